#!/usr/bin/env bash
# Rehearsal of the per-row operators at N > 1 on ONE GPU: the ranks share the card, the exchange goes
# through host memory (gloo) -- correctness of the sharded path, not a scaling number.
# usage: tools/rowop_ranks.sh <tag> [ranks...]
TAG="${1:-rXX}"; shift
RANKS="${*:-2 3}"
mkdir -p gpurun_out
for wl in cfg3_semi_1Mx10M_24chrom cfg3_anti_1Mx10M_24chrom cfg3_count_1Mx10M_24chrom cfg5_nearest_10Mx10M_24chrom; do
  for n in $RANKS; do
    timeout -k 10 400 python3 bench.py --workload "$wl" --gpus "$n" --backend gloo --steps 3 --warmup 1 --verify \
      > "gpurun_out/${TAG}_${wl}_g${n}.json.log" 2> "gpurun_out/${TAG}_${wl}_g${n}.err"
    echo "$wl g$n rc=$?"
    python3 - "gpurun_out/${TAG}_${wl}_g${n}.json.log" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("  n_gpus", d["n_gpus"], "ms", d["ms_per_step"], "rows_out", d["config"]["rows_out"], "parity", (d.get("cpu_baseline") or {}).get("parity", {}).get("equal"))
except Exception as e:
    print("  no line:", e)
PY
  done
done
