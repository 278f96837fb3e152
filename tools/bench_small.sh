#!/usr/bin/env bash
# One bench.py line per small BASELINE config (2 sparse / dense, 3 SEMI / ANTI / COUNT, 5) into gpurun_out/<tag>_<workload>.json.log,
# with ms per step and the in-run parity printed.  usage: tools/bench_small.sh <tag> [extra bench args]
TAG="${1:-rXX}"; shift || true
mkdir -p gpurun_out
for wl in cfg2_sparse_1Mx1M_1chrom cfg2_dense_1Mx1M_1chrom cfg3_semi_1Mx10M_24chrom cfg3_anti_1Mx10M_24chrom cfg3_count_1Mx10M_24chrom cfg5_nearest_10Mx10M_24chrom; do
  timeout -k 10 400 python3 bench.py --workload "$wl" --steps 10 --warmup 3 "$@" > "gpurun_out/${TAG}_${wl}.json.log" 2> "gpurun_out/${TAG}_${wl}.err"
  tail -n 1 "gpurun_out/${TAG}_${wl}.json.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-32s %.3f ms/step' % (d['config']['workload'], d['ms_per_step']), (d.get('cpu_baseline') or {}).get('parity'))"
done
