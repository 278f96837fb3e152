#!/usr/bin/env bash
# End-of-round evidence on the GPU box, one gpurun call: gpu tests, default bench line, rocprofv3 kernel stats + PMC
# traffic, one bench line per small config, the N > 1 rehearsals (SKIP_REHEARSAL=1: not), a soak.  usage: tools/final_evidence.sh <tag> [soak s]
TAG="${1:-rXX}"; SOAK="${2:-150}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "gpurun_out/${TAG}_gputests.log" 2>&1; echo "tests rc=$?"; tail -n 2 "gpurun_out/${TAG}_gputests.log"
timeout -k 10 300 python bench.py > "gpurun_out/${TAG}_bench_default.json.log" 2> "gpurun_out/${TAG}_bench_default.err"; echo "bench rc=$?"
bash tools/profile.sh "${TAG}" | tail -n 3
bash tools/bench_small.sh "${TAG}" 2>&1 | cut -c1-120
timeout -k 10 300 python bench.py --workload cfg4_sorted_10Mx100M_24chrom --steps 10 --warmup 3 > "gpurun_out/${TAG}_cfg4_sorted.json.log" 2>/dev/null; echo "sorted rc=$?"
if [ -z "${SKIP_REHEARSAL:-}" ]; then bash tools/exchange_rehearsal.sh "${TAG}" 2>&1 | grep -v "^$" | cut -c1-300; fi
timeout -k 10 $((SOAK + 120)) python tools/soak.py "${SOAK}" 20261004 2>&1 | tail -n 1 > "gpurun_out/${TAG}_soak.json.log"; cat "gpurun_out/${TAG}_soak.json.log" | cut -c1-600
