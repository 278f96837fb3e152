#!/usr/bin/env bash
# End-of-round evidence on the GPU box, one gpurun call: gpu tests, default bench line, rocprofv3 kernel stats + PMC
# traffic, one bench line per small config, the N > 1 rehearsals (SKIP_REHEARSAL=1: not), a soak.  usage: tools/final_evidence.sh <tag> [soak s]
TAG="${1:-rXX}"; SOAK="${2:-150}"
mkdir -p gpurun_out
# PART=tests: the gpu suite + the soak; PART=bench: every bench line + the profiles; unset: both (two calls fit 1200 s better)
if [ "${PART:-all}" != bench ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "gpurun_out/${TAG}_gputests.log" 2>&1; echo "tests rc=$?"; tail -n 2 "gpurun_out/${TAG}_gputests.log"
fi
if [ "${PART:-all}" != tests ]; then
timeout -k 10 300 python bench.py > "gpurun_out/${TAG}_bench_default.json.log" 2> "gpurun_out/${TAG}_bench_default.err"; echo "bench rc=$?"
bash tools/profile.sh "${TAG}" | tail -n 3
bash tools/bench_small.sh "${TAG}" 2>&1 | cut -c1-120
timeout -k 10 300 python bench.py --workload cfg4_sorted_10Mx100M_24chrom --steps 10 --warmup 3 > "gpurun_out/${TAG}_cfg4_sorted.json.log" 2>/dev/null; echo "sorted rc=$?"
timeout -k 10 300 python bench.py --workload cfg4_indexed_10Mx100M_24chrom --steps 10 --warmup 3 > "gpurun_out/${TAG}_cfg4_indexed.json.log" 2>/dev/null; echo "indexed rc=$?"
timeout -k 10 300 python bench.py --shard-of 8 --shard-rank 0 --steps 10 --warmup 3 > "gpurun_out/${TAG}_shard8_rank0.json.log" 2>/dev/null; echo "shard rc=$?"
timeout -k 10 400 python bench.py --workload dense_40Mx400M_24chrom --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "gpurun_out/${TAG}_dense_40Mx400M.json.log" 2>/dev/null; echo "dense rc=$?"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
if [ -z "${SKIP_REHEARSAL:-}" ]; then bash tools/exchange_rehearsal.sh "${TAG}" 2>&1 | grep -v "^$" | cut -c1-300; fi
fi
if [ "${PART:-all}" != bench ]; then
timeout -k 10 $((SOAK + 120)) python tools/soak.py "${SOAK}" 20261004 2>&1 | tail -n 1 > "gpurun_out/${TAG}_soak.json.log"; cat "gpurun_out/${TAG}_soak.json.log" | cut -c1-600
fi
