#!/usr/bin/env bash
# Collect the round's profiles on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a 1-step run
# Raw outputs land under gpurun_out/prof_<tag>/; tools/pmc_summary.py turns them into
# the CSV / JSON kept under profiles/.
set -uo pipefail
TAG="${1:-r01}"
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${REPO}/gpurun_out/prof_${TAG}"
mkdir -p "${OUT}"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "${OUT}/trace" -o trace -- python3 "${REPO}/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "${OUT}/trace_bench.log" 2>&1
echo "trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc "${c}" --kernel-trace --output-format csv -d "${OUT}/pmc_${c}" -o pmc -- python3 "${REPO}/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "${OUT}/pmc_${c}.log" 2>&1
  echo "pmc ${c} rc=$?"
done
find "${OUT}" -name "*.csv" | head -20
du -sh "${OUT}"
