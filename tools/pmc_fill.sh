#!/usr/bin/env bash
# FETCH_SIZE / WRITE_SIZE of k_fill in several processes: do the slow-mode processes move more bytes?
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for k in 1 2 3 4; do
  for c in FETCH_SIZE; do
    OUT="${REPO}/gpurun_out/pmcfill_${k}_${c}"
    rm -rf "${OUT}"; mkdir -p "${OUT}"
    timeout -k 10 200 rocprofv3 --pmc "${c}" --kernel-trace -d "${OUT}" -o pmc -- python3 "${REPO}/bench.py" --steps 2 --warmup 2 --no-cpu-baseline --no-extras > "${OUT}/bench.log" 2>&1
    echo "run $k $c rc=$?"
    python3 "${REPO}/tools/pmc_kernel.py" "${OUT}/pmc_results.db" "${c}" "k_fill" | tail -n 6
  done
done
