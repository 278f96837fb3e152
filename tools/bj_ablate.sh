#!/usr/bin/env bash
# Stage times of the join-in-the-bucket-stage kernel: rocprofv3 kernel stats of the headline bench with the
# timing-only variants built by tools/build_variant.sh (bj_ab1..3 = stop after the sort / the ranks / the staging).
# usage: tools/bj_ablate.sh name1 name2 ...
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  OUT="${REPO}/gpurun_out/prof_bj_${n}"
  rm -rf "${OUT}"; mkdir -p "${OUT}"
  GIQL_HIP_LIB="${REPO}/build/${n}.so" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "${OUT}" -o t -- python3 "${REPO}/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "${OUT}/bench.log" 2>&1
  f=$(find "${OUT}" -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$n" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_bucket_sort<1, 2" in r["Name"]:
        print("%-10s k_bucket_sort<1,2>: calls %s avg %.1f us" % (sys.argv[2], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
