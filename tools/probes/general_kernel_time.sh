#!/usr/bin/env bash
# kernel time of k_bucket_sort<3, 3> under rocprofv3 for library variants (build/<name>.so). usage: general_kernel_time.sh name...
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  OUT="${REPO}/gpurun_out/prof_g_${n}"; rm -rf "${OUT}"; mkdir -p "${OUT}"
  GIQL_HIP_LIB="${REPO}/build/${n}.so" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "${OUT}" -o t -- python3 "${REPO}/tools/probes/general_time.py" > "${OUT}/run.log" 2>&1
  f=$(find "${OUT}" -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$n" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_bucket_sort<3, 3>" in r["Name"]:
        print("%-10s k_bucket_sort<3,3>: calls %s avg %.1f us" % (sys.argv[2], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
