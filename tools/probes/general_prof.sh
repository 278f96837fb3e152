#!/usr/bin/env bash
# rocprofv3 kernel stats of the one-call general form at 10M x 100M (tools/probes/general_time.py).
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
OUT="${REPO}/gpurun_out/prof_general"; rm -rf "${OUT}"; mkdir -p "${OUT}"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "${OUT}" -o t -- python3 "${REPO}/tools/probes/general_time.py" > "${OUT}/run.log" 2>&1
tail -n 1 "${OUT}/run.log"
f=$(find "${OUT}" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("  %-84s calls %4s avg %9.1f us" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
