#!/usr/bin/env bash
# rocprofv3 kernel trace + stats of one bench.py command on the GPU box; prints the per-kernel table.
# usage: tools/prof_stats.sh <tag> [bench args...]
set -uo pipefail
TAG="${1:-x}"; shift || true
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${REPO}/gpurun_out/prof_${TAG}"
mkdir -p "${OUT}"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "${OUT}/trace" -o trace -- python3 "${REPO}/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras "$@" > "${OUT}/trace_bench.log" 2>&1
echo "trace rc=$?"
tail -n 1 "${OUT}/trace_bench.log" | cut -c1-600
f=$(find "${OUT}/trace" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-90s calls %5s avg %9.1f us total %9.3f ms %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
cp "$f" "${REPO}/gpurun_out/${TAG}_kernel_stats.csv"
