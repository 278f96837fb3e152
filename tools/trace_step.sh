#!/usr/bin/env bash
# Kernel timeline of ONE step of a workload (start-relative us, duration, gap to the previous kernel).
# usage: tools/trace_step.sh <workload> [extra bench args]
WL="$1"; shift
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${REPO}/gpurun_out/trace_${WL}"
rm -rf "${OUT}"; mkdir -p "${OUT}"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace -d "${OUT}" -o t -- python3 "${REPO}/bench.py" --workload "${WL}" --steps 3 --warmup 2 --no-cpu-baseline --no-extras "$@" > "${OUT}/bench.log" 2>&1
python3 - "${OUT}/t_results.db" <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
try:
    rows += [("MEMCPY " + str(r[0]), r[1], r[2]) for r in db.execute("select name, start, end from memory_copies")]
except Exception as e:
    print("no memcpy table", e)
rows.sort(key=lambda r: r[1])
# last step = after the last k_init_minmax
idx = [i for i, r in enumerate(rows) if "k_init_minmax" in r[0]]
seg = rows[idx[-1]:]
t0 = seg[0][1]
prev = None
tot = 0
for name, s, e in seg:
    m = re.search(r"giql::(k_[a-z0-9_]+(<[^>]*>)?)", name)
    nm = m.group(1) if m else name[:40]
    gap = 0 if prev is None else (s - prev) / 1e3
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, nm))
    prev = e
    tot += (e - s) / 1e3
print("span %.1f us, kernel time %.1f us, %d launches" % ((seg[-1][2] - t0) / 1e3, tot, len(seg)))
PY
