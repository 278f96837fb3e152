#!/usr/bin/env bash
# Host + device timeline of ONE step of a workload: HIP runtime calls and kernels on one clock.
# usage: tools/trace_host.sh <workload> [extra bench args]
WL="$1"; shift
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${REPO}/gpurun_out/htrace_${WL}"
rm -rf "${OUT}"; mkdir -p "${OUT}"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d "${OUT}" -o t -- python3 "${REPO}/bench.py" --workload "${WL}" --steps 3 --warmup 2 --no-cpu-baseline --no-extras "$@" > "${OUT}/bench.log" 2>&1
ls "${OUT}"
python3 - "${OUT}" <<'PY'
import csv, sys, re, glob, os
out = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"giql::(k_[a-z0-9_]+)", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU  " + (m.group(1) if m else r["Kernel_Name"][:30])))
for f in glob.glob(os.path.join(out, "**", "*hip_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host " + r["Function"]))
ev.sort()
idx = [i for i, e in enumerate(ev) if "k_init_minmax" in e[2]]
if not idx:
    print("no kernels"); sys.exit(0)
# the last step: from the host call that launched the last k_init_minmax
t_k = ev[idx[-1]][0]
start = max(i for i, e in enumerate(ev) if e[2].startswith("host") and e[0] < t_k and "Launch" in e[2])
t0 = ev[start][0]
for s, e, n in ev[start:]:
    print("%9.1f  +%7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
