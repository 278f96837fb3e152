#!/usr/bin/env bash
# rocprofv3 kernel stats of NEAREST k = 2 and k = 8 at 10M x 10M (tools/probes/nearest_k_time.py's loop, one k at a time).
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for k in "$@"; do
  OUT="${REPO}/gpurun_out/prof_nk${k}"; rm -rf "${OUT}"; mkdir -p "${OUT}"
  cat > /tmp/nk.py <<PY
import sys, time
sys.path.insert(0, '${REPO}')
import torch, bench
from giql_amd.engine import DeviceSide, HipEngine
_op, ha, hb, n_chrom = bench.make_inputs("cfg5_nearest_10Mx10M_24chrom")
a, b = DeviceSide.from_numpy(*ha), DeviceSide.from_numpy(*hb)
eng = HipEngine(0)
ts = []
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = eng.nearest_k(a, b, n_chrom, ${k})
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("k", ${k}, "ms", round(min(ts[1:]), 3), flush=True)
PY
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "${OUT}" -o t -- python3 /tmp/nk.py > "${OUT}/run.log" 2>&1
  tail -n 1 "${OUT}/run.log"
  f=$(find "${OUT}" -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("  %-80s calls %4s avg %9.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
