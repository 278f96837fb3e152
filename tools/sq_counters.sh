#!/usr/bin/env bash
# Where do the waves of the headline's kernels spend their cycles?  One rocprofv3 --pmc pass per counter group (SQ block:
# at most 8 counters per pass) over a 1-step run; per kernel: WAVE_CYCLES split into parked (s_waitcnt / barrier), issue
# stalls and active instructions, + the vector / scalar / LDS / memory instruction shares.  usage: tools/sq_counters.sh <tag>
set -uo pipefail
TAG="${1:-rXX}"
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${REPO}/gpurun_out/sq_${TAG}"
mkdir -p "${OUT}"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --pmc ${grp} --kernel-trace --output-format csv -d "${OUT}/g${i}" -o pmc -- python3 "${REPO}/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extras > "${OUT}/g${i}.log" 2>&1
  echo "group ${i} rc=$?"
done
python3 - "${OUT}" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/g*/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"giql::(k_[a-z0-9_]+)", r.get("Kernel_Name", ""))
        if not m:
            continue
        acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(k)
    print("   of WAVE_CYCLES: parked %.0f %%  issue-stalled %.0f %% (LDS %.0f %%)  active %.0f %%" % (
        100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_LDS", 0) / wc,
        100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc))
    print("   active by kind (share of WAVE_CYCLES): VALU %.0f %%  scalar %.0f %%  LDS %.0f %%  VMEM %.0f %%  flat %.0f %%  misc %.0f %%" % tuple(
        100 * c.get(n, 0) / wc for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC")))
    print("   instructions per wave: VALU %.0f  SALU %.0f  LDS %.0f  VMEM rd %.0f wr %.0f  SMEM %.0f  (waves %.0f)" % tuple(
        [c.get(n, 0) / (c.get("SQ_WAVES", 0) or 1) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")] + [c.get("SQ_WAVES", 0)]))
PY
