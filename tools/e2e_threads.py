"""giql_hip_inner at the headline sizes for several expansion-thread counts (GIQL_HIP_E2E_THREADS) and both store kinds
(GIQL_HIP_E2E_NT): wall ms of the settled call.  usage (GPU box): python3 tools/e2e_threads.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from giql_amd import synth  # noqa: E402
from giql_amd.engine import HipEngine  # noqa: E402

ha = synth.make_table(10_000_000, 5, "peaks")
hb = synth.make_table(100_000_000, 6, "reads")
eng = HipEngine(0)
eng.inner_join_host_timed(ha, hb, 24)
for nt in ("1", "0"):
    for thr in (8, 12, 16, 24, 32, 48, 64):
        os.environ["GIQL_HIP_E2E_THREADS"] = str(thr)
        os.environ["GIQL_HIP_E2E_NT"] = nt
        ms = sorted(eng.inner_join_host_timed(ha, hb, 24)[0] for _ in range(4))
        print(f"non-temporal stores {nt}  threads {thr:3d}: best {ms[0]:.1f} ms  median {ms[1]:.1f}", flush=True)
