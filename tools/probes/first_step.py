#!/usr/bin/env python3
"""Which part of bench.py's row-operator prologue makes the first timed call slow: variants of
(profiled warm-up, stats read, events off, sync) before ten individually timed calls.  usage: first_step.py [op]"""
import os
import sys
import time

ROOT = os.environ.get("GIQL_TREE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch

from giql_amd import synth
from giql_amd.engine import DeviceSide, HipEngine

op = sys.argv[1] if len(sys.argv) > 1 else "count"
ha, hb = synth.make_table(1_000_000, 1, "peaks"), synth.make_table(10_000_000, 2, "reads")
a, b = DeviceSide.from_numpy(*ha), DeviceSide.from_numpy(*hb)
for variant in ("plain", "profiled warm-up", "profiled + stats", "profiled + stats + gc"):
    eng = HipEngine(0)
    fn = {"semi": lambda: eng.semi_join(a, b, 24), "count": lambda: eng.count_overlaps(a, b, 24)}[op]
    if variant != "plain":
        eng.set_profiling(True)
    for _ in range(3):
        res = fn()
    torch.cuda.synchronize()
    if "stats" in variant:
        st = eng.stats()
    if variant != "plain":
        eng.set_profiling(False)
    if "gc" in variant:
        import gc
        gc.collect()
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t = time.perf_counter()
        res = fn()
        ts.append((time.perf_counter() - t) * 1e3)
    print(f"{variant:28s}", " ".join(f"{x:.3f}" for x in ts), flush=True)
    eng.close()
