#!/usr/bin/env python3
"""Per-call wall times of one per-row operator at a BASELINE config (where a step's time goes when the mean moves):
usage: rowop_steps.py [semi|anti|count|nearest] [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

from giql_amd import synth
from giql_amd.engine import DeviceSide, HipEngine

op = sys.argv[1] if len(sys.argv) > 1 else "semi"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if op == "nearest":
    ha, hb = synth.make_table(10_000_000, 1, "peaks"), synth.make_table(10_000_000, 2, "peaks")
else:
    ha, hb = synth.make_table(1_000_000, 1, "peaks"), synth.make_table(10_000_000, 2, "reads")
a, b = DeviceSide.from_numpy(*ha), DeviceSide.from_numpy(*hb)
eng = HipEngine(0)
fn = {"semi": lambda: eng.semi_join(a, b, 24), "anti": lambda: eng.anti_join(a, b, 24),
      "count": lambda: eng.count_overlaps(a, b, 24), "nearest": lambda: eng.nearest(a, b, 24)}[op]
for _ in range(3):
    fn()
torch.cuda.synchronize()
ts = []
for _ in range(calls):
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
print(op, "ms per call:", " ".join(f"{x:.3f}" for x in ts))
print("mean %.3f median %.3f max %.3f" % (sum(ts) / len(ts), sorted(ts)[len(ts) // 2], max(ts)))
