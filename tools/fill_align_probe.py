#!/usr/bin/env python3
"""Probe: does the relative placement of the row_a / row_b output arrays change the
fill time?  (k_fill streams both at once; fill time varies 0.78-0.92 ms between boxes.)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd import synth
from giql_amd.engine import DeviceSide, HipEngine
eng = HipEngine(0)
a = DeviceSide.from_numpy(*synth.make_table(10_000_000, 5, "peaks"))
b = DeviceSide.from_numpy(*synth.make_table(100_000_000, 6, "reads"))
n = eng.inner_plan(a, b, 24)
big = torch.empty(2 * n + (64 << 20), dtype=torch.int32, device="cuda")
base_ptr = big.data_ptr()
eng.set_profiling(True)
for gap_elems in (0, 256, 1024, 4096, 1 << 14, 1 << 16, (1 << 18) + 512, 1 << 20, (1 << 20) + 1024, (1 << 22) + 4096 + 128):
    ra = big[:n]
    rb = big[n + gap_elems: 2 * n + gap_elems]
    ts = []
    for _ in range(4):
        eng.inner_plan(a, b, 24)
        eng.inner_fill(ra, rb)
        ts.append(eng.stats()["phase_ms"]["fill"])
    print(json.dumps({"gap_bytes": gap_elems * 4, "rb_minus_ra_mod_1MiB": ((n + gap_elems) * 4) % (1 << 20), "fill_ms": [round(t, 3) for t in ts[1:]]}), flush=True)
