#!/usr/bin/env python3
"""Probe: k_fill runs in a ~0.80 ms or a ~0.88 ms mode from process to process.  Does the
placement of the OUTPUT buffer decide it?  Time the fill into several distinct buffers."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd import synth
from giql_amd.engine import DeviceSide, HipEngine
eng = HipEngine(0)
a = DeviceSide.from_numpy(*synth.make_table(10_000_000, 5, "peaks"))
b = DeviceSide.from_numpy(*synth.make_table(100_000_000, 6, "reads"))
n = eng.inner_plan(a, b, 24)
bufs = [torch.empty((2, n + 4096), dtype=torch.int32, device="cuda") for _ in range(6)]
eng.set_profiling(True)
for rnd in range(2):
    for k, buf in enumerate(bufs):
        ts = []
        for _ in range(3):
            eng.inner_plan(a, b, 24)
            eng.inner_fill(buf[0, :n], buf[1, :n])
            ts.append(round(eng.stats()["phase_ms"]["fill"], 3))
        print(json.dumps({"round": rnd, "buf": k, "ptr_mod_2MiB": buf.data_ptr() % (2 << 20), "fill_ms": ts}), flush=True)
