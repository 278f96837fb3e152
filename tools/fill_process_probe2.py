#!/usr/bin/env python3
"""Probe 2: reproduce the slow fill mode (a context re-created behind a 30 GB allocation, filling
an OLDER output buffer) and find which array pairing carries it."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GIQL_HIP_DEBUG_ADDR"] = "1"
import torch
from giql_amd import synth
from giql_amd.engine import DeviceSide, HipEngine

a = DeviceSide.from_numpy(*synth.make_table(10_000_000, 5, "peaks"))
b = DeviceSide.from_numpy(*synth.make_table(100_000_000, 6, "reads"))


def fill_ms(eng, ra, rb, reps=4):
    ts = []
    for _ in range(reps):
        eng.inner_plan(a, b, 24)
        eng.inner_fill(ra, rb)
        ts.append(round(eng.stats()["phase_ms"]["fill"], 3))
    return ts[1:]


def report(tag, eng, ra, rb):
    print(json.dumps({"case": tag, "ra": hex(ra.data_ptr()), "rb": hex(rb.data_ptr()), "fill_ms": fill_ms(eng, ra, rb)}), flush=True)


eng = HipEngine(0)
eng.set_profiling(True)
n = eng.inner_plan(a, b, 24)
cap = int(n * 1.05) + 1024
old = torch.empty((2, cap), dtype=torch.int32, device="cuda")
report("old out", eng, old[0, :n], old[1, :n])
big = torch.empty(30 << 30, dtype=torch.uint8, device="cuda")
report("old out, 30 GB block allocated, same context", eng, old[0, :n], old[1, :n])
eng.close()
eng = HipEngine(0)
eng.set_profiling(True)
eng.inner_plan(a, b, 24)
report("new context, old out", eng, old[0, :n], old[1, :n])
time.sleep(1.0)
report("new context, old out, after 1 s", eng, old[0, :n], old[1, :n])
new = torch.empty((2, cap), dtype=torch.int32, device="cuda")
report("new context, new out", eng, new[0, :n], new[1, :n])
report("new context, old out again", eng, old[0, :n], old[1, :n])
report("row_a old / row_b new", eng, old[0, :n], new[1, :n])
report("row_a new / row_b old", eng, new[0, :n], old[1, :n])
report("old rows swapped", eng, old[1, :n], old[0, :n])
report("old out shifted by 1 MiB", eng, old[0, 262144:262144 + n], old[1, 262144:262144 + n])
del big
torch.cuda.empty_cache()
report("30 GB block freed, old out", eng, old[0, :n], old[1, :n])
eng.close()
eng = HipEngine(0)
eng.set_profiling(True)
eng.inner_plan(a, b, 24)
report("third context (block freed), old out", eng, old[0, :n], old[1, :n])
report("third context, new out", eng, new[0, :n], new[1, :n])
