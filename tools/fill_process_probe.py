#!/usr/bin/env python3
"""Probe: k_fill runs in a ~0.73 ms or a ~0.80 ms mode from PROCESS to process (same box, same
code, same virtual layout).  What inside a process flips it?  Times the fill (hipEvents, 4 runs
each) after re-allocating the output buffer, after shifting the allocation history with dummy
buffers, and after re-creating the context (a new arena)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GIQL_HIP_DEBUG_ADDR"] = "1"
import torch
from giql_amd import synth
from giql_amd.engine import DeviceSide, HipEngine

a = DeviceSide.from_numpy(*synth.make_table(10_000_000, 5, "peaks"))
b = DeviceSide.from_numpy(*synth.make_table(100_000_000, 6, "reads"))


def fill_ms(eng, out, n, reps=4):
    ts = []
    for _ in range(reps):
        eng.inner_plan(a, b, 24)
        eng.inner_fill(out[0, :n], out[1, :n])
        ts.append(round(eng.stats()["phase_ms"]["fill"], 3))
    return ts[1:]


def report(tag, eng, out, n):
    print(json.dumps({"case": tag, "out_ptr": hex(out.data_ptr()), "out_mod_2MiB": out.data_ptr() % (2 << 20),
                      "fill_ms": fill_ms(eng, out, n)}), flush=True)


eng = HipEngine(0)
eng.set_profiling(True)
n = eng.inner_plan(a, b, 24)
cap = int(n * 1.05) + 1024
out = torch.empty((2, cap), dtype=torch.int32, device="cuda")
report("first", eng, out, n)
del out
torch.cuda.empty_cache()
out = torch.empty((2, cap), dtype=torch.int32, device="cuda")
report("out re-allocated", eng, out, n)
dummies = [torch.empty(int(x * (1 << 20)), dtype=torch.uint8, device="cuda") for x in (1, 37, 513, 2049)]
out2 = torch.empty((2, cap), dtype=torch.int32, device="cuda")
report("second out behind dummies", eng, out2, n)
report("first out again", eng, out, n)
eng.close()
eng = HipEngine(0)
eng.set_profiling(True)
eng.inner_plan(a, b, 24)
report("new context (new arena), first out", eng, out, n)
report("new context, second out", eng, out2, n)
big = torch.empty(30 << 30, dtype=torch.uint8, device="cuda")  # push later allocations elsewhere
eng.close()
eng = HipEngine(0)
eng.set_profiling(True)
eng.inner_plan(a, b, 24)
report("third context behind a 30 GB block", eng, out, n)
out3 = torch.empty((2, cap), dtype=torch.int32, device="cuda")
report("third context, third out", eng, out3, n)
