import sys, time
sys.path.insert(0, "/root/repo")
import torch
from giql_amd import synth
from giql_amd.engine import DeviceSide, HipEngine
eng = HipEngine(0)
a = DeviceSide.from_numpy(*synth.make_table(10_000_000, 5, "peaks"))
b = DeviceSide.from_numpy(*synth.make_table(100_000_000, 6, "reads"))
n = eng.inner_plan(a, b, 24)
out = torch.empty((2, n + 1024), dtype=torch.int32, device="cuda")
eng.inner_fill(out[0, :n], out[1, :n])
def two():
    m = eng.inner_plan(a, b, 24); eng.inner_fill(out[0, :m], out[1, :m])
def one():
    eng.inner_join_into(a, b, 24, out[0], out[1])
for name, fn in (("two-call", two), ("fused", one), ("two-call", two), ("fused", one)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms", "fused_fill =", eng.stats()["fused_fill"])
