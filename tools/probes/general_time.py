import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from giql_amd.engine import DeviceSide, HipEngine
_op, ha, hb, n_chrom = bench.make_inputs("cfg4_10Mx100M_24chrom")
a, b = DeviceSide.from_numpy(*ha), DeviceSide.from_numpy(*hb)
os.environ["GIQL_HIP_NO_UNIFORM"] = "1"
g = HipEngine(0)
cap = 404376266 + 4096
ra = torch.empty(cap, dtype=torch.int32, device="cuda:0"); rb = torch.empty_like(ra)
n = g.inner_plan(a, b, n_chrom); g.inner_fill(ra[:n], rb[:n])
for _ in range(2): g.inner_join_into(a, b, n_chrom, ra, rb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): n = g.inner_join_into(a, b, n_chrom, ra, rb)
torch.cuda.synchronize()
print("general one-call ms", round((time.perf_counter() - t0) / 5 * 1e3, 3), "pairs", n, g.stats()["bucket_join"])
