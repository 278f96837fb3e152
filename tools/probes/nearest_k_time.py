import sys, time
sys.path.insert(0, '.')
import torch, bench
from giql_amd.engine import DeviceSide, HipEngine
_op, ha, hb, n_chrom = bench.make_inputs("cfg5_nearest_10Mx10M_24chrom")
a, b = DeviceSide.from_numpy(*ha), DeviceSide.from_numpy(*hb)
eng = HipEngine(0)
for k in (1, 2, 3, 8):
    ts = []
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = eng.nearest_k(a, b, n_chrom, k) if k > 1 else eng.nearest(a, b, n_chrom)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("k", k, "ms", round(min(ts[1:]), 3), flush=True)
