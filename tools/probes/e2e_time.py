import sys, time
sys.path.insert(0, '.')
import bench
from giql_amd.engine import HipEngine
_op, ha, hb, n_chrom = bench.make_inputs("cfg4_10Mx100M_24chrom")
eng = HipEngine(0)
for it in range(3):
    ms, n = eng.inner_join_host_timed(ha, hb, n_chrom)
    print("call", it, "ms", round(ms, 1), "pairs", n, flush=True)
