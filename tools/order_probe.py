#!/usr/bin/env python3
"""Probe: the INNER join with the sides in both orders (A small / A large), each form.
usage: python tools/order_probe.py [workload]   (GIQL_HIP_NO_UNIFORM=1 forces the general form)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from giql_amd.engine import DeviceSide, HipEngine

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4_10Mx100M_24chrom"
_op, ha, hb, n_chrom = bench.make_inputs(wl)
a = DeviceSide.from_numpy(*ha)
b = DeviceSide.from_numpy(*hb)
eng = HipEngine(0)
for name, (x, y) in (("A small", (a, b)), ("A large", (b, a)), ("A small", (a, b)), ("A large", (b, a))):
    out = None
    ts = []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if out is None:
            out = eng.inner_join(x, y, n_chrom)
            n = out[0].shape[0]
            out = (torch.empty(n + 4096, dtype=torch.int32, device="cuda"), torch.empty(n + 4096, dtype=torch.int32, device="cuda"))
        else:
            n = eng.inner_join_into(x, y, n_chrom, out[0], out[1])
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    st = eng.stats()
    print(f"{wl} {name}: n={n} ms={min(ts[2:]):.3f} form={st.get('join_form')} swapped={st.get('swapped')}", flush=True)
