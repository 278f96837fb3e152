"""Replay a case dumped by tools/soak.py (gpurun_out/soak_fail.npz) on fresh contexts: default, forced three-stage
sort, and every forced bucket width -- NEAREST k = 1 against the oracle, ties included."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd.engine import DeviceSide, HipEngine
from oracle import pyoracle as ora

z = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "soak_fail.npz"))
o = z["offs"]
a = ora.Side(z["ac"], z["as_"], z["ae"], int(o[0]), int(o[1]))
b = ora.Side(z["bc"], z["bs"], z["be"], int(o[2]), int(o[3]))
nch, signed = int(z["nch"]), bool(z["signed"])
wi, wd = ora.c_nearest_k1(a, b, signed=signed)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.int32)).to("cuda:0")
da = DeviceSide(t(a.chrom), t(a.start), t(a.end), a.start_off, a.end_off)
db = DeviceSide(t(b.chrom), t(b.start), t(b.end), b.start_off, b.end_off)
for env in ({}, {"GIQL_HIP_LOCAL_MIN_ROWS": "1"}, {"GIQL_HIP_LOCAL_MIN_ROWS": "1", "GIQL_HIP_LOCAL_BITS": "15"},
            {"GIQL_HIP_LOCAL_MIN_ROWS": "1", "GIQL_HIP_LOCAL_BITS": "14"}, {"GIQL_HIP_LOCAL_MIN_ROWS": "1", "GIQL_HIP_LOCAL_BITS": "13"}):
    os.environ.update(env)
    e = HipEngine(0)
    for k in env:
        del os.environ[k]
    for call in range(3):
        gi, gd = e.nearest(da, db, nch, signed=signed)
        gi, gd = gi.cpu().numpy(), gd.cpu().numpy()
        hit = gi >= 0
        ok_d = np.array_equal(gd, wd) and np.array_equal(hit, wi >= 0)
        ok_r = np.array_equal(b.start[gi[hit]], b.start[wi[hit]]) and np.array_equal(b.end[gi[hit]], b.end[wi[hit]])
        st = e.stats()
        print(env, "call", call, "distances", ok_d, "rows", ok_r, "sort_local", st["sort_local"], "bits", st["bucket_bits"], "resorted", st["sort_resorted"], flush=True)
    # the same tables through the other operators
    print("   count", np.array_equal(e.count_overlaps(da, db, nch).cpu().numpy(), ora.c_count(a, b, "sweep")),
          "semi", np.array_equal(e.semi_join(da, db, nch).cpu().numpy(), ora.c_semi_anti(a, b, False)), flush=True)
    e.close()
