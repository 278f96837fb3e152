#!/usr/bin/env python3
"""Tables of 2-6M rows whose DENSITY changes from case to case on ONE default context: the bucket width (16 / 15 / 14 /
13 bits, or four global passes) is a guess from the previous call's span, so every change of density is a wrong guess
that has to stay exact -- INNER (plan + fill, then the one-call form twice), COUNT, SEMI and NEAREST against the
oracle's sweep.  usage: soak_density.py [seconds] [seed]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd.engine import DeviceSide, HipEngine
from oracle import pyoracle as ora

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 31
rng = np.random.default_rng(seed)
eng = HipEngine(0)


def dev(s):
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.int32)).to("cuda:0")
    return DeviceSide(t(s.chrom), t(s.start), t(s.end), s.start_off, s.end_off)


t0, it, seen = time.time(), 0, {}
while time.time() - t0 < budget:
    nch = int(rng.choice([1, 2, 5]))
    nb = int(rng.choice([2_200_000, 3_000_000, 6_000_000]))
    per_bucket = float(rng.choice([150, 900, 2_500, 4_000, 9_000, 20_000, 60_000]))   # rows per 65,536 positions
    span = max(50_000, int(nb * 65536.0 / per_bucket / nch))
    fixed = int(rng.choice([0, 100, 100]))
    na = int(rng.choice([2_000, 60_000]))
    bc = rng.integers(0, nch, nb).astype(np.int32)
    bs = rng.integers(0, span, nb).astype(np.int32)
    bl = np.full(nb, fixed, np.int32) if fixed else rng.integers(20, 400, nb).astype(np.int32)
    b = ora.Side(bc, bs, bs + bl)
    as_ = rng.integers(0, span, na).astype(np.int32)
    a = ora.Side(rng.integers(0, nch, na).astype(np.int32), as_, as_ + rng.integers(1, 700, na).astype(np.int32))
    da, db = dev(a), dev(b)
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    ra, rb = eng.inner_join(da, db, nch)
    st = eng.stats()
    assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), want), ("inner", it, nch, nb, per_bucket, fixed, st)
    del ra, rb
    for rep in range(2):
        oa = torch.full((want.shape[0] + 64,), -7, dtype=torch.int32, device="cuda:0")
        ob = torch.full_like(oa, -7)
        n = eng.inner_join_into(da, db, nch, oa, ob)
        st = eng.stats()
        assert n == want.shape[0] and np.array_equal(ora.sort_pairs(oa[:n].cpu().numpy(), ob[:n].cpu().numpy()), want), ("into", rep, it, per_bucket, st)
        assert int((oa[n:] != -7).sum()) == 0
    form = (("buckets 2^%d" % st["bucket_bits"]) if st["sort_local"] else "four passes") + (" + bucket join" if st["bucket_join"] else "")
    seen[form] = seen.get(form, 0) + 1
    assert np.array_equal(eng.count_overlaps(da, db, nch).cpu().numpy(), ora.c_count(a, b, "sweep")), ("count", it, per_bucket)
    assert np.array_equal(eng.semi_join(da, db, nch).cpu().numpy(), ora.c_semi_anti(a, b, False)), ("semi", it, per_bucket)
    gi, gd = eng.nearest(da, db, nch)
    wi, wd = ora.c_nearest_k1(a, b, method="sweep")
    gi = gi.cpu().numpy()
    assert np.array_equal(gd.cpu().numpy(), wd) and np.array_equal(b.start[gi], b.start[wi]) and np.array_equal(b.end[gi], b.end[wi]), ("nearest", it, per_bucket)
    it += 1
    print(json.dumps({"iterations": it, "elapsed_s": round(time.time() - t0, 1), "rows_per_65536": per_bucket, "form": form}), flush=True)
print(json.dumps({"ok": True, "iterations": it, "seed": seed, "forms": seen}))
