#!/usr/bin/env python3
"""Randomized differential soak of the operators tools/soak.py does not reach -- NEAREST k > 1, group_rows, CLUSTER, MERGE,
the table index, the host-buffer INNER join in both download modes -- against the C oracle, on three contexts that live
for the whole run (default; forced three-stage sort; forced 8,192-key buckets), so that the contexts' remembered plans
(two-sort NEAREST, density, layout) cross from case to case.  usage: soak_ops.py [seconds] [seed]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd import _lib
from giql_amd.engine import DeviceSide, HipEngine
from oracle import pyoracle as ora

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
rng = np.random.default_rng(seed)
engines = {"default": HipEngine(0)}
os.environ["GIQL_HIP_LOCAL_MIN_ROWS"] = "1"
engines["local"] = HipEngine(0)
os.environ["GIQL_HIP_LOCAL_BITS"] = "13"
engines["narrow"] = HipEngine(0)
del os.environ["GIQL_HIP_LOCAL_MIN_ROWS"], os.environ["GIQL_HIP_LOCAL_BITS"]


def side(n, nch, span, fixed, piled):
    ch = rng.integers(0, nch, n).astype(np.int32)
    st = rng.integers(0, span, n).astype(np.int32)
    if piled and n:   # long runs of equal starts: the two-sort plans
        st = (rng.integers(0, max(2, n // 300), n) * max(1, span // max(2, n // 300))).astype(np.int32)
    ln = np.full(n, fixed, np.int32) if fixed else rng.integers(1, int(rng.integers(2, 3000)), n).astype(np.int32)
    return ora.Side(ch, st, st + ln)


def dev(s):
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.int32)).to("cuda:0")
    return DeviceSide(t(s.chrom), t(s.start), t(s.end), s.start_off, s.end_off)


def rows_equal(b, got, want):
    ok = want >= 0
    return (np.array_equal(got >= 0, ok) and np.array_equal(b.start[got[ok]], b.start[want[ok]])
            and np.array_equal(b.end[got[ok]], b.end[want[ok]]))


t0, it, seen = time.time(), 0, {}
while time.time() - t0 < budget:
    name = str(rng.choice(list(engines)))
    eng = engines[name]
    nch = int(rng.choice([1, 3, 24, 33]))
    na = int(rng.choice([1, 300, 4_000]))
    nb = int(rng.choice([64, 9_000, 120_000, 300_000]))
    span = int(rng.choice([2_000, 300_000, 40_000_000]))
    piled = bool(rng.random() < 0.3)
    a = side(na, nch, span, 0, False)
    b = side(nb, nch, span, int(rng.choice([0, 0, 150])), piled)
    da, db = dev(a), dev(b)
    tag = (name, "piled" if piled else "plain")
    # NEAREST k > 1: distances and the matched rows, in the reference's order (ABS(distance), start, end)
    k = int(rng.choice([2, 3, 5]))
    signed = bool(rng.random() < 0.5)
    gi, gd = eng.nearest_k(da, db, nch, k, signed=signed)
    wi, wd = ora.c_nearest_k(a, b, k, signed=signed)
    gi, gd = gi.cpu().numpy(), gd.cpu().numpy()
    assert np.array_equal(gd, wd) and rows_equal(b, gi, wi), ("nearest_k", it, tag, k, signed, nch, na, nb, span)
    # group_rows: as many groups as distinct (chrom, start, end), rows of a group identical
    gid, rep = eng.group_rows(db, nch)
    gid, rep = gid.cpu().numpy(), rep.cpu().numpy()
    key = (b.chrom.astype(np.int64) << 44) ^ (b.start.astype(np.int64) << 22) ^ b.end.astype(np.int64)
    trip = np.stack([b.chrom, b.start, b.end], 1)
    assert rep.shape[0] == np.unique(trip, axis=0).shape[0], ("group count", it, tag)
    assert np.array_equal(trip[rep[gid]], trip), ("group rows", it, tag)
    # CLUSTER / MERGE
    dist = int(rng.choice([0, 25, 1000]))
    assert np.array_equal(eng.cluster(db, nch, dist).cpu().numpy(), ora.c_cluster(b, dist)), ("cluster", it, tag, dist)
    mc, ms, me, mn = (x.cpu().numpy() for x in eng.merge(db, nch, dist))
    wc, ws, we, wn = ora.c_merge(b, dist)
    assert (np.array_equal(mc, wc) and np.array_equal(ms, ws) and np.array_equal(me, we) and np.array_equal(mn, wn)), ("merge", it, tag)
    # the table index (tables it takes) and the host-buffer join in both download modes
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    if nch <= 32 and want.shape[0] < 20_000_000:
        try:
            index = eng.index_create(db, nch)
        except _lib.GiqlHipError as exc:
            assert exc.code == _lib.GIQL_ERR_STATE, exc
            index = None
        if index is not None:
            try:
                ra, rb = eng.inner_join_indexed(da, index)
                assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), want), ("indexed", it, tag)
                seen[("indexed", name)] = seen.get(("indexed", name), 0) + 1
            except _lib.GiqlHipError as exc:
                assert exc.code == _lib.GIQL_ERR_STATE, exc
            finally:
                index.close()
        for mode in ("1", "0"):
            os.environ["GIQL_HIP_E2E_COMPACT"] = mode
            ha, hb = eng.inner_join_host((a.chrom, a.start, a.end), (b.chrom, b.start, b.end), nch)
            assert np.array_equal(ora.sort_pairs(ha, hb), want), ("host", mode, it, tag)
        del os.environ["GIQL_HIP_E2E_COMPACT"]
    seen[tag] = seen.get(tag, 0) + 1
    it += 1
    if it % 10 == 0:
        print(json.dumps({"iterations": it, "elapsed_s": round(time.time() - t0, 1)}), flush=True)
print(json.dumps({"ok": True, "iterations": it, "seed": seed, "cases": {" / ".join(map(str, k)): v for k, v in seen.items()}}))
