#!/usr/bin/env python3
"""Randomized differential soak of the HIP path against the C oracle (GPU box): one context for
the whole run, so every form change (fixed-length / general, aligned / tight layout) also crosses
the contexts' speculation.  usage: soak.py [seconds] [seed]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd.engine import DeviceSide, HipEngine
from oracle import pyoracle as ora

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.default_rng(seed)
eng_default = HipEngine(0)
os.environ["GIQL_HIP_LOCAL_MIN_ROWS"] = "1"   # a second context that takes the three-stage sort at every size
eng_local = HipEngine(0)
os.environ["GIQL_HIP_LOCAL_BITS"] = os.environ.get("SOAK_NARROW_BITS", "13")   # a third one with the narrowest buckets (dense tables' form, round 4)
eng_narrow = HipEngine(0)
del os.environ["GIQL_HIP_LOCAL_MIN_ROWS"], os.environ["GIQL_HIP_LOCAL_BITS"]
ENC = list(ora.ENCODING_OFFSETS.values())


def side(n, nch, span, fixed, irregular):
    ch = rng.integers(0, nch, n).astype(np.int32)
    if rng.random() < 0.3:
        ch = np.sort(ch)
    st = rng.integers(0, span, n).astype(np.int32)
    ln = np.full(n, fixed, np.int32) if fixed else rng.integers(1, int(rng.integers(2, 3000)), n).astype(np.int32)
    en = st + ln
    if irregular and n:
        bad = rng.integers(0, n, max(1, n // 500))
        en[bad] = st[bad] - rng.integers(0, 3, bad.shape[0]).astype(np.int32)
    so, eo = ENC[int(rng.integers(0, 4))] if rng.random() < 0.3 else (0, 0)
    return ora.Side(ch, st, en, so, eo)


def dev(s):
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.int32)).to("cuda:0")
    return DeviceSide(t(s.chrom), t(s.start), t(s.end), s.start_off, s.end_off)


t0, it, forms = time.time(), 0, {}
while time.time() - t0 < budget:
    nch = int(rng.choice([1, 3, 24, 31, 32, 33, 40]))
    na = int(rng.choice([1, 77, 5000, 8192, 40_000, 150_000]))
    nb = int(rng.choice([1, 64, 8193, 70_000, 300_000]))
    span = int(rng.choice([2_000, 300_000, 50_000_000, 2_000_000_000 // max(nch, 1)]))
    fixed_b = int(rng.choice([0, 0, 36, 150]))
    fixed_a = int(rng.choice([0, 0, 0, 75]))
    if na * nb * 3000.0 / (float(span) * nch) > 3e7:  # expected pairs (upper estimate): keep the oracle fast
        continue
    a = side(na, nch, span, fixed_a, rng.random() < 0.25)
    b = side(nb, nch, span, fixed_b, rng.random() < 0.15)
    da, db = dev(a), dev(b)
    pick = rng.random()
    eng = eng_local if pick < 0.25 else (eng_narrow if pick < 0.45 else eng_default)
    try:
        ra, rb = eng.inner_join(da, db, nch)
    except Exception as exc:
        assert "span" in str(exc).lower(), exc   # 33+ chromosomes x 2e9 / n positions may not fit 32 bits
        continue
    st = eng.stats()
    forms[(st["join_form"], st["span_hist"])] = forms.get((st["join_form"], st["span_hist"]), 0) + 1
    got = ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy())
    small = na * nb <= 4_000_000_000
    want = ora.sort_pairs(*ora.c_inner(a, b, "brute" if small and (na * nb < 3e8) else "sweep"))
    assert np.array_equal(got, want), ("inner", it, nch, na, nb, span, fixed_a, fixed_b)
    forms[("swapped", st["swapped"])] = forms.get(("swapped", st["swapped"]), 0) + 1
    # the one-call form into caller-owned buffers (the fill launched inside the plan when the guesses hold)
    cap = want.shape[0] + int(rng.integers(0, 3)) * 1000
    oa = torch.empty(cap, dtype=torch.int32, device="cuda:0")
    ob = torch.empty(cap, dtype=torch.int32, device="cuda:0")
    n = eng.inner_join_into(da, db, nch, oa, ob)
    assert n == want.shape[0] and np.array_equal(ora.sort_pairs(oa[:n].cpu().numpy(), ob[:n].cpu().numpy()), want), ("into", it)
    # ... and once more on the same data: now every guess of the context holds, which is when the pairs come straight
    # from the bucket stage of the sort (round 3; either join form, the LDS body, the crowded and the queued buckets)
    oa.fill_(-1)
    n = eng.inner_join_into(da, db, nch, oa, ob)
    assert n == want.shape[0] and np.array_equal(ora.sort_pairs(oa[:n].cpu().numpy(), ob[:n].cpu().numpy()), want), ("into again", it)
    st2 = eng.stats()
    if st2["bucket_join"]:
        forms[("bucket_join", st2["join_form"])] = forms.get(("bucket_join", st2["join_form"]), 0) + 1
    # NEAREST k = 1 (rows with start <= end only: the operator rejects inverted rows)
    if not (np.any(a.end + a.end_off < a.start + a.start_off) or np.any(b.end + b.end_off < b.start + b.start_off)):
        signed = bool(rng.random() < 0.5)
        gi, gd = eng.nearest(da, db, nch, signed=signed)
        wi, wd = ora.c_nearest_k1(a, b, signed=signed)
        gi, gd = gi.cpu().numpy(), gd.cpu().numpy()
        assert np.array_equal(gd, wd) and np.array_equal(gi >= 0, wi >= 0), ("nearest", it, nch, na, nb, span)
        hit = gi >= 0
        if not (np.array_equal(b.start[gi[hit]], b.start[wi[hit]]) and np.array_equal(b.end[gi[hit]], b.end[wi[hit]])):
            bad = np.nonzero(hit)[0][(b.start[gi[hit]] != b.start[wi[hit]]) | (b.end[gi[hit]] != b.end[wi[hit]])]
            name = "narrow" if eng is eng_narrow else ("local" if eng is eng_local else "default")
            rows = [(int(i), int(gd[i]), (int(a.chrom[i]), int(a.start[i]), int(a.end[i])),
                     "got", (int(gi[i]), int(b.start[gi[i]]), int(b.end[gi[i]])),
                     "want", (int(wi[i]), int(b.start[wi[i]]), int(b.end[wi[i]]))) for i in bad[:6]]
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", "soak_fail.npz"), ac=a.chrom, as_=a.start, ae=a.end,
                                bc=b.chrom, bs=b.start, be=b.end, offs=np.array([a.start_off, a.end_off, b.start_off, b.end_off]),
                                nch=nch, signed=signed)
            raise AssertionError(("nearest rows", it, name, signed, nch, na, nb, span, fixed_a, fixed_b,
                                  (a.start_off, a.end_off), (b.start_off, b.end_off), len(bad), rows, eng.stats()))
    assert np.array_equal(eng.semi_join(da, db, nch).cpu().numpy(), ora.c_semi_anti(a, b, False)), ("semi", it)
    if eng.stats()["coarse_b"]:
        forms[("coarse_b", True)] = forms.get(("coarse_b", True), 0) + 1
    assert np.array_equal(eng.anti_join(da, db, nch).cpu().numpy(), ora.c_semi_anti(a, b, True)), ("anti", it)
    assert np.array_equal(eng.count_overlaps(da, db, nch).cpu().numpy(),
                          ora.c_count(a, b, "brute" if na * nb < 3e8 else "sweep")), ("count", it)
    it += 1
    if it % 10 == 0:
        print(json.dumps({"iterations": it, "elapsed_s": round(time.time() - t0, 1)}), flush=True)
def label(k):
    if k[0] == "swapped":
        return "sides exchanged" if k[1] else "sides as given"
    if k[0] == "bucket_join":
        return f"pairs written by the bucket stage ({k[1]})"
    if k[0] == "coarse_b":
        return "SEMI with B sorted without its lowest digit"
    return f"{k[0]}/{'span_hist' if k[1] else 'linearize'}"


print(json.dumps({"ok": True, "iterations": it, "seed": seed, "forms": {label(k): v for k, v in forms.items()}}))
