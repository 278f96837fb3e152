"""Bucket width by density (round 4, VERDICT r03 "Next round" 5) -- needs a GPU.

The three-stage sort's last stage works on the rows sharing ``key >> W``.  W is 16 for tables of up to
~2,800 rows per 65,536 positions; denser tables take W = 15 / 14 / 13 after THREE global passes (bits 8-15,
16-23, 24-31) so that the average bucket stays within what the LDS stage holds, instead of leaving the form
(``giql_amd/csrc/giql_hip.hip`` ``sort_local_bits``, ``bucket_sort.hip.h``).  Reference semantics: the join
itself, ``src/giql/expanders/intersects_duckdb.py:1283-1330`` -- nothing about the result may depend on W.

``GIQL_HIP_LOCAL_BITS=w`` (with ``GIQL_HIP_LOCAL_MIN_ROWS=1``) forces the width at every size, so that small
inputs exercise every path of the narrow forms: plain sorts of every payload shape, the fused range count, the
join in the bucket stage in both forms, queued buckets, crowded windows, both ends of the key axis.  The last
tests take the width from the density alone, on default contexts.
"""

import numpy as np
import pytest

from oracle import pyoracle as ora
from test_gpu_parity import dev, rand_side, uniform_side
from test_sort_stages import _all_ops, _fused_inner, _inner, _join_into, _plain

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(params=[13, 15])   # (14: the density tests below, tests/test_full_size.py, tests/test_index_gpu.py)
def eng_narrow(request, monkeypatch):
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_LOCAL_MIN_ROWS", "1")
    monkeypatch.setenv("GIQL_HIP_LOCAL_BITS", str(request.param))
    e = HipEngine(0)
    monkeypatch.delenv("GIQL_HIP_LOCAL_MIN_ROWS")
    monkeypatch.delenv("GIQL_HIP_LOCAL_BITS")
    e.bits = request.param
    yield e
    e.close()


def _narrow(st, e):
    assert st["sort_local"] and not st["sort_resorted"] and st["bucket_bits"] == e.bits, st


@pytest.mark.parametrize("na,nb,nch,ms,ml", [
    (1, 1, 1, 10, 5),
    (63, 65, 2, 500, 60),
    (5000, 300_000, 24, 200_000_000, 500),
    (40_000, 700_000, 3, 40_000_000, 400),
    (150_000, 300_000, 1, 1_000_000, 300),      # ~2,500 rows per 8192-key bucket, ~20,000 per 65,536 keys
])
def test_narrow_buckets_every_operator(eng_narrow, na, nb, nch, ms, ml):
    a = rand_side(700 + na, na, nch, ms, ml)
    b = rand_side(800 + nb, nb, nch, ms, ml)
    st = _all_ops(eng_narrow, a, b, nch)
    _narrow(st, eng_narrow)


def test_narrow_buckets_uniform_forms_keygen_and_the_fused_count(eng_narrow):
    a = rand_side(901, 60_000, 6, 30_000_000, 900)
    b = uniform_side(902, 900_000, 6, 30_000_000, 150)
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    for _ in range(3):
        assert np.array_equal(_inner(eng_narrow, a, b, 6), want)
        st = eng_narrow.stats()
        assert st["join_form"] == "uniform_b" and st["count_fused"]
        _narrow(st, eng_narrow)
    assert st["span_hist"]            # sorted from the raw columns: the span pass counted bits 8-15 as well
    assert np.array_equal(_inner(eng_narrow, b, a, 6), ora.sort_pairs(*ora.c_inner(b, a, "sweep")))
    assert eng_narrow.stats()["join_form"] == "uniform_a"
    _all_ops(eng_narrow, a, b, 6, nearest=False)


def test_narrow_buckets_irregular_rows_encodings_and_both_ends_of_the_axis(eng_narrow):
    encs = list(ora.ENCODING_OFFSETS)
    for seed in range(4):
        a = rand_side(1100 + seed, 30_000, 5, 90_000_000, 2000, min_len=-3, enc=encs[seed % 4])
        b = rand_side(1200 + seed, 80_000, 5, 90_000_000, 700, min_len=-3 if seed % 2 else 1, enc=encs[(seed + 1) % 4])
        assert np.array_equal(_inner(eng_narrow, a, b, 5), ora.sort_pairs(*ora.c_inner(a, b, "sweep")))
        assert np.array_equal(eng_narrow.count_overlaps(dev(a), dev(b), 5).cpu().numpy(), ora.c_count(a, b, "sweep"))
    top = 2_147_483_000
    w = 1 << eng_narrow.bits
    s = np.array([0, 1, w - 1, w, 65535, 65536, top - 50, top - 10, top - 10], np.int32)
    a = ora.Side(np.zeros(s.size, np.int32), s, s + np.int32(40))
    b = ora.Side(np.zeros(s.size, np.int32), s[::-1].copy(), s[::-1] + np.int32(25))
    _all_ops(eng_narrow, a, b, 1)
    # fixed-length rows at the bucket boundaries, queries that end exactly there
    bs = np.concatenate([np.arange(0, 40 * w, w), np.arange(w - 1, 40 * w, w), np.arange(1, 40 * w, w)]).astype(np.int32)
    u = ora.Side(np.zeros(bs.size, np.int32), bs, bs + np.int32(w // 2))
    qs = np.concatenate([np.arange(0, 40 * w, w // 2), np.arange(3, 40 * w, w)]).astype(np.int32)
    q = ora.Side(np.zeros(qs.size, np.int32), qs, qs + np.int32(w))
    for _ in range(3):
        _fused_inner(eng_narrow, q, u, 1)
    _join_into(eng_narrow, q, u, 1)


def test_narrow_buckets_join_in_the_bucket_stage_both_orders(eng_narrow):
    reads = uniform_side(1601, 400_000, 5, 30_000_000, 150)
    peaks = rand_side(1602, 60_000, 5, 30_000_000, 2000, min_len=200)
    _fused_inner(eng_narrow, peaks, reads, 5)
    st = _join_into(eng_narrow, peaks, reads, 5)
    assert st["count_fused"] and st["fused_fill"] and st["join_form"] == "uniform_b"
    _narrow(st, eng_narrow)
    _join_into(eng_narrow, peaks, reads, 5)
    _fused_inner(eng_narrow, reads, peaks, 5)
    st = _join_into(eng_narrow, reads, peaks, 5)
    assert st["swapped"]
    # a fixed length longer than a bucket, queries longer than several
    long_reads = uniform_side(1603, 200_000, 2, 9_000_000, 20_000)
    long_peaks = rand_side(1604, 5_000, 2, 9_000_000, 30_000, min_len=9_000)
    _fused_inner(eng_narrow, long_peaks, long_reads, 2)
    _join_into(eng_narrow, long_peaks, long_reads, 2, expect_join=None)
    _join_into(eng_narrow, long_peaks, long_reads, 2, expect_join=None)


def test_narrow_buckets_queued_buckets_equal_keys_and_crowded_windows(eng_narrow):
    # ~5,700 rows in every 8192-wide window: every bucket goes through the queue at every width
    reads = uniform_side(1631, 700_000, 1, 1_000_000, 150)
    peaks = rand_side(1632, 8_000, 1, 1_000_000, 800)
    _fused_inner(eng_narrow, peaks, reads, 1)
    _join_into(eng_narrow, peaks, reads, 1)
    r = np.random.default_rng(1633)
    st = (r.integers(0, 500, 100_000) * 37).astype(np.int32)
    piled = ora.Side(np.zeros(st.size, np.int32), st, st + np.int32(150))
    q = rand_side(1634, 1_500, 1, 20_000, 400)
    _fused_inner(eng_narrow, q, piled, 1)
    _join_into(eng_narrow, q, piled, 1)
    reads2 = uniform_side(1635, 300_000, 2, 40_000_000, 150)
    qs = np.concatenate([r.integers(0, 40_000_000, 4_000), r.integers(5_000_000, 5_006_000, 6_000),
                         r.integers(9_000_000, 9_006_000, 2_500)]).astype(np.int32)
    ql = r.integers(50, 1_500, qs.size).astype(np.int32)
    crowd = ora.Side(r.integers(0, 2, qs.size).astype(np.int32), qs, qs + ql)
    _fused_inner(eng_narrow, crowd, reads2, 2)
    _join_into(eng_narrow, crowd, reads2, 2)


def test_narrow_buckets_general_join_both_orders_and_queued_buckets(eng_narrow):
    reads = rand_side(1701, 400_000, 5, 30_000_000, 400, min_len=30)
    peaks = rand_side(1702, 60_000, 5, 30_000_000, 2000, min_len=200)
    _plain(eng_narrow, peaks, reads, 5)
    st = _join_into(eng_narrow, peaks, reads, 5)
    assert st["join_form"] == "general" and st["fused_fill"]
    _narrow(st, eng_narrow)
    _join_into(eng_narrow, peaks, reads, 5)
    _plain(eng_narrow, reads, peaks, 5)
    st = _join_into(eng_narrow, reads, peaks, 5)
    assert st["swapped"] and st["join_form"] == "general"
    dense = rand_side(1703, 600_000, 1, 900_000, 300, min_len=20)      # every bucket through the queue
    q = rand_side(1704, 9_000, 1, 900_000, 900, min_len=1)
    _plain(eng_narrow, q, dense, 1)
    _join_into(eng_narrow, q, dense, 1, expect_join=None)
    _join_into(eng_narrow, q, dense, 1, expect_join=None)


@pytest.mark.parametrize("seed", range(1, 6))   # (seed 0 draws 60K long queries over 300K positions: 30 s of oracle)
def test_narrow_buckets_randomized_sweep(eng_narrow, seed):
    r = np.random.default_rng(8800 + seed)
    n_chrom = int(r.choice([1, 3, 24, 40]))
    span = int(r.choice([300_000, 20_000_000, 2_000_000_000 // n_chrom]))
    nq, nu = int(r.choice([700, 9_000, 60_000])), int(r.choice([30_000, 250_000]))
    fixed = int(r.choice([0, 36, 150, 90_000]))
    encs = list(ora.ENCODING_OFFSETS)
    u = (uniform_side(8900 + seed, nu, n_chrom, span, fixed) if fixed
         else rand_side(8900 + seed, nu, n_chrom, span, int(r.choice([60, 3_000])), min_len=1))
    q = rand_side(9000 + seed, nq, n_chrom, span, int(r.choice([50, 4_000, 30_000])), min_len=1, enc=encs[int(r.integers(0, 4))])
    if r.random() < 0.5:
        k = nq // 2
        q.start[:k] = (span // 3 + r.integers(0, 40_000, k)).astype(np.int32)
        q.end[:k] = q.start[:k] + r.integers(1, 2_000, k).astype(np.int32)
    a, b = (q, u) if r.random() < 0.5 else (u, q)
    for _ in range(3):
        _join_into(eng_narrow, a, b, n_chrom, expect_join=None)
    assert np.array_equal(eng_narrow.count_overlaps(dev(a), dev(b), n_chrom).cpu().numpy(), ora.c_count(a, b, "sweep"))


# ---- the width from the density alone (default contexts) ------------------------------------------------------------
@pytest.mark.parametrize("n_u,span,bits", [
    (3_000_000, 60_000_000, 15),     # 3,277 rows per 65,536 keys: past 2,800 -> buckets of 2^15 keys
    (3_000_000, 30_000_000, 14),     # 6,554
    (4_000_000, 12_000_000, 13),     # 21,845 (2,730 per 8,192 keys)
])
def test_the_density_chooses_the_width(n_u, span, bits):
    """No forcing: a dense fixed-length table keeps the three-stage sort, the fused count and the join in the bucket
    stage with a narrower bucket; count, pairs and the per-row operators equal the oracle's."""
    from giql_amd.engine import HipEngine

    reads = uniform_side(4100 + bits, n_u, 1, span, 100)
    peaks = rand_side(4200 + bits, 40_000, 1, span, 600, min_len=50)
    want = ora.sort_pairs(*ora.c_inner(peaks, reads, "sweep"))
    e = HipEngine(0)
    try:
        for k in range(3):
            got = _inner(e, peaks, reads, 1)
            st = e.stats()
            assert got.shape == want.shape and np.array_equal(got, want)
            assert st["sort_local"] and st["count_fused"] and st["bucket_bits"] == bits and not st["sort_resorted"], st
        st = _join_into(e, peaks, reads, 1)
        assert st["bucket_bits"] == bits
        assert np.array_equal(e.count_overlaps(dev(peaks), dev(reads), 1).cpu().numpy(), ora.c_count(peaks, reads, "sweep"))
        assert np.array_equal(e.semi_join(dev(peaks), dev(reads), 1).cpu().numpy(), ora.c_semi_anti(peaks, reads, False))
    finally:
        e.close()


def test_no_narrow_buckets_switch_keeps_the_round_3_behaviour(monkeypatch):
    from giql_amd.engine import HipEngine

    reads = uniform_side(4301, 3_000_000, 1, 30_000_000, 100)
    peaks = rand_side(4302, 40_000, 1, 30_000_000, 600, min_len=50)
    want = ora.sort_pairs(*ora.c_inner(peaks, reads, "sweep"))
    monkeypatch.setenv("GIQL_HIP_NO_NARROW_BUCKETS", "1")
    e = HipEngine(0)
    monkeypatch.delenv("GIQL_HIP_NO_NARROW_BUCKETS")
    try:
        for _ in range(2):
            assert np.array_equal(_inner(e, peaks, reads, 1), want)
            st = e.stats()
            assert not st["sort_local"] and not st["count_fused"], st
    finally:
        e.close()


# ---- the host-buffer entry point with the compact-plan download (VERDICT r03 "Next round" 8) -------------------------
@pytest.mark.parametrize("mode", ["1", None])
def test_host_join_with_the_compact_plan_download(monkeypatch, mode):
    """``giql_hip_inner`` downloads the plan (per-query {id, first match, count} + the sorted
    ids) and expands it with host threads instead of downloading the pairs.  Same pairs as the oracle in both argument
    orders and with every encoding; plans without a compact form (rows of variable length on both sides, irregular
    rows, no pair at all) are filled and downloaded as before.  Reference semantics: intersects_duckdb.py:1283-1330."""
    from giql_amd.engine import HipEngine

    # mode "1": whenever the plan has the compact form; None (the default): when that form is also the smaller download
    # and the result is large -- the last case below; the small ones take the pairs download
    if mode:
        monkeypatch.setenv("GIQL_HIP_E2E_COMPACT", mode)
    monkeypatch.setenv("GIQL_HIP_E2E_THREADS", "5" if mode else "1")   # (one helper: the calling thread expands too)
    e = HipEngine(0)
    try:
        def host(a, b, nch):
            ra, rb = e.inner_join_host((a.chrom, a.start, a.end), (b.chrom, b.start, b.end), nch,
                                       (a.start_off, a.end_off), (b.start_off, b.end_off))
            return ora.sort_pairs(ra, rb)

        reads = uniform_side(5101, 700_000, 5, 30_000_000, 150)
        peaks = rand_side(5102, 150_000, 5, 30_000_000, 2000, min_len=1)
        want = ora.sort_pairs(*ora.c_inner(peaks, reads, "sweep"))
        for _ in range(3):                      # (the second call speculates, queries grouped coarsely: `lo` not monotone)
            assert np.array_equal(host(peaks, reads, 5), want)
        assert np.array_equal(host(reads, peaks, 5), ora.sort_pairs(*ora.c_inner(reads, peaks, "sweep")))
        encs = list(ora.ENCODING_OFFSETS)
        for k, enc in enumerate(encs):
            q = rand_side(5110 + k, 20_000, 3, 4_000_000, 900, min_len=1, enc=enc)
            u = uniform_side(5120 + k, 90_000, 3, 4_000_000, 75)
            u = ora.Side(u.chrom, u.start, u.end, *ora.ENCODING_OFFSETS[encs[(k + 1) % 4]])
            assert np.array_equal(host(q, u, 3), ora.sort_pairs(*ora.c_inner(q, u, "sweep")))
        # no compact form: the general join, irregular rows among fixed-length ones, an empty result, an empty side
        g1 = rand_side(5131, 60_000, 4, 9_000_000, 700, min_len=1)
        g2 = rand_side(5132, 90_000, 4, 9_000_000, 400, min_len=1)
        assert np.array_equal(host(g1, g2, 4), ora.sort_pairs(*ora.c_inner(g1, g2, "sweep")))
        irr = rand_side(5133, 30_000, 4, 9_000_000, 500, min_len=-5)
        assert np.array_equal(host(irr, reads, 5), ora.sort_pairs(*ora.c_inner(irr, reads, "sweep")))
        far = ora.Side(np.full(1000, 4, np.int32), np.arange(1000, dtype=np.int32) + 900_000_000,
                       np.arange(1000, dtype=np.int32) + 900_000_050)
        assert host(far, reads, 5).shape[0] == 0
        none = ora.Side(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
        assert host(none, reads, 5).shape[0] == 0
        if mode:
            return
        # more than one download chunk of sorted ids (16M ids each) and several expansion blocks
        big = uniform_side(5141, 40_000_000, 2, 100_000_000, 100)
        qs = rand_side(5142, 300_000, 2, 100_000_000, 800, min_len=100)
        got = host(qs, big, 2)
        assert np.array_equal(got, ora.sort_pairs(*ora.c_inner(qs, big, "sweep")))
    finally:
        e.close()


def test_narrow_buckets_keep_the_two_key_sorts_stable(eng_narrow):
    """Pile-ups (long runs of equal starts) put NEAREST and group_rows on their two-sort plan: sort by end, then STABLY by
    start.  With three global passes the first sort ends in the OTHER ping-pong buffer, and the owner of the buffers has
    to follow its view (the soak found ties among equal starts in input order on a context forced to 8,192-key buckets
    that had met pile-ups before: `adopt_by_end`)."""
    rng = np.random.default_rng(77)
    n = 120_000
    st = (rng.integers(0, 400, n) * 50_000).astype(np.int32)        # 400 distinct starts, ~300 rows each
    b = ora.Side(np.zeros(n, np.int32), st, st + rng.integers(1, 3000, n).astype(np.int32))
    qs = rng.integers(0, 20_000_000, 50_000).astype(np.int32)
    a = ora.Side(np.zeros(50_000, np.int32), qs, qs + rng.integers(1, 500, 50_000).astype(np.int32))
    for _ in range(2):  # the first call discovers the pile-ups and switches plans
        idx, dist = eng_narrow.nearest(dev(a), dev(b), 1)
        oi, od = ora.c_nearest_k1(a, b, method="sweep")
        assert np.array_equal(dist.cpu().numpy(), od)
        j = idx.cpu().numpy()
        assert np.array_equal(b.start[j], b.start[oi]) and np.array_equal(b.end[j], b.end[oi])
    gid, rep = eng_narrow.group_rows(dev(b), 1)
    assert rep.shape[0] == len({(int(x), int(y)) for x, y in zip(b.start, b.end)})
    assert eng_narrow.stats()["sort_local"]
    # the context stays on the two-sort plan: a table WITHOUT long runs (many rows per start all the same: ~4 rows on each
    # of 66,000 positions, 33 chromosomes -- the soak's case), every overlapping tie decided by the end
    r = np.random.default_rng(154)
    nb2 = 300_000
    b2 = ora.Side(r.integers(0, 33, nb2).astype(np.int32), r.integers(0, 2_000, nb2).astype(np.int32), np.zeros(nb2, np.int32))
    b2.end[:] = b2.start + r.integers(1, 2_600, nb2).astype(np.int32)
    a2s = r.integers(0, 2_000, 500).astype(np.int32)
    a2 = ora.Side(r.integers(0, 33, 500).astype(np.int32), a2s, a2s + r.integers(1, 100, 500).astype(np.int32))
    for signed in (False, True):
        idx, dist = eng_narrow.nearest(dev(a2), dev(b2), 33, signed=signed)
        oi, od = ora.c_nearest_k1(a2, b2, signed=signed)
        assert np.array_equal(dist.cpu().numpy(), od)
        j = idx.cpu().numpy()
        hit = j >= 0
        assert np.array_equal(hit, oi >= 0)
        assert np.array_equal(b2.start[j[hit]], b2.start[oi[hit]]) and np.array_equal(b2.end[j[hit]], b2.end[oi[hit]])
    # NEAREST k = 3 on the same context (its (end, start) view is a third stable sort)
    k_idx, k_dist = eng_narrow.nearest_k(dev(a2), dev(b2), 33, 3)
    want_i, want_d = ora.c_nearest_k(a2, b2, 3)
    assert np.array_equal(k_dist.cpu().numpy(), want_d)
    ki = k_idx.cpu().numpy()
    ok = want_i >= 0
    assert np.array_equal(ki >= 0, ok)
    assert np.array_equal(b2.start[ki[ok]], b2.start[want_i[ok]]) and np.array_equal(b2.end[ki[ok]], b2.end[want_i[ok]])
