"""The three-stage sort (two global onesweep passes on key bits 16-31 + the in-LDS bucket sort of
``giql_amd/csrc/bucket_sort.hip.h``) against the oracle -- needs a GPU.

Production contexts take that form for sides of 32M rows and more; here
``GIQL_HIP_LOCAL_MIN_ROWS=1`` forces it at every size so that small inputs exercise it: empty and
one-row buckets, buckets at both ends of the key axis, ties, every payload shape (keys only /
+rid / +end / both), the two-key sorts that rely on stability, and the fall-back to the four-pass
sort when a bucket is too large for LDS.
"""

import numpy as np
import pytest

from oracle import pyoracle as ora
from test_gpu_parity import dev, rand_side, uniform_side

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture()
def eng_local(monkeypatch):
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_LOCAL_MIN_ROWS", "1")
    e = HipEngine(0)
    monkeypatch.delenv("GIQL_HIP_LOCAL_MIN_ROWS")
    yield e
    e.close()


def _inner(e, a, b, nch):
    ra, rb = e.inner_join(dev(a), dev(b), nch)
    return ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy())


def _all_ops(e, a, b, nch, nearest=True):
    assert np.array_equal(_inner(e, a, b, nch), ora.sort_pairs(*ora.c_inner(a, b, "sweep")))
    st = e.stats()
    assert np.array_equal(e.semi_join(dev(a), dev(b), nch).cpu().numpy(), ora.c_semi_anti(a, b, False))
    assert np.array_equal(e.anti_join(dev(a), dev(b), nch).cpu().numpy(), ora.c_semi_anti(a, b, True))
    assert np.array_equal(e.count_overlaps(dev(a), dev(b), nch).cpu().numpy(), ora.c_count(a, b, "sweep"))
    if nearest:
        idx, dist = e.nearest(dev(a), dev(b), nch)
        oi, od = ora.c_nearest_k1(a, b, method="sweep")
        assert np.array_equal(dist.cpu().numpy(), od)
        j = idx.cpu().numpy()
        ok = j >= 0
        assert np.array_equal(ok, oi >= 0)
        assert np.array_equal(b.start[j[ok]], b.start[oi[ok]]) and np.array_equal(b.end[j[ok]], b.end[oi[ok]])
        raw = ora.Side(a.chrom, a.start, a.end)
        assert np.array_equal(e.cluster(dev(raw), nch, 25).cpu().numpy(), ora.c_cluster(raw, 25))
    return st


@pytest.mark.parametrize("na,nb,nch,ms,ml", [
    (1, 1, 1, 10, 5),
    (63, 65, 2, 500, 60),
    (5000, 300_000, 24, 200_000_000, 500),      # ~100 rows per 65536-bp bucket
    (200_000, 150_000, 24, 50_000_000, 3000),
    (40_000, 700_000, 3, 40_000_000, 400),      # ~380 rows per bucket, 3 chromosomes
    (300_000, 300_000, 1, 6_000_000, 300),      # ~3300 rows per bucket: near the LDS capacity
])
def test_three_stage_sort_every_operator(eng_local, na, nb, nch, ms, ml):
    a = rand_side(700 + na, na, nch, ms, ml)
    b = rand_side(800 + nb, nb, nch, ms, ml)
    st = _all_ops(eng_local, a, b, nch)
    assert st["sort_local"] and not st["sort_resorted"]


def test_three_stage_sort_uniform_forms_and_keygen(eng_local):
    # fixed-length reads: the big side is sorted straight from its raw columns (KEYGEN first pass,
    # now on bits 16-23), twice so that the second call runs fully speculated and fused
    a = rand_side(901, 60_000, 6, 30_000_000, 900)
    b = uniform_side(902, 900_000, 6, 30_000_000, 150)
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    for _ in range(3):
        assert np.array_equal(_inner(eng_local, a, b, 6), want)
        st = eng_local.stats()
        assert st["join_form"] == "uniform_b" and st["sort_local"]
    assert st["span_hist"]
    want2 = ora.sort_pairs(*ora.c_inner(b, a, "sweep"))
    assert np.array_equal(_inner(eng_local, b, a, 6), want2)
    assert eng_local.stats()["join_form"] == "uniform_a"
    _all_ops(eng_local, a, b, 6, nearest=False)


def test_three_stage_sort_irregular_rows_encodings_and_extremes(eng_local):
    encs = list(ora.ENCODING_OFFSETS)
    for seed in range(6):
        a = rand_side(1100 + seed, 30_000, 5, 90_000_000, 2000, min_len=-3, enc=encs[seed % 4])
        b = rand_side(1200 + seed, 80_000, 5, 90_000_000, 700, min_len=-3 if seed % 2 else 1, enc=encs[(seed + 1) % 4])
        assert np.array_equal(_inner(eng_local, a, b, 5), ora.sort_pairs(*ora.c_inner(a, b, "sweep")))
        assert np.array_equal(eng_local.count_overlaps(dev(a), dev(b), 5).cpu().numpy(), ora.c_count(a, b, "sweep"))
    # keys in the first and the last bucket of the 32-bit axis
    top = 2_147_483_000
    s = np.array([0, 1, 65535, 65536, top - 50, top - 10, top - 10], np.int32)
    a = ora.Side(np.zeros(7, np.int32), s, s + np.int32(40))
    b = ora.Side(np.zeros(7, np.int32), s[::-1].copy(), s[::-1] + np.int32(25))
    _all_ops(eng_local, a, b, 1)


def test_three_stage_sort_is_stable_for_the_two_key_sorts(eng_local):
    # pile-ups (long runs of equal starts) put NEAREST and group_rows on their two-sort plan: sort by
    # end, then STABLY by start -- every stage of the second sort has to keep the order of ties
    rng = np.random.default_rng(77)
    n = 120_000
    st = (rng.integers(0, 400, n) * 50_000).astype(np.int32)        # 400 distinct starts, ~300 rows each
    b = ora.Side(np.zeros(n, np.int32), st, st + rng.integers(1, 3000, n).astype(np.int32))
    qs = rng.integers(0, 20_000_000, 50_000).astype(np.int32)
    a = ora.Side(np.zeros(50_000, np.int32), qs, qs + rng.integers(1, 500, 50_000).astype(np.int32))
    for _ in range(2):  # the first call discovers the pile-ups and switches plans
        idx, dist = eng_local.nearest(dev(a), dev(b), 1)
        oi, od = ora.c_nearest_k1(a, b, method="sweep")
        assert np.array_equal(dist.cpu().numpy(), od)
        j = idx.cpu().numpy()
        assert np.array_equal(b.start[j], b.start[oi]) and np.array_equal(b.end[j], b.end[oi])
    gid, rep = eng_local.group_rows(dev(b), 1)
    assert rep.shape[0] == len({(int(x), int(y)) for x, y in zip(b.start, b.end)})
    assert eng_local.stats()["sort_local"]


def test_buckets_larger_than_lds_are_sorted_by_their_block_through_the_other_buffer(eng_local):
    # pile-ups: 4096 < rows <= 2^18 in one 65536-wide bucket take the in-block two-pass radix sort
    # (bucket_sort_big) -- every payload shape (INNER general: key + end + rid; fixed-length B: key + rid;
    # COUNT: keys only; SEMI general: key + end), next to ordinary buckets, without any fall-back
    rng = np.random.default_rng(91)
    def side(n_hot, n_rest, fixed=None):
        st = np.concatenate([rng.integers(1_000_000, 1_060_000, n_hot), rng.integers(0, 30_000_000, n_rest)]).astype(np.int32)
        ln = np.full(st.shape[0], fixed, np.int32) if fixed else rng.integers(1, 400, st.shape[0]).astype(np.int32)
        perm = rng.permutation(st.shape[0])
        return ora.Side(np.zeros(st.shape[0], np.int32), st[perm], (st + ln)[perm])
    a, b = side(9_000, 20_000), side(70_000, 150_000)
    st = _all_ops(eng_local, a, b, 1)
    assert st["sort_local"] and not st["sort_resorted"]
    bu = side(70_000, 150_000, fixed=150)
    st = _all_ops(eng_local, a, bu, 1, nearest=False)
    assert st["sort_local"] and not st["sort_resorted"] and st["join_form"] == "uniform_b"
    # ties inside a big bucket keep their order: NEAREST's two-sort plan over 300-row pile-ups
    n = 90_000
    s2 = (rng.integers(0, 300, n) * 200).astype(np.int32)               # all inside one bucket
    b2 = ora.Side(np.zeros(n, np.int32), s2, s2 + rng.integers(1, 3000, n).astype(np.int32))
    for _ in range(2):
        idx, dist = eng_local.nearest(dev(a), dev(b2), 1)
        oi, od = ora.c_nearest_k1(a, b2, method="sweep")
        assert np.array_equal(dist.cpu().numpy(), od)
        j = idx.cpu().numpy()
        assert np.array_equal(b2.start[j], b2.start[oi]) and np.array_equal(b2.end[j], b2.end[oi])
    assert not eng_local.stats()["sort_resorted"]


def test_oversized_bucket_falls_back_to_the_four_pass_sort(eng_local):
    # 300,000 rows inside ONE 65536-wide bucket (more than the in-block sort takes): the call is repeated
    # with the four-pass sort, stays exact, and the context keeps that sort afterwards
    a = rand_side(1301, 20_000, 1, 60_000, 300)
    b = rand_side(1302, 300_000, 1, 60_000, 300)
    want_n = int(ora.c_count(a, b, "sweep").sum())
    ra, rb = eng_local.inner_join(dev(a), dev(b), 1)
    assert int(ra.shape[0]) == want_n
    assert eng_local.pairs_checksum(ra, rb) == ora.c_pairs_checksum(*ora.c_inner(a, b, "sweep"))
    st = eng_local.stats()
    assert st["sort_resorted"] and not st["sort_local"]
    assert np.array_equal(eng_local.semi_join(dev(a), dev(b), 1).cpu().numpy(), ora.c_semi_anti(a, b, False))
    assert not eng_local.stats()["sort_local"]


def test_row_operators_recover_from_an_oversized_bucket(eng_local):
    # same, discovered by a per-row operator first (its read-back is the only one of the call)
    a = rand_side(1311, 20_000, 1, 60_000, 300)
    b = rand_side(1312, 300_000, 1, 60_000, 300)
    assert np.array_equal(eng_local.count_overlaps(dev(a), dev(b), 1).cpu().numpy(), ora.c_count(a, b, "sweep"))
    assert eng_local.stats()["sort_resorted"]
    idx, dist = eng_local.nearest(dev(a), dev(b), 1)
    assert np.array_equal(dist.cpu().numpy(), ora.c_nearest_k1(a, b, method="sweep")[1])


@pytest.mark.parametrize("seed", range(12))
def test_three_stage_randomized_sweep(eng_local, seed):
    r = np.random.default_rng(3000 + seed)
    encs = list(ora.ENCODING_OFFSETS)
    n_chrom = int(r.integers(1, 30))
    max_start = int(r.choice([2_000, 400_000, 80_000_000]))
    max_len = int(r.choice([3, 80, 5_000]))
    min_len = -3 if seed % 2 else 1
    na, nb = (int(r.choice([0, 1, 65, 700, 9_000, 120_000])) for _ in range(2))
    a = rand_side(7000 + seed, na, n_chrom, max_start, max_len + 1, min_len=min_len, enc=encs[int(r.integers(0, 4))])
    b = rand_side(8000 + seed, nb, n_chrom, max_start, max_len + 1, min_len=min_len, enc=encs[int(r.integers(0, 4))])
    _all_ops(eng_local, a, b, n_chrom, nearest=min_len >= 0)


@pytest.mark.parametrize("inject", [False, True])
def test_one_call_join_keeps_its_orientation_when_the_plan_is_repeated(monkeypatch, inject):
    # giql_hip_inner_join_dev offers the caller's buffers to the plan, and a plan with
    # the LARGER table first exchanges the sides.  When that plan is repeated (an oversized bucket ->
    # the four-pass sort; a look-back timeout -> ticket order) AFTER its early fill was launched, the
    # repeat must see the buffers in the same orientation: row_a holds A ids, row_b holds B ids.
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_LOCAL_MIN_ROWS", "1")
    if inject:  # the third clean read-back reports a timeout: the end of the second (speculated) call below
        monkeypatch.setenv("GIQL_HIP_INJECT_TIMEOUT", "3")
    e = HipEngine(0)
    monkeypatch.delenv("GIQL_HIP_LOCAL_MIN_ROWS")
    if inject:
        monkeypatch.delenv("GIQL_HIP_INJECT_TIMEOUT")
    try:
        def check(a, b, nch):
            want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
            cap = want.shape[0] + 1024
            ra = torch.full((cap,), -7, dtype=torch.int32, device="cuda:0")
            rb = torch.full((cap,), -7, dtype=torch.int32, device="cuda:0")
            n = e.inner_join_into(dev(a), dev(b), nch, ra, rb)
            assert n == want.shape[0]
            got = ora.sort_pairs(ra[:n].cpu().numpy(), rb[:n].cpu().numpy())
            assert np.array_equal(got, want)  # pair ORIENTATION included: column 0 = row of a
            assert int((ra[n:] != -7).sum()) == 0 and int((rb[n:] != -7).sum()) == 0

        # a first plan (two read-backs) so that the context speculates: fixed-length larger side, no irregular rows
        check(uniform_side(1401, 30_000, 1, 40_000_000, 150), rand_side(1402, 5_000, 1, 40_000_000, 900), 1)
        small = rand_side(1404, 20_000, 1, 60_000, 300)
        if inject:
            # larger table FIRST; the early fill is launched, the read-back reports the (injected) timeout,
            # the call is repeated once in ticket order
            big = uniform_side(1403, 300_000, 1, 40_000_000, 150)
            check(big, small, 1)
            assert e.stats()["sort_order_fallbacks"] == 1
        else:
            # larger table FIRST, 300K rows inside one 65536-wide window: the early fill is launched, the bucket
            # sort reports the oversized bucket at the read-back, the call is repeated once with four passes
            big = uniform_side(1403, 300_000, 1, 60_000, 150)
            check(big, small, 1)
            assert e.stats()["sort_resorted"]
        check(big, small, 1)   # and once more on the settled context
        check(small, big, 1)
    finally:
        e.close()


# ---- the range count fused into the bucket sort (fixed-length INNER form, round 3) ----------------
def _fused_inner(e, a, b, nch, expect_fused=True):
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    got = _inner(e, a, b, nch)
    st = e.stats()
    assert got.shape == want.shape and np.array_equal(got, want)
    assert st["count_fused"] == expect_fused, st
    return st


def test_fused_count_matches_the_count_kernel_in_every_query_order(eng_local, monkeypatch):
    from giql_amd.engine import HipEngine

    reads = uniform_side(1501, 400_000, 5, 30_000_000, 150)
    peaks = rand_side(1502, 60_000, 5, 30_000_000, 2000, min_len=200)
    # first plan: queries sorted on every digit; second: the context's guesses hold -> lowest digit unsorted
    st = _fused_inner(eng_local, peaks, reads, 5)
    assert st["join_form"] == "uniform_b" and st["sort_local"]
    _fused_inner(eng_local, peaks, reads, 5)
    _fused_inner(eng_local, reads, peaks, 5)       # larger table first: the plan exchanges the sides
    # the one-call form on the speculating context (early fill behind the fused count)
    want = ora.sort_pairs(*ora.c_inner(peaks, reads, "sweep"))
    ra = torch.empty(want.shape[0] + 64, dtype=torch.int32, device="cuda:0")
    rb = torch.empty_like(ra)
    n = eng_local.inner_join_into(dev(peaks), dev(reads), 5, ra, rb)
    assert n == want.shape[0] and np.array_equal(ora.sort_pairs(ra[:n].cpu().numpy(), rb[:n].cpu().numpy()), want)
    assert eng_local.stats()["count_fused"] and eng_local.stats()["fused_fill"]
    # and the separate count kernel gives the same pairs
    monkeypatch.setenv("GIQL_HIP_LOCAL_MIN_ROWS", "1")
    monkeypatch.setenv("GIQL_HIP_NO_FUSE_COUNT", "1")
    e2 = HipEngine(0)
    monkeypatch.delenv("GIQL_HIP_LOCAL_MIN_ROWS")
    monkeypatch.delenv("GIQL_HIP_NO_FUSE_COUNT")
    try:
        st2 = _fused_inner(e2, peaks, reads, 5, expect_fused=False)
        assert st2["sort_local"]
    finally:
        e2.close()


def test_fused_count_edges_of_the_axis_empty_buckets_and_irregular_queries(eng_local):
    r = np.random.default_rng(1510)
    # reads at both ends of two chromosomes with a wide empty stretch between (empty buckets that still
    # have bounds to answer), queries reaching below the first read (q.start - L + 1 < 0) and past the last
    st = np.concatenate([r.integers(0, 3_000, 4_000), r.integers(9_000_000, 9_100_000, 4_000)]).astype(np.int32)
    reads = ora.Side(r.integers(0, 2, st.size).astype(np.int32), st, st + np.int32(100))
    qs = np.concatenate([r.integers(0, 200, 500), r.integers(2_000_000, 7_000_000, 500),
                         r.integers(9_050_000, 9_300_000, 500)]).astype(np.int32)
    ql = r.integers(1, 4_000, qs.size).astype(np.int32)
    peaks = ora.Side(r.integers(0, 2, qs.size).astype(np.int32), qs, qs + ql)
    _fused_inner(eng_local, peaks, reads, 2)
    _fused_inner(eng_local, peaks, reads, 2)
    # zero-length and inverted QUERY rows take the literal path beside the fused count
    pe = peaks.end.copy()
    pe[::7] = peaks.start[::7]
    pe[3::11] = peaks.start[3::11] - 5
    irr = ora.Side(peaks.chrom, peaks.start, pe)
    st3 = _fused_inner(eng_local, irr, reads, 2)
    assert st3["n_irregular_a"] > 0
    _fused_inner(eng_local, irr, reads, 2)
    # every encoding of the query side (offsets on both columns)
    for enc in ora.ENCODING_OFFSETS:
        so, eo = ora.ENCODING_OFFSETS[enc]
        _fused_inner(eng_local, ora.Side(peaks.chrom, peaks.start, peaks.end, so, eo), reads, 2)


def test_fused_count_declines_long_query_rows_and_recovers(eng_local):
    reads = uniform_side(1521, 300_000, 3, 20_000_000, 150)
    short = rand_side(1522, 30_000, 3, 20_000_000, 3000)
    long_ = rand_side(1523, 30_000, 3, 20_000_000, 3000)
    long_.end[17] = long_.start[17] + 2_000_000      # one row longer than the windows allow for
    _fused_inner(eng_local, short, reads, 3)
    _fused_inner(eng_local, short, reads, 3)
    # the speculating context launches the fused form, the read-back tells the long row: planned again without
    _fused_inner(eng_local, long_, reads, 3, expect_fused=False)
    _fused_inner(eng_local, long_, reads, 3, expect_fused=False)
    _fused_inner(eng_local, short, reads, 3, expect_fused=False)   # the guess follows the previous plan ...
    _fused_inner(eng_local, short, reads, 3)                        # ... and is back


def test_fused_count_with_buckets_larger_than_lds(eng_local):
    # ~6,500 rows in every 65536-wide window: every bucket goes through the in-block global sort, which
    # answers its window's bounds by binary search
    reads = uniform_side(1531, 700_000, 1, 7_000_000, 150)
    peaks = rand_side(1532, 40_000, 1, 7_000_000, 2500)
    _fused_inner(eng_local, peaks, reads, 1)
    _fused_inner(eng_local, peaks, reads, 1)
    # equal keys galore: 500 distinct read starts (bins with equal sub-values: the gathered-bin rank)
    r = np.random.default_rng(1533)
    st = (r.integers(0, 500, 200_000) * 37).astype(np.int32)
    piled = ora.Side(np.zeros(st.size, np.int32), st, st + np.int32(150))
    q = rand_side(1534, 3_000, 1, 20_000, 400)
    _fused_inner(eng_local, q, piled, 1, expect_fused=True)


# ---- the join itself in the bucket stage (one-call form, round 3) ---------------------------------
def _join_into(e, a, b, nch, expect_join=True, slack=64):
    """inner_join_into on a context whose guesses hold: pairs (orientation included) against the oracle,
    nothing written past the count."""
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    cap = want.shape[0] + slack
    ra = torch.full((cap,), -7, dtype=torch.int32, device="cuda:0")
    rb = torch.full((cap,), -7, dtype=torch.int32, device="cuda:0")
    n = e.inner_join_into(dev(a), dev(b), nch, ra, rb)
    st = e.stats()
    assert n == want.shape[0]
    assert np.array_equal(ora.sort_pairs(ra[:n].cpu().numpy(), rb[:n].cpu().numpy()), want)
    assert int((ra[n:] != -7).sum()) == 0 and int((rb[n:] != -7).sum()) == 0
    if expect_join is not None:
        assert st["bucket_join"] == expect_join, st
    return st


def test_bucket_join_matches_the_oracle_in_both_argument_orders(eng_local):
    from giql_amd import _lib

    reads = uniform_side(1601, 400_000, 5, 30_000_000, 150)
    peaks = rand_side(1602, 60_000, 5, 30_000_000, 2000, min_len=200)
    _fused_inner(eng_local, peaks, reads, 5)            # a first plan: the context learns form, span, lengths
    st = _join_into(eng_local, peaks, reads, 5)
    assert st["count_fused"] and st["fused_fill"] and st["join_form"] == "uniform_b"
    _join_into(eng_local, peaks, reads, 5)
    _fused_inner(eng_local, reads, peaks, 5)
    st = _join_into(eng_local, reads, peaks, 5)         # larger table first: planned with the sides exchanged
    assert st["swapped"]
    # nothing but the pairs left that call: fill and export need a plan of their own
    ra = torch.empty(16, dtype=torch.int32, device="cuda:0")
    with pytest.raises(_lib.GiqlHipError) as exc:
        eng_local.inner_fill(ra, ra.clone())
    assert exc.value.code == _lib.GIQL_ERR_STATE
    n = eng_local.inner_plan(dev(peaks), dev(reads), 5)  # ... and after one they work again
    ra = torch.empty(n, dtype=torch.int32, device="cuda:0")
    rb = torch.empty_like(ra)
    eng_local.inner_fill(ra, rb)
    assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), ora.sort_pairs(*ora.c_inner(peaks, reads, "sweep")))


def test_bucket_join_edges_long_queries_and_every_encoding(eng_local):
    r = np.random.default_rng(1610)
    st = np.concatenate([r.integers(0, 3_000, 4_000), r.integers(9_000_000, 9_100_000, 4_000)]).astype(np.int32)
    reads = ora.Side(r.integers(0, 2, st.size).astype(np.int32), st, st + np.int32(100))
    qs = np.concatenate([r.integers(0, 200, 500), r.integers(2_000_000, 7_000_000, 500),
                         r.integers(9_000_000, 9_300_000, 1500)]).astype(np.int32)
    ql = r.integers(1, 30_000, qs.size).astype(np.int32)   # ranges that span two (and touch three) buckets
    peaks = ora.Side(r.integers(0, 2, qs.size).astype(np.int32), qs, qs + ql)
    _fused_inner(eng_local, peaks, reads, 2)
    _join_into(eng_local, peaks, reads, 2)
    joined = 0
    for enc in ora.ENCODING_OFFSETS:
        # (a 1-based encoding puts the query at start 0 on canonical -1: the aligned layout declines, the call is
        # planned again the ordinary way and the context's next call joins in the bucket stage again)
        so, eo = ora.ENCODING_OFFSETS[enc]
        side = ora.Side(peaks.chrom, peaks.start, peaks.end, so, eo)
        _join_into(eng_local, side, reads, 2, expect_join=None)
        joined += int(_join_into(eng_local, side, reads, 2, expect_join=None)["bucket_join"])
    assert joined >= 2
    _join_into(eng_local, peaks, reads, 2, expect_join=None)
    _join_into(eng_local, peaks, reads, 2)
    # irregular query rows: the guess fails at the read-back, the call is planned again the ordinary way
    pe = peaks.end.copy()
    pe[::7] = peaks.start[::7]
    irr = ora.Side(peaks.chrom, peaks.start, pe)
    _join_into(eng_local, irr, reads, 2, expect_join=False)
    _join_into(eng_local, irr, reads, 2, expect_join=False)
    _join_into(eng_local, peaks, reads, 2, expect_join=False)   # the guess follows the previous plan ...
    _join_into(eng_local, peaks, reads, 2)                      # ... and is back


def test_bucket_join_queued_buckets_equal_keys_and_crowded_windows(eng_local):
    # ~6,500 rows in every 65536-wide window: every bucket goes through the queue (sorted in global memory,
    # its window answered by binary search, its pairs written from there)
    reads = uniform_side(1631, 700_000, 1, 7_000_000, 150)
    peaks = rand_side(1632, 40_000, 1, 7_000_000, 2500)
    _fused_inner(eng_local, peaks, reads, 1)
    _join_into(eng_local, peaks, reads, 1)
    # equal keys galore (bins with equal sub-values: the gathered-bin rank)
    r = np.random.default_rng(1633)
    st = (r.integers(0, 500, 200_000) * 37).astype(np.int32)
    piled = ora.Side(np.zeros(st.size, np.int32), st, st + np.int32(150))
    q = rand_side(1634, 3_000, 1, 20_000, 400)
    _fused_inner(eng_local, q, piled, 1)
    _join_into(eng_local, q, piled, 1)
    # crowds of queries in one window inside a sparse table: 2,500 of them (more than the bucket kernel's three rounds
    # hold: the bucket runs the LDS body with eight rounds in the queue kernel) and 6,000 (past that too: the
    # global-memory path); the other buckets run in the bucket kernel
    reads2 = uniform_side(1635, 300_000, 2, 40_000_000, 150)
    qs = np.concatenate([r.integers(0, 40_000_000, 4_000), r.integers(5_000_000, 5_030_000, 6_000),
                         r.integers(9_000_000, 9_030_000, 2_500)]).astype(np.int32)
    ql = r.integers(50, 1_500, qs.size).astype(np.int32)
    crowd = ora.Side(r.integers(0, 2, qs.size).astype(np.int32), qs, qs + ql)
    _fused_inner(eng_local, crowd, reads2, 2)
    _join_into(eng_local, crowd, reads2, 2)


def test_bucket_join_with_buffers_too_small_reports_the_count_and_keeps_a_plan(eng_local):
    from giql_amd import _lib

    reads = uniform_side(1641, 300_000, 3, 20_000_000, 150)
    peaks = rand_side(1642, 30_000, 3, 20_000_000, 3000)
    want = ora.sort_pairs(*ora.c_inner(peaks, reads, "sweep"))
    _fused_inner(eng_local, peaks, reads, 3)
    ra = torch.full((want.shape[0] // 2,), -7, dtype=torch.int32, device="cuda:0")
    rb = torch.full_like(ra, -7)
    with pytest.raises(_lib.GiqlHipError) as exc:
        eng_local.inner_join_into(dev(peaks), dev(reads), 3, ra, rb)
    assert exc.value.code == _lib.GIQL_ERR_CAPACITY and eng_local.last_pairs == want.shape[0]
    fa = torch.empty(want.shape[0], dtype=torch.int32, device="cuda:0")
    fb = torch.empty_like(fa)
    eng_local.inner_fill(fa, fb)                        # the plan behind the error is an ordinary one
    assert np.array_equal(ora.sort_pairs(fa.cpu().numpy(), fb.cpu().numpy()), want)
    _join_into(eng_local, peaks, reads, 3)              # and with room the same context joins in the bucket stage again


# ---- the general (two-class) join in the bucket stage: rows of any length (round 3) ----------------------------
def _plain(e, a, b, nch):
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    got = _inner(e, a, b, nch)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_general_bucket_join_matches_the_oracle_in_both_argument_orders(eng_local):
    reads = rand_side(1701, 400_000, 5, 30_000_000, 400, min_len=30)     # variable-length reads: the general form
    peaks = rand_side(1702, 60_000, 5, 30_000_000, 2000, min_len=200)
    _plain(eng_local, peaks, reads, 5)               # a first plan: the context learns form, span, lengths
    st = _join_into(eng_local, peaks, reads, 5)
    assert st["join_form"] == "general" and st["fused_fill"] and st["sort_local"]
    _join_into(eng_local, peaks, reads, 5)
    _plain(eng_local, reads, peaks, 5)
    st = _join_into(eng_local, reads, peaks, 5)     # larger table first: planned with the sides exchanged
    assert st["swapped"] and st["join_form"] == "general"
    # equal starts on both sides (class 1 takes them, class 2 must not), starts at the bucket boundaries
    r = np.random.default_rng(1703)
    grid = (r.integers(0, 600, 50_000) * 65536 // 8).astype(np.int32)
    b2 = ora.Side(np.zeros(grid.size, np.int32), grid, grid + r.integers(1, 20_000, grid.size).astype(np.int32))
    a2 = ora.Side(np.zeros(8_000, np.int32), grid[:8_000].copy(), grid[:8_000] + r.integers(1, 30_000, 8_000).astype(np.int32))
    _plain(eng_local, a2, b2, 1)
    _join_into(eng_local, a2, b2, 1)


def test_general_bucket_join_encodings_irregular_rows_and_long_rows(eng_local):
    reads = rand_side(1711, 300_000, 3, 20_000_000, 300, min_len=20)
    peaks = rand_side(1712, 30_000, 3, 20_000_000, 3000)
    _plain(eng_local, peaks, reads, 3)
    _join_into(eng_local, peaks, reads, 3)
    joined = 0
    for enc in ora.ENCODING_OFFSETS:
        so, eo = ora.ENCODING_OFFSETS[enc]
        side = ora.Side(reads.chrom, reads.start, reads.end, so, eo)
        _join_into(eng_local, peaks, side, 3, expect_join=None)
        joined += int(_join_into(eng_local, peaks, side, 3, expect_join=None)["bucket_join"])
    assert joined >= 2
    _join_into(eng_local, peaks, reads, 3, expect_join=None)
    _join_into(eng_local, peaks, reads, 3)
    # an irregular row on the larger side: the guess fails at the read-back, planned again the ordinary way
    re = reads.end.copy()
    re[::9] = reads.start[::9]
    irr = ora.Side(reads.chrom, reads.start, re)
    _join_into(eng_local, peaks, irr, 3, expect_join=False)
    _join_into(eng_local, peaks, irr, 3, expect_join=False)
    _join_into(eng_local, peaks, reads, 3, expect_join=False)
    _join_into(eng_local, peaks, reads, 3)
    # one row longer than the windows allow for: declined at the read-back, and from then on by the guess
    long_ = rand_side(1713, 300_000, 3, 20_000_000, 300, min_len=20)
    long_.end[17] = long_.start[17] + 2_000_000
    _join_into(eng_local, peaks, long_, 3, expect_join=False)
    _join_into(eng_local, peaks, long_, 3, expect_join=False)


def test_general_bucket_join_queued_buckets_and_small_buffers(eng_local):
    from giql_amd import _lib

    # ~6,500 rows in every 65536-wide window: every bucket goes through the queue
    reads = rand_side(1721, 700_000, 1, 7_000_000, 300, min_len=50)
    peaks = rand_side(1722, 40_000, 1, 7_000_000, 2500)
    _plain(eng_local, peaks, reads, 1)
    _join_into(eng_local, peaks, reads, 1)
    # crowded windows in the general form (2,500 and 6,000 A rows inside one bucket's reach)
    r = np.random.default_rng(1723)
    reads2 = rand_side(1724, 300_000, 2, 40_000_000, 300, min_len=40)
    qs = np.concatenate([r.integers(0, 40_000_000, 4_000), r.integers(5_000_000, 5_030_000, 6_000),
                         r.integers(9_000_000, 9_030_000, 2_500)]).astype(np.int32)
    crowd = ora.Side(r.integers(0, 2, qs.size).astype(np.int32), qs, qs + r.integers(50, 1_500, qs.size).astype(np.int32))
    _plain(eng_local, crowd, reads2, 2)
    _join_into(eng_local, crowd, reads2, 2)
    # buffers too small: the count comes back, the plan behind the error is an ordinary one
    want = ora.sort_pairs(*ora.c_inner(peaks, reads, "sweep"))
    ra = torch.full((want.shape[0] // 3,), -7, dtype=torch.int32, device="cuda:0")
    rb = torch.full_like(ra, -7)
    with pytest.raises(_lib.GiqlHipError) as exc:
        eng_local.inner_join_into(dev(peaks), dev(reads), 1, ra, rb)
    assert exc.value.code == _lib.GIQL_ERR_CAPACITY and eng_local.last_pairs == want.shape[0]
    fa = torch.empty(want.shape[0], dtype=torch.int32, device="cuda:0")
    fb = torch.empty_like(fa)
    eng_local.inner_fill(fa, fb)
    assert np.array_equal(ora.sort_pairs(fa.cpu().numpy(), fb.cpu().numpy()), want)


def test_engine_inner_join_takes_the_one_call_form_from_its_second_call(eng_local):
    """HipEngine.inner_join (what execute() calls): buffers sized from the context's previous result, one C-ABI call --
    so the product API gets the join in the bucket stage too; a result that outgrows the guess falls back to the
    exact-size fill behind GIQL_ERR_CAPACITY."""
    reads = uniform_side(1901, 300_000, 3, 20_000_000, 150)
    peaks = rand_side(1902, 30_000, 3, 20_000_000, 3000)
    _plain(eng_local, peaks, reads, 3)
    assert not eng_local.stats()["bucket_join"]          # first call: plan + fill
    _plain(eng_local, peaks, reads, 3)
    _plain(eng_local, peaks, reads, 3)
    assert eng_local.stats()["bucket_join"] and eng_local.stats()["fused_fill"]
    wide = rand_side(1903, 30_000, 3, 20_000_000, 30_000, min_len=10_000)   # ten times the pairs: the guess is too small
    _plain(eng_local, wide, reads, 3)
    _plain(eng_local, wide, reads, 3)
    assert eng_local.stats()["bucket_join"]


def test_bucket_join_with_a_fixed_length_longer_than_a_bucket(eng_local):
    # L = 150,000 > 65536: every query's range [q.start - L + 1, q.end) spans three and more buckets (whole
    # buckets in the middle), the windows reach that far above each bucket
    reads = uniform_side(1961, 120_000, 2, 60_000_000, 150_000)
    peaks = rand_side(1962, 3_000, 2, 60_000_000, 5_000)
    _fused_inner(eng_local, peaks, reads, 2)
    _join_into(eng_local, peaks, reads, 2, expect_join=None)
    _join_into(eng_local, peaks, reads, 2, expect_join=None)


@pytest.mark.parametrize("seed", range(10))
def test_bucket_join_randomized_sweep(eng_local, seed):  # noqa: C901
    """Random shapes through the one-call form twice in a row (the second call runs on the context's settled
    guesses: the join in the bucket stage whenever the shape allows it), fixed-length and general forms, both
    argument orders, every encoding, clustered queries, more chromosomes than the aligned layout holds."""
    r = np.random.default_rng(7700 + seed)
    n_chrom = int(r.choice([1, 3, 24, 40]))
    span = int(r.choice([300_000, 20_000_000, 2_000_000_000 // n_chrom]))
    nq, nu = int(r.choice([700, 9_000, 60_000])), int(r.choice([30_000, 250_000]))
    fixed = int(r.choice([0, 36, 150, 90_000]))
    encs = list(ora.ENCODING_OFFSETS)
    u = (uniform_side(7800 + seed, nu, n_chrom, span, fixed) if fixed
         else rand_side(7800 + seed, nu, n_chrom, span, int(r.choice([60, 3_000])), min_len=1))
    q = rand_side(7900 + seed, nq, n_chrom, span, int(r.choice([50, 4_000, 30_000])), min_len=1, enc=encs[int(r.integers(0, 4))])
    if r.random() < 0.5:   # a crowd of queries in one stretch
        k = nq // 2
        q.start[:k] = (span // 3 + r.integers(0, 40_000, k)).astype(np.int32)
        q.end[:k] = q.start[:k] + r.integers(1, 2_000, k).astype(np.int32)
    a, b = (q, u) if r.random() < 0.5 else (u, q)
    for _ in range(3):
        st = _join_into(eng_local, a, b, n_chrom, expect_join=None)
    _SWEEP_JOINED.append(bool(st["bucket_join"]))


_SWEEP_JOINED: list = []


def test_bucket_join_randomized_sweep_engaged_the_bucket_stage():
    # (the sweep above is only worth its name if a good part of its shapes took the form under test)
    assert len(_SWEEP_JOINED) == 10 and sum(_SWEEP_JOINED) >= 4, _SWEEP_JOINED
