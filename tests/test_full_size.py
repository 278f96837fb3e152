"""BASELINE.json configs 2, 3 and 5 at their STATED sizes against the oracle, through the C ABI --
needs a GPU.  (Config 4's full-size check lives in test_gpu_parity.py; the small-input tests there
cover the edge cases, these cover what only shows at scale: hundreds of sort tiles, the (key, end)
sorts, the prefix-max scans and the equal-start fix-up of NEAREST over 10M rows.)

Inputs follow SURVEY.md section 8(d): PCG64 seeds 1/2 (config 2), 3/4 (config 3), 7/8 (config 5),
hg38 chromosome lengths.  Exact arrays for the per-row operators; pair count + the order-independent
64-bit multiset checksum for the 2.2e8 pairs of the dense config 2 join.
"""

import numpy as np
import pytest

from giql_amd import synth
from oracle import pyoracle as ora
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _engine(monkeypatch, **env):
    from giql_amd.engine import HipEngine

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = HipEngine(0)
    for k in env:
        monkeypatch.delenv(k)
    return e


@pytest.fixture(scope="module")
def cfg3():
    a = ora.Side(*synth.make_table(1_000_000, 3, "peaks"))
    b = ora.Side(*synth.make_table(10_000_000, 4, "reads"))
    want = {"semi": ora.c_semi_anti(a, b, False), "anti": ora.c_semi_anti(a, b, True), "count": ora.c_count(a, b, "sweep")}
    assert want["semi"].shape[0] + want["anti"].shape[0] == a.n and 0 < want["anti"].shape[0] < a.n
    return a, b, want


@pytest.mark.parametrize("env,form", [
    ({}, "uniform_b"),                                   # fixed-length B: one sorted array (coarsely sorted from the second call on), no prefix max
    ({"GIQL_HIP_NO_COARSE_B": "1"}, "uniform_b"),        # ... sorted on every digit, binary searches
    ({"GIQL_HIP_NO_UNIFORM": "1"}, "general"),           # (key, end) sort + prefix max / two sorted arrays
    ({"GIQL_HIP_LOCAL_MIN_ROWS": "1"}, "uniform_b"),     # three-stage sort of both sides
    ({"GIQL_HIP_LOCAL_MIN_ROWS": "1", "GIQL_HIP_NO_UNIFORM": "1"}, "general"),
])
def test_config3_semi_anti_count_1m_x_10m(monkeypatch, cfg3, env, form):
    a, b, want = cfg3
    e = _engine(monkeypatch, **env)
    try:
        da, db = dev(a), dev(b)
        for _ in range(2):  # the second round runs on the context's speculated form
            assert np.array_equal(e.semi_join(da, db, 24).cpu().numpy(), want["semi"])
            assert e.stats()["join_form"] == form
            assert np.array_equal(e.anti_join(da, db, 24).cpu().numpy(), want["anti"])
            assert np.array_equal(e.count_overlaps(da, db, 24).cpu().numpy(), want["count"])
            assert e.stats()["join_form"] == form
        assert e.stats()["sort_local"] == ("GIQL_HIP_LOCAL_MIN_ROWS" in env) and not e.stats()["sort_resorted"]
        assert e.stats()["coarse_b"] == (env == {})
    finally:
        e.close()


@pytest.mark.parametrize("n_b,env", [
    (10_000_000, {}),                                   # config 5 as stated: 99.7 % of the rows overlap a target (distance 0)
    (10_000_000, {"GIQL_HIP_LOCAL_MIN_ROWS": "1"}),
    (1_000_000, {}),                                    # a sparser target table: half of the rows fall between targets
])
def test_config5_nearest_10m_x_10m(monkeypatch, n_b, env):
    a = ora.Side(*synth.make_table(10_000_000, 7, "peaks"))
    b = ora.Side(*synth.make_table(n_b, 8, "peaks"))
    wi, wd = ora.c_nearest_k1(a, b, method="sweep")
    assert (wi >= 0).all() and (wd == 0).sum() > 100_000 and (wd > 0).sum() > (20_000 if n_b == 10_000_000 else 3_000_000)
    e = _engine(monkeypatch, **env)
    try:
        idx, dist = e.nearest(dev(a), dev(b), 24)
        assert np.array_equal(dist.cpu().numpy(), wd)
        j = idx.cpu().numpy()
        assert (j >= 0).all()
        # ties on identical (start, end) are order-ambiguous upstream (nearest.py:366-372): compare coordinates
        assert np.array_equal(b.start[j], b.start[wi]) and np.array_equal(b.end[j], b.end[wi])
        # signed distances and a max_distance on the same tables
        idx2, dist2 = e.nearest(dev(a), dev(b), 24, signed=True, max_distance=500)
        wi2, wd2 = ora.c_nearest_k1(a, b, signed=True, max_distance=500, method="sweep")
        assert np.array_equal(dist2.cpu().numpy(), wd2) and np.array_equal(idx2.cpu().numpy() >= 0, wi2 >= 0)
        # the 8-byte-record output (giql_hip_nearest32_dev: what bench.py times for config 5)
        rec = e.nearest32(dev(a), dev(b), 24).cpu().numpy()
        assert np.array_equal(rec[:, 1], wd) and np.array_equal(b.start[rec[:, 0]], b.start[wi]) \
            and np.array_equal(b.end[rec[:, 0]], b.end[wi])
        rec2 = e.nearest32(dev(a), dev(b), 24, signed=True, max_distance=500).cpu().numpy()
        assert np.array_equal(rec2[:, 1], wd2) and np.array_equal(rec2[:, 0] >= 0, wi2 >= 0)
        assert e.stats()["sort_local"] == bool(env)
    finally:
        e.close()


@pytest.mark.parametrize("genome,env", [
    (10_000_000, {}),                                 # dense: ~2.2e8 pairs, 220 matches per row
    (10_000_000, {"GIQL_HIP_LOCAL_MIN_ROWS": "1"}),   # 6.5K rows per 16-bit bucket: every bucket takes the in-block big path
    (248_956_422, {"GIQL_HIP_LOCAL_MIN_ROWS": "1"}),  # sparse: ~260 rows per bucket in the three-stage sort
])
def test_config2_1m_x_1m_single_chromosome_full_result(monkeypatch, genome, env):
    a = ora.Side(*synth.make_single_chrom(1_000_000, 1, "peaks", genome))
    b = ora.Side(*synth.make_single_chrom(1_000_000, 2, "peaks", genome))
    ra, rb = ora.c_inner(a, b, "sweep")
    want_n, want_sum = int(ra.shape[0]), ora.c_pairs_checksum(ra, rb)
    assert want_n > (150_000_000 if genome == 10_000_000 else 8_000_000)
    counts = np.bincount(ra, minlength=a.n)
    del ra, rb
    e = _engine(monkeypatch, **env)
    try:
        ga, gb = e.inner_join(dev(a), dev(b), 1)
        assert int(ga.shape[0]) == want_n
        assert e.pairs_checksum(ga, gb) == want_sum
        assert np.array_equal(torch.bincount(ga.long(), minlength=a.n).cpu().numpy(), counts)   # per-row multiplicity
        st = e.stats()
        if env:
            assert st["sort_local"] and not st["sort_resorted"]
        assert np.array_equal(e.count_overlaps(dev(a), dev(b), 1).cpu().numpy(), counts)
    finally:
        e.close()


def test_skewed_reads_hot_windows_take_the_big_bucket_path(monkeypatch):
    # real read tables are far from uniform: 40M fixed-length reads, a fifth of them piled into 200 hot
    # 20-kb windows (~40K rows in each of those 16-bit buckets), against 2M peaks.  The three-stage sort
    # keeps its form (the hot buckets go through bucket_sort_big), nothing falls back, results exact.
    rng = np.random.default_rng(42)
    n_b, n_hot, n_a = 40_000_000, 8_000_000, 2_000_000
    lengths = synth.HG38_LENGTHS
    bc, bs, be = synth.make_table(n_b - n_hot, 11, "reads")
    hot_c = rng.integers(0, 24, 200)
    hot_s = (rng.random(200) * (lengths[hot_c] - 100_000)).astype(np.int64)
    pick = rng.integers(0, 200, n_hot)
    hs = (hot_s[pick] + rng.integers(0, 20_000, n_hot)).astype(np.int32)
    b = ora.Side(np.concatenate([bc, hot_c[pick].astype(np.int32)]), np.concatenate([bs, hs]), np.concatenate([be, hs + np.int32(150)]))
    a = ora.Side(*synth.make_table(n_a, 12, "peaks"))
    ra, rb = ora.c_inner(a, b, "sweep")
    want_n, want_sum = int(ra.shape[0]), ora.c_pairs_checksum(ra, rb)
    del ra, rb
    e = _engine(monkeypatch)
    try:
        for _ in range(2):
            ga, gb = e.inner_join(dev(a), dev(b), 24)
            st = e.stats()
            assert int(ga.shape[0]) == want_n and e.pairs_checksum(ga, gb) == want_sum
            assert st["sort_local"] and not st["sort_resorted"] and st["join_form"] == "uniform_b"
            del ga, gb
        assert np.array_equal(e.count_overlaps(dev(a), dev(b), 24).cpu().numpy(), ora.c_count(a, b, "sweep"))
    finally:
        e.close()


def test_config4_sized_join_with_variable_length_reads_both_argument_orders(monkeypatch):
    """10M x 100M rows with VARIABLE read lengths (no fixed-length shortcut applies): the general two-class
    form at the headline size -- the larger side's three-array sort from its raw columns through the
    three-stage sort, class 1 over 100M rows, the plan exchanging the sides when the larger table comes
    first.  Pair count and the order-independent 64-bit multiset checksum of all ~4.5e8 pairs against the
    oracle's sort-merge, for both argument orders, twice each (the second call runs on the context's guesses)."""
    r = np.random.default_rng(606)
    ac, as_, ae = synth.make_table(10_000_000, 5, "peaks")
    bc, bs, be = synth.make_table(100_000_000, 6, "reads")
    be = (be + r.integers(0, 60, be.shape[0])).astype(np.int32)       # lengths 150..209
    a, b = ora.Side(ac, as_, ae), ora.Side(bc, bs, be)
    wa, wb = ora.c_inner(a, b, "sweep")
    n_want = int(wa.shape[0])
    want_sum = ora.c_pairs_checksum(wa, wb)
    want_swapped = ora.c_pairs_checksum(wb, wa)
    del wa, wb
    assert 4.0e8 < n_want < 5.5e8
    e = _engine(monkeypatch)
    try:
        da, db = dev(a), dev(b)
        for it in range(2):
            ra, rb = e.inner_join(da, db, 24)
            st = e.stats()
            assert st["join_form"] == "general" and st["sort_local"] and not st["swapped"] and st["span_hist"]
            assert int(ra.shape[0]) == n_want and e.pairs_checksum(ra, rb) == want_sum, it
            del ra, rb
            rb2, ra2 = e.inner_join(db, da, 24)     # the larger table first
            st = e.stats()
            assert st["join_form"] == "general" and st["swapped"] and st["n_a"] == 100_000_000
            assert int(ra2.shape[0]) == n_want and e.pairs_checksum(rb2, ra2) == want_swapped, it
            del ra2, rb2
    finally:
        e.close()


def test_dense_table_takes_narrow_buckets_once_its_span_is_known(monkeypatch):
    """40M rows on a 2e8-position axis = ~13,000 rows per 65,536 positions, three times what the in-LDS stage
    holds: the span is read back before anything is sorted on a context's first call, so already that call
    takes buckets of 2^13 keys (as does every later one, on the remembered span; round 3 -- and
    GIQL_HIP_NO_NARROW_BUCKETS=1 -- left the three-stage sort for four global passes instead).  A context forced
    into 65,536-key buckets goes through the big-bucket queue.  Same rows out of all of them."""
    r = np.random.default_rng(99)
    nb, na = 40_000_000, 200_000
    sb = r.integers(0, 200_000_000, nb).astype(np.int32)
    b = ora.Side(np.zeros(nb, np.int32), sb, (sb + 100).astype(np.int32))
    sa = r.integers(0, 200_000_000, na).astype(np.int32)
    a = ora.Side(np.zeros(na, np.int32), sa, (sa + r.integers(1, 500, na)).astype(np.int32))
    want = ora.c_count(a, b, "sweep")
    e = _engine(monkeypatch)
    try:
        da, db = dev(a), dev(b)
        for it in range(2):
            got = e.count_overlaps(da, db, 1).cpu().numpy()
            st = e.stats()
            assert st["sort_local"] and st["bucket_bits"] == 13 and not st["sort_resorted"], (it, st)
            assert np.array_equal(got, want), it
        # the INNER plan: first call (span read back mid-plan, the digits recounted for the narrow form), then speculated
        for it in range(2):
            n = e.inner_plan(da, db, 1)
            st = e.stats()
            assert n == int(want.sum()) and st["sort_local"] and st["count_fused"] and st["bucket_bits"] == 13, (it, st)
    finally:
        e.close()
    g = _engine(monkeypatch, GIQL_HIP_NO_NARROW_BUCKETS="1")
    try:
        da, db = dev(a), dev(b)
        got1 = g.count_overlaps(da, db, 1).cpu().numpy()
        first_local = g.stats()["sort_local"]
        got2 = g.count_overlaps(da, db, 1).cpu().numpy()
        assert not first_local and not g.stats()["sort_local"] and not g.stats()["sort_resorted"]
        assert np.array_equal(got1, want) and np.array_equal(got2, want)
        for it in range(2):
            n = g.inner_plan(da, db, 1)
            assert n == int(want.sum()) and not g.stats()["sort_local"], it
    finally:
        g.close()
    f = _engine(monkeypatch, GIQL_HIP_LOCAL_MAX_BUCKET_ROWS="1e9")
    try:
        got3 = f.count_overlaps(dev(a), dev(b), 1).cpu().numpy()
        st = f.stats()
        assert st["sort_local"] and st["bucket_bits"] == 16 and np.array_equal(got3, want)
    finally:
        f.close()


# ---- past 2^32 pairs (the 64-bit offset paths under -m gpu, not in a tool) ------------------
def _check_big_join(eng, A, B, n_chrom, expect_fused):
    """No oracle at this size: the pair count must equal the sum of the COUNT operator's per-row counts, the
    per-row multiplicity of the pairs must equal those counts, ids must be in range, the predicate must hold
    on a strided sample, and the SEMI count must equal the rows with a non-zero count."""
    from giql_amd.engine import DeviceSide

    a, b = DeviceSide.from_numpy(*A), DeviceSide.from_numpy(*B)
    counts = eng.count_overlaps(a, b, n_chrom)
    total = int(counts.sum().item())
    assert total > 2**32
    n = eng.inner_plan(a, b, n_chrom)
    assert n == total
    ra = torch.empty(n, dtype=torch.int32, device="cuda:0")
    rb = torch.empty(n, dtype=torch.int32, device="cuda:0")
    assert eng.inner_join_into(a, b, n_chrom, ra, rb) == n      # the one-call form (speculating context: early fill)
    st = eng.stats()
    assert st["count_fused"] == expect_fused, st
    got = torch.zeros(a.n, dtype=torch.int64, device="cuda:0")
    lo_a, hi_a, lo_b, hi_b, bad = 1 << 40, -1, 1 << 40, -1, 0
    chunk = 1 << 29
    for s in range(0, n, chunk):       # chunked: a 40 GB index_add would not fit beside the pairs
        xa, xb = ra[s:s + chunk].long(), rb[s:s + chunk].long()
        got.index_add_(0, xa, torch.ones(xa.shape[0], dtype=torch.int64, device="cuda:0"))
        lo_a, hi_a = min(lo_a, int(xa.min())), max(hi_a, int(xa.max()))
        lo_b, hi_b = min(lo_b, int(xb.min())), max(hi_b, int(xb.max()))
        sa, sb = xa[::97], xb[::97]
        ok = (a.chrom[sa] == b.chrom[sb]) & (a.start[sa] < b.end[sb]) & (a.end[sa] > b.start[sb])
        bad += int((~ok).sum())
        del xa, xb
    assert lo_a >= 0 and hi_a < a.n and lo_b >= 0 and hi_b < b.n and bad == 0
    assert bool((got == counts).all())
    del ra, rb, got
    semi = eng.semi_join(a, b, n_chrom)
    assert int(semi.shape[0]) == int((counts > 0).sum().item())
    return n, st


def test_join_past_2_pow_32_pairs_dense_tables_narrow_buckets():
    """35M peaks x 350M reads (3.5x the headline sizes): 4.95e9 pairs, 40 GB of output.  ~7,400 rows per 65,536
    positions: since round 4 the 350M-row side keeps the three-stage sort with buckets of 2^14 keys (~1,850 rows,
    three global passes), the fused count and -- in the one-call form -- the join in the bucket stage
    (VERDICT r03 "Next round" 5)."""
    from giql_amd.engine import HipEngine

    A = synth.make_table(35_000_000, 5, "peaks")
    B = synth.make_table(350_000_000, 6, "reads")
    eng = HipEngine(0)
    try:
        n, st = _check_big_join(eng, A, B, 24, expect_fused=True)
        assert n == 4_952_361_736 and st["join_form"] == "uniform_b"
        assert st["sort_local"] and st["bucket_bits"] == 14 and st["bucket_join"] and not st["sort_resorted"], st
    finally:
        eng.close()


def test_join_past_2_pow_32_pairs_four_pass_sort_count_kernel_and_fill(monkeypatch):
    """Past 2^32 pairs WITHOUT the three-stage sort (GIQL_HIP_NO_LOCAL_SORT=1: what tables beyond every bucket width,
    and round 1's pipeline, run): four global passes, the count kernel, the 64-bit scan, the partition and `k_fill`.
    6M long peaks (20-32 kb) x 100M reads, ~850 reads per peak."""
    r = np.random.default_rng(1606)
    ch, st_, _en = synth.make_table(6_000_000, 15, "peaks")
    ln = r.integers(20_000, 32_000, ch.shape[0]).astype(np.int32)
    st_ = np.maximum(st_ - 32_000, 0).astype(np.int32)
    A = (ch, st_, st_ + ln)
    B = synth.make_table(100_000_000, 6, "reads")
    eng = _engine(monkeypatch, GIQL_HIP_NO_LOCAL_SORT="1")
    try:
        n, st = _check_big_join(eng, A, B, 24, expect_fused=False)
        assert st["join_form"] == "uniform_b" and not st["sort_local"] and not st["bucket_join"]
    finally:
        eng.close()


def test_join_past_2_pow_32_pairs_fused_count_and_chained_scan():
    """6M long peaks (20-32 kb) x 100M reads: ~850 reads per peak, > 2^32 pairs with the fused count
    (bucket sort answering the bounds), the chained 64-bit scan and the in-scan partition."""
    from giql_amd.engine import HipEngine

    r = np.random.default_rng(1606)
    ch, st_, _en = synth.make_table(6_000_000, 15, "peaks")
    ln = r.integers(20_000, 32_000, ch.shape[0]).astype(np.int32)
    st_ = np.maximum(st_ - 32_000, 0).astype(np.int32)     # (keeps the ends inside the chromosomes)
    A = (ch, st_, st_ + ln)
    B = synth.make_table(100_000_000, 6, "reads")
    eng = HipEngine(0)
    try:
        n, st = _check_big_join(eng, A, B, 24, expect_fused=True)
        assert st["join_form"] == "uniform_b" and st["sort_local"] and st["fused_fill"]
    finally:
        eng.close()


# ---- the headline kernels at the headline size, on contexts of their own --------------------
@pytest.fixture(scope="module")
def cfg4():
    """BASELINE config 4 (10M peaks x 100M reads, seeds 5 / 6) with the oracle's sweep: pair count, the 64-bit
    multiset checksum in both argument orders and the per-row multiplicities (= count_overlaps)."""
    a = ora.Side(*synth.make_table(10_000_000, 5, "peaks"))
    b = ora.Side(*synth.make_table(100_000_000, 6, "reads"))
    wa, wb = ora.c_inner(a, b, "sweep")
    want = {"n": int(wa.shape[0]), "sum": ora.c_pairs_checksum(wa, wb), "sum_swapped": ora.c_pairs_checksum(wb, wa)}
    del wa, wb
    want["count"] = ora.c_count(a, b, "sweep")
    assert want["n"] == 404_376_266 == int(want["count"].sum())
    return a, b, want


@pytest.mark.parametrize("env,form", [
    ({}, "uniform_b"),                          # k_bucket_sort<1, 2>: the benchmarked kernel
    ({"GIQL_HIP_NO_UNIFORM": "1"}, "general"),  # k_bucket_sort<3, 3>: rows of any length
])
def test_config4_join_in_the_bucket_stage_on_a_fresh_context(monkeypatch, cfg4, env, form):
    """A context of its own (no guess inherited from an earlier, smaller test): the first join is plan + fill, the
    second is the ONE-CALL form in which the bucket stage writes the pairs -- asserted, not assumed -- and must give
    the oracle's pair count, multiset checksum and per-row multiplicities (bag semantics,
    tests/test_duckdb_iejoin.py:4514-4555 of the reference); then the same with the larger table first."""
    a, b, want = cfg4
    e = _engine(monkeypatch, **env)
    try:
        da, db = dev(a), dev(b)
        count = torch.from_numpy(want["count"]).to("cuda:0")
        for it in range(3):
            ra, rb = e.inner_join(da, db, 24)
            st = e.stats()
            assert st["join_form"] == form and st["sort_local"] and not st["sort_resorted"] and not st["swapped"], (it, st)
            if it >= 1:
                assert st["bucket_join"] and st["count_fused"], (it, st)
            assert int(ra.shape[0]) == want["n"] and e.pairs_checksum(ra, rb) == want["sum"], it
            assert bool((torch.bincount(ra, minlength=a.n) == count).all()), it
            assert int(rb.min()) >= 0 and int(rb.max()) < b.n
            del ra, rb
        for it in range(2):
            rb2, ra2 = e.inner_join(db, da, 24)       # the larger table first: planned with the sides exchanged
            st = e.stats()
            # (the stats speak in the caller's labels: the fixed-length table is now side A)
            assert st["join_form"] == {"uniform_b": "uniform_a"}.get(form, form) and st["swapped"] and st["n_a"] == b.n, (it, st)
            if it >= 1:
                assert st["bucket_join"], (it, st)
            assert int(ra2.shape[0]) == want["n"] and e.pairs_checksum(rb2, ra2) == want["sum_swapped"], it
            assert bool((torch.bincount(ra2, minlength=a.n) == count).all()), it
            del ra2, rb2
    finally:
        e.close()


def test_config4_through_transpile_and_execute(monkeypatch, cfg4):
    """The product API at the headline size: transpile(dialect="hip") + execute() on Arrow tables, twice on an engine
    of its own -- the second call takes the one-call join -- with the returned index pairs checked on the host."""
    pa = pytest.importorskip("pyarrow")
    from giql_amd.execute import execute
    from giql_amd.transpile import transpile

    a, b, want = cfg4
    t = {"peaks": pa.table({"chrom": pa.array(a.chrom), "start": pa.array(a.start), "end": pa.array(a.end)}),
         "reads": pa.table({"chrom": pa.array(b.chrom), "start": pa.array(b.start), "end": pa.array(b.end)})}
    plan = transpile("SELECT a.start, b.start AS s2 FROM peaks a JOIN reads b ON a.interval INTERSECTS b.interval",
                     tables=["peaks", "reads"], dialect="hip")
    e = _engine(monkeypatch)
    try:
        for it in range(2):
            ra, rb = execute(plan, t, engine=e, return_indices=True)
            assert ra.shape[0] == want["n"] and ora.c_pairs_checksum(ra, rb) == want["sum"], it
            del ra, rb
        assert e.stats()["bucket_join"]
        out = execute(plan, t, engine=e)       # ... and once with the projected columns gathered on the device
        assert out.num_rows == want["n"]
        import pyarrow.compute as pc
        assert pc.sum(out.column("start")).as_py() == int((a.start.astype(np.int64) * want["count"]).sum())
    finally:
        e.close()
