"""Parity of the HIP path (through the C ABI) against the oracle -- needs a GPU.

Bit-exact: every result is integer row indices / counts / distances.
"""

import ctypes

import numpy as np
import pytest

import _golden as G
from oracle import pyoracle as ora

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

KNOWN = G.load("known_answers.json")
FUZZ = G.load("fuzz_sqlite.json")


@pytest.fixture(scope="module")
def eng():
    from giql_amd.engine import HipEngine

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    e = HipEngine(0)
    yield e
    e.close()


@pytest.fixture()
def eng_fresh():
    """A context of its own (tests of what a context remembers between plans)."""
    from giql_amd.engine import HipEngine

    e = HipEngine(0)
    yield e
    e.close()


def dev(side: ora.Side):
    from giql_amd.engine import DeviceSide

    d = "cuda:0"
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.int32)).to(d)
    return DeviceSide(t(side.chrom), t(side.start), t(side.end), side.start_off, side.end_off)


def n_chrom_of(a, b):
    m = -1
    for s in (a, b):
        if s.n:
            m = max(m, int(s.chrom.max()))
    return m + 1


def gpu_inner(eng, a, b, n_chrom=None):
    ra, rb = eng.inner_join(dev(a), dev(b), n_chrom if n_chrom is not None else n_chrom_of(a, b))
    torch.cuda.synchronize()
    return ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy())


# ----------------------------------------------------------- golden vectors
@pytest.mark.parametrize("case", [c for c in KNOWN if c["kind"] == "inner"], ids=lambda c: c["name"])
def test_inner_known_answers(eng, case):
    a, b = G.sides_of(case)
    got = G.rows_of_pairs(case, gpu_inner(eng, a, b))
    want = sorted(tuple(r) for r in case["expected"])
    if case["mode"] == "contains":
        assert all(w in got for w in want)
    else:
        assert got == want


@pytest.mark.parametrize("case", [c for c in KNOWN if c["kind"] in ("semi", "anti")],
                         ids=lambda c: c["name"])
def test_semi_anti_known_answers(eng, case):
    a, b = G.sides_of(case)
    rows = eng.semi_anti(dev(a), dev(b), n_chrom_of(a, b), case["kind"] == "anti").cpu().numpy()
    assert G.rows_of_left(case, rows) == sorted(tuple(r) for r in case["expected"])


def _nearest(eng, a, b, n_chrom, out32, **kw):
    """NEAREST k = 1 through the int64 ABI (two arrays) or the 8-byte-record one (giql_hip_nearest32_dev)."""
    if not out32:
        return eng.nearest(a, b, n_chrom, **kw)
    rec = eng.nearest32(a, b, n_chrom, **kw)
    assert rec.dtype == torch.int32 and tuple(rec.shape) == (a.n, 2)
    return rec[:, 0].contiguous(), rec[:, 1].to(torch.int64)


@pytest.mark.parametrize("out32", [False, True], ids=["i64", "rec32"])
@pytest.mark.parametrize("case", [c for c in KNOWN if c["kind"] == "nearest"], ids=lambda c: c["name"])
def test_nearest_known_answers(eng, case, out32):
    a, b = G.sides_of(case)
    idx, dist = _nearest(eng, dev(a), dev(b), n_chrom_of(a, b), out32, signed=case["signed"],
                         max_distance=case["max_distance"])
    with_d = len(case["expected"][0]) == 4
    got = G.nearest_rows(case, idx.cpu().numpy(), dist.cpu().numpy(), with_d)
    assert got == sorted(tuple(r) for r in case["expected"])


@pytest.mark.parametrize("case", [c for c in FUZZ if c["kind"] == "join"], ids=lambda c: c["name"])
def test_join_fuzz_vs_sqlite(eng, case):
    a, b = G.sides_of(case)
    nc = max(n_chrom_of(a, b), 1)
    assert np.array_equal(gpu_inner(eng, a, b, nc), np.asarray(case["inner"], np.int64).reshape(-1, 2))
    da, db = dev(a), dev(b)
    assert np.array_equal(eng.count_overlaps(da, db, nc).cpu().numpy(), np.asarray(case["count"], np.int64))
    assert np.array_equal(eng.semi_join(da, db, nc).cpu().numpy().astype(np.int64),
                          np.asarray(case["semi"], np.int64))
    assert np.array_equal(eng.anti_join(da, db, nc).cpu().numpy().astype(np.int64),
                          np.asarray(case["anti"], np.int64))


@pytest.mark.parametrize("out32", [False, True], ids=["i64", "rec32"])
@pytest.mark.parametrize("case", [c for c in FUZZ if c["kind"] == "nearest"], ids=lambda c: c["name"])
def test_nearest_fuzz_vs_sqlite(eng, case, out32):
    a, b = G.sides_of(case)
    idx, dist = _nearest(eng, dev(a), dev(b), max(n_chrom_of(a, b), 1), out32, signed=case["signed"],
                         max_distance=case["max_distance"])
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    for i, exp in enumerate(case["expected"]):
        if not exp:
            assert idx[i] == -1, i
            continue
        _rid, bs, be, d = exp[0]
        j = int(idx[i])
        assert j >= 0, i
        assert (case["b"][j][1], case["b"][j][2], int(dist[i])) == (bs, be, d), i


# --------------------------------------------------- seeded random vs oracle
def rand_side(seed, n, n_chrom, max_start, max_len, min_len=1, enc=("0based", "half_open")):
    r = np.random.default_rng(seed)
    ch = r.integers(0, n_chrom, n).astype(np.int32)
    st = r.integers(0, max_start, n).astype(np.int32)
    ln = r.integers(min_len, max_len, n).astype(np.int32)
    so, eo = ora.ENCODING_OFFSETS[enc]
    return ora.Side(ch, st, st + ln, so, eo)


@pytest.mark.parametrize("na,nb,nch,ms,ml", [
    (1, 1, 1, 10, 5),
    (63, 65, 2, 500, 60),
    (4096, 4097, 3, 100_000, 300),         # exactly one / one+1 radix tiles
    (5000, 300_000, 24, 2_000_000, 500),
    (200_000, 150_000, 24, 50_000_000, 3000),
    (30_000, 30_000, 1, 40_000, 2000),     # dense: ~1500 matches per row
])
def test_inner_random_vs_oracle(eng, na, nb, nch, ms, ml):
    a = rand_side(100 + na, na, nch, ms, ml)
    b = rand_side(200 + nb, nb, nch, ms, ml)
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    got = gpu_inner(eng, a, b, nch)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("enc_a", list(ora.ENCODING_OFFSETS))
@pytest.mark.parametrize("enc_b", list(ora.ENCODING_OFFSETS))
def test_inner_all_encoding_pairs(eng, enc_a, enc_b):
    a = rand_side(1, 3000, 4, 20_000, 50, enc=enc_a)
    b = rand_side(2, 3000, 4, 20_000, 50, enc=enc_b)
    assert np.array_equal(gpu_inner(eng, a, b, 4), ora.sort_pairs(*ora.c_inner(a, b, "sweep")))


def test_inner_with_irregular_rows(eng):
    """Zero-length and inverted rows follow the literal predicate."""
    a = rand_side(5, 4000, 3, 3000, 40, min_len=-15)
    b = rand_side(6, 5000, 3, 3000, 40, min_len=-15)
    want = ora.sort_pairs(*ora.c_inner(a, b, "brute"))
    got = gpu_inner(eng, a, b, 3)
    assert np.array_equal(got, want)
    st = eng.stats()
    assert st["n_irregular_a"] > 0 and st["n_irregular_b"] > 0
    da, db = dev(a), dev(b)
    assert np.array_equal(eng.count_overlaps(da, db, 3).cpu().numpy(), ora.c_count(a, b, "brute"))
    assert np.array_equal(eng.semi_join(da, db, 3).cpu().numpy(), ora.c_semi_anti(a, b, False))
    assert np.array_equal(eng.anti_join(da, db, 3).cpu().numpy(), ora.c_semi_anti(a, b, True))


def test_duplicates_keep_multiplicity(eng):
    ch = np.zeros(2000, np.int32)
    a = ora.Side(ch, np.full(2000, 100, np.int32), np.full(2000, 200, np.int32))
    b = ora.Side(ch[:1500], np.full(1500, 150, np.int32), np.full(1500, 250, np.int32))
    got = gpu_inner(eng, a, b, 1)
    assert got.shape[0] == 2000 * 1500
    assert np.array_equal(got, ora.sort_pairs(*ora.c_inner(a, b, "sweep")))


def test_empty_and_disjoint(eng):
    e = ora.Side(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    a = rand_side(1, 100, 2, 1000, 50)
    assert gpu_inner(eng, e, a, 2).shape[0] == 0
    assert gpu_inner(eng, a, e, 2).shape[0] == 0
    assert gpu_inner(eng, e, e, 0).shape[0] == 0
    b = ora.Side(a.chrom + 2, a.start, a.end)  # disjoint chromosome sets
    assert gpu_inner(eng, a, b, 4).shape[0] == 0
    da, db, de = dev(a), dev(b), dev(e)
    assert eng.semi_join(da, db, 4).shape[0] == 0
    assert np.array_equal(eng.anti_join(da, db, 4).cpu().numpy(), np.arange(100))
    assert np.array_equal(eng.anti_join(da, de, 2).cpu().numpy(), np.arange(100))  # B empty
    assert eng.semi_join(de, da, 2).shape[0] == 0
    assert int(eng.count_overlaps(da, de, 2).sum()) == 0
    idx, _ = eng.nearest(da, db, 4)
    assert bool((idx == -1).all())


def test_extreme_coordinates(eng):
    """INT32_MAX ends with a +1 canonical offset must not overflow."""
    big = 2**31 - 1
    a = ora.Side(np.zeros(3, np.int32), np.array([big - 10, 0, 5], np.int32),
                 np.array([big, 10, big], np.int32), 0, 1)
    b = ora.Side(np.zeros(3, np.int32), np.array([big - 1, 3, big], np.int32),
                 np.array([big, 4, big], np.int32), 0, 1)
    assert np.array_equal(gpu_inner(eng, a, b, 1), ora.sort_pairs(*ora.c_inner(a, b, "brute")))


def test_bad_chrom_id_is_an_error(eng):
    from giql_amd import _lib

    a = rand_side(1, 100, 2, 1000, 50)
    with pytest.raises(_lib.GiqlHipError) as ei:
        eng.inner_join(dev(a), dev(a), 1)  # ids up to 1 but n_chrom = 1
    assert ei.value.code == _lib.GIQL_ERR_CHROM


def test_span_overflow_is_an_error_at_the_c_abi(eng):
    from giql_amd import _lib

    ch = np.arange(4, dtype=np.int32)
    a = ora.Side(ch, np.zeros(4, np.int32), np.full(4, 2**31 - 2, np.int32))
    with pytest.raises(_lib.GiqlHipError) as ei:
        eng.inner_plan(dev(a), dev(a), 4)
    assert ei.value.code == _lib.GIQL_ERR_SPAN


def test_genome_longer_than_32_bits_is_joined_by_chromosome_groups(eng):
    """Spans summing past 2^32: the engine splits the chromosomes into groups that
    fit the u32 axis and joins group by group -- same results as the oracle."""
    r = np.random.default_rng(9)
    n_chrom = 5

    def side(n, max_len):
        ch = r.integers(0, n_chrom, n).astype(np.int32)
        st = r.integers(0, 2**31 - 5000, n).astype(np.int32)
        st[: n // 2] = r.integers(0, 100_000, n // 2)            # a dense corner so rows overlap
        ln = r.integers(1, max_len, n).astype(np.int32)
        return ora.Side(ch, st, st + ln)

    a, b = side(20_000, 3000), side(30_000, 800)
    # make sure every chromosome reaches ~2^31 so the spans sum to ~1e10
    a.start[:n_chrom] = 2**31 - 4000
    a.end[:n_chrom] = 2**31 - 3000
    a.chrom[:n_chrom] = np.arange(n_chrom)
    da, db = dev(a), dev(b)
    assert sum(eng.chrom_spans(da, db, n_chrom)) > 2**32
    ra, rb = eng.inner_join(da, db, n_chrom)
    got = ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy())
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    assert want.shape[0] > 1000 and np.array_equal(got, want)
    assert np.array_equal(eng.semi_join(da, db, n_chrom).cpu().numpy(), ora.c_semi_anti(a, b, False))
    assert np.array_equal(eng.anti_join(da, db, n_chrom).cpu().numpy(), ora.c_semi_anti(a, b, True))
    assert np.array_equal(eng.count_overlaps(da, db, n_chrom).cpu().numpy(), ora.c_count(a, b, "sweep"))
    idx, dist = eng.nearest(da, db, n_chrom)
    oi, od = ora.c_nearest_k1(a, b, method="sweep")
    idx = idx.cpu().numpy()
    assert np.array_equal(dist.cpu().numpy(), od) and np.array_equal(idx >= 0, oi >= 0)
    m = oi >= 0
    assert np.array_equal(b.start[idx[m]], b.start[oi[m]]) and np.array_equal(b.end[idx[m]], b.end[oi[m]])


@pytest.mark.parametrize("na,nb,nch", [(50_000, 400_000, 24), (300_000, 20_000, 5)])
def test_semi_anti_count_random_vs_oracle(eng, na, nb, nch):
    a = rand_side(31, na, nch, 30_000_000, 1500)
    b = rand_side(32, nb, nch - 1, 30_000_000, 200)  # last chrom absent from B
    da, db = dev(a), dev(b)
    assert np.array_equal(eng.semi_join(da, db, nch).cpu().numpy(), ora.c_semi_anti(a, b, False))
    assert np.array_equal(eng.anti_join(da, db, nch).cpu().numpy(), ora.c_semi_anti(a, b, True))
    assert np.array_equal(eng.count_overlaps(da, db, nch).cpu().numpy(), ora.c_count(a, b, "sweep"))


@pytest.mark.parametrize("out32", [False, True], ids=["i64", "rec32"])
@pytest.mark.parametrize("signed,md", [(False, None), (True, None), (False, 500), (True, 2000)])
def test_nearest_random_vs_oracle(eng, signed, md, out32):
    a = rand_side(41, 60_000, 6, 5_000_000, 800)
    b = rand_side(42, 50_000, 5, 5_000_000, 800)
    idx, dist = _nearest(eng, dev(a), dev(b), 6, out32, signed=signed, max_distance=md)
    oi, od = ora.c_nearest_k1(a, b, signed=signed, max_distance=md, method="sweep")
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert np.array_equal(idx >= 0, oi >= 0)
    assert np.array_equal(dist, od)
    m = oi >= 0
    assert np.array_equal(b.start[idx[m]], b.start[oi[m]]) and np.array_equal(b.end[idx[m]], b.end[oi[m]])


def test_nearest32_edges(eng):
    """The 8-byte-record form: no target rows -> {-1, 0} everywhere; a distance past INT32_MAX is refused (the
    int64 entry point answers it); an empty reference table is fine."""
    from giql_amd import _lib
    from giql_amd.engine import DeviceSide

    a = ora.Side(np.zeros(3, np.int32), np.array([10, 50, 70], np.int32), np.array([20, 60, 80], np.int32))
    none = DeviceSide.from_numpy(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    rec = eng.nearest32(dev(a), none, 1).cpu().numpy()
    assert rec.tolist() == [[-1, 0]] * 3
    assert tuple(eng.nearest32(none, dev(a), 1).shape) == (0, 2)
    # a on chromosome 1 only: no target there
    b_other = ora.Side(np.ones(2, np.int32), np.array([5, 9], np.int32), np.array([8, 12], np.int32))
    assert eng.nearest32(dev(a), dev(b_other), 2).cpu().numpy().tolist() == [[-1, 0]] * 3
    far_a = ora.Side(np.zeros(1, np.int32), np.array([-2_000_000_000], np.int32), np.array([-1_999_999_990], np.int32))
    far_b = ora.Side(np.zeros(1, np.int32), np.array([2_000_000_000], np.int32), np.array([2_000_000_010], np.int32))
    i64 = eng.nearest(dev(far_a), dev(far_b), 1)
    assert int(i64[1][0]) == 2_000_000_000 + 1_999_999_990 + 1
    with pytest.raises(_lib.GiqlHipError, match="int32"):
        eng.nearest32(dev(far_a), dev(far_b), 1)


def test_nearest_rejects_inverted_rows(eng):
    from giql_amd import _lib

    a = ora.Side(np.zeros(2, np.int32), np.array([10, 50], np.int32), np.array([20, 40], np.int32))
    b = ora.Side(np.zeros(1, np.int32), np.array([5], np.int32), np.array([8], np.int32))
    with pytest.raises(_lib.GiqlHipError):
        eng.nearest(dev(a), dev(b), 1)


# ---------------------------------------------- BASELINE configs (moderate size)
def test_config2_1m_x_1m_single_chrom(eng):
    from giql_amd import synth

    a = ora.Side(*synth.make_single_chrom(1_000_000, 1, "peaks"))
    b = ora.Side(*synth.make_single_chrom(1_000_000, 2, "peaks"))
    ra, rb = eng.inner_join(dev(a), dev(b), 1)
    wa, wb = ora.c_inner(a, b, "sweep")
    assert ra.shape[0] == wa.shape[0] and ra.shape[0] > 8_000_000
    assert eng.pairs_checksum(ra, rb) == ora.c_pairs_checksum(wa, wb)
    assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), ora.sort_pairs(wa, wb))


def test_host_buffer_entry_points(eng):
    """The Arrow-host-buffer flavour of the C ABI (what a ctypes stub binds)."""
    from giql_amd import _lib

    L = _lib.load()
    a = rand_side(51, 20_000, 4, 1_000_000, 900)
    b = rand_side(52, 30_000, 4, 1_000_000, 300)

    def cs(s):
        return _lib.CSide(s.chrom.ctypes.data, s.start.ctypes.data, s.end.ctypes.data, s.n,
                          s.start_off, s.end_off)

    n = ctypes.c_int64(0)
    pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(L.giql_hip_inner(eng._h, cs(a), cs(b), 4, ctypes.byref(n), ctypes.byref(pa),
                                ctypes.byref(pb)))
    ra = np.ctypeslib.as_array(ctypes.cast(pa, ctypes.POINTER(ctypes.c_int32)), (n.value,)).copy()
    rb = np.ctypeslib.as_array(ctypes.cast(pb, ctypes.POINTER(ctypes.c_int32)), (n.value,)).copy()
    L.giql_hip_free_host(pa)
    L.giql_hip_free_host(pb)
    assert np.array_equal(ora.sort_pairs(ra, rb), ora.sort_pairs(*ora.c_inner(a, b, "sweep")))

    cnt = np.zeros(a.n, np.int64)
    _lib.check(L.giql_hip_count(eng._h, cs(a), cs(b), 4, cnt.ctypes.data))
    assert np.array_equal(cnt, ora.c_count(a, b, "sweep"))

    p = ctypes.c_void_p()
    _lib.check(L.giql_hip_semi_anti(eng._h, cs(a), cs(b), 4, 1, ctypes.byref(n), ctypes.byref(p)))
    rows = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_int32)), (max(n.value, 1),))[: n.value].copy()
    L.giql_hip_free_host(p)
    assert np.array_equal(rows, ora.c_semi_anti(a, b, True))

    idx = np.zeros(a.n, np.int32)
    dist = np.zeros(a.n, np.int64)
    _lib.check(L.giql_hip_nearest(eng._h, cs(a), cs(b), 4, 0, -1, idx.ctypes.data, dist.ctypes.data))
    assert np.array_equal(dist, ora.c_nearest_k1(a, b, method="sweep")[1])


def test_classic_three_launch_sort_path(monkeypatch):
    """GIQL_HIP_SORT=classic keeps the hist/scan/scatter radix passes usable."""
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_SORT", "classic")
    e = HipEngine(0)
    try:
        a = rand_side(61, 70_000, 7, 9_000_000, 1200)
        b = rand_side(62, 90_000, 7, 9_000_000, 400)
        ra, rb = e.inner_join(dev(a), dev(b), 7)
        got = ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy())
        assert np.array_equal(got, ora.sort_pairs(*ora.c_inner(a, b, "sweep")))
    finally:
        e.close()


def test_sorted_input_and_heavy_skew(eng):
    """Chromosome/start-sorted input (BED-like) and one chromosome-spanning row."""
    a = rand_side(71, 120_000, 5, 20_000_000, 900)
    order = np.lexsort((a.start, a.chrom))
    a = ora.Side(a.chrom[order], a.start[order], a.end[order])
    b = rand_side(72, 150_000, 5, 20_000_000, 300)
    b.start[0], b.end[0] = 0, 20_000_000  # overlaps every A row on its chromosome
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    assert np.array_equal(gpu_inner(eng, a, b, 5), want)


# ------------------------------------------- uniform-length form (fixed-length reads)
def uniform_side(seed, n, n_chrom, max_start, length):
    r = np.random.default_rng(seed)
    ch = r.integers(0, n_chrom, n).astype(np.int32)
    st = r.integers(0, max_start, n).astype(np.int32)
    return ora.Side(ch, st, st + np.int32(length))


@pytest.mark.parametrize("which", ["b", "a", "both"])
def test_uniform_length_form(eng, which):
    a = uniform_side(81, 40_000, 6, 3_000_000, 75) if which in ("a", "both") else rand_side(81, 40_000, 6, 3_000_000, 900)
    b = uniform_side(82, 300_000, 6, 3_000_000, 150) if which in ("b", "both") else rand_side(82, 300_000, 6, 3_000_000, 900)
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    assert np.array_equal(gpu_inner(eng, a, b, 6), want)
    form = eng.stats()["join_form"]
    assert form == {"b": "uniform_b", "a": "uniform_a", "both": "uniform_b"}[which]


def test_uniform_length_with_irregular_rows_and_encodings(eng):
    """Zero-length / inverted QUERY rows go through the literal path; the canonical
    offsets change the uniform length (closed intervals are one longer).  An irregular
    row on the fixed-length side itself makes that side non-uniform (general form)."""
    a = rand_side(83, 9000, 3, 40_000, 60, min_len=-5, enc=("1based", "closed"))
    b = uniform_side(84, 20_000, 3, 40_000, 30)
    b.end_off = 1  # 0-based closed: canonical length 31
    want = ora.sort_pairs(*ora.c_inner(a, b, "brute"))
    assert np.array_equal(gpu_inner(eng, a, b, 3), want)
    st = eng.stats()
    assert st["join_form"] == "uniform_b" and st["n_irregular_b"] == 0 and st["n_irregular_a"] > 0
    bad = np.random.default_rng(1).integers(0, b.n, 50)
    b.end[bad] = b.start[bad] - 3  # inverted rows on the fixed-length side
    want = ora.sort_pairs(*ora.c_inner(a, b, "brute"))
    assert np.array_equal(gpu_inner(eng, a, b, 3), want)
    st = eng.stats()
    assert st["join_form"] == "general" and st["n_irregular_b"] > 0 and st["n_irregular_a"] > 0


def test_uniform_form_can_be_disabled(monkeypatch):
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_NO_UNIFORM", "1")
    e = HipEngine(0)
    try:
        a = rand_side(85, 30_000, 4, 2_000_000, 700)
        b = uniform_side(86, 120_000, 4, 2_000_000, 150)
        ra, rb = e.inner_join(dev(a), dev(b), 4)
        assert e.stats()["join_form"] == "general"
        assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()),
                              ora.sort_pairs(*ora.c_inner(a, b, "sweep")))
    finally:
        e.close()


def test_uniform_length_at_chromosome_edges(eng):
    """q.start - L + 1 reaches below a chromosome's first key: no leak across chroms."""
    a = ora.Side(np.array([0, 1, 1, 2], np.int32), np.array([0, 0, 5, 0], np.int32),
                 np.array([10, 3, 9, 1000], np.int32))
    b = ora.Side(np.array([0, 0, 1, 1, 2], np.int32), np.array([0, 990, 0, 2, 0], np.int32),
                 np.array([100, 1090, 100, 102, 100], np.int32))
    assert np.array_equal(gpu_inner(eng, a, b, 3), ora.sort_pairs(*ora.c_inner(a, b, "brute")))
    assert eng.stats()["join_form"] == "uniform_b"


# ------------- per-row operators with a fixed-length B (no prefix max, one sorted array)
def _row_ops_equal(e, a, b, nch, method="sweep"):
    da, db = dev(a), dev(b)
    assert np.array_equal(e.semi_join(da, db, nch).cpu().numpy(), ora.c_semi_anti(a, b, False))
    form_semi = e.stats()["join_form"]
    assert np.array_equal(e.anti_join(da, db, nch).cpu().numpy(), ora.c_semi_anti(a, b, True))
    assert np.array_equal(e.count_overlaps(da, db, nch).cpu().numpy(), ora.c_count(a, b, method))
    return form_semi, e.stats()["join_form"]


def test_row_ops_fixed_length_b(eng_fresh):
    """SEMI / ANTI / COUNT against fixed-length reads: the literal predicate rewritten as a range
    of B starts; irregular A rows, all encodings of A, a closed-interval B (length + 1), rows at
    chromosome edges, an A-only chromosome; repeated calls (first reads the form back, later
    ones speculate on it)."""
    e = eng_fresh
    b = uniform_side(301, 200_000, 6, 3_000_000, 150)
    for k, enc in enumerate(ora.ENCODING_OFFSETS):
        a = rand_side(302 + k, 30_000, 7, 3_000_000, 900, min_len=-3, enc=enc)  # chrom 6 is A-only
        assert _row_ops_equal(e, a, b, 7, "brute") == ("uniform_b", "uniform_b")
    b.end_off = 1  # 0-based closed: canonical length 151
    a = rand_side(310, 30_000, 6, 3_000_000, 900)
    assert _row_ops_equal(e, a, b, 6) == ("uniform_b", "uniform_b")
    # chromosome edges: ranges reaching below a chromosome's first key must not leak
    a = ora.Side(np.array([0, 1, 1, 2, 2], np.int32), np.array([0, 0, 5, 0, 120], np.int32),
                 np.array([10, 3, 9, 1000, 121], np.int32))
    b = ora.Side(np.array([0, 0, 1, 1, 2], np.int32), np.array([0, 990, 0, 2, 0], np.int32),
                 np.array([100, 1090, 100, 102, 100], np.int32))
    assert _row_ops_equal(e, a, b, 3, "brute") == ("uniform_b", "uniform_b")


def test_row_ops_fixed_length_b_sorted_coarsely(monkeypatch):
    """SEMI / ANTI / COUNT against a fixed-length B sorted without its lowest digit, the rows sharing the upper 24 key
    bits looked at one by one (forced at every density here): ranges inside one such group, across many (long A
    rows), empty stretches, chromosome edges, every encoding of A, irregular A rows; the form is taken from the
    context's second call on (the first has no density to go by)."""
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_COARSE_MAX_GROUP_ROWS", "1e12")
    e = HipEngine(0)
    monkeypatch.delenv("GIQL_HIP_COARSE_MAX_GROUP_ROWS")
    try:
        b = uniform_side(341, 200_000, 6, 3_000_000, 150)
        a0 = rand_side(342, 30_000, 7, 3_000_000, 900)
        assert _row_ops_equal(e, a0, b, 7) == ("uniform_b", "uniform_b")
        for k, enc in enumerate(ora.ENCODING_OFFSETS):
            a = rand_side(343 + k, 30_000, 7, 3_000_000, 900, min_len=-3, enc=enc)  # chrom 6 is A-only
            assert _row_ops_equal(e, a, b, 7, "brute") == ("uniform_b", "uniform_b")
            assert e.stats()["coarse_b"]
        # long A rows (ranges over many buckets, some of them empty) on a sparse B
        bs = uniform_side(350, 3_000, 3, 30_000_000, 100)
        al = rand_side(351, 20_000, 3, 30_000_000, 400_000, min_len=1)
        assert _row_ops_equal(e, al, bs, 3) == ("uniform_b", "uniform_b")
        assert _row_ops_equal(e, al, bs, 3) == ("uniform_b", "uniform_b") and e.stats()["coarse_b"]
        # chromosome edges: ranges reaching below a chromosome's first key must not leak
        a = ora.Side(np.array([0, 1, 1, 2, 2], np.int32), np.array([0, 0, 5, 0, 120], np.int32),
                     np.array([10, 3, 9, 1000, 121], np.int32))
        b2 = ora.Side(np.array([0, 0, 1, 1, 2], np.int32), np.array([0, 990, 0, 2, 0], np.int32),
                      np.array([100, 1090, 100, 102, 100], np.int32))
        assert _row_ops_equal(e, a, b2, 3, "brute") == ("uniform_b", "uniform_b")
        # B loses its fixed length: the guess misses, the call is repeated in the general form
        b3 = rand_side(352, 150_000, 6, 3_000_000, 300)
        assert _row_ops_equal(e, a0, b3, 7)[1] == "general" and not e.stats()["coarse_b"]
    finally:
        e.close()


def test_row_ops_form_guess_misses_and_recovers(eng_fresh):
    """fixed-length B -> B with other lengths -> a different fixed length -> an irregular B row:
    a wrong "fixed-length" guess repeats the call, a wrong "general" guess is merely slower (the
    general form is exact on any input) and corrects itself on the next call; results stay exact."""
    e = eng_fresh
    a = rand_side(320, 40_000, 5, 2_000_000, 700)
    b1 = uniform_side(321, 150_000, 5, 2_000_000, 100)
    b2 = rand_side(322, 150_000, 5, 2_000_000, 300)
    b3 = uniform_side(323, 150_000, 5, 2_000_000, 36)
    b4 = uniform_side(324, 150_000, 5, 2_000_000, 36)
    b4.end[77] = b4.start[77] - 2
    for b, want in ((b1, "uniform_b"), (b1, "uniform_b"), (b2, "general"), (b2, "general"), (b3, "uniform_b"),
                    (b4, "general"), (b1, "uniform_b")):
        first, last = _row_ops_equal(e, a, b, 5)
        assert last == want and first in (want, "general")


def test_row_ops_fixed_length_can_be_disabled(monkeypatch):
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_NO_UNIFORM", "1")
    e = HipEngine(0)
    try:
        a = rand_side(330, 20_000, 4, 1_000_000, 500)
        b = uniform_side(331, 90_000, 4, 1_000_000, 150)
        assert _row_ops_equal(e, a, b, 4) == ("general", "general")
    finally:
        e.close()


# ------------- histogram in the span pass: the fixed-length side sorted from its raw columns
def test_span_histogram_form_runs_and_can_be_disabled(monkeypatch):
    """Larger side fixed-length, coordinates >= 0, <= 32 chromosomes: no linearize pass for that
    side (stats span_hist), first call (layout read back) and later calls (speculated) alike;
    GIQL_HIP_NO_SPAN_HIST=1 gives the same pairs through the linearize pass."""
    from giql_amd.engine import HipEngine

    a = rand_side(91, 50_000, 24, 40_000_000, 1500)
    b = uniform_side(92, 400_000, 24, 40_000_000, 150)
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    e = HipEngine(0)
    try:
        for _ in range(3):
            assert np.array_equal(gpu_inner(e, a, b, 24), want)
            st = e.stats()
            assert st["join_form"] == "uniform_b" and st["span_hist"]
    finally:
        e.close()
    monkeypatch.setenv("GIQL_HIP_NO_SPAN_HIST", "1")
    e = HipEngine(0)
    try:
        assert np.array_equal(gpu_inner(e, a, b, 24), want)
        assert e.stats()["join_form"] == "uniform_b" and not e.stats()["span_hist"]
    finally:
        e.close()


def test_span_histogram_layout_misses_and_recovers(eng_fresh):
    """The aligned layout is speculated on after the first plan; inputs it cannot hold -- a
    coordinate below 0 (1-based start 0), buckets past 255 (INT32_MAX-scale coordinates on many
    chromosomes), sorted input (wave-uniform high digits), a bad form guess -- re-plan and stay exact."""
    e = eng_fresh
    a = rand_side(93, 20_000, 8, 5_000_000, 800)
    b = uniform_side(94, 150_000, 8, 5_000_000, 100)
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    assert np.array_equal(gpu_inner(e, a, b, 8), want) and e.stats()["span_hist"]
    assert np.array_equal(gpu_inner(e, a, b, 8), want) and e.stats()["span_hist"]
    # 1-based starts with a 0 among them: canonical -1, the aligned layout does not hold
    b1 = ora.Side(b.chrom.copy(), b.start.copy(), b.end.copy(), start_off=-1, end_off=-1)  # 1-based half-open
    b1.start[:7] = 0
    b1.end[:7] = 100
    want1 = ora.sort_pairs(*ora.c_inner(a, b1, "sweep"))
    assert np.array_equal(gpu_inner(e, a, b1, 8), want1)
    assert e.stats()["join_form"] == "uniform_b" and not e.stats()["span_hist"]
    assert np.array_equal(gpu_inner(e, a, b1, 8), want1) and not e.stats()["span_hist"]
    # huge coordinates on 8 chromosomes: 8 x 128 buckets do not fit 255
    r = np.random.default_rng(95)
    big = r.integers(2_000_000_000, 2_100_000_000, b.n).astype(np.int32)
    b2 = ora.Side(b.chrom.copy(), big, big + np.int32(100))
    a2 = ora.Side(a.chrom.copy(), (a.start // 4 + 2_000_000_000).astype(np.int32),
                  (a.start // 4 + 2_000_000_000 + 5000).astype(np.int32))
    try:
        got = gpu_inner(e, a2, b2, 8)
    except Exception as exc:  # the tight layout does not fit 32 bits either: the documented error
        assert "span" in str(exc).lower()
    else:
        assert np.array_equal(got, ora.sort_pairs(*ora.c_inner(a2, b2, "sweep")))
    # chromosome- and position-sorted fixed-length side (wave-uniform high digits)
    order = np.lexsort((b.start, b.chrom))
    bs = ora.Side(b.chrom[order], b.start[order], b.end[order])
    e2_want = ora.sort_pairs(*ora.c_inner(a, bs, "sweep"))
    assert np.array_equal(gpu_inner(e, a, bs, 8), e2_want)
    assert np.array_equal(gpu_inner(e, a, bs, 8), e2_want)


def test_span_histogram_tile_edges(eng_fresh):
    """Row counts around the 8192-row sort tile and the span pass's 4-row unroll."""
    e = eng_fresh
    a = rand_side(96, 3000, 5, 300_000, 400)
    for n in (1, 63, 64, 8191, 8192, 8193, 16384 + 5, 70_001):
        b = uniform_side(97 + n, n, 5, 300_000, 36)
        want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
        got = gpu_inner(e, a, b, 5)
        assert np.array_equal(got, want), n


# ------------------------------------------- BASELINE full size (cfg 4), properties
def test_config4_full_size_properties(eng):
    """10M x 100M, 24 chromosomes: too large to sort-compare pair by pair in a
    test, so check size-independent properties of the full result on the GPU:
    every pair satisfies the literal predicate, no pair is repeated, the count
    equals the sum of per-row counts from the independent two-sorted-arrays COUNT
    kernel, per-chromosome pair counts equal the oracle's on two sampled
    chromosomes, the uniform and general forms agree (multiset checksum), and that checksum
    equals the oracle's over the whole 404M-pair result."""
    from giql_amd import synth
    from giql_amd.engine import DeviceSide, HipEngine
    import os

    ac, as_, ae = synth.make_table(10_000_000, 5, "peaks")
    bc, bs, be = synth.make_table(100_000_000, 6, "reads")
    a = DeviceSide.from_numpy(ac, as_, ae)
    b = DeviceSide.from_numpy(bc, bs, be)
    ra, rb = eng.inner_join(a, b, 24)
    n = int(ra.shape[0])
    assert eng.stats()["join_form"] == "uniform_b"
    assert 3.9e8 < n < 4.2e8
    la, lb_ = ra.long(), rb.long()
    # 1. predicate holds for every emitted pair
    ok = (a.chrom[la] == b.chrom[lb_]) & (a.start[la] < b.end[lb_]) & (a.end[la] > b.start[lb_])
    assert bool(ok.all())
    del ok
    # 2. completeness: total == sum of COUNT (independent kernels), per-row too
    counts = eng.count_overlaps(a, b, 24)
    assert int(counts.sum()) == n
    per_row = torch.bincount(la, minlength=a.n)
    assert torch.equal(per_row, counts)
    del per_row, counts
    # 3. no repeated pair (rows are distinct, so pairs must be)
    packed = (la << 32) | lb_
    assert int(torch.unique(packed).shape[0]) == n
    del packed
    # 4. two sampled chromosomes against the oracle (full pair set)
    for c in (20, 23):
        ma, mb = ac == c, bc == c
        oa, ob = ora.Side(ac[ma], as_[ma], ae[ma]), ora.Side(bc[mb], bs[mb], be[mb])
        wa, wb = ora.c_inner(oa, ob, "sweep")
        ia, ib = np.nonzero(ma)[0], np.nonzero(mb)[0]
        sel = (a.chrom[la] == c)
        got = ora.sort_pairs(ra[sel].cpu().numpy(), rb[sel].cpu().numpy())
        want = ora.sort_pairs(ia[wa], ib[wb])
        assert np.array_equal(got, want), c
    # 5. the general two-class form returns the same multiset
    chk = eng.pairs_checksum(ra, rb)
    del ra, rb, la, lb_
    os.environ["GIQL_HIP_NO_UNIFORM"] = "1"
    try:
        e2 = HipEngine(0)
        ga, gb = e2.inner_join(a, b, 24)
        assert e2.stats()["join_form"] == "general"
        assert int(ga.shape[0]) == n and e2.pairs_checksum(ga, gb) == chk
        e2.close()
    finally:
        del os.environ["GIQL_HIP_NO_UNIFORM"]
    # 6. the WHOLE pair multiset against the oracle: same count, same order-independent 64-bit
    #    checksum (the oracle's sort-merge over all host cores, a few seconds at this size)
    wa, wb = ora.c_inner(ora.Side(ac, as_, ae), ora.Side(bc, bs, be), "sweep")
    assert wa.shape[0] == n
    assert ora.c_pairs_checksum(wa, wb) == chk


# ---------------------------------------------------------------- projection (take)
@pytest.mark.parametrize("dtype", ["int8", "int16", "int32", "int64", "float32", "float64"])
@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 70_001])
def test_take_fixed_width_matches_numpy(eng, dtype, n):
    rng = np.random.default_rng(n + 7)
    n_rows = 5000
    col = (rng.integers(-100, 100, n_rows) if "int" in dtype else rng.standard_normal(n_rows)).astype(dtype)
    other = rng.integers(0, 2**31 - 1, n_rows).astype(np.int32)
    idx = rng.integers(0, n_rows, n).astype(np.int32)
    if n > 4:
        idx[[1, n - 1]] = -1  # NEAREST's "none": zero-filled
    d = lambda x: torch.from_numpy(x).cuda()
    got = eng.take([d(col), d(other)], d(idx))
    for src, g in zip((col, other), got):
        want = np.where(idx >= 0, src[np.maximum(idx, 0)], 0).astype(src.dtype)
        assert np.array_equal(g.cpu().numpy(), want)


def test_take_many_columns_and_unaligned_views(eng):
    rng = np.random.default_rng(5)
    n_rows, n = 3000, 4099
    cols = [rng.integers(0, 1 << 30, n_rows).astype(np.int32) for _ in range(11)]  # > 8: two launches
    idx_full = torch.from_numpy(rng.integers(0, n_rows, n + 1).astype(np.int32)).cuda()
    idx = idx_full[1:]  # 4-byte-aligned only: scalar path
    got = eng.take([torch.from_numpy(c).cuda() for c in cols], idx.contiguous() if not idx.is_contiguous() else idx)
    ih = idx.cpu().numpy()
    for c, g in zip(cols, got):
        assert np.array_equal(g.cpu().numpy(), c[ih])


def test_take_rejects_out_of_range_ids(eng):
    col = torch.arange(10, dtype=torch.int32).cuda()
    idx = torch.tensor([0, 3, 10], dtype=torch.int32).cuda()
    from giql_amd._lib import GiqlHipError

    with pytest.raises(GiqlHipError) as ei:
        eng.take([col], idx)
    assert ei.value.code == -1


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 10_000])
def test_take_utf8_matches_pyarrow(eng, n):
    pa = pytest.importorskip("pyarrow")
    rng = np.random.default_rng(n)
    words = ["", "a", "chr1", "gene_%d" % 7, "x" * 33, "y" * 200, "z" * 31, "é-ü"]
    vals = [words[i] + str(i) if i % 3 else words[i % len(words)] for i in rng.integers(0, len(words), 500)]
    arr = pa.array(vals, pa.string())
    off = np.frombuffer(arr.buffers()[1], dtype=np.int32, count=len(arr) + 1)
    data = np.frombuffer(arr.buffers()[2], dtype=np.uint8)
    idx = rng.integers(0, len(arr), n).astype(np.int32)
    if n > 2:
        idx[2] = -1
    o, dbytes = eng.take_utf8(torch.from_numpy(off.copy()).cuda(), torch.from_numpy(data.copy()).cuda(),
                                 torch.from_numpy(idx).cuda())
    want = [vals[i] if i >= 0 else "" for i in idx]
    got = pa.Array.from_buffers(pa.string(), n, [None, pa.py_buffer(o.cpu().numpy().tobytes()),
                                                  pa.py_buffer(dbytes.cpu().numpy().tobytes())]).to_pylist()
    assert got == want


# ------------------------------------------------------- residual predicates (select)
_NP_OPS = {"=": np.equal, "!=": np.not_equal, "<": np.less, "<=": np.less_equal, ">": np.greater,
           ">=": np.greater_equal}


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 2047, 2048, 2049, 100_003])
def test_select_pairs_matches_numpy(eng, n):
    rng = np.random.default_rng(n + 1)
    na, nb = 700, 900
    ca_i32 = rng.integers(-50, 50, na).astype(np.int32)
    ca_f64 = rng.standard_normal(na)
    va = rng.random(na) > 0.2
    cb_i64 = rng.integers(-50, 50, nb).astype(np.int64)
    cb_f32 = rng.standard_normal(nb).astype(np.float32)
    cb_u8 = rng.integers(0, 2, nb).astype(np.uint8)
    ia = rng.integers(0, na, n).astype(np.int32)
    ib = rng.integers(0, nb, n).astype(np.int32)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    dia, dib = d(ia), d(ib)
    cases = [
        ([(("a", d(ca_i32)), "<", ("b", d(cb_i64)))], ca_i32[ia] < cb_i64[ib]),
        ([(("a", d(ca_f64)), ">=", ("b", d(cb_f32)))], ca_f64[ia] >= cb_f32[ib].astype(np.float64)),
        ([(("a", d(ca_i32)), "!=", ("lit", 3)), (("b", d(cb_u8)), "=", ("lit", 1))], (ca_i32[ia] != 3) & (cb_u8[ib] == 1)),
        ([(("lit", 0.25), "<", ("a", d(ca_f64), d(va.astype(np.uint8))))], (0.25 < ca_f64[ia]) & va[ia]),
        ([(("a", d(ca_i32)), "<=", ("b", d(cb_f32)))], ca_i32[ia].astype(np.float64) <= cb_f32[ib].astype(np.float64)),
        ([], np.ones(n, bool)),
    ]
    for preds, want in cases:
        ga, gb = eng.select(preds, idx_a=dia, idx_b=dib, n_rows_a=na, n_rows_b=nb)
        assert np.array_equal(ga.cpu().numpy(), ia[want]) and np.array_equal(gb.cpu().numpy(), ib[want])


@pytest.mark.parametrize("op", sorted(_NP_OPS))
def test_select_rows_every_operator(eng, op):
    rng = np.random.default_rng(3)
    col = rng.integers(0, 10, 5000).astype(np.int32)
    got = eng.select([(("a", torch.from_numpy(col).cuda()), op, ("lit", 4))], n=5000, n_rows_a=5000, want=("a",))[0]
    assert np.array_equal(got.cpu().numpy(), np.nonzero(_NP_OPS[op](col, 4))[0])
    got = eng.select([(("b", torch.from_numpy(col).cuda()), op, ("lit", 4.5))], n=5000, n_rows_b=5000, want=("b",))[1]
    assert np.array_equal(got.cpu().numpy(), np.nonzero(_NP_OPS[op](col, 4.5))[0])


@pytest.mark.parametrize("seed", range(6))
def test_select_or_groups_and_null_tests_match_numpy(eng, seed):
    # giql_pred.group: an AND of clauses, a clause = neighbours sharing a non-zero group, OR-ed; IS [NOT] NULL read
    # the left operand's validity only; up to 16 predicates per call
    rng = np.random.default_rng(100 + seed)
    n, na, nb = 30_000, 900, 1100
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    cols = {"a": [rng.integers(0, 6, na).astype(np.int32), rng.standard_normal(na)],
            "b": [rng.integers(0, 6, nb).astype(np.int64), rng.standard_normal(nb).astype(np.float32)]}
    valid = {"a": rng.random(na) > 0.25, "b": rng.random(nb) > 0.25}
    ia, ib = rng.integers(0, na, n).astype(np.int32), rng.integers(0, nb, n).astype(np.int32)
    idx = {"a": ia, "b": ib}
    preds, want, group = [], np.ones(n, bool), 0
    while len(preds) < 16:
        width = int(min(rng.integers(1, 5), 16 - len(preds)))
        group += 1
        clause = np.zeros(n, bool)
        for _ in range(width):
            side = "ab"[rng.integers(0, 2)]
            which = int(rng.integers(0, 2))
            col, with_valid = cols[side][which], rng.random() < 0.6
            vals = col[idx[side]].astype(np.float64)
            ok = valid[side][idx[side]] if with_valid else np.ones(n, bool)
            spec = (side, d(col), d(valid[side].astype(np.uint8))) if with_valid else (side, d(col))
            kind = rng.integers(0, 4)
            if kind == 0:
                op = ["isnull", "notnull"][rng.integers(0, 2)]
                t = ~ok if op == "isnull" else ok
                preds.append((spec, op, ("lit", 0), group if width > 1 else 0))
            else:
                op = sorted(_NP_OPS)[rng.integers(0, 6)]
                lit = int(rng.integers(0, 6)) if which == 0 else float(rng.standard_normal())
                t = _NP_OPS[op](vals, lit) & ok
                preds.append((spec, op, ("lit", lit), group if width > 1 else 0))
            clause |= t
        want &= clause
    assert len(preds) == 16
    ga, gb = eng.select(preds, idx_a=d(ia), idx_b=d(ib), n_rows_a=na, n_rows_b=nb)
    assert np.array_equal(ga.cpu().numpy(), ia[want]) and np.array_equal(gb.cpu().numpy(), ib[want])
    with pytest.raises(ValueError, match="16"):
        eng.select(preds + preds[:1], idx_a=d(ia), idx_b=d(ib), n_rows_a=na, n_rows_b=nb)


def test_select_arithmetic_operands_match_numpy(eng):
    # giql_hip_select_expr_dev: postfix programs over columns and literals; integers stay 64-bit, `/` is a floating
    # division (NULL on a zero divisor), NULL propagates, LEAST / GREATEST skip NULLs
    rng = np.random.default_rng(77)
    n, na, nb = 50_000, 3000, 2500
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    a_s = rng.integers(0, 100_000, na).astype(np.int32)
    a_e = (a_s + rng.integers(1, 500, na)).astype(np.int32)
    b_s = rng.integers(0, 100_000, nb).astype(np.int32)
    b_e = (b_s + rng.integers(1, 500, nb)).astype(np.int32)
    a_sc = rng.integers(-3, 4, na).astype(np.int64)
    b_f = rng.standard_normal(nb)
    va, vb = rng.random(na) > 0.2, rng.random(nb) > 0.2
    ia, ib = rng.integers(0, na, n).astype(np.int32), rng.integers(0, nb, n).astype(np.int32)
    A_s, A_e, B_s, B_e = (("a", d(a_s)), ("a", d(a_e)), ("b", d(b_s)), ("b", d(b_e)))
    A_sc, B_f = ("a", d(a_sc), d(va.astype(np.uint8))), ("b", d(b_f), d(vb.astype(np.uint8)))
    s_a, e_a, s_b, e_b = (x.astype(np.int64) for x in (a_s[ia], a_e[ia], b_s[ib], b_e[ib]))
    ov = np.minimum(e_a, e_b) - np.maximum(s_a, s_b)
    sc, f, ok_a, ok_b = a_sc[ia], b_f[ib], va[ia], vb[ib]
    with np.errstate(divide="ignore", invalid="ignore"):
        cases = [
            # the overlap-fraction recipe on raw columns
            ([(("expr", ("-", ("least", A_e, B_e), ("greatest", A_s, B_s))), ">=",
               ("expr", ("*", ("lit", 0.5), ("-", A_e, A_s))))], ov >= 0.5 * (e_a - s_a)),
            # integer arithmetic, NULL propagation, a comparison with a plain column on the other side
            ([(("expr", ("+", ("*", A_sc, ("lit", 3)), ("neg", A_sc))), "<", B_f)], (sc * 3 - sc < f) & ok_a & ok_b),
            # division: floating, NULL on a zero divisor
            ([(("expr", ("/", ("-", A_e, A_s), A_sc)), ">", ("lit", 40))],
             (np.where(sc != 0, (e_a - s_a) / np.where(sc == 0, 1, sc), 0) > 40) & ok_a & (sc != 0)),
            # LEAST / GREATEST skip a NULL argument; ABS; three arguments
            ([(("expr", ("abs", ("-", ("greatest", A_sc, ("lit", 1), ("lit", -7)), ("least", B_f, ("lit", 0.25))))), "<=",
               ("lit", 1.5))],
             np.abs(np.where(ok_a, np.maximum(sc, 1), 1) - np.where(ok_b, np.minimum(f, 0.25), 0.25)) <= 1.5),
            # two expression predicates in one OR group beside a plain conjunct
            ([(("expr", ("-", A_e, A_s)), ">", ("lit", 400), 1), (("expr", ("*", B_f, B_f)), ">", ("lit", 2.0), 1),
              (A_sc, ">=", ("lit", 0))],
             (((e_a - s_a) > 400) | ((f * f > 2.0) & ok_b)) & (sc >= 0) & ok_a),
        ]
    for preds, want in cases:
        ga, gb = eng.select(preds, idx_a=d(ia), idx_b=d(ib), n_rows_a=na, n_rows_b=nb)
        assert np.array_equal(ga.cpu().numpy(), ia[want]) and np.array_equal(gb.cpu().numpy(), ib[want])
    # boolean programs (round 4): comparisons, IS [NOT] NULL, AND / OR / NOT under three-valued logic, kept when TRUE
    def k3(t, known):        # (value, known) pairs -> Kleene
        return t, known
    def k_and(x, y):
        val = x[0] & y[0]
        known = (x[1] & y[1]) | (x[1] & ~x[0]) | (y[1] & ~y[0])
        return val & known, known
    def k_or(x, y):
        known = (x[1] & y[1]) | (x[1] & x[0]) | (y[1] & y[0])
        val = ((x[0] & x[1]) | (y[0] & y[1]))
        return val, known
    def k_not(x):
        return ~x[0] & x[1], x[1]
    c1 = k3(sc < 2, ok_a)                               # a.score < 2
    c2 = k3(f > 0.3, ok_b)                              # b.f > 0.3
    c3 = k3((e_a - s_a) > 250, np.ones(n, bool))        # a.end - a.start > 250
    c4 = k3(sc * 1.0 >= f, ok_a & ok_b)                 # a.score >= b.f
    isnull_b = k3(~ok_b, np.ones(n, bool))
    prog_cases = [
        (("or", ("and", ("<", A_sc, ("lit", 2)), (">", B_f, ("lit", 0.3))), ("not", (">", ("-", A_e, A_s), ("lit", 250)))),
         k_or(k_and(c1, c2), k_not(c3))),
        (("not", ("or", (">=", A_sc, B_f), ("isnull", B_f))), k_not(k_or(c4, isnull_b))),
        (("and", ("or", ("<", A_sc, ("lit", 2)), ("notnull", B_f), (">=", A_sc, B_f)), ("not", ("and", (">", B_f, ("lit", 0.3)), ("<", A_sc, ("lit", 2))))),
         k_and(k_or(k_or(c1, k_not(isnull_b)), c4), k_not(k_and(c2, c1)))),
    ]
    for tree, (val, known) in prog_cases:
        want = val & known
        ga, gb = eng.select([(("expr", tree), "istrue", ("lit", 0))], idx_a=d(ia), idx_b=d(ib), n_rows_a=na, n_rows_b=nb)
        assert np.array_equal(ga.cpu().numpy(), ia[want]) and np.array_equal(gb.cpu().numpy(), ib[want])
        # ... and beside a plain conjunct
        ga, gb = eng.select([(("expr", tree), "istrue", ("lit", 0)), (A_sc, ">=", ("lit", 0))], idx_a=d(ia), idx_b=d(ib),
                            n_rows_a=na, n_rows_b=nb)
        w2 = want & (sc >= 0) & ok_a
        assert np.array_equal(ga.cpu().numpy(), ia[w2]) and np.array_equal(gb.cpu().numpy(), ib[w2])
    from giql_amd._lib import GiqlHipError
    deep = ("lit", 1)
    for _ in range(13):
        deep = ("+", ("lit", 1), deep)                      # right-nested: thirteen values live at once (twelve fit)
    with pytest.raises(GiqlHipError):
        eng.select([(("expr", deep), ">", ("lit", 0))], n=10, n_rows_a=10, want=("a",))
    deep = ("lit", 1)
    for _ in range(11):
        deep = ("+", ("lit", 1), deep)
    assert int(eng.select([(("expr", deep), ">", ("lit", 0))], n=10, n_rows_a=10, want=("a",))[0].shape[0]) == 10


def test_select_rejects_bad_ids_and_mark_flags(eng):
    from giql_amd._lib import GiqlHipError

    col = torch.arange(10, dtype=torch.int32).cuda()
    bad = torch.tensor([1, 10], dtype=torch.int32).cuda()
    with pytest.raises(GiqlHipError):
        eng.select([(("a", col), ">", ("lit", 0))], idx_a=bad, idx_b=bad, n_rows_a=10, n_rows_b=11)
    flags = eng.mark(torch.tensor([3, 3, 7], dtype=torch.int32).cuda(), 9)
    assert flags.cpu().tolist() == [0, 0, 0, 1, 0, 0, 0, 1, 0]
    with pytest.raises(GiqlHipError):
        eng.mark(torch.tensor([9], dtype=torch.int32).cuda(), 9)


# ------------------------------------------------------------------ CLUSTER / MERGE
CLUSTER = G.load("cluster_merge.json")


def _gpu_cluster_merge(eng, side, n_parts, distance):
    d = dev(side)
    ids = eng.cluster(d, n_parts, distance).cpu().numpy()
    c, s, e, n = (t.cpu().numpy() for t in eng.merge(d, n_parts, distance))
    return ids, list(zip(c, s, e, n))


@pytest.mark.parametrize("case", CLUSTER, ids=lambda c: c["name"])
def test_cluster_merge_golden(eng, case):
    side, parts = G.cluster_side(case)
    ids, merged = _gpu_cluster_merge(eng, side, len(parts), case["distance"])
    G.check_cluster_case(case, ids, merged)


@pytest.mark.parametrize("n,distance", [(1, 0), (5000, 0), (200_000, 0), (200_000, 75), (1_000_000, 1000)])
def test_cluster_merge_random_vs_oracle(eng, n, distance):
    rng = np.random.default_rng(n + distance)
    chrom = rng.integers(0, 5, n).astype(np.int32)
    start = rng.integers(0, 3_000_000, n).astype(np.int32)
    side = ora.Side(chrom, start, start + rng.integers(0, 400, n).astype(np.int32))  # zero-length rows too
    ids, merged = _gpu_cluster_merge(eng, side, 5, distance)
    assert np.array_equal(ids, ora.c_cluster(side, distance))
    c, s, e, cnt = ora.c_merge(side, distance)
    assert merged == list(zip(c, s, e, cnt))


def test_cluster_merge_edge_cases(eng):
    from giql_amd._lib import GiqlHipError

    empty = ora.Side(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    ids, merged = _gpu_cluster_merge(eng, empty, 0, 0)
    assert ids.size == 0 and merged == []
    # duplicates, contained intervals, book-ended rows, a chromosome with one row, INT32_MAX ends
    rows = [(0, 10, 20)] * 3 + [(0, 0, 1000), (0, 20, 30), (0, 1000, 1001), (0, 1002, 1003), (2, 5, 2**31 - 1),
            (1, 7, 7), (1, 7, 9), (1, 8, 8)]
    side = ora.Side(*(np.array([r[k] for r in rows], np.int32) for k in range(3)))
    ids, merged = _gpu_cluster_merge(eng, side, 3, 0)
    assert np.array_equal(ids, ora.c_cluster(side, 0))
    assert merged == list(zip(*ora.c_merge(side, 0)))
    ids1, _ = _gpu_cluster_merge(eng, side, 3, 1)
    assert np.array_equal(ids1, ora.c_cluster(side, 1)) and ids1[6] == ids1[5]  # gap of 1 bridged by distance 1
    inverted = ora.Side(np.zeros(2, np.int32), np.array([5, 50], np.int32), np.array([9, 40], np.int32))
    with pytest.raises(GiqlHipError):
        eng.cluster(dev(inverted), 1, 0)
    with pytest.raises(GiqlHipError):
        eng.merge(dev(inverted), 1, 0)


def test_sort_falls_back_to_ticket_order_after_a_lookback_timeout(monkeypatch):
    # the default tile order assumes in-order workgroup dispatch; a (here: injected) look-back
    # timeout must repeat the call in the assumption-free ticket order and still be exact
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_INJECT_TIMEOUT", "1")
    e = HipEngine(0)
    monkeypatch.delenv("GIQL_HIP_INJECT_TIMEOUT")
    try:
        rng = np.random.default_rng(8)
        def side(n):
            s = rng.integers(0, 2_000_000, n).astype(np.int32)
            return ora.Side(rng.integers(0, 3, n).astype(np.int32), s, s + rng.integers(1, 300, n).astype(np.int32))
        a, b = side(40_000), side(70_000)
        assert e.stats()["sort_tile_order"] == 2
        ra, rb = e.inner_join(dev(a), dev(b), 3)
        st = e.stats()
        assert (st["sort_tile_order"], st["sort_order_fallbacks"]) == (0, 1)
        assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), ora.sort_pairs(*ora.c_inner(a, b, "sweep")))
        assert np.array_equal(e.count_overlaps(dev(a), dev(b), 3).cpu().numpy(), ora.c_count(a, b, "sweep"))
        assert e.stats()["sort_order_fallbacks"] == 1  # stays in ticket order, no further retries
    finally:
        e.close()


@pytest.mark.parametrize("help_after", ["0", "3"])
def test_sort_progress_does_not_depend_on_dispatch_order(monkeypatch, help_after):
    # with help_after ~ 0 every block that sees an unpublished predecessor abandons its tile and
    # computes the predecessor itself (tiles are computed several times over): results stay exact
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_OS_HELP_AFTER", help_after)
    e = HipEngine(0)
    monkeypatch.delenv("GIQL_HIP_OS_HELP_AFTER")
    try:
        rng = np.random.default_rng(21)
        def side(n):
            s = rng.integers(0, 50_000_000, n).astype(np.int32)
            return ora.Side(rng.integers(0, 4, n).astype(np.int32), s, s + rng.integers(1, 300, n).astype(np.int32))
        a, b = side(300_000), side(1_500_000)   # 183 tiles on the B side
        ra, rb = e.inner_join(dev(a), dev(b), 4)
        assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), ora.sort_pairs(*ora.c_inner(a, b, "sweep")))
        assert np.array_equal(e.semi_join(dev(a), dev(b), 4).cpu().numpy(), ora.c_semi_anti(a, b, False))
        assert e.stats()["sort_order_fallbacks"] == 0
    finally:
        e.close()


@pytest.mark.parametrize("n_chrom", [1024, 1025, 5000])
def test_many_chromosomes(eng, n_chrom):
    # scaffolds / contigs: more partitions than the LDS tables of the min/max and linearise
    # kernels hold (their global-memory paths), every operator against the oracle
    rng = np.random.default_rng(n_chrom)
    def side(n):
        s = rng.integers(0, 20_000, n).astype(np.int32)
        return ora.Side(rng.integers(0, n_chrom, n).astype(np.int32), s, s + rng.integers(1, 400, n).astype(np.int32))
    a, b = side(60_000), side(90_000)
    assert np.array_equal(gpu_inner(eng, a, b, n_chrom), ora.sort_pairs(*ora.c_inner(a, b, "sweep")))
    assert np.array_equal(eng.semi_join(dev(a), dev(b), n_chrom).cpu().numpy(), ora.c_semi_anti(a, b, False))
    assert np.array_equal(eng.count_overlaps(dev(a), dev(b), n_chrom).cpu().numpy(), ora.c_count(a, b, "sweep"))
    idx, dist = eng.nearest(dev(a), dev(b), n_chrom)
    _, od = ora.c_nearest_k1(a, b, method="sweep")
    assert np.array_equal(dist.cpu().numpy(), od)
    assert np.array_equal(eng.cluster(dev(a), n_chrom, 10).cpu().numpy(), ora.c_cluster(a, 10))


def test_join_form_speculation_survives_changing_inputs(eng):
    # a context speculates on its previous form decision (uniform_b / uniform_a / general, and the
    # fixed length) and validates it at the end of the plan: every switch must still be exact
    seq = [
        ("uniform_b", rand_side(201, 30_000, 4, 2_000_000, 700), uniform_side(202, 200_000, 4, 2_000_000, 150)),
        ("uniform_b", rand_side(203, 30_000, 4, 2_000_000, 700), uniform_side(204, 200_000, 4, 2_000_000, 150)),
        ("uniform_b", rand_side(205, 30_000, 4, 2_000_000, 700), uniform_side(206, 200_000, 4, 2_000_000, 90)),   # other length
        ("general", rand_side(207, 30_000, 4, 2_000_000, 700), rand_side(208, 100_000, 4, 2_000_000, 300)),
        ("uniform_a", uniform_side(209, 150_000, 4, 2_000_000, 60), rand_side(210, 40_000, 4, 2_000_000, 500)),
        ("general", rand_side(211, 30_000, 4, 2_000_000, 700), rand_side(212, 100_000, 4, 2_000_000, 300)),
        ("uniform_b", rand_side(213, 30_000, 4, 2_000_000, 700), uniform_side(214, 200_000, 4, 2_000_000, 150)),
    ]
    for form, a, b in seq:
        assert np.array_equal(gpu_inner(eng, a, b, 4), ora.sort_pairs(*ora.c_inner(a, b, "sweep"))), form
        assert eng.stats()["join_form"] == form


def test_nearest_equal_starts_short_runs_and_pileups():
    # NEAREST orders ties by (start, end): short runs of equal starts are fixed in place after one
    # sort, a pile-up (> 32 equal starts) makes the context switch to the two-sort plan
    from giql_amd.engine import HipEngine

    e = HipEngine(0)
    try:
        rng = np.random.default_rng(77)
        def check(b):
            n = 4000
            s = rng.integers(0, 3000, n).astype(np.int32)
            a = ora.Side(rng.integers(0, 2, n).astype(np.int32), s, s + rng.integers(0, 40, n).astype(np.int32))
            idx, dist = e.nearest(dev(a), dev(b), 2, signed=True)
            oi, od = ora.c_nearest_k1(a, b, signed=True, method="sweep")
            j = idx.cpu().numpy()
            assert np.array_equal(dist.cpu().numpy(), od)
            ok = j >= 0
            assert np.array_equal(ok, oi >= 0)
            assert np.array_equal(b.start[j[ok]], b.start[oi[ok]]) and np.array_equal(b.end[j[ok]], b.end[oi[ok]])
        # runs of 1-6 equal starts with shuffled ends
        st = np.repeat(rng.integers(0, 3000, 800), rng.integers(1, 7, 800)).astype(np.int32)
        b = ora.Side(rng.integers(0, 2, st.size).astype(np.int32), st, st + rng.integers(1, 60, st.size).astype(np.int32))
        check(b)
        # a pile-up: 200 rows starting at the same base
        st2 = np.concatenate([st, np.full(200, 1500, np.int32)])
        b2 = ora.Side(np.concatenate([b.chrom, np.zeros(200, np.int32)]), st2,
                      st2 + np.concatenate([b.end - b.start, rng.integers(1, 500, 200).astype(np.int32)]))
        check(b2)
        check(b)  # stays correct (two-sort plan from now on)
    finally:
        e.close()


@pytest.mark.parametrize("n_tiles,extra", [(1, 0), (1, 1), (63, 5), (64, 0), (64, 1), (65, 8191), (127, 17), (128, 0), (129, 4096), (200, 1)])
def test_sort_tile_order_edges(eng, n_tiles, extra):
    # the sort maps blocks to tiles through an XCD-aware permutation inside groups of 64 tiles, with an
    # identity tail: exercise tile counts around the group size and partial last tiles (8192-row tiles)
    n_b = n_tiles * 8192 + extra - (8192 if extra else 0) if n_tiles > 1 or extra else 8192
    n_b = max(n_b, 1)
    rng = np.random.default_rng(n_tiles * 10007 + extra)
    sb = rng.integers(0, 40_000_000, n_b).astype(np.int32)
    b = ora.Side(rng.integers(0, 3, n_b).astype(np.int32), sb, sb + rng.integers(1, 200, n_b).astype(np.int32))
    sa = rng.integers(0, 40_000_000, 20_000).astype(np.int32)
    a = ora.Side(rng.integers(0, 3, 20_000).astype(np.int32), sa, sa + rng.integers(1, 3000, 20_000).astype(np.int32))
    assert np.array_equal(eng.count_overlaps(dev(a), dev(b), 3).cpu().numpy(), ora.c_count(a, b, "sweep"))
    assert np.array_equal(gpu_inner(eng, a, b, 3), ora.sort_pairs(*ora.c_inner(a, b, "sweep")))


def test_group_rows_and_segment_sum(eng):
    rng = np.random.default_rng(31)
    n = 120_000
    base = rng.integers(0, 4000, (3000, 2))
    pick = rng.integers(0, 3000, n)
    chrom = rng.integers(0, 3, 3000)[pick].astype(np.int32)
    start = base[pick, 0].astype(np.int32)
    end = (base[pick, 0] + base[pick, 1] % 90 - 5).astype(np.int32)   # some zero-length / inverted rows too
    side = ora.Side(chrom, start, end)
    gid, rep = eng.group_rows(dev(side), 3)
    gid, rep = gid.cpu().numpy(), rep.cpu().numpy()
    keys = {}
    for i in range(n):
        keys.setdefault((int(chrom[i]), int(start[i]), int(end[i])), []).append(i)
    assert rep.size == len(keys) and gid.min() == 0 and gid.max() == len(keys) - 1
    for rows in list(keys.values())[:2000]:
        assert len({int(gid[r]) for r in rows}) == 1
    assert len(set(gid.tolist())) == len(keys)
    assert all((int(chrom[r]), int(start[r]), int(end[r])) in keys and gid[r] == g for g, r in enumerate(rep))
    vals = rng.integers(0, 1000, n).astype(np.int64)
    sums = eng.segment_sum(torch.from_numpy(vals).cuda(), torch.from_numpy(gid).cuda(), rep.size).cpu().numpy()
    assert np.array_equal(sums, np.bincount(gid, weights=vals, minlength=rep.size).astype(np.int64))
    # a pile-up of equal starts (two-sort plan) still groups exactly
    s2 = ora.Side(np.zeros(500, np.int32), np.full(500, 77, np.int32), (77 + rng.integers(0, 7, 500)).astype(np.int32))
    g2, r2 = eng.group_rows(dev(s2), 1)
    assert r2.shape[0] == len(set(s2.end.tolist()))


@pytest.mark.parametrize("seed", range(24))
def test_randomized_sweep_all_operators(eng, seed):
    # hypothesis-style differential sweep: random sizes (incl. 0 / 1), chromosome counts, coordinate
    # ranges from "everything collides" to sparse, zero-length / inverted rows on odd seeds, random
    # encodings -- every operator against the oracle
    r = np.random.default_rng(1000 + seed)
    encs = list(ora.ENCODING_OFFSETS)
    n_chrom = int(r.integers(1, 7))
    max_start = int(r.choice([30, 2_000, 400_000, 80_000_000]))
    max_len = int(r.choice([3, 80, 5_000]))
    min_len = -3 if seed % 2 else 1
    na, nb = (int(r.choice([0, 1, 2, 65, 700, 9_000])) for _ in range(2))
    a = rand_side(5000 + seed, na, n_chrom, max_start, max_len + 1, min_len=min_len, enc=encs[int(r.integers(0, 4))])
    b = rand_side(6000 + seed, nb, n_chrom, max_start, max_len + 1, min_len=min_len, enc=encs[int(r.integers(0, 4))])
    assert np.array_equal(gpu_inner(eng, a, b, n_chrom), ora.sort_pairs(*ora.c_inner(a, b, "brute")))
    assert np.array_equal(eng.semi_join(dev(a), dev(b), n_chrom).cpu().numpy(), ora.c_semi_anti(a, b, False))
    assert np.array_equal(eng.anti_join(dev(a), dev(b), n_chrom).cpu().numpy(), ora.c_semi_anti(a, b, True))
    assert np.array_equal(eng.count_overlaps(dev(a), dev(b), n_chrom).cpu().numpy(), ora.c_count(a, b, "brute"))
    if min_len >= 0:  # NEAREST / CLUSTER / MERGE need start <= end
        signed = bool(seed % 3 == 0)
        md = None if seed % 4 else 50
        idx, dist = eng.nearest(dev(a), dev(b), n_chrom, signed=signed, max_distance=md)
        oi, od = ora.c_nearest_k1(a, b, signed=signed, max_distance=md, method="brute")
        assert np.array_equal(dist.cpu().numpy(), od)
        j = idx.cpu().numpy()
        assert np.array_equal(j >= 0, oi >= 0)
        ok = j >= 0
        assert np.array_equal(b.start[j[ok]], b.start[oi[ok]]) and np.array_equal(b.end[j[ok]], b.end[oi[ok]])
        raw = ora.Side(a.chrom, a.start, a.end)  # CLUSTER / MERGE read raw coordinates
        d = int(r.choice([0, 7, 900]))
        assert np.array_equal(eng.cluster(dev(raw), n_chrom, d).cpu().numpy(), ora.c_cluster(raw, d))
        c, s, e, n = (t.cpu().numpy() for t in eng.merge(dev(raw), n_chrom, d))
        assert list(zip(c, s, e, n)) == list(zip(*ora.c_merge(raw, d)))
    gid, rep = eng.group_rows(dev(a), max(n_chrom, 1))
    assert rep.shape[0] == len({(int(x), int(y), int(z)) for x, y, z in zip(a.chrom, a.start, a.end)})


def test_fused_inner_join_into_caller_buffers():
    # plan + fill in one call: the second call on a context launches the fill inside the plan (no sync
    # in between) when its guesses hold; every other situation must end in the same exact pairs
    from giql_amd._lib import GiqlHipError
    from giql_amd.engine import HipEngine

    e = HipEngine(0)
    try:
        def run(a, b, cap):
            ra = torch.empty(cap, dtype=torch.int32, device="cuda")
            rb = torch.empty(cap, dtype=torch.int32, device="cuda")
            n = e.inner_join_into(dev(a), dev(b), 4, ra, rb)
            return ora.sort_pairs(ra[:n].cpu().numpy(), rb[:n].cpu().numpy())
        a1, b1 = rand_side(301, 20_000, 4, 1_500_000, 700), uniform_side(302, 150_000, 4, 1_500_000, 120)
        want1 = ora.sort_pairs(*ora.c_inner(a1, b1, "sweep"))
        cap = want1.shape[0] + 5000
        assert np.array_equal(run(a1, b1, cap), want1)      # first plan: nothing to guess from
        assert np.array_equal(run(a1, b1, cap), want1)      # fused
        a2, b2 = rand_side(303, 25_000, 4, 1_500_000, 700), uniform_side(304, 140_000, 4, 1_500_000, 120)
        want2 = ora.sort_pairs(*ora.c_inner(a2, b2, "sweep"))
        assert np.array_equal(run(a2, b2, want2.shape[0]), want2)   # fused, exact capacity
        with pytest.raises(GiqlHipError) as ei:             # too small: the plan stays valid
            run(a2, b2, want2.shape[0] - 1)
        assert ei.value.code == -6 and e.last_pairs == want2.shape[0]
        ra = torch.empty(want2.shape[0], dtype=torch.int32, device="cuda")
        rb = torch.empty_like(ra)
        e.inner_fill(ra, rb)
        assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), want2)
        a3 = rand_side(305, 20_000, 4, 1_500_000, 700, min_len=-3)   # irregular query rows: guess fails
        want3 = ora.sort_pairs(*ora.c_inner(a3, b1, "sweep"))
        assert np.array_equal(run(a3, b1, want3.shape[0] + 10), want3)
        b4 = rand_side(306, 90_000, 4, 1_500_000, 400)               # general form
        want4 = ora.sort_pairs(*ora.c_inner(a1, b4, "sweep"))
        assert np.array_equal(run(a1, b4, want4.shape[0] + 10), want4)
        assert np.array_equal(run(a1, b1, cap), want1)
        assert np.array_equal(run(a1, b1, cap), want1)
        empty = ora.Side(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
        assert run(empty, b1, 16).shape[0] == 0
    finally:
        e.close()


def test_fused_fill_never_writes_past_a_short_buffer():
    # the fill launched inside the plan has a grid bounded by the capacity but learns the pair
    # count on the device: with more pairs than capacity it must write NOTHING past the buffers
    # (its last tile would otherwise overrun by up to a whole tile), and the call must report
    # GIQL_ERR_CAPACITY with the plan still valid
    from giql_amd._lib import GIQL_ERR_CAPACITY, GiqlHipError
    from giql_amd.engine import HipEngine

    e = HipEngine(0)
    try:
        a, b = rand_side(311, 20_000, 4, 1_500_000, 700), uniform_side(312, 150_000, 4, 1_500_000, 120)
        want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
        n = want.shape[0]
        big = torch.empty((2, n + 64), dtype=torch.int32, device="cuda")
        assert e.inner_join_into(dev(a), dev(b), 4, big[0], big[1]) == n   # first plan (nothing fused yet)
        assert e.inner_join_into(dev(a), dev(b), 4, big[0], big[1]) == n   # fused: the guesses hold
        guard = 40_000                                                      # > two fill tiles
        for cap in (n // 3 + 17, 1000, n - 1):                              # none a multiple of the 16384-pair tile
            buf = torch.full((2 * (cap + guard),), -7, dtype=torch.int32, device="cuda")
            ra, rb = buf[:cap], buf[cap + guard:2 * cap + guard]
            with pytest.raises(GiqlHipError) as ei:
                e.inner_join_into(dev(a), dev(b), 4, ra, rb)
            assert ei.value.code == GIQL_ERR_CAPACITY and e.last_pairs == n
            torch.cuda.synchronize()
            assert bool((buf[cap:cap + guard] == -7).all()) and bool((buf[2 * cap + guard:] == -7).all())
        ra = torch.empty(n, dtype=torch.int32, device="cuda")
        rb = torch.empty_like(ra)
        e.inner_fill(ra, rb)                                                # the plan is still valid
        assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), want)
    finally:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["general", "uniform"])
def test_inner_larger_side_first_is_planned_swapped(form):
    """A call with the larger table as A is planned with the sides exchanged (B = the larger side):
    same pair set in the caller's labels, stats labelled back, both the plan+fill and the one-call path."""
    import torch

    from giql_amd.engine import DeviceSide, HipEngine

    r = np.random.default_rng(77)

    def side(n, fixed):
        ch = r.integers(0, 5, n).astype(np.int32)
        st = r.integers(0, 3_000_000, n).astype(np.int32)
        ln = np.full(n, 120, np.int32) if fixed else r.integers(1, 700, n).astype(np.int32)
        return ora.Side(ch, st, st + ln)

    big = side(90_000, form == "uniform")
    small = side(7_000, False)
    small.end[:5] = small.start[:5]  # a few irregular rows on the small side (general form only keeps them apart)
    if form == "uniform":
        small = side(7_000, False)
    want = ora.sort_pairs(*ora.c_inner(big, small, "sweep"))
    eng = HipEngine(0)
    da = DeviceSide.from_numpy(big.chrom, big.start, big.end)
    db = DeviceSide.from_numpy(small.chrom, small.start, small.end)
    for _ in range(3):  # the third call runs on the context's guesses
        ra, rb = eng.inner_join(da, db, 5)
        st = eng.stats()
        assert st["swapped"] and st["n_a"] == 90_000 and st["n_b"] == 7_000
        assert st["join_form"] == ("uniform_a" if form == "uniform" else "general")
        assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), want)
    out_a = torch.empty(want.shape[0] + 100, dtype=torch.int32, device="cuda")
    out_b = torch.empty_like(out_a)
    n = eng.inner_join_into(da, db, 5, out_a, out_b)
    assert n == want.shape[0]
    assert np.array_equal(ora.sort_pairs(out_a[:n].cpu().numpy(), out_b[:n].cpu().numpy()), want)
    if form == "uniform":  # the exported plan names the caller's sides, row ids and offsets included
        n = eng.inner_plan(da, db, 5)
        q_is_a, n_q, n_s = eng.plan_sizes()
        assert (q_is_a, n_q, n_s) == (False, 7_000, 90_000)
        i32 = dict(dtype=torch.int32, device="cuda")
        q_rid, lo, cnt, s_rid = (torch.empty(n_q, **i32), torch.empty(n_q, **i32), torch.empty(n_q, **i32),
                                 torch.empty(n_s, **i32))
        eng.plan_export(q_rid, lo, cnt, s_rid, rid_add_a=1000, rid_add_b=50)   # global ids = local + shard base
        assert int(q_rid.min()) >= 50 and int(q_rid.max()) < 50 + 7_000           # the query side is the caller's B
        assert int(s_rid.min()) >= 1000 and int(s_rid.max()) < 1000 + 90_000
        rq, rs = torch.empty(n, **i32), torch.empty(n, **i32)
        assert eng.fill_from_plan(q_rid, lo, cnt, s_rid, rq, rs, n_pairs_expected=n) == n
        got = ora.sort_pairs((rs - 1000).cpu().numpy(), (rq - 50).cpu().numpy())   # (row_a, row_b) in the caller's labels
        assert np.array_equal(got, want)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("local_sort", [False, True])
def test_general_form_sorts_larger_side_from_raw_columns(local_sort, monkeypatch):
    """General form (variable lengths): the larger side's (key, end, rid) sort starts from the raw columns
    (digit histogram in the span pass, no linearize pass) while that side holds no irregular row; an
    irregular row appearing under the same context makes the plan repeat itself the ordinary way."""
    from giql_amd.engine import DeviceSide, HipEngine

    if local_sort:
        monkeypatch.setenv("GIQL_HIP_LOCAL_MIN_ROWS", "1")
    r = np.random.default_rng(91)

    def side(n):
        ch = r.integers(0, 7, n).astype(np.int32)
        st = r.integers(0, 40_000_000, n).astype(np.int32)
        ln = r.integers(1, 5_000, n).astype(np.int32)
        return ora.Side(ch, st, st + ln)

    a, b = side(20_000), side(150_000)
    eng = HipEngine(0)
    da = DeviceSide.from_numpy(a.chrom, a.start, a.end)
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    for it in range(3):
        ra, rb = eng.inner_join(da, DeviceSide.from_numpy(b.chrom, b.start, b.end), 7)
        st = eng.stats()
        assert st["join_form"] == "general" and st["span_hist"], (it, st)
        assert st["sort_local"] == local_sort
        assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), want)
    # offsets on both columns (1-based closed input): keys and end keys built with them
    db1 = DeviceSide.from_numpy(b.chrom, b.start + 1, b.end, encoding=("1based", "closed"))
    ra, rb = eng.inner_join(da, db1, 7)
    assert eng.stats()["span_hist"]
    assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), want)
    # irregular rows on the larger side: the guess fails, the plan is repeated with the linearize pass
    b2 = ora.Side(b.chrom.copy(), b.start.copy(), b.end.copy())
    b2.end[::1000] = b2.start[::1000] - 3
    want2 = ora.sort_pairs(*ora.c_inner(a, b2, "sweep"))
    for it in range(2):
        ra, rb = eng.inner_join(da, DeviceSide.from_numpy(b2.chrom, b2.start, b2.end), 7)
        st = eng.stats()
        assert not st["span_hist"] and st["n_irregular_b"] == 150
        assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), want2)
    # ... and back
    ra, rb = eng.inner_join(da, DeviceSide.from_numpy(b.chrom, b.start, b.end), 7)
    assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), want)
    eng.close()


@pytest.mark.gpu
def test_host_buffer_entry_with_the_larger_table_first():
    """giql_hip_inner (host columns in, pinned host pairs out) plans an (80K, 6K) call with the sides
    exchanged like the device entry points; the pairs come back in the caller's labels."""
    from giql_amd.engine import HipEngine

    r = np.random.default_rng(31)

    def side(n):
        ch = r.integers(0, 4, n).astype(np.int32)
        st = r.integers(0, 2_000_000, n).astype(np.int32)
        return ora.Side(ch, st, st + r.integers(1, 900, n).astype(np.int32))

    a, b = side(80_000), side(6_000)
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    eng = HipEngine(0)
    for _ in range(2):
        ra, rb = eng.inner_join_host((a.chrom, a.start, a.end), (b.chrom, b.start, b.end), 4)
        assert eng.stats()["swapped"]
        assert np.array_equal(ora.sort_pairs(ra, rb), want)
    eng.close()


@pytest.mark.gpu
def test_host_outputs_come_from_a_pool_and_stay_valid_until_released(monkeypatch):
    """Library-owned pinned outputs: a released array is handed out again by the next call of a similar
    size (page-locking costs more than the copy), two results alive at once never share memory, and
    GIQL_HIP_HOST_POOL_MB=0 turns the reuse off."""
    import ctypes

    from giql_amd import _lib
    from giql_amd.engine import HipEngine

    r = np.random.default_rng(8)
    ch = r.integers(0, 3, 50_000).astype(np.int32)
    st = r.integers(0, 1_000_000, 50_000).astype(np.int32)
    en = (st + r.integers(1, 600, 50_000)).astype(np.int32)
    eng = HipEngine(0)
    L = eng._L

    def call():
        ca = _lib.CSide(ch.ctypes.data, st.ctypes.data, en.ctypes.data, 20_000, 0, 0)
        cb = _lib.CSide(ch[20_000:].ctypes.data, st[20_000:].ctypes.data, en[20_000:].ctypes.data, 30_000, 0, 0)
        n, pa_, pb_ = ctypes.c_int64(0), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(L.giql_hip_inner(eng._h, ctypes.byref(ca), ctypes.byref(cb), 3, ctypes.byref(n), ctypes.byref(pa_), ctypes.byref(pb_)))
        return n.value, pa_.value, pb_.value

    def pairs(n, pa_, pb_):
        view = lambda p: np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_int32)), shape=(n,)).copy()
        return ora.sort_pairs(view(pa_), view(pb_))

    want = ora.sort_pairs(*ora.c_inner(ora.Side(ch[:20_000], st[:20_000], en[:20_000]),
                                       ora.Side(ch[20_000:], st[20_000:], en[20_000:]), "sweep"))
    n1, a1, b1 = call()
    n2, a2, b2 = call()                      # the first result is still alive: four distinct arrays
    assert len({a1, b1, a2, b2}) == 4
    assert np.array_equal(pairs(n1, a1, b1), want) and np.array_equal(pairs(n2, a2, b2), want)
    for p in (a1, b1, a2, b2):
        L.giql_hip_free_host(ctypes.c_void_p(p))
    n3, a3, b3 = call()                      # released arrays are handed out again
    assert {a3, b3} <= {a1, b1, a2, b2} and np.array_equal(pairs(n3, a3, b3), want)
    L.giql_hip_free_host(ctypes.c_void_p(a3))
    L.giql_hip_free_host(ctypes.c_void_p(b3))
    eng.close()


@pytest.mark.gpu
def test_two_contexts_on_two_threads():
    """One context per thread (a context is not shared between threads; the error string is thread-local,
    the pinned-output pool is locked): two threads joining different tables at the same time, device and
    host-buffer entry points alike, each get their own oracle's answer."""
    import threading

    import torch

    from giql_amd.engine import DeviceSide, HipEngine

    def tables(seed, fixed):
        r = np.random.default_rng(seed)

        def side(n, f):
            ch = r.integers(0, 5, n).astype(np.int32)
            st = r.integers(0, 5_000_000, n).astype(np.int32)
            ln = np.full(n, 90, np.int32) if f else r.integers(1, 1500, n).astype(np.int32)
            return ora.Side(ch, st, st + ln)

        return side(30_000, False), side(200_000, fixed)

    errors = []

    def worker(seed, fixed):
        try:
            a, b = tables(seed, fixed)
            want = ora.sort_pairs(*ora.c_inner(a, b, "sweep", threads=1))
            want_cnt = ora.c_count(a, b, "sweep", threads=1)
            eng = HipEngine(0)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                da = DeviceSide.from_numpy(a.chrom, a.start, a.end)
                db = DeviceSide.from_numpy(b.chrom, b.start, b.end)
                for _ in range(6):
                    ra, rb = eng.inner_join(da, db, 5)
                    stream.synchronize()
                    assert np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), want)
                    assert np.array_equal(eng.count_overlaps(da, db, 5).cpu().numpy(), want_cnt)
                    ha, hb = eng.inner_join_host((a.chrom, a.start, a.end), (b.chrom, b.start, b.end), 5)
                    assert np.array_equal(ora.sort_pairs(ha, hb), want)
            eng.close()
        except BaseException as exc:  # noqa: BLE001 -- reported by the main thread
            errors.append(repr(exc))

    ts = [threading.Thread(target=worker, args=(s, f)) for s, f in ((101, True), (202, False))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


# ---- inputs that arrive sorted skip their sort -------------------------------------------
def _sorted_side(side):
    order = np.lexsort((side.start, side.chrom))
    return ora.Side(side.chrom[order], side.start[order], side.end[order], side.start_off, side.end_off)


@pytest.mark.parametrize("which", ["both", "a", "b"])
@pytest.mark.parametrize("uniform", [False, True])
def test_sorted_inputs_skip_their_sort_and_unsorted_ones_are_noticed(eng_fresh, which, uniform):
    a = rand_side(2101, 60_000, 7, 5_000_000, 900)
    b = uniform_side(2102, 250_000, 7, 5_000_000, 150) if uniform else rand_side(2102, 250_000, 7, 5_000_000, 700)
    sa = _sorted_side(a) if which in ("both", "a") else a
    sb = _sorted_side(b) if which in ("both", "b") else b
    want = ora.sort_pairs(*ora.c_inner(sa, sb, "sweep"))
    for _ in range(3):   # first plan: read back; then the context's guess
        assert np.array_equal(gpu_inner(eng_fresh, sa, sb, 7), want)
        assert eng_fresh.stats()["presorted"]
    # the one-call form on the speculating context
    cap = want.shape[0] + 100
    ra = torch.empty(cap, dtype=torch.int32, device="cuda:0")
    rb = torch.empty(cap, dtype=torch.int32, device="cuda:0")
    n = eng_fresh.inner_join_into(dev(sa), dev(sb), 7, ra, rb)
    assert n == want.shape[0] and np.array_equal(ora.sort_pairs(ra[:n].cpu().numpy(), rb[:n].cpu().numpy()), want)
    # ... now the same context meets the SHUFFLED tables: its guess is wrong, the plan is repeated
    want_u = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    assert np.array_equal(gpu_inner(eng_fresh, a, b, 7), want_u)
    assert not eng_fresh.stats()["presorted"]
    assert np.array_equal(gpu_inner(eng_fresh, a, b, 7), want_u)
    # one row out of place / one irregular row in an otherwise sorted side: not sorted
    x = _sorted_side(a)
    x.start[1000], x.start[1001] = x.start[1001], x.start[1000]
    x.end[1000], x.end[1001] = max(x.end[1000], x.start[1000] + 1), max(x.end[1001], x.start[1001] + 1)
    if x.start[1000] != x.start[1001] and x.chrom[1000] == x.chrom[1001]:
        want_x = ora.sort_pairs(*ora.c_inner(x, sb, "sweep"))
        assert np.array_equal(gpu_inner(eng_fresh, x, sb, 7), want_x)   # (the guess still says "b is shuffled": only slower)
        assert np.array_equal(gpu_inner(eng_fresh, x, sb, 7), want_x)   # the guess follows the previous plan
        assert eng_fresh.stats()["presorted"] == (which in ("both", "b"))
    y = _sorted_side(a)
    y.end[500] = y.start[500]
    assert np.array_equal(gpu_inner(eng_fresh, y, sb, 7), ora.sort_pairs(*ora.c_inner(y, sb, "sweep")))


def test_sorted_inputs_with_the_three_stage_sort_and_the_fused_count(monkeypatch):
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_LOCAL_MIN_ROWS", "1")
    e = HipEngine(0)
    monkeypatch.delenv("GIQL_HIP_LOCAL_MIN_ROWS")
    try:
        a = _sorted_side(rand_side(2111, 50_000, 5, 30_000_000, 2000, min_len=100))
        b = _sorted_side(uniform_side(2112, 400_000, 5, 30_000_000, 150))
        want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
        for _ in range(3):
            assert np.array_equal(gpu_inner(e, a, b, 5), want)
            st = e.stats()
            assert st["presorted"] and st["count_fused"] and st["join_form"] == "uniform_b"
        assert np.array_equal(gpu_inner(e, b, a, 5), ora.sort_pairs(*ora.c_inner(b, a, "sweep")))
        # encodings on both sides (offsets in the streamed keys)
        a1 = ora.Side(a.chrom, a.start + 1, a.end, -1, 0)
        b1 = ora.Side(b.chrom, b.start + 1, b.end + 1, -1, -1)
        assert np.array_equal(gpu_inner(e, a1, b1, 5), want)
    finally:
        e.close()


def test_host_entry_pipelined_over_row_blocks_of_the_larger_table(monkeypatch, eng):
    """giql_hip_inner with the larger table uploaded block by block (round 3): the union of the blocks' joins, the
    blocked table's row ids offset per block -- whichever table is the larger one, blocks that yield nothing, an
    irregular row (the literal path reads the uploaded columns again at fill time), a ragged last block."""
    monkeypatch.setenv("GIQL_HIP_E2E_BLOCK_ROWS", "7000")
    a = rand_side(1801, 3_000, 4, 600_000, 900)
    b = rand_side(1802, 40_000, 4, 600_000, 400, min_len=0)       # zero-length rows: the irregular path
    b.start[30_000:37_000] = 5_000_000                            # a whole block far from every A row
    b.end[30_000:37_000] = 5_000_100
    want = ora.sort_pairs(*ora.c_inner(a, b, "sweep"))
    for x, y, flip in ((a, b, False), (b, a, True)):
        ra, rb = eng.inner_join_host((x.chrom, x.start, x.end), (y.chrom, y.start, y.end), 4)
        got = ora.sort_pairs(rb, ra) if flip else ora.sort_pairs(ra, rb)
        assert got.shape == want.shape and np.array_equal(got, want)
    monkeypatch.setenv("GIQL_HIP_E2E_BLOCK_ROWS", "0")            # one shot: the same pairs
    ra, rb = eng.inner_join_host((a.chrom, a.start, a.end), (b.chrom, b.start, b.end), 4)
    assert np.array_equal(ora.sort_pairs(ra, rb), want)


def test_nearest_sorts_both_sides_from_their_raw_columns_once_the_layout_is_known(eng_fresh):
    """NEAREST k = 1 (round 3): the first call of a context probes the aligned layout, later calls build the keys in the
    first sort pass of each side (no linearize pass) -- same answers, with every encoding, zero-length rows, signed
    and max_distance; data the layout cannot take (a negative coordinate) repeats the call the ordinary way."""
    e = eng_fresh

    def check(a, b, nch, **kw):
        idx, dist = e.nearest(dev(a), dev(b), nch, **kw)
        oi, od = ora.c_nearest_k1(a, b, method="sweep", **kw)
        idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
        assert np.array_equal(dist, od) and np.array_equal(idx >= 0, oi >= 0)
        m = oi >= 0
        assert np.array_equal(b.start[idx[m]], b.start[oi[m]]) and np.array_equal(b.end[idx[m]], b.end[oi[m]])

    a = rand_side(1951, 60_000, 7, 4_000_000, 900, min_len=0)
    b = rand_side(1952, 90_000, 6, 4_000_000, 400, min_len=0)     # chrom 6 has no target
    check(a, b, 7)                                                 # probe
    for kw in ({}, {"signed": True}, {"signed": True, "max_distance": 700}):
        check(a, b, 7, **kw)                                       # keys from the raw columns
    for enc in ora.ENCODING_OFFSETS:
        so, eo = ora.ENCODING_OFFSETS[enc]
        check(ora.Side(a.chrom, a.start + 5, a.end + 5, so, eo), ora.Side(b.chrom, b.start + 5, b.end + 5, so, eo), 7)
    neg = ora.Side(a.chrom, a.start - 1000, a.end - 1000)        # canonical coordinates below zero: no aligned layout
    check(neg, b, 7)
    check(a, b, 7)
    # more chromosomes than the layout's table holds
    a40 = rand_side(1953, 20_000, 40, 500_000, 300)
    b40 = rand_side(1954, 30_000, 40, 500_000, 300)
    check(a40, b40, 40)
    check(a40, b40, 40)
