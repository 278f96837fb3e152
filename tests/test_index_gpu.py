"""Table index (giql_hip_index_create_dev / giql_hip_inner_join_indexed_dev) -- needs a GPU.

The reference's counterpart is the engine-side ``CREATE INDEX ... (chrom, start, "end")`` its performance guide
advises on both join sides (docs/transpilation/performance.rst:111-130); results are the per-chromosome INNER
plan's (src/giql/expanders/intersects_duckdb.py:1283-1330): same pairs as the ordinary join, bit-exact against the
oracle's sweep."""

import numpy as np
import pytest

from giql_amd import synth
from oracle import pyoracle as ora
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def eng():
    from giql_amd.engine import HipEngine

    e = HipEngine(0)
    yield e
    e.close()


def pairs_of(ra, rb):
    return ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy())


def want_pairs(a, b):
    wa, wb = ora.c_inner(a, b, "sweep")
    return ora.sort_pairs(wa, wb)


def table(n, seed, kind, chroms=None, enc=(0, 0)):
    c, s, e = synth.make_table(n, seed, kind, chroms=chroms)
    return ora.Side(c, (s - enc[0]).astype(np.int32), (e - enc[1]).astype(np.int32), enc[0], enc[1])


@pytest.mark.parametrize("kind_b,form", [("reads", "fixed_length"), ("peaks", "general")])
def test_one_index_serves_three_query_tables(eng, kind_b, form):
    b = table(3_000_000, 11, kind_b)
    index = eng.index_create(dev(b), 24)
    try:
        assert index.n == b.n and index.general == (form == "general")
        assert index.nbytes >= b.n * (12 if index.general else 8)
        for seed, n_a, kind_a in ((21, 400_000, "peaks"), (22, 150_000, "reads"), (23, 1_000, "peaks")):
            a = table(n_a, seed, kind_a)
            for _ in range(2):      # (the second call: buffers sized from the first)
                ra, rb = eng.inner_join_indexed(dev(a), index)
                st = eng.stats()
                assert st["bucket_join"] and st["join_form"] == ("general" if index.general else "uniform_b")
                assert np.array_equal(pairs_of(ra, rb), want_pairs(a, b))
    finally:
        index.close()


def test_index_axis_edges_and_encodings(eng):
    """Query rows on chromosomes the index does not hold, beyond the indexed range, reaching over its end, starting
    below 0 (a 1-based table's first position); every encoding pair."""
    from giql_amd.engine import ENCODING_OFFSETS

    r = np.random.default_rng(5)
    n_b = 600_000
    cb = r.integers(0, 5, n_b).astype(np.int32)
    cb[cb == 3] = 4                                   # chromosome 3: in no indexed row
    sb = r.integers(0, 30_000_000, n_b).astype(np.int32)
    lb = r.integers(1, 400, n_b).astype(np.int32)
    n_a = 200_000
    ca = r.integers(0, 7, n_a).astype(np.int32)       # 5, 6: beyond the index's dictionary
    sa = r.integers(0, 34_000_000, n_a).astype(np.int32)   # some start beyond every indexed row
    la = r.integers(1, 3000, n_a).astype(np.int32)
    sa[:50] = 0
    sa[50:100] = 29_999_990                           # ... and some reach over the end of the indexed range
    for enc_a in ENCODING_OFFSETS.values():
        for enc_b in ENCODING_OFFSETS.values():
            # canonical rows [s, s + l): raw columns are the canonical ones minus the offsets
            a = ora.Side(ca, (sa - enc_a[0]).astype(np.int32), (sa + la - enc_a[1]).astype(np.int32), enc_a[0], enc_a[1])
            b = ora.Side(cb, (sb - enc_b[0]).astype(np.int32), (sb + lb - enc_b[1]).astype(np.int32), enc_b[0], enc_b[1])
            index = eng.index_create(dev(b), 5)
            try:
                ra, rb = eng.inner_join_indexed(dev(a), index)
                assert np.array_equal(pairs_of(ra, rb), want_pairs(a, b)), (enc_a, enc_b)
            finally:
                index.close()


def test_short_buffers_report_the_count_and_write_nothing_past_them(eng):
    from giql_amd import _lib

    a, b = table(200_000, 31, "peaks"), table(2_500_000, 32, "reads")
    index = eng.index_create(dev(b), 24)
    try:
        want = want_pairs(a, b)
        ra = torch.full((1000,), -7, dtype=torch.int32, device="cuda:0")
        rb = torch.full((1000,), -7, dtype=torch.int32, device="cuda:0")
        with pytest.raises(_lib.GiqlHipError) as ei:
            eng.inner_join_indexed_into(dev(a), index, ra[:500], rb[:500])
        assert ei.value.code == _lib.GIQL_ERR_CAPACITY and eng.last_pairs == want.shape[0] > 500
        assert bool((ra[500:] == -7).all()) and bool((rb[500:] == -7).all())
        got = eng.inner_join_indexed(dev(a), index, cap=10)       # the wrapper sizes the second attempt from the count
        assert np.array_equal(pairs_of(*got), want)
        empty = ora.Side(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
        ea, eb = eng.inner_join_indexed(dev(empty), index)
        assert ea.shape[0] == 0 and eb.shape[0] == 0
    finally:
        index.close()


def test_tables_outside_the_indexed_form_are_declined_with_state(eng):
    from giql_amd import _lib

    ok = table(300_000, 41, "reads")
    # irregular rows in the table to index
    bad = ora.Side(ok.chrom.copy(), ok.start.copy(), ok.end.copy())
    bad.end[7] = bad.start[7]
    with pytest.raises(_lib.GiqlHipError) as ei:
        eng.index_create(dev(bad), 24)
    assert ei.value.code == _lib.GIQL_ERR_STATE
    # more chromosomes than the aligned axis holds; a negative coordinate; a table denser than the bucket stage takes
    many = ora.Side((np.arange(300_000) % 40).astype(np.int32), ok.start, ok.end)
    with pytest.raises(_lib.GiqlHipError) as ei:
        eng.index_create(dev(many), 40)
    assert ei.value.code == _lib.GIQL_ERR_STATE
    neg = ora.Side(ok.chrom, (ok.start - 1_000_000_000).astype(np.int32), (ok.end - 1_000_000_000).astype(np.int32))
    with pytest.raises(_lib.GiqlHipError) as ei:
        eng.index_create(dev(neg), 24)
    assert ei.value.code == _lib.GIQL_ERR_STATE
    # more rows in ONE 65,536-key bucket than the bucket stage sorts (2^18): declined ...
    s = np.random.default_rng(1).integers(0, 60_000, 400_000).astype(np.int32)
    piled = ora.Side(np.zeros(400_000, np.int32), s, (s + 100).astype(np.int32))
    with pytest.raises(_lib.GiqlHipError) as ei:
        eng.index_create(dev(piled), 1)
    assert ei.value.code == _lib.GIQL_ERR_STATE
    # ... while buckets past what LDS holds (4096 rows) but below that go through the queue kernel: slow, exact
    s = np.random.default_rng(2).integers(0, 1_000_000, 400_000).astype(np.int32)
    dense = ora.Side(np.zeros(400_000, np.int32), s, (s + 100).astype(np.int32))
    index = eng.index_create(dev(dense), 1)
    try:
        qs = np.random.default_rng(3).integers(0, 1_000_000, 3_000).astype(np.int32)
        q = ora.Side(np.zeros(3_000, np.int32), qs, (qs + np.random.default_rng(4).integers(1, 400, 3_000)).astype(np.int32))
        assert np.array_equal(pairs_of(*eng.inner_join_indexed(dev(q), index)), want_pairs(q, dense))
    finally:
        index.close()
    # an irregular QUERY row: declined per call (the literal predicate may hold for it: the ordinary join's case)
    index = eng.index_create(dev(ok), 24)
    try:
        q = table(50_000, 42, "peaks")
        q.end[3] = q.start[3]
        with pytest.raises(_lib.GiqlHipError) as ei:
            eng.inner_join_indexed(dev(q), index)
        assert ei.value.code == _lib.GIQL_ERR_STATE
        # ... and the context is fine afterwards
        q2 = table(50_000, 43, "peaks")
        assert np.array_equal(pairs_of(*eng.inner_join_indexed(dev(q2), index)), want_pairs(q2, ok))
        assert np.array_equal(pairs_of(*eng.inner_join(dev(q2), dev(ok), 24)), want_pairs(q2, ok))
    finally:
        index.close()


def test_execute_uses_the_index_of_a_pinned_table(monkeypatch):
    pa = pytest.importorskip("pyarrow")
    import giql_amd
    from giql_amd.engine import HipEngine
    from giql_amd.execute import execute
    from giql_amd.transpile import transpile

    names = np.array([f"chr{i + 1}" for i in range(24)])

    def arrow(side, extra_chrom=None):
        chrom = names[side.chrom].astype(object)
        if extra_chrom is not None:
            chrom[:100] = extra_chrom                 # a chromosome the other table does not have
        return pa.table({"chrom": pa.array(chrom, pa.string()), "start": pa.array(side.start), "end": pa.array(side.end),
                         "score": pa.array(np.arange(side.n, dtype=np.int32) % 13)})

    reads = table(1_500_000, 51, "reads")
    plan = transpile("SELECT a.start, a.score, b.start AS bs FROM peaks a JOIN reads b ON a.interval INTERSECTS b.interval",
                     tables=["peaks", "reads"], dialect="hip")
    built = []
    real = HipEngine.index_create
    monkeypatch.setattr(HipEngine, "index_create", lambda self, *a, **k: built.append(1) or real(self, *a, **k))
    with giql_amd.pin(arrow(reads), index=True) as pinned:
        for seed, extra in ((61, None), (62, "chrUn_1"), (63, None)):
            peaks = arrow(table(120_000, seed, "peaks"), extra)
            got = execute(plan, {"peaks": peaks, "reads": pinned})
            want = execute(plan, {"peaks": peaks, "reads": pinned.table})      # the ordinary join
            key = [("start", "ascending"), ("score", "ascending"), ("bs", "ascending")]
            assert got.num_rows == want.num_rows > 10_000 and got.sort_by(key).equals(want.sort_by(key))
        assert len(built) == 1 and pinned.index_info()[0]["rows"] == reads.n
        # the indexed table on the LEFT of the join
        plan2 = transpile("SELECT a.start AS bs, b.start, b.score FROM reads a JOIN peaks b ON a.interval INTERSECTS b.interval",
                          tables=["reads", "peaks"], dialect="hip")
        peaks = arrow(table(120_000, 64, "peaks"))
        got = execute(plan2, {"peaks": peaks, "reads": pinned})
        want = execute(plan2, {"peaks": peaks, "reads": pinned.table})
        key = [("start", "ascending"), ("score", "ascending"), ("bs", "ascending")]
        assert got.sort_by(key).equals(want.sort_by(key)) and len(built) == 1
    assert pinned.index_info() == []


# ---- an index over a DENSE table: narrower buckets (round 4, VERDICT r03 "Next round" 5 + 6) --------------------------
def test_index_of_a_dense_table_takes_narrower_buckets():
    """6M fixed-length rows on 60M positions (~5,900 rows per 65,536): round 3's index declined this table
    (GIQL_ERR_STATE), now it is indexed with buckets of 2^14 keys and serves the same pairs as the oracle."""
    from giql_amd.engine import HipEngine

    r = np.random.default_rng(611)
    s = r.integers(0, 60_000_000, 6_000_000).astype(np.int32)
    b = ora.Side(np.zeros(s.size, np.int32), s, s + np.int32(100))
    e = HipEngine(0)
    try:
        index = e.index_create(dev(b), 1)
        try:
            for seed, nq in ((612, 200_000), (613, 3_000)):
                qs = np.random.default_rng(seed).integers(0, 60_000_000, nq).astype(np.int32)
                q = ora.Side(np.zeros(nq, np.int32), qs, (qs + np.random.default_rng(seed + 9).integers(1, 900, nq)).astype(np.int32))
                for _ in range(2):
                    ra, rb = e.inner_join_indexed(dev(q), index)
                    st = e.stats()
                    assert st["bucket_join"] and st["bucket_bits"] == 14, st
                    assert np.array_equal(pairs_of(ra, rb), want_pairs(q, b))
        finally:
            index.close()
    finally:
        e.close()


@pytest.mark.parametrize("bits", [13, 15])
@pytest.mark.parametrize("kind_b", ["reads", "peaks"])
def test_index_with_a_forced_bucket_width(monkeypatch, bits, kind_b):
    from giql_amd.engine import HipEngine

    monkeypatch.setenv("GIQL_HIP_LOCAL_BITS", str(bits))
    e = HipEngine(0)
    monkeypatch.delenv("GIQL_HIP_LOCAL_BITS")
    try:
        b = table(1_500_000, 31, kind_b)
        index = e.index_create(dev(b), 24)
        try:
            assert index.general == (kind_b == "peaks")
            for seed, n_a, kind_a in ((32, 200_000, "peaks"), (33, 90_000, "reads")):
                a = table(n_a, seed, kind_a)
                ra, rb = e.inner_join_indexed(dev(a), index)
                st = e.stats()
                assert st["bucket_join"] and st["bucket_bits"] == bits, st
                assert np.array_equal(pairs_of(ra, rb), want_pairs(a, b))
            # the ordinary join on the same context afterwards
            a = table(50_000, 34, "peaks")
            assert np.array_equal(pairs_of(*e.inner_join(dev(a), dev(b), 24)), want_pairs(a, b))
        finally:
            index.close()
    finally:
        e.close()
