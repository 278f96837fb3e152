"""End-to-end: transpile(dialect="hip") + execute() on Arrow tables -- needs a GPU.

These read like the reference's execution tests for the DuckDB IEJoin dialect
(tests/test_duckdb_iejoin.py:3606-6608 of the reference): same fixtures, same
expected rows, ``conn.execute(sql)`` replaced by ``execute(plan, tables)``.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

pa = pytest.importorskip("pyarrow")
torch = pytest.importorskip("torch")

from giql_amd.execute import execute  # noqa: E402
from giql_amd.table import Table  # noqa: E402
from giql_amd.transpile import transpile  # noqa: E402


def make_table(rows):
    cols = ["chrom", "start", "end", "name", "score", "strand"]
    if not rows:
        return pa.table({"chrom": pa.array([], pa.string()), "start": pa.array([], pa.int32()),
                         "end": pa.array([], pa.int32()), "name": pa.array([], pa.string()),
                         "score": pa.array([], pa.int32()), "strand": pa.array([], pa.string())})
    data = {c: [r[i] for r in rows] for i, c in enumerate(cols)}
    return pa.table({"chrom": pa.array(data["chrom"], pa.string()),
                     "start": pa.array(data["start"], pa.int32()),
                     "end": pa.array(data["end"], pa.int32()),
                     "name": pa.array(data["name"], pa.string()),
                     "score": pa.array(data["score"], pa.int32()),
                     "strand": pa.array(data["strand"], pa.string())})


def rows_of(tbl):
    return sorted(tuple(d.values()) for d in tbl.to_pylist())


@pytest.fixture
def peaks_genes():
    peaks = make_table([
        ("chr1", 100, 200, "p1", 10, "+"), ("chr1", 300, 400, "p2", 20, "+"),
        ("chr1", 500, 600, "p3", 25, "+"), ("chr2", 100, 200, "p4", 30, "-"),
        ("chr2", 800, 900, "p5", 35, "-")])
    genes = make_table([
        ("chr1", 150, 250, "g1", 1, "+"), ("chr1", 500, 600, "g2", 2, "-"),
        ("chr1", 700, 800, "g3", 3, "+"), ("chr2", 50, 150, "g4", 4, "-"),
        ("chr2", 250, 350, "g5", 5, "+")])
    return {"peaks": peaks, "genes": genes}


EXPECTED = [("chr1", 100, 200, "chr1", 150, 250), ("chr1", 500, 600, "chr1", 500, 600),
            ("chr2", 100, 200, "chr2", 50, 150)]

Q6 = """
    SELECT a.chrom AS a_chrom, a.start AS a_start, a.end AS a_end,
           b.chrom AS b_chrom, b.start AS b_start, b.end AS b_end
    FROM peaks a
    JOIN genes b ON a.interval INTERSECTS b.interval
"""


def test_query_should_return_overlapping_pairs(peaks_genes):
    plan = transpile(Q6, tables=["peaks", "genes"], dialect="hip")
    assert rows_of(execute(plan, peaks_genes)) == sorted(EXPECTED)


def test_query_should_return_empty_set_when_no_chromosomes_intersect():
    t = {"peaks": make_table([("chr1", 100, 200, "p1", 0, "+")]),
         "genes": make_table([("chr2", 100, 200, "g1", 0, "+")])}
    assert rows_of(execute(transpile(Q6, tables=["peaks", "genes"], dialect="hip"), t)) == []


def test_query_should_handle_chrom_with_single_quote_in_name():
    t = {"peaks": make_table([("chr'1", 100, 200, "p1", 0, "+")]),
         "genes": make_table([("chr'1", 150, 250, "g1", 0, "+")])}
    got = rows_of(execute(transpile(Q6, tables=["peaks", "genes"], dialect="hip"), t))
    assert got == [("chr'1", 100, 200, "chr'1", 150, 250)]


@pytest.mark.parametrize("enc,peak,gene,match", [
    (("1based", "closed"), ("chr1", 100, 200), ("chr1", 200, 300), True),
    (("1based", "closed"), ("chr1", 100, 199), ("chr1", 200, 300), False),
    (("0based", "closed"), ("chr1", 100, 200), ("chr1", 200, 300), True),
    (("0based", "half_open"), ("chr1", 100, 200), ("chr1", 200, 300), False),
])
def test_touching_endpoints_follow_the_declared_encoding(enc, peak, gene, match):
    t = {"peaks": make_table([peak + ("p", 0, "+")]), "genes": make_table([gene + ("g", 0, "+")])}
    tables = [Table("peaks", coordinate_system=enc[0], interval_type=enc[1]),
              Table("genes", coordinate_system=enc[0], interval_type=enc[1])]
    got = rows_of(execute(transpile(Q6, tables=tables, dialect="hip"), t))
    assert (len(got) == 1) == match


def test_query_should_handle_mixed_coordinate_systems():
    t = {"peaks": make_table([("chr1", 100, 200, "p", 0, "+")]),
         "genes": make_table([("chr1", 99, 200, "g", 0, "+")])}
    q = "SELECT a.start AS a_s, b.start AS b_s FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval"
    tables = [Table("peaks", coordinate_system="1based", interval_type="closed"),
              Table("genes", coordinate_system="0based", interval_type="half_open")]
    assert rows_of(execute(transpile(q, tables=tables, dialect="hip"), t)) == [(100, 99)]


def test_query_should_apply_one_based_offset_only_to_the_left_table():
    t = {"peaks": make_table([("chr1", 100, 101, "p", 0, "+")]),
         "genes": make_table([("chr1", 99, 100, "g", 0, "+")])}
    q = "SELECT a.start AS a_s, b.start AS b_s FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval"
    off = [Table("peaks", coordinate_system="1based", interval_type="half_open"), Table("genes")]
    assert rows_of(execute(transpile(q, tables=off, dialect="hip"), t)) == [(100, 99)]
    assert rows_of(execute(transpile(q, tables=["peaks", "genes"], dialect="hip"), t)) == []


def test_query_should_return_empty_when_either_table_is_empty():
    q = "SELECT a.start AS a_s, b.start AS b_s FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval"
    plan = transpile(q, tables=["peaks", "genes"], dialect="hip")
    one = make_table([("chr1", 150, 250, "g1", 0, "+")])
    assert rows_of(execute(plan, {"peaks": make_table([]), "genes": one})) == []
    assert rows_of(execute(plan, {"peaks": one, "genes": make_table([])})) == []


def test_query_should_preserve_row_multiplicity_for_duplicate_input_rows():
    t = {"peaks": make_table([("chr1", 100, 200, "p_dup", 1, "+"), ("chr1", 100, 200, "p_dup", 1, "+")]),
         "genes": make_table([("chr1", 150, 250, "g1", 10, "-")])}
    q = "SELECT a.chrom AS c, a.start AS s, a.end AS e FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval"
    got = rows_of(execute(transpile(q, tables=["peaks", "genes"], dialect="hip"), t))
    assert got == [("chr1", 100, 200), ("chr1", 100, 200)]


def test_semi_and_anti_join(peaks_genes):
    q = "SELECT a.chrom, a.start, a.end FROM peaks a {} JOIN genes b ON a.interval INTERSECTS b.interval"
    semi = rows_of(execute(transpile(q.format("SEMI"), tables=["peaks", "genes"], dialect="hip"), peaks_genes))
    anti = rows_of(execute(transpile(q.format("ANTI"), tables=["peaks", "genes"], dialect="hip"), peaks_genes))
    assert semi == [("chr1", 100, 200), ("chr1", 500, 600), ("chr2", 100, 200)]
    assert anti == [("chr1", 300, 400), ("chr2", 800, 900)]


def test_anti_join_preserves_rows_from_left_only_chromosomes():
    t = {"peaks": make_table([("chr1", 10, 20, "p1", 0, "+"), ("chr1", 100, 200, "p2", 0, "+"),
                              ("chr3", 1, 1000, "p3", 0, "+")]),
         "genes": make_table([("chr1", 50, 150, "g1", 0, "+")])}
    q = "SELECT a.chrom, a.start, a.end FROM peaks a ANTI JOIN genes b ON a.interval INTERSECTS b.interval"
    got = rows_of(execute(transpile(q, tables=["peaks", "genes"], dialect="hip"), t))
    assert got == sorted([("chr1", 10, 20), ("chr3", 1, 1000)])


def test_nearest_k1_signed_and_tie_break():
    t = {"peaks": make_table([("chr1", 200, 300, "p", 0, "+")]),
         "genes": make_table([("chr1", 50, 100, "u", 0, "+"), ("chr1", 400, 450, "d", 0, "+")])}
    q = ("SELECT a.start AS a_start, b.start AS b_start, b.distance AS d FROM peaks a "
         "CROSS JOIN LATERAL NEAREST(genes, reference := a.interval, k := 1, signed := true) b")
    got = rows_of(execute(transpile(q, tables=["peaks", "genes"], dialect="hip"), t))
    assert got == [(200, 50, -101)]  # tie at |d| = 101: the lower (start, end) wins


def test_nearest_drops_rows_whose_chromosome_has_no_target():
    t = {"peaks": make_table([("chr1", 200, 300, "p", 0, "+"), ("chr9", 5, 6, "q", 0, "+")]),
         "genes": make_table([("chr1", 280, 290, "g", 0, "+")])}
    q = ("SELECT a.chrom AS c, b.start AS b_start FROM peaks a "
         "CROSS JOIN LATERAL NEAREST(genes, reference := a.interval, k := 1) b")
    assert rows_of(execute(transpile(q, tables=["peaks", "genes"], dialect="hip"), t)) == [("chr1", 280)]


COUNT_Q = ('SELECT a.chrom, a.start, a."end", COUNT(b.chrom) AS n FROM peaks a '
           'LEFT JOIN genes b ON a.interval INTERSECTS b.interval GROUP BY a.chrom, a.start, a."end"')


def _python_count_overlaps(peaks, genes):
    # the reference's test oracle for this shape (tests/test_duckdb_iejoin.py:66-81), restated
    out = []
    for key in set(peaks):
        pc, ps, pe = key
        dup = sum(1 for p in peaks if p == key)
        hits = sum(1 for (gc, gs, ge) in genes if pc == gc and pe > gs and ge > ps)
        out.append((pc, ps, pe, dup * hits))
    return sorted(out)


def test_count_overlaps_zero_fills_and_groups_duplicate_keys(peaks_genes):
    got = rows_of(execute(transpile(COUNT_Q, tables=["peaks", "genes"], dialect="hip"), peaks_genes))
    assert got == [("chr1", 100, 200, 1), ("chr1", 300, 400, 0), ("chr1", 500, 600, 1),
                   ("chr2", 100, 200, 1), ("chr2", 800, 900, 0)]


def test_count_overlaps_matches_python_reference_for_random_inputs():
    # reference tests/test_duckdb_iejoin.py:3520-3564: random rows incl. duplicate keys,
    # left-only chromosomes and zero-overlap keys
    rng = np.random.default_rng(20260209)
    def rows(n, chroms):
        out = []
        for _ in range(n):
            s = int(rng.integers(0, 400))
            out.append((str(rng.choice(chroms)), s, s + int(rng.integers(1, 60))))
        return out
    peak_rows = rows(150, ["chr1", "chr2", "chr3"])
    peak_rows += peak_rows[:20]  # duplicate keys
    gene_rows = rows(200, ["chr1", "chr2"])
    t = {"peaks": make_table([(c, s, e, "p", 0, "+") for c, s, e in peak_rows]),
         "genes": make_table([(c, s, e, "g", 0, "+") for c, s, e in gene_rows])}
    got = rows_of(execute(transpile(COUNT_Q, tables=["peaks", "genes"], dialect="hip"), t))
    assert got == _python_count_overlaps(peak_rows, gene_rows)


def test_count_overlaps_handles_empty_sides():
    plan = transpile(COUNT_Q, tables=["peaks", "genes"], dialect="hip")
    one = make_table([("chr1", 150, 250, "g1", 0, "+")])
    assert rows_of(execute(plan, {"peaks": make_table([]), "genes": one})) == []
    assert rows_of(execute(plan, {"peaks": one, "genes": make_table([])})) == [("chr1", 150, 250, 0)]


def test_count_overlaps_rejects_null_count_column(peaks_genes):
    # COUNT(col) skips NULLs; the kernels count rows, so a nullable argument is refused
    q = COUNT_Q.replace("COUNT(b.chrom)", "COUNT(b.name)")
    genes = peaks_genes["genes"].set_column(3, "name", pa.array(["g1", None, "g3", "g4", "g5"], pa.string()))
    with pytest.raises(ValueError, match="NULL"):
        execute(transpile(q, tables=["peaks", "genes"], dialect="hip"), {"peaks": peaks_genes["peaks"], "genes": genes})


def _rows(n, tag, rng):
    out = []
    for i in range(n):
        s = int(rng.integers(0, 3000))
        out.append((str(rng.choice(["chr1", "chr2", "chr3"])), s, s + int(rng.integers(1, 400)), f"{tag}{i % 17}",
                    int(rng.integers(0, 6)), str(rng.choice(["+", "-"]))))
    return out


def _overlap(p, g):
    return p[0] == g[0] and p[1] < g[2] and p[2] > g[1]


RESIDUAL_CASES = [
    # (ON / WHERE text after the INTERSECTS, python predicate over (peak row, gene row))
    ("AND a.strand = b.strand", "", lambda p, g: p[5] == g[5]),                       # same-strand recipe
    ("AND a.strand != b.strand", "", lambda p, g: p[5] != g[5]),                      # opposite-strand recipe
    ("AND a.score > 2", "", lambda p, g: p[4] > 2),
    ("", "WHERE b.score <= 3 AND a.score >= 1", lambda p, g: g[4] <= 3 and p[4] >= 1),
    ("AND a.score < b.score", "WHERE a.name <> 'p3'", lambda p, g: p[4] < g[4] and p[3] != "p3"),
    ("AND a.strand < b.strand AND 2 < b.score", "WHERE a.name >= 'p3'", lambda p, g: p[5] < g[5] and 2 < g[4] and p[3] >= "p3"),
    ("AND a.score = 2.0", "", lambda p, g: p[4] == 2),
]


@pytest.mark.parametrize("on,where,pred", RESIDUAL_CASES, ids=[c[0] + " " + c[1] for c in RESIDUAL_CASES])
def test_inner_join_with_residual_predicates(on, where, pred):
    # the reference inlines these conjuncts into the per-chromosome join (intersects_duckdb.py:1239-1243);
    # expected rows = brute force over (overlap AND residual)
    rng = np.random.default_rng(42)
    peaks, genes = _rows(300, "p", rng), _rows(500, "g", rng)
    t = {"peaks": make_table(peaks), "genes": make_table(genes)}
    q = ("SELECT a.name AS an, a.start AS s, b.name AS bn, b.end AS e FROM peaks a JOIN genes b "
         f"ON a.interval INTERSECTS b.interval {on} {where}")
    got = rows_of(execute(transpile(q, tables=["peaks", "genes"], dialect="hip"), t))
    want = sorted((p[3], p[1], g[3], g[2]) for p in peaks for g in genes if _overlap(p, g) and pred(p, g))
    assert got == want and len(want) > 0


@pytest.mark.parametrize("kind", ["SEMI", "ANTI"])
@pytest.mark.parametrize("on,where,on_pred,where_pred", [
    ("AND a.strand = b.strand", "", lambda p, g: p[5] == g[5], lambda p: True),
    ("AND b.score > 2", "WHERE a.score < 4", lambda p, g: g[4] > 2, lambda p: p[4] < 4),
    ("AND a.score > 2", "", lambda p, g: p[4] > 2, lambda p: True),   # a left-only ON residual: ANTI KEEPS failing rows
    ("AND a.score <= b.score AND a.score > 0", "WHERE a.name != 'p1'", lambda p, g: 0 < p[4] <= g[4], lambda p: p[3] != "p1"),
], ids=["same_strand", "right_on+left_where", "left_on", "two_sided+where"])
def test_semi_anti_with_residual_predicates(kind, on, where, on_pred, where_pred):
    # ON residuals take part in the existence test, WHERE residuals filter the output (#200,
    # intersects_duckdb.py:1164-1177)
    rng = np.random.default_rng(7)
    peaks, genes = _rows(250, "p", rng), _rows(150, "g", rng)
    peaks.append(("chr9", 5, 50, "p1", 3, "+"))  # left-only chromosome
    t = {"peaks": make_table(peaks), "genes": make_table(genes)}
    q = f"SELECT a.name, a.start, a.score FROM peaks a {kind} JOIN genes b ON a.interval INTERSECTS b.interval {on} {where}"
    got = rows_of(execute(transpile(q, tables=["peaks", "genes"], dialect="hip"), t))
    def exists(p):
        return any(_overlap(p, g) and on_pred(p, g) for g in genes)
    want = sorted((p[3], p[1], p[4]) for p in peaks if where_pred(p) and (exists(p) != (kind == "ANTI")))
    assert got == want and len(want) > 0


REFERENCE_RESIDUAL_CASES = [
    # (source test in the reference's tests/test_duckdb_iejoin.py, peaks, genes, query, expected rows)
    (":347-392 combine ON and WHERE extras",
     [("chr1", 100, 200, "p1", 10, "+"), ("chr1", 300, 400, "p2", 25, "+"), ("chr1", 500, 600, "p3", 30, "+")],
     [("chr1", 150, 250, "g1", 5, "+"), ("chr1", 350, 450, "g2", 5, "+"), ("chr1", 550, 650, "g3", 50, "+")],
     "SELECT a.start AS s FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND a.score > 20 WHERE b.score < 40",
     [(300,)]),
    (":6201-6249 ANTI + WHERE residual is an outer filter (#200)",
     [("chr1", 100, 200, "p1", 1, "+"), ("chr1", 500, 600, "p2", 2, "+"), ("chr1", 700, 800, "p3", 3, "+")],
     [("chr1", 150, 250, "g1", 1, "+"), ("chr1", 500, 600, "g2", 2, "+")],
     "SELECT a.start FROM peaks a ANTI JOIN genes b ON a.interval INTERSECTS b.interval WHERE a.start >= 550",
     [(700,)]),
    (":6251-6289 ANTI + WHERE on a column absent from SELECT",
     [("chr1", 100, 200, "p1", 1, "+"), ("chr1", 700, 800, "p3", 3, "+")],
     [("chr1", 150, 250, "g1", 1, "+")],
     "SELECT a.start FROM peaks a ANTI JOIN genes b ON a.interval INTERSECTS b.interval WHERE a.end > 750",
     [(700,)]),
    (":6291-6338 SEMI + WHERE residual",
     [("chr1", 100, 200, "p1", 1, "+"), ("chr1", 500, 600, "p2", 2, "+"), ("chr1", 700, 800, "p3", 3, "+")],
     [("chr1", 150, 250, "g1", 1, "+"), ("chr1", 500, 600, "g2", 2, "+")],
     "SELECT a.start FROM peaks a SEMI JOIN genes b ON a.interval INTERSECTS b.interval WHERE a.start >= 300",
     [(500,)]),
    (":6413-6463 ANTI + ON residual on the right side stays a join condition",
     [("chr1", 100, 200, "p1", 1, "+"), ("chr1", 500, 600, "p2", 2, "+"), ("chr1", 700, 800, "p3", 3, "+")],
     [("chr1", 150, 250, "g1", 10, "+"), ("chr1", 500, 600, "g2", 50, "+")],
     "SELECT a.start FROM peaks a ANTI JOIN genes b ON a.interval INTERSECTS b.interval AND b.score > 100",
     [(100,), (500,), (700,)]),
]


@pytest.mark.parametrize("src,peaks,genes,query,expected", REFERENCE_RESIDUAL_CASES, ids=[c[0] for c in REFERENCE_RESIDUAL_CASES])
def test_residual_known_answers_from_the_reference(src, peaks, genes, query, expected):
    t = {"peaks": make_table(peaks), "genes": make_table(genes)}
    assert rows_of(execute(transpile(query, tables=["peaks", "genes"], dialect="hip"), t)) == expected


def test_residual_null_operands_never_match_and_type_mismatch_raises(peaks_genes):
    genes = peaks_genes["genes"].set_column(4, "score", pa.array([1, None, 3, 4, 5], pa.int32()))
    t = {"peaks": peaks_genes["peaks"], "genes": genes}
    q = "SELECT a.name, b.name AS g FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND b.score < 100"
    got = rows_of(execute(transpile(q, tables=["peaks", "genes"], dialect="hip"), t))
    assert got == [("p1", "g1"), ("p4", "g4")]  # (p3, g2) overlaps but g2.score is NULL
    bad = "SELECT a.name FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND a.name > 3"
    with pytest.raises(ValueError, match="string with a number"):
        execute(transpile(bad, tables=["peaks", "genes"], dialect="hip"), t)


def test_literal_range_filter_runs_on_the_gpu():
    # BASELINE config 1: SELECT * FROM peaks WHERE interval INTERSECTS 'chr1:1000-2000'
    rng = np.random.default_rng(4)
    rows = _rows(600, "p", rng)
    rows += [("chr1", 900, 1000, "touch_lo", 1, "+"), ("chr1", 2000, 2100, "touch_hi", 1, "+"),
             ("chr1", 999, 1001, "in1", 1, "+"), ("chr1", 1999, 2050, "in2", 1, "-")]
    t = {"peaks": make_table(rows)}
    got = rows_of(execute(transpile("SELECT * FROM peaks WHERE interval INTERSECTS 'chr1:1000-2000'",
                                    tables=["peaks"], dialect="hip"), t))
    want = sorted(r for r in rows if r[0] == "chr1" and r[1] < 2000 and r[2] > 1000)
    assert got == want and ("chr1", 999, 1001, "in1", 1, "+") in got
    assert not any(r[3] in ("touch_lo", "touch_hi") for r in got)
    q = "SELECT name, score AS s FROM peaks p WHERE p.interval INTERSECTS 'chr2:100-900' AND p.score >= 3 AND p.strand = '-'"
    got = rows_of(execute(transpile(q, tables=["peaks"], dialect="hip"), t))
    assert got == sorted((r[3], r[4]) for r in rows if r[0] == "chr2" and r[1] < 900 and r[2] > 100 and r[4] >= 3 and r[5] == "-")
    q = ("SELECT name, score FROM peaks WHERE interval INTERSECTS 'chr2:100-900' AND (score >= 4 OR strand = '-') "
         "AND NOT name IN ('p1', 'p2') AND score BETWEEN 1 AND 4")
    got = rows_of(execute(transpile(q, tables=["peaks"], dialect="hip"), t))
    want = sorted((r[3], r[4]) for r in rows if r[0] == "chr2" and r[1] < 900 and r[2] > 100 and (r[4] >= 4 or r[5] == "-")
                  and r[3] not in ("p1", "p2") and 1 <= r[4] <= 4)
    assert got == want and len(want) > 3
    none = execute(transpile("SELECT * FROM peaks WHERE interval INTERSECTS 'chrZ:1-2'", tables=["peaks"], dialect="hip"), t)
    assert none.num_rows == 0 and none.column_names == t["peaks"].column_names


def test_cluster_query_known_answers():
    # tests/integration/datafusion/test_cross_target_oracle.py:978-1003 and
    # tests/integration/bedtools/test_cluster.py:69-113 (contained intervals share one cluster)
    t = {"peaks": make_table([("chr1", 100, 200, "a", 0, "+"), ("chr1", 150, 300, "b", 0, "+"),
                              ("chr1", 5000, 6000, "c", 0, "+")])}
    q = 'SELECT chrom, start, "end", CLUSTER(interval) AS cid FROM peaks'
    got = rows_of(execute(transpile(q, tables=["peaks"], dialect="hip"), t))
    assert got == [("chr1", 100, 200, 1), ("chr1", 150, 300, 1), ("chr1", 5000, 6000, 2)]
    t2 = {"intervals": make_table([("chr1", 0, 1000, "i1", 100, "+"), ("chr1", 100, 200, "i2", 150, "+"),
                                   ("chr1", 300, 400, "i3", 200, "+")])}
    star = execute(transpile("SELECT *, CLUSTER(interval) AS cluster_id FROM intervals", tables=["intervals"],
                             dialect="hip"), t2)
    assert star.column_names == ["chrom", "start", "end", "name", "score", "strand", "cluster_id"]
    assert set(star.column("cluster_id").to_pylist()) == {1}


def test_merge_query_known_answers():
    # tests/integration/datafusion/test_cross_target_oracle.py:1196-1236;
    # tests/integration/bedtools/test_merge.py:15-43 (book-ended half-open intervals merge)
    t = {"peaks": make_table([("chr1", 100, 200, "a", 0, "+"), ("chr1", 150, 300, "b", 0, "+"),
                              ("chr1", 5000, 6000, "c", 0, "+")])}
    plan = transpile("SELECT MERGE(interval) FROM peaks", tables=["peaks"], dialect="hip")
    out = execute(plan, t)
    assert out.column_names == ["chrom", "start", "end"]
    assert [tuple(d.values()) for d in out.to_pylist()] == [("chr1", 100, 300), ("chr1", 5000, 6000)]
    assert execute(plan, {"peaks": make_table([])}).num_rows == 0
    t2 = {"intervals": make_table([("chr1", 100, 200, "i1", 0, "+"), ("chr1", 200, 300, "i2", 0, "+"),
                                   ("chr1", 300, 400, "i3", 0, "+")])}
    out2 = execute(transpile("SELECT MERGE(interval), COUNT(*) AS n FROM intervals", tables=["intervals"],
                             dialect="hip"), t2)
    assert [tuple(d.values()) for d in out2.to_pylist()] == [("chr1", 100, 400, 3)]


def test_cluster_and_merge_stranded_distance_and_where_vs_brute_force():
    rng = np.random.default_rng(99)
    rows = _rows(400, "f", rng)
    t = {"features": make_table(rows)}
    def brute(rows_in, distance, stranded):
        ids, merged = {}, []
        parts = {}
        for i, r in enumerate(rows_in):
            parts.setdefault((r[0], r[5]) if stranded else (r[0],), []).append(i)
        for key, idx in parts.items():
            idx.sort(key=lambda i: (rows_in[i][1], i))
            cid, run_max = 0, None
            for i in idx:
                if run_max is None or run_max + distance < rows_in[i][1]:
                    cid += 1
                    merged.append([key, rows_in[i][1], rows_in[i][2], 0])
                    run_max = rows_in[i][2]
                run_max = max(run_max, rows_in[i][2])
                merged[-1][2] = max(merged[-1][2], rows_in[i][2])
                merged[-1][3] += 1
                ids[i] = cid
        return ids, merged
    for distance, stranded in [(0, False), (25, True), (300, False)]:
        arg = f"interval, {distance}" + (", stranded := true" if stranded else "")
        q = f'SELECT name, start, CLUSTER({arg}) AS cid FROM features WHERE score >= 2'
        kept = [r for r in rows if r[4] >= 2]
        ids, merged = brute(kept, distance, stranded)
        got = rows_of(execute(transpile(q, tables=["features"], dialect="hip"), t))
        assert got == sorted((r[3], r[1], ids[i]) for i, r in enumerate(kept))
        qm = f"SELECT MERGE({arg}), COUNT(*) AS n FROM features WHERE score >= 2"
        out = execute(transpile(qm, tables=["features"], dialect="hip"), t)
        want = sorted(tuple(k) + (s, e, n) for k, s, e, n in merged)
        assert sorted(tuple(d.values()) for d in out.to_pylist()) == want
        starts = list(zip(out.column("chrom").to_pylist(), out.column("start").to_pylist()))
        assert starts == sorted(starts)  # ORDER BY chrom, start (merge.py:20)


def test_device_projection_matches_host_projection_for_every_column_type():
    # the projected columns are gathered on the GPU (giql_hip_take_*); same rows as pyarrow.take
    rng = np.random.default_rng(11)
    def tbl(n, tag):
        s = rng.integers(0, 5000, n)
        names = [None if i % 7 == 3 else f"{tag}{i}" * (1 + i % 5) for i in range(n)]
        score = [None if i % 5 == 1 else float(i) / 3 for i in range(n)]
        return pa.table({"chrom": pa.array(rng.choice(["chr1", "chr2", "chrX"], n), pa.string()),
                         "start": pa.array(s, pa.int32()), "end": pa.array(s + rng.integers(1, 300, n), pa.int32()),
                         "name": pa.array(names, pa.string()), "score": pa.array(score, pa.float64()),
                         "tiny": pa.array(rng.integers(-5, 5, n), pa.int8()),
                         "flag": pa.array(rng.integers(0, 2, n).astype(bool)),
                         "strand": pa.array(rng.choice(["+", "-"], n)).dictionary_encode()})
    t = {"peaks": tbl(400, "p"), "genes": tbl(700, "g")}
    q = ("SELECT a.name AS an, a.score AS asc_, a.tiny AS at, a.flag AS af, a.strand AS ast, a.start AS s, "
         "b.name AS bn, b.score AS bs, b.end AS be, b.strand AS bst "
         "FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval")
    plan = transpile(q, tables=["peaks", "genes"], dialect="hip")
    dev_rows = execute(plan, t)
    host_rows = execute(plan, t, device_projection=False)
    assert dev_rows.num_rows == host_rows.num_rows > 1000
    key = lambda tb: sorted(map(repr, tb.to_pylist()))
    assert key(dev_rows) == key(host_rows)
    assert dev_rows.schema.types == host_rows.schema.types


def test_nulls_and_out_of_range_are_rejected(peaks_genes):
    bad = peaks_genes["peaks"].set_column(1, "start", pa.array([100, None, 500, 100, 800], pa.int32()))
    plan = transpile(Q6, tables=["peaks", "genes"], dialect="hip")
    with pytest.raises(ValueError, match="NULL"):
        execute(plan, {"peaks": bad, "genes": peaks_genes["genes"]})
    big = peaks_genes["peaks"].set_column(2, "end", pa.array([2**40, 1, 2, 3, 4], pa.int64()))
    with pytest.raises(ValueError, match="int32"):
        execute(plan, {"peaks": big, "genes": peaks_genes["genes"]})


# ------------------------------------------------ outer clauses (reference known answers)
import _golden as G  # noqa: E402

OUTER = G.load("outer_clauses.json")["cases"]


@pytest.mark.parametrize("case", OUTER, ids=lambda c: c["ref"])
def test_outer_clauses_reference_known_answers(case):
    # DISTINCT / GROUP BY + aggregates / ORDER BY / LIMIT / OFFSET ride on the reference's outer SELECT
    # wrapper (intersects_duckdb.py:1336-1400); here they finish on the projected Arrow table
    tables = {"peaks": make_table([tuple(r) for r in case["peaks"]]), "genes": make_table([tuple(r) for r in case["genes"]])}
    out = execute(transpile(case["q"], ["peaks", "genes"], dialect="hip"), tables)
    got = [list(d.values()) for d in out.to_pylist()]
    want = case["want"]
    if not case["ordered"]:
        got, want = sorted(got), sorted(want)
    if case.get("approx"):
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g[0] == w[0] and g[1] == pytest.approx(w[1])
    else:
        assert got == want


def test_outer_clauses_against_brute_force(peaks_genes):
    rng = np.random.default_rng(5)
    n = 400
    def tbl(seed):
        r = np.random.default_rng(seed)
        st = r.integers(0, 20_000, n)
        return make_table([(f"chr{int(c)}", int(s), int(s + l), f"n{i}", int(sc), "+-"[int(k)])
                           for i, (c, s, l, sc, k) in enumerate(zip(r.integers(1, 4, n), st, r.integers(1, 400, n),
                                                                    r.integers(0, 50, n), r.integers(0, 2, n)))])
    tables = {"peaks": tbl(1), "genes": tbl(2)}
    P, Gn = tables["peaks"].to_pylist(), tables["genes"].to_pylist()
    pairs = [(p, g) for p in P for g in Gn if p["chrom"] == g["chrom"] and p["start"] < g["end"] and p["end"] > g["start"]]
    assert len(pairs) > 500
    J = "FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval"
    # GROUP BY two keys + four aggregates
    out = execute(transpile(f"SELECT a.chrom, b.strand AS s, COUNT(*) AS n, SUM(b.score) AS t, MIN(a.start) AS lo, "
                            f"MAX(b.end) AS hi, COUNT(DISTINCT b.name) AS d {J} GROUP BY a.chrom, b.strand",
                            ["peaks", "genes"], dialect="hip"), tables)
    want = {}
    for p, g in pairs:
        k = (p["chrom"], g["strand"])
        w = want.setdefault(k, [0, 0, 10**9, -1, set()])
        w[0] += 1; w[1] += g["score"]; w[2] = min(w[2], p["start"]); w[3] = max(w[3], g["end"]); w[4].add(g["name"])
    assert sorted((d["chrom"], d["s"], d["n"], d["t"], d["lo"], d["hi"], d["d"]) for d in out.to_pylist()) == \
        sorted((k[0], k[1], w[0], w[1], w[2], w[3], len(w[4])) for k, w in want.items())
    # global aggregates, no GROUP BY
    out = execute(transpile(f"SELECT COUNT(*) AS n, AVG(a.score) AS m {J}", ["peaks", "genes"], dialect="hip"), tables)
    assert out.to_pylist() == [{"n": len(pairs), "m": pytest.approx(sum(p["score"] for p, _ in pairs) / len(pairs))}]
    # DISTINCT + ORDER BY two keys (one descending) + OFFSET / LIMIT
    out = execute(transpile(f"SELECT DISTINCT a.chrom, b.score {J} ORDER BY a.chrom DESC, b.score LIMIT 7 OFFSET 3",
                            ["peaks", "genes"], dialect="hip"), tables)
    allrows = sorted({(p["chrom"], g["score"]) for p, g in pairs}, key=lambda t: (-int(t[0][3:]), t[1]))
    assert [(d["chrom"], d["score"]) for d in out.to_pylist()] == allrows[3:10]
    # SEMI join with ORDER BY a hidden column and LIMIT
    out = execute(transpile("SELECT a.name FROM peaks a SEMI JOIN genes b ON a.interval INTERSECTS b.interval "
                            "ORDER BY a.score DESC, a.name LIMIT 5", ["peaks", "genes"], dialect="hip"), tables)
    hit = {p["name"]: p["score"] for p, _ in pairs}
    assert [d["name"] for d in out.to_pylist()] == [n for n, _ in sorted(hit.items(), key=lambda kv: (-kv[1], kv[0]))[:5]]
    # USING (chrom)
    out = execute(transpile("SELECT a.name, b.name AS g FROM peaks a JOIN genes b USING (chrom) "
                            "WHERE a.interval INTERSECTS b.interval", ["peaks", "genes"], dialect="hip"), tables)
    assert sorted((d["name"], d["g"]) for d in out.to_pylist()) == sorted((p["name"], g["name"]) for p, g in pairs)
    # empty result: COUNT(*) is 0, SUM is NULL
    far = {"peaks": tables["peaks"], "genes": make_table([("chr9", 1, 2, "g", 1, "+")])}
    out = execute(transpile(f"SELECT COUNT(*) AS n, SUM(b.score) AS t {J}", ["peaks", "genes"], dialect="hip"), far)
    assert out.to_pylist() == [{"n": 0, "t": None}]


# --------------------------------------------------- NEAREST k > 1 and stranded (reference known answers)
def test_nearest_k2_and_stranded_reference_known_answers():
    X = "CROSS JOIN LATERAL NEAREST(genes, reference := a.interval"
    # tests/integration/datafusion/test_cross_target_oracle.py:293-324
    t = {"peaks": make_table([("chr1", 200, 300, "p", 0, "+")]),
         "genes": make_table([("chr1", 1000, 1100, "g1", 0, "+"), ("chr1", 50, 60, "g2", 0, "+"),
                              ("chr1", 280, 290, "g3", 0, "+"), ("chr1", 310, 320, "g4", 0, "+")])}
    out = execute(transpile(f"SELECT a.start AS a_start, b.start AS b_start FROM peaks a {X}, k := 2) b", ["peaks", "genes"],
                            dialect="hip"), t)
    assert [(d["a_start"], d["b_start"]) for d in out.to_pylist()] == [(200, 280), (200, 310)]
    # :398-424 -- the same-strand gene wins although the opposite-strand one is nearer
    t = {"peaks": make_table([("chr1", 200, 300, "p", 0, "+")]),
         "genes": make_table([("chr1", 280, 290, "g1", 0, "+"), ("chr1", 250, 260, "g2", 0, "-")])}
    q = f"SELECT a.start AS a_start, b.start AS b_start FROM peaks a {X}, k := 1, stranded := true) b"
    assert [(d["a_start"], d["b_start"]) for d in execute(transpile(q, ["peaks", "genes"], dialect="hip"), t).to_pylist()] == [(200, 280)]
    # :482-520 -- co-located opposite-strand reference rows keep their own strand's nearest
    t["peaks"] = make_table([("chr1", 200, 300, "p1", 0, "+"), ("chr1", 200, 300, "p2", 0, "-")])
    q = f"SELECT a.strand AS a_strand, b.start AS b_start FROM peaks a {X}, k := 1, stranded := true) b"
    assert rows_of(execute(transpile(q, ["peaks", "genes"], dialect="hip"), t)) == [("+", 280), ("-", 250)]
    # signed + stranded: a '-' reference flips the sign (_distance.py:88-117): upstream of a '-' row is positive
    q = f"SELECT a.name, b.name AS g, b.distance AS d FROM peaks a {X}, k := 2, stranded := true, signed := true) b"
    t["genes"] = make_table([("chr1", 100, 150, "gm", 0, "-"), ("chr1", 400, 450, "gp", 0, "+"), ("chr1", 350, 360, "gm2", 0, "-")])
    assert rows_of(execute(transpile(q, ["peaks", "genes"], dialect="hip"), t)) == [("p1", "gp", 101), ("p2", "gm", 51), ("p2", "gm2", -51)]
    # '.' targets never pair with '+' / '-' references (tests/test_nearest_k.py holds the '.' / '?' golden rows)
    t["genes"] = make_table([("chr1", 100, 150, "g", 0, ".")])
    assert execute(transpile(q, ["peaks", "genes"], dialect="hip"), t).num_rows == 0
    with pytest.raises(ValueError, match="strands other than"):
        t["genes"] = make_table([("chr1", 100, 150, "g", 0, "x")])
        execute(transpile(q, ["peaks", "genes"], dialect="hip"), t)


def test_having_group_without_aggregate_and_null_placement():
    """HAVING over aggregates (in the SELECT list or hidden), GROUP BY without an aggregate, and the
    per-key NULL placement of ORDER BY -- expected rows restated in Python from the brute-force pair list."""
    n = 300

    def tbl(seed, with_nulls):
        r = np.random.default_rng(seed)
        st = r.integers(0, 15_000, n)
        rows = [(f"chr{int(c)}", int(s), int(s + l), f"n{i}", int(sc), "+-"[int(k)])
                for i, (c, s, l, sc, k) in enumerate(zip(r.integers(1, 5, n), st, r.integers(1, 300, n),
                                                         r.integers(0, 30, n), r.integers(0, 2, n)))]
        t = make_table(rows)
        if with_nulls:   # NULL scores on every seventh row
            sc = [None if i % 7 == 0 else row[4] for i, row in enumerate(rows)]
            t = t.set_column(t.schema.get_field_index("score"), "score", pa.array(sc, pa.int32()))
        return t

    tables = {"peaks": tbl(11, False), "genes": tbl(12, True)}
    P, Gn = tables["peaks"].to_pylist(), tables["genes"].to_pylist()
    pairs = [(p, g) for p in P for g in Gn if p["chrom"] == g["chrom"] and p["start"] < g["end"] and p["end"] > g["start"]]
    assert len(pairs) > 300
    J = "FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval"

    def run(q):
        return execute(transpile(q, ["peaks", "genes"], dialect="hip"), tables).to_pylist()

    groups = {}
    for p, g in pairs:
        w = groups.setdefault(p["name"], [0, 0, 0])
        w[0] += 1
        if g["score"] is not None:
            w[1] += g["score"]
            w[2] += 1
    # HAVING on a SELECT-list aggregate and on a hidden one (SUM ignores NULLs; an all-NULL group's SUM is NULL
    # and the comparison drops it)
    got = run(f"SELECT a.name, COUNT(*) AS n {J} GROUP BY a.name HAVING COUNT(*) >= 3 AND SUM(b.score) > 20")
    want = sorted((k, w[0]) for k, w in groups.items() if w[0] >= 3 and w[2] > 0 and w[1] > 20)
    assert sorted((d["name"], d["n"]) for d in got) == want and want
    # literal on the left, a key column in HAVING
    got = run(f"SELECT a.chrom AS c, COUNT(*) AS n {J} GROUP BY a.chrom HAVING 'chr2' <= a.chrom AND n > 1")
    bych = {}
    for p, _ in pairs:
        bych[p["chrom"]] = bych.get(p["chrom"], 0) + 1
    assert sorted((d["c"], d["n"]) for d in got) == sorted((c, k) for c, k in bych.items() if c >= "chr2" and k > 1)
    # HAVING without GROUP BY: the whole result is one group
    assert run(f"SELECT COUNT(*) AS n {J} HAVING COUNT(*) > 0") == [{"n": len(pairs)}]
    assert run(f"SELECT COUNT(*) AS n {J} HAVING COUNT(*) < 0") == []
    # GROUP BY without any aggregate = one row per key
    got = run(f"SELECT a.chrom, b.strand {J} GROUP BY a.chrom, b.strand")
    assert sorted((d["chrom"], d["strand"]) for d in got) == sorted({(p["chrom"], g["strand"]) for p, g in pairs})
    # NULL placement: unwritten = NULLs are small (first ascending, last descending); written = as written, per key
    vals = sorted({(g["score"], g["name"]) for _, g in pairs}, key=lambda t: (t[0] is not None, t[0] if t[0] is not None else 0, t[1]))
    got = run(f"SELECT DISTINCT b.score, b.name {J} ORDER BY b.score, b.name")
    assert [(d["score"], d["name"]) for d in got] == vals
    got = run(f"SELECT DISTINCT b.score, b.name {J} ORDER BY b.score DESC, b.name")
    want = sorted(vals, key=lambda t: (t[0] is None, -(t[0] or 0), t[1]))
    assert [(d["score"], d["name"]) for d in got] == want
    got = run(f"SELECT DISTINCT b.score, b.name {J} ORDER BY b.score NULLS LAST, b.name DESC")
    want = sorted(sorted(vals, key=lambda t: t[1], reverse=True), key=lambda t: (t[0] is None, t[0] or 0))
    assert [(d["score"], d["name"]) for d in got] == want
    got = run(f"SELECT DISTINCT b.score, b.name {J} ORDER BY b.score DESC NULLS FIRST, b.name LIMIT 12")
    want = sorted(vals, key=lambda t: (t[0] is not None, -(t[0] or 0), t[1]))[:12]
    assert [(d["score"], d["name"]) for d in got] == want


# --------------------------------------------------- execute(devices=[...]): one call over several contexts
def _rand_tables(seed, n_a, n_b, dominant=False, strands="+-"):
    r = np.random.default_rng(seed)

    def tbl(n, tag, uniform_len=None):
        ch = r.integers(1, 6, n)
        if dominant:   # one chromosome heavier than a device's share: cut by row ranges (shard.plan_units)
            ch[r.random(n) < 0.8] = 3
        st = r.integers(0, 60_000, n)
        ln = np.full(n, uniform_len) if uniform_len else r.integers(1, 400, n)
        return pa.table({"chrom": pa.array([f"chr{int(c)}" for c in ch]), "start": pa.array(st, pa.int32()),
                         "end": pa.array(st + ln, pa.int32()), "name": pa.array([f"{tag}{i}" for i in range(n)]),
                         "score": pa.array(r.integers(0, 50, n), pa.int32()),
                         "strand": pa.array([strands[int(k)] for k in r.integers(0, len(strands), n)])})
    return {"peaks": tbl(n_a, "p"), "genes": tbl(n_b, "g", uniform_len=120 if seed % 2 else None)}


@pytest.mark.parametrize("dominant", [False, True])
def test_execute_fans_out_over_devices(dominant):
    """``execute(plan, tables, devices=[...])`` -- chromosomes sharded over the contexts
    (``devices=[0, 0]``: two contexts on the one GPU of this box, a third for an odd count), each shard
    joined and projected on its own, the pieces concatenated -- returns what the single-device call
    returns, for every operator, with a dominant chromosome too."""
    T = ["peaks", "genes"]
    tables = _rand_tables(41 + int(dominant), 3000, 5000, dominant)
    J = "FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval"
    X = "CROSS JOIN LATERAL NEAREST(genes, reference := a.interval"
    queries = [
        f"SELECT a.name, b.name AS g, a.start, b.score {J}",
        f"SELECT a.name, b.name AS g {J} AND a.strand = b.strand WHERE a.score > 10",
        "SELECT a.name, a.score FROM peaks a SEMI JOIN genes b ON a.interval INTERSECTS b.interval",
        "SELECT a.name FROM peaks a ANTI JOIN genes b ON a.interval INTERSECTS b.interval",
        ('SELECT a.chrom, a.start, a."end", COUNT(b.chrom) AS n FROM peaks a LEFT JOIN genes b '
         'ON a.interval INTERSECTS b.interval GROUP BY a.chrom, a.start, a."end"'),
        f"SELECT a.name, b.start AS b_start, b.end AS b_end, b.distance AS d FROM peaks a {X}, k := 1, signed := true) b",
        f"SELECT a.name, b.start AS b_start, b.end AS b_end, b.distance AS d FROM peaks a {X}, k := 3) b",
        f"SELECT a.name, b.start AS b_start, b.end AS b_end, b.distance AS d FROM peaks a {X}, k := 2, stranded := true, signed := true) b",
        f"SELECT a.chrom, COUNT(*) AS n, SUM(b.score) AS t {J} GROUP BY a.chrom ORDER BY a.chrom",
        f"SELECT DISTINCT a.name {J} ORDER BY a.name LIMIT 17 OFFSET 5",
    ]
    for q in queries:
        plan = transpile(q, T, dialect="hip")
        want = execute(plan, tables)
        for devices in ([0, 0], [0, 0, 0]):
            got = execute(plan, tables, devices=devices)
            assert got.column_names == want.column_names, q
            if "ORDER BY" in q:
                assert got.to_pylist() == want.to_pylist(), (q, devices)
            else:
                assert rows_of(got) == rows_of(want), (q, devices)
        assert want.num_rows > 0, q
    # raw indices come back as GLOBAL row ids
    plan = transpile(f"SELECT a.name, b.name AS g {J}", T, dialect="hip")
    ra, rb = execute(plan, tables, return_indices=True)
    ga, gb = execute(plan, tables, devices=[0, 0], return_indices=True)
    assert sorted(zip(ra.tolist(), rb.tolist())) == sorted(zip(ga.tolist(), gb.tolist()))
    plan = transpile("SELECT a.name FROM peaks a ANTI JOIN genes b ON a.interval INTERSECTS b.interval", T, dialect="hip")
    assert np.array_equal(np.sort(execute(plan, tables, return_indices=True)), execute(plan, tables, devices=[0, 0], return_indices=True))
    plan = transpile(f"SELECT a.name, b.distance AS d FROM peaks a {X}, k := 2) b", T, dialect="hip")
    w = execute(plan, tables, return_indices=True)
    g = execute(plan, tables, devices=[0, 0, 0], return_indices=True)
    assert np.array_equal(w[0], g[0]) and np.array_equal(w[2], g[2])
    assert np.array_equal(tables["genes"]["start"].to_numpy()[w[1]], tables["genes"]["start"].to_numpy()[g[1]])
    # one device named explicitly = the single-device call; an empty list is an error
    assert rows_of(execute(plan, tables, devices=[0])) == rows_of(execute(plan, tables))
    with pytest.raises(ValueError):
        execute(plan, tables, devices=[])


def test_execute_devices_with_empty_and_one_sided_shards():
    T = ["peaks", "genes"]
    q = "SELECT a.name, b.name AS g FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval"
    plan = transpile(q, T, dialect="hip")
    # fewer chromosomes than devices; a chromosome present on one side only; no left rows at all
    t = {"peaks": make_table([("chr1", 100, 200, "p1", 0, "+"), ("chr9", 5, 50, "p2", 0, "+")]),
         "genes": make_table([("chr1", 150, 250, "g1", 0, "+"), ("chr2", 1, 9, "g2", 0, "-")])}
    assert rows_of(execute(plan, t, devices=[0, 0, 0])) == [("p1", "g1")]
    anti = transpile("SELECT a.name FROM peaks a ANTI JOIN genes b ON a.interval INTERSECTS b.interval", T, dialect="hip")
    assert rows_of(execute(anti, t, devices=[0, 0, 0])) == [("p2",)]
    empty = {"peaks": make_table([]), "genes": t["genes"]}
    out = execute(plan, empty, devices=[0, 0])
    assert out.num_rows == 0 and out.column_names == ["name", "g"]


def test_stranded_nearest_on_a_genome_wider_than_half_the_axis():
    """folding the strand into the partition id doubled the linearised span, so a
    whole genome with rows on both strands (~6.2e9 for hg38) failed with GIQL_ERR_SPAN although the same
    tables ran unstranded.  The '+' and '-' rows are now two NEAREST problems on the same axis."""
    r = np.random.default_rng(77)
    lens = [248_956_422, 242_193_529, 198_295_559, 190_214_555, 181_538_259, 170_805_979, 159_345_973, 145_138_636,
            138_394_717, 133_797_422, 135_086_622, 133_275_309, 114_364_328, 107_043_718, 101_991_189, 90_338_345,
            83_257_441, 80_373_285, 58_617_616, 64_444_167, 46_709_983, 50_818_468, 156_040_895, 57_227_415]

    def tbl(n, tag):
        ch = r.integers(0, 24, n)
        st = np.array([int(r.integers(0, lens[c] - 5000)) for c in ch])
        st[:24] = [lens[c] - 5000 for c in range(24)]   # every chromosome spans its whole length
        ch[:24] = np.arange(24)
        return pa.table({"chrom": pa.array([f"chr{int(c) + 1}" for c in ch]), "start": pa.array(st, pa.int32()),
                         "end": pa.array(st + r.integers(1, 4000, n), pa.int32()), "name": pa.array([f"{tag}{i}" for i in range(n)]),
                         "score": pa.array(np.zeros(n), pa.int32()), "strand": pa.array(["+-"[int(k)] for k in r.integers(0, 2, n)])})
    t = {"peaks": tbl(4000, "p"), "genes": tbl(6000, "g")}
    q = ("SELECT a.name, b.name AS g, b.distance AS d FROM peaks a CROSS JOIN LATERAL "
         "NEAREST(genes, reference := a.interval, k := 1, stranded := true, signed := true) b")
    out = execute(transpile(q, ["peaks", "genes"], dialect="hip"), t)
    assert out.num_rows == 4000
    P, G = t["peaks"].to_pylist(), t["genes"].to_pylist()
    got = {d["name"]: (d["g"], d["d"]) for d in out.to_pylist()}
    gi = {g["name"]: g for g in G}
    for p in P[:300]:   # brute force on a sample: the nearest same-strand gene's distance (sign flipped for '-')
        best = None
        for g in G:
            if g["chrom"] != p["chrom"] or g["strand"] != p["strand"]:
                continue
            if g["start"] < p["end"] and g["end"] > p["start"]:
                d = 0
            elif g["end"] <= p["start"]:
                d = -(p["start"] - g["end"] + 1)
            else:
                d = g["start"] - p["end"] + 1
            key = (abs(d), g["start"], g["end"])
            if best is None or key < best[0]:
                best = (key, d)
        d = best[1] * (-1 if p["strand"] == "-" else 1)
        assert got[p["name"]][1] == d, p
        g = gi[got[p["name"]][0]]
        assert g["chrom"] == p["chrom"] and g["strand"] == p["strand"]


def test_large_result_columns_come_back_through_pinned_memory(monkeypatch, peaks_genes):
    # execute._to_host: results past a size threshold are copied into page-locked memory from torch's caching host
    # allocator (the Arrow column wraps it); forced here for a small result, with and without NULLs in the column
    from giql_amd import execute as X

    q = "SELECT a.name, a.score, b.start AS s FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval"
    plan = transpile(q, tables=["peaks", "genes"], dialect="hip")
    genes = peaks_genes["genes"]
    peaks = peaks_genes["peaks"].set_column(4, "score", pa.array([10, 20, None, 30, 35], pa.int32()))
    t = {"peaks": peaks, "genes": genes}
    want = rows_of(execute(plan, t))
    monkeypatch.setattr(X, "_PINNED_MIN_BYTES", 1)
    got = execute(plan, t)
    assert rows_of(got) == want and len(want) > 0


def _long_table(n, seed, backing="arrow"):
    names = np.array(["chr1", "chr2", "chr3"])
    r = np.random.default_rng(seed)
    s = r.integers(0, 50_000_000, n).astype(np.int32)
    e = (s + r.integers(1, 300, n)).astype(np.int32)
    rest = {"name": pa.array(np.arange(n).astype(str)), "score": r.integers(0, 9, n).astype(np.int32),
            "strand": pa.array(np.array(["+", "-"])[r.integers(0, 2, n)])}
    cols = {"chrom": pa.array(names[r.integers(0, 3, n)]), "start": pa.array(s), "end": pa.array(e), **rest}
    return pa.table(cols), s, e


_SEMI_Q = ("SELECT a.start, a.score FROM peaks a SEMI JOIN genes b ON a.interval INTERSECTS b.interval "
           "AND a.strand = b.strand")


def test_long_pinned_tables_are_uploaded_once(monkeypatch):
    # execute._device_side keeps the device copy of a long table's (chrom id, start, end) for the next query over the
    # same buffers -- for tables the caller has PINNED (giql_amd.pin: the promise that their memory does not change);
    # another table of the same shape is another entry; unpin / clear_caches drop what was derived
    import giql_amd
    from giql_amd import execute as X
    from giql_amd.engine import DeviceSide

    peaks, genes, genes2 = (giql_amd.pin(_long_table(n, seed)[0]) for n, seed in
                            ((1_100_000, 1), (1_050_000, 2), (1_050_000, 5)))
    t = {"peaks": peaks, "genes": genes}
    plan = transpile(_SEMI_Q, tables=["peaks", "genes"], dialect="hip")
    uploads = []
    real = DeviceSide.from_numpy.__func__
    monkeypatch.setattr(DeviceSide, "from_numpy", classmethod(lambda cls, *a, **k: uploads.append(1) or real(cls, *a, **k)))
    monkeypatch.setattr(X, "_SIDES_CACHE", None)
    first = execute(plan, t)
    n1 = len(uploads)
    second = execute(plan, {"peaks": peaks.table, "genes": genes})   # (the table itself stands for its pin)
    assert n1 == 2 and len(uploads) == 2 and second.equals(first) and first.num_rows > 1000
    info = X.cache_info()
    assert info["sides"] == 2 and info["hbm_bytes"] == 12 * (1_100_000 + 1_050_000)
    third = execute(plan, dict(t, genes=genes2))
    # (both sides again: the chromosome ids belong to the PAIR's shared dictionary)
    assert len(uploads) == 4 and not third.equals(first)
    genes2.unpin()      # ... drops the entries derived from it (both sides of that pair: one key each)
    assert X.cache_info()["sides"] == 3
    held = X.clear_caches()
    assert held["sides"] == 3 and X.cache_info()["sides"] == 0 and X.cache_info()["codes"] == 0
    assert execute(plan, t).equals(first) and len(uploads) == 6     # derived again, still pinned
    assert execute(plan, t).equals(first) and len(uploads) == 6
    monkeypatch.setattr(X, "_SIDES_CACHE_SLOTS", 0)
    assert execute(plan, t).equals(first) and len(uploads) == 8
    peaks.unpin(), genes.unpin()
    assert X.cache_info()["pinned_chunks"] == 0


@pytest.mark.parametrize("backing", ["pa.array(numpy)", "from_pandas"])
def test_tables_changed_in_place_between_calls_give_the_new_rows(backing):
    # an Arrow column built from numpy / pandas aliases that memory zero-copy; a device copy
    # keyed by buffer addresses would serve the OLD coordinates after an in-place update.  Nothing is kept for such
    # tables unless they are pinned.
    import pandas as pd

    from giql_amd import execute as X

    X.clear_caches()
    n = 1_200_000
    _t, s, e = _long_table(n, 11)
    g, gs, ge = _long_table(1_050_000, 12)
    chrom = np.array(["chr1", "chr2", "chr3"])[np.random.default_rng(1).integers(0, 3, n)]
    if backing == "from_pandas":
        df = pd.DataFrame({"start": s, "end": e, "score": np.arange(n, dtype=np.int32) % 9})
        peaks = pa.Table.from_pandas(df, preserve_index=False).append_column("chrom", pa.array(chrom))
        s_mem, e_mem = df["start"].to_numpy(), df["end"].to_numpy()
        assert peaks.column("start").chunk(0).buffers()[1].address == s_mem.ctypes.data     # zero-copy: the premise
    else:
        peaks = pa.table({"chrom": pa.array(chrom), "start": pa.array(s), "end": pa.array(e),
                          "score": pa.array(np.arange(n, dtype=np.int32) % 9)})
        s_mem, e_mem = s, e
        assert peaks.column("start").chunk(0).buffers()[1].address == s.ctypes.data
    q = "SELECT a.start, a.score FROM peaks a SEMI JOIN genes b ON a.interval INTERSECTS b.interval"
    plan = transpile(q, tables=["peaks", "genes"], dialect="hip")
    t = {"peaks": peaks, "genes": g}
    first = execute(plan, t)
    assert X.cache_info()["sides"] == 0          # nothing kept: the memory is not the caller's promise
    # move every peak far away from every gene, in place
    s_mem += 60_000_000
    e_mem += 60_000_000
    second = execute(plan, t)
    assert first.num_rows > 1000 and second.num_rows == 0
    s_mem -= 60_000_000
    e_mem -= 60_000_000
    assert execute(plan, t).equals(first)


def test_pinned_table_refresh_after_an_in_place_change():
    import giql_amd
    from giql_amd import execute as X

    X.clear_caches()
    _t, s, e = _long_table(1_200_000, 21)
    chrom = np.array(["chr1", "chr2", "chr3"])[np.random.default_rng(2).integers(0, 3, s.size)]
    g = _long_table(1_050_000, 22)[0]
    with giql_amd.pin(pa.table({"chrom": pa.array(chrom), "start": pa.array(s), "end": pa.array(e)})) as peaks, \
            giql_amd.pin(g) as genes:
        plan = transpile("SELECT a.start FROM peaks a SEMI JOIN genes b ON a.interval INTERSECTS b.interval",
                         tables=["peaks", "genes"], dialect="hip")
        t = {"peaks": peaks, "genes": genes}
        first = execute(plan, t)
        assert X.cache_info()["sides"] == 2
        s += 60_000_000
        e += 60_000_000
        peaks.refresh()                      # the documented contract after an in-place change
        assert X.cache_info()["sides"] == 1
        assert execute(plan, t).num_rows == 0 and first.num_rows > 1000
    assert X.cache_info()["sides"] == 0 and X.cache_info()["pinned_chunks"] == 0


def test_immutable_buffers_are_kept_without_a_pin(monkeypatch, tmp_path):
    # memory-mapped / IPC-read tables cannot be written behind Arrow's back: kept implicitly
    import pyarrow.ipc as ipc

    from giql_amd import execute as X
    from giql_amd.engine import DeviceSide

    X.clear_caches()
    tabs = {}
    for name, (n, seed) in {"peaks": (1_100_000, 31), "genes": (1_050_000, 32)}.items():
        path = tmp_path / f"{name}.arrow"
        with ipc.new_file(str(path), _long_table(n, seed)[0].schema) as w:
            w.write_table(_long_table(n, seed)[0])
        tabs[name] = ipc.open_file(pa.memory_map(str(path))).read_all()
        assert not tabs[name].column("start").chunk(0).buffers()[1].is_mutable
    uploads = []
    real = DeviceSide.from_numpy.__func__
    monkeypatch.setattr(DeviceSide, "from_numpy", classmethod(lambda cls, *a, **k: uploads.append(1) or real(cls, *a, **k)))
    plan = transpile(_SEMI_Q, tables=["peaks", "genes"], dialect="hip")
    first = execute(plan, tabs)
    assert execute(plan, tabs).equals(first) and len(uploads) == 2 and X.cache_info()["sides"] == 2
    X.clear_caches()
