"""transpile(dialect="hip"): plan structure and the decline / raise matrix.

Modelled on the reference's SQL-structure tests for the DuckDB IEJoin dialect
(tests/test_duckdb_iejoin.py:145-3217, 6610-7365 of the reference): which shapes
engage the path, which decline, which are user errors.  No GPU needed.
"""

import pytest

from giql_amd.plan import JoinPlan, PlanSide, Projection, is_plan_string
from giql_amd.table import Table
from giql_amd.transpile import HipDeclined, build_plan, transpile

Q_INNER = """
    SELECT a.chrom AS a_chrom, a.start AS a_start, a.end AS a_end,
           b.chrom AS b_chrom, b.start AS b_start, b.end AS b_end
    FROM peaks a
    JOIN genes b ON a.interval INTERSECTS b.interval
"""


def test_transpile_returns_a_plan_string_for_hip():
    s = transpile(Q_INNER, tables=["peaks", "genes"], dialect="hip")
    assert isinstance(s, str) and is_plan_string(s)
    plan = JoinPlan.from_string(s)
    assert plan.kind == "INNER"
    assert (plan.left.table, plan.left.alias) == ("peaks", "a")
    assert (plan.right.table, plan.right.alias) == ("genes", "b")
    assert [p.name for p in plan.projection] == ["a_chrom", "a_start", "a_end", "b_chrom", "b_start", "b_end"]
    assert [p.side for p in plan.projection] == ["l", "l", "l", "r", "r", "r"]
    assert JoinPlan.from_string(plan.to_string()) == plan


def test_literal_range_predicate_matches_reference_text():
    # README.md:35-50 / tests/expanders/test_intersects.py:83-85 of the reference
    sql = transpile("SELECT * FROM peaks WHERE interval INTERSECTS 'chr1:1000-2000'", tables=["peaks"])
    assert sql == 'SELECT * FROM peaks WHERE ("chrom" = \'chr1\' AND "start" < 2000 AND "end" > 1000)'


def test_custom_columns_and_encoding_reach_the_plan():
    tables = [
        Table("variants", genomic_col="position", chrom_col="chr", start_col="pos_start",
              end_col="pos_end", strand_col=None, coordinate_system="1based", interval_type="closed"),
        "genes",
    ]
    plan = build_plan(
        "SELECT v.id, g.name FROM variants v JOIN genes g ON v.position INTERSECTS g.interval", tables)
    assert plan.left == PlanSide("variants", "v", "chr", "pos_start", "pos_end", "1based", "closed")
    assert plan.right.encoding == ("0based", "half_open")
    assert plan.projection == (Projection("l", "id", "id"), Projection("r", "name", "name"))


def test_unregistered_table_uses_default_columns():
    plan = build_plan("SELECT a.start FROM x a JOIN y b ON a.interval INTERSECTS b.interval")
    assert plan.left.chrom_col == "chrom" and plan.right.end_col == "end"


@pytest.mark.parametrize("kw,kind", [("", "INNER"), ("INNER ", "INNER"), ("CROSS ", "INNER"),
                                     ("SEMI ", "SEMI"), ("ANTI ", "ANTI"), ("LEFT SEMI ", "SEMI")])
def test_join_kinds(kw, kind):
    plan = build_plan(f"SELECT a.start FROM peaks a {kw}JOIN genes b ON a.interval INTERSECTS b.interval",
                      ["peaks", "genes"])
    assert plan.kind == kind


def test_from_side_orientation_swap():
    # left is always the FROM table, whichever operand the user wrote first
    plan = build_plan("SELECT a.start, b.start FROM peaks a JOIN genes b ON b.interval INTERSECTS a.interval",
                      ["peaks", "genes"])
    assert plan.left.table == "peaks" and plan.right.table == "genes"


def test_implicit_cross_join_in_where_engages():
    plan = build_plan("SELECT a.start, b.start FROM peaks a, genes b WHERE a.interval INTERSECTS b.interval",
                      ["peaks", "genes"])
    assert plan.kind == "INNER"


def test_case_insensitive_aliases():
    plan = build_plan("SELECT A.start FROM peaks A JOIN genes B ON a.interval INTERSECTS b.interval",
                      ["peaks", "genes"])
    assert plan.left.alias == "a"


@pytest.mark.parametrize("query", [
    "SELECT * FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval",
    "SELECT a.* FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval",
    "SELECT a.start FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval",
    "SELECT a.start FROM peaks a RIGHT JOIN genes b ON a.interval INTERSECTS b.interval",
    "SELECT a.start FROM peaks a FULL OUTER JOIN genes b ON a.interval INTERSECTS b.interval",
    "SELECT a.start FROM peaks a NATURAL JOIN genes b",
    "SELECT a.start FROM peaks a JOIN peaks b ON a.interval INTERSECTS b.interval",
    "SELECT a.start FROM peaks a JOIN genes a ON a.interval INTERSECTS a.interval",
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval OR a.score > 5",
    "SELECT a.start FROM peaks a JOIN genes b ON NOT a.interval INTERSECTS b.interval",
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND a.name LIKE 'p%'",
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND a.score IS TRUE",
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND a.score IN (SELECT 1)",
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND a.score IN (1, b.score)",
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND a.score % 2 = 1",
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND 1 + 1 = 2",
    # (13 comparisons once in conjunctive normal form were declined until round 4; a join's residual past the cap now
    # travels as one boolean program: test_an_in_list_past_the_normal_forms_cap_is_a_boolean_program)
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND SQRT(a.score) > 5",
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND 1 = 1",
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND a.interval INTERSECTS b.interval",
    "SELECT a.start FROM peaks a JOIN genes b ON a.score > 5",
    "SELECT a.start + 1 FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval",
    "SELECT a.start FROM peaks a JOIN genes b USING (chrom)",                      # USING without any INTERSECTS
    "SELECT a.chrom FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval GROUP BY a.chrom "
    "HAVING COUNT(*) + 1 > 1",
    "SELECT a.chrom FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval GROUP BY a.chrom "
    "HAVING SUM(a.score) > (SELECT AVG(score) FROM peaks)",
    "SELECT a.start FROM peaks a SEMI JOIN genes b ON TRUE WHERE a.interval INTERSECTS b.interval",
    "SELECT a.start FROM peaks a, genes b, exons c WHERE a.interval INTERSECTS b.interval",
    "WITH x AS (SELECT 1) SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval",
    "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS 'chr1:1-10'",
    "SELECT a.start FROM peaks a CROSS JOIN LATERAL NEAREST(genes, reference := a.interval, k := 1048577) b",
])
def test_valid_but_unsupported_shapes_decline(query):
    with pytest.raises(HipDeclined):
        build_plan(query, ["peaks", "genes"])


def test_an_in_list_past_the_normal_forms_cap_is_a_boolean_program():
    q = ("SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND a.score IN "
         "(1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13)")
    (r,) = build_plan(q, ["peaks", "genes"]).residuals
    assert r.op == "istrue" and r.lhs.kind == "expr" and r.lhs.value[1] == "or" and len(r.lhs.value[2]) == 13
    # ... while the literal-range filter and CLUSTER / MERGE predicates keep the normal form and its cap
    with pytest.raises(HipDeclined, match="too large"):
        build_plan("SELECT * FROM peaks WHERE interval INTERSECTS 'chr1:1-10' AND score IN "
                   "(1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13)", ["peaks"])


def test_boolean_having_arrives_in_conjunctive_normal_form():
    # the reference hands HAVING to the engine verbatim (intersects_duckdb.py:1336-1400)
    q = ("SELECT a.chrom, COUNT(*) AS n FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval GROUP BY a.chrom "
         "HAVING (COUNT(*) > 5 OR NOT SUM(b.score) >= 3) AND a.chrom IN ('chr1', 'chr2') AND MAX(a.score) IS NOT NULL")
    p = build_plan(q, ["peaks", "genes"])
    assert [(h.lhs.value, h.op, h.rhs.value, h.group) for h in p.having] == \
        [("n", ">", 5, 1), ("__giql_h0", "<", 3, 1), ("chrom", "=", "chr1", 2), ("chrom", "=", "chr2", 2),
         ("__giql_h1", "notnull", 0, 0)]
    assert JoinPlan.from_string(p.to_string()) == p


def test_arithmetic_operands_become_expression_trees():
    # the overlap-fraction recipes (docs/recipes/intersect.rst:144-190), inlined as text upstream
    base = "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval "
    p = build_plan(base + "AND (LEAST(a.end, b.end) - GREATEST(a.start, b.start)) >= 0.5 * (a.end - a.start)", ["peaks", "genes"])
    (r,) = p.residuals
    assert (r.lhs.kind, r.op, r.rhs.kind) == ("expr", ">=", "expr")
    assert r.lhs.value == ["fn", "-", [["fn", "least", [["l", "end"], ["r", "end"]]], ["fn", "greatest", [["l", "start"], ["r", "start"]]]]]
    assert r.rhs.value == ["fn", "*", [["float", 0.5], ["fn", "-", [["l", "end"], ["l", "start"]]]]]
    assert JoinPlan.from_string(p.to_string()) == p
    # precedence, unary minus, a parenthesised operand next to a parenthesised condition
    p = build_plan(base + "AND (a.score) > 5 AND (a.score + 1 > 5 OR -a.score * 2 + b.score / 2.0 < ABS(b.score - 3))",
                   ["peaks", "genes"])
    assert [(r.lhs.kind, r.op, r.group) for r in p.residuals] == [("l", ">", 0), ("expr", ">", 1), ("expr", "<", 1)]
    assert p.residuals[2].lhs.value == ["fn", "+", [["fn", "*", [["fn", "neg", [["l", "score"]]], ["int", 2]]],
                                                    ["fn", "/", [["r", "score"], ["float", 2.0]]]]]
    # SEMI / ANTI: a WHERE expression cannot read the right side either; strings do not take part in arithmetic
    with pytest.raises(ValueError, match="right side"):
        build_plan("SELECT a.start FROM peaks a SEMI JOIN genes b ON a.interval INTERSECTS b.interval WHERE a.score + b.score > 1",
                   ["peaks", "genes"])
    with pytest.raises(ValueError, match="arithmetic"):
        build_plan(base + "AND a.score + 'x' > 1", ["peaks", "genes"])
    # HAVING and CLUSTER predicates keep declining arithmetic
    with pytest.raises(HipDeclined):
        build_plan("SELECT *, CLUSTER(interval, predicate := depth + 1 = PREV(depth)) AS cid FROM peaks", ["peaks"])


def _res(plan):
    return [(r.clause, r.lhs.value, r.op, r.rhs.value, r.group) for r in plan.residuals]


def test_boolean_residuals_arrive_in_conjunctive_normal_form():
    # the reference inlines any such extra as SQL text (_classify_extras, intersects_duckdb.py:889-912);
    # here OR / NOT / parentheses / BETWEEN / IN / IS NULL become an AND of OR-groups of comparisons
    base = "SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval "
    p = build_plan(base + "AND (a.score > 5)", ["peaks", "genes"])
    assert _res(p) == [("on", "score", ">", 5, 0)]
    p = build_plan(base + "AND NOT a.score > 5", ["peaks", "genes"])
    assert _res(p) == [("on", "score", "<=", 5, 0)]
    p = build_plan(base + "AND (a.score > 5 OR b.score < 2) WHERE a.name = 'x' OR a.name = 'y' OR b.name <> a.name",
                   ["peaks", "genes"])
    assert _res(p) == [("on", "score", ">", 5, 1), ("on", "score", "<", 2, 1),
                       ("where", "name", "=", "x", 2), ("where", "name", "=", "y", 2), ("where", "name", "!=", "name", 2)]
    # NOT over OR / AND: De Morgan, the comparisons flipped
    p = build_plan(base + "AND NOT (a.score > 5 OR b.score <= 2)", ["peaks", "genes"])
    assert _res(p) == [("on", "score", "<=", 5, 0), ("on", "score", ">", 2, 0)]
    p = build_plan(base + "WHERE NOT (a.score = 5 AND NOT b.score <> 2)", ["peaks", "genes"])
    assert _res(p) == [("where", "score", "!=", 5, 1), ("where", "score", "!=", 2, 1)]
    # OR over AND distributes
    p = build_plan(base + "WHERE (a.score > 1 AND b.score > 2) OR a.score = 0", ["peaks", "genes"])
    assert _res(p) == [("where", "score", ">", 1, 1), ("where", "score", "=", 0, 1),
                       ("where", "score", ">", 2, 2), ("where", "score", "=", 0, 2)]
    p = build_plan(base + "AND a.score BETWEEN 2 AND 4 AND b.score NOT BETWEEN 1 AND 3", ["peaks", "genes"])
    assert _res(p) == [("on", "score", ">=", 2, 0), ("on", "score", "<=", 4, 0),
                       ("on", "score", "<", 1, 1), ("on", "score", ">", 3, 1)]
    p = build_plan(base + "WHERE a.score IN (1, 2) AND b.name NOT IN ('u', 'v') AND a.name IS NOT NULL "
                   "AND NOT b.name IS NULL AND a.score IS NULL", ["peaks", "genes"])
    assert _res(p) == [("where", "score", "=", 1, 1), ("where", "score", "=", 2, 1),
                       ("where", "name", "!=", "u", 0), ("where", "name", "!=", "v", 0),
                       ("where", "name", "notnull", 0, 0), ("where", "name", "notnull", 0, 0),
                       ("where", "score", "isnull", 0, 0)]
    assert JoinPlan.from_string(p.to_string()) == p
    # a parenthesised INTERSECTS is still the join's spatial predicate
    p = build_plan("SELECT a.start FROM peaks a JOIN genes b ON (a.interval INTERSECTS b.interval) AND (a.score > 1)",
                   ["peaks", "genes"])
    assert p.kind == "INNER" and _res(p) == [("on", "score", ">", 1, 0)]
    # SEMI / ANTI: the WHERE clause still cannot name the right side, inside an OR either
    with pytest.raises(ValueError, match="right side"):
        build_plan("SELECT a.start FROM peaks a SEMI JOIN genes b ON a.interval INTERSECTS b.interval "
                   "WHERE a.score > 1 OR b.score > 1", ["peaks", "genes"])


@pytest.mark.parametrize("query,match", [
    ("SELECT start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval", "Unqualified"),
    ("SELECT c.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval", "Unknown table"),
    ("SELECT b.start FROM peaks a SEMI JOIN genes b ON a.interval INTERSECTS b.interval", "right side"),
    ("SELECT b.start FROM peaks a ANTI JOIN genes b ON a.interval INTERSECTS b.interval", "right side"),
    ("SELECT a.start FROM peaks a JOIN genes b ON a.start INTERSECTS b.interval", "genomic"),
])
def test_user_mistakes_raise_value_error(query, match):
    with pytest.raises(ValueError, match=match) as ei:
        build_plan(query, ["peaks", "genes"])
    assert not isinstance(ei.value, HipDeclined)


def test_residual_predicates_are_lowered_into_the_plan():
    # the reference inlines extra ON / WHERE conjuncts into the per-chromosome join
    # (intersects_duckdb.py:1157-1177); the hip plan carries them as comparisons
    q = ("SELECT a.start FROM peaks a JOIN genes b ON a.strand = b.strand AND a.interval INTERSECTS b.interval "
         "AND a.score >= 10 WHERE b.score <= -2.5 AND a.name <> 'x' AND 7 < b.score")
    plan = build_plan(q, ["peaks", "genes"])
    got = [(r.clause, r.lhs.kind, r.lhs.value, r.op, r.rhs.kind, r.rhs.value) for r in plan.residuals]
    assert got == [("on", "l", "strand", "=", "r", "strand"), ("on", "l", "score", ">=", "int", 10),
                   ("where", "r", "score", "<=", "float", -2.5), ("where", "l", "name", "!=", "str", "x"),
                   ("where", "int", 7, "<", "r", "score")]
    assert JoinPlan.from_string(plan.to_string()) == plan
    # implicit cross join: the INTERSECTS and its residuals all sit in WHERE
    q2 = "SELECT a.start FROM peaks a, genes b WHERE a.score != b.score AND a.interval INTERSECTS b.interval"
    p2 = build_plan(q2, ["peaks", "genes"])
    assert p2.kind == "INNER" and [(r.clause, r.op) for r in p2.residuals] == [("where", "!=")]
    # SEMI / ANTI keep the ON / WHERE distinction (WHERE is an outer filter, #200)
    q3 = ("SELECT a.start FROM peaks a ANTI JOIN genes b ON a.interval INTERSECTS b.interval "
          "AND a.strand = b.strand WHERE a.score > 1")
    p3 = build_plan(q3, ["peaks", "genes"])
    assert p3.kind == "ANTI" and [r.clause for r in p3.residuals] == ["on", "where"]
    # no residuals -> empty tuple, and older plan strings without the key still load
    bare = build_plan("SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval", ["peaks", "genes"])
    assert bare.residuals == ()
    assert JoinPlan.from_string(bare.to_string().replace(',"residuals":[]', "")) == bare


@pytest.mark.parametrize("query,match", [
    # qualifier mistakes in a residual are user errors (_validate_extra_qualifiers, :914-959)
    ("SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND score > 5", "qualified"),
    ("SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND c.score > 5", "unknown table"),
    ("SELECT a.start FROM peaks a SEMI JOIN genes b ON a.interval INTERSECTS b.interval WHERE b.score > 5", "right side"),
])
def test_residual_qualifier_mistakes_raise_value_error(query, match):
    with pytest.raises(ValueError, match=match) as ei:
        build_plan(query, ["peaks", "genes"])
    assert not isinstance(ei.value, HipDeclined)


def test_cluster_and_merge_shapes():
    # src/giql/expanders/cluster.py:81-205 / merge.py:62-183; argument spellings of
    # tests/test_cluster_parsing.py:22-75
    p = build_plan("SELECT *, CLUSTER(interval) AS cluster_id FROM peaks", ["peaks"])
    assert (p.kind, p.right, p.distance, p.stranded) == ("CLUSTER", None, 0, False)
    assert [(x.side, x.name) for x in p.projection] == [("star", "*"), ("cluster", "cluster_id")]
    for spelling in ("stranded := true", "stranded = true", "stranded => true"):
        q = f'SELECT chrom, start, "end", CLUSTER(interval, 1000, {spelling}) AS cid FROM peaks p WHERE p.score > 3'
        p = build_plan(q, ["peaks"])
        assert (p.distance, p.stranded, p.strand_col, len(p.residuals)) == (1000, True, "strand", 1)
        assert JoinPlan.from_string(p.to_string()) == p
    m = build_plan("SELECT MERGE(interval, 100), COUNT(*) AS n FROM peaks", ["peaks"])
    assert (m.kind, m.distance, [(x.side, x.name) for x in m.projection]) == ("MERGE", 100, [("count", "n")])
    assert build_plan("SELECT chrom, MERGE(interval) FROM peaks", ["peaks"]).projection == ()


@pytest.mark.parametrize("query,exc,match", [
    ("SELECT *, MERGE(interval) FROM peaks", ValueError, "star projection"),                   # merge.py:96-133
    ("SELECT CLUSTER(interval) AS a, CLUSTER(interval, 5) AS b FROM peaks", ValueError, "Multiple CLUSTER"),
    ("SELECT MERGE(interval), MERGE(interval, 5) FROM peaks", ValueError, "Multiple MERGE"),
    ("SELECT CLUSTER(interval) AS a, MERGE(interval) FROM peaks", ValueError, "cannot be combined"),
    ("SELECT score, MERGE(interval) FROM peaks", ValueError, "non-aggregated"),              # merge.py:279-296
    ("SELECT *, *, CLUSTER(interval) AS c FROM peaks", ValueError, "multiple star"),          # cluster.py:182-196
    ("SELECT CLUSTER(stranded := true) AS c FROM peaks", ValueError, "genomic interval column"),
    ("SELECT *, CLUSTER(start) AS c FROM peaks", ValueError, "genomic column"),
    ("SELECT *, CLUSTER(interval) AS c FROM peaks ORDER BY chrom", HipDeclined, "ORDER"),
    ("SELECT *, CLUSTER(interval) FROM peaks", HipDeclined, "alias"),
    ("SELECT MERGE(interval), SUM(score) AS s FROM peaks", HipDeclined, "function call"),
])
def test_cluster_merge_mistakes_and_declines(query, exc, match):
    with pytest.raises(exc, match=match) as ei:
        build_plan(query, ["peaks"])
    if exc is ValueError:
        assert not isinstance(ei.value, HipDeclined)


def test_literal_range_filter_lowers_to_three_comparisons():
    # BASELINE config 1 (README.md:35-50): chrom = 'chr1' AND start < 2000 AND end > 1000
    # (src/giql/expanders/intersects.py:85-107, 204-222)
    p = build_plan("SELECT * FROM peaks WHERE interval INTERSECTS 'chr1:1000-2000'", ["peaks"])
    assert p.kind == "FILTER" and p.right is None
    assert [(r.lhs.value, r.op, r.rhs.value) for r in p.residuals] == [
        ("chrom", "=", "chr1"), ("start", "<", 2000), ("end", ">", 1000)]
    assert JoinPlan.from_string(p.to_string()) == p
    q = "SELECT name, score AS s FROM peaks p WHERE p.score > 5 AND p.interval INTERSECTS 'chr2:5-9'"
    p = build_plan(q, ["peaks"])
    assert [(x.column, x.name) for x in p.projection] == [("name", "name"), ("score", "s")]
    assert [(r.lhs.value, r.op, r.rhs.value) for r in p.residuals][3:] == [("score", ">", 5)]
    with pytest.raises(ValueError, match="Start must be less than end"):
        build_plan("SELECT * FROM peaks WHERE interval INTERSECTS 'chr1:2000-1000'", ["peaks"])
    with pytest.raises(HipDeclined):
        build_plan("SELECT * FROM peaks WHERE interval INTERSECTS 'chr1:1000-2000' OR score > 1", ["peaks"])
    with pytest.raises(HipDeclined):  # non-canonical encodings keep the reference's SQL path
        build_plan("SELECT * FROM peaks WHERE interval INTERSECTS 'chr1:1000-2000'",
                   [Table("peaks", coordinate_system="1based", interval_type="closed")])


COUNT_Q = ('SELECT a.chrom, a.start, a."end", COUNT(b.chrom) AS n FROM peaks a '
           'LEFT JOIN genes b ON a.interval INTERSECTS b.interval GROUP BY a.chrom, a.start, a."end"')


def test_count_overlaps_shape_engages_the_path():
    # reference tests/test_duckdb_iejoin.py:3220-3250 (the count_overlaps fast path, #209)
    plan = build_plan(COUNT_Q, ["peaks", "genes"])
    assert plan.kind == "COUNT"
    assert [(p.side, p.column, p.name) for p in plan.projection] == [
        ("l", "chrom", "chrom"), ("l", "start", "start"), ("l", "end", "end"), ("count", "chrom", "n")]
    assert JoinPlan.from_string(plan.to_string()) == plan
    outer = build_plan(COUNT_Q.replace("LEFT JOIN", "LEFT OUTER JOIN"), ["peaks", "genes"])
    assert outer == plan


@pytest.mark.parametrize("query", [
    # the reference's decline matrix for this shape, tests/test_duckdb_iejoin.py:3252-3446
    "SELECT a.chrom, COUNT(*) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval GROUP BY a.chrom",
    "SELECT a.chrom, SUM(b.score) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval GROUP BY a.chrom",
    "SELECT COUNT(b.chrom) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval",
    "SELECT a.chrom, COUNT(b.chrom) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval "
    "GROUP BY a.chrom ORDER BY a.chrom",
    "SELECT a.chrom, COUNT(b.chrom) AS n FROM peaks a LEFT JOIN peaks b ON a.interval INTERSECTS b.interval GROUP BY a.chrom",
    "SELECT a.chrom, COUNT(b.chrom) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval "
    "GROUP BY a.chrom, a.start",
    "SELECT a.score AS n, COUNT(b.chrom) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval "
    "GROUP BY a.score",
    "SELECT a.chrom, COUNT(b.chrom) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval "
    "WHERE a.score > 1 GROUP BY a.chrom",
    # shapes the hip target does not take even though the DuckDB matcher does / might
    "SELECT a.chrom, COUNT(DISTINCT b.name) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval "
    "GROUP BY a.chrom",
    "SELECT a.chrom, COUNT(b.chrom) FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval GROUP BY a.chrom",
    "SELECT a.chrom, COUNT(a.start) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval GROUP BY a.chrom",
    "SELECT a.chrom, b.start, COUNT(b.chrom) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval "
    "GROUP BY a.chrom",
    "SELECT a.chrom, COUNT(b.chrom) AS n FROM peaks a LEFT JOIN genes b ON a.interval INTERSECTS b.interval "
    "GROUP BY a.chrom HAVING COUNT(b.chrom) > 1",
])
def test_count_overlaps_lookalikes_decline(query):
    with pytest.raises(HipDeclined):
        build_plan(query, ["peaks", "genes"])


def test_inner_join_with_count_is_a_grouped_inner_join_not_count_overlaps():
    # an INNER join drops the zero-overlap keys, so this is the plain aggregate over the join's rows
    # (intersects_duckdb.py:1402-1644), not the zero-filling count_overlaps shape (:432-548)
    plan = build_plan("SELECT a.chrom, COUNT(b.chrom) AS n FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval "
                      "GROUP BY a.chrom", ["peaks", "genes"])
    assert plan.kind == "INNER" and plan.group_by == ("chrom",)
    assert [(a.func, a.side, a.column, a.name) for a in plan.aggregates] == [("COUNT", "r", "chrom", "n")]


def test_nearest_plan():
    plan = build_plan(
        "SELECT a.start AS a_start, b.start AS b_start, b.distance AS d FROM peaks a "
        "CROSS JOIN LATERAL NEAREST(genes, reference := a.interval, k := 1, max_distance := 100, "
        "signed := true) b", ["peaks", "genes"])
    assert plan.kind == "NEAREST" and plan.k == 1 and plan.max_distance == 100 and plan.signed
    assert plan.projection[2] == Projection("distance", "distance", "d")


def test_unknown_dialect_and_bad_table_config():
    with pytest.raises(ValueError, match="Unknown dialect"):
        transpile(Q_INNER, tables=["peaks", "genes"], dialect="postgres")
    with pytest.raises(ValueError, match="coordinate_system"):
        Table("x", coordinate_system="2based")
    with pytest.raises(ValueError, match="interval_type"):
        Table("x", interval_type="open")


def test_plugin_module_is_import_guarded():
    from giql_amd import plugin

    assert plugin.HAVE_GIQL in (True, False)  # importing never raises without giql/sqlglot


def test_nearest_k_and_stranded_lower_to_the_plan():
    plan = build_plan("SELECT a.start, b.start AS g, b.distance FROM peaks a CROSS JOIN LATERAL "
                      "NEAREST(genes, reference := a.interval, k := 3, stranded := true, signed := true, max_distance := 500) b",
                      ["peaks", "genes"])
    assert (plan.kind, plan.k, plan.stranded, plan.signed, plan.max_distance, plan.strand_col) == \
        ("NEAREST", 3, True, True, 500, "strand,strand")
    from giql_amd.plan import JoinPlan
    assert JoinPlan.from_string(plan.to_string()) == plan
    from giql_amd.table import Table
    with pytest.raises(HipDeclined):     # no strand column to match on
        build_plan("SELECT a.start FROM peaks a CROSS JOIN LATERAL NEAREST(genes, reference := a.interval, stranded := true) b",
                   [Table("peaks", strand_col=None), "genes"])
    with pytest.raises(ValueError, match="positive"):
        build_plan("SELECT a.start FROM peaks a CROSS JOIN LATERAL NEAREST(genes, reference := a.interval, k := 0) b", ["peaks", "genes"])


def test_chrom_encoding_takes_arrows_hash_and_agrees_with_the_numpy_path():
    # execute()'s host side: the shared chromosome dictionary of two Arrow string columns comes from Arrow's own
    # dictionary_encode (a Python-level pass over 110M strings took minutes), sorted like numpy.unique sorts
    np = pytest.importorskip("numpy")
    pa = pytest.importorskip("pyarrow")
    from giql_amd import execute as X

    rng = np.random.default_rng(0)
    names = np.array([f"chr{i}" for i in range(1, 25)] + ["chrX", "chrM"])
    av, bv = names[rng.integers(0, 26, 5000)], names[rng.integers(3, 20, 700)]
    want = X.encode_chroms(av, bv)                                    # numpy arrays: the generic path
    for a, b in [(pa.array(av), pa.array(bv)),
                 (pa.chunked_array([pa.array(av[:1000]), pa.array(av[1000:])]), pa.array(bv).dictionary_encode()),
                 (pa.chunked_array([pa.array(av[:10]).dictionary_encode(), pa.array(av[10:]).dictionary_encode()]),
                  pa.array(bv, pa.large_string()))]:
        ia, ib, d = X.encode_chroms(a, b)
        assert d == want[2] and (ia == want[0]).all() and (ib == want[1]).all() and ia.dtype == np.int32
    ia, ib, d = X.encode_chroms(pa.array([], pa.string()), pa.array(["b", "a", "b"]))
    assert ia.size == 0 and ib.tolist() == [1, 0, 1] and d == ["a", "b"]
    with pytest.raises(ValueError, match="NULL"):
        X.encode_chroms(pa.array(["x", None]), pa.array(["x"]))
    with pytest.raises(ValueError, match="NULL"):
        X.encode_chroms(pa.array(["x"]), pa.array(["x", None]).dictionary_encode())
    assert X._strand_codes(pa.array(["+", "-", None, ".", "?"]), 4).tolist() == [0, 1, 4, 2, 3]
    assert X._strand_codes(np.array(["+", "-", None], dtype=object), 5).tolist() == [0, 1, 5]
    with pytest.raises(ValueError, match="strands other than"):
        X._strand_codes(pa.array(["+", "*"]), 4)


def test_codes_of_long_columns_are_kept_for_the_next_query():
    np = pytest.importorskip("numpy")
    pa = pytest.importorskip("pyarrow")
    from giql_amd import execute as X

    names = np.array(["chr1", "chr2", "chrX"])
    rng = np.random.default_rng(1)
    a = pa.chunked_array([pa.array(names[rng.integers(0, 3, 700_000)]), pa.array(names[rng.integers(0, 3, 400_000)])])
    b = pa.array(names[rng.integers(1, 3, 50_000)])
    old = (X._CODES_CACHE_SLOTS, X._CODES_CACHE)
    pins = []
    try:
        X._CODES_CACHE_SLOTS, X._CODES_CACHE = 3, None
        # nothing is kept for memory that is not the caller's promise (Arrow's own pool buffers are mutable too:
        # giql_amd.pin is the contract) ...
        u1, u2 = X.encode_chroms(a, b), X.encode_chroms(a, b)
        assert u2[0] is not u1[0] and not X._CODES_CACHE
        # ... and everything for pinned tables
        pins = [X.pin(pa.table({"c": a})), X.pin(pa.table({"c": b})), X.pin(pa.table({"c": a.slice(5, 1_050_000)}))]
        r1 = X.encode_chroms(a, b)
        r2 = X.encode_chroms(a, b)
        assert r2[0] is r1[0] and r2[1] is r1[1] and r1[2] == ["chr1", "chr2", "chrX"] and not r1[0].flags.writeable
        assert (r1[0] == np.searchsorted(r1[2], np.concatenate([c.to_numpy(zero_copy_only=False) for c in a.chunks]))).all()
        r3 = X.encode_chroms(a.slice(5, 1_050_000), b)                 # other rows of the same buffers: another entry
        assert r3[0] is not r1[0] and (r3[0] == r1[0][5:1_050_005]).all()
        assert len(X._CODES_CACHE) <= 3                                # oldest out
        X._CODES_CACHE_SLOTS, X._CODES_CACHE = 0, None
        r4 = X.encode_chroms(a, b)
        assert X._CODES_CACHE is None and (r4[0] == r1[0]).all()
        X._CODES_CACHE_SLOTS = 3
        X.encode_chroms(a, b)
        assert len(X._CODES_CACHE) == 2          # the long column and the pair's shared encoding (b is short)
        pins[1].unpin()                          # the end of b's promise takes what was derived from b with it
        assert len(X._CODES_CACHE) == 1
    finally:
        for p in pins:
            p.unpin()
        X._CODES_CACHE_SLOTS, X._CODES_CACHE = old
    assert not X._PINNED
