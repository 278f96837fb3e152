"""giql's plugin hook for dialect='hip' (giql_amd/plugin.py), run on CPU with hand-built stand-ins
for the AST, the ExpansionContext and the resolver's metadata (tests/_ast_doubles.py) -- no giql,
no sqlglot.  The expander must (1) lower from the node and ``ctx.resolution`` -- reading the
canonical offsets off the fragments pass 2 already wrapped, never re-applying them from
``ctx.tables`` --, (2) produce the SAME plan the sqlglot-free mirror produces for the same query,
(3) install a finalizer that swaps the statement for the verbatim plan payload, and (4) defer every
shape the gate declines to the naive predicate without erroring.
"""

import pytest

import _ast_doubles as A
from giql_amd import plugin
from giql_amd.plan import PLAN_PREFIX, JoinPlan
from giql_amd.table import Table, build_tables
from giql_amd.transpile import build_plan

FALLBACK = object()


def run(root, node, tables, resolution="auto"):
    """Run the expander on `node` (the INTERSECTS inside `root`); returns (result, ctx, command)."""
    tbls = build_tables(tables)
    if resolution == "auto":
        cols = {}
        for arg in ("this", "expression"):
            c = node.args[arg]
            al = c.args["table"].args["this"]
            name = {t.args["alias"].args["this"].args["this"] if t.args.get("alias") else t.args["this"].args["this"]:
                    t.args["this"].args["this"] for t in [root.args["from_"].args["this"]] + [j.args["this"] for j in root.args["joins"]]}[al]
            t = tbls.get(name)
            cols[arg] = A.resolved(al, t, *( (t.chrom_col, t.start_col, t.end_col) if t else ()))
        resolution = A.OperatorResolution(columns=cols)
    ctx = A.ExpansionContext(tables=tbls, resolution=resolution)
    calls = []
    expander = plugin.make_expander(lambda n, c: calls.append((n, c)) or FALLBACK, lambda payload: ("COMMAND", payload))
    out = expander(node, ctx)
    return out, ctx, calls


def basic(items, on_extra=(), where=None, kind=None, side=None, swap=False, **kw):
    it = A.intersects(A.col("b", "interval"), A.col("a", "interval")) if swap else A.intersects(A.col("a", "interval"), A.col("b", "interval"))
    root = A.select(items, A.tbl("peaks", "a"), [A.join(A.tbl("genes", "b"), on=A.conj(it, *on_extra), kind=kind, side=side)],
                    where=where, **kw)
    return root, it


def test_inner_join_lowers_to_the_same_plan_as_the_mirror():
    root, it = basic([A.col("a", "name"), A.alias(A.col("b", "name"), "g")],
                     on_extra=[A.cmp("gt", A.col("a", "score"), A.lit(5))],
                     where=A.cmp("eq", A.col("b", "strand"), A.lit("+")),
                     order=[(A.col("a", "start"), True)], limit=3)
    out, ctx, calls = run(root, it, ["peaks", "genes"])
    assert out is it and not calls and len(ctx.finalizers) == 1
    tag, payload = ctx.finalizers[0](root)
    assert tag == "COMMAND" and payload.startswith(PLAN_PREFIX)
    want = build_plan("SELECT a.name, b.name AS g FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND a.score > 5 "
                      "WHERE b.strand = '+' ORDER BY a.start DESC LIMIT 3", ["peaks", "genes"])
    assert JoinPlan.from_string(payload) == want
    assert want.order_by == (("__giql_o0", True, False),) and want.limit == 3 and len(want.residuals) == 2


@pytest.mark.parametrize("enc", [("1based", "closed"), ("1based", "half_open"), ("0based", "closed"), ("0based", "half_open")])
def test_non_canonical_table_offsets_are_read_off_the_fragments_once(enc):
    # pass 2 has ALREADY wrapped a non-canonical operand -- (a."start" - 1) / (a."end" + 1) -- and blanked its
    # table; the plan must carry that table's encoding exactly once (a second "- 1" from ctx.tables would
    # turn 1-based closed into an encoding that does not exist)
    tables = [Table("peaks", coordinate_system=enc[0], interval_type=enc[1]), "genes"]
    root, it = basic([A.col("a", "start"), A.col("b", "start")])
    out, ctx, calls = run(root, it, tables)
    assert out is it and not calls
    plan = JoinPlan.from_string(ctx.finalizers[0](root)[1])
    assert plan.left.encoding == enc and plan.right.encoding == ("0based", "half_open")
    assert plan == build_plan("SELECT a.start, b.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval", tables)
    res = ctx.resolution.column("this")
    if enc != ("0based", "half_open"):
        assert res.table is None and ("- 1" in res.start or "+ 1" in res.end or "- 1" in res.end)


def test_custom_column_names_come_from_the_resolution():
    tables = [Table("peaks", chrom_col="seqid", start_col="s", end_col="e", coordinate_system="1based", interval_type="closed"),
              Table("genes", chrom_col="seqid")]
    root, it = basic([A.col("a", "s")], swap=True)     # operands written b-first: the FROM side stays the left side
    out, ctx, _ = run(root, it, tables)
    plan = JoinPlan.from_string(ctx.finalizers[0](root)[1])
    assert (plan.left.table, plan.left.alias, plan.left.chrom_col, plan.left.start_col, plan.left.end_col) == ("peaks", "a", "seqid", "s", "e")
    assert plan.left.encoding == ("1based", "closed") and plan.right.chrom_col == "seqid"


def test_operand_without_resolution_falls_back_to_the_registry_by_table_name():
    tables = [Table("peaks", coordinate_system="1based", interval_type="closed"), "genes"]
    root, it = basic([A.col("a", "start")])
    out, ctx, calls = run(root, it, tables, resolution=A.OperatorResolution(columns={}))
    assert out is it and not calls
    assert JoinPlan.from_string(ctx.finalizers[0](root)[1]).left.encoding == ("1based", "closed")


@pytest.mark.parametrize("kind", ["SEMI", "ANTI"])
def test_semi_anti_and_count_overlaps_shapes(kind):
    root, it = basic([A.col("a", "name")], kind=kind)
    out, ctx, _ = run(root, it, ["peaks", "genes"])
    assert JoinPlan.from_string(ctx.finalizers[0](root)[1]).kind == kind
    root, it = basic([A.col("a", "name"), A.alias(A.agg("count", A.col("b", "name")), "n")], side="LEFT", group=[A.col("a", "name")])
    out, ctx, _ = run(root, it, ["peaks", "genes"])
    plan = JoinPlan.from_string(ctx.finalizers[0](root)[1])
    assert plan.kind == "COUNT" and [p.side for p in plan.projection] == ["l", "count"]


def test_grouped_aggregates_match_the_mirror():
    root, it = basic([A.alias(A.col("a", "chrom"), "c"), A.alias(A.agg("count"), "n"), A.alias(A.agg("sum", A.col("b", "score")), "s"),
                      A.alias(A.agg("count", A.col("b", "name"), distinct=True), "d")], group=[A.col("a", "chrom")],
                     order=[(A.col(None, "n"), True)])
    out, ctx, _ = run(root, it, ["peaks", "genes"])
    want = build_plan("SELECT a.chrom AS c, COUNT(*) AS n, SUM(b.score) AS s, COUNT(DISTINCT b.name) AS d FROM peaks a JOIN genes b "
                      "ON a.interval INTERSECTS b.interval GROUP BY a.chrom ORDER BY n DESC", ["peaks", "genes"])
    assert JoinPlan.from_string(ctx.finalizers[0](root)[1]) == want


def _declined(root, it, tables=("peaks", "genes")):
    out, ctx, calls = run(root, it, list(tables))
    assert out is FALLBACK and len(calls) == 1 and calls[0][0] is it and not ctx.finalizers


def test_declined_shapes_defer_to_the_naive_predicate():
    _declined(*basic([A.star()]))                                              # #202
    _declined(*basic([A.star("a")]))
    _declined(*basic([A.col("a", "start")], side="LEFT"))                      # outer join
    _declined(*basic([A.col("a", "start")], side="RIGHT"))
    _declined(*basic([A.agg("count", A.star("a"))]))                           # COUNT(a.*), #204
    _declined(*basic([A.N("add", this=A.col("a", "start"), expression=A.lit(1))]))   # a.start + 1, #205
    _declined(*basic([A.N("window", this=A.agg("sum", A.col("a", "score")))]))
    _declined(*basic([A.col("a", "start")], on_extra=[A.N("like", this=A.col("a", "name"), expression=A.lit("p%"))]))
    _declined(*basic([A.col("a", "start")], on_extra=[A.N("is", this=A.col("a", "score"), expression=A.N("boolean", this=True))]))
    _declined(*basic([A.col("a", "start")], on_extra=[A.N("in", this=A.col("a", "score"), query=A.N("subquery"))]))
    _declined(*basic([A.col("a", "start")], on_extra=[A.N("in", this=A.col("a", "score"), expressions=[A.col("b", "score")])]))
    _declined(*basic([A.col("a", "start")], on_extra=[A.N("between", this=A.col("a", "score"), low=A.lit(1), high=A.lit(2),
                                                            symmetric=True)]))
    # the INTERSECTS itself under OR / NOT: the reference falls back too (_classify_extras)
    it = A.intersects(A.col("a", "interval"), A.col("b", "interval"))
    _declined(A.select([A.col("a", "start")], A.tbl("peaks", "a"),
                       [A.join(A.tbl("genes", "b"), on=A.N("or", this=it, expression=A.cmp("gt", A.col("a", "score"), A.lit(1))))]), it)
    it = A.intersects(A.col("a", "interval"), A.col("b", "interval"))
    _declined(A.select([A.col("a", "start")], A.tbl("peaks", "a"), [A.join(A.tbl("genes", "b"), on=A.N("not", this=it))]), it)
    _declined(*basic([A.col("a", "start")], distinct=A.N("distinct", on=A.N("tuple", expressions=[A.col("a", "chrom")]))))
    _declined(*basic([A.col("a", "start")], with_=A.N("with", expressions=[])))
    _declined(*basic([A.col("a", "start")], order=[(A.N("subquery", this=A.N("select", expressions=[])), False)]))
    # SEMI with its INTERSECTS in WHERE (#201)
    it = A.intersects(A.col("a", "interval"), A.col("b", "interval"))
    root = A.select([A.col("a", "name")], A.tbl("peaks", "a"), [A.join(A.tbl("genes", "b"), on=A.N("boolean", this=True), kind="SEMI")], where=it)
    _declined(root, it)
    # self-join, three tables, a table function operand
    it = A.intersects(A.col("a", "interval"), A.col("b", "interval"))
    _declined(A.select([A.col("a", "start")], A.tbl("peaks", "a"), [A.join(A.tbl("peaks", "b"), on=it)]), it)
    it = A.intersects(A.col("a", "interval"), A.col("b", "interval"))
    _declined(A.select([A.col("a", "start")], A.tbl("peaks", "a"), [A.join(A.tbl("genes", "b")), A.join(A.tbl("exons", "c"))], where=it), it)
    it = A.intersects(A.col("a", "interval"), A.col("b", "interval"))
    fn = A.N("table", this=A.N("anonymous", this="DISJOIN"), alias=A.N("tablealias", this=A.ident("b")))
    _declined(A.select([A.col("a", "start")], A.tbl("peaks", "a"), [A.join(fn, on=it)]), it)


def test_boolean_extras_lower_to_the_same_plan_as_the_mirror():
    # OR / NOT / parentheses / BETWEEN / IN / IS NULL beside the INTERSECTS: inlined by the reference
    # (_classify_extras, intersects_duckdb.py:889-912), a conjunction of OR-groups here
    a_s, b_s = A.col("a", "score"), A.col("b", "score")
    extras = [
        A.N("paren", this=A.N("or", this=A.cmp("gt", a_s, A.lit(5)), expression=A.cmp("lt", b_s, A.lit(2)))),
        A.N("not", this=A.N("paren", this=A.N("and", this=A.cmp("eq", a_s, A.lit(3)), expression=A.cmp("neq", b_s, A.lit(4))))),
        A.N("between", this=a_s, low=A.lit(1), high=A.lit(9)),
        A.N("not", this=A.N("in", this=A.col("b", "name"), expressions=[A.lit("u"), A.lit("v")])),
    ]
    where = A.N("and", this=A.N("in", this=a_s, expressions=[A.lit(1), A.lit(2)]),
                expression=A.N("not", this=A.N("is", this=A.col("a", "name"), expression=A.N("null"))))
    root, it = basic([A.col("a", "name")], on_extra=extras, where=where)
    out, ctx, calls = run(root, it, ["peaks", "genes"])
    assert out is it and not calls
    want = build_plan("SELECT a.name FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND (a.score > 5 OR b.score < 2) "
                      "AND NOT (a.score = 3 AND b.score <> 4) AND a.score BETWEEN 1 AND 9 AND b.name NOT IN ('u', 'v') "
                      "WHERE a.score IN (1, 2) AND a.name IS NOT NULL", ["peaks", "genes"])
    got = JoinPlan.from_string(ctx.finalizers[0](root)[1])
    assert got == want
    assert [(r.op, r.group) for r in got.residuals] == [(">", 1), ("<", 1), ("!=", 2), ("=", 2), (">=", 0), ("<=", 0),
                                                       ("!=", 0), ("!=", 0), ("=", 3), ("=", 3), ("notnull", 0)]


def test_arithmetic_extras_lower_to_the_same_plan_as_the_mirror():
    # the overlap-fraction recipe (docs/recipes/intersect.rst:144-160): GTE(Paren(Sub(Least, Greatest)), Mul(0.5, Paren(Sub)))
    a_s, a_e, b_s, b_e = A.col("a", "start"), A.col("a", "end"), A.col("b", "start"), A.col("b", "end")
    ov = A.N("paren", this=A.N("sub", this=A.N("least", this=a_e, expressions=[b_e]),
                               expression=A.N("greatest", this=a_s, expressions=[b_s])))
    frac = A.N("mul", this=A.lit(0.5), expression=A.N("paren", this=A.N("sub", this=a_e, expression=a_s)))
    other = A.cmp("lt", A.N("abs", this=A.N("sub", this=A.col("a", "score"), expression=A.col("b", "score"))),
                  A.N("div", this=A.N("neg", this=A.col("b", "score")), expression=A.lit(-2.0)))
    root, it = basic([A.col("a", "name")], on_extra=[A.cmp("gte", ov, frac)], where=other)
    out, ctx, calls = run(root, it, ["peaks", "genes"])
    assert out is it and not calls
    want = build_plan("SELECT a.name FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND "
                      "(LEAST(a.end, b.end) - GREATEST(a.start, b.start)) >= 0.5 * (a.end - a.start) "
                      "WHERE ABS(a.score - b.score) < -b.score / -2.0", ["peaks", "genes"])
    assert JoinPlan.from_string(ctx.finalizers[0](root)[1]) == want
    # arithmetic this target has no evaluator for declines (modulo, functions), typed division flags too
    _declined(*basic([A.col("a", "start")], on_extra=[A.cmp("gt", A.N("mod", this=a_s, expression=A.lit(2)), A.lit(0))]))
    _declined(*basic([A.col("a", "start")], on_extra=[A.cmp("gt", A.N("sqrt", this=a_s), A.lit(2))]))
    _declined(*basic([A.col("a", "start")], on_extra=[A.cmp("gt", A.N("div", this=a_s, expression=A.lit(2), typed=True), A.lit(2))]))


def test_sibling_spatial_predicate_and_literal_ranges_defer():
    it = A.intersects(A.col("a", "interval"), A.col("b", "interval"))
    other = A.N("contains", this=A.col("a", "interval"), expression=A.col("b", "interval"))
    root = A.select([A.col("a", "start")], A.tbl("peaks", "a"), [A.join(A.tbl("genes", "b"), on=A.conj(it, other))])
    _declined(root, it)
    it = A.intersects(A.col("a", "interval"), A.col("b", "interval"))
    stamped = A.cmp("lt", A.col("a", "start"), A.col("b", "end"))
    stamped.meta[plugin.SPATIAL_PREDICATE_META] = True            # a predicate a generic expander already rewrote
    root = A.select([A.col("a", "start")], A.tbl("peaks", "a"), [A.join(A.tbl("genes", "b"), on=A.conj(it, stamped))])
    _declined(root, it)
    lit_it = A.intersects(A.col(None, "interval"), A.lit("chr1:1000-2000"))
    root = A.select([A.star()], A.tbl("peaks"), [], where=lit_it)
    out, ctx, calls = run(root, lit_it, ["peaks"], resolution=A.OperatorResolution())
    assert out is FALLBACK and not ctx.finalizers


def test_user_mistakes_raise_instead_of_deferring():
    with pytest.raises(ValueError, match="qualified"):
        run(*basic([A.col(None, "start")]), ["peaks", "genes"])
    with pytest.raises(ValueError, match="Unknown table"):
        run(*basic([A.col("c", "start")]), ["peaks", "genes"])
    with pytest.raises(ValueError, match="left-side"):
        run(*basic([A.col("b", "start")], kind="ANTI"), ["peaks", "genes"])


def test_fragment_parser():
    assert plugin._parse_fragment('a."start"') == ("a", "start", 0)
    assert plugin._parse_fragment('(a."start" - 1)') == ("a", "start", -1)
    assert plugin._parse_fragment('("My A"."end pos" + 1)') == ("My A", "end pos", 1)
    assert plugin._parse_fragment('A.chrom') == ("a", "chrom", 0)
    with pytest.raises(ValueError):
        plugin._parse_fragment('COALESCE(a."start", 0)')


def test_having_and_null_placement_lower_like_the_mirror():
    # HAVING COUNT(*) > 1 AND SUM(b.score) >= 2.5 (the second aggregate is not in the SELECT list), ORDER BY with the
    # NULL placement sqlglot's parser leaves on every Ordered node (giql's dialect: NULLs are small)
    root, it = basic([A.alias(A.col("a", "chrom"), "c"), A.alias(A.agg("count"), "n")], group=[A.col("a", "chrom")],
                     having=A.conj(A.cmp("gt", A.agg("count"), A.lit(1)), A.cmp("gte", A.agg("sum", A.col("b", "score")), A.lit(2.5))),
                     order=[(A.col(None, "n"), True, True), (A.col("a", "chrom"), False, True)])
    out, ctx, calls = run(root, it, ["peaks", "genes"])
    assert out is it and not calls
    _tag, payload = ctx.finalizers[0](root)
    want = build_plan("SELECT a.chrom AS c, COUNT(*) AS n FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval "
                      "GROUP BY a.chrom HAVING COUNT(*) > 1 AND SUM(b.score) >= 2.5 ORDER BY n DESC NULLS FIRST, a.chrom",
                      ["peaks", "genes"])
    got = JoinPlan.from_string(payload)
    assert got == want
    assert [(h.lhs.value, h.op, h.rhs.value) for h in got.having] == [("n", ">", 1), ("__giql_h0", ">=", 2.5)]
    assert got.order_by == (("n", True, True), ("c", False, True))
    assert [a.name for a in got.aggregates] == ["n", "__giql_h0"] and got.output == ("c", "n")


def test_having_shapes_this_target_does_not_run_decline():
    sub = A.N("subquery", this=A.N("select"))
    _declined(*basic([A.col("a", "chrom")], group=[A.col("a", "chrom")], having=A.cmp("gt", A.agg("sum", A.col("a", "score")), sub)))
    _declined(*basic([A.col("a", "chrom")], group=[A.col("a", "chrom")],
                     having=A.cmp("gt", A.N("add", this=A.agg("count"), expression=A.lit(1)), A.lit(1))))


def test_boolean_having_lowers_to_the_same_plan_as_the_mirror():
    having = A.N("and", this=A.N("paren", this=A.N("or", this=A.cmp("gt", A.agg("count"), A.lit(1)),
                                                    expression=A.N("not", this=A.cmp("gte", A.agg("sum", A.col("b", "score")), A.lit(3))))),
                 expression=A.N("not", this=A.N("is", this=A.agg("max", A.col("a", "score")), expression=A.N("null"))))
    root, it = basic([A.col("a", "chrom"), A.alias(A.agg("count"), "n")], group=[A.col("a", "chrom")], having=having)
    out, ctx, calls = run(root, it, ["peaks", "genes"])
    assert out is it and not calls
    want = build_plan("SELECT a.chrom, COUNT(*) AS n FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval GROUP BY a.chrom "
                      "HAVING (COUNT(*) > 1 OR NOT SUM(b.score) >= 3) AND MAX(a.score) IS NOT NULL", ["peaks", "genes"])
    got = JoinPlan.from_string(ctx.finalizers[0](root)[1])
    assert got == want and [(h.op, h.group) for h in got.having] == [(">", 1), ("<", 1), ("notnull", 0)]


def test_clauses_the_lowering_does_not_read_decline_instead_of_being_dropped():
    """the lowering reads the statement by ``args`` key; a clause sqlglot stores under a
    key it does not know must DECLINE (the naive predicate then runs the query) -- never be dropped with a
    plan still emitted.  One case per node kind the whitelist guards."""
    cols = [A.col("a", "start")]

    def with_arg(node, **extra):
        node.args.update(extra)
        return node

    # Select: QUALIFY, named WINDOWs, LATERAL VIEW, PIVOT, TABLESAMPLE, SELECT ... INTO, FOR UPDATE, hints
    for key, val in [("qualify", A.N("qualify", this=A.cmp("gt", A.col("a", "score"), A.lit(1)))),
                     ("windows", [A.N("window", this=A.ident("w"))]), ("laterals", [A.N("lateral")]),
                     ("pivots", [A.N("pivot")]), ("sample", A.N("tablesample", percent=A.lit(10))),
                     ("into", A.N("into", this=A.tbl("t"))), ("locks", [A.N("lock", update=True)]),
                     ("hint", A.N("hint", expressions=[A.ident("x")])), ("connect", A.N("connect")),
                     ("distribute", A.N("distribute")), ("sort", A.N("sort")), ("cluster", A.N("cluster"))]:
        root, it = basic(cols)
        with_arg(root, **{key: val})
        _declined(root, it)
    # GROUP BY ROLLUP / CUBE / GROUPING SETS / ALL / WITH TOTALS
    for key, val in [("rollup", [A.col("a", "chrom")]), ("cube", [A.col("a", "chrom")]),
                     ("grouping_sets", [A.N("tuple", expressions=[A.col("a", "chrom")])]), ("all", True), ("totals", True)]:
        root, it = basic([A.col("a", "chrom"), A.agg("count")], group=[A.col("a", "chrom")])
        with_arg(root.args["group"], **{key: val})
        _declined(root, it)
    # Table operand: TABLESAMPLE / PIVOT / hints / time travel; an alias with a column list
    for key, val in [("sample", A.N("tablesample", percent=A.lit(10))), ("pivots", [A.N("pivot")]),
                     ("hints", [A.N("withtablehint")]), ("version", A.N("version")), ("when", A.N("historicaldata")),
                     ("only", True), ("ordinality", True)]:
        for which in ("from", "join"):
            root, it = basic(cols)
            tnode = root.args["from_"].args["this"] if which == "from" else root.args["joins"][0].args["this"]
            with_arg(tnode, **{key: val})
            _declined(root, it)
    root, it = basic(cols)
    with_arg(root.args["joins"][0].args["this"].args["alias"], columns=[A.ident("c1")])
    _declined(root, it)
    # Join: hints, ASOF match_condition, GLOBAL
    for key, val in [("hint", "BROADCAST"), ("match_condition", A.cmp("gt", A.col("a", "start"), A.col("b", "start"))),
                     ("global", True), ("global_", True)]:
        root, it = basic(cols)
        with_arg(root.args["joins"][0], **{key: val})
        _declined(root, it)
    # LIMIT a, b (offset inside Limit) / LIMIT ... BY / WITH TIES; OFFSET with extra expressions; ORDER ... WITH FILL
    for key, val in [("offset", A.lit(2)), ("expressions", [A.col("a", "chrom")]), ("limit_options", A.N("limitoptions"))]:
        root, it = basic(cols, limit=5)
        with_arg(root.args["limit"], **{key: val})
        _declined(root, it)
    root, it = basic(cols, offset=5)
    with_arg(root.args["offset"], expressions=[A.col("a", "chrom")])
    _declined(root, it)
    root, it = basic(cols, order=[(A.col("a", "start"), False)])
    with_arg(root.args["order"].args["expressions"][0], with_fill=A.N("withfill"))
    _declined(root, it)
    root, it = basic(cols, order=[(A.col("a", "start"), False)])
    with_arg(root.args["order"], siblings=True)
    _declined(root, it)
    # empty / falsy values of unknown args are what sqlglot's parser leaves everywhere: still accepted
    root, it = basic(cols, limit=5)
    with_arg(root, qualify=None, windows=[], laterals=None, sample=None, locks=[], hint=None, kind=None, match=None)
    with_arg(root.args["joins"][0], hint=None, match_condition=None)
    with_arg(root.args["limit"], offset=None, expressions=[], limit_options=None, this=None)
    out, ctx, calls = run(root, it, ["peaks", "genes"])
    assert out is it and not calls and len(ctx.finalizers) == 1


def test_user_mistakes_raise_as_upstream_and_are_not_swallowed_as_declines():
    """Which is which (plugin docstring, "Errors"): a catalog / schema-qualified column is a user mistake the
    reference rejects too (intersects_duckdb.py:803-804, 951-958): ValueError, through the expander."""
    q = A.col("a", "score")
    q.args["db"] = A.ident("myschema")
    root, it = basic([A.col("a", "start")], on_extra=[A.cmp("gt", q, A.lit(5))])
    with pytest.raises(ValueError, match="catalog / schema"):
        run(root, it, ["peaks", "genes"])
    from giql_amd.shape import HipDeclined
    try:
        run(root, it, ["peaks", "genes"])
    except ValueError as exc:
        assert not isinstance(exc, HipDeclined)
