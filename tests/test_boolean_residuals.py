"""OR / NOT / parentheses / BETWEEN / IN / IS NULL beside the INTERSECTS.

The reference inlines such extras as SQL text (`_classify_extras`, src/giql/expanders/intersects_duckdb.py:889-912,
1239-1243); the hip target lowers them to a conjunction of OR-groups of comparisons (giql_pred.group).  The
expected rows are sqlite's (tests/golden/make_boolean_residuals.py).  CPU: the plan's normal form, evaluated in
plain Python over the brute-force overlap pairs, must give sqlite's rows (three-valued logic included).  GPU:
transpile + execute must."""
import json
import os

import pytest

from giql_amd.transpile import build_plan, transpile

_DOC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "boolean_residuals.json")))
GOLDEN = _DOC["cases"] + _DOC["arith"]    # boolean combinations; arithmetic operands (the overlap-fraction recipes)
COLS = ["chrom", "start", "end", "name", "score", "strand"]
IDS = [f"{i}:{c['kind']}" for i, c in enumerate(_DOC["cases"])] + [f"arith{i}:{c['kind']}" for i, c in enumerate(_DOC["arith"])]


def _key(r):
    return tuple((x is None, x) for x in r)


def _value(o, p, g):
    if o.kind in ("l", "r"):
        return (p if o.kind == "l" else g)[COLS.index(o.value)]
    if o.kind == "expr":
        return _arith(o.value, p, g)
    return o.value


def _arith(t, p, g):
    """An expression tree of the plan in plain Python, with the reference's execution target's semantics: NULL
    propagates, `/` is a floating division and NULL on a zero divisor, LEAST / GREATEST skip NULLs."""
    if t[0] != "fn":
        return (p if t[0] == "l" else g)[COLS.index(t[1])] if t[0] in ("l", "r") else t[1]
    args = [_arith(c, p, g) for c in t[2]]
    # boolean nodes (round 4: a nested condition travels as one program): three-valued, True / False / None
    if t[1] in ("isnull", "notnull"):
        return (args[0] is None) == (t[1] == "isnull")
    if t[1] == "not":
        return None if args[0] is None else not args[0]
    if t[1] == "and":
        return False if any(a is False for a in args) else (None if any(a is None for a in args) else True)
    if t[1] == "or":
        return True if any(a is True for a in args) else (None if any(a is None for a in args) else False)
    if t[1] in ("=", "!=", "<", "<=", ">", ">="):
        a, b = args
        if a is None or b is None:
            return None
        return {"=": a == b, "!=": a != b, "<": a < b, "<=": a <= b, ">": a > b, ">=": a >= b}[t[1]]
    if t[1] in ("least", "greatest"):
        vals = [a for a in args if a is not None]
        return None if not vals else (min(vals) if t[1] == "least" else max(vals))
    if any(a is None for a in args):
        return None
    if t[1] == "neg":
        return -args[0]
    if t[1] == "abs":
        return abs(args[0])
    a, b = args
    if t[1] == "/":
        return None if b == 0 else a / b
    return {"+": a + b, "-": a - b, "*": a * b}[t[1]]


def _leaf(res, p, g) -> bool:
    a = _value(res.lhs, p, g)
    if res.op == "istrue":
        return a is True
    if res.op in ("isnull", "notnull"):
        return (a is None) == (res.op == "isnull")
    b = _value(res.rhs, p, g)
    if a is None or b is None:
        return False
    return {"=": a == b, "!=": a != b, "<": a < b, "<=": a <= b, ">": a > b, ">=": a >= b}[res.op]


def _holds(residuals, p, g) -> bool:
    """An AND of clauses; neighbours sharing a non-zero group are one OR clause (as the select kernel reads them)."""
    clauses = []
    for r in residuals:
        if clauses and r.group and clauses[-1][-1].group == r.group:
            clauses[-1].append(r)
        else:
            clauses.append([r])
    return all(any(_leaf(r, p, g) for r in c) for c in clauses)


def _overlap(p, g):
    return p[0] == g[0] and p[1] < g[2] and p[2] > g[1]


@pytest.mark.parametrize("case", GOLDEN, ids=IDS)
def test_the_plans_normal_form_gives_sqlites_rows(case):
    plan = build_plan(case["query"], ["peaks", "genes"])
    assert plan.kind == case["kind"]
    peaks, genes = case["peaks"], case["genes"]
    if plan.kind == "INNER":
        got = [[p[3], p[1], g[3], g[2]] for p in peaks for g in genes if _overlap(p, g) and _holds(plan.residuals, p, g)]
    else:
        on = [r for r in plan.residuals if r.clause == "on"]
        where = [r for r in plan.residuals if r.clause == "where"]
        got = [[p[3], p[1], p[4]] for p in peaks
               if _holds(where, p, None) and (any(_overlap(p, g) and _holds(on, p, g) for g in genes) != (plan.kind == "ANTI"))]
    assert sorted(got, key=_key) == case["rows"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLDEN, ids=IDS)
def test_execute_gives_sqlites_rows(case):
    import pyarrow as pa

    from giql_amd.execute import execute

    def table(rows):
        cols = list(zip(*rows)) if rows else [[]] * 6
        types = [pa.string(), pa.int32(), pa.int32(), pa.string(), pa.int32(), pa.string()]
        return pa.table({c: pa.array(list(v), t) for c, v, t in zip(COLS, cols, types)})

    t = {"peaks": table(case["peaks"]), "genes": table(case["genes"])}
    out = execute(transpile(case["query"], tables=["peaks", "genes"], dialect="hip"), t)
    got = sorted(([*d.values()] for d in out.to_pylist()), key=_key)
    assert got == case["rows"]


HAVING = _DOC["having"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", HAVING, ids=[f"having{i}" for i in range(len(HAVING))])
def test_boolean_having_gives_sqlites_groups(case):
    # HAVING rides on the reference's outer wrapper verbatim (intersects_duckdb.py:1336-1400); here the grouped
    # result is filtered by the condition's normal form (Kleene AND / OR over the comparisons' masks)
    import pyarrow as pa

    from giql_amd.execute import execute

    def table(rows):
        types = [pa.string(), pa.int32(), pa.int32(), pa.string(), pa.int32(), pa.string()]
        return pa.table({c: pa.array(list(v), t) for c, v, t in zip(COLS, zip(*rows), types)})

    t = {"peaks": table(case["peaks"]), "genes": table(case["genes"])}
    out = execute(transpile(case["query"], tables=["peaks", "genes"], dialect="hip"), t)
    got = sorted(([*d.values()] for d in out.to_pylist()), key=_key)
    assert got == case["rows"]


RECIPES = [
    # docs/recipes/intersect.rst:120-190 of the reference, as written there (comma join, conditions in WHERE)
    ("same strand", "AND a.strand = b.strand", lambda p, g: p[5] == g[5]),
    ("opposite strands", "AND a.strand != b.strand AND a.strand IN ('+', '-') AND b.strand IN ('+', '-')",
     lambda p, g: p[5] != g[5] and p[5] in "+-" and g[5] in "+-"),
    ("half of A covered", "AND ( LEAST(a.end, b.end) - GREATEST(a.start, b.start) ) >= 0.5 * (a.end - a.start)",
     lambda p, g: min(p[2], g[2]) - max(p[1], g[1]) >= 0.5 * (p[2] - p[1])),
    ("half of B covered", "AND ( LEAST(a.end, b.end) - GREATEST(a.start, b.start) ) >= 0.5 * (b.end - b.start)",
     lambda p, g: min(p[2], g[2]) - max(p[1], g[1]) >= 0.5 * (g[2] - g[1])),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,extra,pred", RECIPES, ids=[r[0] for r in RECIPES])
def test_the_documented_recipes_run_on_the_gpu_path(name, extra, pred):
    import numpy as np
    import pyarrow as pa

    from giql_amd.execute import execute

    rng = np.random.default_rng(9)

    def rows(n, tag):
        out = []
        for i in range(n):
            s = int(rng.integers(0, 4000))
            out.append(("chr%d" % rng.integers(1, 3), s, s + int(rng.integers(1, 500)), f"{tag}{i}", int(rng.integers(0, 9)),
                        str(rng.choice(["+", "-", "."]))))
        return out

    peaks, genes = rows(400, "p"), rows(350, "g")
    types = [pa.string(), pa.int32(), pa.int32(), pa.string(), pa.int32(), pa.string()]
    t = {n: pa.table({c: pa.array(list(v), ty) for c, v, ty in zip(COLS, zip(*r), types)})
         for n, r in (("features_a", peaks), ("features_b", genes))}
    q = f"SELECT a.name, b.name AS b_name FROM features_a a, features_b b WHERE a.interval INTERSECTS b.interval {extra}"
    out = execute(transpile(q, tables=["features_a", "features_b"], dialect="hip"), t)
    got = sorted(zip(out.column("name").to_pylist(), out.column("b_name").to_pylist()))
    want = sorted((p[3], g[3]) for p in peaks for g in genes if _overlap(p, g) and pred(p, g))
    assert got == want and len(want) > 20


@pytest.mark.gpu
@pytest.mark.parametrize("case", [GOLDEN[i] for i in (1, 9, 19, 27, 33, 39, 41, 55)], ids=lambda c: c["kind"])
def test_boolean_and_arithmetic_residuals_through_several_contexts(case):
    # execute(devices=[0, 0]): chromosomes sharded over two contexts, every shard filters its own pairs
    import pyarrow as pa

    from giql_amd.execute import execute

    def table(rows):
        types = [pa.string(), pa.int32(), pa.int32(), pa.string(), pa.int32(), pa.string()]
        return pa.table({c: pa.array(list(v), t) for c, v, t in zip(COLS, zip(*rows), types)})

    t = {"peaks": table(case["peaks"]), "genes": table(case["genes"])}
    out = execute(transpile(case["query"], tables=["peaks", "genes"], dialect="hip"), t, devices=[0, 0])
    assert sorted(([*d.values()] for d in out.to_pylist()), key=_key) == case["rows"]


def _random_condition(rng, depth):
    """A random condition over a.score / b.score / literals as (SQL text, evaluator under Kleene logic)."""
    import operator

    ops = {"=": operator.eq, "<>": operator.ne, "<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge}
    if depth == 0 or rng.random() < 0.25:
        kind = rng.random()
        col = rng.choice(["a.score", "b.score"])
        get = (lambda p, g: p[4]) if col == "a.score" else (lambda p, g: g[4])
        if kind < 0.15:
            neg = rng.random() < 0.5
            return f"{col} IS {'NOT ' if neg else ''}NULL", lambda p, g: (get(p, g) is None) != neg
        if kind < 0.3:
            lo, hi, neg = rng.randrange(0, 4), rng.randrange(2, 6), rng.random() < 0.5
            def between(p, g):
                v = get(p, g)
                return None if v is None else ((lo <= v <= hi) != neg)
            return f"{col} {'NOT ' if neg else ''}BETWEEN {lo} AND {hi}", between
        if kind < 0.45:
            vals, neg = sorted({rng.randrange(0, 6) for _ in range(rng.randrange(1, 4))}), rng.random() < 0.5
            def isin(p, g):
                v = get(p, g)
                return None if v is None else ((v in vals) != neg)
            return f"{col} {'NOT ' if neg else ''}IN ({', '.join(map(str, vals))})", isin
        op = rng.choice(sorted(ops))
        if rng.random() < 0.4:
            def cmp2(p, g):
                return None if p[4] is None or g[4] is None else ops[op](p[4], g[4])
            return f"a.score {op} b.score", cmp2
        lit = rng.randrange(0, 6)
        def cmp1(p, g):
            v = get(p, g)
            return None if v is None else ops[op](v, lit)
        return f"{col} {op} {lit}", cmp1
    kind = rng.random()
    if kind < 0.2:
        text, f = _random_condition(rng, depth - 1)
        return f"NOT ({text})", lambda p, g: (None if f(p, g) is None else not f(p, g))
    kids = [_random_condition(rng, depth - 1) for _ in range(rng.randrange(2, 4))]
    if kind < 0.6:
        def conj(p, g):
            vs = [f(p, g) for _t, f in kids]
            return False if any(v is False for v in vs) else (None if any(v is None for v in vs) else True)
        return "(" + " AND ".join(t for t, _f in kids) + ")", conj
    def disj(p, g):
        vs = [f(p, g) for _t, f in kids]
        return True if any(v is True for v in vs) else (None if any(v is None for v in vs) else False)
    return "(" + " OR ".join(t for t, _f in kids) + ")", disj


@pytest.mark.parametrize("seed", range(60))
def test_random_conditions_normalise_to_what_three_valued_logic_gives(seed):
    # parser + NOT push-down + distribution against a direct Kleene evaluation of the same tree: a filter keeps TRUE only
    import random

    from giql_amd.shape import HipDeclined

    rng = random.Random(seed)
    text, truth = _random_condition(rng, 3)
    q = f"SELECT a.start FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval AND {text}"
    plan = build_plan(q, ["peaks", "genes"])   # (round 4: never declined -- what outgrows the normal form is a program)
    assert HipDeclined is not None
    rows = [("c", 0, 1, "n", v, "+") for v in (None, 0, 1, 2, 3, 4, 5)]
    for p in rows:
        for g in rows:
            assert _holds(plan.residuals, p, g) == (truth(p, g) is True), (text, p[4], g[4])


# ---- what one select call holds: the gate declines, never a run-time error -------------------------
def _nested_sum(depth_levels: int) -> str:
    """1 + (2 + (... + (k + a.score))): a right-nested sum keeps one more value live per level."""
    expr = "a.score"
    for k in range(depth_levels, 0, -1):
        expr = f"{k} + ({expr})"
    return expr


def test_expression_depth_and_size_limits_are_declined_at_plan_time():
    from giql_amd.shape import MAX_EXPR_DEPTH, MAX_EXPR_NODES, HipDeclined, expression_cost

    base = "SELECT a.name FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval WHERE "
    # depth: k nested levels keep k + 1 values live
    ok = build_plan(base + _nested_sum(MAX_EXPR_DEPTH - 1) + " > 3", ["peaks", "genes"])
    nodes, depth = expression_cost(ok.residuals[0].lhs)
    assert depth == MAX_EXPR_DEPTH and nodes == 2 * (MAX_EXPR_DEPTH - 1) + 1
    with pytest.raises(HipDeclined, match="too deep"):
        build_plan(base + _nested_sum(MAX_EXPR_DEPTH) + " > 3", ["peaks", "genes"])
    # a LEFT-nested sum of any length keeps two values live: only the node count bounds it
    half = MAX_EXPR_NODES // 2
    flat = " + ".join(["a.score"] * half)                      # n leaves + (n - 1) operators = 2 n - 1 nodes
    p = build_plan(base + flat + " > b.score", ["peaks", "genes"])
    assert expression_cost(p.residuals[0].lhs) == (MAX_EXPR_NODES - 1, 2)
    with pytest.raises(HipDeclined, match="too large"):
        build_plan(base + " + ".join(["a.score"] * (half + 1)) + " > b.score", ["peaks", "genes"])   # one node too many
    # the nodes of ONE select call add up over its comparisons (here: all two-sided, AND-ed) ...
    recipe = "(LEAST(a.end, b.end) - GREATEST(a.start, b.start)) >= 0.5 * (b.end - b.start)"   # 12 nodes
    one = build_plan(base + recipe, ["peaks", "genes"]).residuals[0]
    assert expression_cost(one.lhs)[0] + expression_cost(one.rhs)[0] == 12
    variants = [recipe.replace("0.5", f"0.{k:02d}") for k in range(1, 40)]
    build_plan(base + " AND ".join(variants[:21]), ["peaks", "genes"])                        # 252 nodes
    with pytest.raises(HipDeclined, match="too large"):
        build_plan(base + " AND ".join(variants[:22]), ["peaks", "genes"])                    # 264 nodes in one call
    # an OR of many recipes outgrows the normal form and travels as ONE boolean program: 13 nodes per comparison
    # + the ORs between them
    prog = build_plan(base + " OR ".join(variants[:15]), ["peaks", "genes"]).residuals
    assert len(prog) == 1 and prog[0].op == "istrue" and expression_cost(prog[0].lhs)[0] == 15 * 13 + 14
    with pytest.raises(HipDeclined, match="too large"):
        build_plan(base + " OR ".join(variants[:20]), ["peaks", "genes"])                     # 279 nodes
    # ... but one-sided conditions run in calls of their own
    left = " + ".join(["a.score"] * (half - 2)) + " > 0"
    right = " + ".join(["b.score"] * (half - 2)) + " > 0"
    assert len(build_plan(base + left + " AND " + right, ["peaks", "genes"]).residuals) == 2
    assert (MAX_EXPR_DEPTH, MAX_EXPR_NODES) == (12, 256)


# ---- conditions past the normal form's cap: boolean programs (round 4) -------------------------
NESTED = _DOC.get("nested", [])


@pytest.mark.parametrize("case", NESTED, ids=[f"nested{i}:{c['kind']}" for i, c in enumerate(NESTED)])
def test_nested_conditions_as_programs_give_sqlites_rows_on_the_cpu(case):
    plan = build_plan(case["query"], ["peaks", "genes"])
    peaks, genes = case["peaks"], case["genes"]
    if plan.kind == "INNER":
        got = [[p[3], p[1], g[3], g[2]] for p in peaks for g in genes if _overlap(p, g) and _holds(plan.residuals, p, g)]
    else:
        on = [r for r in plan.residuals if r.clause == "on"]
        where = [r for r in plan.residuals if r.clause == "where"]
        got = [[p[3], p[1], p[4]] for p in peaks
               if _holds(where, p, None) and (any(_overlap(p, g) and _holds(on, p, g) for g in genes) != (plan.kind == "ANTI"))]
    assert sorted(got, key=_key) == case["rows"]


def test_the_nested_golden_cases_are_the_ones_the_normal_form_could_not_take():
    # a good third of them carry a boolean program (op "istrue": the normal form would have been declined in round 3,
    # the others still fit it); the plan's string form round-trips
    from giql_amd.plan import JoinPlan

    n_prog = 0
    for case in NESTED:
        plan = build_plan(case["query"], ["peaks", "genes"])
        n_prog += any(r.op == "istrue" for r in plan.residuals)
        assert JoinPlan.from_string(plan.to_string()) == plan
    assert len(NESTED) >= 60 and n_prog >= 24


@pytest.mark.gpu
@pytest.mark.parametrize("case", NESTED, ids=[f"nested{i}:{c['kind']}" for i, c in enumerate(NESTED)])
def test_nested_conditions_run_on_the_gpu_path(case):
    import pyarrow as pa

    from giql_amd.execute import execute

    def table(rows):
        cols = list(zip(*rows)) if rows else [[]] * 6
        types = [pa.string(), pa.int32(), pa.int32(), pa.string(), pa.int32(), pa.string()]
        return pa.table({c: pa.array(list(v), t) for c, v, t in zip(COLS, cols, types)})

    t = {"peaks": table(case["peaks"]), "genes": table(case["genes"])}
    out = execute(transpile(case["query"], tables=["peaks", "genes"], dialect="hip"), t)
    got = sorted(([*d.values()] for d in out.to_pylist()), key=_key)
    assert got == case["rows"]
