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

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "boolean_residuals.json")))["cases"]
COLS = ["chrom", "start", "end", "name", "score", "strand"]
IDS = [f"{i}:{c['kind']}" for i, c in enumerate(GOLDEN)]


def _key(r):
    return tuple((x is None, x) for x in r)


def _value(o, p, g):
    if o.kind in ("l", "r"):
        return (p if o.kind == "l" else g)[COLS.index(o.value)]
    return o.value


def _leaf(res, p, g) -> bool:
    a = _value(res.lhs, p, g)
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


HAVING = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "boolean_residuals.json")))["having"]


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
