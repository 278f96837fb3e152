#!/usr/bin/env python3
"""Mint tests/golden/cluster_predicate.json: CLUSTER(interval[, d][, stranded := b], predicate := ...) cases whose
cluster ids come from sqlite3 executing the window SQL the reference emits for them -- the adjacency CASE of
src/giql/expanders/cluster.py:210-300 with the predicate ANDed in and every ``PREV(col)`` rewritten to
``LAG("col") OVER (PARTITION BY "chrom"[, "strand"] ORDER BY "start" NULLS LAST)`` (cluster.py:281-296, 587-640;
the fragments are pinned by tests/test_cluster_predicate_transpilation.py:33-36, 51-58, 76-81, 100-103, 122-125).
Needs only the standard library (no reference code is imported).  Starts are distinct inside a partition: the
predecessor of a row among equal starts is engine-dependent upstream too."""
import json
import os
import random
import sqlite3

HERE = os.path.dirname(os.path.abspath(__file__))


def cluster_pred_sql(distance: int, stranded: bool, pred) -> str:
    part = 'PARTITION BY "chrom"' + (', "strand"' if stranded else "")
    window = f'OVER ({part} ORDER BY "start" NULLS LAST ROWS BETWEEN UNBOUNDED PRECEDING AND 1 PRECEDING)'
    edge = f'MAX("end") {window}' + (f" + {distance}" if distance > 0 else "")
    lag = f'OVER ({part} ORDER BY "start" NULLS LAST)'

    def operand(o):
        kind, v = o
        if kind == "col":
            return f'"{v}"'
        if kind == "prev":
            return f'LAG("{v}") {lag}'
        return repr(v) if not isinstance(v, str) else "'" + v + "'"

    text = render(pred, operand)
    inner = (f'SELECT *, CASE WHEN {edge} >= "start" AND ({text}) THEN 0 ELSE 1 END AS __giql_is_new_cluster '
             "FROM features")
    return (f'SELECT *, SUM(__giql_is_new_cluster) OVER ({part} ORDER BY "start" NULLS LAST) '
            f"AS __giql_cluster_id FROM ({inner}) AS __giql_lag_calc")


def run(rows, distance, stranded, pred):
    conn = sqlite3.connect(":memory:")
    conn.execute('CREATE TABLE features (rid INTEGER, chrom TEXT, "start" INTEGER, "end" INTEGER, strand TEXT, '
                 "depth INTEGER, name TEXT, score REAL)")
    conn.executemany("INSERT INTO features VALUES (?, ?, ?, ?, ?, ?, ?, ?)", [(i, *r) for i, r in enumerate(rows)])
    ids = conn.execute(f"SELECT rid, __giql_cluster_id FROM ({cluster_pred_sql(distance, stranded, pred)}) ORDER BY rid").fetchall()
    # MERGE(..., predicate := ...) is a GROUP BY over that clustered relation (merge.py:201-210, 253-330)
    keys = '"chrom"' + (', "strand"' if stranded else "")
    merged = conn.execute(f'SELECT {keys}, MIN("start"), MAX("end"), COUNT(*) FROM ({cluster_pred_sql(distance, stranded, pred)}) '
                          f"GROUP BY {keys}, __giql_cluster_id ORDER BY {keys}, MIN(\"start\")").fetchall()
    conn.close()
    return [r[1] for r in ids], [list(r) for r in merged]


def render(pred, operand) -> str:
    """A predicate as text: a list of (lhs, op, rhs) comparisons is their conjunction; a tuple is a tree node --
    ("and" | "or", [nodes]), ("not", node), ("cmp", lhs, op, rhs), ("null" | "notnull", operand),
    ("between" | "notbetween", x, lo, hi), ("in" | "notin", x, [literals])."""
    if isinstance(pred, list):
        return " AND ".join(f"{operand(l)} {op} {operand(r)}" for l, op, r in pred)
    k = pred[0]
    if k in ("and", "or"):
        return "(" + f" {k.upper()} ".join(render(c, operand) for c in pred[1]) + ")"
    if k == "not":
        return "NOT " + render(pred[1], operand)
    if k == "cmp":
        return f"{operand(pred[1])} {pred[2]} {operand(pred[3])}"
    if k in ("null", "notnull"):
        return f"{operand(pred[1])} IS {'NOT ' if k == 'notnull' else ''}NULL"
    if k in ("between", "notbetween"):
        return f"{operand(pred[1])} {'NOT ' if k == 'notbetween' else ''}BETWEEN {operand(pred[2])} AND {operand(pred[3])}"
    assert k in ("in", "notin"), k
    return f"{operand(pred[1])} {'NOT ' if k == 'notin' else ''}IN ({', '.join(operand(('lit', v)) for v in pred[2])})"


def giql_text(pred) -> str:
    def operand(o):
        kind, v = o
        return v if kind == "col" else (f"PREV({v})" if kind == "prev" else (repr(v) if not isinstance(v, str) else f"'{v}'"))
    return render(pred, operand)


PREDICATES = [
    [(("col", "depth"), "=", ("prev", "depth"))],                               # the docs' run-length example
    [(("col", "name"), "=", ("prev", "name"))],                                 # a string column
    [(("col", "end"), "=", ("prev", "start"))],                                 # reserved-word genomic columns (never true here)
    [(("col", "depth"), ">=", ("prev", "depth")), (("col", "score"), "<", ("lit", 0.75))],
    [(("prev", "depth"), "!=", ("lit", 3))],
    [(("col", "score"), ">", ("prev", "score"))],                               # floats, with NULLs
    # boolean forms: the reference inlines the text as it stands; the hip target normalises it (AND of OR-groups)
    ("or", [("cmp", ("col", "depth"), "=", ("prev", "depth")), ("cmp", ("col", "name"), "=", ("prev", "name"))]),
    ("not", ("or", [("cmp", ("col", "depth"), "<", ("prev", "depth")), ("cmp", ("col", "score"), ">=", ("lit", 0.5))])),
    ("and", [("or", [("null", ("prev", "score")), ("cmp", ("col", "score"), "<=", ("prev", "score"))]),
             ("notnull", ("col", "name"))]),
    ("or", [("and", [("between", ("col", "depth"), ("lit", 1), ("lit", 2)), ("in", ("prev", "name"), ["x", "y"])]),
            ("notbetween", ("prev", "depth"), ("lit", 1), ("col", "depth"))]),
    ("not", ("and", [("notin", ("col", "depth"), [1, 3]), ("not", ("cmp", ("col", "name"), "!=", ("prev", "name")))])),
]


def main() -> None:
    rng = random.Random(20261005)
    cases = []
    # the documentation's example shape: depth runs over abutting bins (docs/dialect/aggregation-operators.rst:40-75)
    doc = [("chr1", 0, 100, "+", 5, "a", 0.1), ("chr1", 100, 200, "+", 5, "a", 0.2), ("chr1", 200, 300, "+", 7, "b", 0.3),
           ("chr1", 300, 400, "+", 7, "b", None), ("chr1", 400, 500, "+", 5, "a", 0.5), ("chr2", 0, 100, "+", 5, "a", 0.6)]
    ids, merged = run(doc, 0, False, PREDICATES[0])
    assert ids == [1, 1, 2, 2, 3, 1], ids
    assert merged == [["chr1", 0, 200, 2], ["chr1", 200, 400, 2], ["chr1", 400, 500, 1], ["chr2", 0, 100, 1]], merged
    cases.append({"name": "doc_depth_runs", "rows": [list(r) for r in doc], "distance": 0, "stranded": False,
                  "predicate": giql_text(PREDICATES[0]), "ids": ids, "merged": merged})
    idx = 0
    for stranded in (False, True):
        for distance in (0, 25):
            for pred in PREDICATES:
                for n, ms, ml in [(1, 10, 5), (40, 400, 60), (120, 3000, 40)]:
                    rows, used = [], set()
                    while len(rows) < n:
                        c, s = rng.choice(["chr1", "chr2", "chr3"]), rng.randint(0, ms)
                        st = rng.choice("+-")
                        if (c, st if stranded else "", s) in used:
                            continue
                        used.add((c, st if stranded else "", s))
                        rows.append((c, s, s + rng.randint(1, ml), st, rng.choice([1, 2, 3, None]),
                                     rng.choice(["x", "y", None]), rng.choice([None, round(rng.random(), 3)])))
                    ids, merged = run(rows, distance, stranded, pred)
                    cases.append({"name": f"fuzz_pred_{idx}", "rows": [list(r) for r in rows], "distance": distance,
                                  "stranded": stranded, "predicate": giql_text(pred), "ids": ids, "merged": merged})
                    idx += 1
    with open(os.path.join(HERE, "cluster_predicate.json"), "w") as f:
        json.dump({"source": "tests/golden/make_cluster_predicate.py (sqlite3 over the reference's window SQL)",
                   "cases": cases}, f)
    print(f"wrote cluster_predicate.json ({len(cases)} cases)")


if __name__ == "__main__":
    main()
