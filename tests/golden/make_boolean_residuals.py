#!/usr/bin/env python3
"""Mint tests/golden/boolean_residuals.json: INTERSECTS joins whose extra ON / WHERE conditions use OR, NOT,
parentheses, BETWEEN, IN (list) and IS [NOT] NULL.  The reference inlines such an extra verbatim, as SQL text,
into the per-chromosome join's ON clause (`_classify_extras` -> "inline", src/giql/expanders/intersects_duckdb.py:889-912,
1164-1177, 1239-1243), so its rows are the rows of the plain overlap join filtered by that SQL condition -- which
is what sqlite3 computes here from the same condition text (three-valued logic included: `score`, `name` hold NULLs).
SEMI / ANTI: the ON extras take part in the existence test, the WHERE extras filter the left rows (#200,
intersects_duckdb.py:1164-1177, 1254-1282).  Needs only the standard library (no reference code is imported)."""
import json
import os
import random
import sqlite3

HERE = os.path.dirname(os.path.abspath(__file__))
OVERLAP = 'a.chrom = b.chrom AND a."start" < b."end" AND a."end" > b."start"'

# (join kind, extra ON condition or "", WHERE condition or "")
CONDITIONS = [
    ("INNER", "(a.score > 3 OR b.score < 2)", ""),
    ("INNER", "NOT a.score > 3", ""),
    ("INNER", "NOT (a.score > 3 OR b.score <= 2)", ""),
    ("INNER", "", "NOT (a.score = 3 AND NOT b.score <> 2)"),
    ("INNER", "", "(a.score > 1 AND b.score > 2) OR a.score = 0"),
    ("INNER", "a.score BETWEEN 2 AND 4", "b.score NOT BETWEEN 1 AND 3"),
    ("INNER", "a.score IN (1, 2, 5)", "b.name NOT IN ('g1', 'g2', 'g3')"),
    ("INNER", "a.name IS NOT NULL AND b.score IS NULL", ""),
    ("INNER", "", "a.score IS NULL OR b.score IS NULL OR a.score = b.score"),
    ("INNER", "(a.strand = b.strand OR a.name = 'p4') AND (b.score >= 2 OR a.score < b.score)", "NOT a.name IS NULL"),
    ("INNER", "", "a.name = 'p1' OR a.name = 'p2' OR b.name <> a.name"),
    ("INNER", "a.score = 2.0 OR b.score > 3.5", ""),
    ("INNER", "((a.score < 2 OR a.score > 4) AND (b.score < 2 OR b.score > 4)) OR a.strand <> b.strand", ""),
    ("INNER", "NOT (a.score BETWEEN 1 AND 4 AND b.name IN ('g0', 'g5'))", ""),
    ("SEMI", "(a.score > 3 OR b.score < 2)", ""),
    ("SEMI", "b.score IN (0, 5) OR a.strand = b.strand", "a.score IS NOT NULL AND (a.score < 2 OR a.name = 'p3')"),
    ("ANTI", "(a.score > 3 OR b.score < 2)", ""),
    ("ANTI", "NOT (a.strand = b.strand)", "a.score NOT BETWEEN 2 AND 3 OR a.name IS NULL"),
    ("ANTI", "b.score IS NULL", "NOT (a.score = 1 OR a.score = 4)"),
]


# grouped queries with a boolean HAVING (handed to the engine verbatim upstream, intersects_duckdb.py:1336-1400):
# (SELECT list, GROUP BY, HAVING) -- valid in GIQL and, with the overlap predicate spelt out, in sqlite
HAVING = [
    ("a.name, COUNT(*) AS n", "a.name", "COUNT(*) > 8 OR SUM(b.score) < 10"),
    ("a.name, COUNT(*) AS n", "a.name", "NOT (COUNT(*) BETWEEN 3 AND 9) AND a.name IS NOT NULL"),
    ("a.name, COUNT(*) AS n, MAX(b.score) AS m", "a.name",
     "(MIN(b.score) IS NULL OR MAX(b.score) >= 5) AND COUNT(*) IN (1, 2, 3, 10, 11)"),
    ("a.chrom, a.strand, COUNT(*) AS n, SUM(b.score) AS s", "a.chrom, a.strand",
     "SUM(b.score) > 150 OR a.strand = '-' AND NOT COUNT(*) < 20"),
    ("a.score, COUNT(*) AS n", "a.score", "a.score IS NULL OR NOT (a.score IN (1, 3) OR COUNT(*) <= 4)"),
]


# arithmetic operands (the overlap-fraction recipes, docs/recipes/intersect.rst:144-190, and kin).  LEAST / GREATEST /
# ABS are spelt min / max / abs for sqlite; their arguments are never NULL here (sqlite's min(x, NULL) is NULL where
# the reference's DuckDB skips the NULL), `/` always has a floating operand (DuckDB's `/` is a floating division)
ARITH = [
    ("INNER", "", "(LEAST(a.end, b.end) - GREATEST(a.start, b.start)) >= 0.5 * (a.end - a.start)"),
    ("INNER", "(LEAST(a.end, b.end) - GREATEST(a.start, b.start)) >= 0.5 * (b.end - b.start)", ""),
    ("INNER", "", "(LEAST(a.end, b.end) - GREATEST(a.start, b.start)) >= 0.5 * (a.end - a.start) "
                  "AND (LEAST(a.end, b.end) - GREATEST(a.start, b.start)) >= 0.5 * (b.end - b.start)"),
    ("INNER", "ABS(a.score - b.score) <= 1", ""),
    ("INNER", "", "a.score + b.score > 5 OR a.score * 2 < b.score"),
    ("INNER", "-a.score < -2 AND (a.end - a.start) / 2.0 > b.end - b.start", ""),
    ("INNER", "", "GREATEST(a.start, b.start, 700) - LEAST(a.end, b.end, 900) > -150"),
    ("INNER", "(a.end - a.start) / (a.score - 2.0) > 50", ""),
    ("SEMI", "(LEAST(a.end, b.end) - GREATEST(a.start, b.start)) * 2 >= a.end - a.start", "a.end - a.start BETWEEN 20 AND 250"),
    ("ANTI", "ABS(a.start - b.start) < 60", "NOT a.score * a.score > 9"),
    # IS [NOT] NULL over an expression (accepted by the gate, crashed in execute())
    ("INNER", "", "(a.score + 1) IS NULL OR (b.score - a.score) IS NOT NULL AND a.score > 2"),
    ("INNER", "(a.score * b.score) IS NOT NULL", "(a.end - a.start) IS NOT NULL"),
    ("ANTI", "(a.score + b.score) IS NULL", ""),
]


def random_condition(rng, depth):
    """A random nested condition over scores, names and strands (SQL text valid in GIQL and in sqlite)."""
    if depth == 0 or rng.random() < 0.2:
        kind = rng.random()
        col = rng.choice(["a.score", "b.score"])
        if kind < 0.12:
            return f"{rng.choice(['a.score', 'b.score', 'a.name', 'b.name'])} IS {'NOT ' if rng.random() < 0.5 else ''}NULL"
        if kind < 0.24:
            return f"{col} {'NOT ' if rng.random() < 0.5 else ''}BETWEEN {rng.randrange(0, 4)} AND {rng.randrange(2, 6)}"
        if kind < 0.36:
            vals = sorted({rng.randrange(0, 6) for _ in range(rng.randrange(1, 4))})
            return f"{col} {'NOT ' if rng.random() < 0.5 else ''}IN ({', '.join(map(str, vals))})"
        if kind < 0.48:
            return f"a.strand {rng.choice(['=', '<>'])} b.strand"
        if kind < 0.58:
            return f"{rng.choice(['a.name', 'b.name'])} {rng.choice(['=', '<>', '<', '>='])} '{rng.choice('pg')}{rng.randrange(0, 7)}'"
        if kind < 0.66:
            return f"a.score + b.score {rng.choice(['<', '>=', '='])} {rng.randrange(2, 9)}"
        op = rng.choice(["=", "<>", "<", "<=", ">", ">="])
        if rng.random() < 0.4:
            return f"a.score {op} b.score"
        return f"{col} {op} {rng.randrange(0, 6)}"
    kind = rng.random()
    if kind < 0.2:
        return f"NOT ({random_condition(rng, depth - 1)})"
    kids = [random_condition(rng, depth - 1) for _ in range(rng.randrange(2, 4))]
    return "(" + (" AND " if kind < 0.55 else " OR ").join(kids) + ")"


def giql_query(kind: str, on: str, where: str) -> str:
    join = {"INNER": "JOIN", "SEMI": "SEMI JOIN", "ANTI": "ANTI JOIN"}[kind]
    cols = "a.name AS an, a.start AS s, b.name AS bn, b.end AS e" if kind == "INNER" else "a.name, a.start, a.score"
    q = f"SELECT {cols} FROM peaks a {join} genes b ON a.interval INTERSECTS b.interval"
    if on:
        q += f" AND ({on})"
    if where:
        q += f" WHERE {where}"
    return q


def sqlite_rows(conn, kind: str, on: str, where: str):
    quote = lambda t: (t.replace("a.start", 'a."start"').replace("b.start", 'b."start"').replace("b.end", 'b."end"')
                       .replace("a.end", 'a."end"').replace("LEAST(", "min(").replace("GREATEST(", "max(").replace("ABS(", "abs("))
    cond = OVERLAP + (f" AND ({quote(on)})" if on else "")
    if kind == "INNER":
        sql = f'SELECT a.name, a."start", b.name, b."end" FROM peaks a JOIN genes b ON {cond}'
        if where:
            sql += f" WHERE {quote(where)}"
    else:
        ex = "EXISTS" if kind == "SEMI" else "NOT EXISTS"
        sql = f'SELECT a.name, a."start", a.score FROM peaks a WHERE {ex} (SELECT 1 FROM genes b WHERE {cond})'
        if where:
            sql += f" AND ({quote(where)})"
    return conn.execute(sql).fetchall()


def rand_rows(rng, n, tag):
    out = []
    for i in range(n):
        s = rng.randrange(0, 1500)
        score = None if rng.random() < 0.15 else rng.randrange(0, 6)
        name = None if rng.random() < 0.1 else f"{tag}{i % 7}"
        out.append((rng.choice(["chr1", "chr2", "chr3"]), s, s + rng.randrange(1, 300), name, score, rng.choice("+-")))
    return out


def main() -> None:
    rng = random.Random(20261006)
    cases = []
    for kind, on, where in CONDITIONS:
        for n_p, n_g in [(6, 5), (60, 45)]:
            peaks, genes = rand_rows(rng, n_p, "p"), rand_rows(rng, n_g, "g")
            if kind != "INNER":
                peaks.append(("chr9", 5, 50, "p1", 3, "+"))   # a chromosome only the left table has
            conn = sqlite3.connect(":memory:")
            for t, rows in (("peaks", peaks), ("genes", genes)):
                conn.execute(f'CREATE TABLE {t} (chrom TEXT, "start" INTEGER, "end" INTEGER, name TEXT, score INTEGER, strand TEXT)')
                conn.executemany(f"INSERT INTO {t} VALUES (?, ?, ?, ?, ?, ?)", rows)
            got = sqlite_rows(conn, kind, on, where)
            conn.close()
            key = lambda r: tuple((x is None, x) for x in r)
            cases.append({"kind": kind, "query": giql_query(kind, on, where), "peaks": [list(r) for r in peaks],
                          "genes": [list(r) for r in genes], "rows": [list(r) for r in sorted(got, key=key)]})
    assert sum(1 for c in cases if c["rows"]) >= len(cases) * 2 // 3
    having = []
    for sel, group, cond in HAVING:
        for n_p, n_g in [(8, 6), (70, 60)]:
            peaks, genes = rand_rows(rng, n_p, "p"), rand_rows(rng, n_g, "g")
            conn = sqlite3.connect(":memory:")
            for t, rows in (("peaks", peaks), ("genes", genes)):
                conn.execute(f'CREATE TABLE {t} (chrom TEXT, "start" INTEGER, "end" INTEGER, name TEXT, score INTEGER, strand TEXT)')
                conn.executemany(f"INSERT INTO {t} VALUES (?, ?, ?, ?, ?, ?)", rows)
            got = conn.execute(f"SELECT {sel} FROM peaks a JOIN genes b ON {OVERLAP} GROUP BY {group} HAVING {cond}").fetchall()
            conn.close()
            key = lambda r: tuple((x is None, x) for x in r)
            having.append({"query": f"SELECT {sel} FROM peaks a JOIN genes b ON a.interval INTERSECTS b.interval "
                                    f"GROUP BY {group} HAVING {cond}",
                           "peaks": [list(r) for r in peaks], "genes": [list(r) for r in genes],
                           "rows": [list(r) for r in sorted(got, key=key)]})
    assert sum(1 for c in having if c["rows"]) >= len(having) // 2
    arith = []
    for kind, on, where in ARITH:
        for n_p, n_g in [(6, 5), (60, 45)]:
            peaks, genes = rand_rows(rng, n_p, "p"), rand_rows(rng, n_g, "g")
            conn = sqlite3.connect(":memory:")
            for t, rows in (("peaks", peaks), ("genes", genes)):
                conn.execute(f'CREATE TABLE {t} (chrom TEXT, "start" INTEGER, "end" INTEGER, name TEXT, score INTEGER, strand TEXT)')
                conn.executemany(f"INSERT INTO {t} VALUES (?, ?, ?, ?, ?, ?)", rows)
            got = sqlite_rows(conn, kind, on, where)
            conn.close()
            key = lambda r: tuple((x is None, x) for x in r)
            arith.append({"kind": kind, "query": giql_query(kind, on, where), "peaks": [list(r) for r in peaks],
                          "genes": [list(r) for r in genes], "rows": [list(r) for r in sorted(got, key=key)]})
    assert sum(1 for c in arith if c["rows"]) >= len(arith) * 2 // 3
    # nested conditions (round 4): random three-level combinations, most of them past what a conjunctive normal form
    # of 12 comparisons holds -- the hip target runs those as boolean programs.  A generator of its own: the cases
    # above stay what they were.
    rng2 = random.Random(20261104)
    nested = []
    for i in range(72):
        kind = ("INNER", "INNER", "INNER", "SEMI", "ANTI")[i % 5]
        cond = random_condition(rng2, 3)
        on, where = (cond, "") if i % 2 == 0 else ("", cond)
        if kind != "INNER" and where and "b." in where:      # (a SEMI / ANTI WHERE reads the left table only)
            on, where = where, ""
        peaks, genes = rand_rows(rng2, 40, "p"), rand_rows(rng2, 35, "g")
        conn = sqlite3.connect(":memory:")
        for t, rows in (("peaks", peaks), ("genes", genes)):
            conn.execute(f'CREATE TABLE {t} (chrom TEXT, "start" INTEGER, "end" INTEGER, name TEXT, score INTEGER, strand TEXT)')
            conn.executemany(f"INSERT INTO {t} VALUES (?, ?, ?, ?, ?, ?)", rows)
        got = sqlite_rows(conn, kind, on, where)
        conn.close()
        key = lambda r: tuple((x is None, x) for x in r)
        nested.append({"kind": kind, "query": giql_query(kind, on, where), "peaks": [list(r) for r in peaks],
                       "genes": [list(r) for r in genes], "rows": [list(r) for r in sorted(got, key=key)]})
    doc = {"_source": "tests/golden/make_boolean_residuals.py: sqlite3 evaluates the overlap join AND the condition "
                      "text the reference would inline (intersects_duckdb.py:889-912, 1239-1243); rows sorted with NULLs last "
                      "per column; table rows are (chrom, start, end, name, score, strand)",
           "cases": cases, "having": having, "arith": arith, "nested": nested}
    with open(os.path.join(HERE, "boolean_residuals.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print(len(cases), "cases,", sum(len(c["rows"]) for c in cases), "rows;", len(having), "HAVING cases,",
          sum(len(c["rows"]) for c in having), "rows;", len(arith), "arithmetic cases,", sum(len(c["rows"]) for c in arith), "rows;", len(nested), "nested cases,", sum(len(c["rows"]) for c in nested), "rows")


if __name__ == "__main__":
    main()
