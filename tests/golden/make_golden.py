#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/ (run HERE only, never on the GPU box).

Two fixture files are written:

``known_answers.json``
    Inputs and expected rows transcribed as DATA from the reference's own
    known-answer tests for this path (each case cites its source test,
    path:line under /root/reference/).

``fuzz_sqlite.json``
    Seeded random inputs whose expected outputs are minted by stdlib ``sqlite3``
    executing the SQL text the reference emits for this path:

    * the naive overlap predicate
      ``(a."chrom" = b."chrom" AND a."start" < b."end" AND a."end" > b."start")``
      (text pinned by tests/expanders/test_intersects.py:108-110; built by
      src/giql/expanders/intersects.py:149-154), with the canonicalisation
      wrappers ``(x - 1)`` / ``(x + 1)`` exactly as src/giql/canonical.py:16-52
      renders them, in the INNER / EXISTS / NOT EXISTS shapes of
      src/giql/expanders/intersects_duckdb.py:1254-1299;
    * the NEAREST distance CASE produced by the reference's own
      ``generate_distance_case`` (src/giql/expanders/_distance.py:22-117), which
      is sqlglot-free and is loaded BY FILE PATH from /root/reference at
      generation time, inside the ``ORDER BY ABS(distance), start, end LIMIT k``
      wrapper of src/giql/expanders/nearest.py:387-396.

``cluster_merge.json``
    CLUSTER / MERGE: known answers transcribed from the reference's tests, plus seeded
    random inputs whose cluster ids and merged regions are minted by sqlite3
    executing the two-level window SQL the reference emits
    (src/giql/expanders/cluster.py:210-420, src/giql/expanders/merge.py:186-330;
    fragments pinned by tests/expanders/test_cluster.py:149-184).

The reference tree cannot travel to the GPU box, so only the JSON (data) is
committed; this script is the provenance record.
"""

from __future__ import annotations

import importlib.util
import json
import os
import random
import sqlite3

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

ENCODINGS = [
    ("0based", "half_open"),
    ("0based", "closed"),
    ("1based", "half_open"),
    ("1based", "closed"),
]


def _load_by_path(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def canonical_start_sql(raw: str, enc) -> str:
    # src/giql/canonical.py:16-29
    return raw if enc[0] == "0based" else f"({raw} - 1)"


def canonical_end_sql(raw: str, enc) -> str:
    # src/giql/canonical.py:32-52
    if tuple(enc) == ("0based", "closed"):
        return f"({raw} + 1)"
    if tuple(enc) == ("1based", "half_open"):
        return f"({raw} - 1)"
    return raw


def overlap_predicate_sql(enc_a, enc_b) -> str:
    # src/giql/expanders/intersects.py:149-154 over canonicalised fragments
    # (intersects_duckdb.py:1221-1238)
    a_s = canonical_start_sql('a."start"', enc_a)
    a_e = canonical_end_sql('a."end"', enc_a)
    b_s = canonical_start_sql('b."start"', enc_b)
    b_e = canonical_end_sql('b."end"', enc_b)
    return f'(a."chrom" = b."chrom" AND {a_s} < {b_e} AND {a_e} > {b_s})'


def _mk(conn, name, rows):
    conn.execute(f'CREATE TABLE {name} (rid INTEGER, chrom TEXT, "start" INTEGER, "end" INTEGER)')
    conn.executemany(f"INSERT INTO {name} VALUES (?, ?, ?, ?)",
                     [(i, r[0], r[1], r[2]) for i, r in enumerate(rows)])


def run_sqlite_join(a_rows, b_rows, enc_a, enc_b):
    conn = sqlite3.connect(":memory:")
    _mk(conn, "peaks", a_rows)
    _mk(conn, "genes", b_rows)
    pred = overlap_predicate_sql(enc_a, enc_b)
    inner = conn.execute(
        f"SELECT a.rid, b.rid FROM peaks a JOIN genes b ON {pred} ORDER BY 1, 2").fetchall()
    semi = conn.execute(
        f"SELECT a.rid FROM peaks a WHERE EXISTS (SELECT 1 FROM genes b WHERE {pred}) ORDER BY 1"
    ).fetchall()
    anti = conn.execute(
        f"SELECT a.rid FROM peaks a WHERE NOT EXISTS (SELECT 1 FROM genes b WHERE {pred}) ORDER BY 1"
    ).fetchall()
    count = conn.execute(
        f"SELECT a.rid, COUNT(b.rid) FROM peaks a LEFT JOIN genes b ON {pred} "
        "GROUP BY a.rid ORDER BY 1").fetchall()
    conn.close()
    return {
        "inner": [list(r) for r in inner],
        "semi": [r[0] for r in semi],
        "anti": [r[0] for r in anti],
        "count": [r[1] for r in count],
    }


def run_sqlite_nearest(distance_mod, a_rows, b_rows, enc_a, enc_b, k, signed, max_distance):
    """Standalone (literal-reference) form of nearest.py:387-396, once per A row."""
    conn = sqlite3.connect(":memory:")
    _mk(conn, "genes", b_rows)
    out = []
    for (ac, as_, ae) in a_rows:
        ref_chrom = "'" + ac.replace("'", "''") + "'"
        ref_start = canonical_start_sql(str(as_), enc_a)
        ref_end = canonical_end_sql(str(ae), enc_a)
        t_start = canonical_start_sql('genes."start"', enc_b)
        t_end = canonical_end_sql('genes."end"', enc_b)
        case = distance_mod.generate_distance_case(
            ref_chrom, ref_start, ref_end, None,
            'genes."chrom"', t_start, t_end, None,
            stranded=False, signed=signed)
        where = [f'{ref_chrom} = genes."chrom"']
        if max_distance is not None:
            where.append(f"(ABS({case})) <= {max_distance}")
        inner = f'SELECT genes.*, {case} AS distance FROM genes WHERE {" AND ".join(where)}'
        sql = (f'SELECT x.rid, x."start", x."end", x."distance" FROM ({inner}) AS x '
               f'ORDER BY ABS(x."distance"), x."start", x."end" LIMIT {k}')
        out.append([list(r) for r in conn.execute(sql).fetchall()])
    conn.close()
    return out


# --------------------------------------------------------------- known answers
PG_PEAKS = [("chr1", 100, 200), ("chr1", 300, 400), ("chr1", 500, 600),
            ("chr2", 100, 200), ("chr2", 800, 900)]
PG_GENES = [("chr1", 150, 250), ("chr1", 500, 600), ("chr1", 700, 800),
            ("chr2", 50, 150), ("chr2", 250, 350)]
HO = ["0based", "half_open"]


def known_answers():
    """Each case: inputs + expected rows as the reference test states them."""
    K = []

    def inner(name, src, peaks, genes, expected, enc_a=HO, enc_b=HO, mode="equal"):
        K.append({"kind": "inner", "name": name, "source": src,
                  "a": [list(r) for r in peaks], "b": [list(r) for r in genes],
                  "enc_a": list(enc_a), "enc_b": list(enc_b),
                  # expected as (a_chrom, a_start, a_end, b_chrom, b_start, b_end)
                  "expected": [list(r) for r in expected], "mode": mode})

    inner("peaks_genes_truth_table", "tests/test_duckdb_iejoin.py:107-142,3683-3713",
          PG_PEAKS, PG_GENES,
          [("chr1", 100, 200, "chr1", 150, 250), ("chr1", 500, 600, "chr1", 500, 600),
           ("chr2", 100, 200, "chr2", 50, 150)])
    inner("disjoint_chroms_empty", "tests/test_duckdb_iejoin.py:3715-3744",
          [("chr1", 100, 200)], [("chr2", 100, 200)], [])
    inner("quote_in_chrom_name", "tests/test_duckdb_iejoin.py:3780-3810",
          [("chr'1", 100, 200)], [("chr'1", 150, 250)],
          [("chr'1", 100, 200, "chr'1", 150, 250)])
    inner("one_based_closed_touching_match", "tests/test_duckdb_iejoin.py:3812-3846",
          [("chr1", 100, 200)], [("chr1", 200, 300)],
          [("chr1", 100, 200, "chr1", 200, 300)],
          ["1based", "closed"], ["1based", "closed"], mode="contains")
    inner("one_based_closed_non_touching", "tests/test_duckdb_iejoin.py:3848-3882",
          [("chr1", 100, 199)], [("chr1", 200, 300)], [],
          ["1based", "closed"], ["1based", "closed"])
    inner("zero_based_closed_abut_match", "tests/test_duckdb_iejoin.py:3884-3917",
          [("chr1", 100, 200)], [("chr1", 200, 300)],
          [("chr1", 100, 200, "chr1", 200, 300)],
          ["0based", "closed"], ["0based", "closed"], mode="contains")
    inner("half_open_touching_none", "tests/test_duckdb_iejoin.py:4034-4068",
          [("chr1", 100, 200)], [("chr1", 200, 300)], [])
    inner("mixed_1based_closed_vs_0based_half_open", "tests/test_duckdb_iejoin.py:4070-4103",
          [("chr1", 100, 200)], [("chr1", 99, 200)],
          [("chr1", 100, 200, "chr1", 99, 200)], ["1based", "closed"], HO)
    inner("left_1based_half_open_offset_applied", "tests/test_duckdb_iejoin.py:4105-4171",
          [("chr1", 100, 101)], [("chr1", 99, 100)],
          [("chr1", 100, 101, "chr1", 99, 100)], ["1based", "half_open"], HO)
    inner("left_offset_not_applied_none", "tests/test_duckdb_iejoin.py:4105-4171",
          [("chr1", 100, 101)], [("chr1", 99, 100)], [], HO, HO)
    inner("empty_left", "tests/test_duckdb_iejoin.py:4173-4200",
          [], [("chr1", 150, 250)], [])
    inner("empty_right", "tests/test_duckdb_iejoin.py:4202-4229",
          [("chr1", 100, 200)], [], [])
    inner("duplicate_left_rows_keep_multiplicity", "tests/test_duckdb_iejoin.py:4514-4555",
          [("chr1", 100, 200), ("chr1", 100, 200)], [("chr1", 150, 250)],
          [("chr1", 100, 200, "chr1", 150, 250), ("chr1", 100, 200, "chr1", 150, 250)])
    enc_peaks = [("chr1", 100, 200), ("chr1", 300, 400), ("chr1", 700, 800),
                 ("chr2", 100, 200), ("chr2", 500, 600)]
    enc_genes = [("chr1", 150, 250), ("chr1", 350, 450), ("chr2", 50, 180), ("chr2", 550, 700)]
    for enc in ENCODINGS:
        inner(f"encoding_truth_table_{enc[0]}_{enc[1]}", "tests/test_duckdb_iejoin.py:5152-5233",
              enc_peaks, enc_genes,
              [("chr1", 100, 200, "chr1", 150, 250), ("chr1", 300, 400, "chr1", 350, 450),
               ("chr2", 100, 200, "chr2", 50, 180), ("chr2", 500, 600, "chr2", 550, 700)],
              enc, enc)
    inner("cross_target_single_pair",
          "tests/integration/datafusion/test_cross_target_oracle.py:98-127",
          [("chr1", 100, 500), ("chr1", 1000, 2000), ("chr2", 100, 500)],
          [("chr1", 300, 600), ("chr1", 5000, 6000), ("chr2", 9000, 9500)],
          [("chr1", 100, 500, "chr1", 300, 600)])
    inner("touching_intervals_naive_path", "tests/test_intersects_join.py:366-390",
          [("chr1", 100, 200)], [("chr1", 200, 300)], [])

    K.append({"kind": "anti", "name": "anti_keeps_left_only_chroms",
              "source": "tests/test_duckdb_iejoin.py:5873-5919",
              "a": [["chr1", 10, 20], ["chr1", 100, 200], ["chr3", 1, 1000]],
              "b": [["chr1", 50, 150]], "enc_a": HO, "enc_b": HO,
              "expected": [["chr1", 10, 20], ["chr3", 1, 1000]]})
    K.append({"kind": "semi", "name": "semi_peaks_genes",
              "source": "tests/test_duckdb_iejoin.py:49-57,107-142",
              "a": [list(r) for r in PG_PEAKS], "b": [list(r) for r in PG_GENES],
              "enc_a": HO, "enc_b": HO,
              "expected": [["chr1", 100, 200], ["chr1", 500, 600], ["chr2", 100, 200]]})
    K.append({"kind": "anti", "name": "anti_peaks_genes",
              "source": "tests/test_duckdb_iejoin.py:60-63,107-142 (SURVEY 8c: p2,p5)",
              "a": [list(r) for r in PG_PEAKS], "b": [list(r) for r in PG_GENES],
              "enc_a": HO, "enc_b": HO,
              "expected": [["chr1", 300, 400], ["chr2", 800, 900]]})

    X = "tests/integration/datafusion/test_cross_target_oracle.py"

    def nearest(name, src, peaks, genes, expected, k=1, signed=False, max_distance=None):
        K.append({"kind": "nearest", "name": name, "source": src,
                  "a": [list(r) for r in peaks], "b": [list(r) for r in genes],
                  "enc_a": HO, "enc_b": HO, "k": k, "signed": signed,
                  "max_distance": max_distance,
                  # expected as (a_chrom, a_start, b_start[, distance])
                  "expected": [list(r) for r in expected]})

    nearest("k1_three_candidates", X + ":257-292", [("chr1", 200, 300)],
            [("chr1", 1000, 1100), ("chr1", 50, 60), ("chr1", 280, 290)],
            [("chr1", 200, 280)])
    nearest("k1_duplicate_reference_rows_fan_out", X + ":325-348",
            [("chr1", 200, 300), ("chr1", 200, 300)], [("chr1", 280, 290), ("chr1", 50, 60)],
            [("chr1", 200, 280), ("chr1", 200, 280)])
    nearest("k1_partitions_by_chromosome", X + ":350-371",
            [("chr1", 200, 300), ("chr2", 200, 300)],
            [("chr1", 280, 290), ("chr2", 500, 510), ("chr2", 205, 215)],
            [("chr1", 200, 280), ("chr2", 200, 205)])
    nearest("max_distance_100_boundary", X + ":373-396", [("chr1", 200, 300)],
            [("chr1", 360, 400), ("chr1", 5000, 5100)], [("chr1", 200, 360)],
            k=1, max_distance=100)
    nearest("signed_upstream_is_negative", X + ":425-447 (k=2 there; k=1 keeps the nearer)",
            [("chr1", 200, 300)], [("chr1", 50, 60), ("chr1", 360, 400)],
            [("chr1", 200, 360, 61)], k=1, signed=True)
    nearest("signed_upstream_only", X + ":425-447 (upstream row alone: -141)",
            [("chr1", 200, 300)], [("chr1", 50, 60)], [("chr1", 200, 50, -141)],
            k=1, signed=True)
    nearest("tie_breaks_on_lower_start_end", X + ":449-480", [("chr1", 200, 300)],
            [("chr1", 50, 100), ("chr1", 400, 450)], [("chr1", 200, 50)])
    nearest("bookended_distance_is_one",
            "tests/integration/bedtools/test_nearest.py:172-200", [("chr1", 100, 200)],
            [("chr1", 200, 300)], [("chr1", 100, 200, 1)])
    return K


# ------------------------------------------------------------------------ fuzz
def rand_rows(rng, n, chroms, max_start, max_len, min_len=1):
    rows = []
    for _ in range(n):
        c = rng.choice(chroms)
        s = rng.randint(0, max_start)
        rows.append((c, s, s + rng.randint(min_len, max_len)))
    return rows


def fuzz_cases(distance_mod):
    rng = random.Random(20260301)
    cases = []
    # distribution of tests/test_duckdb_iejoin.py:5243-5262 (chr1-3, start 0-200,
    # len 1-50, <= 8 rows) plus denser and longer variants, all 16 encoding pairs
    idx = 0
    for enc_a in ENCODINGS:
        for enc_b in ENCODINGS:
            for (na, nb, ms, ml) in [(8, 8, 200, 50), (30, 40, 300, 80), (25, 25, 60, 40)]:
                a = rand_rows(rng, rng.randint(0, na), ["chr1", "chr2", "chr3"], ms, ml)
                b = rand_rows(rng, rng.randint(0, nb), ["chr1", "chr2", "chr3"], ms, ml)
                exp = run_sqlite_join(a, b, enc_a, enc_b)
                cases.append({"kind": "join", "name": f"fuzz_join_{idx}", "a": a, "b": b,
                              "enc_a": list(enc_a), "enc_b": list(enc_b), **exp})
                idx += 1
    # bedtools-style distribution (tests/integration/bedtools/test_intersect_property.py:23-52)
    for i in range(12):
        a = rand_rows(rng, rng.randint(1, 60), ["chr1", "chr2", "chr3"], 1_000_000, 200_000)
        b = rand_rows(rng, rng.randint(1, 60), ["chr1", "chr2", "chr3", "chr4"], 1_000_000, 200_000)
        exp = run_sqlite_join(a, b, HO, HO)
        cases.append({"kind": "join", "name": f"fuzz_join_bedtools_{i}", "a": a, "b": b,
                      "enc_a": HO, "enc_b": HO, **exp})
    # degenerate rows: zero-length and inverted intervals follow the literal predicate
    for i in range(12):
        a = rand_rows(rng, rng.randint(1, 30), ["chr1", "chr2"], 120, 30, min_len=-10)
        b = rand_rows(rng, rng.randint(1, 30), ["chr1", "chr2"], 120, 30, min_len=-10)
        exp = run_sqlite_join(a, b, HO, HO)
        cases.append({"kind": "join", "name": f"fuzz_join_degenerate_{i}", "a": a, "b": b,
                      "enc_a": HO, "enc_b": HO, **exp})
    # NEAREST k=1: unsigned / signed / max_distance
    idx = 0
    for enc_a in ENCODINGS:
        for enc_b in ENCODINGS:
            for (signed, md) in [(False, None), (True, None), (False, 25), (True, 40)]:
                a = rand_rows(rng, rng.randint(1, 12), ["chr1", "chr2", "chr3"], 400, 40)
                b = rand_rows(rng, rng.randint(0, 14), ["chr1", "chr2"], 400, 40)
                got = run_sqlite_nearest(distance_mod, a, b, enc_a, enc_b, 1, signed, md)
                cases.append({"kind": "nearest", "name": f"fuzz_nearest_{idx}", "a": a, "b": b,
                              "enc_a": list(enc_a), "enc_b": list(enc_b), "k": 1,
                              "signed": signed, "max_distance": md,
                              # per A row: [] or [[rid, start, end, distance]]
                              "expected": got})
                idx += 1
    return cases


# ------------------------------------------------------------- CLUSTER / MERGE
def cluster_sql(distance: int, stranded: bool) -> str:
    """The two-level window form src/giql/expanders/cluster.py:210-420 emits; fragments
    pinned by tests/expanders/test_cluster.py:149-184 and
    tests/test_cluster_predicate_transpilation.py:33-36."""
    part = 'PARTITION BY "chrom"' + (', "strand"' if stranded else "")
    window = f'OVER ({part} ORDER BY "start" NULLS LAST ROWS BETWEEN UNBOUNDED PRECEDING AND 1 PRECEDING)'
    edge = f'MAX("end") {window}' + (f" + {distance}" if distance > 0 else "")
    inner = (f'SELECT *, CASE WHEN {edge} >= "start" THEN 0 ELSE 1 END AS __giql_is_new_cluster '
             "FROM features")
    return (f'SELECT *, SUM(__giql_is_new_cluster) OVER ({part} ORDER BY "start" NULLS LAST) '
            f"AS __giql_cluster_id FROM ({inner}) AS __giql_lag_calc")


def run_sqlite_cluster(rows, distance, stranded):
    conn = sqlite3.connect(":memory:")
    conn.execute('CREATE TABLE features (rid INTEGER, chrom TEXT, "start" INTEGER, "end" INTEGER, strand TEXT)')
    conn.executemany("INSERT INTO features VALUES (?, ?, ?, ?, ?)",
                     [(i, r[0], r[1], r[2], r[3]) for i, r in enumerate(rows)])
    csql = cluster_sql(distance, stranded)
    ids = conn.execute(f"SELECT rid, __giql_cluster_id FROM ({csql}) ORDER BY rid").fetchall()
    keys = '"chrom"' + (', "strand"' if stranded else "")
    # src/giql/expanders/merge.py:186-330: GROUP BY chrom[, strand], cluster id; ORDER BY chrom, start
    merged = conn.execute(
        f'SELECT {keys}, MIN("start") AS "start", MAX("end") AS "end", COUNT(*) FROM ({csql}) AS __giql_clustered '
        f'GROUP BY {keys}, __giql_cluster_id ORDER BY "chrom", "start"' + (', "strand"' if stranded else "")).fetchall()
    conn.close()
    return [r[1] for r in ids], [list(r) for r in merged]


def cluster_merge_cases():
    rng = random.Random(20261004)
    cases = []

    def known(name, src, rows, ids=None, merged=None, same=None, distinct=None):
        cases.append({"kind": "known", "name": name, "source": src, "rows": [list(r) for r in rows],
                      "distance": 0, "stranded": False, "ids": ids, "merged": merged,
                      "same": same, "distinct": distinct})

    P = "+"
    known("cluster_shared_ids", "tests/integration/datafusion/test_cross_target_oracle.py:978-1003",
          [("chr1", 100, 200, P), ("chr1", 150, 300, P), ("chr1", 5000, 6000, P)], ids=[1, 1, 2])
    known("merge_collapses_overlapping", "tests/integration/datafusion/test_cross_target_oracle.py:1196-1217",
          [("chr1", 100, 200, P), ("chr1", 150, 300, P), ("chr1", 5000, 6000, P)],
          merged=[["chr1", 100, 300], ["chr1", 5000, 6000]])
    known("merge_empty", "tests/integration/datafusion/test_cross_target_oracle.py:1219-1236", [], ids=[], merged=[])
    known("cluster_basic", "tests/integration/bedtools/test_cluster.py:15-66",
          [("chr1", 100, 200, P), ("chr1", 150, 250, P), ("chr1", 400, 500, P)], same=[[0, 1]], distinct=[[0, 2]])
    known("cluster_contained", "tests/integration/bedtools/test_cluster.py:69-113",
          [("chr1", 0, 1000, P), ("chr1", 100, 200, P), ("chr1", 300, 400, P)], same=[[0, 1], [1, 2]])
    known("cluster_separated", "tests/integration/bedtools/test_cluster.py:116-160",
          [("chr1", 100, 200, P), ("chr1", 300, 400, P), ("chr1", 500, 600, P)], distinct=[[0, 1], [1, 2], [0, 2]])

    def rows(n, chroms, max_start, max_len):
        return [(c, s, s + rng.randint(1, max_len), rng.choice("+-"))
                for c, s in ((rng.choice(chroms), rng.randint(0, max_start)) for _ in range(n))]

    idx = 0
    for stranded in (False, True):
        for distance in (0, 10, 100):
            for (n, ms, ml) in [(0, 10, 5), (1, 10, 5), (12, 200, 40), (60, 600, 50), (80, 200, 30),
                                (40, 1_000_000, 200_000)]:
                r = rows(n, ["chr1", "chr2", "chr3"], ms, ml)
                # fixtures of tests/integration/bedtools/test_merge.py:15-110 ride along as rows
                if n == 12:
                    r += [("chr1", 100, 200, P), ("chr1", 200, 300, P), ("chr1", 300, 400, P)]
                ids, merged = run_sqlite_cluster(r, distance, stranded)
                cases.append({"kind": "sqlite", "name": f"fuzz_cluster_{idx}", "rows": [list(x) for x in r],
                              "distance": distance, "stranded": stranded, "ids": ids, "merged": merged})
                idx += 1
    return cases


def main() -> None:
    distance_mod = _load_by_path(
        "_ref_distance", os.path.join(REF, "src/giql/expanders/_distance.py"))
    # self-check against the reference's own known answer (-141 / +61,
    # tests/integration/datafusion/test_cross_target_oracle.py:425-447)
    got = run_sqlite_nearest(distance_mod, [("chr1", 200, 300)],
                             [("chr1", 50, 60), ("chr1", 360, 400)], HO, HO, 2, True, None)
    assert [(r[1], r[3]) for r in got[0]] == [(360, 61), (50, -141)], got
    # and the peaks_genes truth table through the predicate text
    pg = run_sqlite_join(PG_PEAKS, PG_GENES, HO, HO)
    assert pg["inner"] == [[0, 0], [2, 1], [3, 3]], pg
    assert pg["anti"] == [1, 4], pg

    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(known_answers(), f, indent=1)
    with open(os.path.join(HERE, "fuzz_sqlite.json"), "w") as f:
        json.dump(fuzz_cases(distance_mod), f)
    # CLUSTER / MERGE: self-check the window SQL against the reference's known answers first
    ids, merged = run_sqlite_cluster([("chr1", 100, 200, "+"), ("chr1", 150, 300, "+"), ("chr1", 5000, 6000, "+")], 0, False)
    assert ids == [1, 1, 2] and [m[:3] for m in merged] == [["chr1", 100, 300], ["chr1", 5000, 6000]], (ids, merged)
    with open(os.path.join(HERE, "cluster_merge.json"), "w") as f:
        json.dump(cluster_merge_cases(), f)
    print("wrote known_answers.json, fuzz_sqlite.json, cluster_merge.json")


if __name__ == "__main__":
    main()
