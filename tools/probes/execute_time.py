#!/usr/bin/env python3
"""Where execute() spends its wall time on Arrow tables of BASELINE size (host side included): two runs of one
INTERSECTS query through transpile + execute with a cProfile of the second.  usage: execute_time.py [n_a n_b]"""
import cProfile
import pstats
import sys
import time

import os

import numpy as np
import pyarrow as pa

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from giql_amd.execute import execute
from giql_amd.synth import HG38_NAMES, make_table
from giql_amd.transpile import transpile

n_a = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n_b = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
names = np.array(HG38_NAMES)


def table(n, seed, kind):
    c, s, e = make_table(n, seed, kind)
    return pa.table({"chrom": pa.array(names[c]), "start": s, "end": e, "score": (s % 1000).astype(np.int32)})


t0 = time.time()
tables = {"peaks": table(n_a, 1, "peaks"), "reads": table(n_b, 2, "reads")}
print(f"tables built in {time.time() - t0:.1f} s", flush=True)
QUERIES = [
    ("pairs", "SELECT a.start AS s, b.start AS t FROM peaks a JOIN reads b ON a.interval INTERSECTS b.interval"),
    ("residual", "SELECT a.start AS s, b.score AS t FROM peaks a JOIN reads b ON a.interval INTERSECTS b.interval AND a.score > b.score"),
    ("overlap fraction", "SELECT a.start AS s, b.start AS t FROM peaks a, reads b WHERE a.interval INTERSECTS b.interval "
                         "AND (LEAST(a.end, b.end) - GREATEST(a.start, b.start)) >= 0.5 * (b.end - b.start)"),
    ("semi", "SELECT a.chrom, a.start FROM peaks a SEMI JOIN reads b ON a.interval INTERSECTS b.interval"),
    ("count", 'SELECT a.chrom, a.start, a."end", COUNT(b.chrom) AS n FROM peaks a LEFT JOIN reads b ON a.interval INTERSECTS b.interval '
              'GROUP BY a.chrom, a.start, a."end"'),
]
if os.environ.get("PROBE_SET") == "other":      # NEAREST / CLUSTER / MERGE instead of the joins
    QUERIES = [
        ("nearest k=1", "SELECT a.start AS s, b.start AS t, b.distance FROM peaks a CROSS JOIN LATERAL "
                        "NEAREST(reads, reference := a.interval) b"),
        ("nearest k=3", "SELECT a.start AS s, b.start AS t, b.distance FROM peaks a CROSS JOIN LATERAL "
                        "NEAREST(reads, reference := a.interval, k := 3) b"),
        ("cluster", "SELECT *, CLUSTER(interval, 100) AS cid FROM reads"),
        ("merge", "SELECT MERGE(interval), COUNT(*) AS n FROM reads"),
        ("cluster pred", "SELECT *, CLUSTER(interval, 100, predicate := score >= PREV(score)) AS cid FROM reads"),
        ("filter", "SELECT * FROM reads WHERE interval INTERSECTS 'chr2:1000000-90000000' AND (score > 500 OR score < 10)"),
    ]
for name, q in QUERIES:
    plan = transpile(q, tables=["peaks", "reads"], dialect="hip")
    t = time.time()
    out = execute(plan, tables)
    first = time.time() - t
    pr = cProfile.Profile()
    t = time.time()
    pr.enable()
    out = execute(plan, tables)
    pr.disable()
    second = time.time() - t
    print(f"== {name}: {out.num_rows} rows; first {first:.2f} s, second {second:.2f} s", flush=True)
    st = pstats.Stats(pr)
    st.sort_stats("cumulative")
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][3])[:14]
    for (fn, line, func), (cc, nc, tt, ct, _callers) in rows:
        print(f"   {ct:7.2f} s cum {tt:7.2f} s own  {func}  ({fn.split('/')[-1]}:{line})")
