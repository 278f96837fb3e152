#!/usr/bin/env python3
"""Print per-dispatch duration and one PMC counter for the kernels whose name matches a pattern,
from a rocprofv3 --pmc <COUNTER> --kernel-trace database.  usage: pmc_kernel.py <db> <counter> <pattern>"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
counter, pat = sys.argv[2], sys.argv[3]
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
rows = db.execute("select * from counters_collection where counter_name = ?", (counter,)).fetchall()
ix = {c: i for i, c in enumerate(cols)}
out = []
for r in rows:
    name = r[ix["kernel_name"]]
    if not re.search(pat, name):
        continue
    dur = (r[ix["end"]] - r[ix["start"]]) if "end" in ix and "start" in ix else None
    out.append((r[ix.get("dispatch_id", 0)], dur, r[ix["value"]]))
out.sort()
for d, dur, v in out:
    print("dispatch %s dur_us %s %s %.4f GB(raw x1024? see guide)" % (d, None if dur is None else round(dur / 1e3, 1), counter, v * 1024.0 / 1e9 if counter.endswith("SIZE") else v))
print("columns:", cols)
