#!/usr/bin/env python3
"""Scale check past 2^32 pairs: 30M peaks x 300M reads (3x the headline sizes, ~3.6e9 pairs, 29 GB of
output).  No oracle at this size: the pair count must equal the sum of the COUNT operator's per-row
counts, every pair must satisfy the predicate on a strided sample, row ids must be in range, and the
SEMI count must equal the number of rows with a non-zero count."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd import synth
from giql_amd.engine import DeviceSide, HipEngine

na, nb = int(os.environ.get("BIG_NA", 30_000_000)), int(os.environ.get("BIG_NB", 300_000_000))
t0 = time.time()
A = synth.make_table(na, 5, "peaks")
B = synth.make_table(nb, 6, "reads")
print(json.dumps({"gen_s": round(time.time() - t0, 1)}), flush=True)
eng = HipEngine(0)
a, b = DeviceSide.from_numpy(*A), DeviceSide.from_numpy(*B)
counts = eng.count_overlaps(a, b, 24)
total = int(counts.sum().item())
n = eng.inner_plan(a, b, 24)
print(json.dumps({"pairs": n, "sum_counts": total, "over_2^32": n > 2**32}), flush=True)
assert n == total
ra = torch.empty(n, dtype=torch.int32, device="cuda")
rb = torch.empty(n, dtype=torch.int32, device="cuda")
eng.set_profiling(True)
t0 = time.perf_counter()
eng.inner_plan(a, b, 24)
eng.inner_fill(ra, rb)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3
st = eng.stats()
assert int(ra.min()) >= 0 and int(ra.max()) < na and int(rb.min()) >= 0 and int(rb.max()) < nb
step = max(n // 50_000_000, 1)
sa, sb = ra[::step].long(), rb[::step].long()
ok = (a.chrom[sa] == b.chrom[sb]) & (a.start[sa] < b.end[sb]) & (a.end[sa] > b.start[sb])
assert bool(ok.all())
# per-row multiplicity of the pairs == the COUNT operator
got = torch.zeros(na, dtype=torch.int64, device="cuda")
got.index_add_(0, ra.long(), torch.ones(n, dtype=torch.int64, device="cuda"))
assert bool((got == counts).all())
semi = eng.semi_join(a, b, 24)
assert int(semi.shape[0]) == int((counts > 0).sum().item())
print(json.dumps({"ok": True, "join_ms": round(ms, 2), "pairs_per_s": round(n / ms * 1e3), "join_form": st["join_form"],
                  "phase_ms": {k: round(v, 3) for k, v in st["phase_ms"].items() if v > 0},
                  "workspace_GB": round(st["workspace_bytes"] / 1e9, 2)}), flush=True)
