#!/usr/bin/env python3
"""Probe: how much of a onesweep pass is scatter-write inefficiency?  Sort 100M rows
whose keys are (a) random, (b) all identical (every pass writes sequentially),
(c) already sorted (long sequential runs per bucket)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd.engine import DeviceSide, HipEngine

eng = HipEngine(0)
n_b, n_a = 100_000_000, 1000
rng = np.random.default_rng(1)
a = DeviceSide.from_numpy(np.zeros(n_a, np.int32), rng.integers(0, 200_000_000, n_a).astype(np.int32), rng.integers(200_000_000, 200_001_000, n_a).astype(np.int32))
cases = {}
st = rng.integers(0, 240_000_000, n_b).astype(np.int32)
cases["random"] = st
cases["identical"] = np.full(n_b, 12345, np.int32)
cases["sorted"] = np.sort(st)
only = os.environ.get("GIQL_PROBE_CASES")
for name, s in cases.items():
    if only and name not in only.split(","):
        continue
    b = DeviceSide.from_numpy(np.zeros(n_b, np.int32), s, s + np.int32(150))
    for _ in range(2):
        eng.semi_join(a, b, 1)   # sorts B as (key, end) + A; tiny A
    torch.cuda.synchronize()
    eng.set_profiling(True)
    acc = {}
    for _ in range(3):
        eng.semi_join(a, b, 1)
        for k, v in eng.stats()["phase_ms"].items():
            acc[k] = acc.get(k, 0) + v / 3
    eng.set_profiling(False)
    print(json.dumps({"case": name, "sort_scatter_ms": round(acc["sort_scatter"], 3), "per_pass_ms": round(acc["sort_scatter"] / 4, 3),
                      "alg_GBps_per_pass": round(16 * n_b / (acc["sort_scatter"] / 4 * 1e-3) / 1e9, 1)}), flush=True)
    del b
