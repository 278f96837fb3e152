#!/usr/bin/env python3
"""Tuning aid: time the span + linearize phases alone on 100M rows (GIQL_LIN_ABLATE
builds stop after B's linearize; build 4 = nothing removed, 1 = no histogram,
2 = no key stores, 3 = neither)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd import synth
from giql_amd.engine import DeviceSide, HipEngine
eng = HipEngine(0)
b = DeviceSide.from_numpy(*synth.make_table(100_000_000, 6, "reads"))
a = DeviceSide.from_numpy(*synth.make_table(1000, 5, "peaks"))
for _ in range(2):
    eng.semi_join(a, b, 24)
eng.set_profiling(True)
acc = {}
for _ in range(5):
    eng.semi_join(a, b, 24)
    for k, v in eng.stats()["phase_ms"].items():
        acc[k] = acc.get(k, 0) + v / 5
print(json.dumps({"lib": os.environ.get("GIQL_HIP_LIB", "default"), "span_ms": round(acc["span"], 3), "linearize_ms": round(acc["linearize"], 3)}))
