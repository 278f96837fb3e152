#!/usr/bin/env python3
"""Experiment: does the LDS serialise same-address lanes of ONE returning atomic in
ascending lane order?  The GIQL_RANK_ATOMIC build ranks with ds_add_rtn instead of the
ballot match; the LSD sort is only correct if that order is stable.  SEMI / COUNT are
bounded-output paths, so a wrong sort cannot fault.  Run with GIQL_HIP_LIB pointing at
the experimental library."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd.engine import DeviceSide, HipEngine
from oracle import pyoracle as ora

eng = HipEngine(0)
ok = True
for seed, (na, nb, span, nch) in enumerate([(200_000, 300_000, 5_000_000, 3), (2_000_000, 3_000_000, 60_000_000, 5),
                                            (50_000, 4_000_000, 1_000, 1), (1_000_000, 1_000_000, 2_000_000_00, 2)]):
    rng = np.random.default_rng(seed)
    def side(n):
        c = rng.integers(0, nch, n).astype(np.int32)
        s = rng.integers(0, span, n).astype(np.int32)
        return c, s, (s + rng.integers(1, 500, n)).astype(np.int32)
    A, B = side(na), side(nb)
    a, b = DeviceSide.from_numpy(*A), DeviceSide.from_numpy(*B)
    oa, ob = ora.Side(*A), ora.Side(*B)
    semi = eng.semi_join(a, b, nch).cpu().numpy()
    cnt = eng.count_overlaps(a, b, nch).cpu().numpy()
    good = bool(np.array_equal(semi, ora.c_semi_anti(oa, ob, False)) and np.array_equal(cnt, ora.c_count(oa, ob, "sweep")))
    ok &= good
    print(json.dumps({"case": [na, nb, span, nch], "parity": good}), flush=True)
print("ALL_OK" if ok else "MISMATCH")
