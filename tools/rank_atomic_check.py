#!/usr/bin/env python3
"""Quick sort-stability check for experimental kernel builds: SEMI and COUNT parity
against the oracle on four inputs with heavy digit collisions.  Both operators have
bounded outputs, so a wrong sort cannot fault.  Point GIQL_HIP_LIB at the library
under test.  (First used to show that ranking with one returning LDS atomic per item
is stable on gfx950 -- the LDS serialises same-address lanes in lane order -- though
only 3.6 % faster than the ballot match, so it was not adopted.)"""
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
