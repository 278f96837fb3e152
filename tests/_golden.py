"""Helpers shared by the parity tests: fixture loading and expected-row shaping."""

from __future__ import annotations

import json
import os

import numpy as np

from oracle import pyoracle as ora

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name: str):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)


def sides_of(case):
    """Dictionary-encode both sides with a SHARED dictionary."""
    ids: dict = {}
    a = ora.make_side([tuple(r) for r in case["a"]], case["enc_a"], ids)
    b = ora.make_side([tuple(r) for r in case["b"]], case["enc_b"], ids)
    return a, b


def rows_of_pairs(case, pairs):
    """(row_a,row_b) -> sorted (a_chrom,a_start,a_end,b_chrom,b_start,b_end) rows."""
    out = []
    for ra, rb in np.asarray(pairs).reshape(-1, 2).tolist():
        out.append(tuple(case["a"][ra]) + tuple(case["b"][rb]))
    return sorted(out)


def rows_of_left(case, rids):
    return sorted(tuple(case["a"][int(r)]) for r in rids)


def nearest_rows(case, idx_b, dist, with_distance: bool):
    """Per A row with a hit: (a_chrom, a_start, b_start[, distance])."""
    out = []
    for i, (j, d) in enumerate(zip(np.asarray(idx_b).tolist(), np.asarray(dist).tolist())):
        if j < 0:
            continue
        row = (case["a"][i][0], case["a"][i][1], case["b"][j][1])
        out.append(row + ((d,) if with_distance else ()))
    return sorted(out)
