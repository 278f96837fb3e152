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


def cluster_side(case):
    """One CLUSTER / MERGE input side: partition ids = dictionary-encoded chrom (sorted),
    with the strand folded in when the case is stranded; raw coordinates (offsets 0).
    Returns ``(side, parts)`` where ``parts[id]`` = the partition's ``(chrom[, strand])``."""
    rows = case["rows"]
    key = (lambda r: (r[0], r[3])) if case["stranded"] else (lambda r: (r[0],))
    parts = sorted({key(r) for r in rows})
    ids = {k: i for i, k in enumerate(parts)}
    side = ora.Side(np.array([ids[key(r)] for r in rows], np.int32),
                    np.array([r[1] for r in rows], np.int32), np.array([r[2] for r in rows], np.int32))
    return side, parts


def check_cluster_case(case, ids, merged_rows):
    """Assert cluster ids (per input row) and merged regions ``(part id, start, end, count)``
    against a cluster_merge.json case."""
    side, parts = cluster_side(case)
    ids = [int(x) for x in ids]
    if case.get("ids") is not None:
        assert ids == case["ids"], case["name"]
    for a, b in case.get("same") or []:
        assert ids[a] == ids[b], case["name"]
    for a, b in case.get("distinct") or []:
        assert ids[a] != ids[b], case["name"]
    if case.get("merged") is not None:
        got = sorted(tuple(parts[int(p)]) + (int(s), int(e), int(c)) for p, s, e, c in merged_rows)
        want = sorted(tuple(m) for m in case["merged"])
        if want and len(want[0]) < len(got[0]):  # known answers carry no COUNT(*)
            got = sorted(g[:-1] for g in got)
        assert got == want, case["name"]
