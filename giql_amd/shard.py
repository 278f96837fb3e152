"""Chromosome sharding for multi-GPU joins (one process per GPU).

The predicate requires ``a.chrom = b.chrom`` and the reference itself partitions
per chromosome and concatenates with ``UNION ALL``
(``src/giql/expanders/_per_chrom.py:3-9, 62-69``;
``src/giql/expanders/intersects_duckdb.py:1317-1330``), so chromosomes are
independent units: each rank joins its own chromosomes with no data-path
collective, and the only exchange is the final gather of the index pairs.
"""

from __future__ import annotations

from typing import Sequence


def lpt_assign(weights: Sequence[float], n_bins: int) -> list[int]:
    """Longest-processing-time-first packing: ``assign[c]`` = bin of item ``c``.

    Deterministic (ties broken by index), so every rank computes the same map.
    """
    if n_bins < 1:
        raise ValueError("n_bins must be >= 1")
    load = [0.0] * n_bins
    assign = [0] * len(weights)
    order = sorted(range(len(weights)), key=lambda c: (-float(weights[c]), c))
    for c in order:
        b = min(range(n_bins), key=lambda k: (load[k], k))
        assign[c] = b
        load[b] += float(weights[c])
    return assign


def span_groups(spans: Sequence[int], limit: int = 2**32 - 2) -> list[list[int]]:
    """Split chromosomes into groups whose summed coordinate span fits 32 bits.

    The HIP path linearises (chrom, position) onto one u32 axis; a genome longer
    than 2^32 is joined group by group (groups are independent for the same
    reason shards are).  First-fit decreasing.
    """
    groups: list[list[int]] = []
    room: list[int] = []
    for c in sorted(range(len(spans)), key=lambda c: (-int(spans[c]), c)):
        s = int(spans[c])
        if s > limit:
            raise ValueError(f"chromosome {c} alone spans {s} > {limit}")
        for g, r in enumerate(room):
            if s <= r:
                groups[g].append(c)
                room[g] -= s
                break
        else:
            groups.append([c])
            room.append(limit - s)
    return [sorted(g) for g in groups]
