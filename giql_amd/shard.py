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


def plan_units(n_a: Sequence[int], n_b: Sequence[int], n_bins: int, split_over: float = 1.0,
               split_side: str | None = None):
    """Work units for ``n_bins`` ranks: ``[(chrom, part, n_parts, split_side, weight)]``.

    A chromosome is normally one unit.  One whose rows (A + B) exceed
    ``split_over`` x the ideal per-rank share is cut into ``n_parts`` units by ROW
    RANGES of its larger side (``split_side`` "a" or "b"); every such unit carries
    the whole of the chromosome's other side, so each (a, b) pair is still found
    exactly once ("if one chromosome dominates, split by row
    ranges and replicate the other side").  ``split_side="a"``: only ever cut along A
    (the per-row operators SEMI / ANTI / COUNT / NEAREST: every A row needs ALL the B
    rows of its chromosome, so B is never the side that is cut).  Deterministic.
    """
    if split_side not in (None, "a", "b"):
        raise ValueError("split_side must be None, 'a' or 'b'")
    if n_bins < 1:
        raise ValueError("n_bins must be >= 1")
    total = float(sum(n_a) + sum(n_b))
    share = total / n_bins if n_bins else total
    units = []
    for c, (na, nb) in enumerate(zip(n_a, n_b)):
        na, nb = int(na), int(nb)
        w = na + nb
        if w == 0:
            continue
        side = split_side or ("a" if na >= nb else "b")
        big, small = (na, nb) if side == "a" else (nb, na)
        k = 1
        if n_bins > 1 and w > split_over * share and big > 1:
            # smallest k whose units (big/k + replicated small) fit the share, capped at n_bins
            k = n_bins
            for cand in range(2, n_bins + 1):
                if big / cand + small <= share:
                    k = cand
                    break
            k = min(k, big)
            if split_side and small >= big and big / k + small > share:
                k = 1  # cutting the smaller side only replicates the larger one: keep the chromosome whole
        for j in range(k):
            units.append((c, j, k, side, big / k + small))
    return units


def assign_units(units, n_bins: int) -> list[int]:
    """LPT assignment of :func:`plan_units` units to ranks."""
    return lpt_assign([u[4] for u in units], n_bins)


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
