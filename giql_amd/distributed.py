"""Multi-GPU execution: one process per GPU, chromosomes sharded, one exchange.

Chromosomes are independent units of the join (the reference partitions per
chromosome and concatenates with UNION ALL: ``src/giql/expanders/_per_chrom.py:3-9,
62-69``), so every rank joins its own chromosomes with NO data-path collective;
the path's single exchange step is the final gather of the (row_a, row_b) index
pairs -- over RCCL (``backend="nccl"`` is RCCL on ROCm) when the tensors are on
GPUs, over gloo in the CPU tests.

xGMI is a point-to-point mesh, so the gather is issued as one all-gather of
equal-sized (padded) shards: every shard crosses each link once, in parallel,
instead of a ring's serial hops.
"""

from __future__ import annotations

from typing import Callable, Sequence

import numpy as np

from .shard import lpt_assign


def shard_rows(chrom: np.ndarray, assign: Sequence[int], rank: int) -> np.ndarray:
    """Row indices (ascending) whose chromosome is assigned to ``rank``."""
    mine = np.asarray([r == rank for r in assign], dtype=bool)
    return np.nonzero(mine[np.asarray(chrom)])[0]


def plan_shards(chrom_a: np.ndarray, chrom_b: np.ndarray, n_chrom: int, world: int) -> list[int]:
    """LPT assignment of chromosomes to ranks by their row counts (A + B)."""
    w = np.bincount(chrom_a, minlength=n_chrom) + np.bincount(chrom_b, minlength=n_chrom)
    return lpt_assign(w.tolist(), world)


def gather_pairs(row_a, row_b, group=None):
    """All-gather variable-length pair shards; returns ``(row_a_all, row_b_all, counts)``.

    Two collectives: the per-rank counts (one int64 each), then the index pairs
    packed as one ``[2, m]`` int32 block per rank, padded to the largest shard.
    Result order: rank order, then local order (the pair *set* is what matters).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = int(row_a.shape[0])
    dev = row_a.device
    count = torch.tensor([n], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, count, group=group)
    counts_h = [int(x) for x in counts.tolist()]
    m = max(counts_h) if counts_h else 0
    if m == 0:
        z = torch.empty(0, dtype=torch.int32, device=dev)
        return z, z.clone(), counts_h
    send = torch.empty((2, m), dtype=torch.int32, device=dev)
    send[0, :n] = row_a
    send[1, :n] = row_b
    recv = torch.empty((world, 2, m), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    out_a = torch.cat([recv[r, 0, : counts_h[r]] for r in range(world)])
    out_b = torch.cat([recv[r, 1, : counts_h[r]] for r in range(world)])
    return out_a, out_b, counts_h


def sharded_inner_join(a, b, n_chrom: int, local_join: Callable, *, device=None, group=None,
                       gather: bool = True):
    """Join host tables ``a``/``b`` = ``(chrom, start, end[, start_off, end_off])``
    across the ranks of ``group``; returns GLOBAL row-id pairs.

    ``local_join(chrom_a, start_a, end_a, offs_a, chrom_b, start_b, end_b, offs_b,
    n_chrom)`` runs this rank's shard and returns local ``(row_a, row_b)`` torch
    int32 tensors -- the HIP engine in production (see :func:`hip_local_join`).
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    ca, sa, ea = (np.asarray(x) for x in a[:3])
    cb, sb, eb = (np.asarray(x) for x in b[:3])
    offs_a = tuple(a[3:5]) if len(a) >= 5 else (0, 0)
    offs_b = tuple(b[3:5]) if len(b) >= 5 else (0, 0)
    assign = plan_shards(ca, cb, n_chrom, world)
    ia = shard_rows(ca, assign, rank)
    ib = shard_rows(cb, assign, rank)
    la, lb = local_join(ca[ia], sa[ia], ea[ia], offs_a, cb[ib], sb[ib], eb[ib], offs_b, n_chrom)
    dev = la.device if device is None else torch.device(device)
    # local -> global row ids (the shard's id maps)
    ga = torch.from_numpy(ia.astype(np.int32)).to(dev)[la.to(dev).long()]
    gb = torch.from_numpy(ib.astype(np.int32)).to(dev)[lb.to(dev).long()]
    if not gather:
        return ga, gb
    out_a, out_b, _ = gather_pairs(ga, gb, group=group)
    return out_a, out_b


def hip_local_join(engine):
    """``local_join`` backed by a :class:`giql_amd.engine.HipEngine`."""
    from .engine import DeviceSide

    def run(ca, sa, ea, offs_a, cb, sb, eb, offs_b, n_chrom):
        a = DeviceSide.from_numpy(ca, sa, ea, device=engine.device)
        a.start_off, a.end_off = offs_a
        b = DeviceSide.from_numpy(cb, sb, eb, device=engine.device)
        b.start_off, b.end_off = offs_b
        return engine.inner_join(a, b, n_chrom)

    return run
