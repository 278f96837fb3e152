"""Multi-GPU execution: one process per GPU, chromosomes sharded, one exchange.

Chromosomes are independent units of the join (the reference partitions per
chromosome and concatenates with UNION ALL: ``src/giql/expanders/_per_chrom.py:3-9,
62-69``), so every rank joins its own chromosomes with NO data-path collective;
the path's single exchange step is the final gather of the (row_a, row_b) index
pairs -- over RCCL (``backend="nccl"`` is RCCL on ROCm) when the tensors are on
GPUs, over gloo in the CPU tests.

Two implementations of that one exchange (``impl=``), both moving the same blocks:

* ``"allgather"`` -- one ``all_gather_into_tensor`` of equal-sized (padded) blocks; RCCL picks the
  algorithm (ring / tree over the xGMI links).
* ``"p2p"`` -- grouped direct ``send`` / ``recv`` (``batch_isend_irecv`` = ``ncclGroupStart`` ... ``End``):
  every rank sends its block, exactly sized, to every peer at once.  xGMI is a point-to-point mesh
  (7 links per GPU), so each block crosses ONE link and the 7 transfers of a rank run side by side;
  what SURVEY.md section 8(e) asked for.

and two result placements (``root=``): every rank ends up with the whole result (``None``, the
reference's ``UNION ALL`` seen from any rank), or only rank ``root`` does (the other ranks send their
block there and skip the expansion -- what a caller that wants ONE table needs).
"""

from __future__ import annotations

from typing import Callable, Sequence

import numpy as np

from .shard import assign_units, lpt_assign, plan_units


def shard_rows(chrom: np.ndarray, assign: Sequence[int], rank: int) -> np.ndarray:
    """Row indices (ascending) whose chromosome is assigned to ``rank``."""
    mine = np.asarray([r == rank for r in assign], dtype=bool)
    return np.nonzero(mine[np.asarray(chrom)])[0]


def plan_shards(chrom_a: np.ndarray, chrom_b: np.ndarray, n_chrom: int, world: int) -> list[int]:
    """LPT assignment of chromosomes to ranks by their row counts (A + B)."""
    w = np.bincount(chrom_a, minlength=n_chrom) + np.bincount(chrom_b, minlength=n_chrom)
    return lpt_assign(w.tolist(), world)


def unit_rows(chrom_a: np.ndarray, chrom_b: np.ndarray, n_chrom: int, world: int, rank: int,
              split_over: float = 1.0, split_side: str | None = None):
    """This rank's (rows_a, rows_b) under the unit plan of :func:`giql_amd.shard.plan_units`.

    Whole-chromosome units contribute all of the chromosome's rows on both sides.
    A split chromosome contributes the rank's row-range slices of the split side
    (slice j of k = the rows whose index within the chromosome falls in
    ``[j*n/k, (j+1)*n/k)``) and ALL rows of the other side.  Row index arrays are
    ascending; across ranks the split side's rows are disjoint, so every pair is
    produced by exactly one rank.
    """
    chrom_a = np.asarray(chrom_a)
    chrom_b = np.asarray(chrom_b)
    na = np.bincount(chrom_a, minlength=n_chrom)
    nb = np.bincount(chrom_b, minlength=n_chrom)
    units = plan_units(na.tolist(), nb.tolist(), world, split_over, split_side)
    owner = assign_units(units, world)
    keep_a = np.zeros(chrom_a.shape[0], dtype=bool)
    keep_b = np.zeros(chrom_b.shape[0], dtype=bool)
    whole = np.zeros(n_chrom, dtype=bool)
    for (c, j, k, side, _w), r in zip(units, owner):
        if r != rank:
            continue
        if k == 1:
            whole[c] = True
            continue
        split_chrom, other_keep, other_chrom = ((chrom_a, keep_b, chrom_b) if side == "a"
                                                else (chrom_b, keep_a, chrom_a))
        rows = np.nonzero(split_chrom == c)[0]
        n = rows.shape[0]
        sl = rows[(j * n) // k:((j + 1) * n) // k]
        (keep_a if side == "a" else keep_b)[sl] = True
        other_keep[other_chrom == c] = True
    keep_a |= whole[chrom_a]
    keep_b |= whole[chrom_b]
    return np.nonzero(keep_a)[0], np.nonzero(keep_b)[0]


IMPLS = ("allgather", "p2p")


def exchange_blocks(send, sizes, recv, group=None, impl="p2p", root=None, async_op=False):
    """Move one 1-D block per rank: rank ``r`` contributes ``send[:sizes[r]]`` and ``recv[r][:sizes[r]]``
    receives it (``recv`` = one preallocated 1-D tensor per rank; this rank's own block is NOT copied --
    the caller reads it from ``send``).  ``root=None``: every rank receives every block; else only rank
    ``root`` does.  Grouped point-to-point operations, exactly sized (``impl="p2p"``; the padded
    all-gather form lives with its callers, which own the padded buffers).  Returns a ``wait()``
    function when ``async_op`` else ``None`` (already waited)."""
    import torch.distributed as dist

    assert impl == "p2p"
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    ops = []
    for r in range(world):
        if r == rank:
            continue
        gr = r if group is None else dist.get_global_rank(group, r)
        if (root is None or root == r) and sizes[rank] > 0:
            ops.append(dist.P2POp(dist.isend, send[: sizes[rank]], gr, group=group))
        if (root is None or root == rank) and sizes[r] > 0:
            ops.append(dist.P2POp(dist.irecv, recv[r][: sizes[r]], gr, group=group))
    works = dist.batch_isend_irecv(ops) if ops else []

    def wait():
        for w in works:
            w.wait()

    if async_op:
        return wait
    wait()
    return None


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


class PairGather:
    """The path's exchange step with persistent buffers and no repacking.

    ``gather_pairs`` above is the simple form (pack, all-gather, concatenate).  At
    BASELINE config 4 the pairs are 3.2 GB, so every extra copy of them costs about
    as much as a kernel of the join itself; this form lets the producer write its
    GLOBAL row ids straight into the send block and hands the result back as one
    zero-copy ``(row_a, row_b)`` view per rank (rank order, then local order -- the
    pair *set* is what matters):

        n = plan(...)                         # this rank's pair count
        counts = xg.counts(n)                 # collective 1: one int64 per rank
        send = xg.send_block(max(counts))     # [2, m] int32 on the exchange device
        ... write send[0, :n], send[1, :n] ...
        blocks = xg.all_gather(counts)        # collective 2: padded all-gather
    """

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist

        self.group = group
        self.device = torch.device(device)
        self.world = dist.get_world_size(group)
        self._m = -1
        self._send = None
        self._recv = None
        self._count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)

    def counts(self, n: int) -> list[int]:
        import torch.distributed as dist

        self._count.fill_(int(n))
        dist.all_gather_into_tensor(self._counts, self._count, group=self.group)
        return [int(x) for x in self._counts.tolist()]

    def send_block(self, m: int):
        """``[2, m]`` int32 send block (grown geometrically, reused across steps)."""
        import torch

        if m > self._m:
            self._send = self._recv = None
            cap = (int(m * 1.05) + 1024 + 1023) // 1024 * 1024   # row 1 starts on a 4 KiB boundary (aligned fill stores)
            self._send = torch.empty((2, cap), dtype=torch.int32, device=self.device)
            self._recv = torch.empty((self.world, 2, cap), dtype=torch.int32, device=self.device)
            self._m = cap
        return self._send

    def all_gather(self, counts):
        """All-gather the send block; returns ``[(row_a_r, row_b_r)]``, views into the
        receive buffer, one per rank."""
        import torch.distributed as dist

        if self._send is None:
            self.send_block(max(counts) if counts else 0)
        dist.all_gather_into_tensor(self._recv.view(-1), self._send.view(-1), group=self.group)
        return [(self._recv[r, 0, :c], self._recv[r, 1, :c]) for r, c in enumerate(counts)]


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
    # chromosome units, LPT-packed; a chromosome heavier than one rank's share is
    # split by row ranges of its larger side (identical to plan_shards otherwise)
    ia, ib = unit_rows(ca, cb, n_chrom, world, rank)
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


class PlanGather:
    """The exchange step in its COMPACT form: instead of the expanded pairs (8 B per pair), ranks
    all-gather the plan that describes them -- per query row ``{row id, first matching position,
    match count}`` plus the other side's row ids in sorted order (12 B per query row + 4 B per
    row: ~65 MB instead of ~400 MB per rank at BASELINE config 4 on 8 GPUs) -- and every receiver
    expands each rank's block locally (``giql_hip_fill_from_plan_dev`` on GPUs).

        xg = PlanGather(device)
        sizes = xg.sizes(n_pairs, n_q, n_s, query_is_a)     # collective 1: four int64 per rank
        q_rid, lo, cnt, s_rid = xg.send_views(sizes)        # int32 views into the send block
        ... write the plan with GLOBAL row ids (giql_hip_inner_plan_export_dev) ...
        blocks = xg.all_gather(sizes)                        # collective 2: one padded all-gather
        for (q_rid_r, lo_r, cnt_r, s_rid_r), (n_pairs_r, _, _, q_is_a_r) in zip(blocks, sizes): expand

    A rank whose plan has no compact form reports ``n_q = -1``; :meth:`compact` is then False on
    every rank and the callers exchange the expanded pairs (:class:`PairGather`) instead.
    """

    def __init__(self, device, group=None, impl="allgather", root=None):
        import torch
        import torch.distributed as dist

        if impl not in IMPLS:
            raise ValueError(f"impl must be one of {IMPLS}")
        self.group = group
        self.device = torch.device(device)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # a gather to ONE rank is point-to-point by nature (the padded all-gather would deliver every
        # block to every rank): root mode always takes the send / recv form
        self.impl = "p2p" if root is not None else impl
        self.root = root
        self._q = self._s = -1
        self._send = None
        self._recv = None
        self._mine = torch.zeros(4, dtype=torch.int64, device=self.device)
        self._all = torch.zeros((self.world, 4), dtype=torch.int64, device=self.device)

    def receives(self, rank=None) -> bool:
        """Does ``rank`` (default: this one) end up with the other ranks' blocks?"""
        r = self.rank if rank is None else rank
        return self.root is None or self.root == r

    def sizes(self, n_pairs: int, n_q: int, n_s: int, query_is_a: bool):
        import torch
        import torch.distributed as dist

        self._mine.copy_(torch.tensor([int(n_pairs), int(n_q), int(n_s), 1 if query_is_a else 0], dtype=torch.int64))
        dist.all_gather_into_tensor(self._all.view(-1), self._mine, group=self.group)
        return [tuple(int(x) for x in row) for row in self._all.tolist()]

    @staticmethod
    def compact(sizes) -> bool:
        return all(s[1] >= 0 for s in sizes)

    def _ensure(self, sizes):
        import torch

        q = max((s[1] for s in sizes), default=0)
        s_ = max((s[2] for s in sizes), default=0)
        if q > self._q or s_ > self._s:
            self._send = self._recv = None
            self._q = (int(q * 1.02) + 1024 + 63) // 64 * 64
            self._s = (int(s_ * 1.02) + 1024 + 63) // 64 * 64
            block = 3 * self._q + self._s
            self._send = torch.empty(block, dtype=torch.int32, device=self.device)
            n_recv = self.world if self.receives() else 1   # a non-root rank of a gather-to-root receives nothing
            self._recv = torch.empty((n_recv, block), dtype=torch.int32, device=self.device)

    def _views(self, block, n_q, n_s):
        # all-gather form: the four arrays at fixed strides of the padded block; p2p form: back to back
        # (the message is exactly 3 n_q + n_s words)
        q = self._q if self.impl == "allgather" else n_q
        return block[:n_q], block[q:q + n_q], block[2 * q:2 * q + n_q], block[3 * q:3 * q + n_s]

    def send_views(self, sizes, rank=None):
        import torch.distributed as dist

        self._ensure(sizes)
        r = dist.get_rank(self.group) if rank is None else rank
        return self._views(self._send, sizes[r][1], sizes[r][2])

    def all_gather(self, sizes):
        """The exchange (blocking): per-rank views ``[(q_rid, lo, cnt, s_rid)]`` in rank order; ``None``
        for the blocks this rank does not receive (gather-to-root on a non-root rank).  This rank's own
        entry is the send block itself."""
        return self.all_gather_async(sizes)()

    def all_gather_async(self, sizes):
        """Start the exchange without blocking the caller's stream; returns a function that waits for
        it (the current stream then waits for the collective) and yields the per-rank views."""
        import torch.distributed as dist

        self._ensure(sizes)
        if self.impl == "allgather":
            work = dist.all_gather_into_tensor(self._recv.view(-1), self._send, group=self.group, async_op=True)

            def wait():
                work.wait()
                return [self._views(self._recv[r], s[1], s[2]) for r, s in enumerate(sizes)]

            return wait
        words = [3 * s[1] + s[2] if s[1] >= 0 else 0 for s in sizes]
        recv = [self._recv[r if self.receives() else 0] for r in range(self.world)]
        wait_p2p = exchange_blocks(self._send, words, recv, group=self.group, impl="p2p", root=self.root, async_op=True)

        def wait():
            wait_p2p()
            out = []
            for r, s in enumerate(sizes):
                if r == self.rank:
                    out.append(self._views(self._send, s[1], s[2]))
                elif self.receives():
                    out.append(self._views(self._recv[r], s[1], s[2]))
                else:
                    out.append(None)
            return out

        return wait

    def bytes_per_rank(self) -> int:
        return 4 * (3 * self._q + self._s) if self._send is not None else 0


def sharded_inner_join_compact(a, b, n_chrom: int, local_plan: Callable, expand: Callable, *, device=None,
                               group=None, impl="allgather", root=None):
    """:func:`sharded_inner_join` with the compact exchange (``impl`` / ``root``: see the module docstring;
    with ``root`` set, the other ranks return their own pairs only).

    ``local_plan(chrom_a, start_a, end_a, offs_a, chrom_b, ..., n_chrom)`` plans this rank's
    shard and returns ``None`` (no compact form) or ``(query_is_a, q_rid, lo, cnt, s_rid,
    n_pairs)`` with shard-LOCAL row ids (torch int32 / int64-convertible tensors);
    ``expand(q_rid, lo, cnt, s_rid, n_pairs)`` turns one plan block into ``(row_q, row_s)``.
    Returns global ``(row_a, row_b)``, every rank the whole result.
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    ca, sa, ea = (np.asarray(x) for x in a[:3])
    cb, sb, eb = (np.asarray(x) for x in b[:3])
    offs_a = tuple(a[3:5]) if len(a) >= 5 else (0, 0)
    offs_b = tuple(b[3:5]) if len(b) >= 5 else (0, 0)
    ia, ib = unit_rows(ca, cb, n_chrom, world, rank)
    plan = local_plan(ca[ia], sa[ia], ea[ia], offs_a, cb[ib], sb[ib], eb[ib], offs_b, n_chrom)
    dev = torch.device("cpu") if device is None else torch.device(device)
    xg = PlanGather(dev, group=group, impl=impl, root=root)
    if plan is None:
        sizes = xg.sizes(0, -1, -1, True)
    else:
        q_is_a, q_rid, lo, cnt, s_rid, n_pairs = plan
        sizes = xg.sizes(n_pairs, int(q_rid.shape[0]), int(s_rid.shape[0]), q_is_a)
    if not xg.compact(sizes):
        return None  # the caller falls back to the expanded exchange (collectively: every rank sees the same sizes)
    q_rid_v, lo_v, cnt_v, s_rid_v = xg.send_views(sizes)
    map_q, map_s = (ia, ib) if q_is_a else (ib, ia)
    q_rid_v.copy_(torch.from_numpy(map_q.astype(np.int32)).to(dev)[q_rid.to(dev).long()])   # local -> global ids
    s_rid_v.copy_(torch.from_numpy(map_s.astype(np.int32)).to(dev)[s_rid.to(dev).long()])
    lo_v.copy_(lo.to(dev).to(torch.int32))
    cnt_v.copy_(cnt.to(dev).to(torch.int32))
    out_a, out_b = [], []
    for blk, (n_r, _nq, _ns, qa_r) in zip(xg.all_gather(sizes), sizes):
        if n_r == 0 or blk is None:   # (None: a block this rank does not receive -- gather to another root)
            continue
        q_r, lo_r, cnt_r, s_r = blk
        row_q, row_s = expand(q_r, lo_r, cnt_r, s_r, n_r)
        out_a.append(row_q if qa_r else row_s)
        out_b.append(row_s if qa_r else row_q)
    z = torch.empty(0, dtype=torch.int32, device=dev)
    return (torch.cat(out_a) if out_a else z), (torch.cat(out_b) if out_b else z.clone())


# ------------------------------------------------------------------ the per-row operators
ROW_OPS = ("semi", "anti", "count", "nearest")


def gather_blocks(block, group=None, impl="allgather", root=None):
    """Gather one ``[k, n]`` block per rank (``n`` differs from rank to rank; same ``k`` and dtype
    everywhere): the per-rank sizes first, then the blocks -- padded to the largest in one all-gather
    (``impl="allgather"``) or exactly sized by grouped send / recv (``"p2p"``; always when ``root`` is
    given: only that rank receives).  Returns ``[block_0, block_1, ...]`` in rank order (``None`` for
    blocks this rank did not receive)."""
    import torch
    import torch.distributed as dist

    if impl not in IMPLS:
        raise ValueError(f"impl must be one of {IMPLS}")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    k, n = int(block.shape[0]), int(block.shape[1])
    dev = block.device
    count = torch.tensor([n], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, count, group=group)
    counts_h = [int(x) for x in counts.tolist()]
    m = max(counts_h) if counts_h else 0
    if m == 0:
        return [block[:, :0] for _ in range(world)]
    if impl == "allgather" and root is None:
        send = block if n == m else torch.cat([block, block.new_zeros((k, m - n))], dim=1)
        recv = torch.empty((world, k, m), dtype=block.dtype, device=dev)
        dist.all_gather_into_tensor(recv.view(-1), send.contiguous().view(-1), group=group)
        return [recv[r, :, : counts_h[r]] for r in range(world)]
    receives = root is None or root == rank
    recv = [torch.empty(k * c if (receives and r != rank) else 0, dtype=block.dtype, device=dev)
            for r, c in enumerate(counts_h)]
    exchange_blocks(block.contiguous().view(-1), [k * c for c in counts_h], recv, group=group, impl="p2p", root=root)
    return [block if r == rank else (recv[r].view(k, counts_h[r]) if receives else None) for r in range(world)]


def _i64_as_i32_rows(x):
    """An int64 vector as two int32 rows (low, high words) and back: the per-row blocks travel as int32."""
    import torch

    return x.contiguous().view(torch.int32).view(-1, 2).t()


def _i32_rows_as_i64(lo_hi):
    import torch

    return lo_hi.t().contiguous().view(-1).view(torch.int64)


def sharded_row_op(op: str, a, b, n_chrom: int, local_op: Callable, *, device=None, group=None,
                   gather: bool = True, impl="allgather", root=None, **kw):
    """SEMI / ANTI / COUNT / NEAREST k=1 of host tables ``a`` / ``b`` across the ranks of ``group``.

    The operators answer per A row from the B rows of the SAME chromosome
    (``intersects_duckdb.py:1254-1282``; ``nearest.py:313-333``), so the units are the A rows'
    chromosomes: LPT-packed like the join's, a chromosome heavier than one rank's share cut by
    row ranges of A with all of its B rows on every part (``unit_rows(..., split_side="a")``).
    No data-path collective; ONE exchange of the per-row results at the end (``impl`` / ``root``: see
    the module docstring) -- as int32 blocks: row ids and counts are int32 (a table has fewer than 2^31
    rows), only NEAREST's distance is 64-bit and travels as two int32 words (4 / 8 / 16 B per A row
    instead of 8 / 16 / 24).

    ``local_op(op, chrom_a, start_a, end_a, offs_a, chrom_b, ..., offs_b, n_chrom, **kw)`` runs
    this rank's shard: SEMI / ANTI -> local A row ids (int32); COUNT -> int64 per local A row;
    NEAREST -> ``(idx_b int32 local or -1, distance int64)`` per local A row.
    Returns, on every rank (``gather=False``: this rank's rows only, as
    ``(global A row ids, ...)``; ``root=r``: the other ranks return ``None``): SEMI / ANTI the ascending
    global A row ids; COUNT one int64 per A row; NEAREST ``(idx_b, distance)`` per A row with GLOBAL B
    row ids.
    """
    import torch
    import torch.distributed as dist

    if op not in ROW_OPS:
        raise ValueError(f"op must be one of {ROW_OPS}")
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    ca, sa, ea = (np.asarray(x) for x in a[:3])
    cb, sb, eb = (np.asarray(x) for x in b[:3])
    offs_a = tuple(a[3:5]) if len(a) >= 5 else (0, 0)
    offs_b = tuple(b[3:5]) if len(b) >= 5 else (0, 0)
    n_a = int(ca.shape[0])
    ia, ib = unit_rows(ca, cb, n_chrom, world, rank, split_side="a")
    res = local_op(op, ca[ia], sa[ia], ea[ia], offs_a, cb[ib], sb[ib], eb[ib], offs_b, n_chrom, **kw)
    dev = torch.device("cpu") if device is None else torch.device(device)
    ga = torch.from_numpy(ia.astype(np.int32)).to(dev)   # local A row -> global A row
    gb = torch.from_numpy(ib.astype(np.int32)).to(dev)
    if op in ("semi", "anti"):
        block = ga[res.to(dev).long()].view(1, -1)
    elif op == "count":
        block = torch.stack([ga, res.to(dev).to(torch.int32)])   # a count is at most the rows of B
    else:
        idx, dist_ = res
        idx = idx.to(dev).long()
        hit = idx >= 0
        gidx = torch.full(idx.shape, -1, dtype=torch.int32, device=dev)
        gidx[hit] = gb[idx[hit]]
        block = torch.cat([torch.stack([ga, gidx]), _i64_as_i32_rows(dist_.to(dev).to(torch.int64))])
    if not gather:
        if op == "nearest":
            return block[0], block[1], _i32_rows_as_i64(block[2:4])
        return tuple(block[k] for k in range(block.shape[0]))
    blocks = gather_blocks(block, group=group, impl=impl, root=root)
    if root is not None and rank != root:
        return None
    if op in ("semi", "anti"):
        return torch.sort(torch.cat([blk[0] for blk in blocks]))[0].to(torch.int32)
    if op == "count":
        out = torch.zeros(n_a, dtype=torch.int64, device=dev)
        for blk in blocks:
            out[blk[0].long()] = blk[1].to(torch.int64)
        return out
    out_i = torch.full((n_a,), -1, dtype=torch.int32, device=dev)
    out_d = torch.zeros(n_a, dtype=torch.int64, device=dev)
    for blk in blocks:
        out_i[blk[0].long()] = blk[1]
        out_d[blk[0].long()] = _i32_rows_as_i64(blk[2:4])
    return out_i, out_d


def hip_local_row_op(engine):
    """``local_op`` of :func:`sharded_row_op` backed by a :class:`giql_amd.engine.HipEngine`."""
    from .engine import DeviceSide

    def run(op, ca, sa, ea, offs_a, cb, sb, eb, offs_b, n_chrom, **kw):
        a = DeviceSide.from_numpy(ca, sa, ea, device=engine.device)
        a.start_off, a.end_off = offs_a
        b = DeviceSide.from_numpy(cb, sb, eb, device=engine.device)
        b.start_off, b.end_off = offs_b
        if op == "semi":
            return engine.semi_join(a, b, n_chrom)
        if op == "anti":
            return engine.anti_join(a, b, n_chrom)
        if op == "count":
            return engine.count_overlaps(a, b, n_chrom)
        return engine.nearest(a, b, n_chrom, **kw)

    return run
