"""World-size-2 (and 3) gloo tests of the N>1 path on CPU: sharding, id maps, the
padded all-gather of variable-length pair shards.  The local join is injected
(the oracle here; the HIP engine in production), everything else is the code
bench.py and giql_amd.distributed run on GPUs.
"""

import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tables(seed=3, skew=False):
    r = np.random.default_rng(seed)
    def side(n, nch):
        ch = r.integers(0, nch, n).astype(np.int32)
        if skew:  # one dominant chromosome (BASELINE config 2 is the limit case: a single chromosome)
            ch[r.random(n) < 0.8] = 2
        st = r.integers(0, 200_000, n).astype(np.int32)
        ln = r.integers(1, 900, n).astype(np.int32)
        return ch, st, st + ln
    return side(4000, 7), side(6000, 6)


def _worker(rank, world, port, out_dir, skew=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from giql_amd import distributed as D
        from oracle import pyoracle as ora

        def local_join(ca, sa, ea, offs_a, cb, sb, eb, offs_b, n_chrom):
            ra, rb = ora.c_inner(ora.Side(ca, sa, ea, *offs_a), ora.Side(cb, sb, eb, *offs_b), "sweep", threads=1)
            return torch.from_numpy(ra), torch.from_numpy(rb)

        a, b = _tables(skew=skew)
        if skew:  # the dominant chromosome must really be spread over the ranks
            ia, ib = D.unit_rows(a[0], b[0], 7, world, rank)
            assert 0 < int((a[0][ia] == 2).sum()) or 0 < int((b[0][ib] == 2).sum())
            assert len(ia) + len(ib) < 0.75 * (len(a[0]) + len(b[0]))
        out_a, out_b = D.sharded_inner_join(a, b, 7, local_join)
        np.save(os.path.join(out_dir, f"pairs_{rank}.npy"),
                np.stack([out_a.numpy(), out_b.numpy()]))
        # empty shards / zero-length gather
        z = torch.empty(0, dtype=torch.int32)
        ea_, eb_, counts = D.gather_pairs(z, z.clone())
        assert ea_.numel() == 0 and counts == [0] * world
        # ragged gather: rank r contributes r+1 pairs
        ra = torch.full((rank + 1,), rank, dtype=torch.int32)
        ga, gb, counts = D.gather_pairs(ra, ra + 10)
        assert counts == [r + 1 for r in range(world)]
        assert ga.tolist() == [r for r in range(world) for _ in range(r + 1)]
        assert gb.tolist() == [r + 10 for r in range(world) for _ in range(r + 1)]
        # the buffered, zero-copy form bench.py uses: two steps with different sizes
        xg = D.PairGather("cpu")
        for step_no in (1, 3):
            n = (rank + 1) * step_no
            counts = xg.counts(n)
            assert counts == [(r + 1) * step_no for r in range(world)]
            send = xg.send_block(max(counts))
            send[0, :n] = rank
            send[1, :n] = rank + 100 * step_no
            blocks = xg.all_gather(counts)
            for r, (ba, bb) in enumerate(blocks):
                assert ba.tolist() == [r] * counts[r] and bb.tolist() == [r + 100 * step_no] * counts[r]
        assert xg.counts(0) == [0] * world
        assert [tuple(x.numel() for x in blk) for blk in xg.all_gather([0] * world)] == [(0, 0)] * world
    finally:
        dist.destroy_process_group()


def _uniform_tables(seed=11, skew=False):
    """A: free lengths; B: fixed-length reads (the shape whose plan has a compact form)."""
    r = np.random.default_rng(seed)
    def chroms(n, nch):
        ch = r.integers(0, nch, n).astype(np.int32)
        if skew:
            ch[r.random(n) < 0.8] = 2
        return ch
    ca, cb = chroms(3000, 7), chroms(9000, 6)
    sa = r.integers(0, 150_000, 3000).astype(np.int32)
    sb = r.integers(0, 150_000, 9000).astype(np.int32)
    return (ca, sa, sa + r.integers(1, 900, 3000).astype(np.int32)), (cb, sb, sb + np.int32(150))


def _np_plan(ca, sa, ea, offs_a, cb, sb, eb, offs_b, n_chrom):
    """numpy stand-in for giql_hip_inner_plan_dev + giql_hip_inner_plan_export_dev on a shard whose
    B rows all have one length L: b overlaps a  <=>  b.start in [a.start - L + 1, a.end)."""
    if len(cb) == 0 or len(ca) == 0:
        z = torch.zeros(0, dtype=torch.int32)
        return True, z, z.clone(), z.clone(), z.clone(), 0
    lens = np.unique(eb.astype(np.int64) - sb)
    if lens.shape[0] != 1 or lens[0] <= 0:
        return None
    big = np.int64(1) << 33
    kb = cb.astype(np.int64) * big + sb
    order = np.argsort(kb, kind="stable")
    ks = kb[order]
    lo = np.searchsorted(ks, ca.astype(np.int64) * big + sa - lens[0] + 1, "left")
    hi = np.searchsorted(ks, ca.astype(np.int64) * big + ea, "left")
    cnt = np.maximum(hi - lo, 0)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.int32))
    return True, t(np.arange(len(ca))), t(lo), t(cnt), t(order), int(cnt.sum())


def _np_expand(q_rid, lo, cnt, s_rid, n_pairs):
    """numpy stand-in for giql_hip_fill_from_plan_dev."""
    c = cnt.numpy().astype(np.int64)
    row_q = np.repeat(q_rid.numpy(), c)
    start = np.repeat(lo.numpy().astype(np.int64), c)
    within = np.arange(int(c.sum())) - np.repeat(np.cumsum(c) - c, c)
    row_s = s_rid.numpy()[start + within]
    assert row_q.shape[0] == n_pairs
    return torch.from_numpy(row_q.astype(np.int32)), torch.from_numpy(row_s.astype(np.int32))


def _compact_worker(rank, world, port, out_dir, skew=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from giql_amd import distributed as D

        a, b = _uniform_tables(skew=skew)
        res = D.sharded_inner_join_compact(a, b, 7, _np_plan, _np_expand)
        assert res is not None
        np.save(os.path.join(out_dir, f"cpairs_{rank}.npy"), np.stack([res[0].numpy(), res[1].numpy()]))
        # the same exchange as grouped direct send / recv, and gathered to ONE rank
        for tag, impl, root in (("p2p", "p2p", None), ("root0", "p2p", 0), ("rootlast", "allgather", world - 1)):
            res = D.sharded_inner_join_compact(a, b, 7, _np_plan, _np_expand, impl=impl, root=root)
            assert res is not None
            np.save(os.path.join(out_dir, f"cpairs_{tag}_{rank}.npy"), np.stack([res[0].numpy(), res[1].numpy()]))
        # a shard without a compact form on ONE rank makes every rank fall back, collectively
        def plan_or_none(*args):
            return None if rank == world - 1 else _np_plan(*args)
        assert D.sharded_inner_join_compact(a, b, 7, plan_or_none, _np_expand) is None
        # the buffered exchange itself: ragged blocks, two steps with growing sizes, an empty rank
        xg = D.PlanGather("cpu")
        for step_no in (1, 4):
            nq, ns = (rank + 1) * step_no, (0 if rank == 1 else 5 * step_no + rank)
            sizes = xg.sizes(nq * 2, nq, ns, rank % 2 == 0)
            assert sizes == [((r + 1) * step_no * 2, (r + 1) * step_no, 0 if r == 1 else 5 * step_no + r, int(r % 2 == 0))
                             for r in range(world)]
            assert D.PlanGather.compact(sizes)
            q, lo, cnt, srid = xg.send_views(sizes)
            q.fill_(rank); lo.fill_(10 + rank); cnt.fill_(2); srid.fill_(100 + rank)
            for r, (q_r, lo_r, cnt_r, s_r) in enumerate(xg.all_gather(sizes)):
                assert q_r.tolist() == [r] * sizes[r][1] and lo_r.tolist() == [10 + r] * sizes[r][1]
                assert cnt_r.tolist() == [2] * sizes[r][1] and s_r.tolist() == [100 + r] * sizes[r][2]
        for impl, root in (("p2p", None), ("p2p", 1), ("allgather", 0)):
            xp = D.PlanGather("cpu", impl=impl, root=root)
            assert xp.impl == "p2p"   # a gather to one rank is point-to-point whatever was asked
            for step_no in (1, 4):
                nq, ns = (rank + 1) * step_no, (0 if rank == 1 else 5 * step_no + rank)
                sizes = xp.sizes(nq * 2, nq, ns, rank % 2 == 0)
                q, lo, cnt, srid = xp.send_views(sizes)
                q.fill_(rank); lo.fill_(10 + rank); cnt.fill_(2); srid.fill_(100 + rank)
                blocks = xp.all_gather(sizes)
                for r, blk in enumerate(blocks):
                    if root is not None and rank != root and r != rank:
                        assert blk is None
                        continue
                    q_r, lo_r, cnt_r, s_r = blk
                    assert q_r.tolist() == [r] * sizes[r][1] and lo_r.tolist() == [10 + r] * sizes[r][1]
                    assert cnt_r.tolist() == [2] * sizes[r][1] and s_r.tolist() == [100 + r] * sizes[r][2]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,skew", [(2, False), (3, False), (2, True), (3, True)])
def test_compact_plan_exchange_matches_single_process(tmp_path, world, skew):
    """The compact exchange (plan blocks all-gathered, expanded by every receiver) returns the same
    global pair set on every rank as one process joining everything -- bit-exact ids."""
    from oracle import pyoracle as ora

    port = _free_port()
    mp.spawn(_compact_worker, args=(world, port, str(tmp_path), skew), nprocs=world, join=True)
    a, b = _uniform_tables(skew=skew)
    want = ora.sort_pairs(*ora.c_inner(ora.Side(*a), ora.Side(*b), "sweep", threads=2))
    assert want.shape[0] > 1000
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"cpairs_{r}.npy"))
        assert np.array_equal(ora.sort_pairs(got[0], got[1]), want), r
        got = np.load(os.path.join(str(tmp_path), f"cpairs_p2p_{r}.npy"))
        assert np.array_equal(ora.sort_pairs(got[0], got[1]), want), ("p2p", r)
    # gathered to one rank: that rank holds everything, the others their own pairs -- which together are the result
    for tag, root in (("root0", 0), ("rootlast", world - 1)):
        parts = [np.load(os.path.join(str(tmp_path), f"cpairs_{tag}_{r}.npy")) for r in range(world)]
        assert np.array_equal(ora.sort_pairs(parts[root][0], parts[root][1]), want), tag
        own = [p for r, p in enumerate(parts) if r != root]
        assert all(p.shape[1] < want.shape[0] for p in own)
        mine_root = want.shape[0] - sum(p.shape[1] for p in own)
        assert mine_root >= 0


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_join_matches_single_process(tmp_path, world):
    from oracle import pyoracle as ora

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = _tables()
    want = ora.sort_pairs(*ora.c_inner(ora.Side(*a), ora.Side(*b), "sweep", threads=2))
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"pairs_{r}.npy"))
        assert np.array_equal(ora.sort_pairs(got[0], got[1]), want), r


@pytest.mark.parametrize("world", [2, 3])
def test_dominant_chromosome_is_split_by_row_ranges(tmp_path, world):
    from oracle import pyoracle as ora

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), True), nprocs=world, join=True)
    a, b = _tables(skew=True)
    want = ora.sort_pairs(*ora.c_inner(ora.Side(*a), ora.Side(*b), "sweep", threads=2))
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"pairs_{r}.npy"))
        assert np.array_equal(ora.sort_pairs(got[0], got[1]), want), r


def test_plan_units_splits_only_dominant_chromosomes():
    from giql_amd import distributed as D
    from giql_amd import shard, synth

    na = synth.rows_per_chrom(10_000_000, 5).tolist()
    nb = synth.rows_per_chrom(100_000_000, 6).tolist()
    for n in (1, 2, 4, 8):  # hg38: no chromosome exceeds one rank's share up to 8 ranks
        assert all(u[2] == 1 for u in shard.plan_units(na, nb, n))
    # single chromosome (config 2): split the larger side into one slice per rank
    units = shard.plan_units([1000], [4000], 4)
    assert [(u[0], u[1], u[2], u[3]) for u in units] == [(0, j, 4, "b") for j in range(4)]
    assert sorted(shard.assign_units(units, 4)) == [0, 1, 2, 3]
    # every row of the split side lands on exactly one rank; the other side is replicated
    ca = np.zeros(1000, np.int32)
    cb = np.zeros(4000, np.int32)
    seen_b = np.zeros(4000, int)
    for r in range(4):
        ia, ib = D.unit_rows(ca, cb, 1, 4, r)
        assert len(ia) == 1000
        seen_b[ib] += 1
    assert (seen_b == 1).all()
    # mixed: one heavy chromosome among light ones; empty chromosomes make no unit
    units = shard.plan_units([100, 10, 0, 10], [900, 10, 0, 10], 2)
    assert [u[0] for u in units if u[2] > 1] == [0, 0] and all(u[0] != 2 for u in units)
    seen = np.zeros(4, int)
    chrom_b = np.array([0, 0, 1, 3, 0, 0], np.int32)
    chrom_a = np.array([0, 1, 3], np.int32)
    for r in range(2):
        ia, ib = D.unit_rows(chrom_a, chrom_b, 4, 2, r)
        seen[np.isin(np.arange(6), ib)[[0, 1, 4, 5]]] += 1
    assert (seen == 1).all()


def test_lpt_assign_and_shard_rows():
    from giql_amd import distributed as D
    from giql_amd import shard, synth

    w = (synth.rows_per_chrom(10_000_000, 5) + synth.rows_per_chrom(100_000_000, 6)).tolist()
    for n in (1, 2, 4, 8):
        assign = shard.lpt_assign(w, n)
        load = [sum(w[c] for c in range(24) if assign[c] == r) for r in range(n)]
        assert max(load) <= 1.06 * (sum(w) / n)  # LPT balances 24 hg38 chroms within 6 %
        assert sorted(set(assign)) == list(range(n))
    chrom = np.array([0, 3, 1, 3, 2, 0], np.int32)
    assert D.shard_rows(chrom, [0, 1, 0, 1], 1).tolist() == [1, 2, 3]
    assert shard.span_groups([2**31, 2**31, 5]) == [[0, 2], [1]] or shard.span_groups([2**31, 2**31, 5]) == [[0], [1, 2]]
    with pytest.raises(ValueError):
        shard.lpt_assign([1.0], 0)


# ------------------------------------------------------------------ the per-row operators, sharded
def _oracle_row_op(op, ca, sa, ea, offs_a, cb, sb, eb, offs_b, n_chrom, **kw):
    from oracle import pyoracle as ora

    a, b = ora.Side(ca, sa, ea, *offs_a), ora.Side(cb, sb, eb, *offs_b)
    if op in ("semi", "anti"):
        return torch.from_numpy(ora.c_semi_anti(a, b, op == "anti", threads=1))
    if op == "count":
        return torch.from_numpy(ora.c_count(a, b, "sweep", threads=1))
    idx, d = ora.c_nearest_k1(a, b, signed=kw.get("signed", False), threads=1)
    return torch.from_numpy(idx), torch.from_numpy(d)


def _row_worker(rank, world, port, out_dir, skew=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from giql_amd import distributed as D

        b, a = _tables(seed=21, skew=skew)   # A = the 6000-row table: with skew, chromosome 2 is A-dominant
        if skew:  # chromosome 2 holds 80 % of A: its A rows must really be spread, each part with all of its B rows
            ia, ib = D.unit_rows(a[0], b[0], 7, world, rank, split_side="a")
            assert int((a[0][ia] == 2).sum()) < int((a[0] == 2).sum())
            assert int((b[0][ib] == 2).sum()) in (0, int((b[0] == 2).sum()))
        out = {}
        for op in ("semi", "anti", "count"):
            out[op] = D.sharded_row_op(op, a, b, 7, _oracle_row_op).numpy()
        idx, d = D.sharded_row_op("nearest", a, b, 7, _oracle_row_op, signed=True)
        out["nearest_idx"], out["nearest_d"] = idx.numpy(), d.numpy()
        # grouped send / recv instead of the padded all-gather; and gathered to rank 0 only
        for op in ("semi", "count"):
            out[op + "_p2p"] = D.sharded_row_op(op, a, b, 7, _oracle_row_op, impl="p2p").numpy()
        r0 = D.sharded_row_op("anti", a, b, 7, _oracle_row_op, root=0)
        assert (r0 is None) == (rank != 0)
        rn = D.sharded_row_op("nearest", a, b, 7, _oracle_row_op, signed=True, root=0, impl="p2p")
        assert (rn is None) == (rank != 0)
        if rank == 0:
            out["anti_root"], out["nearest_idx_root"], out["nearest_d_root"] = r0.numpy(), rn[0].numpy(), rn[1].numpy()
        assert out["count"].dtype == np.int64 and out["nearest_d"].dtype == np.int64 and out["semi"].dtype == np.int32
        np.savez(os.path.join(out_dir, f"rows_{rank}.npz"), **out)
        # ragged blocks, an empty rank
        blk = torch.full((2, 0 if rank == 1 else rank + 2), rank, dtype=torch.int64)
        got = D.gather_blocks(blk)
        assert [tuple(g.shape) for g in got] == [(2, 0 if r == 1 else r + 2) for r in range(world)]
        assert all(bool((g == r).all()) for r, g in enumerate(got))
        assert [tuple(g.shape) for g in D.gather_blocks(torch.zeros((3, 0), dtype=torch.int64))] == [(3, 0)] * world
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,skew", [(2, False), (3, False), (2, True), (3, True)])
def test_sharded_row_ops_match_single_process(tmp_path, world, skew):
    """SEMI / ANTI / COUNT / NEAREST sharded by the A rows' chromosomes (a dominant one cut by row ranges
    of A), results gathered on every rank = the single-process oracle's, global row ids included."""
    from oracle import pyoracle as ora

    port = _free_port()
    mp.spawn(_row_worker, args=(world, port, str(tmp_path), skew), nprocs=world, join=True)
    b, a = _tables(seed=21, skew=skew)
    sa, sb = ora.Side(*a), ora.Side(*b)
    wi, wd = ora.c_nearest_k1(sa, sb, signed=True)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rows_{rank}.npz"))
        assert np.array_equal(got["semi"], np.sort(ora.c_semi_anti(sa, sb, False)))
        assert np.array_equal(got["anti"], np.sort(ora.c_semi_anti(sa, sb, True)))
        assert np.array_equal(got["count"], ora.c_count(sa, sb))
        assert np.array_equal(got["nearest_d"], wd) and np.array_equal(got["nearest_idx"] >= 0, wi >= 0)
        hit = wi >= 0   # ids are tie-ambiguous: compare the matched rows' coordinates
        gi = got["nearest_idx"]
        assert np.array_equal(b[1][gi[hit]], b[1][wi[hit]]) and np.array_equal(b[2][gi[hit]], b[2][wi[hit]])
        assert np.array_equal(b[0][gi[hit]], a[0][hit])
        assert np.array_equal(got["semi_p2p"], got["semi"]) and np.array_equal(got["count_p2p"], got["count"])
        if rank == 0:
            assert np.array_equal(got["anti_root"], got["anti"])
            assert np.array_equal(got["nearest_d_root"], wd) and np.array_equal(got["nearest_idx_root"], got["nearest_idx"])


def test_plan_units_forced_side_never_cuts_b():
    from giql_amd import shard

    # A-dominant chromosome: cut along A; B-dominant one: kept whole (cutting A would only replicate B)
    units = shard.plan_units([1000, 10, 5], [100, 1000, 5], 2, 1.0, split_side="a")
    assert [(c, k, side) for c, _j, k, side, _w in units] == [(0, 2, "a"), (0, 2, "a"), (1, 1, "a"), (2, 1, "a")]
    with pytest.raises(ValueError):
        shard.plan_units([1], [1], 2, 1.0, split_side="x")
