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


def _tables(seed=3):
    r = np.random.default_rng(seed)
    def side(n, nch):
        ch = r.integers(0, nch, n).astype(np.int32)
        st = r.integers(0, 200_000, n).astype(np.int32)
        ln = r.integers(1, 900, n).astype(np.int32)
        return ch, st, st + ln
    return side(4000, 7), side(6000, 6)


def _worker(rank, world, port, out_dir):
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

        a, b = _tables()
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
    finally:
        dist.destroy_process_group()


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
