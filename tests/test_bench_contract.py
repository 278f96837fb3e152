"""bench.py's byte accounting and launch plumbing -- CPU only (no GPU call is made here)."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_sort_bytes_model_by_hand():
    # cfg 4, fixed-length B sorted from its raw columns, three-stage sort for the 100M-row side:
    # B: KEYGEN pass 8 + 8 B/row, second pass 8 + 8, bucket sort 8 + 8; A (key, end, rid), four passes:
    # 8 in + 12 out, then 3 x (12 + 12)
    sc, lo = bench.sort_bytes_model([(10_000_000, 2, False), (100_000_000, 1, True)])
    assert sc == 2 * 16 * 100_000_000 + (20 + 3 * 24) * 10_000_000 == 4_120_000_000
    assert lo == 16 * 100_000_000
    # below 32M rows: four passes, first one reads no row ids; keys only: 8 B/row/pass
    sc, lo = bench.sort_bytes_model([(10_000_000, 0, False), (1_000_000, 2, False)])
    assert sc == 4 * 8 * 10_000_000 + (20 + 3 * 24) * 1_000_000 and lo == 0


def test_phase_bytes_agree_with_the_committed_pmc_run():
    """the bench's algorithmic bytes must follow the code -- within 10 % of what the
    counters saw for the sort, and never above the measured traffic for any phase."""
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    pmc = t["cfg4_10Mx100M_24chrom"]["uniform_b"]
    # (round 3: the settled form of the one-call join writes its pairs from the bucket stage -- no fill phase; the
    # query side is sorted from its raw columns in two passes: 12 + 12 B/row, then 12 + 12)
    alg = bench.inner_phase_bytes(10_000_000, 100_000_000, 404_376_266, "uniform_b", True, bucket_join=True)
    alg["sort_scatter"] = 2 * 16 * 100_000_000 + 2 * 24 * 10_000_000
    assert abs(alg["sort_scatter"] - pmc["sort_scatter"]) / pmc["sort_scatter"] < 0.15
    assert abs(alg["sort_local"] - pmc["sort_local"]) / pmc["sort_local"] < 0.10
    assert alg["span"] <= pmc["span"] * 1.02
    assert "fill" not in pmc and alg["fill"] == 0
    assert "csrc_hash" in t and "commit" in t


def test_pmc_traffic_is_withheld_when_the_kernel_sources_changed(monkeypatch):
    import json as _json

    stamped = _json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    # collected on exactly these sources: the figure is handed out, per launch; on any other sources: withheld
    monkeypatch.setattr(bench, "csrc_hash", lambda: stamped["csrc_hash"])
    per_launch = bench.pmc_traffic("cfg4_10Mx100M_24chrom", "uniform_b", "sort_scatter", 4)
    assert per_launch == round(stamped["cfg4_10Mx100M_24chrom"]["uniform_b"]["sort_scatter"] / 4)
    assert bench.pmc_traffic("cfg4_10Mx100M_24chrom", "no_such_form", "sort_scatter", 4) is None
    monkeypatch.setattr(bench, "csrc_hash", lambda: "0" * 16)
    assert bench.pmc_traffic("cfg4_10Mx100M_24chrom", "uniform_b", "sort_scatter", 4) is None


def test_operator_bytes_follow_survey_8d():
    assert bench.op_bytes("inner", 10, 20, 5) == 12 * 30 + 8 * 5
    assert bench.op_bytes("semi", 10, 20, 5) == 12 * 30 + 4 * 5
    assert bench.op_bytes("count", 10, 20, 0) == 12 * 30 + 8 * 10
    assert bench.op_bytes("nearest", 10, 20, 0) == 12 * 30 + 8 * 10


def test_gpus_n_starts_the_ranks_as_a_child_process(monkeypatch):
    """`python bench.py --gpus N` outside torch.distributed.run must launch the ranks itself (and not
    exec): the command line, and that a rank (WORLD_SIZE set) does not launch again."""
    calls = []

    class Done(Exception):
        pass

    def fake_run(cmd, env=None):
        calls.append((cmd, env))

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    args = bench.parse_args()
    with pytest.raises(SystemExit) as ei:
        bench.self_launch(args)
    assert ei.value.code == 7
    cmd, env = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert "127.0.0.1" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("WORLD_SIZE", "4")
    bench.self_launch(args)          # a rank: returns without launching
    assert len(calls) == 1
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench.self_launch(bench.parse_args())   # N = 1: nothing to launch
    assert len(calls) == 1
