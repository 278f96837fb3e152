#!/usr/bin/env python3
"""bench.py -- headline benchmark of the INTERSECTS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one full pass of the hot path (plan: linearise + sort + count + scan;
fill: pair materialisation) over one batch of synthetic input already resident in
HBM.  Workload (BASELINE.json configs[3], SURVEY.md §8(d) cfg 4): 10M peaks x
100M reads, 24 chromosomes with hg38 lengths, int32 columns, unsorted rows.

N > 1: launched as ``python -m torch.distributed.run --nproc-per-node N ...``;
chromosomes are LPT-packed onto the ranks (giql_amd.shard), every rank joins its
own chromosomes, then the pair counts are all-gathered and the index pairs are
gathered with one RCCL all-gather (padded to the largest shard).  Total work is
fixed as N grows ("strong" scaling).

Prints ONE JSON line on rank 0 (see the task contract): metric / value / unit,
plus ``roofline`` for the dominant kernel (algorithmic bytes per launch over the
hipEvent-measured average launch time) and ``cpu_baseline`` (the oracle's
OpenMP sort-merge port timed on the host cores, bounded sample).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (n_a, kind_a, seed_a, n_b, kind_b, seed_b)
    "cfg4_10Mx100M_24chrom": (10_000_000, "peaks", 5, 100_000_000, "reads", 6),
    "cfg4_small_1Mx10M_24chrom": (1_000_000, "peaks", 5, 10_000_000, "reads", 6),
}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--workload", default="cfg4_10Mx100M_24chrom", choices=sorted(WORKLOADS))
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--force-exchange", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--no-gather", action="store_true",
                   help="N>1: skip the final RCCL gather of the pairs (compute-only scaling)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="N>1 exchange backend; 'gloo' is a rehearsal mode for boxes with fewer "
                        "GPUs than ranks (ranks share GPUs, pairs are exchanged through host memory)")
    p.add_argument("--cpu-sample-chroms", default="18,19,20,21",
                   help="chromosome ids of the bounded cpu_baseline sample")
    return p.parse_args()


def phase_bytes(phase: str, n_a: int, n_b: int, n_out: int, form: str = "general", span_hist: bool = False) -> float:
    """Algorithmic (minimal) HBM bytes moved by ALL launches of one phase per step.

    DESIGN.md §3 states each figure: read every input of the kernel once, write
    every output once, nothing else.  In the uniform-length forms the fixed-length
    side is sorted as (key, rid) -- 8 B/row -- and there is no class-1 stage.  With
    ``span_hist`` (the digit histogram taken in the span pass) that side has no linearize pass
    and its first sort pass reads (chrom, start) -- 8 B/row -- instead of the 4-byte key.
    """
    n = n_a + n_b
    if form == "uniform_b":
        n_q, n_u = n_a, n_b
    elif form == "uniform_a":
        n_q, n_u = n_b, n_a
    else:
        n_q = n_u = 0
    if form == "general":
        sort = 24.0 * n * 4                      # 4 passes x (read + write key,end,rid)
        lin = 12.0 * n + 8.0 * n                 # read 3 cols, write key + end
        count = (8.0 * n_b + 4.0 * n_a) + (8.0 * n_a + 4.0 * n_b + 8.0 * n_a)
        scan = 12.0 * n_a
        fill = 8.0 * n_out + 12.0 * n_b + 4.0 * n_a + 16.0 * n_a + 4.0 * n_b
    else:
        sort = (24.0 * n_q + 16.0 * n_u) * 4     # the uniform side carries (key, rid)
        lin = 12.0 * n + 8.0 * n_q + 4.0 * n_u
        if span_hist:
            sort += 4.0 * n_u
            lin = 12.0 * n_q + 8.0 * n_q
        count = 8.0 * n_q + 4.0 * n_u + 8.0 * n_q
        scan = 12.0 * n_q
        fill = 8.0 * n_out + 16.0 * n_q + 4.0 * n_u
    return {
        "span": 12.0 * n,       # chrom, start, end once
        "linearize": lin,
        "sort_hist": 4.0 * n * 4,
        "sort_scan": 0.0,
        "sort_scatter": sort,
        "count": count,
        "scan": scan,
        "partition": 0.0,
        "fill": fill,
        "irregular": 0.0,
        "aux": 0.0,
    }[phase]


def pmc_traffic(workload: str, form: str, phase: str, launches: int):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/pmc_traffic.json; rocprofv3 cannot run inside the timed process).
    None when no counter run exists for this workload / join form."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        return round(t[workload][form][phase] / max(launches, 1))
    except (OSError, KeyError, ValueError):
        return None


def run_cpu_baseline(args, n_a, kind_a, seed_a, n_b, kind_b, seed_b):
    """Time the oracle's OpenMP sort-merge port on a bounded sample of the workload.

    A probe on four small chromosomes estimates the rate; the reported sample is
    the largest prefix of chromosomes (by id) whose estimated time stays under
    ~20 s -- the whole workload when the host is fast enough.
    """
    from giql_amd import synth
    from oracle import pyoracle as ora

    threads = ora.max_threads()

    n_chrom_all = len(synth.HG38_LENGTHS)
    checksum = [None]

    def run(chroms):
        sel = None if len(chroms) == n_chrom_all else chroms  # None = the GPU run's row order
        sa = synth.make_table(n_a, seed_a, kind_a, chroms=sel)
        sb = synth.make_table(n_b, seed_b, kind_b, chroms=sel)
        oa, ob = ora.Side(*sa), ora.Side(*sb)
        t1 = time.perf_counter()
        ra, rb = ora.c_inner(oa, ob, "sweep", threads=threads)
        dt = time.perf_counter() - t1
        # whole workload only: the same order-independent 64-bit checksum as
        # giql_hip_pairs_checksum_dev, for the parity line (outside the timing)
        checksum[0] = ora.c_pairs_checksum(ra, rb) if len(chroms) == n_chrom_all else None
        return oa.n, ob.n, int(ra.shape[0]), dt

    probe = [int(c) for c in args.cpu_sample_chroms.split(",") if c != ""]
    pa_, pb_, pp, pt = run(probe)
    rows_total = synth.rows_per_chrom(n_a, seed_a) + synth.rows_per_chrom(n_b, seed_b)
    rate = (pa_ + pb_) / max(pt, 1e-3)                    # rows/s on the probe
    budget_rows = rate * 20.0
    chroms, acc = [], 0
    for c in range(len(rows_total)):
        if acc + rows_total[c] > budget_rows and chroms:
            break
        chroms.append(c)
        acc += int(rows_total[c])
    sa_n, sb_n, sp, st_ = run(chroms)
    whole = len(chroms) == len(rows_total)
    return {
        "value": round(sp / st_, 1),
        "unit": "pairs/s",
        "cores": threads,
        "kind": "port",
        "sample": (("the whole workload" if whole else f"chromosome ids 0..{chroms[-1]} of the same workload")
                   + f": {sa_n} x {sb_n} rows -> {sp} pairs in {st_:.2f} s "
                   "(oracle OpenMP sort-merge port, not DuckDB: duckdb is not installed on the box)"),
        "host_cpu_count": os.cpu_count(),
        # popped by main() into the "parity" entry (parity check in the same run)
        "pairs": sp if whole else None,
        "pairs_checksum": checksum[0] if whole else None,
    }


def measured_copy_gbs(dev, mb: int = 1600, reps: int = 5):
    """Device-to-device copy bandwidth of this GPU in GB/s (bytes read + written)."""
    import torch

    try:
        n = mb * (1 << 20) // 4
        x = torch.empty(n, dtype=torch.int32, device=dev)
        y = torch.empty_like(x)
        x.fill_(1)
        y.copy_(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(reps):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps
        return round(2 * n * 4 / ms / 1e6, 1) if ms > 0 else None
    except RuntimeError:  # not enough free HBM next to the workload
        return None


def main() -> None:
    args = parse_args()
    import torch
    import torch.distributed as dist

    from giql_amd import shard, synth
    from giql_amd._lib import GIQL_ERR_CAPACITY, GiqlHipError
    from giql_amd.engine import DeviceSide, HipEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    # --force-exchange: run the N>1 exchange code (process group, counts, take into the send
    # block, all-gather) with a single rank -- the only way to drive it over RCCL on a 1-GPU box
    distributed = world > 1 or args.force_exchange
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    xdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the exchange happens
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    n_a, kind_a, seed_a, n_b, kind_b, seed_b = WORKLOADS[args.workload]
    n_chrom = len(synth.HG38_LENGTHS)

    # ---- shard: chromosomes -> ranks (LPT on expected rows); N=1 keeps all 24
    rows_a = synth.rows_per_chrom(n_a, seed_a)
    rows_b = synth.rows_per_chrom(n_b, seed_b)
    assign = shard.lpt_assign((rows_a + rows_b).tolist(), world)
    my_chroms = [c for c in range(n_chrom) if assign[c] == rank]

    t0 = time.time()
    ac, as_, ae = synth.make_table(n_a, seed_a, kind_a, chroms=None if world == 1 else my_chroms)
    bc, bs, be = synth.make_table(n_b, seed_b, kind_b, chroms=None if world == 1 else my_chroms)
    gen_s = time.time() - t0
    a = DeviceSide.from_numpy(ac, as_, ae, device=dev)
    b = DeviceSide.from_numpy(bc, bs, be, device=dev)
    loc_na, loc_nb = a.n, b.n

    eng = HipEngine(dev_index)
    out_cap = 0
    out = None

    def alloc_out(n):
        """Caller-owned output: row_a / row_b, 5 % head-room, each row a 2 MiB multiple long."""
        cap = (int(n * 1.05) + 1024 + (1 << 19) - 1) >> 19 << 19
        if os.environ.get("GIQL_BENCH_OUT", "split") == "joint":   # probe: both rows in ONE allocation
            return cap, torch.empty((2, cap), dtype=torch.int32, device=dev)
        return cap, (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev))
    # shard-local row index -> global row id: ranks own disjoint chromosome sets and the
    # global table is "rows of lower ranks first", so the map is one offset per side
    # (giql_amd.distributed.sharded_inner_join handles arbitrary row sets with the take kernel)
    base_a = base_b = 0
    if distributed:
        from giql_amd import distributed as D

        sizes = torch.tensor([loc_na, loc_nb], dtype=torch.int64, device=xdev)
        all_sizes = torch.empty((world, 2), dtype=torch.int64, device=xdev)
        dist.all_gather_into_tensor(all_sizes.view(-1), sizes)
        base_a, base_b = (int(x) for x in all_sizes[:rank].sum(0).tolist()) if rank else (0, 0)

    xg = D.PairGather(xdev) if distributed and not args.no_gather else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if distributed else None
    split_ms = [0.0, 0.0]  # [local join, exchange] summed over the timed steps

    phase_ms = {}
    phase_launches = {}
    last_stats = [None]
    last_pairs = [None]  # views of the last step's (row_a, row_b), for the parity line

    def step(collect=False):
        """One pass of the hot path; returns this rank's pair count."""
        nonlocal out, out_cap
        if ev:
            ev[0].record()
        if xg is not None and xdev == dev:
            # the path's one exchange step starts inside the join: the counts are all-gathered
            # after the plan, then the fill writes straight into the send block of the all-gather
            n = eng.inner_plan(a, b, n_chrom)
            counts = xg.counts(n)
            send = xg.send_block(max(counts))
            ra, rb = send[0, :n], send[1, :n]
            eng.inner_fill(ra, rb)
        elif out is not None and os.environ.get("GIQL_BENCH_NO_FUSE"):
            n = eng.inner_plan(a, b, n_chrom)
            ra, rb = out[0][:n], out[1][:n]
            eng.inner_fill(ra, rb)
        elif out is not None:
            # one C-ABI call into the buffers of the previous step (plan + fill, no stream sync
            # between them when the context's guesses hold); a larger result re-allocates
            try:
                n = eng.inner_join_into(a, b, n_chrom, out[0], out[1])
            except GiqlHipError as exc:
                if exc.code != GIQL_ERR_CAPACITY:
                    raise
                n = eng.last_pairs
                out = None
                out_cap, out = alloc_out(n)
                eng.inner_fill(out[0][:n], out[1][:n])
            ra, rb = out[0][:n], out[1][:n]
        else:
            n = eng.inner_plan(a, b, n_chrom)
            out_cap, out = alloc_out(n)
            ra, rb = out[0][:n], out[1][:n]
            eng.inner_fill(ra, rb)
        last_pairs[0] = (ra, rb)
        if ev:
            ev[1].record()
        if collect:  # the join's phase times, before anything else touches the engine
            st = eng.stats()
            last_stats[0] = st
            for k, v in st["phase_ms"].items():
                phase_ms[k] = phase_ms.get(k, 0.0) + v
            for k, v in st["phase_launches"].items():
                phase_launches[k] = phase_launches.get(k, 0) + v
        if xg is not None:
            if xdev == dev:
                if base_a:
                    ra.add_(base_a)  # local -> global row ids, in place in the send block
                if base_b:
                    rb.add_(base_b)
            else:  # gloo rehearsal: the exchange runs through host memory
                counts = xg.counts(n)
                send = xg.send_block(max(counts))
                send[0, :n] = (ra + base_a).to(xdev)
                send[1, :n] = (rb + base_b).to(xdev)
            xg.all_gather(counts)
        if ev:
            ev[2].record()
            ev[2].synchronize()
            split_ms[0] += ev[0].elapsed_time(ev[1])
            split_ms[1] += ev[1].elapsed_time(ev[2])
        return n

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Warm-up steps run with hipEvent pairs around EVERY phase: they give the phase table.
    # The timed steps keep events only around the dominant kernel's launches (the sort passes),
    # which is what the roofline object is computed from: every event pair costs the launch
    # stream a few microseconds of idle time, ~0.1 ms per step with all nine phases timed.
    eng.set_profiling(True)
    for _ in range(args.warmup):
        step(collect=True)
    # the LAST warm-up step's phases (the first one also pays for the arena's first touch)
    warm_phase_ms = dict(last_stats[0]["phase_ms"]) if args.warmup and last_stats[0] else {}
    light = bool(warm_phase_ms)
    phase_ms.clear()
    phase_launches.clear()
    eng.set_profiling(2 if light else True)
    sync_all()
    split_ms[0] = split_ms[1] = 0.0
    t0 = time.perf_counter()
    n_local = 0
    for _ in range(args.steps):
        n_local = step(collect=True)
    sync_all()
    st = last_stats[0]
    elapsed = time.perf_counter() - t0
    eng.set_profiling(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
    tot = torch.tensor([n_local, loc_na, loc_nb], dtype=torch.int64, device=xdev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    n_pairs, tot_na, tot_nb = (int(x) for x in tot.tolist())

    # ---- rank 0: roofline of the dominant kernel + CPU baseline + the JSON line
    if rank == 0:
        per_step_ms = {k: v / args.steps for k, v in phase_ms.items()}
        per_step_launches = {k: v // args.steps for k, v in phase_launches.items()}
        if light:  # the other phases were timed in the warm-up steps only
            per_step_ms = {**warm_phase_ms, "sort_scatter": per_step_ms.get("sort_scatter", 0.0)}
        # the sort passes are the path's dominant kernel; they are the phase timed in the timed steps
        dom = "sort_scatter" if light else max(per_step_ms, key=lambda k: per_step_ms[k])
        dom_ms = per_step_ms[dom]
        dom_launches = max(per_step_launches[dom], 1)
        dom_bytes = phase_bytes(dom, loc_na, loc_nb, n_local, st["join_form"], st.get("span_hist", False))
        achieved = (dom_bytes / dom_launches) / (dom_ms / dom_launches * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        join_bytes = 12.0 * (loc_na + loc_nb) + 8.0 * n_local
        device_ms = sum(per_step_ms.values())
        copy_gbs = measured_copy_gbs(dev)
        roofline = {
            "bound": "hbm",
            "kernel": dom,
            "launches_per_step": dom_launches,
            "avg_launch_ms": round(dom_ms / dom_launches, 4),
            "algorithmic_bytes_per_launch": dom_bytes / dom_launches,
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            # this box's own streaming-copy rate (read + written bytes of a 1.6 GB d2d copy,
            # SURVEY.md 8d) and the kernel's fraction of THAT; "frac" stays against the 8 TB/s peak
            "measured_copy": copy_gbs,
            "frac_of_measured_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
            # the PMC passes were collected on the whole workload on one GPU
            "traffic": pmc_traffic(args.workload, st["join_form"], dom, dom_launches) if world == 1 else None,
            "whole_join": {
                "algorithmic_bytes": join_bytes,
                "device_ms": round(device_ms, 3),
                "achieved": round(join_bytes / (device_ms * 1e-3) / 1e9, 1) if device_ms > 0 else 0.0,
                "frac": round(join_bytes / (device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if device_ms > 0 else 0.0,
            },
            "phase_ms": {k: round(v, 3) for k, v in per_step_ms.items() if v > 0},
            "phase_ms_source": ("sort_scatter: hipEvents in the timed steps; other phases: hipEvents in the "
                                "warm-up steps (the timed steps record events around the sort passes only)"
                                if light else "hipEvents in the timed steps"),
        }

        cpu_baseline = None
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only (the other ranks would idle in the barrier)
            cpu_baseline = run_cpu_baseline(args, n_a, kind_a, seed_a, n_b, kind_b, seed_b)
            cpu_sum = cpu_baseline.pop("pairs_checksum", None)
            cpu_pairs = cpu_baseline.pop("pairs", None)
            if world == 1 and cpu_sum is not None and last_pairs[0] is not None:
                gpu_sum = eng.pairs_checksum(*last_pairs[0])
                cpu_baseline["parity"] = {"pairs_equal": cpu_pairs == n_pairs, "multiset_checksum_equal": gpu_sum == cpu_sum,
                                          "checked": "all %d pairs of the last timed step" % n_pairs}

        value = n_pairs * args.steps / elapsed
        line = {
            "metric": "overlap-pairs/sec, 10Mx100M INTERSECTS inner join",
            "value": round(value, 1),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "n_a": tot_na, "n_b": tot_nb, "n_chrom": n_chrom, "pairs_per_step": n_pairs,
                "parallelism": f"chrom-shard x{world}" + ("" if world == 1 or args.no_gather else
                                                            (" + rccl all-gather of pairs" if args.backend == "nccl"
                                                             else " + gloo (rehearsal) all-gather of pairs")),
                "inputs": "resident in HBM before the timed region",
                "join_form": st["join_form"],
                "span_hist": bool(st.get("span_hist", False)),
            },
            "hbm_algorithmic_GBps": round((12.0 * (tot_na + tot_nb) + 8.0 * n_pairs) * args.steps / elapsed / 1e9, 1),
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "gen_seconds": round(gen_s, 1),
        }
        if distributed:
            # rank 0's view of where a step goes: its own shard's join vs the one
            # exchange (global ids + all-gather of every rank's pairs)
            line["rank0_local_join_ms"] = round(split_ms[0] / args.steps, 3)
            line["rank0_exchange_ms"] = round(split_ms[1] / args.steps, 3)
        print(json.dumps(line), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
