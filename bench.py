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
    p.add_argument("--no-gather", action="store_true",
                   help="N>1: skip the final RCCL gather of the pairs (compute-only scaling)")
    p.add_argument("--cpu-sample-chroms", default="18,19,20,21",
                   help="chromosome ids of the bounded cpu_baseline sample")
    return p.parse_args()


def phase_bytes(phase: str, n_a: int, n_b: int, n_out: int) -> float:
    """Algorithmic (minimal) HBM bytes moved by ALL launches of one phase per step.

    DESIGN.md §kernels states each figure: read every input once, write every
    output once, nothing else.
    """
    n = n_a + n_b
    return {
        "span": 8.0 * n,                  # chrom + (start|end) once
        "linearize": 12.0 * n + 8.0 * n,  # read 3 cols, write key + end
        "sort_hist": 4.0 * n * 4,         # 4 passes x key
        "sort_scan": 0.0,
        "sort_scatter": 24.0 * n * 4,     # 4 passes x (read + write key,end,rid)
        "count": 8.0 * n + 8.0 * n,       # read (start,end) of each query, write lo + cnt
        "scan": 4.0 * n + 8.0 * n,        # read cnt, write u64 offset
        "partition": 0.0,
        "fill": 8.0 * n_out + 4.0 * n,    # write (row_a,row_b); read each rid once
        "irregular": 0.0,
        "aux": 0.0,
    }[phase]


def main() -> None:
    args = parse_args()
    import torch
    import torch.distributed as dist

    from giql_amd import shard, synth
    from giql_amd.engine import DeviceSide, HipEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

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

    eng = HipEngine(local_rank)
    out_cap = 0
    out = None
    count_t = torch.zeros(1, dtype=torch.int64, device=dev)
    gathered = None

    def step():
        """One pass of the hot path; returns this rank's pair count."""
        nonlocal out, out_cap, gathered
        n = eng.inner_plan(a, b, n_chrom)
        if n > out_cap:
            out = None
            out_cap = int(n * 1.05) + 1024
            out = torch.empty((2, out_cap), dtype=torch.int32, device=dev)
        eng.inner_fill(out[0, :n], out[1, :n])
        if distributed and not args.no_gather:
            # the path's one exchange step: counts, then the index pairs (padded)
            count_t.fill_(n)
            counts = torch.empty(world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(counts, count_t)
            m = int(counts.max().item())
            if out_cap < m:  # keep the send buffer at least as large as the pad
                bigger = torch.empty((2, int(m * 1.05) + 1024), dtype=torch.int32, device=dev)
                bigger[:, :n] = out[:, :n]
                out, out_cap = bigger, bigger.shape[1]
            send = out[:, :m].contiguous()
            if gathered is None or gathered.shape[0] < world * 2 * m:
                gathered = torch.empty(world * 2 * m, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(gathered[: world * 2 * m], send.view(-1))
        return n

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    # hipEvent phase timing stays ON inside the timed region (two events per
    # phase on the launch stream); stats() waits for the step's last event.
    eng.set_profiling(True)
    phase_ms = {}
    phase_launches = {}
    sync_all()
    t0 = time.perf_counter()
    n_local = 0
    for _ in range(args.steps):
        n_local = step()
        st = eng.stats()
        for k, v in st["phase_ms"].items():
            phase_ms[k] = phase_ms.get(k, 0.0) + v
        for k, v in st["phase_launches"].items():
            phase_launches[k] = phase_launches.get(k, 0) + v
    sync_all()
    elapsed = time.perf_counter() - t0
    eng.set_profiling(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([n_local, loc_na, loc_nb], dtype=torch.int64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    n_pairs, tot_na, tot_nb = (int(x) for x in tot.tolist())

    # ---- rank 0: roofline of the dominant kernel + CPU baseline + the JSON line
    if rank == 0:
        per_step_ms = {k: v / args.steps for k, v in phase_ms.items()}
        per_step_launches = {k: v // args.steps for k, v in phase_launches.items()}
        dom = max(per_step_ms, key=lambda k: per_step_ms[k])
        dom_ms = per_step_ms[dom]
        dom_launches = max(per_step_launches[dom], 1)
        dom_bytes = phase_bytes(dom, loc_na, loc_nb, n_local)
        achieved = (dom_bytes / dom_launches) / (dom_ms / dom_launches * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        join_bytes = 12.0 * (loc_na + loc_nb) + 8.0 * n_local
        device_ms = sum(per_step_ms.values())
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
            "traffic": None,
            "whole_join": {
                "algorithmic_bytes": join_bytes,
                "device_ms": round(device_ms, 3),
                "achieved": round(join_bytes / (device_ms * 1e-3) / 1e9, 1) if device_ms > 0 else 0.0,
                "frac": round(join_bytes / (device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if device_ms > 0 else 0.0,
            },
            "phase_ms": {k: round(v, 3) for k, v in per_step_ms.items() if v > 0},
        }

        cpu_baseline = None
        if not args.no_cpu_baseline:
            from oracle import pyoracle as ora

            sample = [int(c) for c in args.cpu_sample_chroms.split(",") if c != ""]
            sa = synth.make_table(n_a, seed_a, kind_a, chroms=sample)
            sb = synth.make_table(n_b, seed_b, kind_b, chroms=sample)
            threads = ora.max_threads()
            oa, ob = ora.Side(*sa), ora.Side(*sb)
            t1 = time.perf_counter()
            ra, rb = ora.c_inner(oa, ob, "sweep", threads=threads)
            cpu_s = time.perf_counter() - t1
            cpu_baseline = {
                "value": round(ra.shape[0] / cpu_s, 1),
                "unit": "pairs/s",
                "cores": threads,
                "kind": "port",
                "sample": (f"chromosome ids {sample} of the same workload: {oa.n} x {ob.n} rows -> "
                           f"{ra.shape[0]} pairs in {cpu_s:.2f} s (oracle OpenMP sort-merge port, "
                           "not DuckDB: duckdb is not installed on the box)"),
                "host_cpu_count": os.cpu_count(),
            }

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
                "parallelism": f"chrom-shard x{world}" + ("" if world == 1 or args.no_gather else " + rccl all-gather of pairs"),
                "inputs": "resident in HBM before the timed region",
            },
            "hbm_algorithmic_GBps": round((12.0 * (tot_na + tot_nb) + 8.0 * n_pairs) * args.steps / elapsed / 1e9, 1),
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "gen_seconds": round(gen_s, 1),
        }
        print(json.dumps(line), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
