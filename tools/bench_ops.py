#!/usr/bin/env python3
"""Secondary benchmark: the other BASELINE.json configs on one GPU.

cfg2  1M x 1M INNER, single chromosome (sparse G=248,956,422 and dense G=10,000,000)
cfg3  SEMI and ANTI, 1M peaks x 10M reads, 24 chromosomes (+ COUNT on the same input)
cfg5  NEAREST k=1, 10M x 10M peaks, 24 chromosomes
Prints one JSON line per case: ms per call (median of --reps after --warmup), rows/s,
and the hipEvent phase breakdown.  Inputs are resident in HBM before timing.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch

    from giql_amd import synth
    from giql_amd.engine import DeviceSide, HipEngine

    eng = HipEngine(0)
    dev = "cuda:0"

    def side(cols):
        return DeviceSide.from_numpy(*cols, device=dev)

    def run(name, fn, units, unit_name):
        if args.only and args.only not in name:
            return
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        # wall time with profiling OFF (an event pair costs the stream a few microseconds of idle
        # time per phase: 0.1 ms and more on calls this short) ...
        times, phases = [], {}
        for _ in range(args.reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        # ... and the phase breakdown from separate profiled calls
        eng.set_profiling(True)
        for _ in range(2):
            fn()
            torch.cuda.synchronize()
            for k, v in eng.stats()["phase_ms"].items():
                phases[k] = phases.get(k, 0.0) + v / 2
        eng.set_profiling(False)
        ms = statistics.median(times)
        n_out = int(out) if not hasattr(out, "shape") else int(out.shape[0])
        print(json.dumps({"case": name, "ms": round(ms, 3), "n_out": n_out,
                          unit_name + "_per_s": round(units / (ms * 1e-3), 1),
                          "join_form": eng.stats().get("join_form"),
                          "phase_ms": {k: round(v, 3) for k, v in phases.items() if v > 0}}), flush=True)

    # ---- cfg 2
    for tag, g in (("sparse", 248_956_422), ("dense", 10_000_000)):
        a = side(synth.make_single_chrom(1_000_000, 1, "peaks", g))
        b = side(synth.make_single_chrom(1_000_000, 2, "peaks", g))
        n = eng.inner_plan(a, b, 1)
        out = torch.empty((2, n), dtype=torch.int32, device=dev)

        def inner():
            m = eng.inner_plan(a, b, 1)
            eng.inner_fill(out[0, :m], out[1, :m])
            return m

        run(f"cfg2_inner_1Mx1M_{tag}", inner, n, "pairs")
        del out
    # ---- projection: gather payload columns by the ~7e7 pairs of a 10M x 10M join
    if not args.only or "take" in args.only or "select" in args.only:
        ca = synth.make_table(10_000_000, 7, "peaks")
        cb = synth.make_table(10_000_000, 8, "peaks")
        a, b = side(ca), side(cb)
        ra, rb = eng.inner_join(a, b, 24)
        p = int(ra.shape[0])
        a_cols = [a.start, a.end]
        b_cols = [b.start, torch.arange(b.n, dtype=torch.int64, device=dev)]

        def take_fixed():
            eng.take(a_cols, ra)
            return eng.take(b_cols, rb)[0]

        # algorithmic bytes: 2 idx reads + (4+4) + (4+8) gathered + the same written
        run("take_fixed_4cols", take_fixed, p * (8 + 20 + 20) / 1e9, "GB")
        names = np.array([f"read_{i:08d}" for i in range(100_000)])
        import pyarrow as pa

        arr = pa.array(names[np.arange(b.n) % len(names)], pa.string())
        if isinstance(arr, pa.ChunkedArray):
            arr = arr.combine_chunks()
        off = torch.from_numpy(np.frombuffer(arr.buffers()[1], np.int32, count=len(arr) + 1).copy()).to(dev)
        data = torch.from_numpy(np.frombuffer(arr.buffers()[2], np.uint8).copy()).to(dev)
        run("take_utf8_13B", lambda: eng.take_utf8(off, data, rb)[0], p * (4 + 8 + 4 + 13 + 13) / 1e9, "GB")
        # residual predicates: same-strand filter over the pairs, int-compare filter over rows
        strand_a = torch.randint(0, 2, (a.n,), dtype=torch.int32, device=dev)
        strand_b = torch.randint(0, 2, (b.n,), dtype=torch.int32, device=dev)
        run("select_pairs_same_strand", lambda: eng.select([(("a", strand_a), "=", ("b", strand_b))], idx_a=ra, idx_b=rb,
                                                           n_rows_a=a.n, n_rows_b=b.n)[0], p * (8 + 8 + 8) / 1e9, "GB")
        run("select_rows_10M", lambda: eng.select([(("a", a.start), ">", ("lit", 1_000_000))], n=a.n, n_rows_a=a.n,
                                                  want=("a",))[0], a.n * 8 / 1e9, "GB")
        del ra, rb, off, data
    # ---- CLUSTER / MERGE: 10M peaks, 24 chromosomes
    if not args.only or "cluster" in args.only or "merge" in args.only:
        a = side(synth.make_table(10_000_000, 7, "peaks"))
        run("cluster_10M", lambda: eng.cluster(a, 24, 0), 10_000_000, "input_rows")
        run("merge_10M", lambda: eng.merge(a, 24, 0)[0], 10_000_000, "input_rows")
    # ---- cfg 3
    a = side(synth.make_table(1_000_000, 3, "peaks"))
    b = side(synth.make_table(10_000_000, 4, "reads"))
    run("cfg3_semi_1Mx10M", lambda: eng.semi_join(a, b, 24), 11_000_000, "input_rows")
    run("cfg3_anti_1Mx10M", lambda: eng.anti_join(a, b, 24), 11_000_000, "input_rows")
    run("cfg3_count_1Mx10M", lambda: eng.count_overlaps(a, b, 24), 11_000_000, "input_rows")
    # ---- cfg 5
    a = side(synth.make_table(10_000_000, 7, "peaks"))
    b = side(synth.make_table(10_000_000, 8, "peaks"))
    run("cfg5_nearest_10Mx10M", lambda: eng.nearest(a, b, 24)[0], 20_000_000, "input_rows")
    run("cfg5_count_10Mx10M", lambda: eng.count_overlaps(a, b, 24), 20_000_000, "input_rows")
    run("cfg5_semi_10Mx10M", lambda: eng.semi_join(a, b, 24), 20_000_000, "input_rows")


if __name__ == "__main__":
    main()
