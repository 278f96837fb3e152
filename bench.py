#!/usr/bin/env python3
"""bench.py -- the INTERSECTS hot path on MI355X, one JSON line per run.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A *step* is one full pass of the hot path over one batch of synthetic input already resident
in HBM.  The default workload is BASELINE.json configs[3] (cfg 4): 10M
peaks x 100M reads, 24 chromosomes with hg38 lengths, int32 columns, unsorted rows, INNER join
-> (row_a, row_b) index pairs.  ``--workload`` selects the other BASELINE configs (1M x 1M single
chromosome, SEMI / ANTI / COUNT 1M x 10M, NEAREST 10M x 10M) on the same contract.

N > 1 (default workload): one rank per GPU.  ``python bench.py --gpus N`` without WORLD_SIZE in the
environment starts ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a child
process -- before anything touches the GPU -- and relays rank 0's JSON line and the exit code;
under torch.distributed.run it is a rank.  Chromosomes are LPT-packed onto the ranks
(giql_amd.shard), every rank plans its own chromosomes with no data-path collective, then ONE
exchange: an all-gather of the per-rank sizes, one RCCL all-gather of the COMPACT plan (per query
row {row id, first match, match count} + the other side's sorted row ids, 65 MB per rank instead of
400 MB of pairs at N = 8), and every rank expands the gathered plan into the global pairs.  Total
work is fixed as N grows ("strong" scaling).

The line carries ``roofline`` (dominant kernel: algorithmic bytes per launch, as accounted by
the host code that issues the launches, over the hipEvent-measured launch time of the timed
steps) and ``cpu_baseline`` (the reference's DuckDB path when ``import duckdb`` works on the box,
else the oracle's OpenMP sort-merge port), which is also the run's parity check.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

HG38 = "hg38"
WORKLOADS = {
    # name: op, (n_a, kind_a, seed_a), (n_b, kind_b, seed_b), genome (HG38 or a single-chromosome length)
    "cfg4_10Mx100M_24chrom": ("inner", (10_000_000, "peaks", 5), (100_000_000, "reads", 6), HG38),
    "cfg4_small_1Mx10M_24chrom": ("inner", (1_000_000, "peaks", 5), (10_000_000, "reads", 6), HG38),
    # the SAME rows as cfg4, arriving in (chrom, start) order -- what coordinate-sorted BED / BAM-derived tables look
    # like (docs/transpilation/performance.rst:111-130 of the reference advises an index for them); an extra line,
    # never the headline: BASELINE's generator shuffles
    "cfg4_sorted_10Mx100M_24chrom": ("inner", (10_000_000, "peaks", 5), (100_000_000, "reads", 6), HG38),
    "cfg2_sparse_1Mx1M_1chrom": ("inner", (1_000_000, "peaks", 1), (1_000_000, "peaks", 2), 248_956_422),
    "cfg2_dense_1Mx1M_1chrom": ("inner", (1_000_000, "peaks", 1), (1_000_000, "peaks", 2), 10_000_000),
    "cfg3_semi_1Mx10M_24chrom": ("semi", (1_000_000, "peaks", 3), (10_000_000, "reads", 4), HG38),
    "cfg3_anti_1Mx10M_24chrom": ("anti", (1_000_000, "peaks", 3), (10_000_000, "reads", 4), HG38),
    "cfg3_count_1Mx10M_24chrom": ("count", (1_000_000, "peaks", 3), (10_000_000, "reads", 4), HG38),
    "cfg5_nearest_10Mx10M_24chrom": ("nearest", (10_000_000, "peaks", 7), (10_000_000, "peaks", 8), HG38),
    # the headline tables with the 100M-row one INDEXED (giql_hip_index_create_dev: what the reference's users get from
    # CREATE INDEX, docs/transpilation/performance.rst:111-130): per step only the small side's sort + the bucket stage.
    # An extra line, never the headline -- the index build is outside the timed region and reported beside it
    "cfg4_indexed_10Mx100M_24chrom": ("inner_indexed", (10_000_000, "peaks", 5), (100_000_000, "reads", 6), HG38),
    # 4x the headline tables on the same genome (~8,450 reads per 65,536 positions: whole-genome read sets are this
    # dense and denser): the 400M-row side keeps the three-stage sort with buckets of 2^14 keys (round 4).  6.5e9
    # pairs, 52 GB of output; run it with --no-cpu-baseline (parity at this density: tests/test_full_size.py,
    # tests/test_bucket_width.py).  An extra line, never the headline
    "dense_40Mx400M_24chrom": ("inner", (40_000_000, "peaks", 5), (400_000_000, "reads", 6), HG38),
}
DEFAULT_WORKLOAD = "cfg4_10Mx100M_24chrom"
METRIC = {
    "inner": ("overlap-pairs/sec, {a}x{b} INTERSECTS inner join", "pairs/s"),
    "inner_indexed": ("overlap-pairs/sec, {a}x{b} INTERSECTS inner join against a table index", "pairs/s"),
    "semi": ("input rows/sec, {a}x{b} INTERSECTS SEMI join", "rows/s"),
    "anti": ("input rows/sec, {a}x{b} INTERSECTS ANTI join", "rows/s"),
    "count": ("input rows/sec, {a}x{b} count_overlaps", "rows/s"),
    "nearest": ("input rows/sec, {a}x{b} NEAREST k=1", "rows/s"),
}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true",
                   help="skip the general-form / end-to-end / copy-probe legs that follow the timed region")
    p.add_argument("--exchange", default="plan", choices=["plan", "pairs", "none"],
                   help="N>1: what the ranks all-gather -- the compact plan (default), the expanded pairs, nothing")
    p.add_argument("--no-gather", action="store_true", help="same as --exchange none (compute-only scaling)")
    p.add_argument("--force-exchange", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="N>1 exchange backend; 'gloo' is a rehearsal mode for boxes with fewer GPUs than ranks "
                        "(ranks share GPUs, the exchange runs through host memory)")
    p.add_argument("--cpu-sample-chroms", default="18,19,20,21",
                   help="chromosome ids of the bounded cpu_baseline probe")
    p.add_argument("--exchange-impl", default="allgather", choices=["allgather", "p2p"],
                   help="N>1: how the blocks move -- one padded all-gather (RCCL picks the algorithm) or grouped direct "
                        "send / recv to every peer (ncclGroupStart .. End: each block crosses one xGMI link)")
    p.add_argument("--gather", default="all", choices=["all", "root"],
                   help="N>1: who ends up with the result -- every rank (the all-gatherv of the north star) or rank 0 "
                        "only (the other ranks send their block there and skip the expansion)")
    p.add_argument("--verify", action="store_true", help="(the default at N>1; kept for old command lines)")
    p.add_argument("--no-verify", action="store_true",
                   help="N>1: skip the CPU leg.  By default rank 0 runs it on the WHOLE workload after the timed region "
                        "(the other ranks wait), checks the gathered result against it (count + multiset checksum) and "
                        "the run exits non-zero on a mismatch")
    p.add_argument("--master-port", type=int, default=29531)
    p.add_argument("--shard-of", type=int, default=0,
                   help="--gpus 1 only: measure what ONE rank of an N-rank run of the INNER headline executes -- rank "
                        "--shard-rank's LPT share of the chromosomes through exactly the per-rank code (local plan + "
                        "export; the local one-call join; the expansion of all N ranks' plan blocks), each verified "
                        "against the CPU leg.  An extra line, never the headline")
    p.add_argument("--shard-rank", type=int, default=0)
    a = p.parse_args()
    if a.no_gather:
        a.exchange = "none"
    a.verify = (a.gpus > 1 or a.verify or a.force_exchange) and not a.no_verify
    return a


def self_launch(args) -> None:
    """``python bench.py --gpus N`` (N > 1) outside torch.distributed.run: start the N ranks as a
    child process and relay its output and exit code.  Nothing in THIS process has touched the GPU
    (no torch import yet), and it never execs."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


# --------------------------------------------------------------------------- bytes
def sort_bytes_model(sorts, local_min_rows=1 << 25):
    """Algorithmic bytes of the sort launches, mirroring the host code's own accounting
    (run_sort_onesweep in giql_amd/csrc/giql_hip.hip): ``sorts`` = [(rows, payload_arrays, keygen)].
    A pass reads the key and the payload arrays it carries and writes them, 4 B each; the first
    pass synthesises the row ids instead of reading them; a KEYGEN first pass reads (chrom, start);
    sides of ``local_min_rows`` and more take two global passes + the in-LDS bucket sort.
    Returns (global passes, bucket sort) bytes."""
    scatter = local = 0.0
    for n, payload, keygen in sorts:
        w = 1 + payload
        n_pass = 2 if n >= local_min_rows else 4
        first_in = 2 if keygen else w - (1 if payload else 0)   # the rid array is not read by a first pass
        scatter += 4.0 * n * ((first_in + w) + (n_pass - 1) * 2 * w)
        if n >= local_min_rows:
            local += 8.0 * n * w
    return scatter, local


def inner_phase_bytes(n_a, n_b, n_out, form, span_hist, stats_bytes=None, bucket_join=False):
    """Algorithmic (minimal) HBM bytes of every phase of one INNER join: read every input of a
    kernel once, write every output once (DESIGN.md section 3 states each figure).  The sort
    phases take the library's own accounting (``stats.phase_bytes``) when it is there.
    ``bucket_join``: the one-call form whose bucket stage writes the pairs itself (no count / scan / fill)."""
    n = n_a + n_b
    if form == "uniform_b":
        n_q, n_u = n_a, n_b
    elif form == "uniform_a":
        n_q, n_u = n_b, n_a
    else:
        n_q = n_u = 0
    if form == "general":
        sc, lo = sort_bytes_model([(n_a, 2, False), (n_b, 2, False)])
        lin = 12.0 * n + 8.0 * n                       # read 3 columns, write key + end
        count = (8.0 * n_b + 4.0 * n_a) + (8.0 * n_a + 4.0 * n_b + 8.0 * n_a)
        scan = 12.0 * n_a
        fill = 8.0 * n_out + 12.0 * n_b + 4.0 * n_a + 16.0 * n_a + 4.0 * n_b
    else:
        sc, lo = sort_bytes_model([(n_q, 2, False), (n_u, 1, span_hist)])
        lin = (12.0 + 8.0) * n_q + (0.0 if span_hist else 8.0 * n_u + 4.0 * n_u)
        count = 8.0 * n_q + 4.0 * n_u + 8.0 * n_q      # query (key, end) + the other side's keys -> lo, cnt
        scan = 12.0 * n_q
        fill = 8.0 * n_out + 16.0 * n_q + 4.0 * n_u    # pairs + {off, lo, rid} per query row + sorted rids
    if bucket_join and form != "general":
        # the bucket stage reads the sorted side's (key, rid), the queries' (key, end, rid) and writes the pairs;
        # neither side is linearized (both sorted from their raw columns)
        lo = 8.0 * n_u + 12.0 * n_q + 8.0 * n_out
        count = scan = fill = 0.0
        lin = 0.0
    out = {"span": 12.0 * n, "linearize": lin, "sort_scatter": sc, "sort_local": lo, "count": count,
           "scan": scan, "partition": 0.0, "fill": fill}
    for k in ("sort_scatter", "sort_local"):
        if stats_bytes and stats_bytes.get(k):
            out[k] = float(stats_bytes[k])
    return out


DOMINANT_KERNEL = {
    "sort_scatter": "k_onesweep (phase sort_scatter: the global radix passes of both sides)",
    "sort_local": ("k_bucket_sort<1, 2> (phase sort_local: one block per 16-bit bucket sorts the bucket's rows in LDS "
                   "and writes the bucket's pairs)"),
    "fill": "k_fill (phase fill: pair materialisation)",
    "span": "k_chrom_minmax (phase span: per-chromosome spans, lengths, digit histograms)",
}


def op_bytes(op, n_a, n_b, n_out):
    """the operator-level algorithmic bytes."""
    base = 12.0 * (n_a + n_b)
    return base + {"inner": 8.0 * n_out, "semi": 4.0 * n_out, "anti": 4.0 * n_out, "count": 8.0 * n_a,
                   "nearest": 8.0 * n_a}[op]


def csrc_hash() -> str:
    """Hash of the kernel sources: what ties a committed PMC run to the code that is timed here."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "giql_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, form, phase, launches):
    """HBM bytes per launch of a phase from the committed PMC passes (profiles/pmc_traffic.json;
    rocprofv3 cannot run inside the timed process).  None unless the file was collected on THIS
    kernel source (its csrc_hash) for this workload / join form."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        if t.get("csrc_hash") != csrc_hash():
            return None
        return round(t[workload][form][phase] / max(launches, 1))
    except (OSError, KeyError, ValueError, TypeError):
        return None


# ---------------------------------------------------------------------------- inputs
def workload_table(wl, n, seed, kind, chroms=None):
    """One synthetic table of a workload; the ``_sorted_`` workloads order its rows by
    (chrom, start)."""
    import numpy as np

    from giql_amd import synth

    t = synth.make_table(n, seed, kind, chroms=chroms)
    if "_sorted_" in wl:
        order = np.lexsort((t[1], t[0]))
        t = tuple(np.ascontiguousarray(x[order]) for x in t)
    return t


def make_inputs(wl, chroms=None):
    from giql_amd import synth

    op, (n_a, kind_a, seed_a), (n_b, kind_b, seed_b), genome = WORKLOADS[wl]
    if genome == HG38:
        a = workload_table(wl, n_a, seed_a, kind_a, chroms)
        b = workload_table(wl, n_b, seed_b, kind_b, chroms)
        n_chrom = len(synth.HG38_LENGTHS)
    else:
        a = synth.make_single_chrom(n_a, seed_a, kind_a, genome)
        b = synth.make_single_chrom(n_b, seed_b, kind_b, genome)
        n_chrom = 1
    return op, a, b, n_chrom


def short(n):
    return f"{n // 1_000_000}M" if n % 1_000_000 == 0 else str(n)


# ----------------------------------------------------------------------- CPU baseline
def duckdb_baseline(op, a, b, n_chrom, names):
    """The reference's own CPU path, when the box has DuckDB: the IEJoin SQL the reference emits
    for ``SELECT a.rid, b.rid FROM peaks a JOIN reads b ON a.interval INTERSECTS b.interval``
    (reconstructed from src/giql/expanders/intersects_duckdb.py:1283-1400 and
    _per_chrom.py:46-74), executed on all host cores.  Returns None when ``import duckdb`` fails."""
    try:
        import duckdb  # noqa: F401
        import numpy as np
        import pyarrow as pa
    except Exception:
        return None
    try:
        if op != "inner":
            return {"error": "the DuckDB leg is wired for the INNER join only"}
        con = duckdb.connect()

        def table(cols):
            c, s, e = cols
            return pa.table({"chrom": pa.array(np.asarray(names, dtype=object)[c], pa.string()), "start": pa.array(s),
                             "end": pa.array(e), "rid": pa.array(np.arange(len(c), dtype=np.int32))})
        con.register("peaks", table(a))
        con.register("reads", table(b))
        var = "__giql_iejoin_bench"
        branch = ("'SELECT a.\"rid\" AS __giql_p0, b.\"rid\" AS __giql_p1 FROM (SELECT * FROM peaks WHERE \"chrom\" = ' || "
                  "'''' || replace(chrom, '''', '''''') || '''' || ') a JOIN (SELECT * FROM reads WHERE \"chrom\" = ' || "
                  "'''' || replace(chrom, '''', '''''') || '''' || ') b ON a.\"start\" < b.\"end\" AND a.\"end\" > b.\"start\"'")
        set_sql = (f"SET VARIABLE {var} = COALESCE((SELECT string_agg({branch}, ' UNION ALL ') FROM "
                   "(SELECT DISTINCT \"chrom\" AS chrom FROM peaks INTERSECT SELECT DISTINCT \"chrom\" AS chrom FROM reads)), "
                   "'SELECT a.\"rid\" AS __giql_p0, b.\"rid\" AS __giql_p1 FROM peaks a, reads b WHERE FALSE')")
        t0 = time.perf_counter()
        con.execute(set_sql)
        res = con.execute(f"SELECT __giql_p0 AS ra, __giql_p1 AS rb FROM query(getvariable('{var}')) AS w").fetch_arrow_table()
        dt = time.perf_counter() - t0
        ra = res.column("ra").to_numpy().astype(np.int32)
        rb = res.column("rb").to_numpy().astype(np.int32)
        threads = con.execute("SELECT current_setting('threads')").fetchone()[0]
        return {"seconds": dt, "row_a": ra, "row_b": rb, "threads": int(threads), "version": duckdb.__version__}
    except Exception as exc:  # never let the optional leg take the bench down
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}


def cpu_baseline_inner(args, wl, n_chrom, rank_chroms=None, whole=False):
    """Bounded CPU leg of an INNER workload + the data for the parity line.  DuckDB (the reference's
    path) when importable, else the oracle's OpenMP sort-merge port; a probe on four small
    chromosomes sizes the sample (the whole workload when the host does it in ~20 s)."""
    from giql_amd import synth
    from oracle import pyoracle as ora

    op, (n_a, kind_a, seed_a), (n_b, kind_b, seed_b), genome = WORKLOADS[wl]
    threads = ora.max_threads()
    names = synth.HG38_NAMES if genome == HG38 else ["chr1"]
    out = {"unit": "pairs/s", "host_cpu_count": os.cpu_count()}

    def tables(chroms):
        if genome != HG38:
            return make_inputs(wl)[1:3]
        if chroms is None and rank_chroms is not None:
            # N > 1: the global table is "the rows of lower ranks first" -- the same numbering the ranks'
            # global row ids use -- so the reference is the concatenation of the ranks' shards
            import numpy as np

            parts_a = [workload_table(wl, n_a, seed_a, kind_a, c) for c in rank_chroms]
            parts_b = [workload_table(wl, n_b, seed_b, kind_b, c) for c in rank_chroms]
            return (tuple(np.concatenate([p[k] for p in parts_a]) for k in range(3)),
                    tuple(np.concatenate([p[k] for p in parts_b]) for k in range(3)))
        return (workload_table(wl, n_a, seed_a, kind_a, chroms), workload_table(wl, n_b, seed_b, kind_b, chroms))

    def port(chroms):
        a, b = tables(chroms)
        oa, ob = ora.Side(*a), ora.Side(*b)
        t1 = time.perf_counter()
        ra, rb = ora.c_inner(oa, ob, "sweep", threads=threads)
        return oa.n, ob.n, ra, rb, time.perf_counter() - t1

    if genome == HG38:
        probe = [int(c) for c in args.cpu_sample_chroms.split(",") if c != ""]
        pa_, pb_, _pra, _prb, pt = port(probe)
        rows_total = synth.rows_per_chrom(n_a, seed_a) + synth.rows_per_chrom(n_b, seed_b)
        budget_rows = (pa_ + pb_) / max(pt, 1e-3) * 20.0
        chroms, acc = [], 0
        for c in range(len(rows_total)):
            if acc + rows_total[c] > budget_rows and chroms:
                break
            chroms.append(c)
            acc += int(rows_total[c])
        if whole:   # the N > 1 verification: the gathered result can only be checked against the whole workload
            chroms = list(range(len(rows_total)))
        whole = len(chroms) == len(rows_total)
        sel = None if whole else chroms
    else:
        whole, sel, chroms = True, None, [0]
    sa_n, sb_n, ra, rb, st_ = port(sel)
    sample = ("the whole workload" if whole else f"chromosome ids 0..{chroms[-1]} of the same workload")
    out.update({"value": round(ra.shape[0] / st_, 1), "cores": threads, "kind": "port",
                "sample": f"{sample}: {sa_n} x {sb_n} rows -> {ra.shape[0]} pairs in {st_:.2f} s (oracle OpenMP "
                          "sort-merge port, not DuckDB: duckdb is not installed on the box)"})
    parity_ref = (int(ra.shape[0]), ora.c_pairs_checksum(ra, rb)) if whole else None
    # the reference's real path, when the box has it: same sample, same inputs
    a, b = tables(sel)
    dd = duckdb_baseline(op, a, b, n_chrom, names)
    if dd is not None and "error" not in dd:
        ok = (int(dd["row_a"].shape[0]) == int(ra.shape[0])
              and ora.c_pairs_checksum(dd["row_a"], dd["row_b"]) == ora.c_pairs_checksum(ra, rb))
        out.update({"value": round(dd["row_a"].shape[0] / dd["seconds"], 1), "cores": dd["threads"], "kind": "reference",
                    "sample": f"{sample}: {sa_n} x {sb_n} rows -> {dd['row_a'].shape[0]} pairs in {dd['seconds']:.2f} s "
                              f"(DuckDB {dd['version']} IEJoin SQL of the reference, {dd['threads']} threads)",
                    "port_value": round(ra.shape[0] / st_, 1), "duckdb_equals_port": bool(ok)})
    elif dd is not None:
        out["duckdb_probe"] = dd["error"]
    return out, parity_ref


# ------------------------------------------------- one rank's share, measured on one GPU
def run_shard(args):
    """``--shard-of N [--shard-rank r]``: the parts of a step of an N-rank run that need no second GPU, on the one
    GPU this pool gives: (A) rank r's local plan + the export of its compact plan block with global
    row ids -- what precedes the exchange; (B) the same shard through the one-call join (``--exchange none``,
    ``execute(devices=[...])``: results stay sharded); (C) the expansion of ALL N ranks' plan blocks into the global
    pairs -- what follows the exchange on every rank (the peers' blocks are made here, one shard after the other,
    outside the timed region).  Wire time is not measured (DESIGN.md section 6 keeps a stated model for it).
    Parity: (B) against the oracle on the shard's rows, (A) + (C) against the oracle on the whole workload."""
    import numpy as np
    import torch

    from giql_amd import shard, synth
    from giql_amd.engine import DeviceSide, HipEngine
    from oracle import pyoracle as ora

    wl = args.workload
    op, (n_a, kind_a, seed_a), (n_b, kind_b, seed_b), genome = WORKLOADS[wl]
    if op != "inner" or genome != HG38 or args.gpus != 1:
        raise SystemExit("--shard-of measures a rank of the chromosome-sharded INNER workloads at --gpus 1")
    N, r = int(args.shard_of), int(args.shard_rank)
    if not 0 <= r < N:
        raise SystemExit("--shard-rank must be in [0, --shard-of)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    rows_a, rows_b = synth.rows_per_chrom(n_a, seed_a), synth.rows_per_chrom(n_b, seed_b)
    assign = shard.lpt_assign((rows_a + rows_b).tolist(), N)
    layout = [[c for c in range(len(assign)) if assign[c] == q] for q in range(N)]
    size_a = [int(rows_a[layout[q]].sum()) for q in range(N)]
    size_b = [int(rows_b[layout[q]].sum()) for q in range(N)]
    base_a = [sum(size_a[:q]) for q in range(N)]     # the global table = the rows of lower ranks first
    base_b = [sum(size_b[:q]) for q in range(N)]
    n_chrom = len(synth.HG38_LENGTHS)
    eng = HipEngine(0)

    def shard_sides(q):
        ha = workload_table(wl, n_a, seed_a, kind_a, layout[q])
        hb = workload_table(wl, n_b, seed_b, kind_b, layout[q])
        return ha, hb, DeviceSide.from_numpy(*ha, device=dev), DeviceSide.from_numpy(*hb, device=dev)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(steps):
            fn()
            ev[k + 1].record()
        torch.cuda.synchronize(dev)
        wall = (time.perf_counter() - t0) / steps * 1e3
        ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]
        return {"ms_per_step": round(wall, 3), "median_ms": round(statistics.median(ms), 3), "step_ms": [round(x, 3) for x in ms]}

    def export_block(q_sides, q):
        """Plan shard q and export its compact block with global ids: (n_pairs, q_is_a, [q_rid, lo, cnt, s_rid])."""
        _ha, _hb, da, db = q_sides
        n = eng.inner_plan(da, db, n_chrom)
        q_is_a, n_q, n_s = eng.plan_sizes()
        blk = [torch.empty(n_q, dtype=torch.int32, device=dev) for _ in range(3)] + \
              [torch.empty(n_s, dtype=torch.int32, device=dev)]
        eng.plan_export(*blk, rid_add_a=base_a[q], rid_add_b=base_b[q])
        return n, q_is_a, blk

    # ---- (A) this rank: local plan + export (the block is re-used as the send buffer, as PlanGather does)
    t0 = time.time()
    mine = shard_sides(r)
    gen_s = time.time() - t0
    ha, hb, a, b = mine
    n_mine, qa_mine, blk_mine = export_block(mine, r)

    def plan_export_step():
        n = eng.inner_plan(a, b, n_chrom)
        eng.plan_sizes()
        eng.plan_export(*blk_mine, rid_add_a=base_a[r], rid_add_b=base_b[r])
        return n

    part_a = timed(plan_export_step, args.steps, args.warmup)
    st_plan = eng.stats()
    eng.set_profiling(True)
    plan_export_step()
    ph = eng.stats()
    part_a["phase_ms"] = {k: round(v, 3) for k, v in ph["phase_ms"].items() if v > 0}
    part_a["launches"] = int(sum(ph["phase_launches"].values()))
    eng.set_profiling(False)

    # ---- (B) the same shard through the one-call join (a context of its own: the form is the context's memory)
    eng_b = HipEngine(0)
    cap = (int(n_mine * 1.05) + 1024 + (1 << 19) - 1) >> 19 << 19
    out = (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev))
    n0 = eng_b.inner_plan(a, b, n_chrom)
    eng_b.inner_fill(out[0][:n0], out[1][:n0])
    part_b = timed(lambda: eng_b.inner_join_into(a, b, n_chrom, out[0], out[1]), args.steps, args.warmup)
    st_join = eng_b.stats()
    gpu_local = (n0, eng_b.pairs_checksum(out[0][:n0], out[1][:n0]))
    eng_b.set_profiling(True)
    eng_b.inner_join_into(a, b, n_chrom, out[0], out[1])
    phb = eng_b.stats()
    part_b["phase_ms"] = {k: round(v, 3) for k, v in phb["phase_ms"].items() if v > 0}
    part_b["launches"] = int(sum(phb["phase_launches"].values()))
    wa, wb = ora.c_inner(ora.Side(*ha), ora.Side(*hb), "sweep")
    cpu_local = (int(wa.shape[0]), ora.c_pairs_checksum(wa, wb))
    del wa, wb, out
    eng_b.close()

    # ---- (C) every rank's block (peers planned here one after the other), then the timed expansion of all of them
    blocks = [None] * N
    blocks[r] = (n_mine, qa_mine, blk_mine)
    del mine, a, b
    for q in range(N):
        if q != r:
            sides = shard_sides(q)
            blocks[q] = export_block(sides, q)
            del sides
    total = sum(bk[0] for bk in blocks)
    offs = [sum(bk[0] for bk in blocks[:q]) for q in range(N)]
    gcap = (int(total) + (1 << 19) - 1) >> 19 << 19
    gout = (torch.empty(gcap, dtype=torch.int32, device=dev), torch.empty(gcap, dtype=torch.int32, device=dev))

    def expand_all():
        for q, (n_q_pairs, q_is_a, blk) in enumerate(blocks):
            if n_q_pairs == 0:
                continue
            rq, rs = (gout[0], gout[1]) if q_is_a else (gout[1], gout[0])
            eng.fill_from_plan(blk[0], blk[1], blk[2], blk[3], rq[offs[q]:offs[q] + n_q_pairs],
                               rs[offs[q]:offs[q] + n_q_pairs], n_pairs_expected=n_q_pairs)

    part_c = timed(expand_all, args.steps, args.warmup)
    gpu_all = (int(total), eng.pairs_checksum(gout[0][:total], gout[1][:total]))
    parity = {"local_join_pairs_equal": gpu_local[0] == cpu_local[0] == n_mine,
              "local_join_checksum_equal": gpu_local[1] == cpu_local[1]}
    cpu_baseline = None
    if not args.no_cpu_baseline:
        cpu_baseline, ref = cpu_baseline_inner(args, wl, n_chrom, layout, whole=True)
        parity.update({"expanded_pairs_equal": ref[0] == gpu_all[0], "expanded_checksum_equal": ref[1] == gpu_all[1]})
        cpu_baseline["parity"] = parity
    block_bytes = [4 * sum(int(t.numel()) for t in bk[2]) for bk in blocks]
    line = {
        "metric": f"per-rank parts of a step, rank {r} of {N}, {short(n_a)}x{short(n_b)} INTERSECTS inner join (ONE GPU; no wire)",
        "value": round(n_mine / (part_b["ms_per_step"] * 1e-3), 1), "unit": "pairs/s (this rank's one-call join)",
        "n_gpus": 1, "shard_of": N, "shard_rank": r, "steps": args.steps, "warmup": args.warmup,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": wl, "chromosomes": layout[r], "rows_a": size_a[r], "rows_b": size_b[r], "pairs": n_mine,
                   "all_ranks_rows_b": size_b, "all_ranks_pairs": [bk[0] for bk in blocks]},
        "local_plan_export": {**part_a, "sort_local": st_plan["sort_local"], "count_fused": st_plan["count_fused"],
                              "join_form": st_plan["join_form"], "span_hist": st_plan["span_hist"]},
        "local_one_call_join": {**part_b, "bucket_join": st_join["bucket_join"], "sort_local": st_join["sort_local"],
                                "join_form": st_join["join_form"]},
        "expand_all_blocks": {**part_c, "pairs": int(total), "blocks": N},
        "plan_block_bytes": block_bytes,
        "cpu_baseline": cpu_baseline,
        "parity": parity,
        "gen_seconds": round(gen_s, 1),
    }
    eng.close()
    return line


# ------------------------------------------------------- the join against a table index
def run_indexed(args):
    """``--workload cfg4_indexed_...``: table B indexed once (outside the timed region, its build time reported), a
    step = ``giql_hip_inner_join_indexed_dev`` of table A against the index.  Full parity against the CPU leg."""
    import torch

    from giql_amd.engine import DeviceSide, HipEngine

    wl = args.workload
    op, (n_a, kind_a, seed_a), (n_b, kind_b, seed_b), genome = WORKLOADS[wl]
    if args.gpus != 1:
        raise SystemExit("the indexed workload runs on one GPU")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    t0 = time.time()
    _op, ha, hb, n_chrom = make_inputs(wl)
    gen_s = time.time() - t0
    a, b = DeviceSide.from_numpy(*ha, device=dev), DeviceSide.from_numpy(*hb, device=dev)
    eng = HipEngine(0)
    build_ms = []
    index = None
    for _ in range(3):
        if index is not None:
            index.close()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        index = eng.index_create(b, n_chrom)
        torch.cuda.synchronize(dev)
        build_ms.append((time.perf_counter() - t1) * 1e3)
    ra, rb = eng.inner_join_indexed(a, index)
    n_pairs = int(ra.shape[0])
    cap = (int(n_pairs * 1.05) + 1024 + (1 << 19) - 1) >> 19 << 19
    del ra, rb
    out = (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev))
    eng.set_profiling(True)
    for _ in range(max(args.warmup, 1)):
        eng.inner_join_indexed_into(a, index, out[0], out[1])
    warm = eng.stats()
    eng.set_profiling("sort_local")
    eng.inner_join_indexed_into(a, index, out[0], out[1])
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    dom_ms = 0.0
    t1 = time.perf_counter()
    ev[0].record()
    for k in range(args.steps):
        eng.inner_join_indexed_into(a, index, out[0], out[1])
        dom_ms += eng.stats()["phase_ms"]["sort_local"]
        ev[k + 1].record()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t1
    st = eng.stats()
    eng.set_profiling(False)
    step_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    dom_ms /= args.steps
    dom_bytes = float(st["phase_bytes"]["sort_local"])
    join_bytes = op_bytes("inner", n_a, n_b, n_pairs)
    phases = {k: round(v, 3) for k, v in warm["phase_ms"].items() if v > 0}
    phases["sort_local"] = round(dom_ms, 3)
    cpu_baseline = None
    if not args.no_cpu_baseline:
        cpu_baseline, ref = cpu_baseline_inner(args, "cfg4_10Mx100M_24chrom", n_chrom, None, whole=True)
        gpu_sum = eng.pairs_checksum(out[0][:n_pairs], out[1][:n_pairs])
        cpu_baseline["parity"] = {"pairs_equal": ref[0] == n_pairs, "multiset_checksum_equal": gpu_sum == ref[1],
                                  "checked": "all %d pairs of the last timed step against the CPU leg" % n_pairs}
    metric, unit = METRIC[op]
    line = {
        "metric": metric.format(a=short(n_a), b=short(n_b)), "value": round(n_pairs * args.steps / elapsed, 1), "unit": unit,
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "ms_per_step_median": round(statistics.median(step_ms), 3), "step_ms": [round(x, 3) for x in step_ms],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": wl, "n_a": n_a, "n_b": n_b, "n_chrom": n_chrom, "pairs_per_step": n_pairs,
                   "inputs": "resident in HBM before the timed region; table B indexed before the timed region",
                   "join_form": st["join_form"], "index": {"rows": index.n, "hbm_bytes": index.nbytes,
                                                           "form": "general" if index.general else "fixed_length",
                                                           "build_ms": [round(x, 3) for x in build_ms]}},
        "roofline": {"bound": "hbm", "kernel": DOMINANT_KERNEL["sort_local"], "launches_per_step": 1,
                     "avg_launch_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": round(dom_bytes),
                     "achieved": round(dom_bytes / (dom_ms * 1e-3) / 1e9, 1) if dom_ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dom_ms > 0 else 0.0,
                     "traffic": None, "phase_ms": phases,
                     "whole_join": {"algorithmic_bytes": join_bytes,
                                    "frac_of_wall_step": round(join_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}},
        "cpu_baseline": cpu_baseline, "gen_seconds": round(gen_s, 1),
    }
    index.close()
    eng.close()
    return line


# ------------------------------------------------------------------------ the INNER run
def run_inner(args):
    import torch
    import torch.distributed as dist

    from giql_amd import shard, synth
    from giql_amd._lib import GIQL_ERR_CAPACITY, GIQL_ERR_STATE, GiqlHipError
    from giql_amd.engine import DeviceSide, HipEngine

    wl = args.workload
    op, (n_a, kind_a, seed_a), (n_b, kind_b, seed_b), genome = WORKLOADS[wl]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and genome != HG38:
        raise SystemExit("the single-chromosome workloads run on one GPU (giql_amd.distributed splits a dominant "
                         "chromosome by row ranges; bench.py shards whole chromosomes)")
    distributed = world > 1 or args.force_exchange
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    xdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the exchange happens
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    # ---- shard: chromosomes -> ranks (LPT on expected rows); N = 1 keeps everything
    my_chroms = None
    assign = None
    if genome == HG38 and world > 1:
        rows = synth.rows_per_chrom(n_a, seed_a) + synth.rows_per_chrom(n_b, seed_b)
        assign = shard.lpt_assign(rows.tolist(), world)
        my_chroms = [c for c in range(len(synth.HG38_LENGTHS)) if assign[c] == rank]
    t0 = time.time()
    _op, ha, hb, n_chrom = make_inputs(wl, my_chroms)
    gen_s = time.time() - t0
    a = DeviceSide.from_numpy(*ha, device=dev)
    b = DeviceSide.from_numpy(*hb, device=dev)
    loc_na, loc_nb = a.n, b.n
    eng = HipEngine(dev_index)

    # shard-local row index -> global row id: ranks own disjoint chromosome sets and the global table
    # is "rows of lower ranks first", so the map is one offset per side (giql_amd.distributed handles
    # arbitrary row sets with the take kernel)
    base_a = base_b = 0
    D = None
    if distributed:
        from giql_amd import distributed as D

        sizes_t = torch.tensor([loc_na, loc_nb], dtype=torch.int64, device=xdev)
        all_sizes = torch.empty((world, 2), dtype=torch.int64, device=xdev)
        dist.all_gather_into_tensor(all_sizes.view(-1), sizes_t)
        base_a, base_b = (int(x) for x in all_sizes[:rank].sum(0).tolist()) if rank else (0, 0)
    exchange = args.exchange if distributed else "none"
    root = 0 if (args.gather == "root" and distributed) else None
    xplan = D.PlanGather(xdev, impl=args.exchange_impl, root=root) if exchange == "plan" else None
    xpairs = D.PairGather(xdev) if exchange in ("plan", "pairs") else None   # "plan" falls back to it
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if distributed else None
    split_ms = [0.0, 0.0, 0.0]  # [local join, exchange, expansion] summed over the timed steps
    used = {"plan": 0, "pairs": 0}

    out = None
    gout = None  # the gathered global pairs (every rank holds all of them)

    def alloc_out(n):
        """Caller-owned output rows, 5 % head-room, each a 2 MiB multiple long (a row that starts
        inside a cache line makes every 256-byte wave store of the fill touch three lines)."""
        cap = (int(n * 1.05) + 1024 + (1 << 19) - 1) >> 19 << 19
        if os.environ.get("GIQL_BENCH_OUT", "split") == "joint":   # probe: both rows in ONE allocation
            return torch.empty((2, cap), dtype=torch.int32, device=dev)
        return (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev))

    phase_ms, phase_launches = {}, {}
    last_stats = [None]
    last_pairs = [None]

    def local_join():
        """One pass of the hot path over this rank's rows into `out`; returns the pair count."""
        nonlocal out
        if out is not None and not os.environ.get("GIQL_BENCH_NO_FUSE"):
            # ONE C-ABI call into the buffers of the previous step (plan + fill, no stream sync
            # between them when the context's guesses hold); a larger result re-allocates
            try:
                n = eng.inner_join_into(a, b, n_chrom, out[0], out[1])
            except GiqlHipError as exc:
                if exc.code != GIQL_ERR_CAPACITY:
                    raise
                n = eng.last_pairs
                out = None
                out = alloc_out(n)
                eng.inner_fill(out[0][:n], out[1][:n])
        else:
            n = eng.inner_plan(a, b, n_chrom)
            if out is None or out[0].shape[0] < n:
                out = None
                out = alloc_out(n)
            eng.inner_fill(out[0][:n], out[1][:n])
        return n

    def step(collect=False):
        nonlocal gout
        sizes = None
        n = 0
        if ev:
            ev[0].record()
        if exchange == "plan":
            # the plan is the compact description of this rank's pairs: export it with GLOBAL row
            # ids straight into the send block, all-gather, expand every rank's block locally
            n = eng.inner_plan(a, b, n_chrom)
            try:
                q_is_a, n_q, n_s = eng.plan_sizes()
            except GiqlHipError as exc:
                if exc.code != GIQL_ERR_STATE:
                    raise
                q_is_a, n_q, n_s = True, -1, -1   # no compact form on this rank: everybody falls back
            sizes = xplan.sizes(n, n_q, n_s, q_is_a)
        if exchange == "plan" and D.PlanGather.compact(sizes):
            used["plan"] += 1
            views = xplan.send_views(sizes)
            if xdev == dev:
                eng.plan_export(*views, rid_add_a=base_a, rid_add_b=base_b)
            else:  # gloo rehearsal: through host memory
                tmp = [torch.empty(v.shape[0], dtype=torch.int32, device=dev) for v in views]
                eng.plan_export(*tmp, rid_add_a=base_a, rid_add_b=base_b)
                for v, t_ in zip(views, tmp):
                    v.copy_(t_)
            if collect:
                last_stats[0] = eng.stats()
            if ev:
                ev[1].record()
            total = sum(s[0] for s in sizes)
            keeps = xplan.receives()   # gather-to-root: only the root expands (and holds) the result
            if keeps and (gout is None or gout[0].shape[0] < total):
                gout = None
                gout = alloc_out(total)
            offs = [sum(s[0] for s in sizes[:r]) for r in range(world)]

            def expand(r, blk):
                n_r, _q, _s, qa_r = sizes[r]
                if n_r == 0:
                    return
                q_r, lo_r, cnt_r, s_r = (t_.to(dev) for t_ in blk) if xdev != dev else blk
                rq, rs = (gout[0], gout[1]) if qa_r else (gout[1], gout[0])
                eng.fill_from_plan(q_r, lo_r, cnt_r, s_r, rq[offs[r]:offs[r] + n_r], rs[offs[r]:offs[r] + n_r],
                                   n_pairs_expected=n_r)

            # the all-gather runs on RCCL's own stream; this rank's OWN block is expanded meanwhile, straight
            # from the send block (it needs nothing from the wire)
            work = xplan.all_gather_async(sizes)
            if keeps:
                expand(rank, views)
            blocks = work()
            if ev:
                ev[2].record()
            for r, blk in enumerate(blocks):
                if keeps and r != rank and blk is not None:
                    expand(r, blk)
            last_pairs[0] = (gout[0][:total], gout[1][:total]) if keeps else None
        elif exchange in ("plan", "pairs"):
            used["pairs"] += 1
            if exchange == "pairs":
                n = eng.inner_plan(a, b, n_chrom)
            counts = xpairs.counts(n)
            send = xpairs.send_block(max(counts))
            if xdev == dev:
                ra, rb = send[0, :n], send[1, :n]
                eng.inner_fill(ra, rb)       # straight into the send block of the all-gather
                if base_a:
                    ra.add_(base_a)          # local -> global row ids, in place
                if base_b:
                    rb.add_(base_b)
            else:
                ra = torch.empty(n, dtype=torch.int32, device=dev)
                rb = torch.empty(n, dtype=torch.int32, device=dev)
                eng.inner_fill(ra, rb)
                send[0, :n] = (ra + base_a).to(xdev)
                send[1, :n] = (rb + base_b).to(xdev)
            if collect:
                last_stats[0] = eng.stats()
            if ev:
                ev[1].record()
            last_pairs[0] = xpairs.all_gather(counts)
            if ev:
                ev[2].record()
        else:
            n = local_join()
            last_pairs[0] = (out[0][:n], out[1][:n])
            if collect:
                last_stats[0] = eng.stats()
            if ev:
                ev[1].record()
                ev[2].record()
        if collect and last_stats[0] is not None:
            st_ = last_stats[0]
            for k, v in st_["phase_ms"].items():
                phase_ms[k] = phase_ms.get(k, 0.0) + v
            for k, v in st_["phase_launches"].items():
                phase_launches[k] = phase_launches.get(k, 0) + v
        if ev:
            ev[3].record()
            ev[3].synchronize()
            split_ms[0] += ev[0].elapsed_time(ev[1])
            split_ms[1] += ev[1].elapsed_time(ev[2])
            split_ms[2] += ev[2].elapsed_time(ev[3])
        return n

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Warm-up steps run with hipEvent pairs around EVERY phase: they give the phase table.  The timed
    # steps keep events only around the dominant kernel's launches (the global sort passes), which is
    # what the roofline object is computed from: every event pair costs the launch stream a few
    # microseconds of idle time, ~0.1 ms per step with all the phases timed.
    eng.set_profiling(True)
    for _ in range(args.warmup):
        step(collect=True)
    warm = last_stats[0] if args.warmup else None
    warm_phase_ms = dict(warm["phase_ms"]) if warm else {}
    light = bool(warm_phase_ms)
    # the dominant phase = the one that took longest in the last warm-up step (the global sort passes, or -- when
    # the bucket stage writes the pairs itself -- that stage's one kernel)
    # (in the join-in-the-bucket-stage form that stage's kernel IS the largest single kernel -- rocprofv3's top line;
    # the four global passes are three different kernels -- whatever the warm-up's event overhead made of the phases)
    dom_phase = ("sort_local" if (warm or {}).get("bucket_join") else max(warm_phase_ms, key=warm_phase_ms.get)) if light \
        else "sort_scatter"
    eng.set_profiling(dom_phase if light else True)
    if light:
        step(collect=True)   # one more untimed step in the timed steps' event mode (a switch of mode in front of the
                             # timed region showed up in its first step: see run_rowop)
    phase_ms.clear()
    phase_launches.clear()
    sync_all()
    split_ms[0] = split_ms[1] = split_ms[2] = 0.0
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    n_local = 0
    step_ev[0].record()
    for k in range(args.steps):
        n_local = step(collect=True)
        step_ev[k + 1].record()
    sync_all()
    elapsed = time.perf_counter() - t0
    st = last_stats[0]
    eng.set_profiling(False)
    step_ms = [step_ev[k].elapsed_time(step_ev[k + 1]) for k in range(args.steps)]

    t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
    tot = torch.tensor([n_local, loc_na, loc_nb], dtype=torch.int64, device=xdev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    n_pairs, tot_na, tot_nb = (int(x) for x in tot.tolist())

    # per-rank view of a step (outside the timed region): local join, exchange, expansion, HBM fraction
    per_step_ms = {k: v / args.steps for k, v in phase_ms.items()}
    if light:
        per_step_ms = {**warm_phase_ms, dom_phase: per_step_ms.get(dom_phase, 0.0)}
    if exchange == "plan":   # the expansion's scan / partition / fill are timed by `expand_ms`, not as join phases
        per_step_ms = {k: v for k, v in per_step_ms.items() if k not in ("scan", "partition", "fill") or not used["plan"]}
    device_ms = sum(per_step_ms.values())
    mine = {"rank": rank, "rows_a": loc_na, "rows_b": loc_nb, "pairs": n_local,
            "join_device_ms": round(device_ms, 3),
            "hbm_frac": round(op_bytes("inner", loc_na, loc_nb, n_local) / max(device_ms * 1e-3, 1e-9) / 1e9 / HBM_PEAK_GBS, 4)}
    if distributed:
        mine.update({"local_join_ms": round(split_ms[0] / args.steps, 3), "exchange_ms": round(split_ms[1] / args.steps, 3),
                     "expand_ms": round(split_ms[2] / args.steps, 3)})
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    else:
        per_rank = [mine]

    line = None
    if rank == 0:
        per_step_launches = {k: v // args.steps for k, v in phase_launches.items()}
        form, span_hist = st["join_form"], bool(st.get("span_hist", False))
        pbytes = inner_phase_bytes(loc_na, loc_nb, n_local, form, span_hist, st.get("phase_bytes"),
                                   bool(st.get("bucket_join")))
        dom = dom_phase   # the phase timed in the timed steps
        dom_ms = per_step_ms.get(dom, 0.0)
        dom_launches = max(per_step_launches.get(dom, 0), 1)
        dom_bytes = pbytes[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        join_bytes = op_bytes("inner", loc_na, loc_nb, n_local)
        warm_launches = warm["phase_launches"] if warm else {}
        kernels = {}
        for k, ms in per_step_ms.items():
            if ms <= 0:
                continue
            bts = pbytes.get(k, 0.0)
            kernels[k] = {"ms": round(ms, 3), "launches": int(warm_launches.get(k, per_step_launches.get(k, 0))),
                          "algorithmic_GB": round(bts / 1e9, 4),
                          "achieved_GBps": round(bts / (ms * 1e-3) / 1e9, 1) if bts else None,
                          "frac": round(bts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if bts else None}
        roofline = {
            "bound": "hbm",
            "kernel": DOMINANT_KERNEL.get(dom, dom) if not (dom == "sort_local" and not st.get("bucket_join"))
            else "k_bucket_sort (phase sort_local: every 16-bit bucket sorted in LDS)",
            "launches_per_step": dom_launches,
            "avg_launch_ms": round(dom_ms / dom_launches, 4),
            "algorithmic_bytes_per_launch": round(dom_bytes / dom_launches),
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic(wl, form, dom, dom_launches) if world == 1 else None,
            "bytes_source": ("giql_hip_stats.phase_bytes (accounted by the host code that issues the passes)"
                             if (st.get("phase_bytes") or {}).get(dom) else "bench.sort_bytes_model"),
            "whole_join": {
                "algorithmic_bytes": join_bytes,
                "device_ms": round(device_ms, 3),
                "achieved": round(join_bytes / (device_ms * 1e-3) / 1e9, 1) if device_ms > 0 else 0.0,
                "frac": round(join_bytes / (device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if device_ms > 0 else 0.0,
                # ... and over the WALL time of a timed step (the phase table above comes from a warm-up step with an
                # event pair around every phase, which costs the stream a few microseconds each)
                "frac_of_wall_step": round(join_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if elapsed > 0 else 0.0,
            },
            "kernels": kernels,
            "phase_ms_source": (f"{dom}: hipEvents in the timed steps; other phases: hipEvents in the last warm-up "
                                "step (the timed steps record events around the dominant phase only)"
                                if light else "hipEvents in the timed steps"),
        }

        cpu_baseline = None
        if (not args.no_cpu_baseline and world == 1 and not distributed) or (distributed and args.verify):
            # rank 0 at N = 1 (the contract); at N > 1 the same leg checks the GATHERED result (--no-verify skips it)
            layout = ([[c for c in range(n_chrom) if assign[c] == r] for r in range(world)]
                      if world > 1 and genome == HG38 else None)
            cpu_baseline, ref = cpu_baseline_inner(args, wl, n_chrom, layout, whole=distributed)
            if ref is not None and last_pairs[0] is not None:
                pa_, pb_ = last_pairs[0] if not isinstance(last_pairs[0], list) else (
                    torch.cat([x[0] for x in last_pairs[0]]), torch.cat([x[1] for x in last_pairs[0]]))
                # (the gloo rehearsal gathers the pairs in HOST memory: the checksum kernel wants them on the device)
                gpu_sum = eng.pairs_checksum(pa_.to(dev).contiguous(), pb_.to(dev).contiguous())
                cpu_baseline["parity"] = {"pairs_equal": ref[0] == n_pairs and int(pa_.shape[0]) == n_pairs,
                                          "multiset_checksum_equal": gpu_sum == ref[1],
                                          "checked": "all %d pairs of the last timed step against the CPU leg" % n_pairs}
            elif ref is not None:   # --exchange none: the results stay sharded, only their total can be checked here
                cpu_baseline["parity"] = {"pairs_equal": ref[0] == n_pairs, "multiset_checksum_equal": None,
                                          "checked": "the pair count only (no gathered result: --exchange none)"}

        extras = {}
        if world == 1 and not distributed and not args.no_extras:
            extras = inner_extras(eng, dev_index, a, b, ha, hb, n_chrom, n_local, alloc_out, join_bytes)

        metric, unit = METRIC["inner"]
        value = n_pairs * args.steps / elapsed
        line = {
            "metric": metric.format(a=short(tot_na), b=short(tot_nb)),
            "value": round(value, 1),
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_step_median": round(statistics.median(step_ms), 3) if step_ms else None,
            "step_ms": [round(x, 3) for x in step_ms],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": wl,
                "n_a": tot_na, "n_b": tot_nb, "n_chrom": n_chrom, "pairs_per_step": n_pairs,
                "parallelism": f"chrom-shard x{world}" + ("" if not distributed else {
                    "plan": (f" + {args.backend} {'grouped send/recv' if xplan is not None and xplan.impl == 'p2p' else 'all-gather'}"
                             f" of the compact plan, expanded on {'rank 0 only' if root is not None else 'every rank'}"),
                    "pairs": f" + {args.backend} all-gather of the pairs", "none": ", no gather"}[exchange]),
                "inputs": "resident in HBM before the timed region",
                "join_form": form,
                "presorted_side_skipped_its_sort": bool(st.get("presorted", False)),
                "span_hist": span_hist,
                "sort": (("two global passes + in-LDS bucket sort for sides of 300-2800 rows per 16-bit bucket"
                          if st.get("bucket_bits", 16) == 16 else
                          f"three global passes + in-LDS bucket sort, buckets of 2^{st.get('bucket_bits')} keys (dense table)")
                         if st.get("sort_local") else "four global passes"),
                "pairs_written_by": ("the bucket stage of the sort (k_bucket_sort<1, 2>: no count / scan / fill kernel)"
                                     if st.get("bucket_join") else "k_fill"),
            },
            "hbm_algorithmic_GBps": round(op_bytes("inner", tot_na, tot_nb, n_pairs) * args.steps / elapsed / 1e9, 1),
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "sort_order_fallbacks": st.get("sort_order_fallbacks", 0),
            "sort_resorted": bool(st.get("sort_resorted", False)),
            "per_rank": per_rank,
            "gen_seconds": round(gen_s, 1),
        }
        line.update(extras)
        if distributed:
            line["exchange"] = {"mode": exchange, "impl": (xplan.impl if xplan is not None else "allgather"),
                                "gather": "root" if root is not None else "all",
                                "steps_compact": used["plan"], "steps_expanded": used["pairs"],
                                "bytes_per_rank": (xplan.bytes_per_rank() if used["plan"] else None)}
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    return line


def inner_extras(eng, dev_index, a, b, ha, hb, n_chrom, n_pairs, alloc_out, join_bytes):
    """Legs that follow the timed region at N = 1: the general two-class form on the same inputs (a
    second context under GIQL_HIP_NO_UNIFORM=1), the PCIe-inclusive end-to-end time through the
    host-buffer entry point, and this box's measured copy rate with the library's own copy kernel."""
    import torch

    from giql_amd.engine import HipEngine

    out = {}
    try:
        # the box's streaming ceiling by access shape (giql_hip_stream_probe_dev): best read-only / write-only / copy
        # rate over {1, 4, 8 accesses in flight} x {default, non-temporal} x {4, 8, 16 blocks per CU}, every byte
        # moved counted once, + hipMemcpyDtoDAsync as an outside reference; measured_copy16 is round 3's naive probe
        # (one 16-byte load in flight per thread), kept for comparison
        pr = eng.stream_probe(nbytes=1024 << 20, reps=3)
        out["measured_read_GBps"], out["measured_write_GBps"] = pr["read"], pr["write"]
        out["measured_copy_GBps"], out["measured_memcpy_d2d_GBps"] = pr["copy"], pr["memcpy_d2d"]
        out["measured_best_shapes"] = {k: max((v, n) for n, v in pr["shapes"].items() if n.startswith(k + "/"))[1]
                                       for k in ("read", "write", "copy")}
        out["measured_copy16_GBps"] = round(eng.copy_probe(), 1)
    except Exception as exc:  # e.g. not enough free HBM next to the workload
        out["measured_copy_GBps"] = None
        out["measured_copy_error"] = str(exc)[:200]
    uniform_sum = None
    try:
        os.environ["GIQL_HIP_NO_UNIFORM"] = "1"
        g = HipEngine(dev_index)
    finally:
        del os.environ["GIQL_HIP_NO_UNIFORM"]
    try:
        buf = alloc_out(n_pairs)
        n = eng.inner_plan(a, b, n_chrom)
        eng.inner_fill(buf[0][:n], buf[1][:n])
        uniform_sum = eng.pairs_checksum(buf[0][:n], buf[1][:n])
        reps = 3
        for _ in range(2):   # plan + fill (two calls, the count read back in between)
            n = g.inner_plan(a, b, n_chrom)
            g.inner_fill(buf[0][:n], buf[1][:n])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            n = g.inner_plan(a, b, n_chrom)
            g.inner_fill(buf[0][:n], buf[1][:n])
        torch.cuda.synchronize()
        plan_fill_ms = (time.perf_counter() - t0) / reps * 1e3
        # the one-call form (what the headline times for the fixed-length form): on a settled context the pairs of
        # the general form leave from the bucket stage of the larger side's sort too
        g.set_profiling(True)
        for _ in range(2):
            n = g.inner_join_into(a, b, n_chrom, buf[0], buf[1])
        phases = g.stats()["phase_ms"]
        g.set_profiling(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            n = g.inner_join_into(a, b, n_chrom, buf[0], buf[1])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        dms = sum(phases.values())
        out["general_form"] = {"ms_per_step": round(ms, 3), "pairs": n, "join_form": g.stats()["join_form"],
                               "pairs_written_by": "the bucket stage" if g.stats().get("bucket_join") else "k_c1_emit + k_fill",
                               "plan_then_fill_ms_per_step": round(plan_fill_ms, 3),
                               "device_ms": round(dms, 3),
                               "whole_join_frac": round(join_bytes / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dms > 0 else None,
                               "pairs_equal_default_form": n == n_pairs,
                               "checksum_equal_default_form": g.pairs_checksum(buf[0][:n], buf[1][:n]) == uniform_sum}
        del buf
    finally:
        g.close()
    # host columns in, host index pairs out (giql_hip_inner).  Default since round 4: the columns go up, the COMPACT PLAN
    # comes down (0.52 GB instead of 3.2 GB of pairs) and host threads expand it into plain memory while the sorted ids
    # are still arriving; GIQL_HIP_E2E_COMPACT=0 is round 3's path (pairs downloaded into page-locked arrays, the larger
    # table uploaded block by block).  The default path's pairs are checked against the timed step's checksum.
    try:
        seen = {}

        def inspect(va, vb):
            ta = torch.from_numpy(va).to(eng.device)
            tb = torch.from_numpy(vb).to(eng.device)
            seen["sum"] = eng.pairs_checksum(ta, tb)
            seen["n"] = int(va.shape[0])

        first_ms, n = eng.inner_join_host_timed(ha, hb, n_chrom)
        ms, n = eng.inner_join_host_timed(ha, hb, n_chrom)
        eng.inner_join_host_timed(ha, hb, n_chrom, inspect=inspect)
        out["t_e2e_ms"] = round(ms, 1)
        out["t_e2e_first_call_ms"] = round(first_ms, 1)
        out["t_e2e_parity"] = {"pairs_equal": n == n_pairs and seen.get("n") == n_pairs,
                               "multiset_checksum_equal": seen.get("sum") == uniform_sum}
        out["t_e2e_note"] = ("giql_hip_inner: pageable host columns -> device, join, the compact plan (per-query {id, first "
                             "match, count} + the sorted ids) -> host, expanded by 32 host threads into plain host memory; "
                             "never the headline value")
    except Exception as exc:
        out["t_e2e_ms"] = None
        out["t_e2e_error"] = str(exc)[:200]
    try:
        os.environ["GIQL_HIP_E2E_COMPACT"] = "0"
        first_ms, n = eng.inner_join_host_timed(ha, hb, n_chrom)
        ms, n = eng.inner_join_host_timed(ha, hb, n_chrom)
        out["t_e2e_pairs_download_ms"] = round(ms, 1)
        out["t_e2e_pairs_download_first_call_ms"] = round(first_ms, 1)
    except Exception as exc:
        out["t_e2e_pairs_download_ms"] = None
        out["t_e2e_pairs_download_error"] = str(exc)[:200]
    finally:
        os.environ.pop("GIQL_HIP_E2E_COMPACT", None)
    return out


# ---------------------------------------------------------------------- the row operators
def run_rowop(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    from giql_amd import shard, synth
    from giql_amd.engine import DeviceSide, HipEngine

    wl = args.workload
    op, (n_a, kind_a, seed_a), (n_b, kind_b, seed_b), genome = WORKLOADS[wl]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and genome != HG38:
        raise SystemExit("the single-chromosome workloads run on one GPU")
    distributed = world > 1
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    xdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the exchange happens
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    # ---- shard: the A rows' chromosomes -> ranks (LPT on rows of both sides); every A row needs exactly the
    # B rows of its chromosome, so a rank holds both sides of its chromosomes and there is no data-path collective
    rank_chroms = None
    my_chroms = None
    if distributed:
        rows = synth.rows_per_chrom(n_a, seed_a) + synth.rows_per_chrom(n_b, seed_b)
        assign = shard.lpt_assign(rows.tolist(), world)
        rank_chroms = [[c for c in range(len(synth.HG38_LENGTHS)) if assign[c] == r] for r in range(world)]
        my_chroms = rank_chroms[rank]
    t0 = time.time()
    op, ha, hb, n_chrom = make_inputs(wl, my_chroms)
    gen_s = time.time() - t0
    a = DeviceSide.from_numpy(*ha, device=dev)
    b = DeviceSide.from_numpy(*hb, device=dev)
    eng = HipEngine(dev_index)
    # shard-local row -> global row: the global table is "rows of lower ranks first" (as in run_inner)
    base_a = base_b = 0
    if distributed:
        from giql_amd import distributed as D

        sizes_t = torch.tensor([a.n, b.n], dtype=torch.int64, device=xdev)
        all_sizes = torch.empty((world, 2), dtype=torch.int64, device=xdev)
        dist.all_gather_into_tensor(all_sizes.view(-1), sizes_t)
        base_a, base_b = (int(x) for x in all_sizes[:rank].sum(0).tolist()) if rank else (0, 0)
    root = 0 if (args.gather == "root" and distributed) else None
    local = {"semi": lambda: eng.semi_join(a, b, n_chrom), "anti": lambda: eng.anti_join(a, b, n_chrom),
             "count": lambda: eng.count_overlaps(a, b, n_chrom),
             # N = 1: the (int32 idx, int32 distance) output SURVEY.md section 8 a9 sizes -- one 8-byte record per row
             # (giql_hip_nearest32_dev); the N > 1 exchange carries the int64 ABI's two arrays
             "nearest": (lambda: eng.nearest(a, b, n_chrom)) if distributed
             else (lambda: tuple(eng.nearest32(a, b, n_chrom).unbind(1)))}[op]

    last_local = [0]

    def fn():
        """One pass of the operator over this rank's rows; N > 1: + the path's one exchange, the all-gather
        of the per-row results (global row ids; blocks in rank order = global A row order)."""
        r = local()
        last_local[0] = int((r[0] if isinstance(r, tuple) else r).shape[0])
        if not distributed:
            return r
        # int32 blocks: row ids and counts are int32 (fewer than 2^31 rows per table); only NEAREST's distance is
        # 64-bit and travels as two int32 words
        if op in ("semi", "anti"):
            block = (r + base_a).to(xdev).view(1, -1)
        elif op == "count":
            block = r.to(torch.int32).to(xdev).view(1, -1)
        else:
            idx = r[0]
            block = torch.cat([torch.where(idx >= 0, idx + base_b, idx).view(1, -1), D._i64_as_i32_rows(r[1])]).to(xdev)
        blocks = D.gather_blocks(block, impl=args.exchange_impl, root=root)
        if root is not None and rank != root:
            return r
        if op in ("semi", "anti"):
            return torch.cat([blk[0] for blk in blocks])
        if op == "count":
            return torch.cat([blk[0] for blk in blocks]).to(torch.int64)
        return (torch.cat([blk[0] for blk in blocks]),
                torch.cat([D._i32_rows_as_i64(blk[1:3]) for blk in blocks]))

    def sync_all():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the phase events are recorded in ONE extra call AFTER the timed region: switching them off in front of it left
    # a one-off ~1.1 ms in the first timed call (box-dependent; the calls themselves take 0.3 ms)
    res = None
    for _ in range(max(args.warmup, 1)):
        res = fn()
    sync_all()
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        res = fn()
        step_ms.append((time.perf_counter() - ts) * 1e3)   # (a call ends with its read-back: host time = step time)
    sync_all()
    elapsed = time.perf_counter() - t0
    eng.set_profiling(True)
    fn()
    torch.cuda.synchronize(dev)
    st = eng.stats()
    phases = {k: v for k, v in st["phase_ms"].items() if v > 0}
    eng.set_profiling(False)
    sync_all()
    loc_na, loc_nb = a.n, b.n
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        tot = torch.tensor([loc_na, loc_nb], dtype=torch.int64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(t.item())
        tot_na, tot_nb = (int(x) for x in tot.tolist())
        if rank != 0:
            dist.barrier()   # rank 0 runs the CPU leg below; leave together
            dist.destroy_process_group()
            eng.close()
            return None
        if args.verify:
            # the global tables the gathered result refers to: the ranks' shards, rank after rank
            parts = [make_inputs(wl, cs) for cs in rank_chroms]
            ha = tuple(np.concatenate([p[1][k] for p in parts]) for k in range(3))
            hb = tuple(np.concatenate([p[2][k] for p in parts]) for k in range(3))
    g_na, g_nb = (tot_na, tot_nb) if distributed else (loc_na, loc_nb)   # the whole job's rows
    n_out = int((res[0] if isinstance(res, tuple) else res).shape[0])
    device_ms = sum(phases.values())
    alg = op_bytes(op, loc_na, loc_nb, last_local[0])   # this rank's shard: what its kernels (device_ms) moved
    dom = max(phases, key=lambda k: phases[k]) if phases else None
    dom_bytes = (st.get("phase_bytes") or {}).get(dom, 0) if dom else 0
    roofline = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                "whole_operator": {"algorithmic_bytes": alg, "device_ms": round(device_ms, 3),
                                   "achieved": round(alg / (device_ms * 1e-3) / 1e9, 1) if device_ms else 0.0,
                                   "frac": round(alg / (device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if device_ms else 0.0},
                "phase_ms": {k: round(v, 3) for k, v in phases.items()},
                "phase_launches": {k: v for k, v in st["phase_launches"].items() if v},
                "phase_ms_source": "hipEvents in one extra call after the timed region (the timed calls run without events)"}
    if dom and dom_bytes:
        launches = max(st["phase_launches"].get(dom, 1), 1)
        ach = dom_bytes / (phases[dom] * 1e-3) / 1e9
        roofline.update({"kernel": f"phase {dom}", "launches_per_step": launches,
                         "avg_launch_ms": round(phases[dom] / launches, 4),
                         "algorithmic_bytes_per_launch": round(dom_bytes / launches),
                         "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                         "bytes_source": "giql_hip_stats.phase_bytes"})
    else:  # the dominant phase has no byte accounting: the operator-level figure of SURVEY.md 8(d)
        w = roofline["whole_operator"]
        roofline.update({"kernel": f"whole operator (dominant phase: {dom})", "achieved": w["achieved"], "frac": w["frac"]})

    cpu_baseline = None
    # the CPU leg: at N = 1 (the contract); at N > 1 only to verify the gathered result (--verify)
    if (not args.no_cpu_baseline and not distributed) or (distributed and args.verify):
        from oracle import pyoracle as ora

        oa, ob = ora.Side(*ha), ora.Side(*hb)
        threads = ora.max_threads()
        t1 = time.perf_counter()
        if op in ("semi", "anti"):
            want = ora.c_semi_anti(oa, ob, op == "anti", threads=threads)
            dt = time.perf_counter() - t1
            equal = bool(np.array_equal(res.cpu().numpy(), want))
        elif op == "count":
            want = ora.c_count(oa, ob, "sweep", threads=threads)
            dt = time.perf_counter() - t1
            equal = bool(np.array_equal(res.cpu().numpy(), want))
        else:
            wi, wd = ora.c_nearest_k1(oa, ob, method="sweep", threads=threads)
            dt = time.perf_counter() - t1
            j = res[0].cpu().numpy()
            ok = j >= 0
            equal = bool(np.array_equal(res[1].cpu().numpy(), wd) and np.array_equal(ok, wi >= 0)
                         and np.array_equal(hb[1][j[ok]], hb[1][wi[ok]]) and np.array_equal(hb[2][j[ok]], hb[2][wi[ok]]))
        cpu_baseline = {"value": round((g_na + g_nb) / dt, 1), "unit": "rows/s", "cores": threads, "kind": "port",
                        "sample": f"the whole workload: {g_na} x {g_nb} rows in {dt:.2f} s (oracle OpenMP sweep, not DuckDB)",
                        "host_cpu_count": os.cpu_count(),
                        "parity": {"equal": equal, "checked": "every output row of the last timed call (NEAREST: distances, "
                                                                "presence and the matched (start, end); ids are tie-ambiguous)"}}
    metric, unit = METRIC[op]
    line = {
        "metric": metric.format(a=short(g_na), b=short(g_nb)),
        "value": round((g_na + g_nb) * args.steps / elapsed, 1),
        "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "ms_per_step_median": round(statistics.median(step_ms), 3) if step_ms else None,
        "step_ms": [round(x, 3) for x in step_ms],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": wl, "n_a": g_na, "n_b": g_nb, "n_chrom": n_chrom, "rows_out": n_out,
                   "inputs": "resident in HBM before the timed region", "form": st.get("join_form"),
                   "parallelism": (f"{world} ranks, the A rows' chromosomes LPT-sharded, one "
                                   f"{'gather to rank 0' if root is not None else 'all-gather'} of the per-row results as int32 blocks "
                                   f"({args.backend}, {'grouped send/recv' if (root is not None or args.exchange_impl == 'p2p') else 'padded all-gather'})")
                   if distributed else "1 GPU"},
        "hbm_algorithmic_GBps": round(op_bytes(op, g_na, g_nb, n_out) * args.steps / elapsed / 1e9, 1),
        "per_rank": [{"rank": 0, "rows_a": loc_na, "rows_b": loc_nb, "rows_out": last_local[0],
                      "device_ms": round(device_ms, 3)}],
        "roofline": roofline, "cpu_baseline": cpu_baseline, "gen_seconds": round(gen_s, 1),
    }
    eng.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return line


def main() -> None:
    args = parse_args()
    self_launch(args)
    op = WORKLOADS[args.workload][0]
    line = run_shard(args) if args.shard_of else (run_inner(args) if op == "inner" else
                                                 run_indexed(args) if op == "inner_indexed" else run_rowop(args))
    if line is not None:
        print(json.dumps(line), flush=True)
        parity = ((line.get("cpu_baseline") or {}).get("parity") or {})
        if any(v is False for v in parity.values()):
            sys.stderr.write("bench.py: the result does NOT match the CPU leg: %s\n" % json.dumps(parity))
            sys.exit(3)


if __name__ == "__main__":
    main()
