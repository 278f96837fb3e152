#!/usr/bin/env python3
"""Turn the rocprofv3 databases written by tools/profile.sh into the summaries kept
under profiles/: <tag>_kernel_stats.csv (the --stats view), <tag>_pmc_hbm_traffic.csv
(FETCH_SIZE / WRITE_SIZE per kernel, raw counter x 1024 B) and profiles/pmc_traffic.json
(HBM bytes per join and phase: FETCH_SIZE x 2 + WRITE_SIZE -- the x 2 is the gfx950
correction of MI355X_MICROARCH.md, "HBM").  usage: pmc_summary.py <tag> [joins]   (joins = 1: only the LAST join
of the PMC run is counted -- the settled form; n > 1: everything, divided by n)"""
import csv
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASE_OF = [("k_onesweep", "sort_scatter"), ("k_bucket_sort_big", "aux"), ("k_bucket_sort", "sort_local"), ("k_bucket_bounds_fused", "count"), ("k_bucket_bounds", "sort_local"), ("k_linearize", "linearize"), ("k_digit_offsets", "linearize"), ("k_fold_top", "linearize"),
            ("k_init_minmax", "span"), ("k_chrom_minmax", "span"), ("k_chrom_offsets", "span"),
            ("k_range_count", "count"), ("k_count_partition", "count"), ("k_c1_count", "count"),
            ("k_scan_", "scan"), ("k_partition", "partition"), ("k_fill", "fill"), ("k_c1_emit", "fill")]


def short(name: str) -> str:
    m = re.search(r"giql::(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][:48]


def main() -> None:
    tag = sys.argv[1]
    joins = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    prof = os.path.join(ROOT, "profiles")
    stats_csv = os.path.join(src, "trace", "trace_kernel_stats.csv")
    if os.path.exists(stats_csv):  # rocprofv3 --output-format csv (tools/profile.sh since round 3)
        with open(stats_csv) as f:
            rows = [(r["Name"], int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]), int(r["MinNs"]),
                     int(r["MaxNs"])) for r in csv.DictReader(f)]
        rows.sort(key=lambda r: -r[2])
    else:
        c = sqlite3.connect(os.path.join(src, "trace", "trace_results.db"))
        rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                         "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(os.path.join(prof, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 3), round(100.0 * r[2] / total, 4), r[4], r[5]])
    per_phase = {}
    with open(os.path.join(prof, f"{tag}_pmc_hbm_traffic.csv"), "w", newline="") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 "
                f"({joins} joins), 10Mx100M\n# raw counter x 1024 B; FETCH_SIZE under-reports streaming reads by 2x on "
                "gfx950 (MI355X_MICROARCH.md HBM): double it\n")
        w = csv.writer(f)
        w.writerow(["counter", "kernel", "dispatches", "sum_GB_raw", "max_dispatch_GB_raw"])
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cc_csv = os.path.join(src, f"pmc_{counter}", "pmc_counter_collection.csv")
            if os.path.exists(cc_csv):
                with open(cc_csv) as fc:
                    recs = sorted((r for r in csv.DictReader(fc) if r["Counter_Name"] == counter),
                                  key=lambda r: int(r["Dispatch_Id"]))
                # every join starts with k_init_minmax: the LAST join of the run is the settled form (the first one of a
                # context reads its guesses back and takes the ordinary kernels)
                starts = [i for i, r in enumerate(recs) if "k_init_minmax" in r["Kernel_Name"]]
                if joins == 1 and starts:
                    recs = recs[max(starts[-1] - 1, 0):]   # (- 1: the memset that precedes it)
                values = [(r["Kernel_Name"], float(r["Counter_Value"])) for r in recs]
            else:
                db = sqlite3.connect(os.path.join(src, f"pmc_{counter}", "pmc_results.db"))
                values = db.execute("select kernel_name, value from counters_collection where counter_name = ?",
                                    (counter,)).fetchall()
            agg = {}
            for name, value in values:
                k = short(name)
                a = agg.setdefault(k, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += value * 1024.0
                a[2] = max(a[2], value * 1024.0)
            for k, (n, s, m) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([counter, k, n, round(s / 1e9, 4), round(m / 1e9, 4)])
                for prefix, phase in PHASE_OF:
                    if k.startswith(prefix):
                        per_phase[phase] = per_phase.get(phase, 0.0) + s * (2.0 if counter == "FETCH_SIZE" else 1.0) / joins
                        break
    bench = open(os.path.join(src, "pmc_FETCH_SIZE.log")).read()
    form = re.search(r'"join_form": "([a-z_]+)"', bench)
    workload = re.search(r'"workload": "([A-Za-z0-9_]+)"', bench)
    path = os.path.join(prof, "pmc_traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    sys.path.insert(0, ROOT)
    import subprocess

    import bench  # csrc_hash(): ties these counters to the kernel sources they were collected on

    if data.get("csrc_hash") != bench.csrc_hash():
        data = {}  # counters of older kernels say nothing about these
    data["csrc_hash"] = bench.csrc_hash()
    try:
        data["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                                        text=True).stdout.strip()
    except OSError:
        data["commit"] = None
    data["source"] = (f"profiles/{tag}_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                      "FETCH_SIZE doubled per MI355X_MICROARCH.md)")
    data.setdefault(workload.group(1), {})[form.group(1)] = {k: round(v) for k, v in per_phase.items()}
    json.dump(data, open(path, "w"), indent=1)
    print(json.dumps(data[workload.group(1)][form.group(1)]))


if __name__ == "__main__":
    main()
