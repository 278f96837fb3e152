#!/usr/bin/env python3
"""Print this box's streaming ceiling by access shape (giql_hip_stream_probe_dev): read-only, write-only and copy
rates for 1 / 2 / 4 / 8 sixteen-byte accesses in flight per thread, default and non-temporal cache policy, grids of
n_cu x {4, 8, 16, 32} blocks, + hipMemcpyDtoDAsync.  GB/s, every byte moved counted once.
usage: python tools/stream_probe.py [MiB per buffer] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from giql_amd.engine import HipEngine  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = HipEngine(0)
shapes = [(u, nt, per_cu) for u in (1, 2, 4, 8) for nt in (0, 1) for per_cu in (4, 8, 16, 32)]
for rnd in range(2):      # twice: the boxes need a moment to reach their clocks
    pr = eng.stream_probe(nbytes=mib << 20, reps=reps, shapes=shapes)
print(f"buffers of {mib} MiB, {reps} launches per shape")
for mode in ("read", "write", "copy"):
    print(f"--- {mode}: best {pr[mode]} GB/s")
    for u in (1, 2, 4, 8):
        row = []
        for nt in (0, 1):
            for per_cu in (4, 8, 16, 32):
                pol = "nt" if nt else "dflt"
                row.append("%8.1f" % pr["shapes"][f"{mode}/x{u}/{pol}/{per_cu}perCU"])
        print(f"  x{u} in flight | dflt 4/8/16/32 per CU: {' '.join(row[:4])} | nt: {' '.join(row[4:])}")
print(f"--- hipMemcpyDtoDAsync: {pr['memcpy_d2d']} GB/s (read + written)")
print(f"--- round 3's probe (k_copy16, one load in flight, n_cu x 8 blocks): {eng.copy_probe():.1f} GB/s")
print(json.dumps({k: pr[k] for k in ("read", "write", "copy", "memcpy_d2d")}))
