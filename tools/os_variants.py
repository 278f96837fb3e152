#!/usr/bin/env python3
"""Tuning aid: time the onesweep block-shape variants (GIQL_HIP_OS_VARIANT) on the
headline workload and check each against the oracle on a small input."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {6: "1024x8-all", 0: "1024x8 default", 1: "512x8", 2: "512x16", 3: "256x16", 4: "1024x4", 5: "1024x12"}
CHECK = r"""
import numpy as np, sys
sys.path.insert(0, %r)
from giql_amd.engine import HipEngine, DeviceSide
from oracle import pyoracle as ora
r = np.random.default_rng(5)
def side(n):
    ch = r.integers(0, 6, n).astype(np.int32); st = r.integers(0, 30_000_000, n).astype(np.int32)
    ln = r.integers(1, 900, n).astype(np.int32); return ora.Side(ch, st, st + ln)
a, b = side(150_000), side(260_000)
e = HipEngine(0)
ra, rb = e.inner_join(DeviceSide.from_numpy(a.chrom, a.start, a.end), DeviceSide.from_numpy(b.chrom, b.start, b.end), 6)
ok = np.array_equal(ora.sort_pairs(ra.cpu().numpy(), rb.cpu().numpy()), ora.sort_pairs(*ora.c_inner(a, b, "sweep")))
print("PARITY", ok)
""" % ROOT

for v in (int(x) for x in (sys.argv[1:] or NAMES)):
    env = dict(os.environ, GIQL_HIP_OS_VARIANT=str(v))
    chk = subprocess.run([sys.executable, "-c", CHECK], env=env, capture_output=True, text=True)
    parity = "PARITY True" in chk.stdout
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(v, NAMES[v], "FAILED", out.stderr[-300:])
        continue
    d = json.loads(line[-1])
    ph = d["roofline"]["phase_ms"]
    print(f"variant {v} {NAMES[v]:8s} parity={parity} ms_per_step={d['ms_per_step']:.3f} sort_scatter={ph.get('sort_scatter')}", flush=True)
