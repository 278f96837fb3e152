#!/usr/bin/env python3
"""Phase timeline of the onesweep sort pass (diagnostic).  Needs a -DGIQL_OS_TIMELINE build:
    hipcc ... -DGIQL_OS_TIMELINE -o giql_amd/libv_tl.so giql_amd/csrc/giql_hip.hip
    GIQL_HIP_LIB=$PWD/giql_amd/libv_tl.so python tools/os_timeline.py
Thread 0 of every block stamps the 100 MHz wall clock at the phase boundaries of its tile; the
stamps left after one INNER plan of the headline workload are those of the LAST pass of the
100M-row side's sort.  Prints per-phase medians (microseconds) and how the tiles overlap in time."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from giql_amd import synth, _lib
from giql_amd.engine import DeviceSide, HipEngine

n_a, n_b = int(os.environ.get("TL_NA", 10_000_000)), int(os.environ.get("TL_NB", 100_000_000))
eng = HipEngine(0)
a = DeviceSide.from_numpy(*synth.make_table(n_a, 5, "peaks"))
b = DeviceSide.from_numpy(*synth.make_table(n_b, 6, "reads"))
for _ in range(3):
    eng.inner_plan(a, b, 24)
torch.cuda.synchronize()
L = _lib.load() if hasattr(_lib, "load") else eng._L
n_tiles = (n_b + 8191) // 8192
buf = np.zeros(n_tiles * 16, dtype=np.uint64)
fn = L.giql_hip_debug_timeline
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
fn.restype = ctypes.c_int
assert fn(buf.ctypes.data, buf.size) == 0
raw = buf.reshape(n_tiles, 16).astype(np.int64)[:-1]           # drop the partial last tile
t = raw[:, :12]
us = (t - t[:, :1].min()) / 100.0                             # 100 MHz -> microseconds since the first block started
names = ["start->loads issued", "loads arrive", "rank (wave 0)", "barrier: all waves ranked", "wave bases (16 LDS rmw)",
         "scan + publish + 2 barriers", "positions + key staging", "LDS rounds (keys, payload -> registers)",
         "look-back walk (thread 0's digit)", "barrier + stores issued", "drain (wave 0's stores acked)"]
out = {"tiles": int(t.shape[0]), "pass_us": round(float(us[:, 11].max()), 1)}
d = np.diff(us, axis=1)
out["phases_us_median_p90"] = {names[k]: [round(float(np.median(d[:, k])), 2), round(float(np.percentile(d[:, k], 90)), 2)]
                               for k in range(11)}
life = us[:, 11] - us[:, 0]
out["tile_lifetime_us_median_p90"] = [round(float(np.median(life)), 2), round(float(np.percentile(life, 90)), 2)]
# how many tiles are alive at once (sampled), and when tiles start
grid = np.linspace(0, us[:, 11].max(), 200)
alive = [(int(((us[:, 0] <= g) & (us[:, 11] > g)).sum())) for g in grid]
out["tiles_alive_median_max"] = [int(np.median(alive)), int(max(alive))]
out["sum_of_lifetimes_over_pass_us"] = round(float(life.sum() / us[:, 11].max()), 1)   # average concurrency
polls, walked = raw[1:, 12], raw[1:, 13]
out["lookback_poll_rounds_median_p90_max"] = [float(np.median(polls)), float(np.percentile(polls, 90)), int(polls.max())]
out["lookback_tiles_walked_median_p90_max"] = [float(np.median(walked)), float(np.percentile(walked, 90)), int(walked.max())]
print(json.dumps(out, indent=1))
