#!/usr/bin/env python3
"""Measured HBM copy bandwidth on this GPU ("verify on the box with a
copy kernel and report fraction of measured copy BW as well").  torch's d2d copy,
read-only (sum) and write-only (fill) at several sizes; GB/s counts bytes read + written."""
import json
import sys

import torch


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    out = {}
    for mb in (64, 400, 1600, 6400):
        n = mb * 1024 * 1024 // 4
        x = torch.empty(n, dtype=torch.int32, device="cuda").random_(0, 1000)
        y = torch.empty_like(x)
        t_copy = timed(lambda: y.copy_(x))
        t_fill = timed(lambda: y.fill_(7))
        t_read = timed(lambda: x.sum())
        out[f"{mb}MB"] = {"copy_GBps": round(2 * n * 4 / t_copy / 1e6, 1),
                          "fill_GBps": round(n * 4 / t_fill / 1e6, 1),
                          "read_GBps": round(n * 4 / t_read / 1e6, 1)}
        del x, y
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
