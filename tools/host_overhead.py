#!/usr/bin/env python3
"""Host time of a library call: 192 tiny planes (128 x 64 / 64 x 32, so the kernels take microseconds), wall clock per call with the stream drained every 20 calls.
BoxBlur goes through the prebuilt plane table (the C call alone); the other bindings marshal their arguments per call, so their figures are upper bounds."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import fixtures as fx  # noqa: E402
import vszip_amd  # noqa: E402

d = vszip_amd.Device(0)


def clock(name, step, n=200):
    for _ in range(10):
        step()
    d.sync()
    t0 = time.perf_counter()
    for i in range(n):
        step()
        if i % 20 == 19:
            d.sync()
    d.sync()
    wall = (time.perf_counter() - t0) / n * 1e6
    # the calls alone: 20 back to back without a synchronise (the stream's queue takes them), best of five
    best = 1e9
    for _ in range(5):
        d.sync()
        t0 = time.perf_counter()
        for i in range(20):
            step()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
    d.sync()
    print(f"{name:46s} {wall:8.1f} us per call with the device, {best:8.1f} us the call alone", flush=True)


for dt in (np.uint8, np.uint16, np.float32):
    shapes = [(128, 64), (64, 32), (64, 32)]
    base = [fx.tiled_natural(s, dt, p) for p, s in enumerate(shapes)]
    srcs = [d.upload(b) for f in range(64) for b in base]
    dsts = [d.empty(b.shape[0], b.shape[1], b.dtype) for f in range(64) for b in base]
    table = d.plane_table(srcs, dsts)
    for args in ((13, 1, 13, 1), (2, 1, 2, 1), (30, 1, 30, 1), (5, 3, 5, 3)):
        if min(min(s) for s in shapes) <= 2 * args[0]:
            continue
        clock(f"boxblur {dt.__name__} {args} (table)", lambda: d.boxblur_table(dt, table, *args))
    if dt != np.float32:
        lo, hi = [16] * 192, [200] * 192
        clock(f"limiter {dt.__name__}", lambda: d.limiter(srcs, dsts, lo, hi))
    clock(f"plane_average {dt.__name__}", lambda: d.plane_average(srcs))
    cfg = d.bilateral_cfg([2], [2], yuv=True, ssw=1, ssh=1, hist_len=256 if dt == np.uint8 else 65536)
    idx = [i for f in range(64) for i in range(3)]
    peak = 255.0 if dt == np.uint8 else None
    clock(f"bilateral sigmaS=2 {dt.__name__}", lambda: d.bilateral(srcs, dsts, cfg, idx, peak=peak))
