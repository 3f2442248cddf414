#!/usr/bin/env python3
"""GPU box: the ring kernel's band length (VSZIP_RING_PERIODS, read at every launch) on a FAST and on a SLOW placement."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401

import bench
import vszip_amd

dev = vszip_amd.Device(0)
base = bench.make_frame(0, bench.W4K, bench.H4K)
planes = [np.roll(p, f * 17 + 1, axis=1) for f in range(64) for p in base]
shapes = [p.shape for p in planes]
src = bench.Arena(dev, shapes, np.uint16, 1)
for a, d in zip(planes, src.planes):
    a = np.ascontiguousarray(a)
    dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * 2, a.ctypes.data, a.strides[0], a.shape[1] * 2, a.shape[0]))
dev.sync()


def run(dst, n=30):
    table = dev.plane_table(src.planes, dst.planes)
    for _ in range(4):
        dev.boxblur_table(np.uint16, table, 13, 1, 13, 1)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        dev.boxblur_table(np.uint16, table, 13, 1, 13, 1)
    dev.sync()
    return (time.perf_counter() - t0) / n * 1e6


cands = [bench.Arena(dev, shapes, np.uint16, 100 + k) for k in range(40)]
t = [run(c, 8) for c in cands]
fast, slow = cands[int(np.argmin(t))], cands[int(np.argmax(t))]
print("model's choice: fast %.1f slow %.1f" % (run(fast), run(slow)))
for target in (9, 12, 13, 15, 18, 19, 22, 24, 26, 36):
    os.environ["VSZIP_RING_PERIODS"] = str(target)
    print("periods %2d: fast %.1f slow %.1f" % (target, run(fast), run(slow)), flush=True)
