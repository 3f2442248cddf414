#!/usr/bin/env python3
"""GPU box: BoxBlur launch time against WHERE in VRAM the destination arena lies: one source arena, then destination
arenas allocated one after another and all kept (so each lies further into VRAM), timed one by one.
usage: slab_ballast.py <count> [ballast GiB before the first]"""
import ctypes as C
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
frames, radius = 64, 13
base = bench.make_frame(0, bench.W4K, bench.H4K)
shapes = [p.shape for p in base] * frames
count = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ballast_gib = float(sys.argv[2]) if len(sys.argv) > 2 else 0


def launch_us(src, dst, n=20):
    table = dev.plane_table(src.planes, dst.planes)
    for _ in range(4):
        dev.boxblur_table(np.uint16, table, radius, 1, radius, 1)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        dev.boxblur_table(np.uint16, table, radius, 1, radius, 1)
    dev.sync()
    return (time.perf_counter() - t0) / n * 1e6


src = bench.Arena(dev, shapes, np.uint16, 1)
for i, d in enumerate(src.planes):
    a = np.ascontiguousarray(np.roll(base[i % 3], (i // 3) * 17 + 1, axis=1))
    dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * 2, a.ctypes.data, a.strides[0], a.shape[1] * 2, a.shape[0]))
dev.sync()
hold = []
if ballast_gib:
    p = C.c_void_p()
    dev.check(dev.lib.vszip_dev_alloc(dev.ctx, int(ballast_gib * (1 << 30)), C.byref(p)))
    hold.append(p)
cum = src.nbytes / 2**30 + ballast_gib
for k in range(count):
    d = bench.Arena(dev, shapes, np.uint16, 100 + k)
    hold.append(d)
    t = launch_us(src, d)
    print(f"dst arena {k:2d}  after {cum:6.1f} GiB  va {d.ptr:#x}: {t:6.1f} us", flush=True)
    cum += d.nbytes / 2**30
