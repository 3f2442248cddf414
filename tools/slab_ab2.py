#!/usr/bin/env python3
"""GPU box: the headline BoxBlur launch with all 384 planes (sources and destinations) inside ONE allocation, so that
virtual offsets are physical offsets as far as the allocator keeps the arena contiguous. Arguments: layouts as
align:skew:gap — planes packed at `align` bytes, plane k shifted by (k * skew) % 1 MiB, destinations start `gap` bytes
after the (aligned) end of the sources. Every layout is allocated and measured twice (repeatability)."""
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
from vszip_amd.capi import DevPlane

dev = vszip_amd.Device(0)
frames, radius = 64, 13
base = bench.make_frame(0, bench.W4K, bench.H4K)
shapes = [p.shape for p in base] * frames
host = [np.ascontiguousarray(np.roll(p, 17 + 1, axis=1)) for p in base]


def build(align, skew, gap):
    offs, total = [], 0
    for rep in range(2):
        for k, (h, w) in enumerate(shapes):
            total = (total + align - 1) // align * align
            o = total + (k * skew) % (1 << 20)
            offs.append(o)
            total = o + h * w * 2
        if rep == 0:
            total = (total + (2 << 20) - 1) // (2 << 20) * (2 << 20) + gap
    p = C.c_void_p()
    dev.check(dev.lib.vszip_dev_alloc(dev.ctx, total + 256, C.byref(p)))
    n = len(shapes)
    srcs = [DevPlane(dev, p.value + o, w, h, w, np.uint16, own=False) for o, (h, w) in zip(offs[:n], shapes)]
    dsts = [DevPlane(dev, p.value + o, w, h, w, np.uint16, own=False) for o, (h, w) in zip(offs[n:], shapes)]
    for i, d in enumerate(srcs):
        a = host[i % 3]
        dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * 2, a.ctypes.data, a.strides[0], a.shape[1] * 2, a.shape[0]))
    dev.sync()
    return srcs, dsts, p.value


def measure(srcs, dsts, n=200):
    table = dev.plane_table(srcs, dsts)
    for _ in range(10):
        dev.boxblur_table(np.uint16, table, radius, 1, radius, 1)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        dev.boxblur_table(np.uint16, table, radius, 1, radius, 1)
    dev.sync()
    return (time.perf_counter() - t0) / n * 1e6


for spec in sys.argv[1:]:
    align, skew, gap = (int(x) for x in spec.split(":"))
    res = []
    for trial in range(2):
        s, d, ptr = build(align, skew, gap)
        res.append((measure(s, d), ptr % (1 << 30)))
        dev.lib.vszip_dev_free(dev.ctx, ptr)
    print(f"align {align:8d} skew {skew:7d} gap {gap:8d}: " + "  ".join(f"{t:6.1f} us (va mod 1G {v:#x})" for t, v in res), flush=True)
