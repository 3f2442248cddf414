#!/usr/bin/env python3
"""GPU box: does the headline BoxBlur launch depend on HOW its 384 planes were allocated? Per-plane allocations (what
bench.py does) against one arena for all sources and one for all destinations. Prints us per launch for each layout,
interleaved over a few rounds."""
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


def per_plane():
    srcs, dsts = [], []
    for f in range(frames):
        for p in base:
            srcs.append(dev.upload(np.roll(p, f * 17 + 1, axis=1)))
            dsts.append(dev.empty(p.shape[0], p.shape[1], p.dtype))
    return srcs, dsts, None


rng = np.random.default_rng(1)


def arena(align, skew=0):
    """planes packed at `align`; skew: plane k starts (k * skew) % 1 MiB further on"""
    shapes = [p.shape for p in base] * frames
    offs, total = [], 0
    for k, (h, w) in enumerate(shapes):
        total = (total + align - 1) // align * align
        o = total + ((k * skew) % (1 << 20) if skew >= 0 else int(rng.integers(0, 4096)) * 256)
        offs.append(o)
        total = o + h * w * 2
    slabs = []
    for _ in range(2):
        p = C.c_void_p()
        dev.check(dev.lib.vszip_dev_alloc(dev.ctx, total + 256, C.byref(p)))
        slabs.append(p.value)
    srcs = [DevPlane(dev, slabs[0] + o, w, h, w, np.uint16, own=False) for o, (h, w) in zip(offs, shapes)]
    dsts = [DevPlane(dev, slabs[1] + o, w, h, w, np.uint16, own=False) for o, (h, w) in zip(offs, shapes)]
    for f in range(frames):
        for i, p in enumerate(base):
            a = np.ascontiguousarray(np.roll(p, f * 17 + 1, axis=1))
            d = srcs[f * 3 + i]
            dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * 2, a.ctypes.data, a.strides[0], a.shape[1] * 2, a.shape[0]))
    dev.sync()
    return srcs, dsts, slabs


def measure(srcs, dsts, n=300):
    table = dev.plane_table(srcs, dsts)
    for _ in range(20):
        dev.boxblur_table(np.uint16, table, radius, 1, radius, 1)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        dev.boxblur_table(np.uint16, table, radius, 1, radius, 1)
    dev.sync()
    return (time.perf_counter() - t0) / n * 1e6


layouts = {"per-plane allocations": per_plane(), "two arenas, planes 2 MiB aligned": arena(2 << 20)}
for i, sk in enumerate([int(x) for x in (sys.argv[1:] or ['16640'])]):
    layouts[f"#{i} 2 MiB aligned + " + (f"k x {sk} B" if sk >= 0 else "random x 256 B")] = arena(2 << 20, sk)
pp = layouts["per-plane allocations"]
print("per-plane pointers mod 2 MiB (first 9 src, first 3 dst):", [hex(x.ptr % (2 << 20)) for x in pp[0][:9]], [hex(x.ptr % (2 << 20)) for x in pp[1][:3]])
print("distinct (ptr mod 2 MiB) over the 384 planes:", len({x.ptr % (2 << 20) for x in pp[0] + pp[1]}), " mod 64 KiB:", len({x.ptr % (64 << 10) for x in pp[0] + pp[1]}))
ref = dev.download(layouts["per-plane allocations"][1][5]) if False else None
for r in range(2):
    for name, (s, d, _) in layouts.items():
        print(f"{name:36s} {measure(s, d):8.1f} us/launch", flush=True)
a = dev.download(layouts["per-plane allocations"][1][100])
for name, (s, d, _) in layouts.items():
    assert np.array_equal(dev.download(d[100]), a), name
print("outputs identical")
