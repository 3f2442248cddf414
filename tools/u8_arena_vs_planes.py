#!/usr/bin/env python3
"""Round 4: why does the bench's u8 BoxBlur leg (planes inside two arenas) read 0.44 where the same launch on per-plane allocations reads 0.51?"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import fixtures as fx  # noqa: E402
import vszip_amd  # noqa: E402

dev = vszip_amd.Device(0)
timed = bench.Timed(dev, lambda: None)
for dt in (np.uint8, np.uint16):
    isz = np.dtype(dt).itemsize
    base = [fx.splitmix64_plane(p, s, dt) for p, s in enumerate(bench.yuv420_shapes(bench.W4K, bench.H4K))]
    planes = [np.roll(pl, f * 17 + 1, axis=1) for f in range(64) for pl in base]
    nbytes = 2 * sum(a.nbytes for a in base) * 64

    def time_it(name, srcs, dsts):
        table = dev.plane_table(srcs, dsts)
        step = lambda: dev.boxblur_table(dt, table, 13, 1, 13, 1)
        _, _, dom_ms, n = timed.run(step, 30, 3)
        us = dom_ms / n * 1e3
        print(f"{np.dtype(dt).name} {name:44s} {us:7.1f} us  {nbytes / (us * 1e-6) / 8e12:.3f}", flush=True)

    srcs = [dev.upload(p) for p in planes]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
    time_it("one allocation per plane", srcs, dsts)
    del srcs, dsts
    for label, placement, offsets in (("arenas, placed, random 256-B offsets", 1, True), ("arenas, unplaced, random 256-B offsets", 0, True), ("arenas, placed, 2 MiB boundaries", 1, False),
                                      ("arenas, unplaced, 2 MiB boundaries", 0, False)):
        with dev.options(VSZIP_PLACEMENT=placement):
            a = bench.Arena(dev, [p.shape for p in planes], dt, 1)
            b = bench.Arena(dev, [p.shape for p in planes], dt, 2)
        if not offsets:
            for ar in (a, b):
                ar.offs = [o // (2 << 20) * (2 << 20) for o in ar.offs]
                ar.bind(ar.ptr)
        for h, d in zip(planes, a.planes):
            h = np.ascontiguousarray(h)
            dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * isz, h.ctypes.data, h.strides[0], h.shape[1] * isz, h.shape[0]))
        dev.sync()
        time_it(label + f" (probe {dev.placement_info(a.ptr)['bytes_per_second'] / 1e12:.2f} / {dev.placement_info(b.ptr)['bytes_per_second'] / 1e12:.2f} TB/s)", a.planes, b.planes)
        a.free()
        b.free()
    dev.trim()
dev.close()
