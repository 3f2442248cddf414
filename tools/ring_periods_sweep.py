#!/usr/bin/env python3
"""Round 4: the headline launch on UNPLACED arenas (VSZIP_PLACEMENT=0: what a device without fast regions gives) against the ring kernel's band length
(VSZIP_RING_PERIODS, a -DVSZIP_DEV_VARIANTS build): does another number of concurrent row streams do better in a slow region?"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import vszip_amd  # noqa: E402
from vszip_amd import capi  # noqa: E402

names = sys.argv[1:] or ["base"]
devs = {}
for nm in names:
    capi.LIB_PATH = ROOT / ("vapoursynth-zip_amd/libvszip_hip.so" if nm == "base" else f"tools/ab/{nm}.so")
    capi._lib = None
    devs[nm] = vszip_amd.Device(0)
dev = devs[names[0]]
name = "variants"
base = bench.make_frame(0, bench.W4K, bench.H4K)
planes = [np.roll(p, f * 17 + 1, axis=1) for f in range(64) for p in base]
shapes = [p.shape for p in planes]
timed = bench.Timed(dev, lambda: None)
for placement in (0, 1):
    dev.set_option("VSZIP_PLACEMENT", placement)
    a = bench.Arena(dev, shapes, np.uint16, 1)
    b = bench.Arena(dev, shapes, np.uint16, 2)
    for h, d in zip(planes, a.planes):
        h = np.ascontiguousarray(h)
        dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * 2, h.ctypes.data, h.strides[0], h.shape[1] * 2, h.shape[0]))
    dev.sync()
    table = dev.plane_table(a.planes, b.planes)
    step = lambda: dev.boxblur_table(np.uint16, table, 13, 1, 13, 1)
    out = []
    for nm in names:
        d2 = devs[nm]
        t2 = bench.Timed(d2, lambda: None)
        tab2 = d2.plane_table(a.planes, b.planes)
        _, _, dom_ms, n = t2.run(lambda: d2.boxblur_table(np.uint16, tab2, 13, 1, 13, 1), 30, 3)
        us = dom_ms / n * 1e3
        out.append(f"{nm}: {us:.0f} us ({3185049600 / (us * 1e-6) / 8e12:.3f})")
    print(f"{name} placement {placement} (dst probe {dev.placement_info(b.ptr)['bytes_per_second'] / 1e12:.2f} TB/s): " + " | ".join(out), flush=True)
    a.free()
    b.free()
    dev.trim()
