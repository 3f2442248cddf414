#!/usr/bin/env python3
"""Round 4: the headline batch (64 x 4K YUV420P16, BoxBlur r=13) on arenas {placed by the allocator, plain} x {planes on 2 MiB boundaries,
planes at 2 MiB + a random multiple of 256 B}; a fresh context per configuration (the allocator's state is per context)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import vszip_amd  # noqa: E402

base = bench.make_frame(0, bench.W4K, bench.H4K)
planes = [np.roll(p, f * 17 + 1, axis=1) for f in range(64) for p in base]
shapes = [p.shape for p in planes]
for rnd in range(2):
    for placement in (1, 0):
        for aligned in (True, False):
            dev = vszip_amd.Device(0)
            dev.set_option("VSZIP_PLACEMENT", placement)
            timed = bench.Timed(dev, lambda: None)
            a = bench.Arena(dev, shapes, np.uint16, 1)
            b = bench.Arena(dev, shapes, np.uint16, 2)
            if aligned:
                for ar in (a, b):
                    ar.offs = [o // (2 << 20) * (2 << 20) for o in ar.offs]
                    ar.bind(ar.ptr)
            for h, d in zip(planes, a.planes):
                h = np.ascontiguousarray(h)
                dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * 2, h.ctypes.data, h.strides[0], h.shape[1] * 2, h.shape[0]))
            dev.sync()
            table = dev.plane_table(a.planes, b.planes)
            step = lambda: dev.boxblur_table(np.uint16, table, 13, 1, 13, 1)
            _, _, dom_ms, n = timed.run(step, 40, 3)
            us = dom_ms / n * 1e3
            pa, pb = dev.placement_info(a.ptr), dev.placement_info(b.ptr)
            print(f"round {rnd} placement {placement} planes {'on 2 MiB boundaries' if aligned else '2 MiB + random 256 B  '}: {us:7.1f} us  {3185049600 / (us * 1e-6) / 8e12:.3f}   "
                  f"probe src {pa['bytes_per_second'] / 1e12:.2f} dst {pb['bytes_per_second'] / 1e12:.2f} TB/s, walks {pb['walks']}, probed {pb['probed']}, exhausted {pb['exhausted']}", flush=True)
            a.free()
            b.free()
            dev.close()
