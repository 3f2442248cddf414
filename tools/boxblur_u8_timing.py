"""Development: BoxBlur r=13 on 3840x2160 YUV420P8 (64 frames per launch), kernel-probe timing."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401

import bench
import fixtures as fx
import vszip_amd

dev = vszip_amd.Device(0)
timed = bench.Timed(dev, lambda: None)
for dt in (np.uint8, np.uint16):
    base = [fx.splitmix64_plane(p, s, dt) for p, s in enumerate(bench.yuv420_shapes(bench.W4K, bench.H4K))]
    srcs, dsts = [], []
    for f in range(64):
        for pl in base:
            srcs.append(dev.upload(np.roll(pl, f * 17 + 1, axis=1)))
            dsts.append(dev.empty(pl.shape[0], pl.shape[1], pl.dtype))
    table = dev.plane_table(srcs, dsts)
    for radius in (13, 3):
        step = lambda: dev.boxblur_table(dt, table, radius, 1, radius, 1)
        dt_s, region_ms, dom_ms, launches = timed.run(step, 10, 2)
        nbytes = 2 * sum(a.size * a.itemsize for a in base) * 64
        print(f"{np.dtype(dt).name} r={radius}: {64 * 10 / dt_s:9.0f} fps, kernel {dom_ms / launches * 1e3:7.1f} us/launch, {nbytes / (dom_ms / launches * 1e-3) / 8e12:.3f} of HBM peak", flush=True)
    del srcs, dsts, table
dev.close()
