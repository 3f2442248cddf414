"""Development: BoxBlur r=13 on 1920x1080 YUV420P16 (BASELINE config 0's geometry), kernel-probe timing."""
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
base = [fx.splitmix64_plane(p, s, np.uint16) for p, s in enumerate(bench.yuv420_shapes(1920, 1080))]
for frames in (16, 64):
    srcs, dsts = [], []
    for f in range(frames):
        for pl in base:
            srcs.append(dev.upload(np.roll(pl, f * 17 + 1, axis=1)))
            dsts.append(dev.empty(pl.shape[0], pl.shape[1], pl.dtype))
    table = dev.plane_table(srcs, dsts)
    step = lambda: dev.boxblur_table(np.uint16, table, 13, 1, 13, 1)
    dt_s, region_ms, dom_ms, launches = timed.run(step, 20, 3)
    nbytes = 2 * sum(a.size * a.itemsize for a in base) * frames
    print(f"1080p u16 r=13, {frames} frames/launch: {frames * 20 / dt_s:9.0f} fps, kernel {dom_ms / launches * 1e3:7.1f} us/launch, {nbytes / (dom_ms / launches * 1e-3) / 8e12:.3f} of HBM peak", flush=True)
dev.close()
