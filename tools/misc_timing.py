#!/usr/bin/env python3
"""Timings of paths no bench leg covers: Bilateral algorithm 1 (PBFIC), BoxBlur RT float."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch  # noqa
import fixtures as fx
import vszip_amd

dev = vszip_amd.Device(0)
def timeit(fn, n=5):
    fn(); dev.sync(); t = time.perf_counter()
    for _ in range(n): fn()
    dev.sync(); return (time.perf_counter() - t) / n

shapes = [(1080, 1920), (540, 960), (540, 960)]
planes = [fx.tiled_natural(s, np.uint16, p) for p, s in enumerate(shapes)]
for num in (4, 16):
    cfg = dev.bilateral_cfg([8], [0.1], algorithm=[1], pbficnum=[num], yuv=True, ssw=1, ssh=1, hist_len=65536)
    srcs = [dev.upload(p) for p in planes]; dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
    t = timeit(lambda: dev.bilateral(srcs, dsts, cfg, [0, 1, 2]))
    print(f"Bilateral alg1 PBFICnum={num} (chroma {cfg[1].pbficnum}) 1080p YUV420P16: {1/t:.1f} fps ({t*1e3:.2f} ms/frame)")
    dev.bilateral_free(cfg)
pf = [(p / 65535.0).astype(np.float32) for p in planes]
srcs = [dev.upload(p) for p in pf]; dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in pf]
t = timeit(lambda: dev.boxblur(srcs, dsts, 30, 1, 30, 1))
print(f"BoxBlur RT float r=30 1080p YUV420PS: {1/t:.1f} fps ({t*1e3:.2f} ms/frame)")

# BoxBlur r=13 on 4K YUV420P8 (8-bit video is the common case)
import bench as _b
shapes4k = _b.yuv420_shapes(3840, 2160)
base8 = [(fx.splitmix64_plane(50 + p, s, np.uint16) >> 8).astype(np.uint8) for p, s in enumerate(shapes4k)]
srcs, dsts = [], []
for f in range(32):
    for p in base8:
        srcs.append(dev.upload(np.roll(p, f, 1))); dsts.append(dev.empty(p.shape[0], p.shape[1], np.uint8))
table = dev.plane_table(srcs, dsts)
t = timeit(lambda: dev.boxblur_table(np.uint8, table, 13, 1, 13, 1), n=20)
fb = 2 * sum(p.nbytes for p in base8) * 32
print(f"BoxBlur r=13 4K YUV420P8: {32/t:.0f} fps, {fb/t/1e9:.0f} GB/s = {fb/t/8e12:.3f} of peak")
