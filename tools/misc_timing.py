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
