#!/usr/bin/env python3
"""GPU box: Bilateral 1080p YUV420P8 (the most common clip format), 64 frames per call, at several parameter sets."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, bench, vszip_amd, fixtures as fx
dev = vszip_amd.Device(0)
base = [fx.tiled_natural(s, np.uint8, p) for p, s in enumerate(bench.yuv420_shapes(1920, 1080))]
F = 64
for sS, sR in ((2, 2), (2, 0.02), (3, 0.02), (1, 0.02)):
    cfg = dev.bilateral_cfg([sS], [sR], yuv=True, ssw=1, ssh=1, hist_len=256)
    srcs, dsts, idx = [], [], []
    for f in range(F):
        for i, p in enumerate(base):
            srcs.append(dev.upload(np.roll(p, f * 13, axis=1))); dsts.append(dev.empty(p.shape[0], p.shape[1], p.dtype)); idx.append(i)
    for _ in range(2): dev.bilateral(srcs, dsts, cfg, idx, peak=255.0)
    dev.sync(); t = time.perf_counter()
    for _ in range(5): dev.bilateral(srcs, dsts, cfg, idx, peak=255.0)
    dev.sync(); dt = (time.perf_counter() - t) / 5
    print(f"u8 sigmaS={sS} sigmaR={sR}: radius/step luma {cfg[0].radius}/{cfg[0].step} chroma {cfg[1].radius}/{cfg[1].step}: {F / dt:9.0f} fps", flush=True)
    dev.bilateral_free(cfg); del srcs, dsts
