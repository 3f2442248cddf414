#!/usr/bin/env python3
"""GPU box: Bilateral 1080p YUV420P16, 64 frames per call, at the BASELINE's (2, 2), the plugin bench's (2, 0.02) and the filter's DEFAULT (3, 0.02) parameters."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, bench, vszip_amd
dev = vszip_amd.Device(0)
base = bench.natural_frame(1920, 1080)
F = 64
for sS, sR in ((2, 2), (2, 0.02), (3, 0.02), (3, 2), (1, 0.02), (2, 0.1), (2, 0.3), (2, 0.5)):
    cfg = dev.bilateral_cfg([sS], [sR], yuv=True, ssw=1, ssh=1, hist_len=65536)
    srcs, dsts, idx = [], [], []
    for f in range(F):
        for i, p in enumerate(base):
            srcs.append(dev.upload(np.roll(p, f * 13, axis=1))); dsts.append(dev.empty(p.shape[0], p.shape[1], p.dtype)); idx.append(i)
    for _ in range(2): dev.bilateral(srcs, dsts, cfg, idx)
    dev.sync(); t = time.perf_counter()
    for _ in range(5): dev.bilateral(srcs, dsts, cfg, idx)
    dev.sync(); dt = (time.perf_counter() - t) / 5
    print(f"sigmaS={sS} sigmaR={sR}: radius/step luma {cfg[0].radius}/{cfg[0].step} chroma {cfg[1].radius}/{cfg[1].step} alg {cfg[0].algorithm}: {F / dt:9.0f} fps", flush=True)
    dev.bilateral_free(cfg); del srcs, dsts
