#!/usr/bin/env python3
"""GPU box: Bilateral sigmaS=2 sigmaR=2 on 8K RGBS planes (the pipeline config's first stage), walk kernel against the tile kernel; and the 8K pipeline."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, bench, vszip_amd, fixtures as fx
dev = vszip_amd.Device(0)
planes = [np.ascontiguousarray(fx.tiled_natural((4320, 7680), np.float32, p)) for p in range(3)]
cfg = dev.bilateral_cfg([2], [2], yuv=False, ssw=0, ssh=0, hist_len=65536)
srcs = [dev.upload(p, 1) for p in planes for _ in range(2)]
dsts = [dev.empty(4320, 7680, np.float32, 1) for _ in srcs]
idx = [i // 2 for i in range(len(srcs))]
for rnd in range(2):
    for env in ("", "1"):
        if env: os.environ["VSZIP_BILATERAL_NO_WALK"] = "1"
        else: os.environ.pop("VSZIP_BILATERAL_NO_WALK", None)
        for _ in range(2): dev.bilateral(srcs, dsts, cfg, idx)
        dev.sync(); t = time.perf_counter()
        for _ in range(5): dev.bilateral(srcs, dsts, cfg, idx)
        dev.sync(); dt = (time.perf_counter() - t) / 5
        print("tile kernel" if env else "walk kernel", round(dt * 1e6), "us per 2 x 8K RGBS frames", round(2 / dt), "fps", flush=True)
