#!/usr/bin/env python3
"""GPU box: SSIMULACRA2 at 1080p (what most comparisons run at), YUV420P8 and linear RGBS sources, 1 / 4 / 16 pairs per call, one context."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, bench, vszip_amd
dev = vszip_amd.Device(0)
for src in ("yuv420p8", "rgbs"):
    for pairs in (1, 4, 16):
        step, keep = (bench.setup_ssimulacra2_yuv420p8 if src == "yuv420p8" else bench.setup_ssimulacra2)(dev, 1920, 1080, pairs)
        for _ in range(3): step()
        n = max(5, 64 // pairs)
        t = time.perf_counter()
        for _ in range(n): step()
        dt = (time.perf_counter() - t) / n
        print(f"{src} {pairs:2d} pairs per call: {dt * 1e6:8.0f} us per call, {pairs / dt:8.0f} pairs/s", flush=True)
        del step, keep
