#!/usr/bin/env python3
"""SSIMULACRA2 from linear RGBS planes over frame sizes (about 130 Mpixel of pairs per call): Gpixel/s of pair pixels — looking for sizes off the tuned path."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import fixtures as fx  # noqa: E402
import vszip_amd  # noqa: E402

d = vszip_amd.Device(0)
timed = bench.Timed(d, d.sync)
timed.prewarm_s = 0.2
rng = np.random.default_rng(1)
for w, h in [(640, 360), (854, 480), (1280, 720), (1366, 768), (1918, 1078), (1920, 1080), (1921, 1081), (2560, 1440), (3838, 2158), (3840, 2160), (4096, 2160), (7680, 4320)]:
    pairs = max(1, int(130e6 / (w * h)))
    ref = [np.ascontiguousarray(fx.tiled_natural((h, w), np.float32, p)) for p in range(3)]
    dis = [np.clip(p + rng.normal(0, 0.02, p.shape).astype(np.float32), 0, 1).astype(np.float32) for p in ref]
    r, s = [], []
    for k in range(pairs):
        r += [d.upload(np.roll(x, k * 7, axis=1), 1) for x in ref]
        s += [d.upload(np.roll(x, k * 7, axis=1), 1) for x in dis]
    step = lambda: d.ssimulacra2(r, s)
    _, region_ms, *_ = timed.run(step, 4, 1)
    print(f"{w}x{h}: {pairs:4d} pairs per call, {pairs * 4 / (region_ms * 1e-3):9.1f} pairs/s, {pairs * 4 * w * h / (region_ms * 1e-3) / 1e9:7.2f} Gpx/s", flush=True)
    del r, s
