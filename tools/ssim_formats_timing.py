#!/usr/bin/env python3
"""GPU box: SSIMULACRA2 at 4K (16 pairs a call) from RGB24 / RGB48 / RGB30 / gamma-encoded RGBS clips: what the transfer-table gather of the
16-bit and float pre-stages costs against the 8-bit one (its 256-entry table sits in LDS)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import bench, fixtures as fx, vszip_amd
dev = vszip_amd.Device(0)
w, h, pairs = bench.W4K, bench.H4K, 16
rng = np.random.default_rng(1)
ref8 = [fx.tiled_natural((h, w), np.uint8, p) for p in range(3)]
dis8 = [np.clip(p.astype(np.int16) + rng.integers(-5, 6, p.shape, dtype=np.int16), 0, 255).astype(np.uint8) for p in ref8]
cases = {
  "RGB24": (np.uint8, 8, lambda p: p),
  "RGB48": (np.uint16, 16, lambda p: p.astype(np.uint16) * 257),
  "RGB30": (np.uint16, 10, lambda p: (p.astype(np.uint16) << 2)),
  "RGBS gamma": (np.float32, 32, lambda p: (p.astype(np.float32) / 255.0)),
}
for name, (dt, bits, cv) in cases.items():
    ref, dis = [cv(p) for p in ref8], [cv(p) for p in dis8]
    fmt = dev.ssim_source("RGB", dt, bits, True)
    r, d = [], []
    for p in range(pairs):
        r += [dev.upload(np.roll(x, p * 7, axis=1)) for x in ref]
        d += [dev.upload(np.roll(x, p * 7, axis=1)) for x in dis]
    dev.ssimulacra2_src(fmt, r, d); dev.sync()
    t0 = time.perf_counter()
    for _ in range(5): dev.ssimulacra2_src(fmt, r, d)
    dev.sync()
    print(name, round(80 / (time.perf_counter() - t0), 1), "pairs/s", flush=True)
    del r, d
