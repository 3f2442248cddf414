#!/usr/bin/env python3
"""GPU box: vszip_xpsnr_wsse_batch on 1080p / 4K YUV420 at 8 and 10 bits, 64 / 16 frames per call (whole call and the strip kernel's share of
the HBM peak by the bench's own byte count: org + rec + one previous luma). One line per case; the last line is a JSON summary."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401

import bench
import fixtures as fx
import vszip_amd

dev = vszip_amd.Device(0)
timed = bench.Timed(dev, dev.sync)
out = {}
for name, w, h, depth, batch in (("1080p8", 1920, 1080, 8, 64), ("1080p10", 1920, 1080, 10, 64), ("4k8", 3840, 2160, 8, 16), ("4k10", 3840, 2160, 10, 16)):
    dt = np.uint8 if depth == 8 else np.uint16
    peak = (1 << depth) - 1
    rng = np.random.default_rng(3)
    base = [fx.tiled_natural(s, dt, p) for p, s in enumerate(bench.yuv420_shapes(w, h))]
    if depth > 8:
        base = [(p >> (16 - depth)).astype(dt) if p.max() > peak else p for p in base]
    noise = [rng.integers(-3, 4, p.shape).astype(np.int32) for p in base]
    org = [[np.roll(p, 5 * f, axis=1) for p in base] for f in range(batch)]
    rec = [[np.clip(p.astype(np.int32) + np.roll(nz, f, axis=0), 0, peak).astype(dt) for p, nz in zip(fr, noise)] for f, fr in enumerate(org)]
    dorg = [[dev.upload(p) for p in fr] for fr in org]
    drec = [[dev.upload(p) for p in fr] for fr in rec]
    p1s = [dorg[f - 1][0] if f >= 1 else None for f in range(batch)]
    p2s = [dorg[f - 2][0] if f >= 2 else None for f in range(batch)]
    call = dev.xpsnr_batch_call(dorg, drec, p1s, p2s, depth=depth, frame_rate=24)
    dtb, _, dom_ms, launches = timed.run(call, 10, 2)
    fb = (2 * sum(s[0] * s[1] for s in bench.yuv420_shapes(w, h)) + w * h) * np.dtype(dt).itemsize
    us = dom_ms / launches * 1e3
    out[name] = {"frames_per_s": round(batch * 10 / dtb, 1), "strip_us": round(us, 1), "strip_frac": round(batch * fb / (us * 1e-6) / 8e12, 3)}
    print(name, out[name], flush=True)
    del dorg, drec
print(json.dumps(out))
