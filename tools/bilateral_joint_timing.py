#!/usr/bin/env python3
"""GPU box: vszip.Bilateral with a `ref` clip (joint), 64 x 1080p YUV420P16 per call, fps."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch  # noqa: F401
import bench
import vszip_amd

dev = vszip_amd.Device(0)
timed = bench.Timed(dev, dev.sync)
dt_ = np.float32 if "f32" in sys.argv else np.uint16
base = bench.natural_frame(bench.W1080, bench.H1080)
if dt_ == np.float32:
    base = [(p.astype(np.float32) / 65535.0) for p in base]
cfg = dev.bilateral_cfg([2], [2], yuv=True, ssw=1, ssh=1, hist_len=65536)
srcs, refs, dsts, idx = [], [], [], []
for f in range(64):
    for i, p in enumerate(base):
        srcs.append(dev.upload(np.roll(p, f * 13, axis=1)))
        refs.append(dev.upload(np.roll(p, f * 13 + 2, axis=1)))
        dsts.append(dev.empty(p.shape[0], p.shape[1], p.dtype))
        idx.append(i)
for joint in (True, False):
    fn = (lambda: dev.bilateral(srcs, dsts, cfg, idx, refs=refs)) if joint else (lambda: dev.bilateral(srcs, dsts, cfg, idx))
    dt, _, _, _ = timed.run(fn, 10, 2)
    print("joint" if joint else "plain", dt_.__name__, round(64 * 10 / dt), "fps", flush=True)
