#!/usr/bin/env python3
"""GPU box: SSIMULACRA2 scores (as hex doubles) of a fixed set of pairs - to compare library variants bit for bit."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch  # noqa: F401
import bench  # noqa: F401
import fixtures as fx
import vszip_amd

dev = vszip_amd.Device(0)
rng = np.random.default_rng(11)
out = []
for (h, w) in [(2160, 3840), (631, 313), (270, 480)]:
    ref = [np.ascontiguousarray(fx.tiled_natural((h, w), np.float32, p)) for p in range(3)]
    for sigma in (0.0, 0.002, 0.02, 0.3):
        dis = [np.clip(p + rng.normal(0, sigma, p.shape).astype(np.float32), 0, 1).astype(np.float32) for p in ref]
        r = [dev.upload(p) for p in ref]
        d = [dev.upload(p) for p in dis]
        out.append(dev.ssimulacra2(r, d)[0])
    flat = [np.full((h, w), 0.5, np.float32) for _ in range(3)]
    out.append(dev.ssimulacra2([dev.upload(p) for p in ref], [dev.upload(p) for p in flat])[0])
print(" ".join(float(x).hex() for x in out))
