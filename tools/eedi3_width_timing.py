#!/usr/bin/env python3
"""EEDI3 dh=1 on float planes of several line widths (the vertical-consistency chain kernel changes with the width: LDS rings up to kVcLdsMaxL
columns, the global-memory chain up to 4096, the plain one beyond): ms per call of 16 planes and Mpixel/s of output."""
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
for w, h in [(1920, 1080), (2048, 1080), (2160, 1080), (2560, 1080), (3840, 1080)]:
    base = np.ascontiguousarray(fx.tiled_natural((h, w), np.float32, 0))
    srcs = [d.upload(np.roll(base, 7 * f, axis=1)) for f in range(16)]
    step = lambda: d.eedi3(srcs, 1, dh=True)
    _, region_ms, *_ = timed.run(step, 5, 2)
    print(f"w={w} h={h}: {region_ms / 5:8.3f} ms per 16 planes, {16 * w * 2 * h / (region_ms / 5 * 1e-3) / 1e6:9.0f} Mpx/s out", flush=True)
