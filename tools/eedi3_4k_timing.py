#!/usr/bin/env python3
"""EEDI3 field=1 (same height) and dh=1 on 3840x2160 YUV420PS frames, 8 and 16 frames per call, outputs preallocated by the first call's planes being reused
is not possible through this binding, so the figures include the output planes' allocation: compare builds, not absolute rates."""
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

from vszip_amd import capi  # noqa: E402

names = sys.argv[1:] or ["base"]
for rnd in range(2):
  for name in names:
    capi.LIB_PATH = ROOT / ("vapoursynth-zip_amd/libvszip_hip.so" if name == "base" else f"tools/ab/{name}.so")
    capi._lib = None
    d = vszip_amd.Device(0)
    timed = bench.Timed(d, d.sync)
    print(f"-- {name}", flush=True)
    for nf in (8, 16):
        base = [np.ascontiguousarray(fx.tiled_natural(s, np.float32, p)) for p, s in enumerate([(2160, 3840), (1080, 1920), (1080, 1920)])]
        srcs = [d.upload(np.roll(pl, 7 * f, axis=1)) for f in range(nf) for pl in base]
        for kw in (dict(dh=False), dict(dh=True)):
            step = lambda: d.eedi3(srcs, 1, **kw)
            _, region_ms, *_ = timed.run(step, 4, 2)
            print(f"4K YUV420PS {nf} frames {kw}: {region_ms / 4:8.3f} ms per call, {nf / (region_ms / 4 * 1e-3):7.1f} fps", flush=True)
        del srcs
    d.close()
