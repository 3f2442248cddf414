#!/usr/bin/env python3
"""BoxBlur r=13 on 8-bit YUV 4:2:0 frames of several sizes (about 200 Mpixel a call): 16 against 8 pixels a lane (VSZIP_CT_U8_PX8), Gpixel/s from the stream clock, plane tables prebuilt."""
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
radii = [int(a) for a in sys.argv[1:]] or [13]
for w, h in [(1280, 720), (1920, 1080), (1920, 1088), (2560, 1440), (3840, 2160), (4096, 2160)]:
    frames = max(2, int(200e6 / (w * h * 1.5)))
    base = [fx.tiled_natural(s, np.uint8, p) for p, s in enumerate([(h, w), (h // 2, w // 2), (h // 2, w // 2)])]
    srcs = [d.upload(np.ascontiguousarray(np.roll(b, 3 * f, axis=1))) for f in range(frames) for b in base]
    dsts = [d.empty(b.shape[0], b.shape[1], b.dtype) for f in range(frames) for b in base]
    table = d.plane_table(srcs, dsts)
    for r in radii:
        out = []
        for px8 in (0, 1):
            with d.options(VSZIP_CT_U8_PX8=px8):
                step = lambda: d.boxblur_table(np.uint8, table, r, 1, r, 1)
                _, region_ms, *_ = timed.run(step, 6, 2)
            out.append(frames * w * h * 1.5 * 6 / (region_ms * 1e-3) / 1e9)
        print(f"{w}x{h} x{frames:3d} r={r:2d}: 16 px {out[0]:7.1f}   8 px {out[1]:7.1f} Gpx/s   ratio {out[0] / out[1]:.3f}", flush=True)
    del srcs, dsts
