#!/usr/bin/env python3
"""Multi-pass integer BoxBlur: the horizontal passes through the one-launch small-radius kernel (default) against one ring-kernel row pass each (VSZIP_RT_NO_HSMALL=1), Gpixel/s."""
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
CASES = [(1, 2, 1, 2), (3, 2, 3, 2), (2, 3, 2, 3), (5, 3, 5, 3), (10, 2, 10, 2), (13, 5, 13, 5), (3, 2, 0, 0), (13, 5, 0, 0)]
for dt in (np.uint8, np.uint16):
    for w, h, frames in ((1920, 1080, 64), (3840, 2160, 16)):
        base = [fx.tiled_natural(s, dt, p) for p, s in enumerate([(h, w), (h // 2, w // 2), (h // 2, w // 2)])]
        srcs = [d.upload(np.ascontiguousarray(np.roll(b, 3 * f, axis=1))) for f in range(frames) for b in base]
        dsts = [d.empty(b.shape[0], b.shape[1], b.dtype) for f in range(frames) for b in base]
        table = d.plane_table(srcs, dsts)
        row = []
        for args in CASES:
            out = []
            for opt in (0, 1):
                with d.options(VSZIP_RT_NO_HSMALL=opt):
                    step = lambda: d.boxblur_table(dt, table, *args)
                    _, region_ms, *_ = timed.run(step, 4, 1)
                out.append(frames * w * h * 1.5 * 4 / (region_ms * 1e-3) / 1e9)
            row.append(f"{args[0]}x{args[1]}/{args[2]}x{args[3]}: {out[0]:5.0f} / {out[1]:5.0f}")
        print(f"{dt.__name__:7s} {w}x{h}: " + " | ".join(row), flush=True)
        del srcs, dsts
