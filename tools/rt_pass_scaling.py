#!/usr/bin/env python3
"""README bench 3's two halves by pass count: 32 x 1080p YUV420P16, r = 13, horizontal-only and vertical-only with 1 ... 5 passes (us per call)."""
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
R = int(sys.argv[1]) if len(sys.argv) > 1 else 13
for dt, w, h, frames in ((np.uint16, 1920, 1080, 32), (np.uint16, 3840, 2160, 8), (np.uint8, 1920, 1080, 64)):
    base = [fx.tiled_natural(s, dt, p) for p, s in enumerate([(h, w), (h // 2, w // 2), (h // 2, w // 2)])]
    srcs = [d.upload(np.ascontiguousarray(np.roll(b, 3 * f, axis=1))) for f in range(frames) for b in base]
    dsts = [d.empty(b.shape[0], b.shape[1], b.dtype) for f in range(frames) for b in base]
    table = d.plane_table(srcs, dsts)
    out = []
    for args in [(R, p, 0, 0) for p in range(1, 6)] + [(0, 0, R, p) for p in range(1, 6)] + [(R, 5, R, 5)]:
        step = lambda: d.boxblur_table(dt, table, *args)
        _, region_ms, *_ = timed.run(step, 6, 2)
        out.append(f"h{args[0]}x{args[1]} v{args[2]}x{args[3]}: {region_ms / 6 * 1e3:6.0f}")
    print(f"{np.dtype(dt).name} {w}x{h} x{frames} r={R}: " + " | ".join(out), flush=True)
    del srcs, dsts
