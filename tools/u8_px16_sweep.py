#!/usr/bin/env python3
"""Round 4: BoxBlur CT on 64 x 4K YUV420P8, every radius, 16 pixels a lane against 8 (VSZIP_CT_U8_PX8), interleaved in one process."""
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

dev = vszip_amd.Device(0)
timed = bench.Timed(dev, lambda: None)
dt = np.uint8
base = [fx.splitmix64_plane(p, s, dt) for p, s in enumerate(bench.yuv420_shapes(bench.W4K, bench.H4K))]
srcs = [dev.upload(np.roll(pl, f * 17 + 1, axis=1)) for f in range(64) for pl in base]
dsts = [dev.empty(pl.shape[0], pl.shape[1], pl.dtype) for f in range(64) for pl in base]
table = dev.plane_table(srcs, dsts)
nbytes = 2 * sum(a.nbytes for a in base) * 64
for r in range(1, 23):
    res = {}
    for rnd in range(2):
        for opt in (0, 1):
            dev.set_option("VSZIP_CT_U8_PX8", opt)
            _, _, dom, n = timed.run(lambda: dev.boxblur_table(dt, table, r, 1, r, 1), 20, 3)
            res.setdefault(opt, []).append(dom / n * 1e3)
    a, b = min(res[0]), min(res[1])
    print(f"r {r:2d}: 16 px {a:6.1f} us ({nbytes / (a * 1e-6) / 8e12:.3f})   8 px {b:6.1f} us ({nbytes / (b * 1e-6) / 8e12:.3f})   16/8 = {a / b:.3f}", flush=True)
