#!/usr/bin/env python3
"""GPU box: EEDI3 field=1 dh=1 1080p YUV420PS, one context, F frames per call, with and without the second (CU-masked) stream that
runs the tall planes' vertical-consistency chains beside the short planes' line kernel. Where does the overlap start to pay?"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
import fixtures as fx  # noqa: E402
import vszip_amd  # noqa: E402

dev = vszip_amd.Device(0)
base = [np.ascontiguousarray(fx.tiled_natural(s, np.float32, p)) for p, s in enumerate(bench.yuv420_shapes(1920, 1080))]
for frames in (1, 2, 4, 8, 16):
    srcs = []
    for f in range(frames):
        srcs += [dev.upload(np.roll(p, f * 11, axis=1)) for p in base]
    dsts = dev.eedi3(srcs, 1, dh=True)
    table = dev.plane_table(srcs, dsts)
    prm = bench._eedi3_params()
    row = []
    for env in ("", "1"):
        if env:
            os.environ["VSZIP_EEDI3_NO_OVERLAP"] = "1"
        else:
            os.environ.pop("VSZIP_EEDI3_NO_OVERLAP", None)
        n = max(4, 48 // frames)
        for _ in range(2):
            dev.check(dev.lib.vszip_eedi3(dev.ctx, table, None, None, len(srcs), 1, 0, prm))
        dev.sync()
        t = time.perf_counter()
        for _ in range(n):
            dev.check(dev.lib.vszip_eedi3(dev.ctx, table, None, None, len(srcs), 1, 0, prm))
        dev.sync()
        row.append(frames * n / (time.perf_counter() - t))
    print(f"frames per call {frames:2d}: overlap {row[0]:8.1f} fps   no overlap {row[1]:8.1f} fps", flush=True)
