#!/usr/bin/env python3
"""GPU box: BoxBlur at the radii scripts actually use (1 ... 5, single pass and two passes), 1080p YUV420P8 / P16, 64 frames per call."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, bench, vszip_amd, fixtures as fx
dev = vszip_amd.Device(0)
F = int(os.environ.get("FRAMES", "64"))
ARGS = os.environ.get("ARGS")  # e.g. "2,3,2,3;13,2,13,2": these instead of the default list
for dt in (np.uint8, np.uint16):
    base = [fx.tiled_natural(s, dt, p) for p, s in enumerate(bench.yuv420_shapes(1920, 1080))]
    srcs = [dev.upload(np.roll(p, f * 3, axis=1)) for f in range(F) for p in base]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for f in range(F) for p in base]
    table = dev.plane_table(srcs, dsts)
    fb = 2 * sum(p.nbytes for p in base) * F
    for args in ([tuple(int(v) for v in a.split(",")) for a in ARGS.split(";")] if ARGS else ((1, 1, 1, 1), (2, 1, 2, 1), (3, 1, 3, 1), (5, 1, 5, 1), (13, 1, 13, 1), (1, 2, 1, 2), (2, 3, 2, 3))):
        step = lambda: dev.boxblur_table(dt, table, *args)
        for _ in range(3): step()
        dev.sync(); t = time.perf_counter()
        for _ in range(10): step()
        dev.sync(); d = (time.perf_counter() - t) / 10
        print(f"{np.dtype(dt).name:7s} r={args[0]}x{args[1]} / {args[2]}x{args[3]}: {F / d:9.0f} fps  {fb / d / 8e12:5.3f} of the HBM peak", flush=True)
