#!/usr/bin/env python3
"""GPU box: the vertical pass chain for library variants in one process: us per call for v-only r = 13 / r = 5, 2 / 3 / 5 passes, 1080p x 32 and 4K x 8 (u16)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa
import bench, fixtures as fx, vszip_amd
from vszip_amd import capi
names = sys.argv[1:] or ["base"]
for rnd in range(2):
    for n in names:
        capi.LIB_PATH = ROOT / ("vapoursynth-zip_amd/libvszip_hip.so" if n == "base" else f"tools/ab/{n}.so")
        capi._lib = None
        d = vszip_amd.Device(0)
        timed = bench.Timed(d, d.sync); timed.prewarm_s = 0.1
        out = []
        for dt, w, h, frames in ((np.uint16, 1920, 1080, 32), (np.uint16, 3840, 2160, 8)):
            base = [fx.tiled_natural(s, dt, p) for p, s in enumerate([(h, w), (h // 2, w // 2), (h // 2, w // 2)])]
            srcs = [d.upload(np.ascontiguousarray(np.roll(b, 3 * f, axis=1))) for f in range(frames) for b in base]
            dsts = [d.empty(b.shape[0], b.shape[1], b.dtype) for f in range(frames) for b in base]
            table = d.plane_table(srcs, dsts)
            for args in ((0, 0, 13, 2), (0, 0, 13, 3), (0, 0, 13, 5), (0, 0, 5, 3), (0, 0, 5, 5)):
                _, ms, *_ = timed.run(lambda: d.boxblur_table(dt, table, *args), 8, 2)
                out.append(f"{w}:{args[2]}x{args[3]} {ms / 8 * 1e3:5.0f}")
            del srcs, dsts
        print(f"{n:8s} " + " | ".join(out), flush=True)
        d.close()
