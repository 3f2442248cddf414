#!/usr/bin/env python3
"""GPU box: 8-bit BoxBlur CT at r = 15 ... 20, 1080p and 4K (64 frames a call), the default lane width against 8 pixels a lane (VSZIP_CT_U8_PX8).
Found with it (round 5, left as found): at 1080p the ring periods of 42 / 44 rows (r = 18, 19 by default) run 145-150 us where 36 / 46 / 48 rows run 121-128."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch, bench, vszip_amd, fixtures as fx
dev = vszip_amd.Device(0)
for (w, h, F) in ((1920, 1080, 64), (3840, 2160, 64)):
    base = [fx.tiled_natural(s, np.uint8, p) for p, s in enumerate(bench.yuv420_shapes(w, h))]
    srcs = [dev.upload(np.roll(p, f * 3, axis=1)) for f in range(F) for p in base]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for f in range(F) for p in base]
    table = dev.plane_table(srcs, dsts)
    fb = 2 * sum(p.nbytes for p in base) * F
    for r in (15, 16, 17, 18, 19, 20):
        res = {}
        for rnd in range(2):
            for px8 in (0, 1):
                dev.set_option("VSZIP_CT_U8_PX8", px8)
                step = lambda: dev.boxblur_table(np.uint8, table, r, 1, r, 1)
                for _ in range(3): step()
                dev.sync(); t = time.perf_counter()
                for _ in range(20): step()
                dev.sync(); d = (time.perf_counter() - t) / 20
                res.setdefault(px8, []).append(round(d * 1e6, 1))
        print(f"{w}x{h} u8 r={r}: default {res[0]} us, 8 pixels a lane {res[1]} us", flush=True)
    del srcs, dsts
