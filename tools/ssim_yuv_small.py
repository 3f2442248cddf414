#!/usr/bin/env python3
"""GPU box: SSIMULACRA2 from YUV420P8 in SMALL calls (1080p / 4K, 1 - 8 pairs a call): where the pre-stage pass's table staging (150 KB a
workgroup) stops paying against the fused tile kernel. pairs/s, split / fused, interleaved."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401

import bench
import vszip_amd

dev = vszip_amd.Device(0)
for w, h in ((1920, 1080), (3840, 2160), (1280, 720)):
    ref, dis = bench.yuv420p8_pair(w, h)
    fmt = dev.ssim_source("YUV", np.uint8, 8, ssw=1, ssh=1, matrix=1, chroma_loc=0)
    for pairs in (1, 2, 4, 8):
        r, d = [], []
        for p in range(pairs):
            r += [dev.upload(np.roll(x, p * 8, axis=1)) for x in ref]
            d += [dev.upload(np.roll(x, p * 8, axis=1)) for x in dis]
        out = {}
        n = max(4, 64 // pairs)
        for rnd in range(2):
            for name, off in (("split", 0), ("fused", 1)):
                dev.set_option("VSZIP_SSIM_NO_YUV420_LDS", off)
                dev.ssimulacra2_src(fmt, r, d)
                dev.sync()
                t0 = time.perf_counter()
                for _ in range(n):
                    dev.ssimulacra2_src(fmt, r, d)
                dev.sync()
                out.setdefault(name, []).append(round(pairs * n / (time.perf_counter() - t0), 1))
        print(f"{w}x{h} pairs/call {pairs}: {out}", flush=True)
        del r, d
