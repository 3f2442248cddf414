#!/usr/bin/env python3
"""GPU box: SSIMULACRA2 from 4K YUV 4:2:0 clips at 8, 10 and 16 bits, 16 pairs a call: the pre-stage pass + f32 pyramid pass (default) against
the fused tile kernel (VSZIP_SSIM_NO_YUV420_LDS=1), interleaved. One line per case."""
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
ref8, dis8 = bench.yuv420p8_pair(bench.W4K, bench.H4K)
for bits in (8, 10, 16):
    if bits == 8:
        ref, dis, dt = ref8, dis8, np.uint8
    else:
        ref, dis, dt = [p.astype(np.uint16) << (bits - 8) for p in ref8], [p.astype(np.uint16) << (bits - 8) for p in dis8], np.uint16
    fmt = dev.ssim_source("YUV", dt, bits, ssw=1, ssh=1, matrix=1, chroma_loc=0)
    r, d = [], []
    for p in range(16):
        r += [dev.upload(np.roll(x, p * 8, axis=1)) for x in ref]
        d += [dev.upload(np.roll(x, p * 8, axis=1)) for x in dis]
    out = {}
    for rnd in range(2):
        for name, off in (("split", 0), ("fused", 1)):
            dev.set_option("VSZIP_SSIM_NO_YUV420_LDS", off)
            s0 = dev.ssimulacra2_src(fmt, r, d)
            dev.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                s = dev.ssimulacra2_src(fmt, r, d)
            dev.sync()
            out.setdefault(name, []).append(round(80 / (time.perf_counter() - t0), 1))
            out.setdefault(name + "_score", s[0])
    print(bits, out, "equal scores" if out["split_score"] == out["fused_score"] else "SCORES DIFFER", flush=True)
    del r, d
