#!/usr/bin/env python3
"""GPU box: kernel-only rates (HIP-event probe) of the pure streaming filters on 16 4K YUV420P16 frames —
Limiter, LimitFilter, PlaneAverage, PlaneMinMax — and SSIMULACRA2 from f32 / u16 / u8 sources."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
import fixtures as fx
import vszip_amd
import bench as B

dev = vszip_amd.Device(0)
frames = 16
base = B.make_frame(1, B.W4K, B.H4K)
srcs, flts, dsts = [], [], []
for f in range(frames):
    for p in base:
        srcs.append(dev.upload(np.roll(p, f + 1, axis=1))); flts.append(dev.upload(np.roll(p, f + 3, axis=1))); dsts.append(dev.empty(p.shape[0], p.shape[1], p.dtype))
n = len(srcs)
fb = sum(p.nbytes for p in base) * frames

def probe(fn, reps=20):
    fn(); dev.sync(); dev.probe_enable(True)
    for _ in range(reps): fn()
    ms, launches, each = dev.probe_read_each(); dev.probe_enable(False)
    a = np.sort(np.array(each)) * 1e3
    return a[len(a) // 2], launches // reps

for name, fn, streams in (("Limiter", lambda: dev.limiter(srcs, dsts, [4096] * n, [60160] * n), 2),
                          ("LimitFilter", lambda: dev.limit_filter(flts, srcs, dsts, [2056.0] * n, [2056.0] * n, [3.0] * n), 3),
                          ("PlaneAverage", lambda: dev.plane_average(srcs, [-1]), 1), ("PlaneMinMax", lambda: dev.plane_minmax(srcs), 1)):
    try:
        us, per = probe(fn)
    except Exception as e:
        print(f"{name:14s} failed: {e}")
        continue
    print(f"{name:14s} {us * per:8.1f} us/call ({per} launch)  {streams * fb / (us * per * 1e-6) / 1e9:7.0f} GB/s = {streams * fb / (us * per * 1e-6) / 8e12:.3f} of peak")

# SSIMULACRA2: 16 pairs per call, from linear f32 and from 16- / 8-bit RGB sources
import time
h, w = B.H4K, B.W4K
nat = [fx.tiled_natural((h, w), np.uint8, p) for p in range(3)]
rng = np.random.default_rng(1)
for label, dt, bits in (("RGBS linear", np.float32, 32), ("RGB48", np.uint16, 16), ("RGB24", np.uint8, 8)):
    if dt == np.float32:
        ref, dis = B.rgbs_pair(w, h)
        fmt = dev.ssim_source("RGB", np.float32, 32, linearize=False)
    else:
        sc = 257 if bits == 16 else 1
        ref = [(p.astype(dt) * dt(sc)) for p in nat]
        dis = [np.clip(p.astype(np.int32) + rng.integers(-3 * sc, 3 * sc + 1, p.shape), 0, 255 * sc).astype(dt) for p in ref]
        fmt = dev.ssim_source("RGB", dt, bits)
    r, d = [], []
    for k in range(16):
        r += [dev.upload(np.roll(x, 7 * k, axis=1)) for x in ref]; d += [dev.upload(np.roll(x, 7 * k, axis=1)) for x in dis]
    dev.ssimulacra2_src(fmt, r, d)
    t0 = time.perf_counter()
    for _ in range(5): dev.ssimulacra2_src(fmt, r, d)
    dt_ = (time.perf_counter() - t0) / 5
    print(f"SSIMULACRA2 4K from {label:12s}: {16 / dt_:7.0f} pairs/s ({dt_ * 1e3 / 16:.3f} ms/pair), upload bytes/pair {2 * sum(x.nbytes for x in ref) / 1e6:.0f} MB")
    del r, d
