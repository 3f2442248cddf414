#!/usr/bin/env python3
"""GPU box: output Mpixel/s of the pixel filters across frame sizes and sample types — looking for CLIFFS: a size or a type that falls off a tuned path
(alignment rules, width limits) shows as a row far below its neighbours. Frames per call scale so that a call moves about 200 Mpixel.
usage: cliff_sweep.py [filter ...]   (boxblur13 boxblur2 boxblur30 boxblur5x3 bilateral bilateral_default limiter average eedi3 xpsnr)"""
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

SIZES = [(1280, 720), (1366, 768), (1916, 1076), (1918, 1078), (1920, 1080), (1920, 1088), (2560, 1440), (3838, 2158), (3840, 2160), (4096, 2160), (854, 480), (1001, 563)]
TYPES = [("u8", np.uint8), ("u16", np.uint16), ("f32", np.float32)]
only = sys.argv[1:]
d = vszip_amd.Device(0)
timed = bench.Timed(d, d.sync)
timed.prewarm_s = 0.2


def planes_420(w, h, dt, frames, gray=False):
    shapes = [(h, w)] if gray else [(h, w), ((h + 1) // 2, (w + 1) // 2), ((h + 1) // 2, (w + 1) // 2)]
    base = [fx.tiled_natural(s, dt, p) for p, s in enumerate(shapes)]
    srcs = [d.upload(np.ascontiguousarray(np.roll(b, 3 * f, axis=1))) for f in range(frames) for b in base]
    dsts = [d.empty(b.shape[0], b.shape[1], b.dtype) for f in range(frames) for b in base]
    return srcs, dsts, [i for f in range(frames) for i in range(len(base))]


def run(name, step, mpx):
    try:
        _, region_ms, *_ = timed.run(step, 4, 1)
        return mpx * 4 / (region_ms * 1e-3) / 1e3  # Gpx/s
    except Exception as e:  # noqa: BLE001
        return str(e)[:60]


def sweep(name, make):
    if only and name not in only:
        return
    print(f"== {name}: Gpixel/s of luma+chroma samples (YUV 4:2:0 unless noted)", flush=True)
    for tn, dt in TYPES:
        row = []
        for w, h in SIZES:
            frames = max(2, int(200e6 / (w * h * 1.5)))
            got = make(w, h, dt, frames)
            if got is None:
                row.append(f"{w}x{h}: --")
                continue
            step, mpx, keep = got
            r = run(name, step, mpx)
            row.append(f"{w}x{h}: {r:6.1f}" if isinstance(r, float) else f"{w}x{h}: ERR {r}")
            del keep, step
        print(f"  {tn:4s} " + " | ".join(row), flush=True)


def mk_boxblur(args):
    def make(w, h, dt, frames):
        srcs, dsts, _ = planes_420(w, h, dt, frames)
        return (lambda: d.boxblur(srcs, dsts, *args)), frames * w * h * 1.5 / 1e6, (srcs, dsts)
    return make


def mk_bilateral(ss, sr):
    def make(w, h, dt, frames):
        srcs, dsts, idx = planes_420(w, h, dt, frames)
        cfg = d.bilateral_cfg([ss], [sr], yuv=True, ssw=1, ssh=1, hist_len=256 if dt == np.uint8 else 65536)
        peak = 255.0 if dt == np.uint8 else None
        return (lambda: d.bilateral(srcs, dsts, cfg, idx, peak=peak)), frames * w * h * 1.5 / 1e6, (srcs, dsts, cfg)
    return make


def mk_limiter(w, h, dt, frames):
    if dt == np.float32:
        lo, hi = [0.1] * 3, [0.9] * 3
    else:
        peak = 255 if dt == np.uint8 else 65535
        lo, hi = [peak // 16] * 3, [peak - peak // 16] * 3
    srcs, dsts, idx = planes_420(w, h, dt, frames)
    los = [lo[i] for i in idx]
    his = [hi[i] for i in idx]
    return (lambda: d.limiter(srcs, dsts, los, his)), frames * w * h * 1.5 / 1e6, (srcs, dsts)


def mk_average(w, h, dt, frames):
    srcs, _, _ = planes_420(w, h, dt, frames)
    return (lambda: d.plane_average(srcs)), frames * w * h * 1.5 / 1e6, (srcs,)


def mk_eedi3(w, h, dt, frames):
    if dt != np.float32:
        return None
    frames = max(2, frames // 8)
    srcs, _, _ = planes_420(w, h, dt, frames)
    return (lambda: d.eedi3(srcs, 1, dh=False)), frames * w * h * 1.5 / 1e6, (srcs,)


def mk_xpsnr(w, h, dt, frames):
    if dt == np.float32:
        return None
    frames = min(frames, 32)
    a, _, _ = planes_420(w, h, dt, frames)
    b, _, _ = planes_420(w, h, dt, frames)
    orgs = [a[3 * f:3 * f + 3] for f in range(frames)]
    recs = [b[3 * f:3 * f + 3] for f in range(frames)]
    call = d.xpsnr_batch_call(orgs, recs, depth=8 if dt == np.uint8 else 16, temporal=False)
    return call, frames * w * h * 1.5 / 1e6, (a, b)


sweep("boxblur13", mk_boxblur((13, 1, 13, 1)))
sweep("boxblur2", mk_boxblur((2, 1, 2, 1)))
sweep("boxblur30", mk_boxblur((30, 1, 30, 1)))
sweep("boxblur5x3", mk_boxblur((5, 3, 5, 3)))
sweep("bilateral", mk_bilateral(2, 2))
sweep("bilateral_default", mk_bilateral(3, 0.02))
sweep("limiter", mk_limiter)
sweep("average", mk_average)
sweep("eedi3", mk_eedi3)
sweep("xpsnr", mk_xpsnr)
