#!/usr/bin/env python3
"""GPU box: thresholded PlaneMinMax on 16-bit planes, the default two sweeps against the single-read path (a development variant:
build a -DVSZIP_DEV_VARIANTS library into tools/ab/dev.so first), 64 x 4K YUV420P16 per call, on noise / a picture x 257 / a picture with noisy low bits."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import vszip_amd  # noqa: E402
from vszip_amd import capi  # noqa: E402

capi.LIB_PATH = ROOT / "tools/ab/dev.so"
capi._lib = None
dev = vszip_amd.Device(0)
frames = 64


def run(planes, tag):
    res = {}
    for single in (1, 0, 1, 0):
        dev.set_option("VSZIP_MINMAX_SINGLE_READ", single)
        for _ in range(3):
            r = dev.plane_minmax(planes, 0.1, 0.1)
        t = time.perf_counter()
        for _ in range(20):
            r = dev.plane_minmax(planes, 0.1, 0.1)
        dt = (time.perf_counter() - t) / 20
        res.setdefault(single, []).append(dt * 1e6)
        key = (r[0][:3], r[1][:3])
        res.setdefault("ans", set()).add(str(key))
    alg = sum(p.h * p.w * 2 for p in planes)
    print(f"{tag:18s} single read {res[1][0]:7.1f} / {res[1][1]:7.1f} us ({alg / (min(res[1]) * 1e-6) / 8e12:.3f} of peak)   two sweeps {res[0][0]:7.1f} / {res[0][1]:7.1f} us ({alg / (min(res[0]) * 1e-6) / 8e12:.3f})   same answers: {len(res['ans']) == 1}", flush=True)


base = bench.make_frame(7, 3840, 2160)
run([dev.upload(np.roll(p, f * 3, axis=1)) for f in range(frames) for p in base], "noise")
nat = bench.natural_frame(3840, 2160)
run([dev.upload(np.roll(p, f * 3, axis=1)) for f in range(frames) for p in nat], "natural x 257")
rng = np.random.default_rng(1)
nat2 = [(p.astype(np.int64) - 128 + rng.integers(0, 256, p.shape)).clip(0, 65535).astype(np.uint16) for p in nat]
run([dev.upload(np.roll(p, f * 3, axis=1)) for f in range(frames) for p in nat2], "natural + low bits")
