"""Development: per-stage times of BASELINE config 5 (Bilateral -> BoxBlur -> SSIMULACRA2 on 7680x4320
RGBS, planes resident in HBM) on one GPU."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401

import fixtures as fx
import vszip_amd

dev = vszip_amd.Device(0)
h, w = 4320, 7680
rgb = [np.clip(fx.tiled_natural((h, w), np.float32, p), 0, 1).astype(np.float32) for p in range(3)]
cfg = dev.bilateral_cfg([2], [2], yuv=False, ssw=0, ssh=0, hist_len=65536)
srcs = [dev.upload(p, 1) for p in rgb]
mid = [dev.empty(h, w, np.float32, 1) for _ in range(3)]
out = [dev.empty(h, w, np.float32, 1) for _ in range(3)]


def t(fn, n=5):
    fn()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dev.sync()
    return (time.perf_counter() - t0) / n * 1e3


tb = t(lambda: dev.bilateral(srcs, mid, cfg, [0, 1, 2]))
tx2 = t(lambda: dev.boxblur(mid, out, 2, 1, 2, 1))
tx13 = t(lambda: dev.boxblur(mid, out, 13, 1, 13, 1))
ts = t(lambda: dev.ssimulacra2(srcs, out))
print(f"8K RGBS per frame: Bilateral {tb:.2f} ms, BoxBlur r=2 {tx2:.2f} ms (r=13 {tx13:.2f} ms), SSIMULACRA2 {ts:.2f} ms -> {1e3 / (tb + tx2 + ts):.0f} frames/s per GPU")
dev.close()
