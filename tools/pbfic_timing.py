"""Development: Bilateral algorithm 1 (PBFIC) on 3840x2160 Gray16, per-kernel times via the stream timer."""
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
h, w = 2160, 3840
src = dev.upload(fx.tiled_natural((h, w), np.uint16, 0))
dst = dev.empty(h, w, np.uint16)
for num in (4, 16):
    cfg = dev.bilateral_cfg([3], [0.1], algorithm=[1], pbficnum=[num], hist_len=65536)
    dev.bilateral([src], [dst], cfg, [0])
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        dev.bilateral([src], [dst], cfg, [0])
    dev.sync()
    print(f"PBFICnum={num}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per 4K Gray16 plane", flush=True)
    dev.bilateral_free(cfg)
dev.close()
