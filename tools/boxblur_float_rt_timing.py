"""Development: the float runtime BoxBlur path (running f32 sums, sequential per line) on 4K planes."""
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
frames = int(os.environ.get("FRAMES", "8"))
H = int(os.environ.get("HEIGHT", "2160"))  # 2160: 4K, 1080: 1080p
planes = [fx.tiled_natural(s, np.float32, p) for p, s in enumerate([(H, H * 16 // 9), (H // 2, H * 8 // 9), (H // 2, H * 8 // 9)])]
srcs = [dev.upload(np.roll(p, f, axis=1)) for f in range(frames) for p in planes]
dsts = [dev.empty(p.shape[0], p.shape[1], np.float32) for f in range(frames) for p in planes]
for args in ((30, 1, 30, 1), (5, 3, 5, 3), (5, 3, 0, 0), (0, 0, 5, 3), (2, 2, 2, 2), (13, 5, 13, 5), (13, 1, 13, 1)):
    dev.boxblur(srcs, dsts, *args)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        dev.boxblur(srcs, dsts, *args)
    dev.sync()
    print(f"float BoxBlur {args}: {10 * frames / (time.perf_counter() - t0):8.1f} frames/s ({H}p YUV420PS, {frames} frames per call)", flush=True)
dev.close()
