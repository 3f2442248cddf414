"""GPU box: what a synchronous vszip_plane_minmax call costs beside its kernels - time per call for 1 / 4 / 16 / 64 4K YUV420P16 frames, thresholds 0.1 (predicted) and none,
on argument blocks built once; and the same for tiny planes (64 x 64: the fixed cost alone)."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
import vszip_amd

dev = vszip_amd.Device(0)
out = []
base = bench.make_frame(4, 3840, 2160)
tiny = [np.ascontiguousarray(p[:64, :64]) for p in base]
for label, frame in (("4K", base), ("64x64", tiny)):
    for frames in (1, 4, 16, 64):
        srcs = [dev.upload(np.roll(p, f, axis=1)) for f in range(frames) for p in frame]
        for thr in (0.1, 0.0):
            run = dev.prepared_plane_minmax(srcs, thr, thr)
            run(); run()
            n = 200 if label != "4K" or frames < 16 else 30
            t0 = time.perf_counter()
            for _ in range(n):
                run()
            us = (time.perf_counter() - t0) / n * 1e6
            out.append(f"{label} x{frames} thr={thr}: {us:.1f} us")
        del srcs
print(" | ".join(out))
