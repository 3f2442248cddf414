#!/usr/bin/env python3
"""GPU box: the float pass-chain kernels alone — 4K YUV420PS, FRAMES per call (default 8), r = 5 x 3 passes on one axis."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import fixtures as fx
import vszip_amd
dev = vszip_amd.Device(0)
frames = int(os.environ.get("FRAMES", "8"))
planes = [fx.tiled_natural(s, np.float32, p) for p, s in enumerate([(2160, 3840), (1080, 1920), (1080, 1920)])]
srcs = [dev.upload(np.roll(p, f, axis=1)) for f in range(frames) for p in planes]
dsts = [dev.empty(p.shape[0], p.shape[1], np.float32) for f in range(frames) for p in planes]
out = []
for args in ((5, 3, 0, 0), (0, 0, 5, 3), (1, 2, 1, 2)):
    dev.boxblur(srcs, dsts, *args); dev.sync()
    t0 = time.perf_counter()
    for _ in range(10): dev.boxblur(srcs, dsts, *args)
    dev.sync()
    out.append("%s %.0f us" % (args, (time.perf_counter() - t0) / 10 * 1e6))
print("  ".join(out))
