#!/usr/bin/env python3
"""GPU box: BoxBlur CT float, 8 (or with `one`: 1) x 4K YUV420PS per call, fps per radius (ring kernel vs VSZIP_BOXBLUR_NO_FLOAT_RING=1)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch  # noqa: F401
import bench
import vszip_amd

dev = vszip_amd.Device(0)
timed = bench.Timed(dev, dev.sync)
dt_ = np.float16 if "f16" in sys.argv else np.float32
NF = 1 if "one" in sys.argv else 8
base = [(p.astype(np.float32) / 65535.0).astype(dt_) for p in bench.make_frame(3, bench.W4K, bench.H4K)]
srcs, dsts = [], []
for f in range(NF):
    for p in base:
        srcs.append(dev.upload(np.roll(p, f + 1, axis=1)))
        dsts.append(dev.empty(p.shape[0], p.shape[1], p.dtype))
table = dev.plane_table(srcs, dsts)
fb = 2 * sum(p.nbytes for p in base) * NF
for r in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 3, 5, 8, 11, 13, 15, 17, 18, 22]:
    dt, kms, _, _ = timed.run(lambda: dev.boxblur_table(dt_, table, r, 1, r, 1), 5, 1)
    print("r=%2d  %8.0f fps  %.3f of HBM" % (r, NF * 5 / dt, fb * 5 / (kms * 1e-3) / 1e9 / 8000), flush=True)
