#!/usr/bin/env python3
"""BoxBlur r=13 on float YUV420 frames of one size (argv: w h frames), a few calls — for rocprofv3 --kernel-trace --stats."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import fixtures as fx  # noqa: E402
import vszip_amd  # noqa: E402

w, h, frames = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = vszip_amd.Device(0)
shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
base = [fx.tiled_natural(s, np.float32, p) for p, s in enumerate(shapes)]
srcs = [d.upload(np.ascontiguousarray(np.roll(b, 3 * f, axis=1))) for f in range(frames) for b in base]
dsts = [d.empty(b.shape[0], b.shape[1], b.dtype) for f in range(frames) for b in base]
for _ in range(6):
    d.boxblur(srcs, dsts, 13, 1, 13, 1)
d.sync()
