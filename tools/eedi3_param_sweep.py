#!/usr/bin/env python3
"""EEDI3 field=1 dh=1 on 16 x 1080p YUV420PS per call over its parameters (each picks a kernel instantiation): frames/s (with the output planes' allocation inside)."""
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

d = vszip_amd.Device(0)
timed = bench.Timed(d, d.sync)
timed.prewarm_s = 0.2
base = [np.ascontiguousarray(fx.tiled_natural(s, np.float32, p)) for p, s in enumerate([(1080, 1920), (540, 960), (540, 960)])]
srcs = [d.upload(np.roll(pl, 7 * f, axis=1)) for f in range(16) for pl in base]
for kw in (dict(), dict(mdis=10), dict(mdis=15), dict(mdis=19), dict(mdis=21), dict(mdis=30), dict(mdis=40), dict(nrad=0), dict(nrad=1), dict(nrad=3), dict(hp=True), dict(hp=True, mdis=10),
           dict(vcheck=0), dict(vcheck=3), dict(gamma=0.0)):
    step = lambda: d.eedi3(srcs, 1, dh=True, **kw)
    _, region_ms, *_ = timed.run(step, 3, 1)
    print(f"{str(kw):28s} {16 * 3 / (region_ms * 1e-3):8.1f} fps", flush=True)
