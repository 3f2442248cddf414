#!/usr/bin/env python3
"""Round 4: the banded integer pass chain against the whole-column chain and one launch per pass (vertical passes only and whole filters)."""
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

dev = vszip_amd.Device(0)
timed = bench.Timed(dev, lambda: None)
for dt, (w, h), frames, args in [(np.uint16, (1920, 1080), 32, (13, 5, 13, 5)), (np.uint16, (1920, 1080), 16, (13, 5, 13, 5)), (np.uint16, (1920, 1080), 16, (0, 0, 13, 5)), (np.uint16, (1920, 1080), 1, (13, 5, 13, 5)),
                                 (np.uint16, (3840, 2160), 8, (5, 3, 5, 3)), (np.uint16, (3840, 2160), 8, (0, 0, 5, 3)), (np.uint8, (1920, 1080), 64, (1, 2, 1, 2)), (np.uint8, (1920, 1080), 64, (3, 3, 3, 3)),
                                 (np.uint16, (1920, 1080), 64, (0, 0, 5, 2)), (np.uint8, (3840, 2160), 8, (2, 5, 2, 5))]:
    base = [fx.tiled_natural(s, dt, p) for p, s in enumerate(bench.yuv420_shapes(w, h))]
    srcs = [dev.upload(np.roll(p, f * 3, axis=1)) for f in range(frames) for p in base]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for f in range(frames) for p in base]
    table = dev.plane_table(srcs, dsts)
    step = lambda: dev.boxblur_table(dt, table, *args)
    nbytes = 2 * sum(p.nbytes for p in base) * frames
    out = []
    for name, opts in (("default", {}), ("banded everywhere", {"VSZIP_RT_ICHAIN_ALL": 1}), ("whole columns", {"VSZIP_RT_NO_BANDED": 1}), ("whole columns everywhere", {"VSZIP_RT_NO_BANDED": 1, "VSZIP_RT_ICHAIN_ALL": 1}),
                       ("per pass", {"VSZIP_RT_NO_ICHAIN": 1})):
        with dev.options(**opts):
            dts, kms, _, _ = timed.run(step, 10, 2)
        out.append(f"{name} {kms / 10 * 1e3:7.1f} us ({nbytes * 10 / (kms * 1e-3) / 8e12:.3f})")
    print(f"{np.dtype(dt).name} {w}x{h} x{frames} {args}: " + " | ".join(out), flush=True)
    del srcs, dsts, table
dev.close()
