#!/usr/bin/env python3
"""Round 4: band count sweep of the banded integer pass chain (VSZIP_RT_ICHAIN_BANDS), vertical passes only."""
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
for dt, (w, h), frames, args in [(np.uint16, (1920, 1080), 32, (0, 0, 13, 5)), (np.uint16, (1920, 1080), 16, (0, 0, 13, 5)), (np.uint16, (1920, 1080), 4, (0, 0, 13, 5)), (np.uint16, (1920, 1080), 1, (0, 0, 13, 5)),
                                 (np.uint16, (3840, 2160), 8, (0, 0, 5, 3)), (np.uint16, (3840, 2160), 1, (0, 0, 5, 3)), (np.uint16, (1920, 1080), 64, (0, 0, 5, 2)), (np.uint16, (1920, 1080), 64, (0, 0, 2, 3)),
                                 (np.uint8, (1920, 1080), 64, (0, 0, 3, 3)), (np.uint8, (1920, 1080), 64, (0, 0, 1, 2)), (np.uint16, (1920, 1080), 16, (0, 0, 30, 2)), (np.uint16, (1920, 1080), 16, (0, 0, 8, 4))]:
    base = [fx.tiled_natural(s, dt, p) for p, s in enumerate(bench.yuv420_shapes(w, h))]
    srcs = [dev.upload(np.roll(p, f * 3, axis=1)) for f in range(frames) for p in base]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for f in range(frames) for p in base]
    table = dev.plane_table(srcs, dsts)
    step = lambda: dev.boxblur_table(dt, table, *args)
    out = []
    with dev.options(VSZIP_RT_NO_ICHAIN=1):
        _, kms, _, _ = timed.run(step, 10, 2)
    out.append(f"per-pass {kms / 10 * 1e3:6.1f}")
    with dev.options(VSZIP_RT_ICHAIN_ALL=1, VSZIP_RT_NO_BANDED=1):
        _, kms, _, _ = timed.run(step, 10, 2)
    out.append(f"whole {kms / 10 * 1e3:6.1f}")
    for nb in (2, 3, 4, 6, 8, 12, 16):
        with dev.options(VSZIP_RT_ICHAIN_ALL=1, VSZIP_RT_ICHAIN_BANDS=nb):
            _, kms, _, _ = timed.run(step, 10, 2)
        out.append(f"{nb}: {kms / 10 * 1e3:6.1f}")
    cg = sum((p.shape[1] + 63) // 64 for p in base) * frames
    print(f"{np.dtype(dt).name} {w}x{h} x{frames} {args} colgroups {cg}: " + " | ".join(out) + " us", flush=True)
    del srcs, dsts, table
dev.close()
