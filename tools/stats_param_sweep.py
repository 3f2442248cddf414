#!/usr/bin/env python3
"""PlaneAverage / PlaneMinMax on 64 x 4K (and 1080p) YUV 4:2:0 frames over their parameters and sample types: frames/s and the fraction of the 8 TB/s a single
read of the planes would be (whole call: kernels + the results' way back + the synchronise)."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import fixtures as fx  # noqa: E402
import vszip_amd  # noqa: E402

d = vszip_amd.Device(0)


def clock(step, n=12):
    for _ in range(3):
        step()
    d.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    d.sync()
    return (time.perf_counter() - t0) / n


for dt in (np.uint8, np.uint16, np.float32):
    for w, h, frames in ((3840, 2160, 64), (1920, 1080, 64)):
        base = [fx.tiled_natural(s, dt, p) for p, s in enumerate([(h, w), (h // 2, w // 2), (h // 2, w // 2)])]
        srcs = [d.upload(np.ascontiguousarray(np.roll(b, 3 * f, axis=1))) for f in range(frames) for b in base]
        refs = [d.upload(np.ascontiguousarray(np.roll(b, 3 * f + 1, axis=1))) for f in range(frames) for b in base]
        nbytes = sum(a.size * a.itemsize for a in base) * frames
        cases = [("average", lambda: d.plane_average(srcs)), ("average exclude=[0]", lambda: d.plane_average(srcs, exclude=[0] if dt != np.float32 else [])),
                 ("average exclude=[16,235]", lambda: d.plane_average(srcs, exclude=[16, 235] if dt != np.float32 else [])), ("average + ref", lambda: d.plane_average(srcs, refs=refs)),
                 ("minmax", lambda: d.plane_minmax(srcs)), ("minmax thr 0.01", lambda: d.plane_minmax(srcs, 0.01, 0.01)), ("minmax thr 0.3", lambda: d.plane_minmax(srcs, 0.3, 0.3)),
                 ("minmax + ref", lambda: d.plane_minmax(srcs, refs=refs)), ("minmax thr + ref", lambda: d.plane_minmax(srcs, 0.01, 0.01, refs=refs))]
        row = []
        for name, step in cases:
            try:
                t = clock(step)
                reads = 2 if "ref" in name else 1
                row.append(f"{name}: {frames / t / 1e3:6.1f}k ({reads * nbytes / t / 8e12:.2f})")
            except Exception as e:  # noqa: BLE001
                row.append(f"{name}: ERR {str(e)[:30]}")
        print(f"{dt.__name__:8s} {w}x{h}: " + " | ".join(row), flush=True)
        del srcs, refs
