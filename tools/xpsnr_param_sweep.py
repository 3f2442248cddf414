#!/usr/bin/env python3
"""XPSNR over frame sizes, sample depths and the temporal term: frames/s of the batch call (32 frames) and the fraction of the HBM peak its compulsory reads are."""
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
for w, h in ((1280, 720), (1920, 1080), (2560, 1440), (3840, 2160)):
    for dt, depth in ((np.uint8, 8), (np.uint16, 10), (np.uint16, 16)):
        frames = 32
        shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
        base = [fx.tiled_natural(s, dt, p) >> (16 - depth if dt == np.uint16 else 0) for p, s in enumerate(shapes)]
        a = [d.upload(np.ascontiguousarray(np.roll(b, 3 * f, axis=1))) for f in range(frames) for b in base]
        b2 = [d.upload(np.ascontiguousarray(np.roll(np.clip(b.astype(np.int32) + 2, 0, (1 << depth) - 1).astype(dt), 3 * f, axis=1))) for f in range(frames) for b in base]
        orgs = [a[3 * f:3 * f + 3] for f in range(frames)]
        recs = [b2[3 * f:3 * f + 3] for f in range(frames)]
        row = []
        for temporal in (False, True):
            p1 = [None] + [orgs[f - 1][0] for f in range(1, frames)] if temporal else None
            p2 = [None, None] + [orgs[f - 2][0] for f in range(2, frames)] if temporal else None
            for order2 in ((False, True) if temporal else (False,)):
                call = d.xpsnr_batch_call(orgs, recs, p1, p2 if order2 else None, depth=depth, frame_rate=60 if order2 else 24, temporal=temporal)
                _, region_ms, *_ = timed.run(call, 6, 2)
                nbytes = 2 * sum(x.size * x.itemsize for x in base) * frames + (base[0].size * base[0].itemsize * frames * (2 if order2 else 1) if temporal else 0)
                row.append(f"{'t2' if order2 else ('t1' if temporal else 'spatial')}: {frames * 6 / (region_ms * 1e-3) / 1e3:7.1f}k fps ({nbytes * 6 / (region_ms * 1e-3) / 8e12:.2f})")
        print(f"{w}x{h} depth {depth:2d}: " + " | ".join(row), flush=True)
        del a, b2
