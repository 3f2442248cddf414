#!/usr/bin/env python3
"""Bilateral 1080p YUV420P16 / P8, 64 frames per call, over sigmaS (its radius / step pick the kernel) and two sigmaR: frames/s."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import vszip_amd  # noqa: E402

d = vszip_amd.Device(0)
timed = bench.Timed(d, d.sync)
timed.prewarm_s = 0.2
for b8 in (False, True):
    for sr in (2, 0.02):
        row = []
        for ss in (0.5, 1, 1.5, 2, 2.5, 3, 4, 5, 7):
            try:
                step, keep = bench.setup_bilateral(d, bench.W1080, bench.H1080, 64, ss, sr, b8)
                _, region_ms, *_ = timed.run(step, 4, 1)
                row.append(f"sS={ss}: {64 * 4 / (region_ms * 1e-3) / 1e3:6.1f}k")
                del step, keep
            except Exception as e:  # noqa: BLE001
                row.append(f"sS={ss}: ERR {str(e)[:40]}")
        print(f"{'P8 ' if b8 else 'P16'} sigmaR={sr}: " + " | ".join(row), flush=True)
