#!/usr/bin/env python3
"""Round 4: EEDI3 1080p -> 2160 (YUV420P8, 16 frames per call): library variants in one process, per-kernel time from the probe."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import vszip_amd  # noqa: E402
from vszip_amd import capi  # noqa: E402

names = sys.argv[1:] or ["base"]
for rnd in range(2):
    for n in names:
        capi.LIB_PATH = ROOT / ("vapoursynth-zip_amd/libvszip_hip.so" if n == "base" else f"tools/ab/{n}.so")
        capi._lib = None
        d = vszip_amd.Device(0)
        timed = bench.Timed(d, lambda: None)
        leg = bench.eedi3_leg(d, timed, True)
        print(f"round {rnd} {n:14s} {leg['value']:8.1f} fps   line kernel {leg['roofline'].get('avg_launch_us', 0):8.1f} us/launch", flush=True)
        d.close()
