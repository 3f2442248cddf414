#!/usr/bin/env python3
"""Round 4: SSIMULACRA2 4K RGBS, 16 pairs per call: library variants / options in one process."""
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
devs = {}
for n in names:
    lib, _, opt = n.partition("+")
    capi.LIB_PATH = ROOT / ("vapoursynth-zip_amd/libvszip_hip.so" if lib == "base" else f"tools/ab/{lib}.so")
    capi._lib = None
    d = vszip_amd.Device(0)
    if opt:
        d.set_option(opt, 1)
    devs[n] = d
d0 = devs[names[0]]
steps = {}
keeps = []
for n in names:
    st, keep = bench.setup_ssimulacra2(devs[n], bench.W4K, bench.H4K, 16)
    steps[n] = st
    keeps.append(keep)
scores = {}
for rnd in range(3):
    for n in names:
        timed = bench.Timed(devs[n], lambda: None)
        dt, kms, _, _ = timed.run(steps[n], 5, 1)
        print(f"round {rnd} {n:28s} {16 * 5 / dt:8.1f} pairs/s  {kms / 5:7.3f} ms per 16 pairs", flush=True)
