#!/usr/bin/env python3
"""Round 4: Bilateral 1080p legs (sigmaS=2 sigmaR=2 P16; the filter's defaults sigmaS=3 sigmaR=0.02 P16; sigmaS=2 sigmaR=2 P8; 4K P16) for library
variants (tools/variant.sh) in one process, frames/s from the stream clock. usage: bil_ab.py base name ..."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import vszip_amd  # noqa: E402
from vszip_amd import capi  # noqa: E402

names = sys.argv[1:] or ["base"]
CASES = [("s2_r2_p16_1080p", bench.W1080, bench.H1080, 64, 2, 2, False), ("defaults_p16_1080p", bench.W1080, bench.H1080, 64, 3, 0.02, False),
         ("s2_r2_p8_1080p", bench.W1080, bench.H1080, 64, 2, 2, True), ("s2_r2_p16_4k", bench.W4K, bench.H4K, 16, 2, 2, False)]
for rnd in range(2):
    for n in names:
        capi.LIB_PATH = ROOT / ("vapoursynth-zip_amd/libvszip_hip.so" if n == "base" else f"tools/ab/{n}.so")
        capi._lib = None
        d = vszip_amd.Device(0)
        timed = bench.Timed(d, d.sync)
        out = []
        for name, w, h, nf, ss, sr, b8 in CASES:
            step, keep = bench.setup_bilateral(d, w, h, nf, ss, sr, b8)
            _, region_ms, *_ = timed.run(step, 10, 2)
            out.append(f"{name} {nf * 10 / (region_ms * 1e-3):9.0f}")
            del keep, step
        print(f"round {rnd} {n:12s} " + "   ".join(out), flush=True)
        d.close()
