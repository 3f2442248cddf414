#!/usr/bin/env python3
"""Run single bench legs (eedi3 | xpsnr | planestats) for rocprofv3 --kernel-trace --stats."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401  (its HIP runtime first)

import bench
import vszip_amd

dev = vszip_amd.Device(0)
timed = bench.Timed(dev, dev.sync)
for leg in sys.argv[1:]:
    if leg == "eedi3":
        print(json.dumps(bench.eedi3_leg(dev, timed, True)))
    elif leg == "xpsnr":
        print(json.dumps(bench.xpsnr_leg(dev, timed, True)))
    elif leg == "boxblur_other":
        print(json.dumps(bench.boxblur_other_paths_leg(dev, timed)))
    elif leg == "limiter":
        print(json.dumps(bench.limiter_leg(dev, timed)))
        print(json.dumps(bench.limit_filter_leg(dev, timed)))
    elif leg == "pbfic":
        import runpy

        runpy.run_path(str(ROOT / "tools" / "pbfic_timing.py"))
    elif leg == "ssim_yuv":
        st, keep = bench.setup_ssimulacra2_yuv420p8(dev, bench.W4K, bench.H4K, 16)
        dt, _, _, _ = timed.run(st, 5, 1)
        print(json.dumps({"ssimulacra2_4k_yuv420p8_pairs_s": 16 * 5 / dt}))
        del keep
    elif leg == "planestats":
        print(json.dumps(bench.planestats_leg(dev, timed)))
