#!/usr/bin/env python3
"""Run single bench legs for rocprofv3 --kernel-trace --stats (tools/prof_all.sh, tools/roofline_check.sh). One JSON line per leg:
the leg's record(s) and `calls` = how often the timed call ran in this process (prewarm + warm-up + timed), so that a profile's
total kernel time divides into a per-call time."""
import json
import os
import sys
from pathlib import Path

os.environ.setdefault("VSZIP_BENCH_NO_FIRST_US", "1")

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401  (its HIP runtime first)

import bench
import vszip_amd

dev = vszip_amd.Device(0)
timed = bench.Timed(dev, dev.sync)


def emit(name, rec):
    print(json.dumps({"leg": name, "calls": getattr(timed, "calls", 0), "record": rec}), flush=True)
    timed.calls = 0


for leg in sys.argv[1:]:
    if leg == "eedi3":
        emit(leg, bench.eedi3_leg(dev, timed, True))
    elif leg == "xpsnr":
        emit(leg, bench.xpsnr_leg(dev, timed, True))
    elif leg == "xpsnr_batch":
        emit(leg, bench.xpsnr_leg(dev, timed, True, per_frame=False))
    elif leg == "boxblur_other":
        emit(leg, bench.boxblur_other_paths_leg(dev, timed))
    elif leg.startswith("boxblur_rt_") or leg.startswith("boxblur_ct_"):
        emit(leg, bench.boxblur_other_paths_leg(dev, timed, only=leg)[leg])
    elif leg == "limiter":
        emit(leg, bench.limiter_leg(dev, timed)["limiter_4k"])
    elif leg == "limit_filter":
        emit(leg, bench.limit_filter_leg(dev, timed)["limit_filter_4k"])
    elif leg == "boxblur_1080p":
        emit(leg, bench.boxblur_1080p_leg(dev, timed, True))
    elif leg == "boxblur_1080p_5pass":
        emit(leg, bench.boxblur_1080p_5pass_leg(dev, timed, True))
    elif leg == "boxblur_1080p_r1x2_yuv420p8":
        emit(leg, bench.boxblur_gauss_leg(dev, timed))
    elif leg in ("plane_average_4k", "plane_minmax_4k", "plane_minmax_thr_4k"):
        emit(leg, bench.planestats_leg(dev, timed, only=leg)[leg])
    elif leg == "planestats":
        emit(leg, bench.planestats_leg(dev, timed))
    elif leg == "pbfic":
        import runpy

        runpy.run_path(str(ROOT / "tools" / "pbfic_timing.py"))
    elif leg == "ssim_yuv":
        st, keep = bench.setup_ssimulacra2_yuv420p8(dev, bench.W4K, bench.H4K, 16)
        dt, _, _, _ = timed.run(st, 5, 1)
        emit(leg, {"ssimulacra2_4k_yuv420p8_pairs_s": 16 * 5 / dt})
        del keep
    elif leg.startswith("bilateral_"):
        w, h, nf = (bench.W4K, bench.H4K, 16) if leg == "bilateral_4k" else (bench.W1080, bench.H1080, 64)
        st, keep = bench.setup_bilateral(dev, w, h, nf)
        dt2, kms, dms, nl = timed.run(st, 10, 2)
        fb2 = sum(2 * s_[0] * s_[1] for s_ in bench.yuv420_shapes(w, h))
        emit(leg, {"value": nf * 10 / dt2, "unit": "frames/s", "roofline": {"frac": 2 * fb2 * nf * 10 / (dms * 1e-3) / 8e12, "alg_bytes_per_call": 2 * fb2 * nf, "basis": "dominant kernel",
                                                                             "kernel_match": "bilateral_walk16_kernel"}})
        del keep
    elif leg == "ssimulacra2_4k":
        st, keep = bench.setup_ssimulacra2(dev, bench.W4K, bench.H4K, 16)
        dt3, kms, _, _ = timed.run(st, 5, 1)
        emit(leg, {"value": 80 / dt3, "unit": "pairs/s", "roofline": {"frac": 2 * 3 * bench.W4K * bench.H4K * 4 * 80 / dt3 / 8e12, "alg_bytes_per_call": 2 * 3 * bench.W4K * bench.H4K * 4 * 16, "basis": "wall"}})
        del keep
