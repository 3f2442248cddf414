#!/usr/bin/env python3
"""GPU box: run the same BoxBlur / Bilateral / SSIMULACRA2 / EEDI3 launches many times and compare every output with the
first run's, bit for bit - timing-dependent hardware hazards and stream races show up as rare mismatches."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch  # noqa: F401
import bench
import fixtures as fx
import vszip_amd

dev = vszip_amd.Device(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad_total = 0


def planes_of(dtype, w, h, frames, seed):
    base = [fx.splitmix64_plane(seed + p, s, dtype) for p, s in enumerate(bench.yuv420_shapes(w, h))]
    return [np.roll(p, f, axis=1) for f in range(frames) for p in base]


def repeat(name, run, outs, n=N):
    global bad_total
    run()
    dev.sync()
    first = [dev.download(o).copy() for o in outs]
    bad = 0
    for it in range(n):
        run()
        dev.sync()
        for k in range(it % 3, len(outs), 3):  # a third of the planes per iteration
            if not np.array_equal(dev.download(outs[k]).view(np.uint8), first[k].view(np.uint8)):
                bad += 1
    print(f"{name}: {n} runs, mismatching planes {bad}", flush=True)
    bad_total += bad


ONLY_MINMAX = len(sys.argv) > 3 and sys.argv[3] == "minmax"  # python tools/stress_repeat.py 60 200 minmax
if not ONLY_MINMAX:
    for dtype, r, frames in [(np.uint16, 13, 16), (np.uint8, 13, 16), (np.uint16, 5, 16), (np.float32, 3, 4), (np.float32, 13, 4), (np.float16, 7, 4), (np.float32, 22, 4)]:
        pl = planes_of(dtype, bench.W4K, bench.H4K, frames, 7)
        srcs = [dev.upload(p) for p in pl]
        dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in pl]
        table = dev.plane_table(srcs, dsts)
        repeat(f"boxblur {np.dtype(dtype).name} r={r} x{frames} 4K", lambda: dev.boxblur_table(dtype, table, r, 1, r, 1), dsts)
        del srcs, dsts, table

    pl = planes_of(np.uint16, bench.W1080, bench.H1080, 16, 11)
    srcs = [dev.upload(p) for p in pl]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in pl]
    cfg = dev.bilateral_cfg([2], [2], yuv=True, ssw=1, ssh=1, hist_len=65536)
    idx = [i % 3 for i in range(len(pl))]
    repeat("bilateral u16 1080p x16", lambda: dev.bilateral(srcs, dsts, cfg, idx), dsts)

    plf = [np.ascontiguousarray(fx.tiled_natural(s, np.float32, i % 3)) for i, s in enumerate(bench.yuv420_shapes(bench.W1080, bench.H1080) * 4)]
    fs = [dev.upload(p) for p in plf]
    outs = dev.eedi3(fs, 1, dh=True)
    table = dev.plane_table(fs, outs)
    prm = bench._eedi3_params()
    os.environ["VSZIP_EEDI3_FORCE_OVERLAP"] = "1"  # (the library forks the second stream from 12 frames per call on)
    repeat("eedi3 1080p YUV420PS x4 (two plane heights: overlap path)", lambda: dev.check(dev.lib.vszip_eedi3(dev.ctx, table, None, None, len(fs), 1, 0, prm)), outs, n=max(10, N // 3))

    ref = [np.ascontiguousarray(fx.tiled_natural((1080, 1920), np.float32, p)) for p in range(3)]
    rng = np.random.default_rng(3)
    rr, dd = [], []
    for i in range(6):
        rr += [dev.upload(p) for p in ref]
        dd += [dev.upload(np.clip(p + rng.normal(0, 0.01 * (i + 1), p.shape).astype(np.float32), 0, 1).astype(np.float32)) for p in ref]
    first = dev.ssimulacra2(rr, dd)
    bad = sum(1 for _ in range(N) if [float(x).hex() for x in dev.ssimulacra2(rr, dd)] != [float(x).hex() for x in first])
    print(f"ssimulacra2 1080p x6 pairs (halves on two streams): {N} runs, mismatching calls {bad}")
    bad_total += bad

# Thresholded PlaneMinMax: the sweeps hand their tables to each plane's last workgroup without a device-scope fence (planestats.hip "THE RULE"):
# 192 planes a call (every plane's last workgroup reads tables built on all eight XCDs), 8- and 16-bit and float, with and without a second clip,
# every result of every repeat against the first call's (the hazard hunter now includes hist_sweep_kernel / hist_refine_kernel).
NM = int(sys.argv[2]) if len(sys.argv) > 2 else max(200, N)
for dtype, w, h in [(np.uint16, 960, 540), (np.uint8, 960, 540), (np.float32, 640, 360), (np.uint16, 1920, 1080)]:
    rng = np.random.default_rng(17)
    frames = 64 if w < 1920 else 16
    if dtype == np.float32:
        pl = [np.ascontiguousarray(fx.tiled_natural(s, np.float32, i % 3)) for i, s in enumerate(bench.yuv420_shapes(w, h) * frames)]
        pl = [np.roll(p, i, axis=1) for i, p in enumerate(pl)]
    else:
        pl = planes_of(dtype, w, h, frames, 23)
        # picture-like content too: every third frame a natural tile (long runs in one histogram bin)
        for f in range(0, frames, 3):
            for k, s in enumerate(bench.yuv420_shapes(w, h)):
                pl[3 * f + k] = np.ascontiguousarray(fx.tiled_natural(s, dtype, k))
    a = [dev.upload(p) for p in pl]
    b = [dev.upload(np.roll(p, 5, axis=0)) for p in pl]
    for thr, other in (((0.1, 0.1), None), ((0.02, 0.3), b), ((0.0, 0.5), None)):
        first = dev.plane_minmax(a, thr[0], thr[1], other)
        bad = 0
        for _ in range(NM):
            got = dev.plane_minmax(a, thr[0], thr[1], other)
            bad += sum(1 for x, y in zip(got, first) if not np.array_equal(np.asarray(x), np.asarray(y)))
        print(f"plane_minmax {np.dtype(dtype).name} {w}x{h} x{len(a)} planes thr={thr} {'with a second clip' if other else ''}: {NM} runs, mismatching result arrays {bad}", flush=True)
        bad_total += bad
    del a, b
print("TOTAL MISMATCHES", bad_total)
sys.exit(1 if bad_total else 0)
