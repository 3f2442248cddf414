#!/usr/bin/env python3
"""GPU box: the float pass-chain kernels for library variants in one process — 4K YUV420PS, FRAMES per call (default 8): us per call for 3 horizontal /
3 vertical passes of r = 5, and bit-equality of the outputs with the first variant's."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch  # noqa
import fixtures as fx
import vszip_amd
from vszip_amd import capi
names = sys.argv[1:] or ["base"]
frames = int(os.environ.get("FRAMES", "8"))
planes = [fx.tiled_natural(s, np.float32, p) for p, s in enumerate([(2160, 3840), (1080, 1920), (1080, 1920)])]
ref = {}
for rnd in range(2):
    for n in names:
        capi.LIB_PATH = ROOT / ("vapoursynth-zip_amd/libvszip_hip.so" if n == "base" else f"tools/ab/{n}.so")
        capi._lib = None
        dev = vszip_amd.Device(0)
        srcs = [dev.upload(np.roll(p, f, axis=1)) for f in range(frames) for p in planes]
        dsts = [dev.empty(p.shape[0], p.shape[1], np.float32) for f in range(frames) for p in planes]
        out = []
        for args in ((5, 3, 0, 0), (0, 0, 5, 3), (13, 2, 0, 0)):
            dev.boxblur(srcs, dsts, *args); dev.sync()
            t0 = time.perf_counter()
            for _ in range(10): dev.boxblur(srcs, dsts, *args)
            dev.sync()
            got = [dev.download(d) for d in dsts[:3]]
            same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, ref.setdefault(args, got)))
            out.append("%s %.0f us%s" % (args, (time.perf_counter() - t0) / 10 * 1e6, "" if same else " DIFFERENT"))
        print(f"{n:8s} " + "  ".join(out), flush=True)
        del srcs, dsts
        dev.close()
