"""GPU box: SSIMULACRA2 from 4K YUV444P8 / YUV422P8 / YUV444P16 pairs, the row pre-stage pass (default) against the fused tile kernel (VSZIP_SSIM_NO_YUV420_LDS=1), interleaved."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench
import vszip_amd

dev = vszip_amd.Device(0)
W, H, PAIRS = 3840, 2160, 16
ref, dis = bench.yuv420p8_pair(W, H)
up2 = lambda p, sx, sy: np.ascontiguousarray(np.repeat(np.repeat(p, sy, axis=0), sx, axis=1))
for name, ssw, bits in (("YUV444P8", 0, 8), ("YUV422P8", 1, 8), ("YUV444P16", 0, 16), ("YUV422P10", 1, 10)):
    sx = 1 if ssw else 2
    conv = (lambda p: p) if bits == 8 else (lambda p: (p.astype(np.uint16) << (bits - 8)))
    r3 = [conv(ref[0]), conv(up2(ref[1], sx, 2)), conv(up2(ref[2], sx, 2))]
    d3 = [conv(dis[0]), conv(up2(dis[1], sx, 2)), conv(up2(dis[2], sx, 2))]
    fmt = dev.ssim_source("YUV", r3[0].dtype, bits, ssw=ssw, ssh=0, matrix=1, chroma_loc=0)
    r, d = [], []
    for p in range(PAIRS):
        r += [dev.upload(np.roll(x, p * 8, axis=1)) for x in r3]
        d += [dev.upload(np.roll(x, p * 8, axis=1)) for x in d3]
    res = {}
    for rnd in range(3):
        for mode in (0, 1):
            with dev.options(VSZIP_SSIM_NO_YUV420_LDS=mode):
                s = dev.ssimulacra2_src(fmt, r, d)
                t0 = time.perf_counter()
                for _ in range(4):
                    s2 = dev.ssimulacra2_src(fmt, r, d)
                dt = time.perf_counter() - t0
            res.setdefault(mode, []).append(PAIRS * 4 / dt)
            res.setdefault(("s", mode), s2)
    print(name, "row pass", [round(v) for v in res[0]], "fused tile kernel", [round(v) for v in res[1]], "equal scores", res[("s", 0)] == res[("s", 1)], flush=True)
    del r, d
