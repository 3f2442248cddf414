#!/usr/bin/env python3
"""A/B of library variants (tools/variant.sh -> tools/ab/<name>.so) on ONE set of device planes in one process:
BoxBlur r=13 on 64 x 3840x2160 YUV420P8 (or P16 with --u16). usage: ab_u8_inproc.py [--u16] base name name ..."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import fixtures as fx  # noqa: E402
import vszip_amd  # noqa: E402
from vszip_amd import capi  # noqa: E402


def device_of(name):
    if name != "base":
        capi.LIB_PATH = ROOT / "tools" / "ab" / f"{name}.so"
        capi._lib = None
    else:
        capi.LIB_PATH = ROOT / "vapoursynth-zip_amd" / "libvszip_hip.so"
        capi._lib = None
    return vszip_amd.Device(0)


def main():
    args = sys.argv[1:]
    dt = np.uint16 if "--u16" in args else np.uint8
    names = [a for a in args if not a.startswith("--")] or ["base"]
    radius = next((int(a.split("=")[1]) for a in args if a.startswith("--radius=")), 13)
    dev0 = device_of("base")
    base = [fx.splitmix64_plane(p, s, dt) for p, s in enumerate(bench.yuv420_shapes(bench.W4K, bench.H4K))]
    srcs, dsts = [], []
    for f in range(64):
        for pl in base:
            srcs.append(dev0.upload(np.roll(pl, f * 17 + 1, axis=1)))
            dsts.append(dev0.empty(pl.shape[0], pl.shape[1], pl.dtype))
    dev0.sync()
    want = None
    devs = {n: (dev0 if n == "base" else device_of(n)) for n in names}
    nbytes = 2 * sum(a.size * a.itemsize for a in base) * 64
    for rnd in range(3):
        for n in names:
            d = devs[n]
            table = d.plane_table(srcs, dsts)
            timed = bench.Timed(d, lambda: None)
            step = lambda: d.boxblur_table(dt, table, radius, 1, radius, 1)
            _, _, dom_ms, launches = timed.run(step, 30, 3)
            got = dev0.download(dsts[0])[:64].copy()
            if want is None:
                want = got
            ok = np.array_equal(got, want)
            us = dom_ms / launches * 1e3
            print(f"round {rnd} {n:12s} {us:7.1f} us/launch  {nbytes / (us * 1e-6) / 8e12:.3f} of HBM  {'same' if ok else 'DIFFERENT'}", flush=True)


if __name__ == "__main__":
    main()
