#!/usr/bin/env python3
"""GPU box: variants of the ring kernel (store / load cache policy, prefetch depth ...) on a SLOW and on a FAST placement of the
destination arena, all inside one process: variant libraries (tools/variant.sh, -DVSZIP_ST_AUX=n) are loaded side by
side and run on the same arenas."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401

import bench
import vszip_amd

dev = vszip_amd.Device(0)
import os

frames, radius = 64, 13
DT = np.uint8 if os.environ.get("AB_U8") else np.uint16  # AB_U8=1: the 8-bit kernel on YUV420P8
base = bench.make_frame(0, bench.W4K, bench.H4K)
if DT == np.uint8:
    base = [(p >> 8).astype(np.uint8) for p in base]
planes = [np.roll(p, f * 17 + 1, axis=1) for f in range(frames) for p in base]
shapes = [p.shape for p in planes]
src = bench.Arena(dev, shapes, DT, 1)
for a, d in zip(planes, src.planes):
    a = np.ascontiguousarray(a)
    dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * a.itemsize, a.ctypes.data, a.strides[0], a.shape[1] * a.itemsize, a.shape[0]))
dev.sync()


CODE = 0 if DT == np.uint8 else 1


def run(lib, ctx, table, n=40):
    for _ in range(5):
        assert lib.vszip_boxblur(ctx, CODE, table, len(table), radius, 1, radius, 1) == 0
    lib.vszip_ctx_sync(ctx)
    t0 = time.perf_counter()
    for _ in range(n):
        lib.vszip_boxblur(ctx, CODE, table, len(table), radius, 1, radius, 1)
    lib.vszip_ctx_sync(ctx)
    return (time.perf_counter() - t0) / n * 1e6


cands = [bench.Arena(dev, shapes, DT, 100 + k) for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30)]
times = [run(dev.lib, dev.ctx, dev.plane_table(src.planes, c.planes), 10) for c in cands]
order = np.argsort(times)
picks = {"fast": cands[order[0]], "median": cands[order[len(order) // 2]], "slow": cands[order[-1]]}
print("candidates:", [round(t) for t in times])
libs = {"aux2 (built: nt)": (dev.lib, dev.ctx)}
for name in sys.argv[2:]:
    lib = C.CDLL(str(ROOT / "tools" / "ab" / f"{name}.so"))
    lib.vszip_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.vszip_boxblur.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.vszip_ctx_sync.argtypes = [C.c_void_p]
    ctx = C.c_void_p()
    assert lib.vszip_ctx_create(0, C.byref(ctx)) == 0
    libs[name] = (lib, ctx)
for rnd in range(2):
    for lname, (lib, ctx) in libs.items():
        row = []
        for pname, arena in picks.items():
            row.append(f"{pname} {run(lib, ctx, dev.plane_table(src.planes, arena.planes)):6.1f}")
        print(f"{lname:18s} " + "   ".join(row), flush=True)
