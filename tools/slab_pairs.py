#!/usr/bin/env python3
"""GPU box: is the BoxBlur launch time a property of the source arena, of the destination arena, or of the pair?
Three source arenas and three destination arenas (separate allocations, identical layout: planes 2 MiB aligned),
all nine pairings, plus a read-only pass (PlaneAverage) over every arena."""
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
from vszip_amd.capi import DevPlane

dev = vszip_amd.Device(0)
frames, radius = 64, 13
base = bench.make_frame(0, bench.W4K, bench.H4K)
shapes = [p.shape for p in base] * frames
host = [np.ascontiguousarray(np.roll(p, 18, axis=1)) for p in base]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def _hip():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError("no HIP runtime mapped")


HIP = _hip()
HIP.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 4 = hipDeviceMallocContiguous
SKEW = len(sys.argv) > 3 and sys.argv[3] == "skew"  # plane k starts a random multiple of 256 B (< 1 MiB) further on
rng = np.random.default_rng(2)


def arena(fill):
    offs, total = [], 0
    for h, w in shapes:
        total = (total + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        o = total + (int(rng.integers(0, 4096)) * 256 if SKEW else 0)
        offs.append(o)
        total = o + h * w * 2
    p = C.c_void_p()
    if FLAGS:
        rc = HIP.hipExtMallocWithFlags(C.byref(p), total + 256, FLAGS)
        assert rc == 0 and p.value, ("hipExtMallocWithFlags", rc)
    else:
        dev.check(dev.lib.vszip_dev_alloc(dev.ctx, total + 256, C.byref(p)))
    planes = [DevPlane(dev, p.value + o, w, h, w, np.uint16, own=False) for o, (h, w) in zip(offs, shapes)]
    if fill:
        for i, d in enumerate(planes):
            a = host[i % 3] if host is not None else np.ascontiguousarray(np.roll(base[i % 3], (i // 3) * 17 + 1, axis=1))
            dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * 2, a.ctypes.data, a.strides[0], a.shape[1] * 2, a.shape[0]))
        dev.sync()
    return planes, p.value


def timeit(fn, n=150):
    for _ in range(8):
        fn()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dev.sync()
    return (time.perf_counter() - t0) / n * 1e6


OPTS = sys.argv[4:]
hold = []
if "prealloc" in OPTS:  # 384 per-plane allocations first, like bench.py's own buffers
    for h, w in shapes * 2:
        hold.append(dev.empty(h, w, np.uint16))
if "roll" in OPTS:  # every frame its own content
    host = None
S = [arena(True) for _ in range(N)]
D = [arena(False) for _ in range(N)]
print("src arena VAs:", [hex(v) for _, v in S], " dst arena VAs:", [hex(v) for _, v in D])
for i, (s, _) in enumerate(S):
    row = []
    for j, (d, _) in enumerate(D):
        table = dev.plane_table(s, d)
        row.append(timeit(lambda: dev.boxblur_table(np.uint16, table, radius, 1, radius, 1)))
    print(f"S{i} x D0..D{N - 1}: " + "  ".join(f"{t:6.1f}" for t in row), flush=True)
# a source arena blurred into another source arena (arena roles swapped), and in place is not allowed; read-only pass:
for name, arr in (("S", S), ("D", D)):
    for i, (pl, _) in enumerate(arr):
        t = timeit(lambda: dev.plane_average(pl[:48]), 40)
        print(f"PlaneAverage over 16 frames of {name}{i}: {t:7.1f} us", flush=True)
