#!/usr/bin/env python3
"""GPU box: does the ring kernel's placement sensitivity follow the VIRTUAL or the PHYSICAL address of the destination
arena? (1) Several physical allocations (hipMemCreate), each mapped at one virtual range: the usual spread. (2) ONE
physical allocation mapped at several virtual ranges (different alignments): if the time changes with the alias, it is
the virtual address (TLB); if all aliases of one physical arena give the same time, it is the physical one."""
import ctypes as C
import subprocess
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

so = ROOT / "tools" / "vmm" / "libvmm_alias.so"
if not so.is_file():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-O2", "-o", str(so), str(so.with_name("vmm_alias.hip"))])
vmm = C.CDLL(str(so))
vmm.vmm_granularity.restype = C.c_size_t
vmm.vmm_create.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
vmm.vmm_map.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]

dev = vszip_amd.Device(0)
frames, radius = 64, 13
base = bench.make_frame(0, bench.W4K, bench.H4K)
planes = [np.roll(p, f * 17 + 1, axis=1) for f in range(frames) for p in base]
shapes = [p.shape for p in planes]
src = bench.Arena(dev, shapes, np.uint16, 1)
for a, d in zip(planes, src.planes):
    a = np.ascontiguousarray(a)
    dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * 2, a.ctypes.data, a.strides[0], a.shape[1] * 2, a.shape[0]))
dev.sync()
gran = vmm.vmm_granularity(0)
offs, total = [], 0
for h, w in shapes:
    total = (total + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    offs.append(total)
    total += h * w * 2
total = (total + gran - 1) // gran * gran
print("granularity", gran, "arena bytes", total)


def views(va):
    return [DevPlane(dev, va + o, w, h, w, np.uint16, own=False) for o, (h, w) in zip(offs, shapes)]


def run(dst_planes, n=30):
    table = dev.plane_table(src.planes, dst_planes)
    for _ in range(4):
        dev.boxblur_table(np.uint16, table, radius, 1, radius, 1)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        dev.boxblur_table(np.uint16, table, radius, 1, radius, 1)
    dev.sync()
    return (time.perf_counter() - t0) / n * 1e6


nphys = int(sys.argv[1]) if len(sys.argv) > 1 else 24
phys = []
for k in range(nphys):
    h = C.c_void_p()
    assert vmm.vmm_create(0, total, C.byref(h)) == 0
    va = C.c_void_p()
    assert vmm.vmm_map(0, h, total, 2 << 20, 0, C.byref(va)) == 0
    t = run(views(va.value), 10)
    phys.append((t, h, va.value))
    print(f"physical arena {k:2d} at va {va.value:#x}: {t:6.1f} us", flush=True)
order = sorted(range(nphys), key=lambda k: phys[k][0])
for label, k in (("fastest", order[0]), ("slowest", order[-1]), ("median", order[nphys // 2])):
    t0, h, va0 = phys[k]
    row = [f"first mapping {run(views(va0)):6.1f}"]
    for skew in (0, 0, 4096, 65536, 1 << 20):  # a skewed alias: virtual and physical addresses disagree modulo 2 MiB
        va = C.c_void_p()
        assert vmm.vmm_map(0, h, total, 2 << 20, skew, C.byref(va)) == 0
        row.append(f"alias+{skew} {run(views(va.value)):6.1f}")
    print(f"{label} physical arena ({t0:.1f} us): " + "  ".join(row), flush=True)
