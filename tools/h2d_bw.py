"""GPU box: host-to-device rate of 128 MiB planes by the kind of host memory: hipHostMalloc, plain malloc (the runtime pins it in place),
2 MiB-aligned anonymous memory with MADV_HUGEPAGE, the context's arena (CPU copy + DMA). Round 6 ran it with a registration cache as well
(profiles/r06_pin_cache_attempt.patch): every row but the arena's reads 56.5 GB/s."""
import ctypes as C
import mmap
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
import vszip_amd

if "--bind" in sys.argv:
    import torch  # noqa: F401

    print("numa node", bench.bind_to_gpu_numa(0))
print("THP:", Path("/sys/kernel/mm/transparent_hugepage/enabled").read_text().strip())
dev = vszip_amd.Device(0)
H, W = 4096, 8192  # x4 bytes = 128 MiB
N = 6
d = dev.empty(H, W, np.float32)


def rate(name, arrs, pin=False):
    for rep in range(2):
        t0 = time.perf_counter()
        for a in arrs:
            dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * 4, a.ctypes.data, a.strides[0], W * 4, H))
        dev.sync()
        dt = time.perf_counter() - t0
    print(f"{name:52s} {len(arrs) * H * W * 4 / dt / 1e9:6.1f} GB/s")


pinned = [dev.pinned_array((H, W), np.float32) for _ in range(N)]
for a in pinned:
    a[...] = 1.0
rate("hipHostMalloc", pinned, False)
plain = [np.ones((H, W), np.float32) for _ in range(N)]
rate("malloc, runtime's pageable path", plain, False)
maps = []
huge = []
for _ in range(N):
    m = mmap.mmap(-1, H * W * 4 + (2 << 20))
    m.madvise(mmap.MADV_HUGEPAGE)
    base = C.addressof(C.c_char.from_buffer(m))
    off = (-base) % (2 << 20)
    a = np.frombuffer(m, dtype=np.float32, count=H * W, offset=off).reshape(H, W)
    a[...] = 1.0
    maps.append(m)
    huge.append(a)
rate("2 MiB-aligned + MADV_HUGEPAGE, runtime's pageable path", huge, False)
dev.set_staging(1)
rate("malloc through the context's arena (CPU copy + DMA)", plain, False)
dev.set_staging(0)
