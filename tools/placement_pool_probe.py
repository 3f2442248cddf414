#!/usr/bin/env python3
"""Round 4: calibrate the allocator's placement probe (ctx.hip placement_probe_kernel) against the real ring kernel.

1. placement off: N candidate arenas of the headline batch's size, all held; for each the probe's rate (vszip_dev_probe_region)
   and the time of the real BoxBlur r=13 launch with the candidate as DESTINATION (source = arena 0) and as SOURCE (destination = arena 1).
2. everything freed, placement on: source + destination arenas through plain vszip_dev_alloc, the launch time on them, and
   what the allocator says (vszip_dev_placement_info). Repeated a few times (free -> park -> reuse).
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402


def main():
    import torch  # noqa: F401  (its HIP runtime first)
    import vszip_amd

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    dev = vszip_amd.Device(0)
    dev.set_option("VSZIP_PLACEMENT", 0)
    frames = 64
    base = bench.make_frame(0, bench.W4K, bench.H4K)
    planes = [np.roll(p, f * 17 + 1, axis=1) for f in range(frames) for p in base]
    shapes = [p.shape for p in planes]
    src = bench.Arena(dev, shapes, np.uint16, 1)
    for a, d in zip(planes, src.planes):
        a = np.ascontiguousarray(a)
        dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * 2, a.ctypes.data, a.strides[0], a.shape[1] * 2, a.shape[0]))
    dev.sync()
    lay = bench.Arena(dev, shapes, np.uint16, 2, ptr=0)

    def launch_us(s, d, nrep=10):
        table = dev.plane_table(s.planes, d.planes)
        for _ in range(2):
            dev.boxblur_table(np.uint16, table, 13, 1, 13, 1)
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(nrep):
            dev.boxblur_table(np.uint16, table, 13, 1, 13, 1)
        dev.sync()
        return (time.perf_counter() - t0) / nrep * 1e6

    import ctypes as C

    cands = []
    print(f"# arena bytes {lay.nbytes / 2**30:.3f} GiB; columns: index, probe TB/s (x3), ring us as destination, ring us as source", flush=True)
    dst0 = None
    for k in range(n):
        p = C.c_void_p()
        if dev.lib.vszip_dev_alloc(dev.ctx, lay.nbytes, C.byref(p)) != 0:
            break
        cands.append(p.value)
        d = lay.view(p.value)
        t_dst = launch_us(src, d)
        pair = [dev.probe_region(p.value, min(lay.nbytes, src.nbytes), src.ptr) / 1e12 for _ in range(2)]
        t_dst2 = launch_us(src, d)
        rates = [dev.probe_region(p.value, lay.nbytes) / 1e12 for _ in range(3)]
        if dst0 is None:
            dst0 = d
        # as a source: copy the planes in (the probe overwrote them), then time against the first candidate as destination
        s2 = src.view(p.value)
        for a_, b_ in zip(src.planes, s2.planes):
            dev.check(dev.lib.vszip_copy_d2d_2d(dev.ctx, b_.ptr, b_.stride * 2, a_.ptr, a_.stride * 2, a_.w * 2, a_.h))
        dev.sync()
        t_src = launch_us(s2, dst0) if k > 0 else float("nan")
        print(f"{k:3d}  {rates[0]:.3f} {rates[1]:.3f} {rates[2]:.3f}   dst {t_dst:7.1f} {t_dst2:7.1f}   src {t_src:7.1f}   pair-probe {pair[0]:.3f} {pair[1]:.3f}", flush=True)
    for p in cands:
        dev.lib.vszip_dev_free(dev.ctx, p)
    src.free()
    dev.sync()

    # 2: the allocator on its own
    dev.set_option("VSZIP_PLACEMENT", 1)
    for rnd in range(4):
        t0 = time.perf_counter()
        s = bench.Arena(dev, shapes, np.uint16, 1)
        t1 = time.perf_counter()
        d = bench.Arena(dev, shapes, np.uint16, 2)
        t2 = time.perf_counter()
        for a, q in zip(planes, s.planes):
            a = np.ascontiguousarray(a)
            dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, q.ptr, q.stride * 2, a.ctypes.data, a.strides[0], a.shape[1] * 2, a.shape[0]))
        dev.sync()
        us = launch_us(s, d, 20)
        print(f"placed round {rnd}: alloc src {1e3 * (t1 - t0):.0f} ms, dst {1e3 * (t2 - t1):.0f} ms; ring {us:.1f} us = {3185049600 / (us * 1e-6) / 8e12:.3f} of 8 TB/s; "
              f"src {dev.placement_info(s.ptr)}, dst rate {dev.placement_info(d.ptr)['bytes_per_second'] / 1e12:.3f} TB/s", flush=True)
        s.free()
        d.free()
    print("trim:", dev.trim(), "bytes;", dev.placement_info())
    # 3: how long do hipMalloc / hipFree of such arenas take once memory has been used before? (placement off)
    dev.set_option("VSZIP_PLACEMENT", 0)
    held = []
    t0 = time.perf_counter()
    for k in range(12):
        p = C.c_void_p()
        dev.lib.vszip_dev_alloc(dev.ctx, lay.nbytes, C.byref(p))
        held.append(p.value)
    t1 = time.perf_counter()
    for p in held:
        dev.probe_region(p, lay.nbytes)
    t2 = time.perf_counter()
    for p in held:
        dev.lib.vszip_dev_free(dev.ctx, p)
    t3 = time.perf_counter()
    print(f"12 arenas: hipMalloc {1e3 * (t1 - t0) / 12:.1f} ms each, probe {1e3 * (t2 - t1) / 12:.1f} ms each, hipFree {1e3 * (t3 - t2) / 12:.1f} ms each", flush=True)
    # 4: the same device, bench.py's round-3 search (3 walks of 24 candidates timing the real launch)
    step, keep, info = bench.placed_batch(dev, planes, np.uint16, (13, 1, 13, 1), 0, 24, (12, 1, 12, 1))
    table = dev.plane_table(keep[0].planes, keep[1].planes)
    print("bench search (24 tries):", {k: v for k, v in info.items() if k != "note"}, flush=True)
    print(f"  -> ring {launch_us(keep[0], keep[1], 20):.1f} us; probe rate of its arenas: src {dev.probe_region(keep[0].ptr, keep[0].nbytes) / 1e12:.3f}, dst {dev.probe_region(keep[1].ptr, keep[1].nbytes) / 1e12:.3f} TB/s")
    dev.close()


if __name__ == "__main__":
    main()
