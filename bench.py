#!/usr/bin/env python3
"""Headline bench: vszip.BoxBlur(hradius=vradius=13) on 3840x2160 YUV420P16.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--no-cpu]

A step is one pass of the hot path over one batch of F synthetic frames that are
already resident in HBM (F*3 planes -> one or two kernel launches). For N > 1 the
driver starts one process per GPU through torch.distributed.run; frames shard
across ranks with no data-path collective (weak scaling: every rank owns its own
F frames), the only collective is the max-reduce of the timings.
Rank 0 prints ONE JSON line (see README / DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

W4K, H4K = 3840, 2160
RADIUS = 13
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def yuv420_shapes(w, h):
    return [(h, w), (h // 2, w // 2), (h // 2, w // 2)]


def make_frame(seed: int, w: int, h: int):
    """Deterministic noise planes (splitmix64), u16 full range."""
    import fixtures as fx

    return [fx.splitmix64_plane(0x5A170000 + 16 * seed + p, s, np.uint16) for p, s in enumerate(yuv420_shapes(w, h))]


def cpu_baseline(seconds_budget: float = 15.0):
    """The CPU oracle (a scalar C++ port of the reference arithmetic) timed on this
    box's host cores, one 4K frame per thread, on a bounded sample."""
    from oracle import oracle as orc

    orc.build()
    cores = os.cpu_count() or 1
    frame = make_frame(0, W4K, H4K)

    def one(_):
        for p in frame:
            orc.boxblur(p, RADIUS, 1, RADIUS, 1)

    t0 = time.perf_counter()
    one(0)
    t1 = time.perf_counter() - t0  # single-thread time per frame
    nframes = max(cores, int(seconds_budget / max(t1, 1e-3)) // 1)
    nframes = min(nframes, 64 * cores)
    nframes = (nframes // cores) * cores or cores
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(one, range(nframes)))
        dt = time.perf_counter() - t0
    return {
        "value": nframes / dt, "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"{nframes} frames 3840x2160 YUV420P16 BoxBlur r=13, one frame per thread, {cores} threads, {dt:.1f}s",
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=16, help="frames per step per GPU")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--radius", type=int, default=RADIUS, help="development: BoxBlur radius (headline = 13)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        print(f"WORLD_SIZE={world} != --gpus {a.gpus}", file=sys.stderr)

    # torch first: its bundled HIP runtime must be the one libvszip_hip.so binds to
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import vszip_amd

    dev = vszip_amd.Device(local_rank)
    F = a.frames
    srcs, dsts = [], []
    base = make_frame(rank, W4K, H4K)
    for f in range(F):
        for p, plane in enumerate(base):
            # distinct buffers per frame; content = noise rolled by the frame index
            srcs.append(dev.upload(np.roll(plane, f * 17 + 1, axis=1)))
            dsts.append(dev.empty(plane.shape[0], plane.shape[1], plane.dtype))
    table = dev.plane_table(srcs, dsts)

    def step():
        dev.boxblur_table(np.uint16, table, a.radius, 1, a.radius, 1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dev.sync()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    dev.timer_start()
    for _ in range(a.steps):
        step()
    kernel_ms = dev.timer_stop_ms()  # HIP events on the kernels' stream; synchronises it
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        frame_bytes = sum(2 * s[0] * s[1] for s in yuv420_shapes(W4K, H4K))  # 24 883 200
        launches_per_step = -(-len(srcs) // 48)
        alg_bytes_per_launch = 2 * frame_bytes * F / launches_per_step  # read once + write once
        avg_launch_s = kernel_ms * 1e-3 / (a.steps * launches_per_step)
        achieved = alg_bytes_per_launch / avg_launch_s / 1e9
        out = {
            "metric": "frames/sec at 4K YUV420P16: BoxBlur r=13 (Bilateral, SSIMULACRA2 reported by their own workloads)",
            "value": world * F * a.steps / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt * 1e3 / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
            "config": {
                "workload": "vszip.BoxBlur hradius=vradius=13, 3840x2160 YUV420P16, splitmix64 noise, HBM-resident",
                "frames_per_step_per_gpu": F, "parallelism": f"frame-parallel x{world}",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": "boxblur_ct_int_kernel<u16,13>", "avg_launch_us": avg_launch_s * 1e6,
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            },
        }
        if not a.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    dev.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
