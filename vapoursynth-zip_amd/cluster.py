"""Multi-GPU layer: frames of a clip shard embarrassingly across the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in
the CPU tests). There is NO data-path collective: frame n belongs to rank n mod world
(the same rule the plugin applies inside one process, n mod #GPUs). The only exchange is the
per-clip reduction of metric scalars, the one place the reference aggregates over a clip:
XPSNR's {sum_wdist[3], sum_xpsnr[3], num_frames} under its mutex
(src/vapoursynth/xpsnr.zig:89-96, consumed by getAvgXPSNR src/filters/xpsnr.zig:359-368) and —
a build-side addition — SSIMULACRA2's {sum_score, count}. That is at most 8 f64 per clip:
one latency-bound all-reduce on its own at clip end, never per frame.
"""
from __future__ import annotations

import numpy as np


def frames_of_rank(num_frames: int, rank: int, world: int):
    """Frame indices owned by `rank`: round-robin by frame index."""
    return range(rank, num_frames, world)


def owner_of_frame(n: int, world: int) -> int:
    return n % world


def allreduce_clip_scalars(local: np.ndarray, group=None, device=None) -> np.ndarray:
    """SUM-all-reduce a small f64 vector of per-clip accumulators across ranks."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(local, np.float64).copy()
    t = torch.tensor(np.asarray(local, np.float64), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def xpsnr_clip_average(sum_wdist, sum_xpsnr, num_frames, widths, heights, depth, capi_lib):
    """getAvgXPSNR per plane from globally reduced accumulators (what xpsnrFree prints)."""
    return [capi_lib.vszip_xpsnr_average(float(sum_wdist[c]), float(sum_xpsnr[c]), int(widths[c]), int(heights[c]), int(depth), int(num_frames)) for c in range(len(widths))]
