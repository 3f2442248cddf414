"""bench.py's own N-rank arithmetic under a world_size-4 gloo group (CPU). No rank of this pool has ever run
beside another on GPUs, so everything bench.py does ACROSS ranks runs here with the GPU calls replaced by a host stand-in: the
timing rule (MAX over ranks), `value` (all ranks' frames over the slowest rank's time), the gathered per-rank records, and
`xpsnr_clip_leg` - XPSNR's per-clip accumulators from each rank's frames (frame n on rank n mod world), SUM-all-reduced and
checked against a single-rank pass (reference src/vapoursynth/xpsnr.zig:89-96, src/filters/xpsnr.zig:359-368)."""
import json
import os
import socket
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _HostDev:
    """what xpsnr_clip_leg calls on a Device, on the host: planes stay numpy arrays, the weighted SSE is a deterministic integer
    function of the frame (NOT the XPSNR arithmetic - the oracle tests own that; the reduction only needs per-frame integers)"""

    def __init__(self, lib):
        self.lib = lib  # the real library's host-side functions (vszip_xpsnr_value / _average need no GPU)

    def upload(self, a):
        return a

    def xpsnr_wsse_batch(self, orgs, recs, p1, p2, depth, frame_rate, temporal):
        out = []
        for o, r, q in zip(orgs, recs, p1):
            w = [int(((a.astype(np.int64) - b.astype(np.int64)) ** 2).sum()) + 1 for a, b in zip(o, r)]
            if q is not None:
                w[0] += int(np.abs(o[0].astype(np.int64) - q.astype(np.int64)).sum() % 1000003)
            out.append(w)
        return out


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist

    import bench
    import vszip_amd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    res["max_dt"] = bench.ranks_max(0.125 * (rank + 1), True, None)  # the slowest rank took 0.5 s
    res["value"] = bench.whole_job_value(64, 20, world, res["max_dt"])
    recs = bench.gather_rank_records({"rank": rank, "local_rank": rank, "device": rank, "frac": 0.7 - 0.01 * rank}, True)
    res["records"] = recs
    res["xpsnr_clip"] = bench.xpsnr_clip_leg(_HostDev(vszip_amd.capi.load()), vszip_amd, rank, world, None, frames_per_rank=3)
    (Path(out_dir) / f"rank{rank}.json").write_text(json.dumps(res))
    dist.destroy_process_group()


def test_bench_reductions_over_four_ranks(tmp_path):
    import torch.multiprocessing as mp

    world = 4
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    res = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(world)]
    for r in res:
        assert r["max_dt"] == 0.5                      # every rank learns the slowest rank's time
        assert r["value"] == 4 * 64 * 20 / 0.5         # all ranks' frames over it
        assert [q["rank"] for q in r["records"]] == [0, 1, 2, 3] and [q["device"] for q in r["records"]] == [0, 1, 2, 3]
        x = r["xpsnr_clip"]
        assert x["frames"] == 12 and x["reduced_over_ranks"] == 4
        assert x["avg_xpsnr_yuv"] == res[0]["xpsnr_clip"]["avg_xpsnr_yuv"]  # the all-reduce leaves every rank with the same clip average
    x0 = res[0]["xpsnr_clip"]
    assert x0["matches_single_rank"] is True and x0["max_rel_diff_vs_single_rank"] <= 1e-12  # rank 0 re-did the whole clip alone
    assert all(40.0 < v < 200.0 or v == float("inf") for v in x0["avg_xpsnr_yuv"]), x0
    # the committed stand-in for a first real multi-rank line (profiles/r05_ranks_cpu.json) is what this test produces
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "r5_ranks_cpu.json").write_text(json.dumps({"world": world, "backend": "gloo", "rank0": res[0]}, indent=1))


def test_dry_launch_starts_one_child_per_gpu():
    """`python bench.py --gpus N` without a rendezvous: the ranks are CHILD processes started before anything touches a GPU"""
    import subprocess

    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-launch"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["dry_launch"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd and cmd[-4:] == ["--steps", "5", "--warmup", "1"] or "--gpus" in cmd
    assert "--dry-launch" not in cmd
