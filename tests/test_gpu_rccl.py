"""RCCL in the driver's own test run: the pool hands out one GPU, so the 8-GPU launch never executes here — but everything
the ranks do with the collective backend does with ONE rank: a child `python -m torch.distributed.run --nproc-per-node=1 bench.py` (a fresh
process, started before it touches the GPU; this process is never replaced) initialises a real "nccl" process group on the device, runs the
barrier inside the clock, the MAX-reduce of the timings and the SUM-all-reduce of XPSNR's per-clip accumulators, and prints the line."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_one_rank_over_a_real_nccl_group(tmp_path):
    env = dict(os.environ, VSZIP_BENCH_FORCE_DIST="1", VSZIP_BENCH_DETAIL=str(tmp_path / "detail.json"), MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--exchange-only"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    assert d["collective_backend"] == "nccl" and d["rccl_ranks"] == 1  # a real RCCL group, not the single-rank shortcut
    assert d["config"]["barrier_ms"] is not None and d["config"]["barrier_ms"] >= 0  # the closing barrier was timed (it is inside the clock)
    xc = d["config"]["xpsnr_clip"]
    assert "error" not in xc, xc
    assert xc["reduced_over_ranks"] == 1 and xc["matches_single_rank"] is True and xc["frames"] == 8
    assert all(40 < v < 70 for v in xc["avg_xpsnr_yuv"])
    lm = d["config"]["clip_mean_luma"]
    assert lm["reduced_over_ranks"] == 1 and 0.4 < lm["value"] < 0.6
