"""world_size-2 gloo test of the multi-GPU layer (CPU): round-robin frame ownership covers
every frame exactly once, and the per-clip scalar all-reduce reproduces the serial
accumulation of XPSNR's {sum_wdist, sum_xpsnr, num_frames} (reference
src/vapoursynth/xpsnr.zig:89-96) and the SSIMULACRA2 mean."""
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


def _worker(rank, world, port, nframes, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist

    import vszip_amd
    from vszip_amd import cluster

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng_all = np.random.default_rng(123)
    wsse = rng_all.integers(1, 10**9, size=(nframes, 3)).astype(np.float64)  # synthetic per-frame wsse64
    score = rng_all.uniform(20, 95, size=nframes)
    mine = list(cluster.frames_of_rank(nframes, rank, world))
    acc = np.zeros(9)
    for n in mine:
        acc[0:3] += np.sqrt(wsse[n])
        acc[3:6] += 10.0 * np.log10(1e12 / wsse[n])
        acc[6] += 1
        acc[7] += score[n]
        acc[8] += 1
    tot = cluster.allreduce_clip_scalars(acc)
    np.save(Path(out_dir) / f"r{rank}.npy", tot)
    np.save(Path(out_dir) / f"own{rank}.npy", np.array(mine))
    dist.destroy_process_group()


def test_frame_sharding_and_scalar_allreduce(tmp_path):
    import torch.multiprocessing as mp

    world, nframes = 2, 37
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, nframes, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    owned = np.concatenate([np.load(tmp_path / f"own{r}.npy") for r in range(world)])
    assert sorted(owned.tolist()) == list(range(nframes))
    rng_all = np.random.default_rng(123)
    wsse = rng_all.integers(1, 10**9, size=(nframes, 3)).astype(np.float64)
    score = rng_all.uniform(20, 95, size=nframes)
    serial = np.concatenate([np.sqrt(wsse).sum(0), (10.0 * np.log10(1e12 / wsse)).sum(0), [nframes], [score.sum()], [nframes]])
    for r in range(world):
        tot = np.load(tmp_path / f"r{r}.npy")
        np.testing.assert_allclose(tot, serial, rtol=1e-12)
    assert tot[7] / tot[8] == np.float64(score.sum() / nframes) or abs(tot[7] / tot[8] - score.mean()) < 1e-12


def test_owner_rule_matches_plugin():
    """The plugin picks device n mod #GPUs (vszip_plugin.cpp gpu_for_frame); ranks use the same rule."""
    sys.path.insert(0, str(ROOT))
    import vszip_amd
    from vszip_amd import cluster

    for world in (1, 2, 4, 8):
        for n in range(50):
            assert cluster.owner_of_frame(n, world) == n % world
            assert n in cluster.frames_of_rank(50, n % world, world)


def test_bench_gpus_n_launches_n_ranks():
    """`python bench.py --gpus N` must start N ranks itself (round 1: --gpus was parsed and ignored).
    --dry-launch prints the child command instead of running it; no GPU needed."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "1", "--dry-launch"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["dry_launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "7", "--warmup", "1"]  # the ranks get the same arguments, minus --dry-launch


def test_bench_rejects_world_size_mismatch():
    """Started by an outer launcher with the wrong number of ranks, bench.py must not print an n_gpus it did not run."""
    import subprocess

    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""
