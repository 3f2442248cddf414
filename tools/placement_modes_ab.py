"""Round 5: the headline launch (64 x 4K YUV420P16, BoxBlur r = 13) on arenas from the allocator's
forms, interleaved in ONE process on one device: VSZIP_PLACEMENT = 0 (plain hipMalloc) against 1 (striped: physical pieces
taken evenly spaced from a transient pool, mapped side by side) at several piece and pool sizes.
    python tools/placement_modes_ab.py [rounds=3]
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    import torch  # noqa: F401  (its HIP runtime first)
    import vszip_amd

    dev = vszip_amd.Device(0)
    timed = bench.Timed(dev, dev.sync, prewarm_s=0.3)
    F = 64
    alg = 2 * F * 24883200
    keepers = []
    print("mode piece_MiB    tries  alloc_s  launch_us  frac   (u16 r=13, 64 x 4K)")
    for rnd in range(rounds):
        for mode, piece, pool in ((0, 0, 1), (1, 0, 1), (1, 0, 8), (1, 0, 16), (1, 0, 32)):
            with dev.options(VSZIP_PLACEMENT=mode, VSZIP_PLACEMENT_TRIES=pool):
                step, keep = bench.setup_boxblur(dev, 0, F, 13)
            _, _, dom, n = timed.run(step, 200, 5)
            us = dom * 1e3 / n
            ar = keep[2]["arena"]
            print(f"{mode:4d} {piece:9d} {pool:8d} {keep[2]['alloc_seconds']:8.3f} {us:10.1f}  {alg / (us * 1e-6) / 8e12:.4f}   dst: {ar['candidates']} candidates, probe {ar['probe_bytes_per_second'] / 1e12:.2f} TB/s", flush=True)
            keepers.append(keep)  # held for two rounds: later allocations lie elsewhere
            del step
        if rnd % 2 == 1:
            keepers.clear()
    dev.close()


if __name__ == "__main__":
    main()
