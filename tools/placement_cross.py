"""Round 5: is a slow batch slow because of its SOURCE arena, its DESTINATION arena, or the pair? Three batches built one
after another in one fresh process (all held); the launch timed on every (source i, destination j).
    python tools/placement_cross.py <placement 0|1> <piece_mib> <pool_gib>
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    mode, piece, pool = (int(x) for x in sys.argv[1:4])
    import torch  # noqa: F401
    import vszip_amd

    dev = vszip_amd.Device(0)
    timed = bench.Timed(dev, dev.sync, prewarm_s=0.2)
    keeps = []
    for _ in range(3):
        with dev.options(VSZIP_PLACEMENT=mode, VSZIP_PLACEMENT_PIECE_MIB=piece, VSZIP_PLACEMENT_POOL_GIB=pool):
            step, keep = bench.setup_boxblur(dev, 0, 64, 13)
        keeps.append(keep)
    # every source holds the same frames (same seed)
    print(f"mode {mode} piece {piece} pool {pool}; rows = source batch, columns = destination batch (us per launch)")
    for rep in range(2):
        for i, ks in enumerate(keeps):
            row = []
            for j, kd in enumerate(keeps):
                table = dev.plane_table(ks[0].planes, kd[1].planes)
                _, _, dom, n = timed.run(lambda: dev.boxblur_table(np.uint16, table, 13, 1, 13, 1), 60, 3)
                row.append(dom * 1e3 / n)
            print(f"  src {i}: " + " ".join(f"{u:6.0f}" for u in row), flush=True)
        print()
    dev.close()


if __name__ == "__main__":
    main()
