"""Round 5: does WHERE in physical memory the pieces come from matter beyond their being apart? One fresh process per
line (a process's first allocations get the lowest free addresses): `ballast` GiB of plain hipMalloc held first, then the
headline batch (64 x 4K YUV420P16, BoxBlur r = 13) on arenas allocated with the given allocator options.
    python tools/placement_ballast.py <ballast_gib> <placement 0|1> <piece_mib> <pool_gib> [repeat=2]
"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    ballast, mode, piece, pool = (int(x) for x in sys.argv[1:5])
    repeat = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    import torch  # noqa: F401
    import vszip_amd

    dev = vszip_amd.Device(0)
    held = []
    with dev.options(VSZIP_PLACEMENT=0):
        for _ in range(ballast // 8):  # 8 GiB blocks
            p = C.c_void_p()
            dev.check(dev.lib.vszip_dev_alloc(dev.ctx, 8 << 30, C.byref(p)))
            held.append(p.value)
    timed = bench.Timed(dev, dev.sync, prewarm_s=0.2)
    alg = 2 * 64 * 24883200
    res, keep_all = [], []
    for _ in range(repeat):
        with dev.options(VSZIP_PLACEMENT=mode, VSZIP_PLACEMENT_PIECE_MIB=piece, VSZIP_PLACEMENT_POOL_GIB=pool):
            step, keep = bench.setup_boxblur(dev, 0, 64, 13)
        _, _, dom, n = timed.run(step, 100, 5)
        res.append(dom * 1e3 / n)
        keep_all.append(keep)
    print(f"ballast {ballast:3d} GiB mode {mode} piece {piece:3d} pool {pool:2d}: " + " ".join(f"{u:.0f} us ({alg / (u * 1e-6) / 8e12:.3f})" for u in res), flush=True)
    dev.close()


if __name__ == "__main__":
    main()
