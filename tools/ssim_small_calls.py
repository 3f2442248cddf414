"""GPU box: SSIMULACRA2 on linear RGBS, 1 / 2 / 3 / 4 pairs a call at 1080p and 4K: pairs/s (for library A/B with tools/ab_script.sh)."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench
import vszip_amd

dev = vszip_amd.Device(0)
out = []
for (w, h) in ((1920, 1080), (3840, 2160)):
    ref, dis = bench.rgbs_pair(w, h)
    r = [dev.upload(p) for p in ref]
    d = [dev.upload(p) for p in dis]
    for n in ((1, 2, 3, 4) if len(sys.argv) < 2 else [int(v) for v in sys.argv[1].split(',')]):
        rr, dd = r * n, d * n
        dev.ssimulacra2(rr, dd)
        reps = max(4, 48 // n)
        t0 = time.perf_counter()
        for _ in range(reps):
            dev.ssimulacra2(rr, dd)
        out.append(f"{w}x{h} x{n}: {n * reps / (time.perf_counter() - t0):.0f}")
print(" | ".join(out))
