#!/usr/bin/env python3
"""GPU box: Bilateral sigmaS=2 sigmaR=2 YUV420P16 at 1080p (64 frames per call) and 4K (16 frames per call), frames/s — for A/B of library variants."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench, vszip_amd
dev = vszip_amd.Device(0)
out = []
for w, h, f in ((1920, 1080, 64), (3840, 2160, 16)):
    step, keep = bench.setup_bilateral(dev, w, h, f)
    for _ in range(3): step()
    dev.sync(); t = time.perf_counter()
    for _ in range(12): step()
    dev.sync(); out.append(f * 12 / (time.perf_counter() - t))
    del step, keep
print("1080p %.0f fps   4K %.0f fps" % tuple(out))
