#!/usr/bin/env python3
"""GPU box: the RT legs that take the small-radius kernels — 4K r=5 3+3 passes (8 frames), 1080p r=13 5+5 passes (32 frames), 1080p r=1 2+2 passes u8 (64 frames) — for library A/B."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench, vszip_amd
dev = vszip_amd.Device(0)
timed = bench.Timed(dev, lambda: None)
o = bench.boxblur_other_paths_leg(dev, timed)
p5 = bench.boxblur_1080p_5pass_leg(dev, timed, True)
g = bench.boxblur_gauss_leg(dev, timed)
print("r5x3_4k %.0f  5pass_1080p %.0f  r1x2_1080p_u8 %.0f fps" % (o["boxblur_rt_r5x3_4k"]["value"], p5["value"], g["value"]))
