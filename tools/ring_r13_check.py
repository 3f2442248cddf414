"""GPU box: BoxBlur r = 13 on u16 planes, the library in place against the CPU oracle on a few geometries, then the 64 x 4K launch timed (us per launch, events)."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import bench
import fixtures as fx
import vszip_amd
from oracle import oracle as orc

orc.build()
dev = vszip_amd.Device(0)
bad = 0
for i, shape in enumerate([(270, 960), (135, 1920), (540, 3840), (100, 976), (61, 480), (300, 1440), (77, 2880)]):
    a = fx.splitmix64_plane(50 + i, shape, np.uint16)
    d = dev.upload(a)
    o = dev.empty(shape[0], shape[1], np.uint16)
    dev.boxblur([d], [o], 13, 1, 13, 1)
    got = dev.download(o)
    want = orc.boxblur(a, 13, 1, 13, 1)
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH", shape, int((got != want).sum()), np.argwhere(got != want)[:3].tolist())
print("parity:", "ok" if not bad else f"{bad} geometries differ")
timed = bench.Timed(dev, dev.sync)
step, keep = bench.setup_boxblur(dev, 0, 64, 13)
dt, region_ms, dom_ms, launches = timed.run(step, 300, 5)
print(f"launch {dom_ms * 1e3 / launches:.1f} us  frac {2 * 24883200 * 64 / (dom_ms * 1e-3 / launches) / 8e12:.4f}  arena {keep[2]['arena']}")
