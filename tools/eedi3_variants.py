"""Development: EEDI3 throughput of the option sets that take the general line kernel (hp, mdis > 31,
mclip) next to the default, 1920x1080 YUV420PS dh, 4 frames per call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401

import fixtures as fx
import vszip_amd

dev = vszip_amd.Device(0)
frames = 4
planes = []
for f in range(frames):
    for p, s in enumerate([(1080, 1920), (540, 960), (540, 960)]):
        planes.append(dev.upload(np.roll(fx.tiled_natural(s, np.float32, p), 3 * f, axis=1)))
for name, kw in (("default", {}), ("hp", dict(hp=True)), ("mdis40", dict(mdis=40)), ("mdis31", dict(mdis=31)), ("vcheck0", dict(vcheck=0))):
    dev.eedi3(planes, 1, dh=True, **kw)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        dev.eedi3(planes, 1, dh=True, **kw)
    dev.sync()
    print(f"{name:8s} {3 * frames / (time.perf_counter() - t0):8.1f} frames/s", flush=True)
dev.close()
