"""GPU box: thresholded PlaneMinMax, 64 x 4K YUV420P16 a call (the bench leg's shape), frames/s on noise / the test picture / the picture x 257
(8-bit content in 16 bits), steady state (predicted) and with prediction off. `trace`: noise, predicted, a few calls (for tools/ktrace.sh)."""
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

dev = vszip_amd.Device(0)
out = []
trace = len(sys.argv) > 1 and sys.argv[1] == "trace"
for name in ("noise",) if trace else ("noise", "picture", "picture x 257"):
    if name == "noise":
        base = bench.make_frame(4, 3840, 2160)
    else:
        base = [fx.tiled_natural(s, np.uint16, p) for p, s in enumerate(bench.yuv420_shapes(3840, 2160))]
        if name.endswith("257"):
            base = [((p >> 8).astype(np.uint16) * 257).astype(np.uint16) for p in base]
    srcs = [dev.upload(np.roll(p, f, axis=1)) for f in range(64) for p in base]
    for mode in (0,) if trace else (0, 1):
        with dev.options(VSZIP_MINMAX_NO_PREDICT=mode):
            run = dev.prepared_plane_minmax(srcs, 0.1, 0.1)
            run()
            run()
            t0 = time.perf_counter()
            for _ in range(10):
                r = run()
            dt = time.perf_counter() - t0
        out.append(f"{name} {'two sweeps' if mode else 'predicted'}: {640 / dt / 1e3:.1f} k")
    del srcs
print(" | ".join(out))
