"""Development: XPSNR leg of bench.py alone (batched + per-frame calls)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: F401  (before the library: see INTEGRATION.md)

import bench
import vszip_amd

dev = vszip_amd.Device(0)
timed = bench.Timed(dev, lambda: None)
print(json.dumps(bench.xpsnr_leg(dev, timed, True)))
dev.close()
