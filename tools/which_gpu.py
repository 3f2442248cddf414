#!/usr/bin/env python3
"""GPU box: which physical GPU is this, and how do its candidate placements spread? (one line)"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch

import bench
import vszip_amd

dev = vszip_amd.Device(0)
base = bench.make_frame(0, bench.W4K, bench.H4K)
planes = [np.roll(p, f * 17 + 1, axis=1) for f in range(64) for p in base]
_, keep, info = bench.placed_batch(dev, planes, np.uint16, (13, 1, 13, 1), 0, 32, (12, 1, 12, 1))
c = np.array(info["destination_candidates_us"])
props = torch.cuda.get_device_properties(0)
bus = getattr(props, "pci_bus_id", None)
print("visible:", {k: v for k, v in os.environ.items() if "VISIBLE" in k}, "pci bus", bus, "uuid", getattr(props, "uuid", None),
      "| candidates min %.0f p25 %.0f median %.0f max %.0f, under 600 us: %d of %d" % (c.min(), np.percentile(c, 25), np.median(c), c.max(), int((c < 600).sum()), len(c)))
