#!/usr/bin/env python3
"""Regenerate tests/golden/* from the reference's own test DATA (not source).

Run in the build container only (it reads /root/reference, which does not
exist on the GPU box):   python tests/golden/make_fixtures.py

Produces
  crop_rgb24.npy      the 640x320 RGB24 source every reference golden is taken
                      on: tests/image.png cropped to x in [1280,1920), y in
                      [0,320)  (reference tests/conftest.py:72-77, Crop(left=w-640,
                      bottom=h-320)).  uint8 [3][320][640] planar.
  ref_goldens.json    the subset of the reference's tests/goldens/*.json whose
                      inputs are reproducible without VapourSynth/zimg
                      (RGB24 untouched, RGBS = v * f32(1/255), GRAY8 = limited
                      range BT.709 luma).
Only data (pixels, expected numbers) is copied; no reference source text.
"""
import json
import sys
from pathlib import Path

import numpy as np
from PIL import Image

REF = Path("/root/reference/tests")
OUT = Path(__file__).resolve().parent

REACHABLE_PREFIXES = ("RGB24|", "RGBS|", "GRAY8|")
FILES = ["boxblur", "bilateral", "eedi3", "eedi3h", "planeaverage",
         "planeminmax", "ssimulacra2", "limiter", "limitfilter"]


def main() -> int:
    if not REF.is_dir():
        print("reference tree not present; fixtures are already committed", file=sys.stderr)
        return 1
    img = np.asarray(Image.open(REF / "image.png").convert("RGB"))
    h, w, _ = img.shape
    assert (w, h) == (1920, 1080), (w, h)
    crop = img[0:320, w - 640:w, :]
    planar = np.ascontiguousarray(crop.transpose(2, 0, 1)).astype(np.uint8)
    np.save(OUT / "crop_rgb24.npy", planar)

    out = {}
    for name in FILES:
        data = json.loads((REF / "goldens" / f"{name}.json").read_text())
        out[name] = {k: v for k, v in data.items() if k.startswith(REACHABLE_PREFIXES)}
    # soft-pinned (approximate GRAY16 / GRAYS luma fixtures): kept for
    # magnitude checks at a looser tolerance
    soft = {}
    for name in FILES:
        data = json.loads((REF / "goldens" / f"{name}.json").read_text())
        soft[name] = {k: v for k, v in data.items() if k.startswith(("GRAY16|", "GRAYS|"))}
    (OUT / "ref_goldens.json").write_text(json.dumps({"exact": out, "soft": soft}, indent=1, sort_keys=True) + "\n")
    print("wrote", OUT / "crop_rgb24.npy", planar.shape, "and ref_goldens.json")
    return 0


if __name__ == "__main__":
    sys.exit(main())
