#!/usr/bin/env python3
"""Regenerate tests/golden/* from the reference's own test DATA (not source).

Run in the build container only (it reads /root/reference, which does not
exist on the GPU box):   python tests/golden/make_fixtures.py

Produces
  crop_rgb24.npy      the 640x320 RGB24 source every reference golden is taken
                      on: tests/image.png cropped to x in [1280,1920), y in
                      [0,320)  (reference tests/conftest.py:72-77, Crop(left=w-640,
                      bottom=h-320)), plus rows 320 and 321: frame n of the
                      reference's 3-frame temporal fixture is rows [n, n+320)
                      (tests/conftest.py:138-148).  uint8 [3][322][640] planar.
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
# round 3: oracle/vs_host.py restates zimg's matrix + chroma resampler exactly enough (plane sums of
# planeaverage.json match to the last unit), so the YUV keys and the 16-bit / float Gray keys are reachable too
YUV_PREFIXES = ("YUV420P8|", "YUV420P16|", "YUV444P16|", "YUV444PS|", "YUV420PS|", "YUV420P10|", "GRAYH|")
FILES = ["boxblur", "bilateral", "eedi3", "eedi3h", "planeaverage",
         "planeminmax", "ssimulacra2", "limiter", "limitfilter", "adaptive_binarize"]
# XPSNR accepts YUV only, but XPSNR_Y depends on the luma planes alone and the luma of the
# reference's YUV fixtures is reproducible (limited-range BT.709 from the RGB24 crop): the Y entries
# of its goldens are reachable for the 8-bit keys (10-bit: approximate luma, soft).


def main() -> int:
    if not REF.is_dir():
        print("reference tree not present; fixtures are already committed", file=sys.stderr)
        return 1
    img = np.asarray(Image.open(REF / "image.png").convert("RGB"))
    h, w, _ = img.shape
    assert (w, h) == (1920, 1080), (w, h)
    crop = img[0:322, w - 640:w, :]  # 320 rows + the two extra rows of the 3-frame temporal fixture
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
    yuv = {}
    for name in FILES + ["xpsnr"]:
        data = json.loads((REF / "goldens" / f"{name}.json").read_text())
        yuv[name] = {k: v for k, v in data.items() if k.startswith(YUV_PREFIXES)}
    # plane 0 of a YUV420P8 clip is the GRAY8 fixture (same limited-range BT.709 luma; the golden's
    # avg[0] equals the GRAY8 average to every digit), so the luma entries of YUV keys are reachable
    pa = json.loads((REF / "goldens" / "planeaverage.json").read_text())
    k = "YUV420P8|full|exclude=[-1],planes=[0,1,2]|ref3"
    extra = {"planeaverage": {k + "#luma": {"avg": pa[k]["avg"][0], "diff": pa[k]["diff"][0]}}}
    xp = json.loads((REF / "goldens" / "xpsnr.json").read_text())
    extra["xpsnr_Y"] = {k: v["Y"] for k, v in xp.items() if k.startswith(("YUV420P8|", "YUV420P10|"))}
    # x+k distortions: chroma SSE is k^2 per sample whatever the chroma content -> U/V reachable as well
    extra["xpsnr_UV"] = {k: {"U": v["U"], "V": v["V"]} for k, v in xp.items() if k.startswith("YUV420P8|") and ("|bright|" in k or "|shift|" in k)}
    (OUT / "ref_goldens.json").write_text(json.dumps({"exact": out, "soft": soft, "luma_of_yuv": extra, "yuv": yuv}, indent=1, sort_keys=True) + "\n")
    print("wrote", OUT / "crop_rgb24.npy", planar.shape, "and ref_goldens.json")
    return 0


if __name__ == "__main__":
    sys.exit(main())
