"""TEST INFRASTRUCTURE ONLY — numpy restatement of the HOST-side steps the reference's tests
and wrappers lean on but which live outside /root/reference: VapourSynth's `std.BoxBlur` and the
zimg conversions behind `resize.*` (depth conversion, gray -> RGB, sRGB -> linear light).

Third-party code, absent from the reference tree and from this image:
  * VapourSynth core (`pyproject.toml:13`: VapourSynth>=75) — `std.BoxBlur`, used by the
    reference's tests to build distorted / companion clips (tests/test_ssimulacra2.py:19-26,
    tests/test_adaptive_binarize.py:12-14, tests/test_planeaverage.py:71-73);
  * zimg (version unpinned, whatever the VapourSynth build carries) — `resize.Bicubic(format=RGBS,
    matrix_in=...)` in `hz.toRGBS` (src/helper.zig:225-243) and `resize.Bicubic(transfer=LINEAR)`
    in `sRGBtoLinearRGB` (src/vapoursynth/ssimulacra2.zig:132-162).
Neither source is available here, so every rule below is restated from the published behaviour
and PINNED BY THE REFERENCE'S OWN GOLDENS (tests/test_oracle_vs_host.py):
  * integer std.BoxBlur = horizontal then vertical running box, replicated edges,
    dst = (sum + 2r) / (2r+1)  [sic: the rounding term is 2r, not r] — reproduces all six
    `adaptive_binarize.json` GRAY8|full keys and the RGB24 ones to the pixel count, and the
    `planeaverage.json` YUV420P8 `|ref3` luma diff to every digit;
  * float std.BoxBlur = the same walk with an f32 running sum times f32(1/(2r+1)), replicated
    edges — `planeaverage.json` `RGBS|...|ref3`;
  * zimg integer -> float: (v - offset) * f32(1/range), full range for RGB, limited for Gray/YUV;
  * zimg sRGB -> linear with VapourSynth's default approximate_gamma=1: a 2^16+1-entry table over
    [-0.5, 1.5] indexed by rint(x * 32768 + 16384), entries from the sRGB EOTF with zimg's
    constants (alpha 1.055010718947587, beta 0.003041282560128) — with it the oracle meets
    `ssimulacra2.json` `RGBS|full|dist=blur1` to 7e-6 relative (the textbook EOTF: 8e-4),
    and the RGB24 / GRAY8 keys to < 1e-4.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

SRGB_ALPHA = 1.055010718947587
SRGB_BETA = 0.003041282560128


# ---------------------------------------------------------------------------------------
# std.BoxBlur (VapourSynth core)
# ---------------------------------------------------------------------------------------
def _rep_index(n: int, r: int) -> np.ndarray:
    return np.clip(np.arange(-r, n + r), 0, n - 1)


def _box_rows_int(a: np.ndarray, r: int) -> np.ndarray:
    h, w = a.shape
    p = a[:, _rep_index(w, r)].astype(np.int64)
    c = np.concatenate([np.zeros((h, 1), np.int64), np.cumsum(p, 1)], 1)
    s = c[:, 2 * r + 1:] - c[:, :-(2 * r + 1)]
    return ((s + 2 * r) // (2 * r + 1)).astype(a.dtype)


def _box_rows_float(a: np.ndarray, r: int) -> np.ndarray:
    """f32 running sum along each row: acc starts as r copies of the first sample plus the first r
    samples; per output: acc += entering, dst = acc * div, acc -= leaving (edges replicated)."""
    h, w = a.shape
    div = np.float32(1.0 / (2 * r + 1))
    src = a.astype(np.float32)
    acc = np.float32(r) * src[:, 0]
    for x in range(r):
        acc = acc + src[:, min(x, w - 1)]
    out = np.empty_like(src)
    for x in range(w):
        acc = acc + src[:, min(x + r, w - 1)]
        out[:, x] = acc * div
        acc = acc - src[:, max(x - r, 0)]
    return out


def std_boxblur(plane: np.ndarray, hradius: int = 1, vradius: int = 1) -> np.ndarray:
    """VapourSynth `std.BoxBlur(hradius, vradius)`, one pass per axis, on one plane."""
    fn = _box_rows_int if plane.dtype.kind == "u" else _box_rows_float
    out = np.ascontiguousarray(plane)
    if hradius > 0:
        out = fn(out, hradius)
    if vradius > 0:
        out = np.ascontiguousarray(fn(np.ascontiguousarray(out.T), vradius).T)
    return out


# ---------------------------------------------------------------------------------------
# zimg: depth conversion, gray -> RGB, sRGB -> linear
# ---------------------------------------------------------------------------------------
def int_to_float(plane: np.ndarray, bits: int, limited: bool, chroma: bool = False) -> np.ndarray:
    """zimg integer -> f32: (v - offset) * f32(1 / range)."""
    if limited:
        off = (128 if chroma else 16) << (bits - 8)
        rng = (224 if chroma else 219) << (bits - 8)
    else:
        off = (1 << (bits - 1)) if chroma else 0
        rng = (1 << bits) - 1
    return ((plane.astype(np.float32) - np.float32(off)) * np.float32(1.0 / rng)).astype(np.float32)


def srgb_eotf(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float64)
    lin = np.where(x < 12.92 * SRGB_BETA, x / 12.92, ((np.maximum(x, 0.0) + (SRGB_ALPHA - 1.0)) / SRGB_ALPHA) ** 2.4)
    return lin.astype(np.float32)


_LUT = None


def srgb_to_linear_lut() -> np.ndarray:
    """The 65537-entry table of zimg's approximate-gamma path: entry i = EOTF(i / 65536 * 2 - 0.5)."""
    global _LUT
    if _LUT is None:
        x = (np.arange(65537, dtype=np.float32) / np.float32(65536) * np.float32(2) - np.float32(0.5)).astype(np.float32)
        _LUT = srgb_eotf(x)
    return _LUT


def srgb_to_linear(x: np.ndarray) -> np.ndarray:
    idx = np.rint(x.astype(np.float32) * np.float32(32768) + np.float32(16384)).astype(np.int64)
    return srgb_to_linear_lut()[np.clip(idx, 0, 65536)]


def to_rgbs(planes, family: str, bits: int = 32) -> list:
    """`hz.toRGBS` for the inputs that need no resampler: RGB24/48 (full range), RGBS (as is),
    Gray8/16 (limited range; R = G = B = Y through any YUV -> RGB matrix since U = V = 0)."""
    if family == "RGBS":
        return [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
    if family == "RGB":
        return [int_to_float(p, bits, False) for p in planes]
    if family == "GRAY":
        y = planes[0].astype(np.float32) if planes[0].dtype.kind == "f" else int_to_float(planes[0], bits, True)
        return [y, y, y]
    raise ValueError(family)


def to_linear_rgbs(planes, family: str, bits: int = 32) -> list:
    """toRGBS followed by sRGBtoLinearRGB — what vszip.SSIMULACRA2 feeds its kernel."""
    return [srgb_to_linear(p) for p in to_rgbs(planes, family, bits)]
