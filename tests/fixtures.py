"""Shared, VapourSynth-free test inputs.

`crop_rgb24()` is the reference suite's own source picture (reference
tests/conftest.py:72-77); the derived formats restate what zimg does for the
only conversions that are reproducible without zimg:
  RGBS  = v * f32(1/255)                       (exact)
  GRAY8 = limited-range BT.709 luma, f64, +0.5 floor   (exact)
  GRAY16 / GRAYS / GRAYH luma, YUV4xxP8/P16/PS  (exact since round 3: oracle/vs_host.py restates zimg's
                                               matrix and chroma resampler; the plane sums of the reference's
                                               planeaverage.json goldens match to the last unit)
"""
from __future__ import annotations

import json
from functools import lru_cache
from pathlib import Path

import numpy as np

GOLDEN_DIR = Path(__file__).resolve().parent / "golden"


@lru_cache(maxsize=None)
def crop_rgb24() -> np.ndarray:
    a = np.ascontiguousarray(_crop_rows322()[:, :320])
    a.setflags(write=False)
    return a


@lru_cache(maxsize=None)
def _crop_rows322() -> np.ndarray:
    a = np.load(GOLDEN_DIR / "crop_rgb24.npy")
    assert a.shape == (3, 322, 640) and a.dtype == np.uint8
    return a


def temporal_rgb24(n: int) -> np.ndarray:
    """Frame n (0..2) of the reference's temporal fixture (tests/conftest.py:138-148): the crop
    shifted down n rows."""
    assert 0 <= n <= 2
    return np.ascontiguousarray(_crop_rows322()[:, n:n + 320])


def luma8(rgb: np.ndarray) -> np.ndarray:
    """zimg RGB24 -> 8-bit limited-range BT.709 luma (exact): plane 0 of the reference's
    GRAY8 and YUV4xxP8 fixtures."""
    r, g, b = (rgb[i].astype(np.float64) for i in range(3))
    return np.floor((0.2126 * r + 0.7152 * g + 0.0722 * b) * 219.0 / 255.0 + 16.0 + 0.5).astype(np.uint8)


@lru_cache(maxsize=None)
def crop_rgbs() -> np.ndarray:
    a = crop_rgb24().astype(np.float32) * np.float32(1.0 / 255.0)
    a.setflags(write=False)
    return a


@lru_cache(maxsize=None)
def crop_gray8() -> np.ndarray:
    a = luma8(crop_rgb24())
    a.setflags(write=False)
    return a


@lru_cache(maxsize=None)
def crop_grays() -> np.ndarray:
    """zimg RGB24 -> GRAYS (matrix=1): the FMA-chain luma (oracle/vs_host.py); planeaverage.json
    `GRAYS|full|exclude=[-1]` reproduces to every digit."""
    from oracle import vs_host as vh

    y = vh.rgb24_to_yuv(crop_rgb24(), sample="f32", gray=True)[0]
    y.setflags(write=False)
    return y


@lru_cache(maxsize=None)
def crop_gray16() -> np.ndarray:
    """zimg RGB24 -> GRAY16 limited range; the plane sum equals planeaverage.json `GRAY16|full|exclude=[-1]`."""
    from oracle import vs_host as vh

    a = vh.rgb24_to_yuv(crop_rgb24(), 16, gray=True)[0]
    a.setflags(write=False)
    return a


@lru_cache(maxsize=None)
def crop_yuv(bits: int = 8, ssw: int = 1, ssh: int = 1, sample: str = "int", temporal: int = 0) -> tuple:
    """The reference's YUV fixtures (tests/conftest.py:88-102: resize.Bilinear(format=YUV..., matrix=1) of the
    RGB24 crop; `temporal` = frame n of the 3-frame shifted clip)."""
    from oracle import vs_host as vh

    planes = vh.rgb24_to_yuv(temporal_rgb24(temporal), bits, ssw, ssh, sample=sample)
    for p in planes:
        p.setflags(write=False)
    return tuple(planes)


def yuv_geometry(planes, geometry: str, ssw: int = 1, ssh: int = 1) -> list:
    """reference tests/conftest.py:108-122 on a subsampled clip: `odd` crops the subsampling modulus off the
    right / bottom, `tiny` is CropAbs(13 - 13 % wmod, 7 - 7 % hmod, left=200, top=100)."""
    if geometry == "full":
        return [np.ascontiguousarray(p) for p in planes]
    wm, hm = 1 << ssw, 1 << ssh
    y, u, v = planes
    if geometry == "odd":
        return [np.ascontiguousarray(y[:-hm, :-wm]), np.ascontiguousarray(u[:-1, :-1]), np.ascontiguousarray(v[:-1, :-1])]
    if geometry == "tiny":
        tw, th = 13 - 13 % wm, 7 - 7 % hm
        cw, ch = tw >> ssw, th >> ssh
        return [np.ascontiguousarray(y[100:100 + th, 200:200 + tw]), np.ascontiguousarray(u[100 >> ssh:(100 >> ssh) + ch, 200 >> ssw:(200 >> ssw) + cw]),
                np.ascontiguousarray(v[100 >> ssh:(100 >> ssh) + ch, 200 >> ssw:(200 >> ssw) + cw])]
    raise ValueError(geometry)


@lru_cache(maxsize=None)
def ref_goldens() -> dict:
    return json.loads((GOLDEN_DIR / "ref_goldens.json").read_text())


def plane_stats(p: np.ndarray) -> dict:
    """std.PlaneStats as the reference's golden_stats uses it (tests/golden.py:106-121):
    avg normalised by peak for integer formats, min/max raw."""
    if p.dtype.kind == "u":
        peak = float((1 << (8 * p.dtype.itemsize)) - 1)
        avg = float(p.astype(np.uint64).sum()) / p.size / peak
        return {"avg": avg, "min": int(p.min()), "max": int(p.max())}
    q = p.astype(np.float32)
    return {"avg": float(q.astype(np.float64).sum() / q.size), "min": float(q.min()), "max": float(q.max())}


def splitmix64_plane(seed: int, shape, dtype) -> np.ndarray:
    """Deterministic noise plane: splitmix64 stream -> full-range
    integers or uniform [0,1) floats."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        x = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    dtype = np.dtype(dtype)
    if dtype == np.uint8:
        out = (x >> np.uint64(56)).astype(np.uint8)
    elif dtype == np.uint16:
        out = (x >> np.uint64(48)).astype(np.uint16)
    elif dtype == np.float32:
        out = ((x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)
    elif dtype == np.float16:
        out = ((x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float16)
    else:
        raise TypeError(dtype)
    return out.reshape(shape)


def tiled_natural(shape, dtype, plane: int = 0) -> np.ndarray:
    """Natural-content plane of any size: the reference crop tiled."""
    h, w = shape
    dtype = np.dtype(dtype)
    base = crop_rgb24()[plane]
    reps = (-(-h // base.shape[0]), -(-w // base.shape[1]))
    t = np.tile(base, reps)[:h, :w]
    if dtype == np.uint8:
        return np.ascontiguousarray(t)
    if dtype == np.uint16:
        return (t.astype(np.uint16) * np.uint16(257)).astype(np.uint16)
    return (t.astype(np.float32) * np.float32(1.0 / 255.0)).astype(dtype)


def has_dev_variants(dev) -> bool:
    """does libvszip_hip.so contain the development variants (built with -DVSZIP_DEV_VARIANTS)?"""
    from vszip_amd.capi import VszipError

    try:
        dev.get_option("VSZIP_RT_FUSED")
        return True
    except VszipError as e:
        if e.code == -3:
            return False
        raise


def set_dev_option(dev, name: str, value: int):
    """set an option that only a -DVSZIP_DEV_VARIANTS build has; the default build skips the test"""
    import pytest
    from vszip_amd.capi import VszipError

    try:
        dev.set_option(name, value)
    except VszipError as e:
        if e.code == -3:
            pytest.skip(f"{name}: development variant, not in the default build")
        raise
