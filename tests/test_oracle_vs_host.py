"""Pins for oracle/vs_host.py (the VapourSynth-core / zimg steps that sit outside the reference
repo) and, through them, the SSIMULACRA2 and AdaptiveBinarize oracles: every key of the
reference's goldens that is reachable once `std.BoxBlur` and zimg's depth / gray / sRGB->linear
conversions are restated (tests/goldens/{ssimulacra2,adaptive_binarize,planeaverage}.json,
cases in tests/test_ssimulacra2.py:36-62, tests/test_adaptive_binarize.py:17-50,
tests/test_planeaverage.py:60-80)."""
import numpy as np
import pytest

import fixtures as fx
from oracle import vs_host as vh


def _geometry(p: np.ndarray, geometry: str) -> np.ndarray:
    """reference tests/conftest.py:108-122 for non-subsampled formats."""
    if geometry == "full":
        return np.ascontiguousarray(p)
    if geometry == "odd":
        return np.ascontiguousarray(p[:-1, :-1])
    if geometry == "tiny":
        return np.ascontiguousarray(p[100:107, 200:213])
    raise ValueError(geometry)


# ---- std.BoxBlur ----------------------------------------------------------------------------
def test_std_boxblur_u8_pinned_by_planeaverage_luma():
    """planeaverage.json YUV420P8|...|ref3, plane 0 (= the GRAY8 fixture): psmDiff of the clip
    against std.BoxBlur(3,3) of itself, to every digit."""
    g = fx.ref_goldens()["luma_of_yuv"]["planeaverage"]["YUV420P8|full|exclude=[-1],planes=[0,1,2]|ref3#luma"]
    y = fx.crop_gray8()
    assert float(y.astype(np.uint64).sum()) / y.size / 255.0 == g["avg"]
    b = vh.std_boxblur(y, 3, 3)
    diff = float(np.abs(y.astype(np.int64) - b.astype(np.int64)).sum()) / y.size / 255.0
    assert diff == pytest.approx(g["diff"], rel=1e-14)


def test_std_boxblur_f32_pinned_by_planeaverage_rgbs(oracle):
    """planeaverage.json RGBS|...|ref3: f32 running sum, replicated edges. The blue plane is nearly
    constant (diff 5.9e-5), so the running sum's drift shows: an exact window sum misses it by 2e-3."""
    g = fx.ref_goldens()["exact"]["planeaverage"]["RGBS|full|exclude=[-1],planes=[0,1,2]|ref3"]
    for i, p in enumerate(fx.crop_rgbs()):
        p = np.ascontiguousarray(p)
        avg, diff = oracle.plane_average(p, [-1], ref=vh.std_boxblur(p, 3, 3))
        assert avg == pytest.approx(g["avg"][i], rel=1e-6)
        assert diff == pytest.approx(g["diff"][i], rel=1e-6)


# ---- AdaptiveBinarize: every GRAY8 / RGB24 golden (clip2 = std.BoxBlur(5) or (12)) ----------------
def _ab_keys():
    g = fx.ref_goldens()["exact"]["adaptive_binarize"]
    return sorted(g)


@pytest.mark.parametrize("key", _ab_keys())
def test_adaptive_binarize_goldens(oracle, key):
    g = fx.ref_goldens()["exact"]["adaptive_binarize"][key]
    parts = key.split("|")
    fmt, geometry, c = parts[0], parts[1], int(parts[2].split("=")[1])
    r = 12 if parts[-1] == "wide" else 5
    planes = [fx.crop_gray8()] if fmt == "GRAY8" else list(fx.crop_rgb24())
    assert len(g) == len(planes)
    for i, p in enumerate(planes):
        src = _geometry(p, geometry)
        out = oracle.adaptive_binarize(src, vh.std_boxblur(src, r, r), c)
        st = fx.plane_stats(out)
        e = g[f"p{i}"]
        assert st["avg"] == pytest.approx(e["avg"], rel=1e-12, abs=1e-15) and st["min"] == e["min"] and st["max"] == e["max"], (key, i, st, e)


# ---- SSIMULACRA2: all five reachable keys, far inside the reference's own rel=1e-3 ---------------
@pytest.mark.parametrize("key,family,radius", [
    ("RGBS|full|dist=blur1", "RGBS", 1),
    ("RGB24|full|dist=blur1", "RGB", 1),
    ("RGB24|full|dist=blur3", "RGB", 3),
    ("GRAY8|full|dist=blur1", "GRAY", 1),
    ("GRAY8|full|dist=blur3", "GRAY", 3),
])
def test_ssimulacra2_goldens(oracle, key, family, radius):
    """reference tests/test_ssimulacra2.py:56-61 (rel=1e-3 there). ref -> toRGBS -> linear, dist =
    std.BoxBlur(ref) in the clip's own format -> the same; measured agreement 3e-6 .. 7e-5."""
    g = fx.ref_goldens()["exact"]["ssimulacra2"][key]
    if family == "RGBS":
        ref = [np.ascontiguousarray(p) for p in fx.crop_rgbs()]
    elif family == "RGB":
        ref = [np.ascontiguousarray(p) for p in fx.crop_rgb24()]
    else:
        ref = [fx.crop_gray8()]
    dis = [vh.std_boxblur(p, radius, radius) for p in ref]
    s = oracle.ssimulacra2(vh.to_linear_rgbs(ref, family, 8), vh.to_linear_rgbs(dis, family, 8))
    assert s == pytest.approx(g, rel=2e-4), (key, s, g, s / g - 1)


def test_ssimulacra2_soft_gray16(oracle):
    """GRAY16|full|dist=blur1 on the approximate GRAY16 fixture: still inside rel=1e-3."""
    g = fx.ref_goldens()["soft"]["ssimulacra2"]["GRAY16|full|dist=blur1"]
    y = np.ascontiguousarray(fx.crop_gray16())
    s = oracle.ssimulacra2(vh.to_linear_rgbs([y], "GRAY", 16), vh.to_linear_rgbs([vh.std_boxblur(y, 1, 1)], "GRAY", 16))
    assert s == pytest.approx(g, rel=1e-3)


def test_ssimulacra2_ordering(oracle):
    """reference tests/test_ssimulacra2.py:78-83: identical > blur1 > blur3."""
    ref = [np.ascontiguousarray(p) for p in fx.crop_rgb24()]
    lin = vh.to_linear_rgbs(ref, "RGB", 8)
    s1 = oracle.ssimulacra2(lin, vh.to_linear_rgbs([vh.std_boxblur(p, 1, 1) for p in ref], "RGB", 8))
    s3 = oracle.ssimulacra2(lin, vh.to_linear_rgbs([vh.std_boxblur(p, 3, 3) for p in ref], "RGB", 8))
    assert oracle.ssimulacra2(lin, lin) > s1 > s3


def test_srgb_lut_shape():
    lut = vh.srgb_to_linear_lut()
    assert lut.shape == (65537,) and lut.dtype == np.float32
    assert lut[16384] == 0.0 and lut[49152] == pytest.approx(1.0, abs=1e-6)  # x = 0 and x = 1
    assert np.all(np.diff(lut[16384:]) >= 0)
