"""The reference's own structural tests (tests/test_*.py of the reference) that no other module of this suite restates,
ported one to one onto the plugin boundary: `libvszip.so` loaded by the VapourSynth-free test host, clips built from
numpy planes. Same test names, same arguments, same assertions; where the reference leans on another VapourSynth
plugin (std.BoxBlur, std.Crop, resize) the equivalent numpy construction is written out. None of these needs the
oracle: they are properties of the filters themselves."""
import math
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def vs():
    from fakevs import fakevs

    return fakevs


def _gray(vs, fmt, crop=None):
    """the reference's `to_gray(fmt)`: its test image as one Gray plane of the format (here: the committed crop)."""
    g8 = np.ascontiguousarray(fx.crop_gray8() if crop is None else fx.crop_gray8()[crop])
    if fmt == vs.GRAY8:
        return g8
    if fmt == vs.GRAY16:
        return g8.astype(np.uint16) * 257
    return (g8.astype(np.float32) / np.float32(255.0)).astype(np.float16 if fmt == vs.GRAYH else np.float32)


def _clip(vs, fmt, planes, **kw):
    return vs.source([[np.ascontiguousarray(p) for p in planes]], fmt, **kw)


def _yuv(vs):
    """the reference's `to_yuv(YUV420P16)`: three planes with real content, chroma subsampled"""
    y = _gray(vs, vs.GRAY16)
    h, w = (y.shape[0] // 2) * 2, (y.shape[1] // 2) * 2
    y = np.ascontiguousarray(y[:h, :w])
    u = np.ascontiguousarray(np.roll(y, 5, axis=1)[::2, ::2])
    v = np.ascontiguousarray(np.roll(y, 11, axis=0)[::2, ::2])
    return [y, u, v]


# ---- BoxBlur: reference tests/test_boxblur.py -------------------------------------------------------------------------


@pytest.mark.parametrize("fmt_name", ["GRAY8", "GRAY16", "GRAYS"])
@pytest.mark.parametrize("radius", [1, 8, 22, 23, 40])  # 1..22 compile-time path, 23+ runtime path
def test_matches_std_boxblur(vs, fmt_name, radius):
    """:73-84 — interior pixels match a plain box mean (std.BoxBlur there; here the exact (2r+1)^2 mean in f64), within
    the reference's own tolerances for the fixed-point reciprocal: 2 / 16 / 1e-5. Independent of the oracle."""
    fmt = getattr(vs, fmt_name)
    src = _gray(vs, fmt)
    out = _clip(vs, fmt, [src]).vszip.BoxBlur(hradius=radius, vradius=radius).get_frame(0)[0]
    k = 2 * radius + 1
    c = np.cumsum(np.cumsum(np.pad(src.astype(np.float64), ((1, 0), (1, 0))), axis=0), axis=1)
    mean = (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]) / (k * k)  # windows centred on [r, h-r) x [r, w-r)
    m = radius + 2
    ours = out.astype(np.float64)[m:-m, m:-m]
    want = mean[m - radius:mean.shape[0] - (m - radius), m - radius:mean.shape[1] - (m - radius)]
    assert ours.shape == want.shape and ours.size > 0
    tol = {"GRAY8": 2, "GRAY16": 16, "GRAYS": 1e-5}[fmt_name]
    assert np.abs(ours - want).max() <= tol


def test_h_and_v_compose(vs):
    """:96-101 — hradius + vradius in one call equals separate h-only and v-only calls."""
    src = _clip(vs, vs.GRAY16, [_gray(vs, vs.GRAY16)])
    both = src.vszip.BoxBlur(hradius=4, vradius=9).get_frame(0)[0]
    split = src.vszip.BoxBlur(hradius=4, vradius=0, vpasses=0).vszip.BoxBlur(hradius=0, hpasses=0, vradius=9).get_frame(0)[0]
    assert np.array_equal(both, split)


def test_f16_runs(vs):
    """:104-108"""
    out = _clip(vs, vs.GRAYH, [_gray(vs, vs.GRAYH)]).vszip.BoxBlur(hradius=5, vradius=5)
    assert out.format_id == vs.GRAYH
    assert 0.0 < float(out.get_frame(0)[0].astype(np.float32).mean()) < 1.0


def test_planes(vs):
    """:111-120 — untouched planes are copied; the processed plane equals the blur of that plane alone."""
    planes = _yuv(vs)
    src = _clip(vs, vs.YUV420P16, planes)
    out = src.vszip.BoxBlur(planes=[0], hradius=5, vradius=5).get_frame(0)
    assert np.array_equal(out[1], planes[1]) and np.array_equal(out[2], planes[2])
    assert not np.array_equal(out[0], planes[0])
    y_blur = _clip(vs, vs.GRAY16, [planes[0]]).vszip.BoxBlur(hradius=5, vradius=5).get_frame(0)[0]
    assert np.array_equal(out[0], y_blur)


@pytest.mark.parametrize("fmt_name", ["GRAY8", "GRAY16", "GRAYS"])
@pytest.mark.parametrize("radius", [10, 30])  # compile-time and runtime paths
def test_stride_handling(vs, fmt_name, radius):
    """:123-128 — a clip cropped by 27 columns (odd width, offset plane pointer, stride > width) against its repacked
    copy."""
    fmt = getattr(vs, fmt_name)
    cropped = np.ascontiguousarray(_gray(vs, fmt)[:, 27:])
    isz = cropped.dtype.itemsize
    a = _clip(vs, fmt, [cropped], extra_stride=27 * isz + 64, offset=27 * isz).vszip.BoxBlur(hradius=radius, vradius=radius).get_frame(0)[0]
    b = _clip(vs, fmt, [cropped]).vszip.BoxBlur(hradius=radius, vradius=radius).get_frame(0)[0]
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


# ---- Bilateral: reference tests/test_bilateral.py ---------------------------------------------------------------------


def test_bilateral_planes(vs):
    """:71-81 — every plane is processed by default (unlike the wiki's claim); planes=[0] copies the others."""
    planes = _yuv(vs)
    src = _clip(vs, vs.YUV420P16, planes)
    out = src.vszip.Bilateral(sigmaS=2, sigmaR=2).get_frame(0)
    for p in range(3):
        assert not np.array_equal(out[p], planes[p])
    luma_only = src.vszip.Bilateral(sigmaS=2, sigmaR=2, planes=[0]).get_frame(0)
    assert not np.array_equal(luma_only[0], planes[0])
    assert np.array_equal(luma_only[1], planes[1]) and np.array_equal(luma_only[2], planes[2])


def test_bilateral_f16_runs(vs):
    """:90-93"""
    out = _clip(vs, vs.GRAYH, [_gray(vs, vs.GRAYH)]).vszip.Bilateral(sigmaS=2, sigmaR=2)
    assert out.format_id == vs.GRAYH
    assert 0.0 < float(out.get_frame(0)[0].astype(np.float32).mean()) < 1.0


def test_bilateral_stride_handling(vs):
    """:96-99"""
    cropped = np.ascontiguousarray(_gray(vs, vs.GRAY16)[:, 27:])
    a = _clip(vs, vs.GRAY16, [cropped], extra_stride=27 * 2 + 64, offset=27 * 2).vszip.Bilateral(sigmaS=2, sigmaR=2).get_frame(0)[0]
    b = _clip(vs, vs.GRAY16, [cropped]).vszip.Bilateral(sigmaS=2, sigmaR=2).get_frame(0)[0]
    assert np.array_equal(a, b)


def test_algorithm1_does_not_crash():
    """:117-131 — in a subprocess there because the failure mode was process death; kept that way."""
    script = (
        f"import sys; sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(ROOT / 'tests')!r})\n"
        "from fakevs import fakevs as vs\n"
        "src = vs.blank(vs.GRAY16, 64, 64, 0)\n"
        "src.vszip.Bilateral(sigmaS=3, sigmaR=0.1, algorithm=1).get_frame(0)\n"
        "print('OK')\n"
    )
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize(("w", "h"), [(20, 4), (5, 20), (4, 4), (3, 30)])
@pytest.mark.parametrize("fmt_name", ["GRAY8", "GRAY16", "GRAYS"])
def test_small_frame_errors(vs, fmt_name, w, h):
    """:133-143 — a plane smaller than 2 * radius on either axis (radius 5 at the default sigmaS = 3) is rejected at
    creation."""
    fmt = getattr(vs, fmt_name)
    src = vs.blank(fmt, w, h, 0.5 if fmt == vs.GRAYS else 100)
    with pytest.raises(vs.Error, match="plane too small for the spatial radius"):
        src.vszip.Bilateral()


def test_small_frame_subsampled_chroma_errors(vs):
    """:146-153 — the check is per processed plane: the 32 x 32 chroma plane with sigmaS = 20 fails, luma would not."""
    src = vs.blank(vs.YUV420P8, 64, 64, [100, 128, 128])
    with pytest.raises(vs.Error, match="plane too small for the spatial radius"):
        src.vszip.Bilateral(sigmaS=[2, 20], algorithm=2)


@pytest.mark.parametrize(("w", "h"), [(5, 5), (4, 30), (8, 8)])
def test_small_frame_algorithm1_ok(vs, w, h):
    """:156-164 — algorithm 1 is size-agnostic: small frames produce output."""
    out = vs.blank(vs.GRAY16, w, h, 100).vszip.Bilateral(sigmaS=3, sigmaR=0.1, algorithm=1)
    assert (out.width, out.height) == (w, h)
    out.get_frame(0)


# ---- SSIMULACRA2: reference tests/test_ssimulacra2.py -----------------------------------------------------------------


def test_identical_real_image(vs):
    """:65-71 — a constant clip against itself scores exactly 100, a real image more than 99.9."""
    ref = [np.ascontiguousarray(p) for p in fx.crop_rgbs()]
    a = vs.source([ref], vs.RGBS, props={"_Transfer": 8})
    assert a.vszip.SSIMULACRA2(a).get_frame(0).props["SSIMULACRA2"] > 99.9
    c = vs.source([[np.full((64, 64), v, np.float32) for v in (0.3, 0.2, 0.5)]], vs.RGBS, props={"_Transfer": 8})
    assert c.vszip.SSIMULACRA2(c).get_frame(0).props["SSIMULACRA2"] == 100.0


def test_dimension_and_length_errors(vs):
    """:103-116"""
    a = vs.blank(vs.RGBS, 64, 64, [0.1, 0.2, 0.3])
    with pytest.raises(vs.Error, match="clips must have the same dimensions"):
        a.vszip.SSIMULACRA2(vs.blank(vs.RGBS, 64, 48, [0.1, 0.2, 0.3]))
    with pytest.raises(vs.Error, match="clips must have the same length"):
        vs.blank(vs.RGBS, 64, 64, [0.1, 0.2, 0.3], length=3).vszip.SSIMULACRA2(vs.blank(vs.RGBS, 64, 64, [0.1, 0.2, 0.3], length=2))


# ---- XPSNR: reference tests/test_xpsnr.py -----------------------------------------------------------------------------


def _pair(vs, n=4, w=320, h=180):
    rng = np.random.default_rng(5)
    shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    frames = [[np.roll(fx.tiled_natural(s, np.uint8, p), 5 * f, axis=1) for p, s in enumerate(shapes)] for f in range(n)]
    dist = [[np.clip(p.astype(np.int16) + rng.integers(-6, 7, p.shape), 0, 255).astype(np.uint8) for p in fr] for fr in frames]
    return frames, dist, vs.source(frames, vs.YUV420P8), vs.source(dist, vs.YUV420P8)


def test_temporal_differs_from_spatial(vs):
    """:247-251 — (from frame 1 on: frame 0 has no previous frame)"""
    _, _, a, b = _pair(vs)
    t = [a.vszip.XPSNR(b, temporal=1, verbose=0).get_frame(n).props["XPSNR_Y"] for n in range(4)]
    s = [a.vszip.XPSNR(b, temporal=0, verbose=0).get_frame(n).props["XPSNR_Y"] for n in range(4)]
    assert all(math.isfinite(x) for x in t + s)
    assert all(t[n] != s[n] for n in range(1, 4))


def test_output_frame_is_distorted_copy(vs):
    """:254-256"""
    _, dist, a, b = _pair(vs)
    out = a.vszip.XPSNR(b, verbose=0)
    for n in range(4):
        f = out.get_frame(n)
        for p in range(3):
            assert np.array_equal(f[p], dist[n][p])


def test_no_teardown_corruption():
    """:335-349 — freeing an XPSNR node; in a subprocess as there."""
    script = (
        f"import sys; sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(ROOT / 'tests')!r})\n"
        "from fakevs import fakevs as vs\n"
        "a = vs.blank(vs.YUV420P8, 64, 64, [0, 128, 128])\n"
        "c = a.vszip.XPSNR(a, verbose=0)\n"
        "c.get_frame(0)\n"
        "del c, a\n"
        "print('OK')\n"
    )
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


# ---- PlaneMinMax / PlaneAverage: reference tests/test_planeminmax.py, tests/test_planeaverage.py ----------------------


def test_threshold_drop_semantics(vs):
    """:99-110 — both halves: minthr / maxthr drop that fraction of extreme pixels before picking."""
    p = np.full((32, 64), 200, np.uint8)
    p[:8] = 0  # 25 % of the pixels
    src = _clip(vs, vs.GRAY8, [p])
    assert src.vszip.PlaneMinMax(minthr=0.2).get_frame(0).props["psmMin"] == 0
    assert src.vszip.PlaneMinMax(minthr=0.3).get_frame(0).props["psmMin"] == 200
    q = np.full((32, 64), 100, np.uint8)
    q[24:] = 255
    src2 = _clip(vs, vs.GRAY8, [q])
    assert src2.vszip.PlaneMinMax(maxthr=0.2).get_frame(0).props["psmMax"] == 255
    assert src2.vszip.PlaneMinMax(maxthr=0.3).get_frame(0).props["psmMax"] == 100


def test_diff_ignores_thresholds(vs):
    """:139-144 — psmDiff is computed on all pixels; the thresholds only move min / max."""
    src = _clip(vs, vs.GRAY16, [_gray(vs, vs.GRAY16)])
    blur = src.vszip.BoxBlur(hradius=1, vradius=1)
    d0 = src.vszip.PlaneMinMax(minthr=0, maxthr=0, clipb=blur).get_frame(0).props["psmDiff"]
    d1 = src.vszip.PlaneMinMax(minthr=0.2, maxthr=0.3, clipb=blur).get_frame(0).props["psmDiff"]
    assert d0 == d1 and d0 > 0


@pytest.mark.parametrize("fmt_name", ["GRAY16", "GRAYS"])
def test_thr_one_no_counter_overflow(vs, fmt_name):
    """:226-240 — minthr / maxthr = 1.0 on a clip with a 65536-bin histogram: dropping 100 % yields peak as min, 0 as max."""
    fmt = getattr(vs, fmt_name)
    src = vs.blank(fmt, 64, 64, 0.5 if fmt == vs.GRAYS else 30000)
    pmin = src.vszip.PlaneMinMax(minthr=1.0).get_frame(0).props["psmMin"]
    pmax = src.vszip.PlaneMinMax(maxthr=1.0).get_frame(0).props["psmMax"]
    if fmt == vs.GRAY16:
        assert pmin == 65535 and pmax == 0
    else:
        assert math.isfinite(pmin) and math.isfinite(pmax)  # (the reference asserts values for 16-bit only)


def test_float_no_thr_exact_minmax(vs):
    """:124-128 — without thresholds a float clip reports its exact extremes, values outside [0, 1] included (no
    histogram quantisation, no clamping)."""
    g = _gray(vs, vs.GRAYS).copy()
    g[3, 5] = np.float32(-2.0)
    g[7, 9] = np.float32(1.987654321)
    p = _clip(vs, vs.GRAYS, [g]).vszip.PlaneMinMax().get_frame(0).props
    assert p["psmMin"] == float(g.min()) and p["psmMax"] == float(g.max())
