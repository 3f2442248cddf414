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


def _two_tone(vs, fmt, lo, hi):
    dt = np.uint16 if fmt == vs.GRAY16 else np.float32
    return _clip(vs, fmt, [np.concatenate([np.full((32, 64), lo, dt), np.full((32, 64), hi, dt)], axis=0)])


def test_matches_std_planestats(vs):
    """test_planeaverage.py:108-113 — with a never-matching exclude value this is the plain plane mean (exact: the sum of
    16-bit samples is an integer)."""
    g = _gray(vs, vs.GRAY16)
    ours = _clip(vs, vs.GRAY16, [g]).vszip.PlaneAverage(exclude=[-1]).get_frame(0).props["psmAvg"]
    assert ours == pytest.approx(float(g.astype(np.float64).mean()) / 65535.0, rel=1e-12)


def test_exclude_exact(vs):
    """test_planeaverage.py:116-121"""
    src = _two_tone(vs, vs.GRAY16, 1000, 3000)
    assert src.vszip.PlaneAverage(exclude=[1000]).get_frame(0).props["psmAvg"] == 3000 / 65535
    assert src.vszip.PlaneAverage(exclude=[3000]).get_frame(0).props["psmAvg"] == 1000 / 65535
    assert src.vszip.PlaneAverage(exclude=[1000, 3000]).get_frame(0).props["psmAvg"] == 0.0


def test_exclude_float_clip(vs):
    """test_planeaverage.py:124-126"""
    assert _two_tone(vs, vs.GRAYS, 3.0, 1.0).vszip.PlaneAverage(exclude=[3]).get_frame(0).props["psmAvg"] == 1.0


def test_clipb_diff_matches_std(vs):
    """test_planeaverage.py:129-133 — psmDiff is the mean absolute difference over the peak."""
    g = _gray(vs, vs.GRAY16)
    b = np.ascontiguousarray(np.roll(g, 3, axis=1))
    p = _clip(vs, vs.GRAY16, [g]).vszip.PlaneAverage(exclude=[-1], clipb=_clip(vs, vs.GRAY16, [b])).get_frame(0).props
    assert p["psmDiff"] == pytest.approx(float(np.abs(g.astype(np.float64) - b).mean()) / 65535.0, rel=1e-12)


def test_planes_and_prop_rename(vs):
    """test_planeaverage.py:136-146"""
    src = vs.blank(vs.YUV420P16, 64, 32, [6777, 32768, 0])
    out = src.vszip.PlaneAverage(exclude=[300, 5000]).vszip.PlaneAverage(exclude=[300, 5000], prop="avg_test")
    p = out.get_frame(0).props
    assert p["psmAvg"] == 0.10341039139391164  # 6777 / 65535
    assert p["avg_testAvg"] == p["psmAvg"]
    multi = src.vszip.PlaneAverage(exclude=[-1], planes=[0, 1, 2]).get_frame(0).props["psmAvg"]
    assert multi == [6777 / 65535, 32768 / 65535, 0.0]


def test_planeaverage_errors(vs):
    """test_planeaverage.py:149-175"""
    with pytest.raises(vs.Error, match="32-bit integer"):
        vs.blank(vs.GRAY32, 64, 32, 123456).vszip.PlaneAverage(exclude=[-1])
    src = _clip(vs, vs.YUV420P16, _yuv(vs))
    with pytest.raises(vs.Error, match="plane index out of range"):
        src.vszip.PlaneAverage(exclude=[-1], planes=[3])
    with pytest.raises(vs.Error, match="plane specified twice"):
        src.vszip.PlaneAverage(exclude=[-1], planes=[0, 0])
    with pytest.raises(vs.Error, match="exclude"):
        src.vszip.PlaneAverage()
    with pytest.raises(vs.Error, match="second clip has less frames than input clip"):
        vs.blank(vs.GRAY8, 64, 32, 0, length=5).vszip.PlaneAverage(exclude=[-1], clipb=vs.blank(vs.GRAY8, 64, 32, 0, length=3))


# ---- EEDI3: reference tests/test_eedi3.py, "behavioral contract" ------------------------------------------------------


@pytest.fixture(scope="module")
def grays(vs):
    g = _gray(vs, vs.GRAYS)
    h, w = (g.shape[0] // 2) * 2, (g.shape[1] // 2) * 2
    return _clip(vs, vs.GRAYS, [g[:h, :w]]), np.ascontiguousarray(g[:h, :w])


def _y(clip, n=0):
    return clip.get_frame(n)[0]


def test_eedi3_geometry(vs, grays):
    """:89-110 — field doubles the height with dh, keeps the size without; field 2 doubles the frame count; EEDI3H
    doubles the width."""
    c, g = grays
    h, w = g.shape
    out = c.vszip.EEDI3(field=1, dh=True)
    assert (out.width, out.height) == (w, h * 2)
    out = c.vszip.EEDI3(field=1)
    assert (out.width, out.height) == (w, h)
    four = vs.source([[g]] * 4, vs.GRAYS)
    assert four.vszip.EEDI3(field=2).num_frames == 8
    out = c.vszip.EEDI3H(field=1, dh=True)
    assert (out.width, out.height) == (w * 2, h)


def test_eedi3h_matches_transpose_eedi3(vs, grays):
    """:113-120 — EEDI3H is bit-exact to Transpose -> EEDI3 -> Transpose for every option combination."""
    c, g = grays
    t = _clip(vs, vs.GRAYS, [np.ascontiguousarray(g.T)])
    for kw in (dict(field=1), dict(field=0, vcheck=0), dict(field=1, dh=True), dict(field=1, hp=True, vcheck=3), dict(field=1, nrad=3, mdis=40)):
        assert np.array_equal(_y(c.vszip.EEDI3H(**kw)), _y(t.vszip.EEDI3(**kw)).T), kw


def test_all_planes_processed(vs, grays):
    """:123-126"""
    _, g = grays
    planes = [g, np.ascontiguousarray(np.roll(g, 7, axis=1)) - np.float32(0.5), np.ascontiguousarray(np.roll(g, 9, axis=0)) - np.float32(0.5)]
    out = _clip(vs, vs.YUV444PS, planes).vszip.EEDI3(field=1).get_frame(0)
    for p in range(3):
        assert not np.array_equal(out[p], planes[p])


def test_options_change_the_output(vs, grays):
    """:129-143 — a larger mdis, hp (half-pel steps are implemented, unlike eedi3m) and vcheck each change the result."""
    c, _ = grays
    assert not np.array_equal(_y(c.vszip.EEDI3(field=1, mdis=1)), _y(c.vszip.EEDI3(field=1, mdis=40)))
    assert not np.array_equal(_y(c.vszip.EEDI3(field=1, hp=True)), _y(c.vszip.EEDI3(field=1, hp=False)))
    assert not np.array_equal(_y(c.vszip.EEDI3(field=1, vcheck=0)), _y(c.vszip.EEDI3(field=1, vcheck=3)))


def test_float_output_is_finite(vs, grays):
    """:146-152 — the 4-tap cubic overshoots the nominal range and is not clamped; nothing blows up."""
    _, g = grays
    planes = [g, g - np.float32(0.5), np.float32(0.5) - g]
    out = _clip(vs, vs.YUV444PS, planes).vszip.EEDI3(field=1).get_frame(0)
    for p in range(3):
        assert -2.0 < float(out[p].min()) <= float(out[p].max()) < 2.0


def test_eedi3_stride_handling(vs, grays):
    """:155-161 — odd width (cropped by 19 columns, offset plane pointer) against the repacked clip."""
    _, g = grays
    cropped = np.ascontiguousarray(g[:, 19:])
    a = _clip(vs, vs.GRAYS, [cropped], extra_stride=19 * 4 + 64, offset=19 * 4).vszip.EEDI3(field=1, mdis=10)
    b = _clip(vs, vs.GRAYS, [cropped]).vszip.EEDI3(field=1, mdis=10)
    assert np.array_equal(_y(a), _y(b))


def test_sclip_and_mclip_change_the_output(vs, grays):
    """:163-186 — a custom sclip changes the vcheck blend; an edge mask differs from an empty mask (plain cubic); a
    float Gray mask is converted internally and runs."""
    c, g = grays
    blurred = c.vszip.BoxBlur(hradius=4, vradius=4)
    assert not np.array_equal(_y(c.vszip.EEDI3(field=1, vcheck=3, sclip=blurred)), _y(c.vszip.EEDI3(field=1, vcheck=3)))
    gy, gx = np.gradient(g.astype(np.float64))
    edge = ((np.hypot(gx, gy) > 0.05) * 255).astype(np.uint8)
    masked = c.vszip.EEDI3(field=1, mclip=_clip(vs, vs.GRAY8, [edge]))
    cubic = c.vszip.EEDI3(field=1, mclip=vs.blank(vs.GRAY8, g.shape[1], g.shape[0], 0))
    assert not np.array_equal(_y(masked), _y(cubic))
    vs.core_standins(True)  # (a float mask goes through the host's resize; the test host has a stand-in)
    try:
        c.vszip.EEDI3(field=1, mclip=_clip(vs, vs.GRAYS, [(edge / 255).astype(np.float32)])).get_frame(0)
    finally:
        vs.core_standins(False)


def test_eedi3_rejections(vs, grays):
    """:191-246"""
    c, g = grays
    with pytest.raises(vs.Error, match="32-bit float"):
        _clip(vs, vs.GRAY16, [_gray(vs, vs.GRAY16)]).vszip.EEDI3(field=1).get_frame(0)
    with pytest.raises(vs.Error, match="height must be mod 2"):
        _clip(vs, vs.GRAYS, [g[:-1]]).vszip.EEDI3(field=1).get_frame(0)
    with pytest.raises(vs.Error, match="width must be mod 2"):
        _clip(vs, vs.GRAYS, [g[:, :-1]]).vszip.EEDI3H(field=1).get_frame(0)
    yuv = _clip(vs, vs.YUV444PS, [g, g - np.float32(0.5), g - np.float32(0.5)])
    with pytest.raises(vs.Error, match="mclip must be Gray"):
        yuv.vszip.EEDI3(field=1, mclip=yuv).get_frame(0)
    wide = np.pad(g, ((0, 0), (0, 2)))
    with pytest.raises(vs.Error, match="mclip's dimensions"):
        c.vszip.EEDI3(field=1, mclip=_clip(vs, vs.GRAY8, [(wide > 0.1).astype(np.uint8)])).get_frame(0)
    with pytest.raises(vs.Error, match="sclip"):
        c.vszip.EEDI3(field=1, vcheck=2, sclip=_clip(vs, vs.GRAYS, [wide])).get_frame(0)


@pytest.mark.parametrize("fmt", ["GRAYS", "YUV420PS", "YUV422PS", "YUV444PS", "RGBS"])
def test_all_float_formats_run(vs, grays, fmt):
    """:249-251"""
    _, g = grays
    fid = {"YUV422PS": vs.fmt_id(vs.YUV, vs.FLOAT, 32, 1, 0)}.get(fmt) or getattr(vs, fmt)
    ssw, ssh = (fid >> 8) & 0xFF, fid & 0xFF
    n = 1 if fmt == "GRAYS" else 3
    planes = [np.ascontiguousarray(g if p == 0 else g[:: 1 << ssh, :: 1 << ssw]) for p in range(n)]
    _clip(vs, fid, planes).vszip.EEDI3(field=1).get_frame(0)


# ---- Limiter: reference tests/test_limiter.py -------------------------------------------------------------------------


def _stacked(vs, fid):
    """top half all-peak, bottom half all-floor (floor is -2 for float)"""
    is_float = (fid >> 24) & 0xF == vs.FLOAT
    bits = (fid >> 16) & 0xFF
    peak, floor = (2.0, -2.0) if is_float else ((1 << bits) - 1, 0)
    dt = np.float32 if is_float else (np.uint8 if bits <= 8 else np.uint16)
    ssw, ssh = (fid >> 8) & 0xFF, fid & 0xFF
    planes = []
    for p in range(3):
        h, w = 64 >> (ssh if p else 0), 64 >> (ssw if p else 0)
        planes.append(np.concatenate([np.full((h // 2, w), peak, dt), np.full((h // 2, w), floor, dt)], axis=0))
    return _clip(vs, fid, planes), planes


def _min_max(frame):
    return [p.min().item() for p in frame.planes], [p.max().item() for p in frame.planes]


TV_RANGE = [(8, [16, 16, 16], [235, 240, 240]), (9, [32, 32, 32], [470, 480, 480]), (10, [64, 64, 64], [940, 960, 960]),
            (12, [256, 256, 256], [3760, 3840, 3840]), (14, [1024, 1024, 1024], [15040, 15360, 15360]), (16, [4096, 4096, 4096], [60160, 61440, 61440]),
            (32, [0.0, -0.5, -0.5], [1.0, 0.5, 0.5])]


@pytest.mark.parametrize(("bits", "lo", "hi"), TV_RANGE)
def test_tv_range(vs, bits, lo, hi):
    """:112-115 — YUV420P8/9/10/12/14/16 and YUV420PS"""
    fid = vs.fmt_id(vs.YUV, vs.FLOAT if bits == 32 else vs.INTEGER, bits, 1, 1)
    src, _ = _stacked(vs, fid)
    assert _min_max(src.vszip.Limiter(tv_range=True).get_frame(0)) == (lo, hi)
    if bits <= 10:  # :133-137 — the compile-time tv_range path and the runtime min / max path agree
        a, b = src.vszip.Limiter(tv_range=True).get_frame(0), src.vszip.Limiter(min=lo, max=hi).get_frame(0)
        assert all(np.array_equal(a[p], b[p]) for p in range(3))


def test_limiter_defaults(vs):
    """:118-130 — mask clamps float chroma to 0..1; float default clamps to the full range; integer default is a no-op."""
    src, _ = _stacked(vs, vs.YUV420PS)
    assert _min_max(src.vszip.Limiter(tv_range=True, mask=True).get_frame(0)) == ([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    assert _min_max(src.vszip.Limiter().get_frame(0)) == ([0.0, -0.5, -0.5], [1.0, 0.5, 0.5])
    src16, planes = _stacked(vs, vs.YUV420P16)
    out = src16.vszip.Limiter().get_frame(0)
    assert all(np.array_equal(out[p], planes[p]) for p in range(3))


# ---- AdaptiveBinarize: reference tests/test_adaptive_binarize.py ------------------------------------------------------


def test_output_is_binary_and_full_range(vs):
    """:64-68"""
    g = _gray(vs, vs.GRAY8)
    src = _clip(vs, vs.GRAY8, [g])
    f = src.vszip.AdaptiveBinarize(src.vszip.BoxBlur(hradius=5, vradius=5)).get_frame(0)
    assert set(np.unique(f[0]).tolist()) <= {0, 255} and len(np.unique(f[0])) == 2
    assert f.props["_ColorRange"] == 0  # RANGE_FULL


@pytest.mark.parametrize("c", [0, 3, 10])
def test_threshold_rule_exact(vs, c):
    """:71-82 — out = 255 where src <= clip2 - c, else 0 (OpenCV ADAPTIVE_THRESH_MEAN_C with THRESH_BINARY_INV), on a
    0..255 ramp against a constant 128."""
    ramp = np.tile(np.arange(256, dtype=np.uint8), (2, 1))
    out = _clip(vs, vs.GRAY8, [ramp]).vszip.AdaptiveBinarize(vs.blank(vs.GRAY8, 256, 2, 128), c=c).get_frame(0)[0]
    assert out[0].tolist() == [255 if x <= 128 - c else 0 for x in range(256)]


def test_higher_c_is_stricter(vs):
    """:85-90"""
    src = _clip(vs, vs.GRAY8, [_gray(vs, vs.GRAY8)])
    blur = src.vszip.BoxBlur(hradius=5, vradius=5)
    a3 = src.vszip.AdaptiveBinarize(blur, c=3).get_frame(0)[0].mean()
    a10 = src.vszip.AdaptiveBinarize(blur, c=10).get_frame(0)[0].mean()
    assert a10 < a3


def test_non_8bit_error(vs):
    """:93-96"""
    src16 = _clip(vs, vs.GRAY16, [_gray(vs, vs.GRAY16)])
    with pytest.raises(vs.Error, match="only 8 bit int format supported"):
        src16.vszip.AdaptiveBinarize(src16)
