"""CPU oracle vs the reference's goldens / known-answer tests for Bilateral, EEDI3(H),
PlaneAverage, PlaneMinMax, SSIMULACRA2, XPSNR. Keys are the reference's own
(tests/goldens/*.json; cases in tests/test_*.py) on the inputs that are reproducible
without VapourSynth/zimg."""
import math

import numpy as np
import pytest

import fixtures as fx

REL = 1e-6


def _check(stats, gold, rel=REL):
    for k in ("avg", "min", "max"):
        assert stats[k] == pytest.approx(gold[k], rel=rel, abs=1e-9), (k, stats[k], gold[k])


# ---- Bilateral --------------------------------------------------------------
def _bilateral(oracle, planes, sigmaS, sigmaR, **kw):
    prm = oracle.bilateral_params([sigmaS], [sigmaR], **kw)
    out = []
    for i, p in enumerate(planes):
        out.append(oracle.bilateral_plane(np.ascontiguousarray(p), prm["sigmaS"][i], prm["sigmaR"][i], prm["algorithm"][i], prm["radius"][i],
                                          prm["step"][i], prm["PBFICnum"][i]))
    return out, prm


def test_bilateral_params_baseline(oracle):
    """BASELINE config: sigmaS=2 sigmaR=2 on YUV420 -> sigmaS [2,1,1], radius/step luma 3/2 chroma 2/1,
    algorithm 2 everywhere, PBFICnum [4,5,5]."""
    prm = oracle.bilateral_params([2], [2], yuv=True, ssw=1, ssh=1)
    assert prm["sigmaS"] == [2.0, 1.0, 1.0]
    assert prm["radius"] == [3, 2, 2] and prm["step"] == [2, 1, 1] and prm["samples"] == [2, 2, 2]
    assert prm["algorithm"] == [2, 2, 2] and prm["PBFICnum"] == [4, 5, 5]


def test_bilateral_golden_rgb24(oracle):
    g = fx.ref_goldens()["exact"]["bilateral"]["RGB24|full|sigmaR=2,sigmaS=2"]
    out, _ = _bilateral(oracle, fx.crop_rgb24(), 2, 2)
    for p in range(3):
        _check(fx.plane_stats(out[p]), g[f"p{p}"])


def test_bilateral_golden_rgbs(oracle):
    g = fx.ref_goldens()["exact"]["bilateral"]["RGBS|full|sigmaR=2,sigmaS=2"]
    out, _ = _bilateral(oracle, fx.crop_rgbs(), 2, 2)
    for p in range(3):
        _check(fx.plane_stats(out[p]), g[f"p{p}"])


def test_bilateral_golden_gray8(oracle):
    g = fx.ref_goldens()["exact"]["bilateral"]["GRAY8|full|sigmaR=2,sigmaS=2"]
    out, _ = _bilateral(oracle, [fx.crop_gray8()], 2, 2)
    _check(fx.plane_stats(out[0]), g["p0"])


@pytest.mark.parametrize("key,sS,sR,alg,num", [
    ("GRAY16|full|PBFICnum=4,algorithm=1,sigmaR=0.1,sigmaS=3", 3, 0.1, 1, 4),
    ("GRAY16|full|PBFICnum=32,algorithm=1,sigmaR=0.1,sigmaS=3", 3, 0.1, 1, 32),
    ("GRAY16|full|algorithm=2,sigmaR=0.02,sigmaS=3", 3, 0.02, 2, 0),
    ("GRAY16|full|sigmaR=2,sigmaS=5", 5, 2, 0, 0),
    ("GRAY16|full|sigmaR=0.02,sigmaS=0.8", 0.8, 0.02, 0, 0),
])
def test_bilateral_soft_gray16(oracle, key, sS, sR, alg, num):
    g = fx.ref_goldens()["soft"]["bilateral"][key]["p0"]
    prm = oracle.bilateral_params([sS], [sR], algorithm=[alg], pbficnum=[num])
    out = oracle.bilateral_plane(np.ascontiguousarray(fx.crop_gray16()), prm["sigmaS"][0], prm["sigmaR"][0], prm["algorithm"][0], prm["radius"][0],
                                 prm["step"][0], prm["PBFICnum"][0])
    st = fx.plane_stats(out)
    assert st["avg"] == pytest.approx(g["avg"], rel=1e-7)
    assert abs(st["min"] - g["min"]) <= 1 and abs(st["max"] - g["max"]) <= 1


def test_bilateral_soft_gray16_joint_ref(oracle):
    """The joint path (`ref` clip, reference tests/test_bilateral.py:40-44: ref = std.BoxBlur(5,5) of the
    source): GRAY16|full|sigmaR=0.05,sigmaS=2|ref on the approximate GRAY16 fixture."""
    from oracle import vs_host as vh

    g = fx.ref_goldens()["soft"]["bilateral"]["GRAY16|full|sigmaR=0.05,sigmaS=2|ref"]["p0"]
    src = np.ascontiguousarray(fx.crop_gray16())
    prm = oracle.bilateral_params([2], [0.05])
    out = oracle.bilateral_plane(src, prm["sigmaS"][0], prm["sigmaR"][0], prm["algorithm"][0], prm["radius"][0], prm["step"][0], prm["PBFICnum"][0],
                                 ref=vh.std_boxblur(src, 5, 5))
    st = fx.plane_stats(out)
    assert st["avg"] == pytest.approx(g["avg"], rel=1e-7)
    assert abs(st["min"] - g["min"]) <= 1 and abs(st["max"] - g["max"]) <= 1


# ---- EEDI3 / EEDI3H -----------------------------------------------------------
def test_eedi3_golden_rgbs(oracle):
    g = fx.ref_goldens()["exact"]["eedi3"]["RGBS|full|field=1"]
    for p in range(3):
        _check(fx.plane_stats(oracle.eedi3(fx.crop_rgbs()[p], field=1)), g[f"p{p}"])


def test_eedi3h_golden_rgbs(oracle):
    g = fx.ref_goldens()["exact"]["eedi3h"]["RGBS|full|field=1"]
    for p in range(3):
        _check(fx.plane_stats(oracle.eedi3(fx.crop_rgbs()[p], field=1, horizontal=True)), g[f"p{p}"])


def test_eedi3h_is_transposed_eedi3(oracle):
    """EEDI3H == Transpose o EEDI3 o Transpose, bit-exact (reference tests/test_eedi3.py:111-118)."""
    src = fx.crop_rgbs()[0][:96, :128]
    a = oracle.eedi3(src, field=1, horizontal=True)
    b = oracle.eedi3(np.ascontiguousarray(src.T), field=1).T
    assert np.array_equal(a, b)


@pytest.mark.parametrize("key,kw", [
    ("GRAYS|full|dh=1,field=1", dict(field=1, dh=True)),
    ("GRAYS|full|field=0", dict(field=0)),
    ("GRAYS|full|field=1,mdis=40,nrad=3", dict(field=1, mdis=40, nrad=3)),
    ("GRAYS|full|field=1,gamma=0", dict(field=1, gamma=0.0)),
    ("GRAYS|full|field=1,hp=1", dict(field=1, hp=True)),
])
def test_eedi3_soft_grays(oracle, key, kw):
    """Approximate GRAYS fixture: EEDI3's discrete path can flip on ulp-level input
    differences, so this is a magnitude check (1e-5), not a gate."""
    g = fx.ref_goldens()["soft"]["eedi3"][key]["p0"]
    st = fx.plane_stats(oracle.eedi3(fx.crop_grays(), **kw))
    assert st["avg"] == pytest.approx(g["avg"], rel=1e-5)


# ---- PlaneAverage / PlaneMinMax ----------------------------------------------------
def test_planeaverage_goldens(oracle):
    g = fx.ref_goldens()["exact"]["planeaverage"]
    rgb, rgbs, g8 = fx.crop_rgb24(), fx.crop_rgbs(), fx.crop_gray8()
    avg = lambda p, ex: oracle.plane_average(np.ascontiguousarray(p), ex)[0]
    assert avg(rgb[0], [-1]) == pytest.approx(g["RGB24|full|exclude=[-1]"]["avg"], rel=1e-12)
    assert [avg(rgb[i], [-1]) for i in range(3)] == pytest.approx(g["RGB24|full|exclude=[-1],planes=[0,1,2]"]["avg"], rel=1e-12)
    assert [avg(rgbs[i], [-1]) for i in range(3)] == pytest.approx(g["RGBS|full|exclude=[-1],planes=[0,1,2]"]["avg"], rel=1e-12)
    assert avg(g8, [-1]) == pytest.approx(g["GRAY8|full|exclude=[-1]"]["avg"], rel=1e-12)
    assert avg(g8, [128]) == pytest.approx(g["GRAY8|full|exclude=[128]"]["avg"], rel=1e-12)
    assert avg(g8, [100, 150, 200]) == pytest.approx(g["GRAY8|full|exclude=[100,150,200]"]["avg"], rel=1e-12)
    assert avg(g8[:-1, :-1], [-1]) == pytest.approx(g["GRAY8|odd|exclude=[-1]"]["avg"], rel=1e-12)
    assert avg(g8[100:107, 200:213], [-1]) == pytest.approx(g["GRAY8|tiny|exclude=[-1]"]["avg"], rel=1e-12)


def test_planeaverage_known_answers(oracle):
    """reference tests/test_planeaverage.py:118-147 (two-tone exclude, 6777/65535)."""
    two = np.concatenate([np.full((32, 64), 1000, np.uint16), np.full((32, 64), 3000, np.uint16)], axis=1)
    assert oracle.plane_average(two, [1000])[0] == 3000 / 65535
    assert oracle.plane_average(two, [3000])[0] == 1000 / 65535
    assert oracle.plane_average(two, [1000, 3000])[0] == 0.0
    assert oracle.plane_average(np.full((32, 64), 6777, np.uint16), [-1])[0] == 0.10341039139391164
    f = np.concatenate([np.full((32, 64), 1.0, np.float32), np.full((32, 64), 3.0, np.float32)], axis=1)
    assert oracle.plane_average(f, [3])[0] == 1.0


def test_planeminmax_goldens(oracle):
    g = fx.ref_goldens()["exact"]["planeminmax"]
    rgb, rgbs, g8 = fx.crop_rgb24(), fx.crop_rgbs(), fx.crop_gray8()
    mm = lambda p, lo, hi, ref=None: oracle.plane_minmax(np.ascontiguousarray(p), lo, hi, ref)
    e = g["RGB24|full|maxthr=0.1,minthr=0.1"]
    assert mm(rgb[0], 0.1, 0.1)[:2] == (e["Min"], e["Max"])
    e = g["GRAY8|full|maxthr=0.1,minthr=0.1"]
    assert mm(g8, 0.1, 0.1)[:2] == (e["Min"], e["Max"])
    e = g["RGBS|full|maxthr=0.3,minthr=0.2,planes=[0,1,2]"]
    for i in range(3):
        lo, hi, _ = mm(rgbs[i], 0.2, 0.3)
        assert lo == pytest.approx(e["Min"][i], rel=1e-7) and hi == pytest.approx(e["Max"][i], rel=1e-7)
    # clipb = vszip.BoxBlur(1,1) (reference tests/test_planeminmax.py:73-75): pins the CT integer BoxBlur too
    e = g["RGB24|full|maxthr=0.1,minthr=0.1,planes=[0,1,2]|ref"]
    for i in range(3):
        p = np.ascontiguousarray(rgb[i])
        lo, hi, df = mm(p, 0.1, 0.1, oracle.boxblur(p, 1, 1, 1, 1))
        assert (lo, hi) == (e["Min"][i], e["Max"][i])
        assert df == pytest.approx(e["Diff"][i], rel=1e-12)


def test_planeminmax_known_answers(oracle):
    """reference tests/test_planeminmax.py:99-110,228-236: 25% zeros + thresholds; minthr=1 -> peak, maxthr=1 -> 0."""
    p = np.full((32, 64), 200, np.uint8)
    p[:8, :] = 0
    assert oracle.plane_minmax(p, 0.2, 0.0)[:2] == (0, 200)
    assert oracle.plane_minmax(p, 0.3, 0.0)[:2] == (200, 200)
    q = np.full((32, 64), 1234, np.uint16)
    assert oracle.plane_minmax(q, 1.0, 0.0)[0] == 65535
    assert oracle.plane_minmax(q, 0.0, 1.0)[1] == 0


# ---- SSIMULACRA2 -----------------------------------------------------------------------
def _srgb_to_linear(v):
    v = v.astype(np.float64)
    return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4).astype(np.float32)


def test_ssimulacra2_identical_is_100(oracle):
    """reference tests/test_ssimulacra2.py:65-71"""
    c = [np.full((64, 64), v, np.float32) for v in (0.3, 0.2, 0.5)]
    assert oracle.ssimulacra2(c, c) == 100.0
    lin = [_srgb_to_linear(p) for p in fx.crop_rgbs()]
    assert oracle.ssimulacra2(lin, lin) > 99.9


def test_ssimulacra2_vec_size_sensitivity(oracle):
    """The oracle (and the HIP kernels) restate the reference as built with vec_size = 8 (x86_64_v3, the
    CI build and the haswell wheel); the znver4 wheel has vec_size = 16 (hatch_build.py:13-17), which moves
    the column where blurV stops fusing and the f64 lane-sum order. Measured on the reference's test crop
    (RGB24, dist = std.BoxBlur(1,1), score 4.41): 3.5e-5 relative at 640x320 (the 40- and 20-column scales of
    the pyramid have 8 resp. 4 tail columns under one build and 8 resp. 4 + 0 under the other — the deep scales
    carry the large weights), 4.8e-5 at 631x313, 7e-8 at 13x7. I.e. the reference's own two wheels disagree
    by ~4e-5 relative; anything below that is not a parity statement about "the" reference. Bound asserted:
    1e-4 (the reference's own tolerance is rel=1e-3)."""
    from oracle import vs_host as vh

    ref8 = [np.ascontiguousarray(p) for p in fx.crop_rgb24()]
    worst = {}
    for name, crop in (("640x320", lambda p: p), ("631x313", lambda p: np.ascontiguousarray(p[:313, :631])), ("13x7", lambda p: np.ascontiguousarray(p[100:107, 200:213]))):
        ref = [crop(p) for p in ref8]
        lin_r = vh.to_linear_rgbs(ref, "RGB", 8)
        lin_d = vh.to_linear_rgbs([vh.std_boxblur(p, 1, 1) for p in ref], "RGB", 8)
        s8 = oracle.ssimulacra2(lin_r, lin_d)
        try:
            assert oracle.ssim_set_vec(16) == 8
            s16 = oracle.ssimulacra2(lin_r, lin_d)
        finally:
            oracle.ssim_set_vec(8)
        worst[name] = abs(s16 - s8) / max(abs(s8), 1e-30)
    print("SSIMULACRA2 vec_size 8 vs 16, relative score change:", {k: f"{v:.2e}" for k, v in worst.items()})
    assert 0 < max(worst.values()) <= 1e-4


# The reference's golden keys for SSIMULACRA2 (RGBS / RGB24 / GRAY8, blur1 / blur3) are pinned in
# tests/test_oracle_vs_host.py, together with the std.BoxBlur / zimg restatements they need.


# ---- XPSNR -------------------------------------------------------------------------------
def test_xpsnr_identical_is_inf(oracle):
    """reference tests/test_xpsnr.py:222-225"""
    y = fx.splitmix64_plane(3, (64, 96), np.uint8)
    u = fx.splitmix64_plane(4, (32, 48), np.uint8)
    w = oracle.xpsnr_wsse([y, u, u], [y, u, u], depth=8)
    assert w == [0, 0, 0]
    assert math.isinf(oracle.xpsnr_frame(0, 96, 64, 8))


def _xpsnr_dist(y, kind):
    """reference tests/test_xpsnr.py:60-72: the distorted clip of a golden case (luma plane)."""
    from oracle import vs_host as vh

    if kind == "box2":
        return vh.std_boxblur(y, 2, 2)
    if kind == "box5":
        return vh.std_boxblur(y, 5, 5)
    add = 12 if kind == "bright" else 1  # std.Expr("x 12 +") / ("x 1 +"), clamped to the format's peak
    peak = 255 if y.dtype == np.uint8 else 1023
    return np.minimum(y.astype(np.int32) + add, peak).astype(y.dtype)


@pytest.mark.parametrize("kind", ["box2", "box5", "bright", "shift"])
@pytest.mark.parametrize("temporal", [0, 1])
def test_xpsnr_golden_luma_yuv420p8(oracle, kind, temporal):
    """reference tests/test_xpsnr.py:113-131 + tests/goldens/xpsnr.json, YUV420P8 keys, XPSNR_Y of frames
    0..2 (rel=1e-6 there; reproduced to the last bit or one ulp here). XPSNR_Y depends on the luma planes
    only, and plane 0 of the 3-frame Point-converted fixture is the exact 8-bit BT.709 luma of the
    shifted crops; the clip runs at ImageRead's 30 fps (first-order temporal, image_read.zig:420).
    Pins SSE, the 3x3 spatial activity, tempDiff1 with and without a previous frame, the <=640x480
    min-smoothing, the weighting and avg_act (src/filters/xpsnr.zig:174-212,253-357,376-474)."""
    g = fx.ref_goldens()["luma_of_yuv"]["xpsnr_Y"]
    for n in range(3):
        y = fx.luma8(fx.temporal_rgb24(n))
        p1 = fx.luma8(fx.temporal_rgb24(n - 1)) if (temporal and n > 0) else None
        w = oracle.xpsnr_wsse([y], [_xpsnr_dist(y, kind)], prv1=p1, depth=8, frame_rate=30, temporal=bool(temporal))
        assert oracle.xpsnr_frame(w[0], 640, 320, 8) == pytest.approx(g[f"YUV420P8|full|temporal={temporal}|{kind}|n{n}"], rel=1e-14)


@pytest.mark.parametrize("kind", ["bright", "shift"])
@pytest.mark.parametrize("temporal", [0, 1])
def test_xpsnr_golden_chroma_of_offset_distortions(oracle, kind, temporal):
    """For the `x + k` distortions the chroma SSE is k^2 per sample whatever the chroma content (limited-range
    chroma never reaches the clamp), so XPSNR_U/V of those goldens are reachable too: pins the chroma branch —
    luma weights reused per block, bx/by scaling for 4:2:0, avg_act (src/filters/xpsnr.zig:476-521)."""
    import json

    gold = json.loads((fx.GOLDEN_DIR / "ref_goldens.json").read_text())["luma_of_yuv"]["xpsnr_UV"]
    for n in range(3):
        y = fx.luma8(fx.temporal_rgb24(n))
        p1 = fx.luma8(fx.temporal_rgb24(n - 1)) if (temporal and n > 0) else None
        c = np.full((160, 320), 128, np.uint8)
        w = oracle.xpsnr_wsse([y, c, c], [_xpsnr_dist(y, kind), _xpsnr_dist(c, kind), _xpsnr_dist(c, kind)], prv1=p1, depth=8, frame_rate=30,
                              temporal=bool(temporal))
        e = gold[f"YUV420P8|full|temporal={temporal}|{kind}|n{n}"]
        assert oracle.xpsnr_frame(w[1], 320, 160, 8) == pytest.approx(e["U"], rel=1e-14)
        assert oracle.xpsnr_frame(w[2], 320, 160, 8) == pytest.approx(e["V"], rel=1e-14)


def test_xpsnr_golden_luma_yuv420p10_soft(oracle):
    """10-bit keys on an approximate luma (zimg's f32 operation order is not recovered: a few samples
    are off by one): still within 2e-4 of the goldens."""
    g = fx.ref_goldens()["luma_of_yuv"]["xpsnr_Y"]

    def luma10(rgb):
        f = np.float32
        r, gg, b = (rgb[i].astype(np.float32) * f(1.0 / 255.0) for i in range(3))
        yy = (f(0.2126) * r + f(0.7152) * gg) + f(0.0722) * b
        return np.floor(yy * f(876.0) + f(64.0) + f(0.5)).astype(np.uint16)

    for kind in ("box2", "shift"):
        for n in range(3):
            y = luma10(fx.temporal_rgb24(n))
            p1 = luma10(fx.temporal_rgb24(n - 1)) if n > 0 else None
            w = oracle.xpsnr_wsse([y], [_xpsnr_dist(y, kind)], prv1=p1, depth=10, frame_rate=30, temporal=True)
            assert oracle.xpsnr_frame(w[0], 640, 320, 10) == pytest.approx(g[f"YUV420P10|full|temporal=1|{kind}|n{n}"], rel=2e-4)


def _xpsnr_flat_expected(w, h, depth, delta, act_t):
    """Hand evaluation of getWSSE (src/filters/xpsnr.zig:376-474) for a flat reference frame and a
    reconstruction that is off by `delta` everywhere: the spatial activity is 0, so a block's ms_act is its
    temporal activity `act_t` (0 with temporal=False), floored at 2^(depth-6) and squared (:343-350); every
    weight is 1/sqrt(ms_act) and the <=640x480 smoothing cannot change equal weights."""
    r = (w * h) / (3840.0 * 2160.0)
    b = int(32.0 * math.sqrt(r) + 0.5) * 4
    avg_act = math.sqrt(16.0 * float(1 << (2 * depth - 9)) / math.sqrt(max(0.00001, r)))
    ms = max(float(act_t), float(1 << (depth - 6)))
    wgt = 1.0 / math.sqrt(ms * ms)
    tot = 0.0
    for y0 in range(0, h, b):
        for x0 in range(0, w, b):
            tot += float(delta * delta * min(b, w - x0) * min(b, h - y0)) * wgt
    return b, int(tot * avg_act + 0.5)


@pytest.mark.parametrize("w,h,depth,b_expect", [(1920, 1080, 8, 64), (3840, 2160, 8, 128), (1920, 1080, 10, 64), (640, 480, 8, 24), (2560, 1440, 10, 84)])
def test_xpsnr_known_answer_flat_spatial(oracle, w, h, depth, b_expect):
    """Both block-size regimes (b = 64 at 1080p; `highds` above 2048x1152), the <=640x480 smoothing branch and
    10 bit, temporal=False: wsse in closed form, luma and 4:2:0 chroma."""
    dt = np.uint8 if depth == 8 else np.uint16
    y, c = np.full((h, w), 100, dt), np.full((h // 2, w // 2), 90, dt)
    got = oracle.xpsnr_wsse([y, c, c], [y + dt(1), c + dt(1), c + dt(1)], depth=depth, temporal=False)
    b, wl = _xpsnr_flat_expected(w, h, depth, 1, 0)
    assert b == b_expect
    assert got[0] == wl
    assert got[1] == got[2] and got[1] == pytest.approx(wl / 4, abs=1)  # same weights, a quarter of the samples
    # weight 1/2^(depth-6): e.g. 1080p 8 bit -> 2073600 / 4 * 64
    if (w, h, depth) == (1920, 1080, 8):
        assert wl == 33177600


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160)])
@pytest.mark.parametrize("fps,cur,p1,p2,act", [
    (24, 100, 90, None, 2 * 10),         # first order: XPSNR_GAMMA * |cur - p1|
    (24, 100, None, None, 2 * 100),      # frame 0: the missing previous frame counts as zeros
    (60, 100, 90, 70, 2 * 10),           # second order: |cur - 2 p1 + p2| = |100 - 180 + 70|
    (60, 100, 90, None, 2 * 80),         # frame 1 at >= 32 fps: |cur - 2 p1|
    (31, 100, 99, None, 2 * 1),          # activity 2 < 2^(8-6): floored at 4
])
def test_xpsnr_known_answer_flat_temporal(oracle, w, h, fps, cur, p1, p2, act):
    """tempDiff1/2 (<= 2048x1152) and diff1st/2nd (2x2 sums, above) on flat frames, fps < 32 vs >= 32
    (src/filters/xpsnr.zig:66-170, 312-341)."""
    y = np.full((h, w), cur, np.uint8)
    mk = lambda v: None if v is None else np.full((h, w), v, np.uint8)
    got = oracle.xpsnr_wsse([y], [y + np.uint8(3)], prv1=mk(p1), prv2=mk(p2), depth=8, frame_rate=fps, temporal=True)
    assert got[0] == _xpsnr_flat_expected(w, h, 8, 3, act)[1]


def test_xpsnr_known_answer_checkerboard(oracle):
    """A one-sample checkerboard a/b has 3x3 activity |12c - 2(l+r+u+d) - (4 diagonals)| = 8|a-b| at every
    sample, so ms_act = (8|a-b|)^2 and every weight is 1/(8|a-b|) (src/filters/xpsnr.zig:174-212)."""
    w, h, a, b_ = 1920, 1080, 120, 118
    yy, xx = np.mgrid[0:h, 0:w]
    y = np.where((xx + yy) % 2 == 0, a, b_).astype(np.uint8)
    got = oracle.xpsnr_wsse([y], [y + np.uint8(2)], depth=8, temporal=False)
    tot = 0.0
    for y0 in range(0, h, 64):
        for x0 in range(0, w, 64):
            tot += float(4 * min(64, w - x0) * min(64, h - y0)) * (1.0 / math.sqrt(16.0 * 16.0))
    assert got[0] == int(tot * 64.0 + 0.5)


def test_xpsnr_ordering(oracle):
    """reference tests/test_xpsnr.py:178-190 in spirit: a stronger blur scores lower."""
    y = fx.luma8(fx.temporal_rgb24(0))
    s = [oracle.xpsnr_frame(oracle.xpsnr_wsse([y], [_xpsnr_dist(y, k)], depth=8, temporal=False)[0], 640, 320, 8) for k in ("shift", "bright", "box2", "box5")]
    assert s[0] > s[1] > s[2] > s[3]


# ---- Limiter ---------------------------------------------------
@pytest.mark.parametrize("key,lo,hi,planes", [
    ("RGB24|full|max=[180,200,250],min=[20,20,100]", [20, 20, 100], [180, 200, 250], [0, 1, 2]),
    ("RGB24|full|max=[180,200,250],min=[20,20,100],planes=[0,2]", [20, 20, 100], [180, 200, 250], [0, 2]),
    ("RGB24|full|tv_range=1", None, None, [0, 1, 2]),
])
def test_limiter_golden_rgb24(oracle, key, lo, hi, planes):
    g = fx.ref_goldens()["exact"]["limiter"][key]
    if lo is None:  # the comptime tv_range table of an RGB clip (src/filters/limiter.zig:83)
        lo, hi = oracle.limiter_default_range(False, 8, False, True)
        assert (lo, hi) == ([16.0] * 3, [235.0] * 3)
    src = fx.crop_rgb24()
    for p in range(3):
        out = oracle.limiter(src[p], lo[p], hi[p]) if p in planes else src[p]
        _check(fx.plane_stats(out), g[f"p{p}"])


def test_limiter_golden_gray8_and_rgbs(oracle):
    g = fx.ref_goldens()["exact"]["limiter"]
    _check(fx.plane_stats(oracle.limiter(fx.crop_gray8(), 50, 200)), g["GRAY8|full|max=[200],min=[50]"]["p0"])
    lo, hi = [0.1, 0.1, 0.1], [0.7, 0.7, 0.99]
    for p in range(3):
        out = oracle.limiter(np.ascontiguousarray(fx.crop_rgbs()[p]), np.float32(lo[p]), np.float32(hi[p]))
        _check(fx.plane_stats(out), g["RGBS|full|max=[0.7,0.7,0.99],min=[0.1,0.1,0.1]"][f"p{p}"])


@pytest.mark.parametrize("key,lo,hi", [("GRAY16|full|max=[50000],min=[10000]", 10000, 50000), ("GRAY16|full|max=[30000],min=[0]", 0, 30000),
                                        ("GRAY16|full|max=[65535],min=[30000]", 30000, 65535)])
def test_limiter_soft_gray16(oracle, key, lo, hi):
    g = fx.ref_goldens()["soft"]["limiter"][key]["p0"]
    st = fx.plane_stats(oracle.limiter(fx.crop_gray16(), lo, hi))
    assert st["avg"] == pytest.approx(g["avg"], rel=1e-7) and abs(st["min"] - g["min"]) <= 1 and abs(st["max"] - g["max"]) <= 1


def test_limiter_default_tables(oracle):
    """reference tests/test_limiter.py:96-127 (TV_RANGE table, mask, float default) and the u32 tables :150-191"""
    assert oracle.limiter_default_range(False, 16, True, True) == ([4096.0] * 3, [60160.0, 61440.0, 61440.0])
    assert oracle.limiter_default_range(False, 10, True, True) == ([64.0] * 3, [940.0, 960.0, 960.0])
    assert oracle.limiter_default_range(False, 32, True, True) == ([268435456.0] * 3, [3942645760.0, 4026531840.0, 4026531840.0])
    assert oracle.limiter_default_range(False, 32, False, True)[1] == [3942645760.0] * 3
    assert oracle.limiter_default_range(False, 16, True, False) == ([0.0] * 3, [65535.0] * 3)
    assert oracle.limiter_default_range(True, 32, True, True) == ([0.0, -0.5, -0.5], [1.0, 0.5, 0.5])
    assert oracle.limiter_default_range(True, 32, False, True) == ([0.0] * 3, [1.0] * 3)  # mask / RGB / Gray
    ramp = np.tile(np.arange(256, dtype=np.uint8), (2, 1))
    assert oracle.limiter(ramp, 10, 200)[0].tolist() == [min(max(x, 10), 200) for x in range(256)]


# ---- LimitFilter -------------------------------------------------
# The reference's goldens are taken on flt = src.vszip.BoxBlur(hradius=2, vradius=2), LimitFilter(flt, src)
# (the construction of tests/test_int_parity.py:158-167; confirmed by the goldens themselves: the
# RGB24 key reproduces to every printed digit).
def _limit_filter(oracle, planes, dark, bright, elast, is_float=False, which=(0, 1, 2)):
    out = []
    for p, s in enumerate(planes):
        s = np.ascontiguousarray(s)
        flt = oracle.boxblur(s, 2, 1, 2, 1)
        if p not in which:
            out.append(flt)  # unprocessed planes are copied from flt (newVideoFrame2 on flt)
            continue
        d = oracle.scale_value_from_8bit(dark, is_float, 32 if is_float else 8 * s.itemsize, False)
        b = oracle.scale_value_from_8bit(bright, is_float, 32 if is_float else 8 * s.itemsize, False)
        out.append(oracle.limit_filter(flt, s, None, d, b, elast))
    return out


def test_limit_filter_goldens(oracle):
    g = fx.ref_goldens()["exact"]["limitfilter"]
    for key, planes, args in (("RGB24|full|bright_thr=8,dark_thr=8,elast=3", list(fx.crop_rgb24()), dict(dark=8, bright=8, elast=3)),
                              ("RGB24|full|bright_thr=8,dark_thr=8,planes=[0,2]", list(fx.crop_rgb24()), dict(dark=8, bright=8, elast=2, which=(0, 2))),
                              ("GRAY8|full|bright_thr=4,dark_thr=4,elast=2", [fx.crop_gray8()], dict(dark=4, bright=4, elast=2)),
                              ("RGBS|full|bright_thr=8,dark_thr=8,elast=3", list(fx.crop_rgbs()), dict(dark=8, bright=8, elast=3, is_float=True))):
        out = _limit_filter(oracle, planes, **args)
        for p in range(len(planes)):
            _check(fx.plane_stats(out[p]), g[key][f"p{p}"])


def test_scale_value(oracle):
    """hz.scaleValue defaults (src/helper.zig:312-336): identity at 8 bit, x257 full range / x256-ish limited at 16 bit, /255 for float"""
    assert oracle.scale_value_from_8bit(8, False, 8, True) == 8.0
    assert oracle.scale_value_from_8bit(8, False, 16, False) == float(np.round(np.float32(8) * (np.float32(65535) / np.float32(255))))
    assert oracle.scale_value_from_8bit(8, False, 16, True) == float(np.round(np.float32(8) * (np.float32(60160 - 4096) / np.float32(219))))
    assert oracle.scale_value_from_8bit(8, True, 32, False) == float(np.float32(8) * (np.float32(1) / np.float32(255)))
