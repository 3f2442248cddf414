"""End-to-end through the plugin boundary on the GPU: fake VS host -> libvszip.so ->
C ABI -> HIP kernels -> frames/props, compared with the CPU oracle and the reference's
known-answer tests (tests/test_*.py of the reference, cited per test)."""
import math

import numpy as np
import pytest

import fixtures as fx
from fakevs import fakevs as vs

pytestmark = pytest.mark.gpu


def _yuv420p16(seed=0, w=384, h=216):
    return [fx.splitmix64_plane(seed + p, s, np.uint16) for p, s in enumerate([(h, w), (h // 2, w // 2), (h // 2, w // 2)])]


def test_boxblur_yuv420p16_matches_oracle(oracle):
    planes = _yuv420p16()
    out = vs.source([planes], vs.YUV420P16).vszip.BoxBlur(hradius=13, vradius=13).get_frame(0)
    for p in range(3):
        assert np.array_equal(out[p], oracle.boxblur(planes[p], 13, 1, 13, 1))


def test_boxblur_planes_and_props(oracle):
    """planes=[0] copies the other planes (reference tests/test_boxblur.py:111-121); props pass through."""
    planes = _yuv420p16(3)
    out = vs.source([planes], vs.YUV420P16, props={"_Matrix": 1}).vszip.BoxBlur(planes=[0], hradius=5, vradius=5).get_frame(0)
    assert np.array_equal(out[1], planes[1]) and np.array_equal(out[2], planes[2])
    assert np.array_equal(out[0], oracle.boxblur(planes[0], 5, 1, 5, 1))
    assert out.props["_Matrix"] == 1


@pytest.mark.parametrize("fmt,dtype", [(vs.GRAY8, np.uint8), (vs.GRAY16, np.uint16), (vs.GRAYS, np.float32)])
@pytest.mark.parametrize("radius", [10, 30])
def test_boxblur_stride_handling(oracle, fmt, dtype, radius):
    """offset plane pointers + stride > width (reference tests/test_boxblur.py:122-128)."""
    p = fx.splitmix64_plane(9, (120, 613), dtype)
    a = vs.source([[p]], fmt, extra_stride=96, offset=54).vszip.BoxBlur(hradius=radius, vradius=radius).get_frame(0)
    b = vs.source([[p]], fmt).vszip.BoxBlur(hradius=radius, vradius=radius).get_frame(0)
    assert np.array_equal(a[0].view(np.uint8), b[0].view(np.uint8))
    assert np.array_equal(a[0].view(np.uint8), oracle.boxblur(p, radius, 1, radius, 1).view(np.uint8))


def test_boxblur_pass_composition():
    """reference tests/test_boxblur.py:86-101"""
    src = vs.source([[fx.splitmix64_plane(4, (90, 160), np.uint16)]], vs.GRAY16)
    two = src.vszip.BoxBlur(hradius=7, hpasses=2, vradius=0, vpasses=0).get_frame(0)[0]
    one = src.vszip.BoxBlur(hradius=7, vradius=0, vpasses=0).vszip.BoxBlur(hradius=7, vradius=0, vpasses=0).get_frame(0)[0]
    assert np.array_equal(two, one)


def test_bilateral_yuv420p16(oracle):
    planes = [fx.tiled_natural(s, np.uint16, p) for p, s in enumerate([(216, 384), (108, 192), (108, 192)])]
    out = vs.source([planes], vs.YUV420P16).vszip.Bilateral(sigmaS=2.0, sigmaR=2.0).get_frame(0)
    prm = oracle.bilateral_params([2], [2], yuv=True, ssw=1, ssh=1)
    for p in range(3):
        want = oracle.bilateral_plane(planes[p], prm["sigmaS"][p], prm["sigmaR"][p], prm["algorithm"][p], prm["radius"][p], prm["step"][p], prm["PBFICnum"][p])
        assert np.array_equal(out[p], want)


def test_bilateral_pbfic_gray16(oracle):
    """algorithm=1 through the plugin (reference tests/test_bilateral.py:22-23: sigmaS=3 sigmaR=0.1, PBFICnum 4 and 32)."""
    p = fx.tiled_natural((120, 200), np.uint16, 0)
    for num in (4, 32):
        out = vs.source([[p]], vs.GRAY16).vszip.Bilateral(sigmaS=3.0, sigmaR=0.1, algorithm=1, PBFICnum=num).get_frame(0)
        prm = oracle.bilateral_params([3], [0.1], algorithm=[1], pbficnum=[num])
        want = oracle.bilateral_plane(p, prm["sigmaS"][0], prm["sigmaR"][0], 1, prm["radius"][0], prm["step"][0], prm["PBFICnum"][0])
        assert np.array_equal(out[0], want)


def test_bilateral_sigma_zero_is_passthrough():
    """reference tests/test_bilateral.py:84-87"""
    planes = _yuv420p16(5, 128, 64)
    out = vs.source([planes], vs.YUV420P16).vszip.Bilateral(sigmaS=0.0).get_frame(0)
    for p in range(3):
        assert np.array_equal(out[p], planes[p])


def test_planeaverage_known_answers():
    """reference tests/test_planeaverage.py:118-147"""
    two = np.concatenate([np.full((32, 64), 1000, np.uint16), np.full((32, 64), 3000, np.uint16)], axis=1)
    src = vs.source([[two]], vs.GRAY16)
    assert src.vszip.PlaneAverage(exclude=[1000]).get_frame(0).props["psmAvg"] == 3000 / 65535
    assert src.vszip.PlaneAverage(exclude=[3000]).get_frame(0).props["psmAvg"] == 1000 / 65535
    assert src.vszip.PlaneAverage(exclude=[1000, 3000]).get_frame(0).props["psmAvg"] == 0.0
    y = vs.blank(vs.YUV420P16, 64, 32, [6777, 32768, 0])
    p = y.vszip.PlaneAverage(exclude=[-1], prop="avg_test").get_frame(0).props
    assert p["avg_testAvg"] == 0.10341039139391164
    multi = y.vszip.PlaneAverage(exclude=[-1], planes=[0, 1, 2]).get_frame(0).props["psmAvg"]
    assert multi == [6777 / 65535, 32768 / 65535, 0.0]
    f = vs.source([[np.concatenate([np.full((32, 64), 1.0, np.float32), np.full((32, 64), 3.0, np.float32)], axis=1)]], vs.GRAYS)
    assert f.vszip.PlaneAverage(exclude=[3]).get_frame(0).props["psmAvg"] == 1.0


def test_planeminmax_known_answers(oracle):
    """reference tests/test_planeminmax.py:99-110,228-236"""
    p = np.full((32, 64), 200, np.uint8)
    p[:8, :] = 0
    src = vs.source([[p]], vs.GRAY8)
    a = src.vszip.PlaneMinMax(minthr=0.2).get_frame(0).props
    b = src.vszip.PlaneMinMax(minthr=0.3).get_frame(0).props
    assert (a["psmMin"], a["psmMax"], b["psmMin"]) == (0, 200, 200)
    q = vs.blank(vs.GRAY16, 64, 32, 1234)
    assert q.vszip.PlaneMinMax(minthr=1.0).get_frame(0).props["psmMin"] == 65535
    assert q.vszip.PlaneMinMax(maxthr=1.0).get_frame(0).props["psmMax"] == 0
    g = fx.crop_gray8()
    pr = vs.source([[g]], vs.GRAY8).vszip.PlaneMinMax(minthr=0.1, maxthr=0.1, clipb=vs.source([[g]], vs.GRAY8).vszip.BoxBlur(hradius=1, vradius=1)).get_frame(0).props
    gold = fx.ref_goldens()["exact"]["planeminmax"]["GRAY8|full|maxthr=0.1,minthr=0.1"]
    assert (pr["psmMin"], pr["psmMax"]) == (gold["Min"], gold["Max"])
    assert pr["psmDiff"] == oracle.plane_minmax(g, 0.1, 0.1, oracle.boxblur(np.ascontiguousarray(g), 1, 1, 1, 1))[2]


def test_ssimulacra2_props():
    """reference tests/test_ssimulacra2.py:65-71 (linear RGBS fed directly: _Transfer = 8)."""
    c = vs.blank(vs.RGBS, 64, 64, [0.3, 0.2, 0.5])
    lin = vs.source([[np.full((64, 64), v, np.float32) for v in (0.3, 0.2, 0.5)]], vs.RGBS, props={"_Transfer": 8})
    assert lin.vszip.SSIMULACRA2(lin).get_frame(0).props["SSIMULACRA2"] == 100.0
    with pytest.raises(vs.Error, match="linear-light RGBS"):
        c.vszip.SSIMULACRA2(c)


def test_ssimulacra2_matches_oracle(oracle):
    rng = np.random.default_rng(2)
    ref = [np.ascontiguousarray(p) for p in fx.crop_rgbs()]
    dis = [np.clip(p + rng.normal(0, 0.03, p.shape).astype(np.float32), 0, 1) for p in ref]
    a = vs.source([ref], vs.RGBS, props={"_Transfer": 8})
    b = vs.source([dis], vs.RGBS, props={"_Transfer": 8})
    out = a.vszip.SSIMULACRA2(b).get_frame(0)
    assert out.props["SSIMULACRA2"] == pytest.approx(oracle.ssimulacra2(ref, dis), abs=1e-7)
    assert np.array_equal(out[0], ref[0])  # the output clip is the reference clip + prop


def test_xpsnr_props(oracle):
    """identical -> +inf (reference tests/test_xpsnr.py:222-225); a distorted clip matches the oracle."""
    frames = []
    rng = np.random.default_rng(0)
    for f in range(3):
        frames.append([np.roll(fx.tiled_natural((288, 352), np.uint8), f, 0), fx.tiled_natural((144, 176), np.uint8, 1), fx.tiled_natural((144, 176), np.uint8, 2)])
    dist = [[np.clip(p.astype(np.int16) + rng.integers(-2, 3, p.shape), 0, 255).astype(np.uint8) for p in fr] for fr in frames]
    a, b = vs.source(frames, vs.YUV420P8), vs.source(dist, vs.YUV420P8)
    same = a.vszip.XPSNR(a, verbose=0).get_frame(1).props
    assert math.isinf(same["XPSNR_Y"]) and math.isinf(same["XPSNR_U"])
    got = a.vszip.XPSNR(b, verbose=0).get_frame(2).props
    w = oracle.xpsnr_wsse(frames[2], dist[2], frames[1][0], None, depth=8, frame_rate=24, temporal=True)
    assert got["XPSNR_Y"] == oracle.xpsnr_frame(w[0], 352, 288, 8)
    assert got["XPSNR_V"] == oracle.xpsnr_frame(w[2], 176, 144, 8)


def test_eedi3_through_plugin(oracle):
    planes = [np.ascontiguousarray(fx.crop_rgbs()[0][:96, :200]), np.ascontiguousarray(fx.crop_rgbs()[1][:48, :100]), np.ascontiguousarray(fx.crop_rgbs()[2][:48, :100])]
    src = vs.source([planes] * 2, vs.YUV420PS)
    out = src.vszip.EEDI3(field=1, dh=1).get_frame(0)
    for p in range(3):
        assert np.array_equal(out[p], oracle.eedi3(planes[p], 1, dh=True))
    assert out.props["_FieldBased"] == 0
    # _FieldBased on the source frame overrides `field` (src/vapoursynth/eedi3.zig:166-172): 1 = bottom -> field 0
    bff = vs.source([planes], vs.YUV420PS, props={"_FieldBased": 1}).vszip.EEDI3(field=1).get_frame(0)
    assert np.array_equal(bff[0], oracle.eedi3(planes[0], 0))
    # double rate: frame 1 takes the opposite field
    dbl = src.vszip.EEDI3(field=3)
    assert np.array_equal(dbl.get_frame(0)[0], oracle.eedi3(planes[0], 1)) and np.array_equal(dbl.get_frame(1)[0], oracle.eedi3(planes[0], 0))
    h = vs.source([[planes[0]]], vs.GRAYS).vszip.EEDI3H(field=1).get_frame(0)
    assert np.array_equal(h[0], oracle.eedi3(planes[0], 1, horizontal=True))
    # (round 6: only the interpolated lines come down the link, the kept field's lines are copied from the source frame on the host) every plane, both
    # fields, with and without dh; EEDI3H with dh downloads whole planes
    for field in (0, 1):
        for dh in (False, True):
            got = src.vszip.EEDI3(field=field, dh=dh).get_frame(1)
            for p in range(3):
                assert np.array_equal(got[p], oracle.eedi3(planes[p], field, dh=dh)), (field, dh, p)
    hd = vs.source([[planes[0]]], vs.GRAYS).vszip.EEDI3H(field=0, dh=True).get_frame(0)
    assert np.array_equal(hd[0], oracle.eedi3(planes[0], 0, dh=True, horizontal=True))


def test_eedi3_hp_mdis40_mclip_through_plugin(oracle):
    """The rest of the EEDI3 signature: hp=1, mdis up to 40, and a Gray8 mclip that drives every
    plane of a YUV clip (its top-left region for the subsampled planes, src/vapoursynth/eedi3.zig:215-218)."""
    planes = [np.ascontiguousarray(fx.crop_rgbs()[0][:96, :240]), np.ascontiguousarray(fx.crop_rgbs()[1][:48, :120]), np.ascontiguousarray(fx.crop_rgbs()[2][:48, :120])]
    src = vs.source([planes], vs.YUV420PS)
    hp = src.vszip.EEDI3(field=1, hp=1).get_frame(0)
    wide = src.vszip.EEDI3(field=0, mdis=40, nrad=3).get_frame(0)
    for p in range(3):
        assert np.array_equal(hp[p], oracle.eedi3(planes[p], 1, hp=True))
        assert np.array_equal(wide[p], oracle.eedi3(planes[p], 0, mdis=40, nrad=3))
    rng = np.random.default_rng(11)
    mask = (rng.random(planes[0].shape) < 0.02).astype(np.uint8) * 255
    mask[30:40] = 0
    m = vs.source([[mask]], vs.GRAY8)
    out = src.vszip.EEDI3(field=1, mclip=m).get_frame(0)
    for p in range(3):
        h, w = planes[p].shape
        assert np.array_equal(out[p], oracle.eedi3(planes[p], 1, mclip=np.ascontiguousarray(mask[:h, :w])))
    with pytest.raises(vs.Error, match="mclip's dimensions don't match"):
        src.vszip.EEDI3(field=1, mclip=vs.source([[mask[:50]]], vs.GRAY8))


def test_concurrent_get_frame_threads(oracle):
    """fmParallel: VapourSynth calls getFrame from several worker threads at once. Eight threads pull
    frames of a BoxBlur and a Bilateral clip concurrently (one context + slab per (thread, GPU) in
    the plugin); every frame must equal the oracle's."""
    import threading

    frames = [[fx.splitmix64_plane(100 + 3 * n + p, s, np.uint16) for p, s in enumerate([(120, 200), (60, 100), (60, 100)])] for n in range(12)]
    src = vs.source(frames, vs.YUV420P16)
    blur = src.vszip.BoxBlur(hradius=5, vradius=5)
    bil = src.vszip.Bilateral(sigmaS=1.0, sigmaR=0.05)
    prm = oracle.bilateral_params([1.0], [0.05], yuv=True, ssw=1, ssh=1)
    errors = []

    def worker(tid):
        try:
            for rep in range(3):
                for n in range(tid % 4, 12, 4):
                    out = blur.get_frame(n)
                    for p in range(3):
                        if not np.array_equal(out[p], oracle.boxblur(frames[n][p], 5, 1, 5, 1)):
                            errors.append(("boxblur", tid, n, p))
                    out = bil.get_frame(n)
                    for p in range(3):
                        want = oracle.bilateral_plane(frames[n][p], prm["sigmaS"][p], prm["sigmaR"][p], prm["algorithm"][p], prm["radius"][p], prm["step"][p], prm["PBFICnum"][p])
                        if not np.array_equal(out[p], want):
                            errors.append(("bilateral", tid, n, p))
        except Exception as e:  # noqa: BLE001
            errors.append(("exception", tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_second_clips_through_plugin(oracle):
    """The optional second clips of the signatures: PlaneAverage clipb (+ a 32-bit integer clip),
    Bilateral ref, EEDI3 sclip."""
    a = fx.tiled_natural((96, 160), np.uint16, 0)
    b = fx.tiled_natural((96, 160), np.uint16, 2)
    ca, cb = vs.source([[a]], vs.GRAY16), vs.source([[b]], vs.GRAY16)
    pr = ca.vszip.PlaneAverage(exclude=[-1], clipb=cb).get_frame(0).props
    oa, od = oracle.plane_average(a, ref=b)
    assert (pr["psmAvg"], pr["psmDiff"]) == (oa, od)
    u32 = (a.astype(np.uint32) << 16) | b
    p32 = vs.source([[u32]], vs.GRAY32).vszip.PlaneAverage(exclude=[]).get_frame(0).props
    assert p32["psmAvg"] == oracle.plane_average(u32)[0]
    with pytest.raises(vs.Error, match="exclude is not supported for 32-bit integer"):
        vs.source([[u32]], vs.GRAY32).vszip.PlaneAverage(exclude=[0])
    out = ca.vszip.Bilateral(ref=cb, sigmaS=2.0, sigmaR=0.05).get_frame(0)
    prm = oracle.bilateral_params([2.0], [0.05])
    want = oracle.bilateral_plane(a, prm["sigmaS"][0], prm["sigmaR"][0], prm["algorithm"][0], prm["radius"][0], prm["step"][0], prm["PBFICnum"][0], ref=b)
    assert np.array_equal(out[0], want)
    f = np.ascontiguousarray(fx.crop_rgbs()[1][:64, :200])
    sc = np.ascontiguousarray(fx.crop_rgbs()[2][:128, :200])
    e = vs.source([[f]], vs.GRAYS).vszip.EEDI3(field=1, dh=1, sclip=vs.source([[sc]], vs.GRAYS)).get_frame(0)
    assert np.array_equal(e[0], oracle.eedi3(f, 1, dh=True, sclip=sc))
    # (round 6: only the sclip's interpolated lines are uploaded) the other field, without dh, every vcheck level, and EEDI3H (whole sclip, transposed on the device)
    for field, dh, vcheck in ((0, True, 2), (0, False, 1), (1, False, 3)):
        scl = sc if dh else sc[:64]
        e = vs.source([[f]], vs.GRAYS).vszip.EEDI3(field=field, dh=dh, vcheck=vcheck, sclip=vs.source([[scl]], vs.GRAYS)).get_frame(0)
        assert np.array_equal(e[0], oracle.eedi3(f, field, dh=dh, vcheck=vcheck, sclip=scl)), (field, dh, vcheck)
    sch = np.ascontiguousarray(fx.crop_rgbs()[2][:64, :400])
    eh = vs.source([[f]], vs.GRAYS).vszip.EEDI3H(field=0, dh=1, sclip=vs.source([[sch]], vs.GRAYS)).get_frame(0)
    assert np.array_equal(eh[0], oracle.eedi3(f, 0, dh=True, sclip=sch, horizontal=True))


def test_depth_conversions_are_delegated_to_the_host(oracle):
    """mclip that is not 8-bit Gray: std.SetFrameProps(_Range=1) + resize.Point(format=Gray8)
    (src/vapoursynth/eedi3.zig:411-432); XPSNR clips of different depth: resize.Point with
    dither_type=none on the shallower one (src/helper.zig:470-494, xpsnr.zig:163-169). The host
    functions are the test host's stand-ins; what is checked is the delegation (calls, arguments,
    ownership) and that the filter then runs on the converted clip."""
    f = np.ascontiguousarray(fx.crop_rgbs()[1][:64, :200])
    mask16 = np.zeros((64, 200), np.uint16)
    mask16[10:30, 50:120] = 40000
    mask16[40:50, 10:40] = 90
    src = vs.source([[f]], vs.GRAYS)
    with pytest.raises(vs.Error, match="needs the host's std and resize plugins"):
        src.vszip.EEDI3(field=1, mclip=vs.source([[mask16]], vs.GRAY16))
    y8 = fx.tiled_natural((96, 160), np.uint8, 0)
    c8 = [fx.tiled_natural((48, 80), np.uint8, 1 + i) for i in range(2)]
    ref8 = [[y8] + c8] * 2
    dis10 = [[np.clip(p.astype(np.int32) * 4 + 5, 0, 1023).astype(np.uint16) for p in fr] for fr in ref8]
    with pytest.raises(vs.Error, match="different bit depth need the host's resize plugin"):
        vs.source(ref8, vs.YUV420P8).vszip.XPSNR(vs.source(dis10, vs.YUV420P10))
    vs.core_standins(True)
    try:
        got = src.vszip.EEDI3(field=1, mclip=vs.source([[mask16]], vs.GRAY16)).get_frame(0)
        assert vs.standin_log() == ["std.SetFrameProps _Range=1", f"resize.Point format={vs.GRAY8}"]
        mask8 = np.floor(mask16.astype(np.float64) * 255 / 65535 + 0.5).astype(np.uint8)
        assert np.array_equal(got[0], oracle.eedi3(f, 1, mclip=mask8))
        assert not np.array_equal(got[0], oracle.eedi3(f, 1))  # the mask does gate something
        vs.core_standins(True)  # clears the log
        clip = vs.source(ref8, vs.YUV420P8).vszip.XPSNR(vs.source(dis10, vs.YUV420P10), verbose=0)
        assert vs.standin_log() == [f"resize.Point format={vs.YUV420P10} dither_type=none"]
        props = clip.get_frame(1).props
        ref10 = [(p.astype(np.uint16) << 2) for p in ref8[0]]
        w = oracle.xpsnr_wsse(ref10, dis10[0], ref10[0], None, depth=10, frame_rate=24, temporal=True)
        assert props["XPSNR_Y"] == oracle.xpsnr_frame(w[0], 160, 96, 10)
        assert props["XPSNR_U"] == oracle.xpsnr_frame(w[1], 80, 48, 10)
    finally:
        vs.core_standins(False)


def test_limiter_through_plugin(oracle):
    """reference tests/test_limiter.py: tv_range tables per format, mask, float default, integer default
    no-op, tv_range == explicit min/max, planes subset, explicit min/max pixels, 32-bit tables."""
    def stacked(fmt, dtype, peak, lo_val=0, chroma=False):
        h, w = 32, 64
        ssw, ssh = (fmt >> 8) & 0xFF, fmt & 0xFF
        npl = 1 if (fmt >> 28) == vs.GRAY else 3
        planes = []
        for p in range(npl):
            ph, pw = h >> (ssh if p else 0), w >> (ssw if p else 0)
            a = np.empty((ph, pw), dtype)
            a[: ph // 2] = peak if not (chroma and p) else 0.5
            a[ph // 2:] = lo_val if not (chroma and p) else -0.5
            planes.append(a)
        return vs.source([planes], fmt), planes

    def mm(clip):
        f = clip.get_frame(0)
        return [np.asarray(f[p]).min().item() for p in range(len(f))], [np.asarray(f[p]).max().item() for p in range(len(f))]

    for fmt, dtype, peak, lo, hi in ((vs.YUV420P8, np.uint8, 255, [16, 16, 16], [235, 240, 240]), (vs.YUV420P10, np.uint16, 1023, [64, 64, 64], [940, 960, 960]),
                                     (vs.YUV420P16, np.uint16, 65535, [4096] * 3, [60160, 61440, 61440]), (vs.RGB24, np.uint8, 255, [16] * 3, [235] * 3)):
        src, planes = stacked(fmt, dtype, peak)
        assert mm(src.vszip.Limiter(tv_range=True)) == (lo, hi), fmt
        explicit = src.vszip.Limiter(min=lo, max=hi).get_frame(0)  # the comptime and the runtime path agree
        tv = src.vszip.Limiter(tv_range=True).get_frame(0)
        for p in range(3):
            assert np.array_equal(explicit[p], tv[p])
        out = src.vszip.Limiter().get_frame(0)  # integer default: full range, a no-op
        for p in range(3):
            assert np.array_equal(out[p], planes[p])
    srcf, _ = stacked(vs.YUV420PS, np.float32, 2.0, -2.0)
    assert mm(srcf.vszip.Limiter()) == ([0.0, -0.5, -0.5], [1.0, 0.5, 0.5])
    assert mm(srcf.vszip.Limiter(tv_range=True, mask=True)) == ([0.0] * 3, [1.0] * 3)
    src16, planes16 = stacked(vs.YUV420P16, np.uint16, 65535)
    part = src16.vszip.Limiter(tv_range=True, planes=[0]).get_frame(0)
    assert np.array_equal(part[1], planes16[1]) and np.array_equal(part[2], planes16[2]) and np.asarray(part[0]).min() == 4096
    ramp = np.tile(np.arange(256, dtype=np.uint8), (2, 1))
    assert np.asarray(vs.source([[ramp]], vs.GRAY8).vszip.Limiter(min=[10], max=[200]).get_frame(0)[0])[0].tolist() == [min(max(x, 10), 200) for x in range(256)]
    src32, _ = stacked(vs.GRAY32, np.uint32, 4294967295)
    assert mm(src32.vszip.Limiter()) == ([0], [4294967295]) and mm(src32.vszip.Limiter(tv_range=True)) == ([268435456], [3942645760])
    nat = [fx.tiled_natural((96, 160), np.uint16, p) for p in range(3)]
    got = vs.source([nat], vs.YUV444P16).vszip.Limiter(min=[10000, 20000, 10000], max=[50000, 55000, 45000]).get_frame(0)
    for p, (lo, hi) in enumerate(zip([10000, 20000, 10000], [50000, 55000, 45000])):
        assert np.array_equal(got[p], oracle.limiter(nat[p], lo, hi))


def test_limit_filter_through_plugin(oracle):
    """flt = BoxBlur(2,2) of src like the reference's own cases; thresholds arrive on the 8-bit scale
    and are carried to the clip's depth / range by the wrapper (hz.scaleValue): identity at 8 bit, full
    range for RGB, limited for YUV without a range prop; WITH a range prop the reference's build resolves the
    opposite range (its 38 integer LimitFilter goldens on zimg's limited-flagged clips carry full-range thresholds:
    tests/test_oracle_zimg_goldens.py::test_limit_filter_keys), mirrored by the plugin; per-plane arrays; the
    optional ref clip; planes subset copies from flt."""
    src8 = list(fx.crop_rgb24()[:, :96, :160])
    clip8 = vs.source([src8], vs.RGB24)
    flt8 = clip8.vszip.BoxBlur(hradius=2, vradius=2)
    out = flt8.vszip.LimitFilter(src=clip8, dark_thr=8, bright_thr=8, elast=3).get_frame(0)
    for p in range(3):
        want = oracle.limit_filter(oracle.boxblur(np.ascontiguousarray(src8[p]), 2, 1, 2, 1), src8[p], None, 8, 8, 3)
        assert np.array_equal(out[p], want)
    nat = [fx.tiled_natural(s, np.uint16, p) for p, s in enumerate([(96, 160), (48, 80), (48, 80)])]
    for props, limited in ((None, True), ({"_ColorRange": 0}, True), ({"_ColorRange": 1}, False), ({"_Range": 0}, True), ({"_Range": 1}, True)):
        c16 = vs.source([nat], vs.YUV420P16, props=props)
        f16 = c16.vszip.BoxBlur(hradius=2, vradius=2)
        r16 = c16.vszip.BoxBlur(hradius=4, vradius=4)
        got = f16.vszip.LimitFilter(src=c16, ref=r16, dark_thr=[16, 4], bright_thr=[8, 2], elast=[4, 2]).get_frame(0)
        for p in range(3):
            dk = oracle.scale_value_from_8bit([16, 4, 4][p], False, 16, limited)
            br = oracle.scale_value_from_8bit([8, 2, 2][p], False, 16, limited)
            fl = oracle.boxblur(nat[p], 2, 1, 2, 1)
            rf = oracle.boxblur(nat[p], 4, 1, 4, 1)
            assert np.array_equal(got[p], oracle.limit_filter(fl, nat[p], rf, dk, br, [4, 2, 2][p])), (props, p)
    c16 = vs.source([nat], vs.YUV420P16)
    f16 = c16.vszip.BoxBlur(hradius=2, vradius=2)
    part = f16.vszip.LimitFilter(src=c16, dark_thr=8, bright_thr=8, planes=[0]).get_frame(0)
    flt_frame = f16.get_frame(0)
    assert np.array_equal(part[1], flt_frame[1]) and np.array_equal(part[2], flt_frame[2]) and not np.array_equal(part[0], flt_frame[0])
    srcf = [np.ascontiguousarray(p[:64, :128]) for p in fx.crop_rgbs()]
    cf = vs.source([srcf], vs.RGBS)
    ff = cf.vszip.BoxBlur(hradius=2, vradius=2)
    gotf = ff.vszip.LimitFilter(src=cf, dark_thr=8, bright_thr=8, elast=3).get_frame(0)
    t = oracle.scale_value_from_8bit(8, True, 32, False)
    for p in range(3):
        assert np.array_equal(gotf[p], oracle.limit_filter(oracle.boxblur(srcf[p], 2, 1, 2, 1), srcf[p], None, t, t, 3))


def test_adaptive_binarize_through_plugin(oracle):
    """255 where clip2 - clip >= c else 0 on every plane, c clamped to [-256, 256], output tagged full
    range (reference tests/test_adaptive_binarize.py:58-70); clip2 here is vszip.BoxBlur(5, 5)."""
    planes = [fx.tiled_natural(s, np.uint8, p) for p, s in enumerate([(98, 162), (49, 81), (49, 81)])]
    src = vs.source([planes], vs.YUV420P8)
    blur = src.vszip.BoxBlur(hradius=5, vradius=5)
    bf = blur.get_frame(0)
    for c in (3, 0, 12, -5, 1000, -1000):
        out = src.vszip.AdaptiveBinarize(clip2=blur, c=c).get_frame(0)
        for p in range(3):
            assert np.array_equal(out[p], oracle.adaptive_binarize(planes[p], bf[p], c)), (c, p)
            assert set(np.unique(out[p]).tolist()) <= {0, 255}
        assert out.props["_ColorRange"] == 0
    assert np.all(src.vszip.AdaptiveBinarize(clip2=blur, c=1000).get_frame(0)[0] == 0)
    assert np.all(src.vszip.AdaptiveBinarize(clip2=blur, c=-1000).get_frame(0)[0] == 255)


def test_xpsnr_reference_behaviours(oracle):
    """reference tests/test_xpsnr.py: the temporal order switches at exactly 32 fps (24 == 31; 32 agrees
    on frame 0 only) :178-196; a width whose stride exceeds it :352-361; a clip so small that the block
    size is 0 :364-369; the output frame is the distorted frame :254-257; verbose changes nothing :266-273."""
    rng = np.random.default_rng(8)

    def clip(w, h, n, fps, seed):
        r = np.random.default_rng(seed)
        shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
        frames = [[np.roll(fx.tiled_natural(s, np.uint8, p), 3 * f, axis=1) for p, s in enumerate(shapes)] for f in range(n)]
        dist = [[np.clip(p.astype(np.int16) + r.integers(-4, 5, p.shape), 0, 255).astype(np.uint8) for p in fr] for fr in frames]
        return frames, dist, vs.source(frames, vs.YUV420P8, fps=(fps, 1)), vs.source(dist, vs.YUV420P8, fps=(fps, 1))

    def ys(fps):
        fr, ds, a, b = clip(640, 360, 5, fps, 1)
        out = a.vszip.XPSNR(b, verbose=0)
        got = [out.get_frame(n).props["XPSNR_Y"] for n in range(5)]
        for n in range(5):  # and each equals the oracle with the temporal order the fps selects
            w = oracle.xpsnr_wsse(fr[n], ds[n], fr[n - 1][0] if n >= 1 else None, fr[n - 2][0] if n >= 2 else None, depth=8, frame_rate=fps, temporal=True)
            assert got[n] == oracle.xpsnr_frame(w[0], 640, 360, 8)
        return got

    s24, s31, s32 = ys(24), ys(31), ys(32)
    assert s24 == s31 and s32[0] == s31[0] and all(s32[n] != s31[n] for n in range(1, 5))
    for w, h in ((100, 80), (32, 32)):
        fr, ds, a, b = clip(w, h, 3, 24, w)
        out = a.vszip.XPSNR(b, temporal=1, verbose=0)
        quiet = [out.get_frame(n) for n in range(3)]
        loud = [a.vszip.XPSNR(b, temporal=1).get_frame(n) for n in range(3)]
        for n in range(3):
            assert math.isfinite(quiet[n].props["XPSNR_Y"]) and quiet[n].props == loud[n].props
            for p in range(3):
                assert np.array_equal(quiet[n][p], ds[n][p])  # the distorted frame, passed through
            wsse = oracle.xpsnr_wsse(fr[n], ds[n], fr[n - 1][0] if n >= 1 else None, None, depth=8, frame_rate=24, temporal=True)
            assert quiet[n].props["XPSNR_U"] == oracle.xpsnr_frame(wsse[1], w // 2, h // 2, 8)


def test_ssimulacra2_monotone_and_output_clip(oracle):
    """reference tests/test_ssimulacra2.py:79-100: more blur scores lower; the output clip is the RGBS reference."""
    ref = [np.ascontiguousarray(p[:128, :192]) for p in fx.crop_rgbs()]
    a = vs.source([ref], vs.RGBS, props={"_Transfer": 8})
    s = []
    for r in (0, 1, 3):
        b = a if r == 0 else vs.source([[oracle.boxblur(p, r, 1, r, 1) for p in ref]], vs.RGBS, props={"_Transfer": 8})
        out = a.vszip.SSIMULACRA2(b)
        f = out.get_frame(0)
        s.append(f.props["SSIMULACRA2"])
        assert out.format_id == vs.RGBS and (out.width, out.height) == (192, 128)
        for p in range(3):
            assert np.array_equal(f[p], ref[p])
    assert s[0] > s[1] > s[2] and s[0] > 99.9


def test_chained_filters_like_a_script(oracle):
    """BASELINE config 5 written the way a .vpy script chains it — Bilateral -> BoxBlur ->
    SSIMULACRA2(source, processed) — over a 3-frame clip: every frame's pixels equal the oracle's
    chain bit for bit and every frame carries its own score."""
    base = [np.ascontiguousarray(p) for p in fx.crop_rgbs()]
    frames = [[np.ascontiguousarray(np.roll(p, 37 * f, axis=1)) for p in base] for f in range(3)]
    src = vs.source(frames, vs.RGBS, props={"_Transfer": 8})
    proc = src.vszip.Bilateral(sigmaS=2.0, sigmaR=2.0).vszip.BoxBlur(hradius=2, vradius=2)
    scored = src.vszip.SSIMULACRA2(proc)
    prm = oracle.bilateral_params([2], [2], yuv=False, ssw=0, ssh=0)
    scores = []
    for n in (2, 0, 1):  # out of order, like a seeking host
        want = []
        for i, p in enumerate(frames[n]):
            b = oracle.bilateral_plane(p, prm["sigmaS"][i], prm["sigmaR"][i], prm["algorithm"][i], prm["radius"][i], prm["step"][i], prm["PBFICnum"][i])
            want.append(oracle.boxblur(b, 2, 1, 2, 1))
        got = proc.get_frame(n)
        for i in range(3):
            assert np.array_equal(got[i], want[i])
        assert got.props["_Transfer"] == 8
        s = scored.get_frame(n).props["SSIMULACRA2"]
        assert s == pytest.approx(oracle.ssimulacra2(frames[n], want), abs=1e-7)
        scores.append(s)
    # a horizontal roll only moves the content across tile borders: the scores stay close, not equal
    assert max(scores) - min(scores) < 5.0 and len(set(scores)) == 3


def test_heavy_filters_switch_to_the_pinned_arena_under_load(oracle):
    """Bilateral and EEDI3 stage through the context's pinned arena once four getFrame calls are
    in flight (HeavyFrameScope); BoxBlur on the same worker threads keeps the direct copies. The
    pixels must not depend on the staging mode: 16 workers pull a Bilateral -> BoxBlur chain and an
    EEDI3 clip, and every frame equals the single-caller result."""
    planes = [fx.tiled_natural(s, np.uint16, p) for p, s in enumerate([(216, 384), (108, 192), (108, 192)])]
    frames = [[np.ascontiguousarray(np.roll(p, 3 * f, axis=1)) for p in planes] for f in range(12)]
    chain = vs.source(frames, vs.YUV420P16).vszip.Bilateral(sigmaS=2.0, sigmaR=2.0).vszip.BoxBlur(hradius=3, vradius=3)
    single = [[np.array(chain.get_frame(n)[p]) for p in range(3)] for n in range(12)]
    chain.pull(96, 16, warm_per_thread=1)
    import concurrent.futures as cf

    with cf.ThreadPoolExecutor(16) as ex:
        got = list(ex.map(lambda n: [np.array(chain.get_frame(n % 12)[p]) for p in range(3)], range(64)))
    for n, fr in enumerate(got):
        for p in range(3):
            assert np.array_equal(fr[p], single[n % 12][p])
    g = [np.ascontiguousarray(np.roll(fx.tiled_natural((120, 200), np.float32, 0), 5 * f, axis=1)) for f in range(8)]
    e3 = vs.source([[x] for x in g], vs.GRAYS).vszip.EEDI3(field=1, dh=True)
    want = [oracle.eedi3(x, 1, dh=True) for x in g]
    with cf.ThreadPoolExecutor(16) as ex:
        got = list(ex.map(lambda n: np.array(e3.get_frame(n % 8)[0]), range(48)))
    for n, o in enumerate(got):
        assert np.array_equal(o, want[n % 8])


def test_gate_with_fewer_slots_than_workers():
    """VSZIP_MAX_IN_FLIGHT=2 with 16 workers on a Bilateral -> BoxBlur chain plus a metric: every
    worker gets its turn (no deadlock, no failed frame), results equal the ungated single caller's.
    Runs in a child process because the limit is read once per process."""
    import os
    import subprocess
    import sys

    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
import fixtures as fx
from fakevs import fakevs as vs
planes = [fx.tiled_natural(s, np.uint16, p) for p, s in enumerate([(144, 256), (72, 128), (72, 128)])]
frames = [[np.ascontiguousarray(np.roll(p, 3 * f, axis=1)) for p in planes] for f in range(8)]
src = vs.source(frames, vs.YUV420P16)
chain = src.vszip.Bilateral(sigmaS=2.0, sigmaR=2.0).vszip.BoxBlur(hradius=3, vradius=3).vszip.PlaneAverage(exclude=[-1])
single = [(np.array(chain.get_frame(n)[0]), chain.get_frame(n).props["psmAvg"]) for n in range(8)]
chain.pull(128, 16)
import concurrent.futures as cf
with cf.ThreadPoolExecutor(16) as ex:
    got = list(ex.map(lambda n: (np.array(chain.get_frame(n % 8)[0]), chain.get_frame(n % 8).props["psmAvg"]), range(64)))
for n, (px, avg) in enumerate(got):
    assert np.array_equal(px, single[n % 8][0]) and avg == single[n % 8][1]
print("gate ok")
'''
    env = dict(os.environ, VSZIP_MAX_IN_FLIGHT="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gate ok" in r.stdout, r.stdout + r.stderr


def _ssim_prestage_case(oracle):
    from oracle import vs_host as vh

    ref8 = [np.ascontiguousarray(p[:160, :256]) for p in fx.crop_rgb24()]
    dis8 = [vh.std_boxblur(p, 1, 1) for p in ref8]
    g8 = np.ascontiguousarray(fx.crop_gray8()[:160, :256])
    want_rgb = oracle.ssimulacra2(vh.to_linear_rgbs(ref8, "RGB", 8), vh.to_linear_rgbs(dis8, "RGB", 8))
    want_gray = oracle.ssimulacra2(vh.to_linear_rgbs([g8], "GRAY", 8), vh.to_linear_rgbs([vh.std_boxblur(g8, 3, 3)], "GRAY", 8))
    return ref8, dis8, g8, want_rgb, want_gray


def test_ssimulacra2_device_colour_prestage(oracle):
    """RGB24 / Gray8 / gamma RGBS clips go to the GPU as they are; hz.toRGBS + sRGBtoLinearRGB
    (src/helper.zig:225-243, src/vapoursynth/ssimulacra2.zig:132-162) run fused into the first SSIMULACRA2 pass.
    The OUTPUT clip is still the host-converted reference (ssimulacra2.zig:53), here through the test host's
    resize / std stand-ins (the same restatement as oracle/vs_host.py)."""
    from oracle import vs_host as vh

    ref8, dis8, g8, want_rgb, want_gray = _ssim_prestage_case(oracle)
    vs.core_standins(True)
    try:
        a, b = vs.source([ref8], vs.RGB24), vs.source([dis8], vs.RGB24)
        out = a.vszip.SSIMULACRA2(b)
        f = out.get_frame(0)
        assert f.props["SSIMULACRA2"] == pytest.approx(want_rgb, abs=1e-7)
        assert out.format_id == vs.RGBS
        lin = vh.to_linear_rgbs(ref8, "RGB", 8)
        for p in range(3):
            assert np.array_equal(np.asarray(f[p]).view(np.uint32), lin[p].view(np.uint32))
        # only the reference went through the host's resize (format + transfer); the distorted clip never did
        log = vs.standin_log()
        assert sum("resize.Bicubic" in l for l in log) == 2, log
        # Gray: limited range, R = G = B
        ga, gb = vs.source([[g8]], vs.GRAY8), vs.source([[vh.std_boxblur(g8, 3, 3)]], vs.GRAY8)
        assert ga.vszip.SSIMULACRA2(gb).get_frame(0).props["SSIMULACRA2"] == pytest.approx(want_gray, abs=1e-7)
        # mixed formats: RGB24 reference against a gamma RGBS distorted clip (converted on the device one by one)
        dis_f = [(p.astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float32) for p in dis8]
        c = vs.source([dis_f], vs.RGBS)
        assert a.vszip.SSIMULACRA2(c).get_frame(0).props["SSIMULACRA2"] == pytest.approx(want_rgb, abs=1e-7)
        # a distorted clip that is already linear needs no pre-stage at all
        d = vs.source([vh.to_linear_rgbs(dis8, "RGB", 8)], vs.RGBS, props={"_Transfer": 8})
        assert a.vszip.SSIMULACRA2(d).get_frame(0).props["SSIMULACRA2"] == pytest.approx(want_rgb, abs=1e-7)
    finally:
        vs.core_standins(False)


def test_ssimulacra2_host_colour_switch(oracle, tmp_path):
    """VSZIP_SSIM_HOST_COLOR=1 keeps the whole conversion on the host (the round-1 path): same score."""
    import subprocess
    import sys
    import textwrap

    ref8, dis8, _, want_rgb, _ = _ssim_prestage_case(oracle)
    np.save(tmp_path / "ref.npy", np.stack(ref8))
    np.save(tmp_path / "dis.npy", np.stack(dis8))
    code = textwrap.dedent(f"""
        import sys, numpy as np
        sys.path.insert(0, {str(vs.ROOT)!r}); sys.path.insert(0, {str(vs.ROOT / 'tests')!r})
        from fakevs import fakevs as vs
        vs.core_standins(True)
        ref, dis = np.load({str(tmp_path / 'ref.npy')!r}), np.load({str(tmp_path / 'dis.npy')!r})
        a, b = vs.source([list(ref)], vs.RGB24), vs.source([list(dis)], vs.RGB24)
        print(repr(a.vszip.SSIMULACRA2(b).get_frame(0).props["SSIMULACRA2"]), sum("resize.Bicubic" in l for l in vs.standin_log()))
    """)
    import os

    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, VSZIP_SSIM_HOST_COLOR="1"), timeout=600)
    assert r.returncode == 0, r.stderr
    score, nconv = r.stdout.strip().split()
    assert float(score) == pytest.approx(want_rgb, abs=1e-7) and int(nconv) == 4  # both clips: format + transfer on the host


def _materialize(clip, fmt, n):
    """The clip's frames as a fresh source clip: a non-vszip node in between, i.e. the UNFUSED path."""
    out = []
    for i in range(n):
        fr = clip.get_frame(i)
        out.append([np.array(fr[p]) for p in range(3)])
    return vs.source(out, fmt)


def test_fused_chain_equals_unfused(oracle):
    """SURVEY 8f rank 4, second half: chained vszip filters run as ONE getFrame — one upload of the root frame,
    every stage's kernels on the device, one download. The fused result is bit-identical to the same chain
    with every intermediate clip taken through host memory, including `planes` subsets (a plane an upstream
    stage wrote and the last stage does not process must come back from the device, not from the root frame)."""
    frames = [[fx.splitmix64_plane(10 * f + p, s, np.uint16) for p, s in enumerate([(120, 208), (60, 104), (60, 104)])] for f in range(3)]
    src = vs.source(frames, vs.YUV420P16)
    f0, s0 = vs.fusion_stats()
    # fused: Bilateral (all planes) -> BoxBlur (luma only) -> Limiter (chroma only)
    fused = src.vszip.Bilateral(sigmaS=2.0, sigmaR=0.05).vszip.BoxBlur(hradius=3, vradius=3, planes=[0]).vszip.Limiter(min=[0, 20000, 20000], max=[65535, 40000, 40000], planes=[1, 2])
    got = []
    for n in (1, 0, 2):
        fr = fused.get_frame(n)
        got.append([np.array(fr[p]) for p in range(3)])
    f1, s1 = vs.fusion_stats()
    assert (f1 - f0, s1 - s0) == (3, 6)  # three getFrame calls, two upstream stages each
    # unfused twin: every hop through host memory
    a = _materialize(src.vszip.Bilateral(sigmaS=2.0, sigmaR=0.05), vs.YUV420P16, 3)
    b = _materialize(a.vszip.BoxBlur(hradius=3, vradius=3, planes=[0]), vs.YUV420P16, 3)
    c = b.vszip.Limiter(min=[0, 20000, 20000], max=[65535, 40000, 40000], planes=[1, 2])
    assert vs.fusion_stats() == (f1, s1)  # nothing fused here
    for k, n in enumerate((1, 0, 2)):
        want = c.get_frame(n)
        for p in range(3):
            assert np.array_equal(got[k][p], want[p]), (n, p)
    # and against the oracle, stage by stage
    prm = oracle.bilateral_params([2], [0.05], yuv=True, ssw=1, ssh=1)
    for k, n in enumerate((1, 0, 2)):
        bl = [oracle.bilateral_plane(p, prm["sigmaS"][i], prm["sigmaR"][i], prm["algorithm"][i], prm["radius"][i], prm["step"][i], prm["PBFICnum"][i]) for i, p in enumerate(frames[n])]
        want = [oracle.boxblur(bl[0], 3, 1, 3, 1), oracle.limiter(bl[1], 20000, 40000), oracle.limiter(bl[2], 20000, 40000)]
        for p in range(3):
            assert np.array_equal(got[k][p], want[p]), (n, p)
    # the upstream instances still serve other consumers on their own
    up = src.vszip.Bilateral(sigmaS=2.0, sigmaR=0.05)
    down = up.vszip.BoxBlur(hradius=3, vradius=3, planes=[0])
    assert np.array_equal(np.array(up.get_frame(0)[1]), oracle.bilateral_plane(frames[0][1], prm["sigmaS"][1], prm["sigmaR"][1], prm["algorithm"][1], prm["radius"][1], prm["step"][1], prm["PBFICnum"][1]))
    assert np.array_equal(np.array(down.get_frame(0)[0]), got[1][0])
    # a joint-bilateral instance (ref clip) is not a fusable stage: its consumer takes the staged path
    f2, s2 = vs.fusion_stats()
    j = src.vszip.Bilateral(ref=src, sigmaS=2.0, sigmaR=0.05).vszip.BoxBlur(hradius=1, vradius=1)
    j.get_frame(0)
    assert vs.fusion_stats() == (f2, s2)


def test_fused_two_input_and_metric_sinks(oracle):
    """Round 3: create-time fusion reaches the two-input filter and the metric sinks.
      * LimitFilter(src.vszip.BoxBlur(2,2), src) — the reference's own canonical construction (tests/test_int_parity.py:158-167):
        ONE upload of src (flt and src share the root), BoxBlur + LimitFilter on the device, one download; with a `planes` subset
        the planes LimitFilter does not process come back as flt's device planes; a `ref` chain of the same root as well;
      * PlaneAverage / PlaneMinMax on a pixel chain: the statistics read the chain's planes on the device, the output frame
        carries them (the reference returns clipa's frame), clipb may be another chain of the same root;
      * XPSNR(src, src.vszip.BoxBlur(..)): the distorted clip's chain runs on the device, src is requested and uploaded once.
    Every fused result is bit-identical to its unfused twin (every hop through host memory); vszip_plugin_fusion_stats counts them."""
    frames = [[fx.tiled_natural(s, np.uint16, p) if f == 0 else fx.splitmix64_plane(10 * f + p, s, np.uint16) for p, s in enumerate([(96, 160), (48, 80), (48, 80)])] for f in range(3)]
    src = vs.source(frames, vs.YUV420P16)
    same = lambda a, b: all(np.array_equal(np.array(a[p]), np.array(b[p])) for p in range(3))

    # ---- LimitFilter
    f0, s0 = vs.fusion_stats()
    fused = src.vszip.BoxBlur(hradius=2, vradius=2).vszip.LimitFilter(src, dark_thr=8, bright_thr=8, elast=3)
    got = [fused.get_frame(n) for n in range(3)]
    assert vs.fusion_stats() == (f0 + 3, s0 + 3)
    flt = _materialize(src.vszip.BoxBlur(hradius=2, vradius=2), vs.YUV420P16, 3)
    twin = flt.vszip.LimitFilter(_materialize(src, vs.YUV420P16, 3), dark_thr=8, bright_thr=8, elast=3)
    f1, s1 = vs.fusion_stats()
    for n in range(3):
        assert same(got[n], twin.get_frame(n)), n
    assert vs.fusion_stats() == (f1, s1)
    # a planes subset (unprocessed planes are flt's: from the device) and a `ref` chain of the same root
    part = src.vszip.BoxBlur(hradius=2, vradius=2).vszip.LimitFilter(src, ref=src.vszip.BoxBlur(hradius=4, vradius=4), dark_thr=[16, 4], bright_thr=[8, 2], planes=[0])
    ptwin = flt.vszip.LimitFilter(_materialize(src, vs.YUV420P16, 3), ref=_materialize(src.vszip.BoxBlur(hradius=4, vradius=4), vs.YUV420P16, 3),
                                  dark_thr=[16, 4], bright_thr=[8, 2], planes=[0])
    assert same(part.get_frame(1), ptwin.get_frame(1))
    # against the oracle (thresholds: the reference's build scales a clip without a range prop as limited)
    fr = part.get_frame(1)
    for p in range(3):
        fl = oracle.boxblur(frames[1][p], 2, 1, 2, 1)
        if p == 0:
            dk, br = oracle.scale_value_from_8bit(16, False, 16, True), oracle.scale_value_from_8bit(8, False, 16, True)
            fl = oracle.limit_filter(fl, frames[1][p], oracle.boxblur(frames[1][p], 4, 1, 4, 1), dk, br, 2.0)
        assert np.array_equal(np.array(fr[p]), fl), p

    # ---- PlaneAverage / PlaneMinMax as sinks
    f2, s2 = vs.fusion_stats()
    chain = src.vszip.Bilateral(sigmaS=2.0, sigmaR=2.0).vszip.BoxBlur(hradius=3, vradius=3, planes=[0])
    pa = chain.vszip.PlaneAverage(exclude=[-1], planes=[0, 1, 2], clipb=src.vszip.BoxBlur(hradius=1, vradius=1))
    pm = chain.vszip.PlaneMinMax(minthr=0.1, maxthr=0.1, planes=[0, 1, 2], clipb=src)
    ga, gm = pa.get_frame(0), pm.get_frame(0)
    assert vs.fusion_stats() == (f2 + 2, s2 + 3 + 2)
    mat = _materialize(chain, vs.YUV420P16, 3)
    ta = mat.vszip.PlaneAverage(exclude=[-1], planes=[0, 1, 2], clipb=_materialize(src.vszip.BoxBlur(hradius=1, vradius=1), vs.YUV420P16, 3)).get_frame(0)
    tm = mat.vszip.PlaneMinMax(minthr=0.1, maxthr=0.1, planes=[0, 1, 2], clipb=_materialize(src, vs.YUV420P16, 3)).get_frame(0)
    assert same(ga, ta) and same(gm, tm)  # the output frames carry the chain's pixels
    for k in ("psmAvg", "psmDiff"):
        assert ga.props[k] == ta.props[k], k
    for k in ("psmMin", "psmMax", "psmDiff"):
        assert gm.props[k] == tm.props[k], k

    # ---- XPSNR with a fused distorted clip of the reference's own root
    s8 = vs.source([[(p >> 8).astype(np.uint8) for p in fr_] for fr_ in frames], vs.YUV420P8)
    f3, s3 = vs.fusion_stats()
    xp = s8.vszip.XPSNR(s8.vszip.BoxBlur(hradius=2, vradius=2), verbose=False)
    gx = [xp.get_frame(n) for n in range(3)]
    assert vs.fusion_stats() == (f3 + 3, s3 + 3)
    tx = s8.vszip.XPSNR(_materialize(s8.vszip.BoxBlur(hradius=2, vradius=2), vs.YUV420P8, 3), verbose=False)
    for n in range(3):
        want = tx.get_frame(n)
        assert same(gx[n], want), n
        for k in ("XPSNR_Y", "XPSNR_U", "XPSNR_V"):
            assert gx[n].props[k] == want.props[k], (n, k)


def test_fused_pipeline_into_ssimulacra2(oracle):
    """BASELINE config 5 through libvszip.so: SSIMULACRA2(src, src.Bilateral().BoxBlur()) uploads the source frame
    ONCE (both inputs share the root), runs Bilateral and BoxBlur on the device and returns a score — the
    processed frame never crosses the host link."""
    base = [np.ascontiguousarray(p[:160, :256]) for p in fx.crop_rgbs()]
    src = vs.source([base], vs.RGBS, props={"_Transfer": 8})
    f0, s0 = vs.fusion_stats()
    u0, d0 = vs.upload_stats()
    s = src.vszip.SSIMULACRA2(src.vszip.Bilateral(sigmaS=2.0, sigmaR=2.0).vszip.BoxBlur(hradius=2, vradius=2)).get_frame(0).props["SSIMULACRA2"]
    assert vs.fusion_stats() == (f0 + 1, s0 + 2)
    # round 6: "uploads the source frame ONCE" is counted — three planes cross the link, the chain's root is answered by the reference's copy
    # (until then the frame went up twice: the 8K pipeline's trace showed six plane copies a frame, 70 fps where three give 140)
    assert vs.upload_stats() == (u0 + 3, d0 + 3)
    prm = oracle.bilateral_params([2], [2], yuv=False, ssw=0, ssh=0)
    want = [oracle.boxblur(oracle.bilateral_plane(p, prm["sigmaS"][i], prm["sigmaR"][i], prm["algorithm"][i], prm["radius"][i], prm["step"][i], prm["PBFICnum"][i]), 2, 1, 2, 1)
            for i, p in enumerate(base)]
    assert s == pytest.approx(oracle.ssimulacra2(base, want), abs=1e-7)


def test_ssimulacra2_yuv_sources_on_the_device(oracle):
    """SURVEY 8f rank 1, the YUV half (round 3): a YUV clip goes to the GPU as it is — 12 MB per 4K YUV420P8 frame instead
    of 100 MB of RGBS — and hz.toRGBS's chroma upsampling + matrix + sRGBtoLinearRGB run fused into the first pass. The
    output clip is still the host-converted reference (ssimulacra2.zig:53), here through the test host's stand-in (which
    must agree with oracle/vs_host.py bit for bit). `_Matrix` of the frames wins over toRGBS's matrix_in = 601 / 709."""
    from oracle import vs_host as vh

    vs.core_standins(True)
    try:
        for fmt_id, bits in ((vs.YUV420P8, 8), (vs.YUV420P16, 16)):
            ref = [np.ascontiguousarray(p[:80, :128] if i else p[:160, :256]) for i, p in enumerate(fx.crop_yuv(bits))]
            dis = [vh.std_boxblur(p, 1, 1) for p in ref]
            props = {"_Matrix": 1, "_ColorRange": 1, "_ChromaLocation": 0}
            a, b = vs.source([ref], fmt_id, props=props), vs.source([dis], fmt_id, props=props)
            out = a.vszip.SSIMULACRA2(b)
            f = out.get_frame(0)
            lin_r, lin_d = vh.yuv_to_linear_rgbs(ref, bits, 1, 1, 1, 0), vh.yuv_to_linear_rgbs(dis, bits, 1, 1, 1, 0)
            assert f.props["SSIMULACRA2"] == pytest.approx(oracle.ssimulacra2(lin_r, lin_d), abs=1e-7)
            assert out.format_id == vs.RGBS
            for p in range(3):
                assert np.array_equal(np.asarray(f[p]).view(np.uint32), lin_r[p].view(np.uint32)), (bits, p)
            # only the reference went through the host's resize (format, then transfer); the distorted clip never did
            assert sum("resize.Bicubic" in l for l in vs.standin_log()) == 2, vs.standin_log()
            vs.core_standins(True)  # (clears the log)
        # no _Matrix on the frames: toRGBS's own choice — BT.601 for a clip of <= 650 rows (src/helper.zig:231)
        ref = [np.ascontiguousarray(p[:80, :128] if i else p[:160, :256]) for i, p in enumerate(fx.crop_yuv(8))]
        dis = [vh.std_boxblur(p, 2, 2) for p in ref]
        a, b = vs.source([ref], vs.YUV420P8), vs.source([dis], vs.YUV420P8)
        want = oracle.ssimulacra2(vh.yuv_to_linear_rgbs(ref, 8, 1, 1, 6, 0), vh.yuv_to_linear_rgbs(dis, 8, 1, 1, 6, 0))
        assert a.vszip.SSIMULACRA2(b).get_frame(0).props["SSIMULACRA2"] == pytest.approx(want, abs=1e-7)
        # mixed: YUV420P8 reference against the same picture as RGB24 (converted on the device one by one)
        rgb = [np.ascontiguousarray(p[:160, :256]) for p in fx.crop_rgb24()]
        c = vs.source([rgb], vs.RGB24)
        want = oracle.ssimulacra2(vh.yuv_to_linear_rgbs(ref, 8, 1, 1, 6, 0), vh.to_linear_rgbs(rgb, "RGB", 8))
        assert a.vszip.SSIMULACRA2(c).get_frame(0).props["SSIMULACRA2"] == pytest.approx(want, abs=1e-7)
        # a vszip chain on the YUV clip stays on the device: one upload of the 4:2:0 planes, BoxBlur there, score
        f0, s0 = vs.fusion_stats()
        got = a.vszip.SSIMULACRA2(a.vszip.BoxBlur(hradius=2, vradius=2)).get_frame(0).props["SSIMULACRA2"]
        assert vs.fusion_stats() == (f0 + 1, s0 + 1)
        blur = [oracle.boxblur(p, 2, 1, 2, 1) for p in ref]
        assert got == pytest.approx(oracle.ssimulacra2(vh.yuv_to_linear_rgbs(ref, 8, 1, 1, 6, 0), vh.yuv_to_linear_rgbs(blur, 8, 1, 1, 6, 0)), abs=1e-7)
    finally:
        vs.core_standins(False)


def test_ssimulacra2_fused_reference_against_its_own_linear_root(oracle):
    """lin.vszip.BoxBlur().vszip.SSIMULACRA2(lin) on a linear RGBS clip — the reference is a fused chain
    whose root IS the distorted clip, and the distorted clip needs no pre-stage (raw1 set, raw2 not). Both inputs are one
    node: requested once, and the distorted planes are that frame's own upload (it used to be a null frame)."""
    base = [np.ascontiguousarray(p[:96, :160]) for p in fx.crop_rgbs()]
    lin = vs.source([base], vs.RGBS, props={"_Transfer": 8})
    blur = [oracle.boxblur(p, 2, 1, 2, 1) for p in base]
    s = lin.vszip.BoxBlur(hradius=2, vradius=2).vszip.SSIMULACRA2(lin).get_frame(0).props["SSIMULACRA2"]
    assert s == pytest.approx(oracle.ssimulacra2(blur, base), abs=1e-7)
    # and the mirrored order (which always worked)
    s2 = lin.vszip.SSIMULACRA2(lin.vszip.BoxBlur(hradius=2, vradius=2)).get_frame(0).props["SSIMULACRA2"]
    assert s2 == pytest.approx(oracle.ssimulacra2(base, blur), abs=1e-7)


def test_frames_shard_over_devices_like_one_device(oracle):
    """frame n runs on GPU n mod #GPUs (vszip_plugin.cpp device_of_frame), with per-device slots,
    contexts and Bilateral LUTs. On a multi-GPU host every frame must equal the single-device result; a worker
    thread serves frames of every GPU, so the C ABI sets the device itself in its copy / sync paths."""
    import ctypes as C

    import vszip_amd

    n_dev = 0
    lib = vszip_amd.capi.load()
    for d in range(16):
        ctx = C.c_void_p()
        if lib.vszip_ctx_create(d, C.byref(ctx)) != 0:
            break
        lib.vszip_ctx_destroy(ctx)
        n_dev += 1
    if n_dev < 2:
        pytest.skip("one visible GPU: the n mod #GPUs path needs a multi-GPU host")
    frames = [_yuv420p16(seed=7 * f) for f in range(2 * n_dev)]
    src = vs.source(frames, vs.YUV420P16)
    out = src.vszip.Bilateral(sigmaS=2.0, sigmaR=0.05).vszip.BoxBlur(hradius=5, vradius=5)
    prm = oracle.bilateral_params([2], [0.05], yuv=True, ssw=1, ssh=1)
    out.pull(len(frames), 4)  # several workers, every device
    for n in range(len(frames)):
        got = out.get_frame(n)
        for i, p in enumerate(frames[n]):
            b = oracle.bilateral_plane(p, prm["sigmaS"][i], prm["sigmaR"][i], prm["algorithm"][i], prm["radius"][i], prm["step"][i], prm["PBFICnum"][i])
            assert np.array_equal(got[i], oracle.boxblur(b, 5, 1, 5, 1)), (n, i)
