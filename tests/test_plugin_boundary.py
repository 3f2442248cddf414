"""The VapourSynth plugin boundary (libvszip.so) driven by the VapourSynth-free host
tests/fakevs: registration and create-time validation need no GPU and mirror the
reference's tests (tests/test_boxblur.py:131-163, test_eedi3.py:192-216, test_planeminmax.py,
test_bilateral.py, test_ssimulacra2.py, test_xpsnr.py error cases)."""
import numpy as np
import pytest

from fakevs import fakevs as vs

# src/vszip.zig:48,64,186,194,210,218 and src/vapoursynth/eedi3.zig:494 — byte-identical signatures
SIGNATURES = {
    "AdaptiveBinarize": "clip:vnode;clip2:vnode;c:int:opt;",  # src/vszip.zig:40
    "Bilateral": "clip:vnode;ref:vnode:opt;sigmaS:float[]:opt;sigmaR:float[]:opt;planes:int[]:opt;algorithm:int[]:opt;PBFICnum:int[]:opt",
    "BoxBlur": "clip:vnode;planes:int[]:opt;hradius:int:opt;hpasses:int:opt;vradius:int:opt;vpasses:int:opt",
    "EEDI3": "clip:vnode;field:int;dh:int:opt;alpha:float:opt;beta:float:opt;gamma:float:opt;nrad:int:opt;mdis:int:opt;hp:int:opt;vcheck:int:opt;"
             "vthresh0:float:opt;vthresh1:float:opt;vthresh2:float:opt;sclip:vnode:opt;mclip:vnode:opt;",
    "LimitFilter": "flt:vnode;src:vnode;ref:vnode:opt;dark_thr:float[]:opt;bright_thr:float[]:opt;elast:float[]:opt;planes:int[]:opt;",  # src/vszip.zig:154
    "Limiter": "clip:vnode;min:float[]:opt;max:float[]:opt;tv_range:int:opt;mask:int:opt;planes:int[]:opt;",  # src/vszip.zig:162
    "PlaneAverage": "clipa:vnode;exclude:int[];clipb:vnode:opt;planes:int[]:opt;prop:data:opt;",
    "PlaneMinMax": "clipa:vnode;minthr:float:opt;maxthr:float:opt;clipb:vnode:opt;planes:int[]:opt;prop:data:opt;",
    "SSIMULACRA2": "reference:vnode;distorted:vnode;",
    "XPSNR": "reference:vnode;distorted:vnode;temporal:int:opt;verbose:int:opt;",
}
SIGNATURES["EEDI3H"] = SIGNATURES["EEDI3"]


def test_registration():
    pid, version, nfuncs = vs.plugin_info()
    assert pid == "com.julek.vszip" and nfuncs == len(SIGNATURES)
    for fn, sig in SIGNATURES.items():
        assert vs.signature_string(fn) == sig, fn


def _yuv(fmt=vs.YUV420P8, w=64, h=32, length=1):
    return vs.blank(fmt, w, h, [16, 128, 128], length)


@pytest.mark.parametrize("args,msg", [
    (dict(hradius=0, vradius=0, hpasses=0, vpasses=0), "nothing to be performed"),
    (dict(hradius=5, vradius=5, hpasses=0, vpasses=0), "nothing to be performed"),
    (dict(planes=[3]), "plane index out of range"),
    (dict(planes=[-1]), "plane index out of range"),
    (dict(planes=[0, 0]), "plane specified twice"),
    (dict(hradius=40), "hradius too large"),
    (dict(vradius=16), "vradius too large"),
])
def test_boxblur_validation(args, msg):
    with pytest.raises(vs.Error, match=msg):
        _yuv().vszip.BoxBlur(**args)


def test_unsupported_int_format():
    with pytest.raises(vs.Error, match="not supported Int format"):
        vs.blank(vs.GRAY32, 64, 64, 0).vszip.BoxBlur(hradius=1, vradius=1)


@pytest.mark.parametrize("args,msg", [
    (dict(sigmaS=-1.0), "Invalid \"sigmaS\""),
    (dict(PBFICnum=1), "Invalid \"PBFICnum\""),
    (dict(sigmaR=[1.0, 2.0, 3.0, 4.0]), "too many elements"),
    (dict(algorithm=3), "above maximum"),
    (dict(sigmaR=-0.5), "below minimum"),
    (dict(sigmaS=40.0, algorithm=2), "plane too small"),
])
def test_bilateral_validation(args, msg):
    with pytest.raises(vs.Error, match=msg):
        _yuv(vs.YUV420P16).vszip.Bilateral(**args)


def test_bilateral_ref_mismatch():
    a, b = vs.blank(vs.GRAY8, 64, 32, 0, 5), vs.blank(vs.GRAY8, 64, 32, 0, 3)
    with pytest.raises(vs.Error, match="second clip has less frames"):
        a.vszip.Bilateral(ref=b)
    with pytest.raises(vs.Error, match="same width and height"):
        a.vszip.Bilateral(ref=vs.blank(vs.GRAY8, 48, 32, 0, 5))


@pytest.mark.parametrize("args,msg", [
    (dict(field=4), "field must be 0, 1, 2, or 3"),
    (dict(field=2, dh=1), "field must be 0 or 1 when dh=True"),
    (dict(field=1, alpha=1.5), "alpha must be between"),
    (dict(field=1, alpha=0.8, beta=0.5), "alpha \\+ beta"),
    (dict(field=1, gamma=-1.0), "gamma must be greater"),
    (dict(field=1, nrad=4), "nrad must be between"),
    (dict(field=1, mdis=41), "mdis must be between"),
    (dict(field=1, vcheck=5), "vcheck must be"),
    (dict(field=1, vthresh0=0.0), "vthresh0, vthresh1 and vthresh2"),
])
def test_eedi3_validation(args, msg):
    src = vs.blank(vs.GRAYS, 128, 64, 0.5)
    for fn in ("EEDI3", "EEDI3H"):
        with pytest.raises(vs.Error, match=msg):
            getattr(src.vszip, fn)(**args)


def test_eedi3_format_and_required_field():
    with pytest.raises(vs.Error, match="only 32-bit float input is supported"):
        vs.blank(vs.GRAY8, 128, 64, 0).vszip.EEDI3(field=1)
    with pytest.raises(vs.Error, match="field is required"):
        vs.blank(vs.GRAYS, 128, 64, 0.5).vszip.EEDI3()
    with pytest.raises(vs.Error, match="height must be mod 2"):
        vs.blank(vs.GRAYS, 128, 63, 0.5).vszip.EEDI3(field=1)
    with pytest.raises(vs.Error, match="width must be mod 2"):
        vs.blank(vs.GRAYS, 127, 64, 0.5).vszip.EEDI3H(field=1)


def test_eedi3_output_geometry():
    src = vs.blank(vs.GRAYS, 128, 64, 0.5, length=3, fps=(24, 1))
    assert (src.vszip.EEDI3(field=1, dh=1).width, src.vszip.EEDI3(field=1, dh=1).height) == (128, 128)
    assert (src.vszip.EEDI3H(field=1, dh=1).width, src.vszip.EEDI3H(field=1, dh=1).height) == (256, 64)
    dbl = src.vszip.EEDI3(field=3)
    assert dbl.num_frames == 6 and dbl.fps == (48, 1)


def test_planestats_validation():
    y = _yuv(vs.YUV420P16)
    with pytest.raises(vs.Error, match="exclude is required"):
        y.vszip.PlaneAverage()
    with pytest.raises(vs.Error, match="minthr should be a float between 0.0 and 1.0"):
        y.vszip.PlaneMinMax(minthr=1.5)
    with pytest.raises(vs.Error, match="maxthr should be a float between 0.0 and 1.0"):
        y.vszip.PlaneMinMax(maxthr=-0.1)
    with pytest.raises(vs.Error, match="float chroma"):
        vs.blank(vs.YUV444PS, 64, 32, [0.5, 0.0, 0.0]).vszip.PlaneMinMax(minthr=0.1, planes=[0, 1, 2])
    with pytest.raises(vs.Error, match="exclude is not supported for 32-bit integer clips"):
        vs.blank(vs.GRAY32, 64, 32, 7).vszip.PlaneAverage(exclude=[1])
    with pytest.raises(vs.Error, match="second clip has less frames"):
        vs.blank(vs.GRAY8, 64, 32, 0, 5).vszip.PlaneAverage(exclude=[-1], clipb=vs.blank(vs.GRAY8, 64, 32, 0, 3))


def test_ssimulacra2_and_xpsnr_validation():
    a = vs.blank(vs.RGBS, 64, 64, [0.1, 0.2, 0.3])
    with pytest.raises(vs.Error, match="SSIMULACRA2 : clips must have the same dimensions"):
        a.vszip.SSIMULACRA2(vs.blank(vs.RGBS, 48, 64, [0.1, 0.2, 0.3]))
    with pytest.raises(vs.Error, match="SSIMULACRA2 : clips must have the same length"):
        a.vszip.SSIMULACRA2(vs.blank(vs.RGBS, 64, 64, [0.1, 0.2, 0.3], length=2))
    with pytest.raises(vs.Error, match="half-float"):
        a.vszip.SSIMULACRA2(vs.blank(vs.RGBH, 64, 64, [0.1, 0.2, 0.3]))
    with pytest.raises(vs.Error, match="XPSNR : only supports YUV format clips"):
        vs.blank(vs.RGB24, 64, 64, [1, 2, 3]).vszip.XPSNR(vs.blank(vs.RGB24, 64, 64, [1, 2, 3]))
    with pytest.raises(vs.Error, match="XPSNR : only supports 8 or 10 bit clips"):
        _yuv(vs.YUV420P16).vszip.XPSNR(_yuv(vs.YUV420P16))
    with pytest.raises(vs.Error, match="only supports even width and height"):
        vs.blank(vs.YUV444P16 & ~(0xFF << 16) | (8 << 16), 63, 64, [1, 2, 3]).vszip.XPSNR(vs.blank(vs.YUV444P16 & ~(0xFF << 16) | (8 << 16), 63, 64, [1, 2, 3]))


def test_unknown_argument_rejected():
    with pytest.raises(vs.Error, match="no argument named"):
        _yuv().vszip.BoxBlur(radius=3)


@pytest.mark.parametrize("args,msg", [  # reference tests/test_limiter.py:189-205
    (dict(min=[0, 0, 0]), "min array is set but max array is not"),
    (dict(max=[255, 255, 255]), "max array is set but min array is not"),
    (dict(min=[0, 0], max=[255, 255, 255]), "min array must have the same number of elements as planes"),
    (dict(min=[0, 0, 0], max=[255, 255]), "max array must have the same number of elements as planes"),
    (dict(min=[-1, 0, 0], max=[255, 255, 255]), "min value must be greater than or equal to 0"),
    (dict(min=[0, 0, 0], max=[255, 255, 256]), "max value must be less than or equal to peak value"),
    (dict(min=[300, 0, 0], max=[255, 255, 255]), "min value must be less than or equal to peak value"),
    (dict(min=[200, 0, 0], max=[100, 255, 255]), "min value must be less than or equal to max value"),
    (dict(planes=[3]), "plane index out of range"),
    (dict(planes=[-1]), "plane index out of range"),
    (dict(planes=[0, 0]), "plane specified twice"),
])
def test_limiter_validation(args, msg):
    with pytest.raises(vs.Error, match=msg):
        _yuv().vszip.Limiter(**args)


def test_limiter_32bit_explicit_bounds_are_unreachable():
    """getPeakValue overflows for 32-bit integer clips (peak -1), so every explicit bound is above it
    (reference tests/test_limiter.py:150-160)."""
    with pytest.raises(vs.Error, match="min value must be less than or equal to peak value"):
        vs.blank(vs.GRAY32, 64, 32, 0).vszip.Limiter(min=[0], max=[10])


@pytest.mark.parametrize("args,msg", [
    (dict(dark_thr=[1, 2, 3, 4]), "dark_thr has too many elements"),
    (dict(bright_thr=-1.0), "below minimum"),
    (dict(dark_thr=256.0), "above maximum"),
    (dict(elast=70000.0), "above maximum"),
    (dict(planes=[3]), "plane index out of range"),
    (dict(planes=[1, 1]), "plane specified twice"),
])
def test_limit_filter_validation(args, msg):
    with pytest.raises(vs.Error, match=msg):
        _yuv().vszip.LimitFilter(src=_yuv(), **args)


def test_limit_filter_clip_mismatch():
    with pytest.raises(vs.Error, match="same width and height"):
        _yuv().vszip.LimitFilter(src=_yuv(w=48))
    with pytest.raises(vs.Error, match="same bit depth"):
        _yuv().vszip.LimitFilter(src=_yuv(vs.YUV420P10))
    with pytest.raises(vs.Error, match="same length"):
        _yuv(length=3).vszip.LimitFilter(src=_yuv(length=2))
    with pytest.raises(vs.Error, match="same length"):
        _yuv(length=3).vszip.LimitFilter(src=_yuv(length=3), ref=_yuv(length=2))


def test_adaptive_binarize_validation():
    with pytest.raises(vs.Error, match="only 8 bit int format supported"):
        _yuv(vs.YUV420P16).vszip.AdaptiveBinarize(clip2=_yuv(vs.YUV420P16))
    with pytest.raises(vs.Error, match="clip2 is required"):
        _yuv().vszip.AdaptiveBinarize()
    with pytest.raises(vs.Error, match="same width and height"):
        _yuv().vszip.AdaptiveBinarize(clip2=_yuv(w=48))
    with pytest.raises(vs.Error, match="second clip has less frames"):
        _yuv(length=3).vszip.AdaptiveBinarize(clip2=_yuv(length=2))
