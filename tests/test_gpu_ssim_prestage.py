"""GPU parity of SSIMULACRA2's colour pre-stage (hz.toRGBS + sRGBtoLinearRGB on the device) and of the
scores computed straight from the clips' own sample types (vszip_ssimulacra2_src):
  * vszip_to_rgbs_linear is BIT-EXACT against oracle/vs_host.py (zimg's integer -> float conversion, Gray ->
    R = G = B, the sRGB EOTF through zimg's approximate-gamma table) for RGB24/30/48, RGBS, GRAY8/10/16, GRAYS;
  * the fused path scores within 1e-7 of the oracle fed with the oracle's converted planes;
  * the reference's own goldens (tests/goldens/ssimulacra2.json; rel=1e-3 in tests/test_ssimulacra2.py:60)
    come out of the GPU from the raw RGB24 / RGBS / GRAY8 crop within 2e-4."""
import numpy as np
import pytest

import fixtures as fx
from oracle import vs_host as vh

pytestmark = pytest.mark.gpu
TOL = 1e-7


@pytest.fixture(scope="module")
def dev():
    import vszip_amd

    d = vszip_amd.Device(0)
    yield d
    d.close()


def _at_depth(p8, bits):
    return p8 if bits == 8 else np.floor(p8.astype(np.float64) * ((1 << bits) - 1) / 255.0 + 0.5).astype(np.uint16)


def _sources():
    rgb8 = [np.ascontiguousarray(p) for p in fx.crop_rgb24()]
    g8 = fx.crop_gray8()
    out = {
        "RGB24": ("RGB", rgb8, 8), "RGB30": ("RGB", [_at_depth(p, 10) for p in rgb8], 10), "RGB48": ("RGB", [_at_depth(p, 16) for p in rgb8], 16),
        "RGBS": ("RGBS", [np.ascontiguousarray(p) for p in fx.crop_rgbs()], 32),
        "GRAY8": ("GRAY", [g8], 8), "GRAY10": ("GRAY", [(g8.astype(np.uint16) << 2)], 10), "GRAY16": ("GRAY", [np.ascontiguousarray(fx.crop_gray16())], 16),
        "GRAYS": ("GRAY", [np.ascontiguousarray(fx.crop_grays())], 32),
    }
    return out


def _fmt(dev, family, planes, bits, linearize=True):
    return dev.ssim_source("GRAY" if family == "GRAY" else "RGB", planes[0].dtype, bits, linearize)


@pytest.mark.parametrize("name", ["RGB24", "RGB30", "RGB48", "RGBS", "GRAY8", "GRAY10", "GRAY16", "GRAYS"])
@pytest.mark.parametrize("crop", [None, (313, 631), (7, 13)])
def test_prestage_bit_exact(dev, name, crop):
    family, planes, bits = _sources()[name]
    if crop:
        planes = [np.ascontiguousarray(p[:crop[0], :crop[1]]) for p in planes]
    want = vh.to_linear_rgbs(planes, family, bits)
    got = [dev.download(d) for d in dev.to_rgbs_linear(_fmt(dev, family, planes, bits), [dev.upload(p) for p in planes])]
    for c in range(3):
        assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), (name, crop, c, np.abs(got[c] - want[c]).max())
    # _Transfer == LINEAR: the depth conversion alone
    want0 = vh.to_rgbs(planes, family, bits)
    got0 = [dev.download(d) for d in dev.to_rgbs_linear(_fmt(dev, family, planes, bits, linearize=False), [dev.upload(p) for p in planes])]
    for c in range(3):
        assert np.array_equal(got0[c].view(np.uint32), want0[c].view(np.uint32)), (name, crop, c)


@pytest.mark.parametrize("name", ["RGB24", "RGB30", "RGB48", "RGBS", "GRAY8", "GRAY16", "GRAYS"])
@pytest.mark.parametrize("crop", [None, (313, 631), (135, 241), (64, 66), (7, 13)])
def test_fused_score_matches_oracle(dev, oracle, name, crop):
    family, ref, bits = _sources()[name]
    if crop:
        ref = [np.ascontiguousarray(p[:crop[0], :crop[1]]) for p in ref]
    dis = [vh.std_boxblur(p, 1, 1) for p in ref]
    want = oracle.ssimulacra2(vh.to_linear_rgbs(ref, family, bits), vh.to_linear_rgbs(dis, family, bits))
    fmt = _fmt(dev, family, ref, bits)
    # odd row pitches too: the kernel's 4-sample vector loads need alignment, everything else takes the scalar path
    for align in (32, 1):
        got = dev.ssimulacra2_src(fmt, [dev.upload(p, align) for p in ref], [dev.upload(p, align) for p in dis])[0]
        assert got == pytest.approx(want, abs=TOL), (name, crop, align, got, want)


@pytest.mark.parametrize("key,name,radius", [
    ("RGBS|full|dist=blur1", "RGBS", 1), ("RGB24|full|dist=blur1", "RGB24", 1), ("RGB24|full|dist=blur3", "RGB24", 3),
    ("GRAY8|full|dist=blur1", "GRAY8", 1), ("GRAY8|full|dist=blur3", "GRAY8", 3),
])
def test_reference_goldens_from_raw_sources(dev, key, name, radius):
    g = fx.ref_goldens()["exact"]["ssimulacra2"][key]
    family, ref, bits = _sources()[name]
    dis = [vh.std_boxblur(p, radius, radius) for p in ref]
    got = dev.ssimulacra2_src(_fmt(dev, family, ref, bits), [dev.upload(p) for p in ref], [dev.upload(p) for p in dis])[0]
    assert got == pytest.approx(g, rel=2e-4), (key, got, g)


def test_batch_of_pairs_and_identity(dev, oracle):
    family, ref, bits = _sources()["RGB24"]
    d1 = [vh.std_boxblur(p, 1, 1) for p in ref]
    d3 = [vh.std_boxblur(p, 3, 3) for p in ref]
    up = lambda ps: [dev.upload(p) for p in ps]
    fmt = _fmt(dev, family, ref, bits)
    got = dev.ssimulacra2_src(fmt, up(ref) + up(ref) + up(ref), up(d1) + up(d3) + up(ref))
    lin = lambda ps: vh.to_linear_rgbs(ps, family, bits)
    assert got[0] == pytest.approx(oracle.ssimulacra2(lin(ref), lin(d1)), abs=TOL)
    assert got[1] == pytest.approx(oracle.ssimulacra2(lin(ref), lin(d3)), abs=TOL)
    assert got[2] > 99.9 and got[0] > got[1]
    c = [np.full((64, 64), v, np.uint8) for v in (77, 51, 128)]
    assert dev.ssimulacra2_src(fmt, up(c), up(c))[0] == 100.0  # reference tests/test_ssimulacra2.py:65-67


def test_depth_parity_of_the_score(dev):
    """reference tests/test_int_parity.py:368-395: the score is depth-independent by construction (each depth
    normalised by its own peak); 8 / 10 / 16 bit within 0.5 (measured there: ~0.16)."""
    s = {}
    for name in ("RGB24", "RGB30", "RGB48"):
        family, ref, bits = _sources()[name]
        dis = [vh.std_boxblur(p, 1, 1) for p in ref]
        s[name] = dev.ssimulacra2_src(_fmt(dev, family, ref, bits), [dev.upload(p) for p in ref], [dev.upload(p) for p in dis])[0]
    assert abs(s["RGB24"] - s["RGB48"]) <= 0.5 and abs(s["RGB30"] - s["RGB48"]) <= 0.5, s
