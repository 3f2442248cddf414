"""GPU parity of SSIMULACRA2's colour pre-stage for YUV clips (round 3): hz.toRGBS
(src/helper.zig:225-243: resize.Bicubic(format=RGBS, matrix_in=...)) + sRGBtoLinearRGB
(src/vapoursynth/ssimulacra2.zig:132-162) on the device, straight from the clip's own planes.
  * vszip_to_rgbs_linear(VSZIP_CF_YUV) is BIT-EXACT against oracle/vs_host.py::yuv_to_linear_rgbs (zimg's
    integer -> float conversion, Catmull-Rom chroma resampler with its accumulation order, YUV -> RGB FMA chain,
    approximate-gamma table) for 4:2:0 / 4:2:2 / 4:4:4 / 4:1:0, 8 / 10 / 16 bit and f32, every chroma siting,
    BT.709 / 601 / 2020, odd, ragged and tiny geometries;
  * the fused pass (vszip_ssimulacra2_src) scores within 1e-7 of the oracle fed with the oracle's converted planes;
  * the reference's seven YUV goldens (tests/goldens/ssimulacra2.json; rel = 1e-3 in tests/test_ssimulacra2.py:60)
    come out of the GPU from the raw YUV planes within 2e-4 (`tiny`: 5e-3, see tests/test_oracle_zimg_goldens.py)."""
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


def _clip(bits, ssw, ssh, sample="int", crop=None, loc=0):
    rgb = fx.crop_rgb24()
    if crop:
        rgb = np.ascontiguousarray(rgb[:, :crop[0], :crop[1]])
    return [np.ascontiguousarray(p) for p in vh.rgb24_to_yuv(rgb, bits, ssw, ssh, sample=sample, loc=loc)]


def _fmt(dev, planes, bits, ssw, ssh, matrix=1, loc=0, linearize=True):
    return dev.ssim_source("YUV", planes[0].dtype, bits, linearize, ssw=ssw, ssh=ssh, matrix=matrix, chroma_loc=loc)


FORMATS = {"YUV420P8": (8, 1, 1, "int"), "YUV420P10": (10, 1, 1, "int"), "YUV420P16": (16, 1, 1, "int"), "YUV420PS": (32, 1, 1, "f32"),
           "YUV422P8": (8, 1, 0, "int"), "YUV444P16": (16, 0, 0, "int"), "YUV440P8": (8, 0, 1, "int"), "YUV410P8": (8, 2, 2, "int"), "YUV411P8": (8, 2, 0, "int")}


@pytest.mark.parametrize("name", sorted(FORMATS))
@pytest.mark.parametrize("crop", [None, (318, 638), (316, 628), (12, 20), (8, 12)])
def test_yuv_prestage_bit_exact(dev, name, crop):
    bits, ssw, ssh, sample = FORMATS[name]
    if crop and (crop[0] % (1 << ssh) or crop[1] % (1 << ssw)):
        pytest.skip("VapourSynth has no such clip")
    planes = _clip(bits, ssw, ssh, sample, crop)
    for linearize in (True, False):
        want = vh.yuv_to_linear_rgbs(planes, bits, ssw, ssh, 1, 0) if linearize else vh.yuv_to_rgbs(planes, bits, ssw, ssh, 1, 0)
        got = [dev.download(d) for d in dev.to_rgbs_linear(_fmt(dev, planes, bits, ssw, ssh, linearize=linearize), [dev.upload(p) for p in planes])]
        for c in range(3):
            assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), (name, crop, linearize, c, np.abs(got[c] - want[c]).max())


@pytest.mark.parametrize("loc", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("matrix", [1, 6, 9])
def test_yuv_prestage_siting_and_matrix(dev, loc, matrix):
    planes = _clip(8, 1, 1, "int", (64, 96), loc)
    want = vh.yuv_to_linear_rgbs(planes, 8, 1, 1, matrix, loc)
    got = [dev.download(d) for d in dev.to_rgbs_linear(_fmt(dev, planes, 8, 1, 1, matrix, loc), [dev.upload(p) for p in planes])]
    for c in range(3):
        assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), (loc, matrix, c)


def test_yuv_full_range_and_noise(dev):
    """full-range integer YUV (a `_ColorRange = 0` clip) and white-noise planes (every table entry, values far outside
    [0, 1] after the matrix: the transfer table's clamps)."""
    planes = [fx.splitmix64_plane(11, (64, 96), np.uint8), fx.splitmix64_plane(12, (32, 48), np.uint8), fx.splitmix64_plane(13, (32, 48), np.uint8)]
    for limited in (True, False):
        fmt = dev.ssim_source("YUV", np.uint8, 8, True, limited=limited, ssw=1, ssh=1, matrix=6, chroma_loc=0)
        want = [vh.srgb_to_linear(p) for p in vh.yuv_to_rgbs(planes, 8, 1, 1, 6, 0, limited=limited)]
        got = [dev.download(d) for d in dev.to_rgbs_linear(fmt, [dev.upload(p) for p in planes])]
        for c in range(3):
            assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32)), (limited, c)


@pytest.mark.parametrize("name", ["YUV420P8", "YUV420P16", "YUV420PS", "YUV422P8", "YUV444P16", "YUV410P8"])
@pytest.mark.parametrize("crop", [None, (316, 628), (136, 244), (64, 68), (8, 12)])
def test_yuv_fused_score_matches_oracle(dev, oracle, name, crop):
    bits, ssw, ssh, sample = FORMATS[name]
    ref = _clip(bits, ssw, ssh, sample, crop)
    dis = [vh.std_boxblur(p, 1, 1) for p in ref]
    want = oracle.ssimulacra2(vh.yuv_to_linear_rgbs(ref, bits, ssw, ssh, 1, 0), vh.yuv_to_linear_rgbs(dis, bits, ssw, ssh, 1, 0))
    fmt = _fmt(dev, ref, bits, ssw, ssh)
    for align in (32, 1):  # odd row pitches too: the vector loads need alignment, everything else takes the scalar path
        got = dev.ssimulacra2_src(fmt, [dev.upload(p, align) for p in ref], [dev.upload(p, align) for p in dis])[0]
        assert got == pytest.approx(want, abs=TOL), (name, crop, align, got, want)


@pytest.mark.parametrize("loc", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("name", ["YUV420P8", "YUV420P16", "YUV422P8"])
def test_yuv_fused_score_every_siting(dev, oracle, name, loc):
    """The fused pass under every _ChromaLocation, on a frame of several tiles each way: co-sited chroma (left / top) gives resampling
    tables whose first tap is NOT monotonic (a sample that sits on a chroma sample has one non-zero tap, its neighbour four, starting
    one sample earlier), and a tile's chroma rows / columns start at the SMALLEST first tap of its rows / columns (round 5: the tile
    took its first row's, and top-sited clips read one row off)."""
    bits, ssw, ssh, sample = FORMATS[name]
    ref = _clip(bits, ssw, ssh, sample, None, loc)
    dis = [vh.std_boxblur(p, 1, 1) for p in ref]
    fmt = _fmt(dev, ref, bits, ssw, ssh, loc=loc)
    got = dev.ssimulacra2_src(fmt, [dev.upload(p, 32) for p in ref], [dev.upload(p, 32) for p in dis])[0]
    want = oracle.ssimulacra2(vh.yuv_to_linear_rgbs(ref, bits, ssw, ssh, 1, loc), vh.yuv_to_linear_rgbs(dis, bits, ssw, ssh, 1, loc))
    assert got == pytest.approx(want, abs=TOL), (name, loc, got, want)


@pytest.mark.parametrize("loc", [0, 1, 2, 3])
@pytest.mark.parametrize("bits,crop", [(8, None), (16, None), (10, None), (8, (316, 628)), (8, (136, 244)), (16, (64, 68))])
def test_yuv420_split_pass_is_the_tile_kernel(dev, oracle, bits, crop, loc):
    """4:2:0 integer clips (round 5) go through ssim_yuv420_rgb_kernel (persistent workgroups, transfer table in LDS, taps from the tables'
    period; the frame's outermost tiles from the tables) + the f32 pyramid pass; VSZIP_SSIM_NO_YUV420_LDS=1 sends them through the fused tile
    kernel like every other YUV format. Same f32 operations in the same order: the scores are EQUAL, for every chroma siting (co-sited
    chroma makes the tables' first taps non-monotonic) and on frames of one tile, of ragged tiles and of several tiles each way."""
    ref = _clip(bits, 1, 1, "int", crop, loc)
    dis = [vh.std_boxblur(p, 1, 1) for p in ref]
    fmt = _fmt(dev, ref, bits, 1, 1, loc=loc)
    up = lambda ps: [dev.upload(p, 32) for p in ps]
    got = dev.ssimulacra2_src(fmt, up(ref) + up(dis) + up(ref), up(dis) + up(ref) + up(ref))
    with dev.options(VSZIP_SSIM_NO_YUV420_LDS=1):
        tile = dev.ssimulacra2_src(fmt, up(ref) + up(dis) + up(ref), up(dis) + up(ref) + up(ref))
    assert got == tile, (bits, crop, loc, got, tile)
    assert got[2] > 99.9


@pytest.mark.parametrize("loc", [0, 1])
@pytest.mark.parametrize("ssw", [0, 1])
@pytest.mark.parametrize("bits,crop", [(8, None), (16, None), (10, None), (8, (316, 628)), (8, (136, 244)), (16, (64, 68)), (8, (8, 12)), (16, (318, 636))])
def test_yuv444_and_422_row_pass_is_the_tile_kernel(dev, oracle, bits, crop, ssw, loc):
    """4:4:4 and 4:2:2 integer clips (round 6) go through ssim_yuvrow_rgb_kernel (persistent workgroups, transfer table in LDS; 4:2:2: taps from
    the horizontal table's period, the frame's first and last columns from the table) + the f32 pyramid pass; VSZIP_SSIM_NO_YUV420_LDS=1 sends
    them through the fused tile kernel. Same f32 operations in the same order: the scores are EQUAL - left- and centre-sited chroma, 8 / 10 / 16
    bit, widths of one group, of several, ragged against the tile kernel's 256-sample tiles - and equal to the oracle's."""
    if ssw == 0 and loc:
        pytest.skip("4:4:4 has no chroma siting")
    ref = _clip(bits, ssw, 0, "int", crop, loc)
    dis = [vh.std_boxblur(p, 1, 1) for p in ref]
    fmt = _fmt(dev, ref, bits, ssw, 0, loc=loc)
    up = lambda ps: [dev.upload(p, 32) for p in ps]
    got = dev.ssimulacra2_src(fmt, up(ref) + up(dis) + up(ref), up(dis) + up(ref) + up(ref))
    with dev.options(VSZIP_SSIM_NO_YUV420_LDS=1):
        tile = dev.ssimulacra2_src(fmt, up(ref) + up(dis) + up(ref), up(dis) + up(ref) + up(ref))
    assert got == tile, (bits, crop, ssw, loc, got, tile)
    assert got[2] > 99.9
    want = oracle.ssimulacra2(vh.yuv_to_linear_rgbs(ref, bits, ssw, 0, 1, loc), vh.yuv_to_linear_rgbs(dis, bits, ssw, 0, 1, loc))
    assert got[0] == pytest.approx(want, abs=TOL)


@pytest.mark.parametrize("key", sorted(fx.ref_goldens()["yuv"]["ssimulacra2"]))
def test_reference_goldens_from_raw_yuv(dev, key):
    """The reference's own YUV keys, computed by the GPU from the YUV planes (the fixture carries `_Matrix = 1`,
    which VapourSynth's resize prefers over toRGBS's matrix_in)."""
    g = fx.ref_goldens()["yuv"]["ssimulacra2"][key]
    fmt_name, geometry, d = key.split("|")
    bits = 8 if fmt_name.endswith("P8") else 16
    ref = fx.yuv_geometry(fx.crop_yuv(bits), geometry)
    kind = d.split("=")[1]
    if kind.startswith("blur"):
        dis = [vh.std_boxblur(p, int(kind[4:]), int(kind[4:])) for p in ref]
    else:
        h, w = ref[0].shape
        dis = vh.resize_yuv_int(vh.resize_yuv_int(ref, bits, w * 2, h * 2), bits, w, h)
    got = dev.ssimulacra2_src(_fmt(dev, ref, bits, 1, 1), [dev.upload(p) for p in ref], [dev.upload(p) for p in dis])[0]
    assert got == pytest.approx(g, rel=5e-3 if geometry == "tiny" else 2e-4), (key, got, g, got / g - 1)


def test_yuv_batch_identity_and_determinism(dev, oracle):
    ref = _clip(8, 1, 1)
    d1 = [vh.std_boxblur(p, 1, 1) for p in ref]
    d3 = [vh.std_boxblur(p, 3, 3) for p in ref]
    up = lambda ps: [dev.upload(p) for p in ps]
    fmt = _fmt(dev, ref, 8, 1, 1)
    got = dev.ssimulacra2_src(fmt, up(ref) * 4, up(d1) + up(d3) + up(ref) + up(d1))
    lin = lambda ps: vh.yuv_to_linear_rgbs(ps, 8, 1, 1, 1, 0)
    assert got[0] == pytest.approx(oracle.ssimulacra2(lin(ref), lin(d1)), abs=TOL)
    assert got[1] == pytest.approx(oracle.ssimulacra2(lin(ref), lin(d3)), abs=TOL)
    assert got[2] > 99.9 and got[0] > got[1] and got[3] == got[0]
    c = [np.full((64, 64), 120, np.uint8), np.full((32, 32), 90, np.uint8), np.full((32, 32), 160, np.uint8)]
    assert dev.ssimulacra2_src(fmt, up(c), up(c))[0] == 100.0  # reference tests/test_ssimulacra2.py:65-67 (a YUV420 BlankClip)


def test_yuv_fused_equals_prestage_then_score_at_1080p(dev):
    """Full size: the fused pass against the two-step route (pre-stage kernel, then the linear-RGBS entry point) on a
    1080p YUV420P8 pair — the same f32 planes by construction, so the scores are identical."""
    h, w = 1080, 1920
    rgb = np.stack([fx.tiled_natural((h, w), np.uint8, c) for c in range(3)])
    ref = [np.ascontiguousarray(p) for p in vh.rgb24_to_yuv(rgb, 8, 1, 1)]
    dis = [vh.std_boxblur(p, 2, 2) for p in ref]
    fmt = dev.ssim_source("YUV", np.uint8, 8, True, ssw=1, ssh=1, matrix=1, chroma_loc=0)
    fused = dev.ssimulacra2_src(fmt, [dev.upload(p) for p in ref], [dev.upload(p) for p in dis])[0]
    a = dev.to_rgbs_linear(fmt, [dev.upload(p) for p in ref])
    b = dev.to_rgbs_linear(fmt, [dev.upload(p) for p in dis])
    assert fused == dev.ssimulacra2(a, b)[0]
    want = vh.yuv_to_linear_rgbs(ref, 8, 1, 1, 1, 0)
    got = [dev.download(p) for p in a]
    for c in range(3):
        assert np.array_equal(got[c].view(np.uint32), want[c].view(np.uint32))
