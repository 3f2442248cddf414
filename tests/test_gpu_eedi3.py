"""GPU parity: vszip_eedi3 vs the CPU oracle, bit-exact (the discrete Viterbi path makes any
arithmetic deviation visible), plus the reference's own goldens and the EEDI3H identity
(reference tests/test_eedi3.py:23-62,111-118)."""
import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import vszip_amd

    d = vszip_amd.Device(0)
    yield d
    d.close()


def _gpu(dev, src, field, **kw):
    s = dev.upload(np.ascontiguousarray(src, np.float32))
    (d,) = dev.eedi3([s], field, **kw)
    return dev.download(d)


CASES = [
    dict(field=1), dict(field=0), dict(field=1, dh=True), dict(field=0, dh=True),
    dict(field=1, vcheck=0), dict(field=1, vcheck=1), dict(field=1, vcheck=3),
    dict(field=1, mdis=31, nrad=3), dict(field=1, mdis=5, nrad=0), dict(field=1, gamma=0.0),
    dict(field=1, alpha=0.4, beta=0.3, gamma=40.0), dict(field=1, alpha=0.9, beta=0.05, gamma=2.0, mdis=30),
    # the default mdis = 20 instantiation (window sums in registers through DPP, round 4) at every nrad: 0 has no shift, odd ones store their sums unaligned
    dict(field=1, nrad=0), dict(field=0, nrad=1, dh=True), dict(field=1, nrad=3, vcheck=1), dict(field=0, nrad=3, alpha=0.1, beta=0.9, gamma=0.0),
    dict(field=1, mdis=20, nrad=2, vcheck=3, vthresh0=8.0, vthresh1=4.0, vthresh2=1.0),
    # mdis below the default: the default's layout with the directions beyond mdis masked out (whole passes skipped, a pass cut in the middle, one direction)
    dict(field=1, mdis=1), dict(field=0, mdis=3, nrad=1), dict(field=1, mdis=4, dh=True), dict(field=1, mdis=8, nrad=3), dict(field=0, mdis=12, gamma=0.0),
    dict(field=1, mdis=17, nrad=0, vcheck=1), dict(field=1, mdis=19, alpha=0.5, beta=0.1),
]


@pytest.mark.parametrize("kw", CASES, ids=[",".join(f"{k}={v}" for k, v in c.items()) for c in CASES])
def test_matches_oracle(dev, oracle, kw):
    src = fx.crop_rgbs()[1][:120, :333]
    kw = dict(kw)
    field = kw.pop("field")
    got = _gpu(dev, src, field, **kw)
    want = oracle.eedi3(src, field, **kw)
    assert np.array_equal(got, want), (kw, int((got != want).sum()))


@pytest.mark.parametrize("scale", [1e30, 3e37, 1e-30], ids=["1e30", "3e37", "1e-30"])
def test_costs_beyond_the_sentinel_and_denormals(dev, oracle, scale):
    """Samples whose costs overflow (the reference clamps a path cost at 0.9 * FLT_MAX, eedi3.zig:536-548: the Viterbi step's v_min against that bound
    and its +inf edge lanes) or are denormal: the path and the interpolation still equal the oracle's."""
    src = (fx.crop_rgbs()[1][:64, :200] * np.float32(scale)).astype(np.float32)
    assert np.isfinite(src).all()
    for kw in (dict(), dict(dh=True, vcheck=0)):
        got = _gpu(dev, src, 1, **kw)
        want = oracle.eedi3(src, 1, **kw)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (scale, kw, int((got.view(np.uint32) != want.view(np.uint32)).sum()))


GENERAL = [
    dict(field=1, hp=True), dict(field=0, hp=True, dh=True), dict(field=1, hp=True, vcheck=0), dict(field=1, hp=True, vcheck=3, mdis=10, nrad=1),
    dict(field=1, hp=True, mdis=40, nrad=3), dict(field=1, mdis=40, nrad=3), dict(field=1, mdis=40, nrad=0), dict(field=0, mdis=33, gamma=0.0, vcheck=1),
    dict(field=1, hp=True, mdis=7, nrad=0, alpha=0.4, beta=0.3, gamma=40.0),
]


@pytest.mark.parametrize("kw", GENERAL, ids=[",".join(f"{k}={v}" for k, v in c.items()) for c in GENERAL])
def test_general_kernel_matches_oracle(dev, oracle, kw):
    """hp=True (half-pel directions, +-2 transitions) and mdis up to 40 go through the general line
    kernel (reference tests/test_eedi3.py: hp=1, mdis=40 cases)."""
    src = fx.crop_rgbs()[1][:96, :401]
    kw = dict(kw)
    field = kw.pop("field")
    got = _gpu(dev, src, field, **kw)
    want = oracle.eedi3(src, field, **kw)
    assert np.array_equal(got, want), (kw, int((got != want).sum()))


@pytest.mark.parametrize("hp", [False, True])
@pytest.mark.parametrize("horizontal", [False, True])
def test_mclip(dev, oracle, hp, horizontal):
    """mclip: only pixels with a mask sample within +-mdis are connected (buildBmask), an all-zero
    mask line falls back to the vertical cubic."""
    src = np.ascontiguousarray(fx.crop_rgbs()[0][:90, :300])
    rng = np.random.default_rng(5)
    mask = (rng.random(src.shape) < 0.01).astype(np.uint8) * 255
    mask[10:20] = 0          # lines without any sample
    mask[:, 120:220] = 0     # a wide unmasked band
    s = dev.upload(src)
    m = dev.upload(mask)
    (d,) = dev.eedi3([s], 1, hp=hp, horizontal=horizontal, mclips=[m])
    want = oracle.eedi3(src, 1, hp=hp, horizontal=horizontal, mclip=mask)
    got = dev.download(d)
    assert np.array_equal(got, want), int((got != want).sum())
    (d0,) = dev.eedi3([s], 1, hp=hp, horizontal=horizontal, mclips=[dev.upload(np.zeros_like(mask))])
    assert np.array_equal(dev.download(d0), oracle.eedi3(src, 1, hp=hp, horizontal=horizontal, mclip=np.zeros_like(mask)))


def _mclip_masks(shape, rng):
    h, w = shape
    sparse = (rng.random(shape) < 0.01).astype(np.uint8) * 255
    sparse[h // 4:h // 4 + 6] = 0
    dense = (rng.random(shape) < 0.9).astype(np.uint8) * 255
    edges = np.zeros(shape, np.uint8)
    edges[:, : min(3, w)] = 255                      # only the first columns (the rest of a long line: blocks without any column)
    far = np.zeros(shape, np.uint8)
    far[:, w - 1] = 255                              # only the last column
    far[::3] = 0
    blocks = np.zeros(shape, np.uint8)               # whole stretches without a sample: the tuned kernel skips those blocks' costs
    for x0 in range(150, w, 300):
        blocks[:, x0:x0 + 7] = 255
    first_off = np.full(shape, 255, np.uint8)
    first_off[:, : min(40, w)] = 0                   # column 1 outside bmask on some lines (:494-496) and inside on others
    first_off[1::2, : min(22, w)] = 255
    return dict(sparse=sparse, dense=dense, ones=np.full(shape, 255, np.uint8), first=edges, last=far, blocks=blocks, first_off=first_off)


@pytest.mark.parametrize("geom", [(40, 401), (24, 64), (24, 65), (20, 130), (16, 33), (12, 700)], ids=lambda g: f"{g[1]}x{g[0]}")
@pytest.mark.parametrize("kw", [dict(field=1), dict(field=0, dh=True), dict(field=1, mdis=12, nrad=1), dict(field=0, mdis=5, nrad=3, vcheck=0), dict(field=1, mdis=19, gamma=0.0, vcheck=3)],
                         ids=lambda c: ",".join(f"{k}={v}" for k, v in c.items()))
def test_mclip_on_the_tuned_line_kernel(dev, oracle, geom, kw):
    """Round 6 (late): mclip with mdis <= 20 runs on the tuned line kernel (eedi3_line_kernel<..., MCLIP>): bmask words a block on the scalar unit, a column outside
    bmask repeats the codes before it, a block without any column in bmask skips its costs. Masks that exercise each rule, widths around the 64-column blocks, the
    mdis = 20 and the mdis < 20 layouts; and mdis = 25 calls, which the general kernel serves - all bit-exact against the oracle."""
    h, w = geom
    src = np.ascontiguousarray(np.tile(fx.crop_rgbs()[2], (1, 3))[:h, :w])
    kw = dict(kw)
    field = kw.pop("field")
    s = dev.upload(src)
    for name, mask in _mclip_masks((h, w), np.random.default_rng(h * 1000 + w)).items():
        want = oracle.eedi3(src, field, mclip=mask, **kw)
        (d,) = dev.eedi3([s], field, mclips=[dev.upload(mask)], **kw)
        got = dev.download(d)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, geom, kw, int((got.view(np.uint32) != want.view(np.uint32)).sum()))
        if name in ("sparse", "blocks") and "mdis" not in kw:  # the general kernel's masked path keeps its own check: mdis > 20 sends an mclip call there
            (d,) = dev.eedi3([s], field, mclips=[dev.upload(mask)], mdis=25, **kw)
            want25 = oracle.eedi3(src, field, mclip=mask, mdis=25, **kw)
            assert np.array_equal(dev.download(d).view(np.uint32), want25.view(np.uint32)), ("general kernel", name, geom, kw)


def test_mclip_on_some_planes_of_a_call(dev, oracle):
    """planes with and without an mclip in one call (a YUV clip whose mclip is Gray: the plugin passes the mask for the luma only), and EEDI3H's transposed masks"""
    rng = np.random.default_rng(9)
    planes = [np.ascontiguousarray(fx.crop_rgbs()[i][:60 + 10 * i, :200 + 37 * i]) for i in range(3)]
    masks = [(rng.random(planes[0].shape) < 0.02).astype(np.uint8) * 255, None, (rng.random(planes[2].shape) < 0.3).astype(np.uint8) * 255]
    for horizontal in (False, True):
        ds = dev.eedi3([dev.upload(p) for p in planes], 1, dh=True, horizontal=horizontal, mclips=[dev.upload(m) if m is not None else None for m in masks])
        for i, p in enumerate(planes):
            want = oracle.eedi3(p, 1, dh=True, horizontal=horizontal, mclip=masks[i])
            assert np.array_equal(dev.download(ds[i]).view(np.uint32), want.view(np.uint32)), (horizontal, i)


def test_horizontal_matches_oracle_and_identity(dev, oracle):
    src = fx.crop_rgbs()[0][:150, :200]
    for dh in (False, True):
        got = _gpu(dev, src, 1, dh=dh, horizontal=True)
        assert np.array_equal(got, oracle.eedi3(src, 1, dh=dh, horizontal=True))
        vert = _gpu(dev, np.ascontiguousarray(src.T), 1, dh=dh)
        assert np.array_equal(got, vert.T)


def test_sclip(dev, oracle):
    src = fx.crop_rgbs()[2][:96, :160]
    sc = fx.splitmix64_plane(9, (96, 160), np.float32)
    s, c = dev.upload(src), dev.upload(sc)
    (d,) = dev.eedi3([s], 1, sclips=[c])
    assert np.array_equal(dev.download(d), oracle.eedi3(src, 1, sclip=sc))


def test_reference_goldens(dev):
    g = fx.ref_goldens()["exact"]
    planes = [dev.upload(np.ascontiguousarray(p)) for p in fx.crop_rgbs()]
    for key, horiz in (("eedi3", False), ("eedi3h", True)):
        outs = dev.eedi3(planes, 1, horizontal=horiz)
        for p in range(3):
            st = fx.plane_stats(dev.download(outs[p]))
            for k in ("avg", "min", "max"):
                assert st[k] == pytest.approx(g[key]["RGBS|full|field=1"][f"p{p}"][k], rel=1e-6, abs=1e-9)


def test_errors(dev):
    import vszip_amd

    s = dev.upload(np.zeros((64, 128), np.float32))
    for kw, msg in ((dict(alpha=0.8, beta=0.5), "alpha \\+ beta"), (dict(nrad=4), "nrad must be"), (dict(mdis=0), "mdis must be"), (dict(vcheck=4), "vcheck must be")):
        with pytest.raises(vszip_amd.VszipError, match=msg):
            dev.eedi3([s], 1, **kw)
    odd = dev.upload(np.zeros((63, 128), np.float32))
    with pytest.raises(vszip_amd.VszipError, match="height must be mod 2"):
        dev.eedi3([odd], 1)


@pytest.mark.parametrize("w", [4097, 7680, 8200, 16500])
def test_vcheck_on_wide_lines(dev, oracle, w):
    """Lines wider than the pipelined vcheck kernels serve (8K frames) take the wide-line kernel;
    beyond 8192 samples the blended line pair lives in global scratch instead of LDS."""
    rng = np.random.default_rng(w)
    src = rng.random((16, w)).astype(np.float32)
    for kw in (dict(vcheck=2), dict(vcheck=3, dh=True)):
        got = _gpu(dev, src, 1, **kw)
        assert np.array_equal(got, oracle.eedi3(src, 1, **kw)), kw


@pytest.mark.parametrize("w", [1919, 1921, 2047, 2048, 2049, 2160, 2559, 2560, 2561, 3000, 3840, 4095, 4096])
def test_vcheck_chain_kernels_by_line_width(dev, oracle, w):
    """The vertical-consistency chain runs from LDS rings with two columns a thread up to 2048 samples a line, three up to 2560 (round 4: the second pass
    of a 2x upscale of 1080p is 2160 wide), four up to 4096 (the luma of a 4K frame), from global memory beyond (test_vcheck_on_wide_lines): every width
    around the switches, enough lines for several turns of the six-line ring period, with an sclip and half-pel directions too."""
    rng = np.random.default_rng(w)
    src = fx.tiled_natural((48, w), np.float32, 1) + rng.random((48, w)).astype(np.float32) * np.float32(0.05)
    sc = rng.random((96, w)).astype(np.float32)
    for kw in (dict(), dict(vcheck=3, dh=True), dict(vcheck=1, hp=True, mdis=6)):
        got = _gpu(dev, src, 1, **kw)
        assert np.array_equal(got, oracle.eedi3(src, 1, **kw)), (w, kw)
    s, c = dev.upload(np.ascontiguousarray(src)), dev.upload(sc)
    (d,) = dev.eedi3([s], 0, dh=True, sclips=[c])
    assert np.array_equal(dev.download(d), oracle.eedi3(src, 0, dh=True, sclip=sc)), w


@pytest.mark.parametrize("w", [5, 8, 17, 31, 43, 45, 63])
@pytest.mark.parametrize("kw", [dict(), dict(hp=True), dict(mdis=40, nrad=3), dict(mdis=31, nrad=0, vcheck=3), dict(dh=True)],
                         ids=["default", "hp", "mdis40", "mdis31", "dh"])
def test_short_lines(dev, oracle, w, kw):
    """Lines shorter than the direction reach 2*mdis+nrad: every direction the reference evaluates at
    x is bounded by min(x, w-1-x), so the result stays defined; also the horizontal variant on a
    short height."""
    src = np.ascontiguousarray(fx.crop_rgbs()[0][40:72, 100:100 + w])
    got = _gpu(dev, src, 1, **kw)
    want = oracle.eedi3(src, 1, **kw)
    assert np.array_equal(got, want), (w, kw, int((got != want).sum()))
    if not kw.get("dh"):
        srct = np.ascontiguousarray(src.T)
        if srct.shape[1] % 2 == 0:
            got = _gpu(dev, srct, 0, horizontal=True, **kw)
            assert np.array_equal(got, oracle.eedi3(srct, 0, horizontal=True, **kw))


@pytest.mark.parametrize("kw", [dict(dh=True), dict(), dict(horizontal=True), dict(vcheck=3, nrad=1)], ids=["dh", "plain", "eedi3h", "vcheck3"])
def test_two_plane_heights_overlap_equals_sequential_and_oracle(dev, oracle, kw, monkeypatch):
    """YUV 4:2:0 batches: the tall planes' vertical-consistency chains run beside the short planes' line kernel (second,
    CU-masked stream; plane slots reordered tall-first). Same bits as the sequential order, and as the oracle, with the
    planes handed over in frame order (Y, U, V, Y, U, V)."""
    shapes = [(96, 160), (48, 80), (48, 80)] * 2
    planes = [np.ascontiguousarray(fx.tiled_natural(s, np.float32, i % 3)) + np.float32(0.01 * (i // 3)) for i, s in enumerate(shapes)]
    srcs = [dev.upload(p) for p in planes]
    dev.set_option("VSZIP_EEDI3_FORCE_OVERLAP", 1)  # (the library overlaps from 12 frames per call on: below that it does not pay)
    a = [dev.download(d) for d in dev.eedi3(srcs, 1, **kw)]
    dev.set_option("VSZIP_EEDI3_FORCE_OVERLAP", 0)
    dev.set_option("VSZIP_EEDI3_NO_OVERLAP", 1)
    b = [dev.download(d) for d in dev.eedi3(srcs, 1, **kw)]
    dev.set_option("VSZIP_EEDI3_NO_OVERLAP", 0)
    okw = {k: v for k, v in kw.items() if k != "horizontal"}
    for p, x, y in zip(planes, a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        # EEDI3H == T . EEDI3 . T (src/vapoursynth/eedi3.zig:220-246)
        want = np.ascontiguousarray(oracle.eedi3(np.ascontiguousarray(p.T), 1, **okw).T) if kw.get("horizontal") else oracle.eedi3(p, 1, **okw)
        assert np.array_equal(x.view(np.uint32), want.view(np.uint32))
