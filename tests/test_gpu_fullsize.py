"""GPU parity at BASELINE.json's full sizes: the HIP path against the
CPU oracle on whole frames where the oracle finishes in seconds, and size-independent
identities (pass composition, EEDI3H == T.EEDI3.T, determinism) where it does not."""
import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu

W4K, H4K = 3840, 2160
W1080, H1080 = 1920, 1080


@pytest.fixture(scope="module")
def dev():
    import vszip_amd

    d = vszip_amd.Device(0)
    yield d
    d.close()


def _yuv420(shape):
    h, w = shape
    return [(h, w), (h // 2, w // 2), (h // 2, w // 2)]


def _lin(v):
    v = v.astype(np.float64)
    return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4).astype(np.float32)


def _boxblur(dev, planes, *args):
    srcs = [dev.upload(p) for p in planes]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
    dev.boxblur(srcs, dsts, *args)
    return [dev.download(d) for d in dsts], dsts


@pytest.mark.parametrize("content", ["noise", "natural"])
def test_boxblur_r13_4k_yuv420p16(dev, oracle, content):
    """config 1 at the north-star size: BoxBlur hradius=vradius=13 on 3840x2160 YUV420P16, bit-exact."""
    if content == "noise":
        planes = [fx.splitmix64_plane(0x5A170000 + p, s, np.uint16) for p, s in enumerate(_yuv420((H4K, W4K)))]
    else:
        planes = [fx.tiled_natural(s, np.uint16, p) for p, s in enumerate(_yuv420((H4K, W4K)))]
    got, _ = _boxblur(dev, planes, 13, 1, 13, 1)
    for g, p in zip(got, planes):
        assert np.array_equal(g, oracle.boxblur(p, 13, 1, 13, 1))


@pytest.mark.parametrize("dtype,args,frames,shape", [(np.uint16, (13, 5, 13, 5), 16, (H1080, W1080)), (np.uint16, (5, 3, 5, 3), 8, (H4K, W4K)), (np.uint8, (3, 2, 3, 6), 16, (H1080, W1080)),
                                                     (np.float32, (5, 3, 5, 3), 8, (H4K, W4K)), (np.float16, (2, 2, 2, 2), 4, (H4K, W4K))])
def test_boxblur_runtime_passes_batches_take_the_pass_chains(dev, oracle, dtype, args, frames, shape, monkeypatch):
    """Calls large enough for the pass-chain kernels' own conditions (900+ column groups for integer planes' vertical passes, 12 000+ rows for float planes' horizontal
    ones) — the sizes the bench legs run at: the default paths bit-identical to one launch per pass on every plane of the batch, and to the oracle on the first frame."""
    import os

    base = [fx.tiled_natural(s, dtype, p) for p, s in enumerate(_yuv420(shape))]
    planes = [np.ascontiguousarray(np.roll(p, 7 * f, axis=1)) for f in range(frames) for p in base]
    got, _ = _boxblur(dev, planes, *args)
    with dev.options(VSZIP_RT_NO_FCHAIN=1, VSZIP_RT_NO_ICHAIN=1):
        per_pass, _ = _boxblur(dev, planes, *args)
    assert dev.get_option("VSZIP_RT_NO_ICHAIN") == 0 and "VSZIP_RT_NO_ICHAIN" not in os.environ
    for i, (a, b) in enumerate(zip(got, per_pass)):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (i, args, np.argwhere(a != b)[:3].tolist())
    for a, p in zip(got[:3], planes[:3]):
        assert np.array_equal(a.view(np.uint8), oracle.boxblur(p, *args).view(np.uint8))


@pytest.mark.parametrize("dtype,r", [(np.float32, 13), (np.float16, 13), (np.float32, 22), (np.float32, 3)])
def test_boxblur_ct_float_4k_yuv420(dev, oracle, dtype, r):
    """The float CT path at 4K (register-ring kernel over several column tiles and bands that end inside a ring
    period, the bottom / left / right strips through the tile kernel): every byte equals the oracle's."""
    planes = [fx.splitmix64_plane(0x5A170100 + p, s, dtype) for p, s in enumerate(_yuv420((H4K, W4K)))]
    got, _ = _boxblur(dev, planes, r, 1, r, 1)
    for g, p in zip(got, planes):
        want = oracle.boxblur(p, r, 1, r, 1)
        bad = g.view(np.uint8) != want.view(np.uint8)
        assert not bad.any(), (dtype, r, p.shape, np.argwhere(bad)[:3].tolist())


def test_boxblur_pass_composition_4k(dev):
    """BoxBlur(h=7, hpasses=2) == BoxBlur(h=7) o BoxBlur(h=7) and BoxBlur(4, 9) == v9 o h4 (reference
    tests/test_boxblur.py:86-101), exact, on a 4K u16 plane."""
    p = fx.splitmix64_plane(77, (H4K, W4K), np.uint16)
    (two,), _ = _boxblur(dev, [p], 7, 2, 0, 0)
    (one,), _ = _boxblur(dev, [p], 7, 1, 0, 0)
    (again,), _ = _boxblur(dev, [one], 7, 1, 0, 0)
    assert np.array_equal(two, again)
    (hv,), _ = _boxblur(dev, [p], 4, 1, 9, 1)
    (h4,), _ = _boxblur(dev, [p], 4, 1, 0, 0)
    (v9,), _ = _boxblur(dev, [h4], 0, 0, 9, 1)
    assert np.array_equal(hv, v9)


@pytest.mark.parametrize("shape", [(H1080, W1080), (H4K, W4K)], ids=["1080p", "4k"])
def test_bilateral_yuv420p16(dev, oracle, shape):
    """config 2 (and its 4K variant): Bilateral sigmaS=2 sigmaR=2 on YUV420P16, bit-exact."""
    planes = [fx.tiled_natural(s, np.uint16, p) for p, s in enumerate(_yuv420(shape))]
    cfg = dev.bilateral_cfg([2], [2], yuv=True, ssw=1, ssh=1, hist_len=65536)
    srcs = [dev.upload(p) for p in planes]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
    dev.bilateral(srcs, dsts, cfg, [0, 1, 2])
    prm = oracle.bilateral_params([2], [2], yuv=True, ssw=1, ssh=1)
    for i, (d, p) in enumerate(zip(dsts, planes)):
        want = oracle.bilateral_plane(p, prm["sigmaS"][i], prm["sigmaR"][i], prm["algorithm"][i], prm["radius"][i], prm["step"][i], prm["PBFICnum"][i])
        assert np.array_equal(dev.download(d), want), i
    dev.bilateral_free(cfg)


def test_ssimulacra2_4k_rgbs(dev, oracle):
    """config 3: SSIMULACRA2 ref vs dist on 3840x2160 linear RGBS (dist = ref + noise sigma 0.02).
    Float result: bound 1e-5 (north star); the f64 pooling order is the only difference, asserted 1e-7."""
    rng = np.random.default_rng(1)
    ref = [_lin(fx.tiled_natural((H4K, W4K), np.float32, p)) for p in range(3)]
    dis = [np.clip(p + rng.normal(0, 0.02, p.shape).astype(np.float32), 0, 1).astype(np.float32) for p in ref]
    r = [dev.upload(p, 1) for p in ref]
    d = [dev.upload(p, 1) for p in dis]
    (got,) = dev.ssimulacra2(r, d)
    want = oracle.ssimulacra2(ref, dis)
    assert got == pytest.approx(want, abs=1e-7), (got, want)
    # identical inputs: the reference's a*a + b*b plane (addSquare, ssimulacra2.zig:228-245) is not
    # 2*(a*b) in f32, so a large natural frame scores just under 100 (99.9559 here) — in the
    # oracle and on the GPU alike
    (same,) = dev.ssimulacra2(r, r)
    assert same == pytest.approx(oracle.ssimulacra2(ref, ref), abs=1e-7) and 99.9 < same <= 100.0


def test_eedi3_1080p_double_height(dev, oracle):
    """config 4: EEDI3 field=1 dh=1, 1920x1080 -> 1920x2160 float luma, bit-exact; and the
    EEDI3H == transpose o EEDI3 o transpose identity at the same size."""
    src = np.ascontiguousarray(fx.tiled_natural((H1080, W1080), np.float32, 0))
    s = dev.upload(src)
    (d,) = dev.eedi3([s], 1, dh=True)
    got = dev.download(d)
    assert got.shape == (2 * H1080, W1080)
    assert np.array_equal(got, oracle.eedi3(src, 1, dh=True))
    st = dev.upload(np.ascontiguousarray(src.T))
    (dh_,) = dev.eedi3([st], 1, dh=True, horizontal=True)
    assert np.array_equal(dev.download(dh_), got.T)


def _pipeline_gpu(dev, rgb):
    """Bilateral(sigmaS=2, sigmaR=2) -> BoxBlur(r=2) -> SSIMULACRA2(original, processed), planes stay on the device."""
    cfg = dev.bilateral_cfg([2], [2], yuv=False, ssw=0, ssh=0, hist_len=65536)
    srcs = [dev.upload(p, 1) for p in rgb]
    mid = [dev.empty(p.shape[0], p.shape[1], p.dtype, 1) for p in rgb]
    out = [dev.empty(p.shape[0], p.shape[1], p.dtype, 1) for p in rgb]
    dev.bilateral(srcs, mid, cfg, [0, 1, 2])
    dev.boxblur(mid, out, 2, 1, 2, 1)
    (score,) = dev.ssimulacra2(srcs, out)
    dev.bilateral_free(cfg)
    return score, out


def test_pipeline_matches_oracle_540p(dev, oracle):
    """config 5's chain at a size the oracle finishes quickly: every stage bit-exact, score to 1e-7."""
    rgb = [_lin(fx.tiled_natural((540, 960), np.float32, p)) for p in range(3)]
    score, out = _pipeline_gpu(dev, rgb)
    prm = oracle.bilateral_params([2], [2], yuv=False, ssw=0, ssh=0)
    want = []
    for i, p in enumerate(rgb):
        b = oracle.bilateral_plane(p, prm["sigmaS"][i], prm["sigmaR"][i], prm["algorithm"][i], prm["radius"][i], prm["step"][i], prm["PBFICnum"][i])
        want.append(oracle.boxblur(b, 2, 1, 2, 1))
    for o, w in zip(out, want):
        assert np.array_equal(dev.download(o), w)
    assert score == pytest.approx(oracle.ssimulacra2(rgb, want), abs=1e-7)


def test_pipeline_8k_rgbs_device_resident(dev):
    """config 5 at 7680x4320 RGBS: the chain runs with every intermediate resident in HBM and is
    deterministic run to run (pixels and score)."""
    rgb = [_lin(fx.tiled_natural((4320, 7680), np.float32, p)) for p in range(3)]
    s1, out1 = _pipeline_gpu(dev, rgb)
    first = [dev.download(o) for o in out1]
    s2, out2 = _pipeline_gpu(dev, rgb)
    assert s1 == s2 and np.isfinite(s1) and s1 < 100.0  # SSIMULACRA2 goes negative for heavy smoothing
    for a, o in zip(first, out2):
        assert np.array_equal(a, dev.download(o))
