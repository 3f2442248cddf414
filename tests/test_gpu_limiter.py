"""GPU parity: vszip_limiter vs the CPU oracle, bit-exact for every sample type, plus the
reference's goldens and known answers (reference tests/test_limiter.py)."""
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


def _run(dev, planes, lo, hi, align=32):
    ds = [dev.upload(np.ascontiguousarray(p), align) for p in planes]
    dd = [dev.empty(p.shape[0], p.shape[1], p.dtype, align) for p in planes]
    dev.limiter(ds, dd, lo, hi)
    return [dev.download(d) for d in dd]


@pytest.mark.parametrize("dtype,lo,hi", [(np.uint8, 40, 200), (np.uint16, 10000, 50000), (np.uint32, 268435456, 3942645760),
                                          (np.float32, 0.1, 0.7), (np.float16, 0.2, 0.8), (np.float32, -0.4, 0.4)])
@pytest.mark.parametrize("shape,align", [((33, 70), 1), ((270, 481), 32), ((64, 256), 8), ((7, 13), 1)])
def test_matches_oracle(dev, oracle, dtype, lo, hi, shape, align):
    a = fx.splitmix64_plane(3, shape, dtype) if np.dtype(dtype) != np.uint32 else \
        (fx.splitmix64_plane(3, shape, np.uint16).astype(np.uint32) << 16 | fx.splitmix64_plane(4, shape, np.uint16))
    if np.dtype(dtype).kind == "f":
        a = (a.astype(np.float32) * 2 - 0.5).astype(dtype)  # values on both sides of the window
        a[0, 0] = np.nan  # @max/@min return the non-NaN operand: a NaN sample becomes lo
    (got,) = _run(dev, [a], [lo], [hi], align)
    want = oracle.limiter(a, lo, hi)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dtype, shape, align)


def test_reference_goldens_and_known_answers(dev, oracle):
    g = fx.ref_goldens()["exact"]["limiter"]
    out = _run(dev, list(fx.crop_rgb24()), [20, 20, 100], [180, 200, 250])
    for p in range(3):
        st = fx.plane_stats(out[p])
        for k in ("avg", "min", "max"):
            assert st[k] == pytest.approx(g["RGB24|full|max=[180,200,250],min=[20,20,100]"][f"p{p}"][k], rel=1e-6, abs=1e-9)
    lo, hi = oracle.limiter_default_range(False, 8, False, True)
    out = _run(dev, list(fx.crop_rgb24()), lo, hi)
    for p in range(3):
        assert fx.plane_stats(out[p])["min"] >= 16 and fx.plane_stats(out[p])["max"] <= 235
        assert fx.plane_stats(out[p])["avg"] == pytest.approx(g["RGB24|full|tv_range=1"][f"p{p}"]["avg"], rel=1e-6)
    ramp = np.tile(np.arange(256, dtype=np.uint8), (2, 1))  # reference tests/test_limiter.py:136-142
    assert _run(dev, [ramp], [10], [200])[0][0].tolist() == [min(max(x, 10), 200) for x in range(256)]


def test_many_planes_and_errors(dev, oracle):
    import vszip_amd

    rng = np.random.default_rng(9)
    planes = [rng.integers(0, 65536, size=(20 + i % 5, 64 + 8 * (i % 4)), dtype=np.uint16) for i in range(100)]
    lo = [1000 + 10 * i for i in range(100)]
    hi = [60000 - 10 * i for i in range(100)]
    out = _run(dev, planes, lo, hi)
    for i in (0, 47, 48, 99):
        assert np.array_equal(out[i], oracle.limiter(planes[i], lo[i], hi[i]))
    with pytest.raises(vszip_amd.VszipError, match="min value must be less than or equal to max value"):
        _run(dev, planes[:1], [10], [5])


def test_adaptive_binarize_matches_oracle(dev, oracle):
    """vszip_adaptive_binarize (src/vapoursynth/adaptive_binarize.zig:26-73) on aligned and tight strides."""
    for shape, align in (((61, 203), 1), ((128, 512), 32), ((5, 9), 1)):
        a = fx.splitmix64_plane(31, shape, np.uint8)
        b = fx.splitmix64_plane(32, shape, np.uint8)
        for c in (3, 0, -5, 255, -255, 256, -256, 300):
            da, db = dev.upload(a, align), dev.upload(b, align)
            dd = dev.empty(shape[0], shape[1], np.uint8, align)
            dev.adaptive_binarize([da], [db], [dd], c)
            assert np.array_equal(dev.download(dd), oracle.adaptive_binarize(a, b, c)), (shape, c)
