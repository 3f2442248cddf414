"""GPU parity: vszip_xpsnr_wsse vs the CPU oracle — exact u64 equality (integer block sums
on the device, the reference's f64 weighting in its own block order on the host)."""
import math

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


def _frames(shape, dtype, depth, n=3, seed=0):
    rng = np.random.default_rng(seed)
    h, w = shape
    peak = (1 << depth) - 1
    out = []
    for f in range(n):
        y = np.roll(fx.tiled_natural((h, w), np.uint8), f * 3, axis=0).astype(np.int32) * (peak // 255)
        u = np.roll(fx.tiled_natural((h // 2, w // 2), np.uint8, 1), f, axis=1).astype(np.int32) * (peak // 255)
        v = np.roll(fx.tiled_natural((h // 2, w // 2), np.uint8, 2), f, axis=1).astype(np.int32) * (peak // 255)
        org = [np.clip(p, 0, peak).astype(dtype) for p in (y, u, v)]
        rec = [np.clip(p.astype(np.int32) + rng.integers(-3, 4, p.shape), 0, peak).astype(dtype) for p in org]
        out.append((org, rec))
    return out


@pytest.mark.parametrize("shape,dtype,depth", [((288, 352), np.uint8, 8), ((480, 640), np.uint8, 8), ((540, 960), np.uint16, 10),
                                               ((1080, 1920), np.uint8, 8), ((1200, 2100), np.uint16, 10)])
@pytest.mark.parametrize("fps,temporal", [(24, True), (60, True), (24, False)])
def test_matches_oracle(dev, oracle, shape, dtype, depth, fps, temporal):
    fr = _frames(shape, dtype, depth)
    for n, (org, rec) in enumerate(fr):
        p1 = fr[n - 1][0][0] if (temporal and n > 0) else None
        p2 = fr[n - 2][0][0] if (temporal and fps >= 32 and n > 1) else None
        want = oracle.xpsnr_wsse(org, rec, p1, p2, depth=depth, frame_rate=fps, temporal=temporal)
        d_org, d_rec = [dev.upload(p) for p in org], [dev.upload(p) for p in rec]
        got = dev.xpsnr_wsse(d_org, d_rec, dev.upload(p1) if p1 is not None else None, dev.upload(p2) if p2 is not None else None,
                             depth=depth, frame_rate=fps, temporal=temporal)
        assert got == want, (shape, n, fps, temporal, got, want)
        for c in range(3):
            a = dev.lib.vszip_xpsnr_value(got[c], org[c].shape[1], org[c].shape[0], depth)
            assert a == oracle.xpsnr_frame(want[c], org[c].shape[1], org[c].shape[0], depth)


def test_identical_is_inf(dev):
    """reference tests/test_xpsnr.py:222-225"""
    (org, _), = _frames((288, 352), np.uint8, 8, n=1)
    d = [dev.upload(p) for p in org]
    assert dev.xpsnr_wsse(d, d) == [0, 0, 0]
    assert math.isinf(dev.lib.vszip_xpsnr_value(0, 352, 288, 8))
