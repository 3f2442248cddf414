"""Host staging modes of the C ABI (vszip_ctx_set_staging): copies through the context's pinned
arena must move exactly the bytes the direct copies move, for any pitch, across arena growth and
across several planes in flight before one sync."""
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


def _roundtrip(dev, a):
    d = dev.upload(a)
    return dev.download(d)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_pinned_arena_roundtrip_equals_direct(dev, dtype):
    a = fx.splitmix64_plane(3, (257, 1001), dtype)
    strided = np.zeros((257, 1100), dtype)[:, 17:1018]
    strided[...] = a
    try:
        dev.set_staging(1)
        assert np.array_equal(_roundtrip(dev, a).view(np.uint8), a.view(np.uint8))
        assert np.array_equal(_roundtrip(dev, strided).view(np.uint8), a.view(np.uint8))
    finally:
        dev.set_staging(0)
    assert np.array_equal(_roundtrip(dev, a).view(np.uint8), a.view(np.uint8))


def test_pinned_arena_many_planes_one_sync_and_growth(dev, oracle):
    """Several uploads, a kernel, several downloads, ONE sync; then a plane larger than the arena
    (64 MiB initially) forces a drain + regrow while copies are pending."""
    planes = [fx.splitmix64_plane(10 + i, (300 + i, 640), np.uint16) for i in range(4)]
    big = fx.splitmix64_plane(99, (4200, 8200), np.uint16)  # 68.9 MB
    try:
        dev.set_staging(1)
        srcs = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
        dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
        for s, p in zip(srcs, planes):
            dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, s.ptr, s.stride * 2, p.ctypes.data, p.strides[0], p.shape[1] * 2, p.shape[0]))
        dev.boxblur(srcs, dsts, 5, 1, 5, 1)
        outs = [np.full(p.shape, 0xABCD, np.uint16) for p in planes]
        for d, o in zip(dsts, outs):
            dev.check(dev.lib.vszip_copy_d2h_2d(dev.ctx, o.ctypes.data, o.strides[0], d.ptr, d.stride * 2, d.w * 2, d.h))
        # a copy larger than what is left of the arena drains the stream and lands the pending planes
        bd = dev.empty(big.shape[0], big.shape[1], big.dtype)
        dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, bd.ptr, bd.stride * 2, big.ctypes.data, big.strides[0], big.shape[1] * 2, big.shape[0]))
        dev.sync()
        for o, p in zip(outs, planes):
            assert np.array_equal(o, oracle.boxblur(p, 5, 1, 5, 1))
        assert np.array_equal(dev.download(bd), big)
        # abort: staged output is dropped, the destination keeps its old contents
        keep = np.full(planes[0].shape, 7, np.uint16)
        dev.check(dev.lib.vszip_copy_d2h_2d(dev.ctx, keep.ctypes.data, keep.strides[0], dsts[0].ptr, dsts[0].stride * 2, dsts[0].w * 2, dsts[0].h))
        dev.check(dev.lib.vszip_ctx_abort(dev.ctx))
        dev.sync()
        assert np.all(keep == 7)
    finally:
        dev.set_staging(0)
