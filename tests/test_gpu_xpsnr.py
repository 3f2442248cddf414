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


def _rand_clip(rng, shapes, dtype, depth, n, full_range=False):
    peak = np.iinfo(dtype).max if full_range else (1 << depth) - 1
    frames = [[rng.integers(0, peak + 1, size=s, dtype=np.int64).astype(dtype) for s in shapes] for _ in range(n)]
    recs = [[np.clip(p.astype(np.int64) + rng.integers(-9, 10, p.shape), 0, peak).astype(dtype) for p in fr] for fr in frames]
    return frames, recs


def _want(oracle, frames, recs, depth, fps, temporal=True):
    return [oracle.xpsnr_wsse(frames[n], recs[n], frames[n - 1][0] if n >= 1 else None, frames[n - 2][0] if n >= 2 else None, depth=depth, frame_rate=fps,
                              temporal=temporal) for n in range(len(frames))]


# (luma shape, chroma subsampling (ssw, ssh) or None for Gray): plain 4:2:0, widths that are 2 mod 4, the
# highds regime (> 2048x1152) with ragged last blocks, 4:4:4 / 4:2:2 / 4:1:1 chroma, a one-plane clip,
# frames so small that the whole plane is one block (b < 4)
_GEOMS = [((288, 352), (1, 1)), ((290, 354), (1, 1)), ((1156, 2054), (1, 1)), ((1200, 2100), (0, 0)), ((486, 720), (1, 0)), ((480, 704), (2, 0)),
          ((240, 426), None), ((40, 44), (1, 1)), ((2160, 3840), (1, 1))]


@pytest.mark.parametrize("geom", range(len(_GEOMS)))
@pytest.mark.parametrize("dtype,depth", [(np.uint8, 8), (np.uint16, 10)])
def test_batch_matches_oracle(dev, oracle, geom, dtype, depth):
    """vszip_xpsnr_wsse_batch (strip kernel: one launch for the whole batch) == the oracle frame by
    frame, with the frame table inline (3 frames) and in device memory (10 frames)."""
    (h, w), ss = _GEOMS[geom]
    shapes = [(h, w)] + ([] if ss is None else [(h >> ss[1], w >> ss[0])] * 2)
    nf = 10 if h * w < 1e6 else 3
    rng = np.random.default_rng(9100 + geom)
    frames, recs = _rand_clip(rng, shapes, dtype, depth, nf)
    fps = 60 if geom % 2 else 24
    want = _want(oracle, frames, recs, depth, fps)
    dfr = [[dev.upload(p) for p in fr] for fr in frames]
    drc = [[dev.upload(p) for p in fr] for fr in recs]
    p1 = [dfr[n - 1][0] if n >= 1 else None for n in range(nf)]
    p2 = [dfr[n - 2][0] if n >= 2 else None for n in range(nf)]
    assert dev.xpsnr_wsse_batch(dfr, drc, p1, p2, depth=depth, frame_rate=fps) == want
    assert dev.xpsnr_wsse_batch(dfr[:3], drc[:3], p1[:3], p2[:3], depth=depth, frame_rate=fps) == want[:3]


def test_temporal_off_and_no_prev(dev, oracle):
    rng = np.random.default_rng(5)
    for shape in ((288, 352), (1200, 2100)):
        shapes = [shape, (shape[0] // 2, shape[1] // 2), (shape[0] // 2, shape[1] // 2)]
        frames, recs = _rand_clip(rng, shapes, np.uint8, 8, 2)
        dfr = [[dev.upload(p) for p in fr] for fr in frames]
        drc = [[dev.upload(p) for p in fr] for fr in recs]
        want_off = [oracle.xpsnr_wsse(frames[n], recs[n], None, None, depth=8, frame_rate=24, temporal=False) for n in range(2)]
        assert dev.xpsnr_wsse_batch(dfr, drc, None, None, depth=8, frame_rate=24, temporal=False) == want_off
        # temporal on but no previous frames at all (frame 0 of a clip): the previous frame counts as zeros
        want_np = [oracle.xpsnr_wsse(frames[n], recs[n], None, None, depth=8, frame_rate=60, temporal=True) for n in range(2)]
        assert dev.xpsnr_wsse_batch(dfr, drc, None, None, depth=8, frame_rate=60, temporal=True) == want_np


def test_block_kernels_agree(dev, oracle, monkeypatch):
    """The one-workgroup-per-block kernels (planes that miss the strip kernel's alignment rules) and
    the strip kernel give the same sums; an unaligned view takes the block kernels by itself."""
    rng = np.random.default_rng(77)
    shapes = [(486, 720), (243, 360), (243, 360)]
    frames, recs = _rand_clip(rng, shapes, np.uint16, 10, 3, full_range=True)
    want = _want(oracle, frames, recs, 10, 60)
    dfr = [[dev.upload(p) for p in fr] for fr in frames]
    drc = [[dev.upload(p) for p in fr] for fr in recs]
    p1 = [dfr[n - 1][0] if n >= 1 else None for n in range(3)]
    p2 = [dfr[n - 2][0] if n >= 2 else None for n in range(3)]
    strip = dev.xpsnr_wsse_batch(dfr, drc, p1, p2, depth=10, frame_rate=60)
    assert strip == want
    with dev.variant(VSZIP_XPSNR_BLOCKS=1) as built:  # (a development variant since round 5: the forcing switch, not the kernel the geometry rules route to)
        if built:
            assert dev.xpsnr_wsse_batch(dfr, drc, p1, p2, depth=10, frame_rate=60) == want


@pytest.mark.parametrize("shape,fps", [((290, 354), 24), ((290, 354), 60), ((1156, 2054), 24), ((1156, 2054), 60)])
def test_packed_8bit_strip_agrees(dev, oracle, monkeypatch, shape, fps):
    """8-bit clips take the packed-arithmetic strip (v_dot4 / v_sad_u8 / packed i16); the generic
    strip on the same frames (VSZIP_XPSNR_UNPACKED) and the oracle give the same sums. Full-range
    noise + extreme values so that every intermediate reaches its bound."""
    rng = np.random.default_rng(shape[0] + fps)
    shapes = [shape, (shape[0] // 2, shape[1] // 2), (shape[0] // 2, shape[1] // 2)]
    frames, recs = _rand_clip(rng, shapes, np.uint8, 8, 3)
    frames[1][0][::2, ::2] = 255
    frames[1][0][1::2, 1::2] = 0
    recs[1][0][:] = 255 - frames[1][0]
    want = _want(oracle, frames, recs, 8, fps)
    dfr = [[dev.upload(p) for p in fr] for fr in frames]
    drc = [[dev.upload(p) for p in fr] for fr in recs]
    p1 = [dfr[n - 1][0] if n >= 1 else None for n in range(3)]
    p2 = [dfr[n - 2][0] if n >= 2 else None for n in range(3)]
    packed = dev.xpsnr_wsse_batch(dfr, drc, p1, p2, depth=8, frame_rate=fps)
    assert packed == want
    with dev.variant(VSZIP_XPSNR_UNPACKED=1) as built:  # (a development variant since round 5: the forcing switch, not the kernel the geometry rules route to)
        if built:
            assert dev.xpsnr_wsse_batch(dfr, drc, p1, p2, depth=8, frame_rate=fps) == want


@pytest.mark.parametrize("shape", [(288, 352), (480, 640), (1080, 1920), (1200, 2100), (40, 44), (96, 160)])
def test_device_weighting_is_the_host_weighting(dev, oracle, monkeypatch, shape):
    """The f64 weighting (getWSSE :437-521) runs on the device by default; the host evaluation of
    the same expressions (VSZIP_XPSNR_HOST_WEIGH) must give the same u64s, including the
    <=640x480 weight smoothing."""
    rng = np.random.default_rng(shape[0])
    shapes = [shape, (shape[0] // 2, shape[1] // 2), (shape[0] // 2, shape[1] // 2)]
    frames, recs = _rand_clip(rng, shapes, np.uint8, 8, 4)
    # smooth content so that the block weights differ a lot (the smoothing step then matters)
    for fr in frames:
        fr[0][: shape[0] // 2] = fx.tiled_natural((shape[0] // 2, shape[1]), np.uint8, 1)
    want = _want(oracle, frames, recs, 8, 24)
    dfr = [[dev.upload(p) for p in fr] for fr in frames]
    drc = [[dev.upload(p) for p in fr] for fr in recs]
    p1 = [dfr[n - 1][0] if n >= 1 else None for n in range(4)]
    on_dev = dev.xpsnr_wsse_batch(dfr, drc, p1, None, depth=8, frame_rate=24)
    assert on_dev == want
    with dev.variant(VSZIP_XPSNR_HOST_WEIGH=1) as built:  # (a development variant since round 5: the forcing switch, not the kernel the geometry rules route to)
        if built:
            assert dev.xpsnr_wsse_batch(dfr, drc, p1, None, depth=8, frame_rate=24) == want


@pytest.mark.parametrize("seed", range(14))
def test_batch_random_geometries(dev, oracle, seed):
    """Seeded random even sizes on both sides of the 2048x1152 switch (3x3 vs highds activity), both
    depths, both temporal orders, 4:2:0 / 4:4:4 / 4:2:2: the strip kernel (and, for widths that are
    2 mod 4 with an unpadded stride, the per-block kernels) against the oracle."""
    rng = np.random.default_rng(31000 + seed)
    big = seed % 3 == 0
    h = int(rng.integers(600, 760)) * 2 if big else int(rng.integers(9, 420)) * 2
    w = int(rng.integers(900, 1100)) * 2 if big else int(rng.integers(9, 640)) * 2
    ssw, ssh = [(1, 1), (0, 0), (1, 0)][seed % 3]
    dtype, depth = [(np.uint8, 8), (np.uint16, 10)][seed % 2]
    shapes = [(h, w), (h >> ssh, w >> ssw), (h >> ssh, w >> ssw)]
    nf = 3
    frames, recs = _rand_clip(rng, shapes, dtype, depth, nf)
    fps = 60 if seed % 4 < 2 else 24
    want = _want(oracle, frames, recs, depth, fps)
    pitch = 1 if seed % 5 == 0 else 256  # tight rows now and then: stride == width
    dfr = [[dev.upload(p, pitch) for p in fr] for fr in frames]
    drc = [[dev.upload(p, pitch) for p in fr] for fr in recs]
    p1 = [dfr[n - 1][0] if n >= 1 else None for n in range(nf)]
    p2 = [dfr[n - 2][0] if n >= 2 else None for n in range(nf)]
    assert dev.xpsnr_wsse_batch(dfr, drc, p1, p2, depth=depth, frame_rate=fps) == want, (seed, h, w, ssw, ssh, depth, fps, pitch)
