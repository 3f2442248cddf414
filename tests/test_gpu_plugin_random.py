"""Seeded random filter GRAPHS through libvszip.so: whatever create-time fusion does with a script — chains of pixel filters into
pixel, two-input and metric sinks, inputs that share a root or do not — the frames and the frame properties must be bit-identical
to the same script with every intermediate clip taken through host memory (no vszip node feeds another, nothing fuses).
Covers what the hand-written twins of tests/test_gpu_plugin.py cannot enumerate: stage orders, `planes` subsets, two chains on one
root, chains on different roots, a chain as `ref` / `clipb`."""
import os

import numpy as np
import pytest

import fixtures as fx
from fakevs import fakevs as vs

pytestmark = pytest.mark.gpu
SEED_BASE = 100003 * int(os.environ.get("VSZIP_TEST_SEED_BASE", "0"))
NFR = 3


def _frames(rng, fmt):
    dt = {vs.YUV420P16: np.uint16, vs.YUV420P8: np.uint8, vs.RGBS: np.float32}[fmt]
    shapes = [(72, 128)] * 3 if fmt == vs.RGBS else [(72, 128), (36, 64), (36, 64)]
    out = []
    for f in range(NFR):
        fr = []
        for p, s in enumerate(shapes):
            a = fx.tiled_natural(s, dt, p) if rng.integers(0, 2) else fx.splitmix64_plane(int(rng.integers(0, 1 << 30)), s, dt)
            fr.append(np.roll(a, int(rng.integers(0, 9)), axis=1))
        out.append(fr)
    return out


def _stage(rng):
    kind = int(rng.integers(0, 3))
    planes = [None, [0], [0, 1, 2], [1, 2]][int(rng.integers(0, 4))]
    kw = {} if planes is None else {"planes": planes}
    if kind == 0:
        r = int(rng.integers(1, 9))
        if rng.integers(0, 3) == 0:  # the runtime path
            return ("BoxBlur", dict(hradius=r, hpasses=int(rng.integers(1, 3)), vradius=int(rng.integers(1, 9)), vpasses=int(rng.integers(1, 3)), **kw))
        return ("BoxBlur", dict(hradius=r, vradius=r, **kw))
    if kind == 1:
        return ("Bilateral", dict(sigmaS=float(rng.choice([1.0, 2.0])), sigmaR=float(rng.choice([0.1, 2.0])), **kw))
    return ("Limiter", dict(tv_range=True, **kw))


def _apply(clip, st):
    return getattr(clip.vszip, st[0])(**st[1])


def _materialize(clip, fmt):
    out = []
    for i in range(NFR):
        fr = clip.get_frame(i)
        out.append([np.array(fr[p]) for p in range(3)])
    return vs.source(out, fmt, props={"_Transfer": 8} if fmt == vs.RGBS else None)


def _chain(root, stages, fmt, fused):
    clip = root
    for st in stages:
        clip = _apply(clip, st)
        if not fused:
            clip = _materialize(clip, fmt)
    return clip


def _same_frames(a, b):
    for n in range(NFR):
        fa, fb = a.get_frame(n), b.get_frame(n)
        for p in range(3):
            assert np.array_equal(np.array(fa[p]), np.array(fb[p])), (n, p)


@pytest.mark.parametrize("seed", range(18))
def test_random_graph_fused_equals_unfused(seed):
    rng = np.random.default_rng(SEED_BASE + 9000 + seed)
    sink = ["pixel", "limitfilter", "average", "minmax", "xpsnr", "ssim"][seed % 6]
    fmt = vs.YUV420P8 if sink == "xpsnr" else (vs.RGBS if sink == "ssim" else vs.YUV420P16)
    props = {"_Transfer": 8} if fmt == vs.RGBS else None  # (linear light: SSIMULACRA2 takes the planes as they are)
    root_a = vs.source(_frames(rng, fmt), fmt, props=props)
    root_b = vs.source(_frames(rng, fmt), fmt, props=props) if rng.integers(0, 3) == 0 else root_a  # a second root now and then
    st_a = [_stage(rng) for _ in range(int(rng.integers(1, 4)))]
    st_b = [_stage(rng) for _ in range(int(rng.integers(0, 3)))]
    st_c = [_stage(rng) for _ in range(int(rng.integers(0, 2)))]
    f0, s0 = vs.fusion_stats()

    def build(fused):
        a = _chain(root_a, st_a, fmt, fused)
        b = _chain(root_b, st_b, fmt, fused)
        if sink == "pixel":
            return _apply(a, ("BoxBlur", dict(hradius=2, vradius=2)))
        if sink == "limitfilter":
            kw = dict(dark_thr=8, bright_thr=[8, 4], elast=3)
            if st_c:
                kw["ref"] = _chain(root_a, st_c, fmt, fused)
            if rng_planes is not None:
                kw["planes"] = rng_planes
            return a.vszip.LimitFilter(b, **kw)
        if sink == "average":
            return a.vszip.PlaneAverage(exclude=[-1], planes=[0, 1, 2], clipb=b)
        if sink == "minmax":
            return a.vszip.PlaneMinMax(minthr=thr, maxthr=thr, planes=[0, 1, 2], clipb=b)
        if sink == "ssim":
            return a.vszip.SSIMULACRA2(b)
        return b.vszip.XPSNR(a, verbose=False)

    rng_planes = [None, [0], [1, 2]][int(rng.integers(0, 3))]
    thr = float(rng.choice([0.0, 0.1]))
    fused, twin = build(True), build(False)
    got = [fused.get_frame(n) for n in range(NFR)]
    assert vs.fusion_stats()[0] > f0, "the fused graph did not fuse"
    for n in range(NFR):
        want = twin.get_frame(n)
        for p in range(3):
            assert np.array_equal(np.array(got[n][p]), np.array(want[p])), (seed, sink, n, p, st_a, st_b, st_c)
        keys = {"average": ("psmAvg", "psmDiff"), "minmax": ("psmMin", "psmMax", "psmDiff"), "xpsnr": ("XPSNR_Y", "XPSNR_U", "XPSNR_V"),
                "ssim": ("SSIMULACRA2",)}.get(sink, ())
        for k in keys:
            a, b = got[n].props[k], want.props[k]
            assert a == b or (a != a and b != b), (seed, sink, n, k, a, b)
