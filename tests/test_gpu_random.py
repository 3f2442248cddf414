"""Seeded random-geometry parity: plane sizes, strides, radii and batch compositions that no
hand-written case covers (band splits of the ring kernel, launch-table limits, tiles that are
all border, lines barely longer than the search reach). Bit-exact against the oracle."""
import os

import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu
# soak runs: VSZIP_TEST_SEED_BASE=<n> shifts every seed below (tools/soak_random.sh loops over bases); the committed suite runs base 0
SEED_BASE = 100003 * int(os.environ.get("VSZIP_TEST_SEED_BASE", "0"))


@pytest.fixture(scope="module")
def dev():
    import vszip_amd

    d = vszip_amd.Device(0)
    yield d
    d.close()


def _plane(rng, shape, dtype):
    if np.dtype(dtype).kind == "u":
        peak = np.iinfo(dtype).max
        kind = rng.integers(0, 3)
        if kind == 0:
            return rng.integers(0, peak + 1, size=shape, dtype=dtype)
        if kind == 1:
            return fx.tiled_natural(shape, dtype, int(rng.integers(0, 3)))
        return np.full(shape, peak if rng.integers(0, 2) else 0, dtype)  # extremes: no overflow at full scale
    return rng.random(shape).astype(dtype)


@pytest.mark.parametrize("seed", range(12))
def test_boxblur_int_random_batches(dev, oracle, seed):
    """CT (ring + generic kernel) and RT integer paths on random batches of mixed plane sizes."""
    rng = np.random.default_rng(SEED_BASE + 1000 + seed)
    dtype = [np.uint8, np.uint16][seed % 2]
    ct = seed % 3 != 2
    r = int(rng.integers(1, 23)) if ct else int(rng.integers(1, 40))
    args = (r, 1, r, 1) if ct else (r, int(rng.integers(1, 3)), int(rng.integers(1, 40)), int(rng.integers(0, 3)))
    need = 2 * max(args[0], args[2]) + 1
    planes = []
    for _ in range(int(rng.integers(1, 9))):
        h = int(rng.integers(max(need, 54), 400))
        w = int(rng.integers(max(need, 24), 700))
        if rng.integers(0, 3):
            w = (w + 7) // 8 * 8  # the ring kernel's fast variant
        planes.append(_plane(rng, (h, w), dtype))
    align = int(rng.choice([1, 8, 32]))
    srcs = [dev.upload(p, align) for p in planes]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype, align) for p in planes]
    dev.boxblur(srcs, dsts, *args)
    for p, d in zip(planes, dsts):
        assert np.array_equal(dev.download(d), oracle.boxblur(p, *args)), (seed, p.shape, args, align)


def test_boxblur_ring_many_planes(dev, oracle):
    """More planes than one launch table holds (192) and more than the old 48: exercises the
    block -> plane map and the launch split."""
    rng = np.random.default_rng(7)
    planes = [rng.integers(0, 65536, size=(int(rng.integers(60, 90)), 64 + 8 * int(rng.integers(0, 8))), dtype=np.uint16) for _ in range(230)]
    srcs = [dev.upload(p) for p in planes]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
    dev.boxblur(srcs, dsts, 5, 1, 5, 1)
    for i in (0, 47, 48, 191, 192, 229):
        assert np.array_equal(dev.download(dsts[i]), oracle.boxblur(planes[i], 5, 1, 5, 1)), i


@pytest.mark.parametrize("seed", range(6))
def test_boxblur_float_random(dev, oracle, seed):
    rng = np.random.default_rng(SEED_BASE + 2000 + seed)
    dtype = [np.float32, np.float16][seed % 2]
    r = int(rng.integers(1, 23))
    h, w = int(rng.integers(2 * r + 1, 200)), int(rng.integers(2 * r + 1, 300))
    p = _plane(rng, (h, w), dtype)
    s, d = dev.upload(p), dev.empty(h, w, dtype)
    dev.boxblur([s], [d], r, 1, r, 1)
    assert np.array_equal(dev.download(d).view(np.uint8), oracle.boxblur(p, r, 1, r, 1).view(np.uint8)), (seed, r, h, w)


@pytest.mark.parametrize("seed", range(12))
def test_boxblur_float_runtime_passes_random(dev, oracle, seed):
    """The runtime float path: 1 ... 6 passes on either axis (2 ... 5 take the pass-chain kernels, 1 and 6 a launch per pass), radii up to 30, f32 and f16,
    batches of planes of mixed sizes, lines from barely 2 R + 2 samples to several prefetch groups / tiles; the pass chain against a launch per pass as well."""
    rng = np.random.default_rng(SEED_BASE + 2300 + seed)
    dtype = [np.float32, np.float16][seed % 2]
    hr, vr = int(rng.integers(1, 31)), int(rng.integers(1, 31))
    if seed % 3 == 0:
        hr, vr = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    hp, vp = int(rng.integers(1, 7)), int(rng.integers(1, 7))
    if seed % 4 == 1:
        hr = 0  # vertical passes only
    if seed % 4 == 3:
        vr = 0
    planes = []
    for _ in range(int(rng.integers(1, 4))):
        h = int(rng.integers(2 * vr + 2, 2 * vr + 3 + [6, 80, 400][int(rng.integers(0, 3))]))
        w = int(rng.integers(2 * hr + 2, 2 * hr + 3 + [6, 80, 400][int(rng.integers(0, 3))]))
        planes.append(_plane(rng, (h, w), dtype))

    def run():
        srcs = [dev.upload(p) for p in planes]
        dsts = [dev.empty(p.shape[0], p.shape[1], dtype) for p in planes]
        dev.boxblur(srcs, dsts, hr, hp, vr, vp)
        return [dev.download(d) for d in dsts]

    dflt = run()  # (small calls: the vertical chain, the horizontal passes a launch each)
    with dev.options(VSZIP_RT_FCHAIN_ALL=1):
        got = run()
    with dev.options(VSZIP_RT_NO_FCHAIN=1):
        per_pass = run()
    for p, a, b, c in zip(planes, got, per_pass, dflt):
        want = oracle.boxblur(p, hr, hp, vr, vp)
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (seed, (hr, hp, vr, vp), p.shape, "chain vs per pass", np.argwhere(a != b)[:3].tolist())
        assert np.array_equal(a.view(np.uint8), want.view(np.uint8)), (seed, (hr, hp, vr, vp), p.shape, np.argwhere(a != want)[:3].tolist())
        assert np.array_equal(c.view(np.uint8), want.view(np.uint8)), (seed, (hr, hp, vr, vp), p.shape, "default paths")


@pytest.mark.parametrize("seed", range(12))
def test_boxblur_int_runtime_passes_random(dev, oracle, seed):
    """The runtime integer path with several passes: whatever the library picks (all horizontal passes in one launch, two vertical stages, the vertical pass
    chain, a launch per pass), the vertical pass chain forced wherever it can run, and a launch per pass — all three the oracle's bits. u8 / u16, radii to 30,
    1 ... 7 passes, batches of planes of mixed sizes, widths that are not whole sample groups, lines from 2 R + 2 samples up."""
    rng = np.random.default_rng(SEED_BASE + 2400 + seed)
    dtype = [np.uint16, np.uint8][seed % 2]
    hr, vr = int(rng.integers(1, 31)), int(rng.integers(1, 31))
    if seed % 3 == 0:
        hr, vr = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    hp, vp = int(rng.integers(1, 8)), int(rng.integers(2, 8))
    if seed % 4 == 1:
        hr = 0
    planes = []
    for _ in range(int(rng.integers(1, 4))):
        h = int(rng.integers(2 * vr + 2, 2 * vr + 3 + [6, 80, 400][int(rng.integers(0, 3))]))
        w = int(rng.integers(2 * hr + 2, 2 * hr + 3 + [6, 80, 400][int(rng.integers(0, 3))]))
        planes.append(_plane(rng, (h, w), dtype))

    def run(opts):
        with dev.options(**{k: 1 for k in opts}):
            srcs = [dev.upload(p) for p in planes]
            dsts = [dev.empty(p.shape[0], p.shape[1], dtype) for p in planes]
            dev.boxblur(srcs, dsts, hr, hp, vr, vp)
            return [dev.download(d) for d in dsts]

    outs = {"default": run([]), "chain": run(["VSZIP_RT_ICHAIN_ALL"]), "per pass": run(["VSZIP_RT_NO_ICHAIN", "VSZIP_RT_NO_HSMALL"])}
    if fx.has_dev_variants(dev):  # boxblur_rt_vsmall_kernel: a -DVSZIP_DEV_VARIANTS build only
        outs["two vertical stages"] = run(["VSZIP_RT_VSMALL"])
    for i, p in enumerate(planes):
        want = oracle.boxblur(p, hr, hp, vr, vp)
        for name, o in outs.items():
            assert np.array_equal(o[i], want), (seed, name, (hr, hp, vr, vp), p.shape, np.argwhere(o[i] != want)[:3].tolist())


@pytest.mark.parametrize("seed", range(10))
def test_boxblur_float_ring_random(dev, oracle, seed):
    """Planes large enough for the float register-ring kernel, in batches of mixed sizes: column-tile counts, bands that
    end inside a ring period, ragged right strips, odd row counts, one call over all of them."""
    rng = np.random.default_rng(SEED_BASE + 2500 + seed)
    dtype = [np.float32, np.float16][seed % 2]
    r = int(rng.integers(1, 23))
    planes = [_plane(rng, (int(rng.integers(2 * r + 40, 700)), int(rng.integers(270, 1500))), dtype) for _ in range(int(rng.integers(1, 5)))]
    if seed % 3 == 0:
        planes.append(_plane(rng, (int(rng.integers(2 * r + 1, 60)), int(rng.integers(2 * r + 1, 200))), dtype))  # one that stays on the tile kernel
    srcs = [dev.upload(p) for p in planes]
    dsts = [dev.empty(p.shape[0], p.shape[1], dtype) for p in planes]
    dev.boxblur(srcs, dsts, r, 1, r, 1)
    for i, p in enumerate(planes):
        bad = dev.download(dsts[i]).view(np.uint8) != oracle.boxblur(p, r, 1, r, 1).view(np.uint8)
        assert not bad.any(), (seed, r, p.shape, np.argwhere(bad)[:3].tolist())


@pytest.mark.parametrize("seed", range(8))
def test_bilateral_random(dev, oracle, seed):
    rng = np.random.default_rng(SEED_BASE + 3000 + seed)
    dtype = [np.uint8, np.uint16, np.float32, np.float16][seed % 4]
    sS = float(rng.choice([0.6, 1.0, 2.0, 3.5, 6.0]))
    sR = float(rng.choice([0.02, 0.1, 0.5, 2.0]))
    cfg = dev.bilateral_cfg([sS], [sR], hist_len=(1 << (8 * np.dtype(dtype).itemsize)) if np.dtype(dtype).kind == "u" else 65536)
    rad = cfg[0].radius if cfg[0].algorithm == 2 else 1
    h, w = int(rng.integers(2 * rad + 1, 150)), int(rng.integers(2 * rad + 1, 260))
    p = _plane(rng, (h, w), dtype)
    ref = _plane(rng, (h, w), dtype) if seed % 2 else None
    s, d = dev.upload(p), dev.empty(h, w, dtype)
    dev.bilateral([s], [d], cfg, [0], [dev.upload(ref)] if ref is not None else None)
    want = oracle.bilateral_plane(p, cfg[0].sigmaS, cfg[0].sigmaR, cfg[0].algorithm, cfg[0].radius, cfg[0].step, cfg[0].pbficnum, ref=ref)
    assert np.array_equal(dev.download(d).view(np.uint8), want.view(np.uint8)), (seed, sS, sR, cfg[0].algorithm, h, w)
    dev.bilateral_free(cfg)


@pytest.mark.parametrize("seed", range(8))
def test_eedi3_random(dev, oracle, seed):
    rng = np.random.default_rng(SEED_BASE + 4000 + seed)
    hp = bool(seed % 2)
    mdis = int(rng.integers(1, 41))
    nrad = int(rng.integers(0, 4))
    dh = bool(rng.integers(0, 2))
    w = int(rng.integers(2 * mdis + nrad + 4, 2 * mdis + nrad + 200))
    h = int(rng.integers(4, 40)) * 2
    p = rng.random((h, w)).astype(np.float32)
    kw = dict(dh=dh, hp=hp, mdis=mdis, nrad=nrad, vcheck=int(rng.integers(0, 4)), gamma=float(rng.choice([0.0, 5.0, 20.0, 80.0])))
    field = int(rng.integers(0, 2))
    s = dev.upload(p)
    (d,) = dev.eedi3([s], field, **kw)
    assert np.array_equal(dev.download(d), oracle.eedi3(p, field, **kw)), (seed, field, kw, p.shape)


@pytest.mark.parametrize("seed", range(8))
def test_eedi3_mclip_random(dev, oracle, seed):
    """mclip on the tuned line kernel (mdis <= 20) and, for larger mdis, on the general one: random widths around the 64-column blocks, masks from empty to dense
    made of samples, runs, empty lines and empty stretches; sclip on some; two planes a call, the second without a mask on some"""
    rng = np.random.default_rng(SEED_BASE + 4500 + seed)
    mdis = int(rng.integers(1, 21)) if seed % 4 else int(rng.integers(21, 32))
    nrad = int(rng.integers(0, 4))
    dh = bool(rng.integers(0, 2))
    w = int(rng.integers(2 * mdis + nrad + 4, 2 * mdis + nrad + 330))
    h = int(rng.integers(4, 30)) * 2
    field = int(rng.integers(0, 2))
    kw = dict(dh=dh, mdis=mdis, nrad=nrad, vcheck=int(rng.integers(0, 4)), gamma=float(rng.choice([0.0, 5.0, 20.0, 80.0])))
    planes, masks = [], []
    for k in range(2):
        p = rng.random((h, w)).astype(np.float32)
        dens = float(rng.choice([0.0, 0.002, 0.02, 0.3, 0.95]))
        m = (rng.random((h, w)) < dens).astype(np.uint8) * int(rng.choice([1, 128, 255]))
        for _ in range(int(rng.integers(0, 4))):  # runs of samples
            y, x0, n = int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(1, 90))
            m[y, x0:x0 + n] = 255
        m[int(rng.integers(0, h))] = 0
        x0 = int(rng.integers(0, w))
        m[:, x0:x0 + int(rng.integers(0, 200))] = 0
        if rng.integers(0, 3) == 0:
            m[:, :2] = 255
        planes.append(p)
        masks.append(m if (k == 0 or seed % 3) else None)
    sclips = None
    if seed % 2 and kw["vcheck"] > 0:
        sclips = [rng.random((2 * h if dh else h, w)).astype(np.float32) for _ in planes]
    ds = dev.eedi3([dev.upload(p) for p in planes], field, mclips=[dev.upload(m) if m is not None else None for m in masks],
                   sclips=[dev.upload(x) for x in sclips] if sclips else None, **kw)
    for k, p in enumerate(planes):
        want = oracle.eedi3(p, field, mclip=masks[k], sclip=sclips[k] if sclips else None, **kw)
        got = dev.download(ds[k])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, k, field, kw, p.shape, int((got.view(np.uint32) != want.view(np.uint32)).sum()))


@pytest.mark.parametrize("seed", range(6))
def test_planestats_random(dev, oracle, seed):
    rng = np.random.default_rng(SEED_BASE + 5000 + seed)
    dtype = [np.uint8, np.uint16, np.float32][seed % 3]
    planes = [_plane(rng, (int(rng.integers(1, 200)), int(rng.integers(1, 500))), dtype) for _ in range(int(rng.integers(1, 6)))]
    align = int(rng.choice([1, 8, 32]))
    srcs = [dev.upload(p, align) for p in planes]
    excl = [int(x) for x in rng.integers(0, 256, size=int(rng.integers(0, 4)))]
    avg, _ = dev.plane_average(srcs, exclude=excl)
    thr = float(rng.choice([0.0, 0.05, 0.3]))
    mn, mx, _ = dev.plane_minmax(srcs, thr, thr)
    for i, p in enumerate(planes):
        oa = oracle.plane_average(p, exclude=excl)[0]
        assert avg[i] == (oa if np.dtype(dtype).kind == "u" else pytest.approx(oa, rel=1e-12)), (seed, i)
        omn, omx = oracle.plane_minmax(p, thr, thr)[:2]
        assert (mn[i], mx[i]) == (omn, omx), (seed, i, thr)


@pytest.mark.parametrize("seed", range(8))
def test_minmax_prediction_random_sequences(dev, oracle, seed):
    """Thresholded PlaneMinMax over a SEQUENCE of calls (round 6: a call of the previous call's shape reads its planes once over the ranges the previous
    answers predict; a miss takes the two sweeps): the same frame again, a nearby frame, unrelated content, a change of thresholds or of the plane set -
    whatever the prediction table holds (this test's own history and every earlier test's on the shared context), each result is the oracle's."""
    rng = np.random.default_rng(SEED_BASE + 5500 + seed)
    dtype = [np.uint16, np.float32, np.float16, np.uint16][seed % 4]
    kind = np.dtype(dtype).kind
    nplanes = int(rng.integers(1, 6))
    shapes = [(int(rng.integers(2, 220)), int(rng.integers(8, 520))) for _ in range(nplanes)]
    thr = (float(rng.choice([0.01, 0.1, 0.25])), float(rng.choice([0.02, 0.1, 0.4])))
    frame = [_plane(rng, s, dtype) for s in shapes]
    for step in range(7):
        what = int(rng.integers(0, 5))
        if what == 1:  # the next frame: a small drift
            frame = [np.clip(p.astype(np.float64) + (rng.integers(-150, 151) if kind == "u" else rng.uniform(-0.002, 0.002)), 0, 65535 if kind == "u" else 1).astype(dtype) for p in frame]
        elif what == 2:  # a cut
            frame = [_plane(rng, s, dtype) for s in shapes]
        elif what == 3:
            thr = (float(rng.choice([0.0, 0.01, 0.1, 0.25, 1.0])), float(rng.choice([0.02, 0.1, 0.4])))
        elif what == 4 and nplanes > 1:  # another plane set (another shape: no prediction)
            shapes = shapes[::-1]
            frame = [_plane(rng, s, dtype) for s in shapes]
        srcs = [dev.upload(p, int(rng.choice([1, 8, 32]))) for p in frame]
        mn, mx, _ = dev.plane_minmax(srcs, thr[0], thr[1])
        for i, p in enumerate(frame):
            omn, omx = oracle.plane_minmax(p, thr[0], thr[1])[:2]
            assert (mn[i], mx[i]) == (omn, omx), (seed, step, what, i, thr, p.shape)


@pytest.mark.parametrize("seed", range(6))
def test_ssimulacra2_random_sizes(dev, oracle, seed):
    """Odd and tiny frames (down to one 8x8 tile at scale 0; five halvings) and a batch of pairs."""
    rng = np.random.default_rng(SEED_BASE + 6000 + seed)
    h, w = int(rng.integers(8, 300)), int(rng.integers(8, 400))
    npairs = int(rng.integers(1, 4))
    ref, dis = [], []
    for _ in range(npairs):
        r = [rng.random((h, w)).astype(np.float32) for _ in range(3)]
        ref.append(r)
        dis.append([np.clip(p + rng.normal(0, 0.05, p.shape).astype(np.float32), 0, 1).astype(np.float32) for p in r])
    rr = [dev.upload(p, 1) for pr in ref for p in pr]
    dd = [dev.upload(p, 1) for pr in dis for p in pr]
    got = dev.ssimulacra2(rr, dd)
    for i in range(npairs):
        assert got[i] == pytest.approx(oracle.ssimulacra2(ref[i], dis[i]), abs=1e-7), (seed, i, h, w)


@pytest.mark.parametrize("seed", range(6))
def test_xpsnr_random(dev, oracle, seed):
    rng = np.random.default_rng(SEED_BASE + 7000 + seed)
    dtype, depth = [(np.uint8, 8), (np.uint16, 10)][seed % 2]
    h, w = int(rng.integers(16, 140)) * 2, int(rng.integers(16, 200)) * 2
    peak = (1 << depth) - 1
    shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    frames = [[rng.integers(0, peak + 1, size=s).astype(dtype) for s in shapes] for _ in range(3)]
    recs = [[np.clip(p.astype(np.int32) + rng.integers(-6, 7, p.shape), 0, peak).astype(dtype) for p in fr] for fr in frames]
    dfr = [[dev.upload(p) for p in fr] for fr in frames]
    drc = [[dev.upload(p) for p in fr] for fr in recs]
    fps = int(rng.choice([24, 60]))
    for n in range(3):
        p1, p2 = (dfr[n - 1][0] if n >= 1 else None), (dfr[n - 2][0] if n >= 2 else None)
        got = dev.xpsnr_wsse(dfr[n], drc[n], p1, p2, depth=depth, frame_rate=fps)
        want = oracle.xpsnr_wsse(frames[n], recs[n], frames[n - 1][0] if n >= 1 else None, frames[n - 2][0] if n >= 2 else None, depth=depth, frame_rate=fps)
        assert got == want, (seed, n, h, w, fps)


def test_boxblur_rt_huge_radius(dev, oracle):
    """A horizontal radius beyond one 512-column chunk takes the whole-row-prefix kernel instead of
    the ring one; a vertical radius larger than the band heuristic's minimum."""
    p = fx.splitmix64_plane(12, (700, 1400), np.uint16)
    s, d = dev.upload(p), dev.empty(700, 1400, np.uint16)
    dev.boxblur([s], [d], 600, 1, 300, 2)
    assert np.array_equal(dev.download(d), oracle.boxblur(p, 600, 1, 300, 2))
    p8 = (fx.splitmix64_plane(13, (64, 2300), np.uint16) >> 8).astype(np.uint8)
    s, d = dev.upload(p8), dev.empty(64, 2300, np.uint8)
    dev.boxblur([s], [d], 1030, 1, 0, 0)
    assert np.array_equal(dev.download(d), oracle.boxblur(p8, 1030, 1, 0, 0))


def test_boxblur_rt_very_wide_rows(dev, oracle):
    """Rows far wider than any video line (the ring kernel keeps three chunks, not the row, in LDS)."""
    p = fx.splitmix64_plane(14, (24, 20000), np.uint16)
    s, d = dev.upload(p), dev.empty(24, 20000, np.uint16)
    dev.boxblur([s], [d], 30, 2, 9, 1)
    assert np.array_equal(dev.download(d), oracle.boxblur(p, 30, 2, 9, 1))


def test_more_planes_than_one_kernel_table(dev, oracle):
    """Calls with more planes than a kernel-argument table holds (48) are split into batches inside
    the library: plane statistics, EEDI3, Bilateral."""
    rng = np.random.default_rng(4242)
    planes = [rng.integers(0, 65536, size=(40 + i % 7, 72 + 8 * (i % 5)), dtype=np.uint16) for i in range(110)]
    dp = [dev.upload(p) for p in planes]
    avg, _ = dev.plane_average(dp, exclude=[-1])
    mn, mx, _ = dev.plane_minmax(dp, 0.05, 0.05)
    for i, p in enumerate(planes):
        assert avg[i] == oracle.plane_average(p)[0]
        assert (mn[i], mx[i]) == oracle.plane_minmax(p, 0.05, 0.05)[:2]
    fl = [rng.random((24, 96 + 8 * (i % 3))).astype(np.float32) for i in range(60)]
    outs = dev.eedi3([dev.upload(p) for p in fl], 1)
    for i in (0, 47, 48, 59):
        assert np.array_equal(dev.download(outs[i]), oracle.eedi3(fl[i], 1))


@pytest.mark.parametrize("seed", range(10))
def test_ssim_yuv_sources_random(dev, oracle, seed):
    """SSIMULACRA2 from YUV clips (round 5, after a fused-pass bug that only top-sited 4:2:0 showed): random subsampling, depth, chroma siting,
    matrix, range and geometry (several tiles each way, ragged last tiles, odd pitches) - the conversion alone is BIT-EXACT against the
    oracle's zimg restatement and the fused pass scores what the oracle scores on the oracle's converted planes."""
    from oracle import vs_host as vh

    rng = np.random.default_rng(SEED_BASE + 8000 + seed)
    ssw, ssh = [(1, 1), (1, 1), (1, 0), (0, 0), (2, 2), (0, 1), (2, 0)][int(rng.integers(0, 7))]
    bits, dtype = [(8, np.uint8), (10, np.uint16), (16, np.uint16), (32, np.float32)][int(rng.integers(0, 4))]
    loc, matrix = int(rng.integers(0, 6)), int(rng.choice([1, 6, 9]))
    limited = bool(rng.integers(0, 4))  # mostly limited range, like the clips VapourSynth hands over
    h = int(rng.integers(3, 60 if seed % 3 else 180)) << max(ssh, 1)
    w = int(rng.integers(3, 90 if seed % 3 else 300)) << max(ssw, 1)
    cs = ((h + (1 << ssh) - 1) >> ssh, (w + (1 << ssw) - 1) >> ssw)

    def clip():
        if dtype == np.float32:
            return [rng.random((h, w)).astype(np.float32)] + [(rng.random(cs) - 0.5).astype(np.float32) for _ in range(2)]
        lo, hi = ((16 << (bits - 8), 235 << (bits - 8)) if limited else (0, (1 << bits) - 1))
        y = fx.tiled_natural((h, w), np.uint16, seed % 3).astype(np.float64) / 65535.0
        return [(lo + y * (hi - lo)).astype(dtype)] + [rng.integers(0, 1 << bits, size=cs).astype(dtype) for _ in range(2)]

    ref = clip()
    dis = [np.clip(p.astype(np.float64) + rng.normal(0, 0.01 * (1.0 if dtype == np.float32 else float(1 << bits)), p.shape), 0 if dtype != np.float32 else -0.5,
                   1.0 if dtype == np.float32 else (1 << bits) - 1).astype(dtype) for p in ref]
    fmt = dev.ssim_source("YUV", dtype, bits, True, limited=limited, ssw=ssw, ssh=ssh, matrix=matrix, chroma_loc=loc)
    pitch = 1 if seed % 4 == 0 else 32
    want_rgb = [vh.srgb_to_linear(p) for p in vh.yuv_to_rgbs(ref, bits, ssw, ssh, matrix, loc, limited=limited)]
    got_rgb = [dev.download(d) for d in dev.to_rgbs_linear(fmt, [dev.upload(p, pitch) for p in ref])]
    what = (seed, (h, w), (ssw, ssh), bits, loc, matrix, limited, pitch)
    for c in range(3):
        assert np.array_equal(got_rgb[c].view(np.uint32), want_rgb[c].view(np.uint32)), what + (c,)
    if min(h, w) >= 8:
        want_dis = [vh.srgb_to_linear(p) for p in vh.yuv_to_rgbs(dis, bits, ssw, ssh, matrix, loc, limited=limited)]
        got = dev.ssimulacra2_src(fmt, [dev.upload(p, pitch) for p in ref], [dev.upload(p, pitch) for p in dis])[0]
        assert got == pytest.approx(oracle.ssimulacra2(want_rgb, want_dis), abs=1e-7), what


@pytest.mark.parametrize("seed", range(8))
def test_xpsnr_batch_random(dev, oracle, seed):
    """The strip kernel's lane widths (round 5: 8 pixels a lane where a block is whole 8-sample groups, 4 otherwise; halo samples from the
    neighbouring lanes) over random even sizes on both sides of the 2048 x 1152 switch, both depths, 4:2:0 / 4:4:4 / 4:2:2, tight rows."""
    rng = np.random.default_rng(SEED_BASE + 8500 + seed)
    big = seed % 4 == 0
    h = int(rng.integers(600, 700)) * 2 if big else int(rng.integers(9, 380)) * 2
    w = int(rng.integers(900, 1100)) * 2 if big else int(rng.integers(9, 700)) * 2
    ssw, ssh = [(1, 1), (0, 0), (1, 0)][int(rng.integers(0, 3))]
    dtype, depth = [(np.uint8, 8), (np.uint16, 10)][int(rng.integers(0, 2))]
    peak = (1 << depth) - 1
    shapes = [(h, w), (h >> ssh, w >> ssw), (h >> ssh, w >> ssw)]
    nf = 3
    frames = [[rng.integers(0, peak + 1, size=s).astype(dtype) for s in shapes] for _ in range(nf)]
    recs = [[np.clip(p.astype(np.int64) + rng.integers(-9, 10, p.shape), 0, peak).astype(dtype) for p in fr] for fr in frames]
    fps = int(rng.choice([24, 60]))
    want = [oracle.xpsnr_wsse(frames[n], recs[n], frames[n - 1][0] if n >= 1 else None, frames[n - 2][0] if n >= 2 else None, depth=depth, frame_rate=fps) for n in range(nf)]
    pitch = 1 if seed % 3 == 0 else 256
    dfr = [[dev.upload(p, pitch) for p in fr] for fr in frames]
    drc = [[dev.upload(p, pitch) for p in fr] for fr in recs]
    p1 = [dfr[n - 1][0] if n >= 1 else None for n in range(nf)]
    p2 = [dfr[n - 2][0] if n >= 2 else None for n in range(nf)]
    assert dev.xpsnr_wsse_batch(dfr, drc, p1, p2, depth=depth, frame_rate=fps) == want, (seed, h, w, ssw, ssh, depth, fps, pitch)


@pytest.mark.parametrize("seed", range(8))
def test_point_filters_random(dev, oracle, seed):
    """Limiter, LimitFilter (with and without the third clip) and AdaptiveBinarize: random plane sizes (down to one sample, odd pitches, several
    planes a call), sample types, bounds and thresholds - bit-exact."""
    rng = np.random.default_rng(SEED_BASE + 9500 + seed)
    dtype = [np.uint8, np.uint16, np.float32][int(rng.integers(0, 3))]
    n = int(rng.integers(1, 5))
    shapes = [(int(rng.integers(1, 200)), int(rng.integers(1, 700))) for _ in range(n)]
    peak = float(np.iinfo(dtype).max) if np.dtype(dtype).kind == "u" else 1.0
    mk = lambda s: _plane(rng, s, dtype)
    pitch = 1 if seed % 2 else 64
    # Limiter
    srcs = [mk(s) for s in shapes]
    lo = [float(rng.uniform(0, 0.6)) * peak for _ in range(n)]
    hi = [l + float(rng.uniform(0, 0.4)) * peak for l in lo]
    if np.dtype(dtype).kind == "u":
        lo, hi = [float(int(v)) for v in lo], [float(int(v)) for v in hi]
    ds, dd = [dev.upload(p, pitch) for p in srcs], [dev.empty(s[0], s[1], dtype) for s in shapes]
    dev.limiter(ds, dd, lo, hi)
    for i in range(n):
        assert np.array_equal(dev.download(dd[i]), oracle.limiter(srcs[i], lo[i], hi[i])), ("limiter", seed, i, shapes[i], lo[i], hi[i])
    # LimitFilter
    flts = [mk(s) for s in shapes]
    refs = [mk(s) for s in shapes] if seed % 3 else None
    dark = [float(rng.uniform(0, 0.05)) * peak for _ in range(n)]
    bright = [float(rng.uniform(0, 0.05)) * peak for _ in range(n)]
    elast = [float(rng.uniform(1.0, 4.0)) for _ in range(n)]
    df, dr = [dev.upload(p, pitch) for p in flts], ([dev.upload(p, pitch) for p in refs] if refs else None)
    dev.limit_filter(df, ds, dd, dark, bright, elast, refs=dr)
    for i in range(n):
        want = oracle.limit_filter(flts[i], srcs[i], refs[i] if refs else None, np.float32(dark[i]), np.float32(bright[i]), np.float32(elast[i]))
        got = dev.download(dd[i])
        assert np.array_equal(got.view(np.uint32) if dtype == np.float32 else got, want.view(np.uint32) if dtype == np.float32 else want), ("limit_filter", seed, i, shapes[i])
    # AdaptiveBinarize (8-bit only)
    a = [rng.integers(0, 256, size=s, dtype=np.uint8) for s in shapes]
    b = [np.clip(p.astype(np.int16) + rng.integers(-6, 7, p.shape), 0, 255).astype(np.uint8) for p in a]
    c = int(rng.integers(0, 8))
    da, db, do = [dev.upload(p, pitch) for p in a], [dev.upload(p, pitch) for p in b], [dev.empty(s[0], s[1], np.uint8) for s in shapes]
    dev.adaptive_binarize(da, db, do, c)
    for i in range(n):
        assert np.array_equal(dev.download(do[i]), oracle.adaptive_binarize(a[i], b[i], c)), ("adaptive_binarize", seed, i, shapes[i], c)
