"""GPU parity: vszip_bilateral (algorithm 2) vs the CPU oracle — bit-exact for every
sample type (the kernel keeps the reference's f32 operation order), plus the
reference's own goldens on the reproducible inputs."""
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


def _gpu(dev, planes, sigmaS, sigmaR, refs=None, **kw):
    dt = planes[0].dtype
    hist = (1 << (8 * dt.itemsize)) if dt.kind == "u" else 65536
    cfg = dev.bilateral_cfg([sigmaS], [sigmaR], hist_len=hist, **kw)
    srcs = [dev.upload(np.ascontiguousarray(p)) for p in planes]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
    rr = [dev.upload(np.ascontiguousarray(r)) for r in refs] if refs is not None else None
    dev.bilateral(srcs, dsts, cfg, list(range(len(planes))), rr)
    out = [dev.download(d) for d in dsts]
    cfgs = [(cfg[i].sigmaS, cfg[i].sigmaR, cfg[i].algorithm, cfg[i].radius, cfg[i].step, cfg[i].pbficnum) for i in range(3)]
    dev.bilateral_free(cfg)
    return out, cfgs


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float16])
@pytest.mark.parametrize("sig", [(2, 2), (3, 0.02), (0.8, 0.05), (5, 2)])
def test_matches_oracle(dev, oracle, dtype, sig):
    sS, sR = sig
    for shape in [(120, 200), (67, 131)]:
        src = fx.tiled_natural(shape, dtype, 1)
        (got,), cfgs = _gpu(dev, [src], sS, sR, algorithm=[2])
        c = cfgs[0]
        want = oracle.bilateral_plane(src, c[0], c[1], c[2], c[3], c[4], c[5])
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dtype, sig, shape)


def _noise(shape, dtype):
    n = fx.splitmix64_plane(9, shape, np.uint16)
    if np.dtype(dtype) == np.uint8:
        return (n >> 8).astype(np.uint8)
    if np.dtype(dtype) == np.uint16:
        return n
    return (n / 65535.0).astype(dtype)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float16])
@pytest.mark.parametrize("case", [dict(sS=3, sR=0.1, num=4), dict(sS=3, sR=0.1, num=32), dict(sS=8, sR=2, num=0), dict(sS=1.5, sR=0.05, num=7)])
def test_pbfic_matches_oracle(dev, oracle, dtype, case):
    """algorithm 1 (PBFIC + recursive Gaussian): the IIR recursion keeps the reference's operation
    order line by line, so the result is bit-exact for every sample type."""
    for shape in [(120, 200), (67, 131), (3, 5), (1, 70), (70, 1)]:
        src = fx.tiled_natural(shape, dtype, 1) if min(shape) > 8 else _noise(shape, dtype)
        (got,), cfgs = _gpu(dev, [src], case["sS"], case["sR"], algorithm=[1], pbficnum=[case["num"]])
        sS, sR, alg, rad, step, num = cfgs[0]
        assert alg == 1
        want = oracle.bilateral_plane(np.ascontiguousarray(src), sS, sR, alg, rad, step, num)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dtype, case, shape)


def test_pbfic_joint_ref_and_auto_selection(dev, oracle):
    """A separate ref clip drives the range weights; algorithm 0 auto-selects PBFIC for a wide
    spatial sigma with a wide range sigma (bilateral.zig(vs):196)."""
    src = fx.tiled_natural((96, 160), np.uint16, 0)
    ref = fx.tiled_natural((96, 160), np.uint16, 2)
    (got,), cfgs = _gpu(dev, [src], 8, 2, refs=[ref])
    sS, sR, alg, rad, step, num = cfgs[0]
    assert alg == 1
    want = oracle.bilateral_plane(src, sS, sR, alg, rad, step, num, ref=ref)
    assert np.array_equal(got, want)


def test_ten_bit_clip_lut_in_lds(dev, oracle):
    """10-bit samples in a u16 container: hist_len 1024, peak 1023 — the range LUT is staged in LDS."""
    src = (fx.tiled_natural((120, 200), np.uint16, 0) >> 6).astype(np.uint16)
    ref = (fx.tiled_natural((120, 200), np.uint16, 2) >> 6).astype(np.uint16)
    for refs in (None, [ref]):
        cfg = dev.bilateral_cfg([2], [0.1], hist_len=1024)
        s, d = dev.upload(src), dev.empty(120, 200, np.uint16)
        dev.bilateral([s], [d], cfg, [0], [dev.upload(ref)] if refs else None, peak=1023.0)
        want = oracle.bilateral_plane(src, cfg[0].sigmaS, cfg[0].sigmaR, cfg[0].algorithm, cfg[0].radius, cfg[0].step, cfg[0].pbficnum, ref=ref if refs else None, bits=10)
        assert np.array_equal(dev.download(d), want)
        dev.bilateral_free(cfg)


def test_joint_ref_and_noise(dev, oracle):
    src = fx.splitmix64_plane(1, (90, 150), np.uint16)
    ref = fx.tiled_natural((90, 150), np.uint16)
    (got,), cfgs = _gpu(dev, [src], 2, 0.05, refs=[ref])
    c = cfgs[0]
    want = oracle.bilateral_plane(src, c[0], c[1], c[2], c[3], c[4], c[5], ref=ref)
    assert np.array_equal(got, want)


def test_derive_matches_oracle_yuv420(dev, oracle):
    cfg = dev.bilateral_cfg([2], [2], yuv=True, ssw=1, ssh=1)
    o = oracle.bilateral_params([2], [2], yuv=True, ssw=1, ssh=1)
    for i in range(3):
        assert (cfg[i].sigmaS, cfg[i].algorithm, cfg[i].pbficnum, cfg[i].radius, cfg[i].step, cfg[i].samples) == (
            o["sigmaS"][i], o["algorithm"][i], o["PBFICnum"][i], o["radius"][i], o["step"][i], o["samples"][i])
    dev.bilateral_free(cfg)


def test_reference_goldens(dev):
    g = fx.ref_goldens()["exact"]["bilateral"]
    for key, planes in (("RGB24|full|sigmaR=2,sigmaS=2", fx.crop_rgb24()), ("RGBS|full|sigmaR=2,sigmaS=2", fx.crop_rgbs())):
        out, _ = _gpu(dev, list(planes), 2, 2)
        for p in range(3):
            st = fx.plane_stats(out[p])
            for k in ("avg", "min", "max"):
                assert st[k] == pytest.approx(g[key][f"p{p}"][k], rel=1e-6, abs=1e-9)


def test_errors(dev):
    import vszip_amd

    with pytest.raises(vszip_amd.VszipError):
        dev.bilateral_cfg([2], [2], pbficnum=[1])
    src = np.zeros((5, 5), np.uint8)
    with pytest.raises(vszip_amd.VszipError, match="plane too small"):
        _gpu(dev, [src], 2, 2)


@pytest.mark.parametrize("dtype", [np.uint16, np.float32, np.float16])
@pytest.mark.parametrize("sig", [(2, 2), (3, 0.2), (1, 0.5)])
def test_lds16_path_equals_gathered_path(dev, oracle, dtype, sig, monkeypatch):
    """16-bit / float clips: the kernel that keeps the whole 65536-entry range LUT packed in LDS (exact:
    bits = base[i >> 6] - delta[i]) against the kernel that gathers it from global memory, and both against
    the oracle; joint `ref` clip and an odd shape included. Steep tables (small sigmaR) do not pack and take
    the gathered path by themselves — covered by test_matches_oracle's (3, 0.02) and (0.8, 0.05) cases."""
    sS, sR = sig
    for shape, joint in [((120, 200), False), ((67, 131), False), ((96, 160), True)]:
        src = fx.tiled_natural(shape, dtype, 1)
        refs = [np.ascontiguousarray(np.roll(src, 3, axis=1))] if joint else None
        (a,), cfgs = _gpu(dev, [src], sS, sR, refs=refs, algorithm=[2])
        dev.set_option("VSZIP_BILATERAL_NO_LDS16", 1)
        (b,), _ = _gpu(dev, [src], sS, sR, refs=refs, algorithm=[2])
        dev.set_option("VSZIP_BILATERAL_NO_LDS16", 0)
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (dtype, sig, shape, joint)
        c = cfgs[0]
        want = oracle.bilateral_plane(src, c[0], c[1], c[2], c[3], c[4], c[5], ref=refs[0] if joint else None)
        assert np.array_equal(a.view(np.uint8), want.view(np.uint8)), (dtype, sig, shape, joint)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("sig", [(2, 2), (1, 2), (2, 1.0), (2, 0.3), (2, 0.02), (1, 0.05), (2, 0.0625), (2, 0.004), (3, 0.02), (3, 2), (3, 1.0), (2, 0.1), (2, 0.3), (1, 0.2), (3, 0.15), (2, 0.07), (2, 0.41)])
@pytest.mark.parametrize("content", ["natural", "noise", "edges"])
def test_walk16_paths_agree(dev, oracle, sig, content, dtype, monkeypatch):
    """Round 3: 16-bit clips without `ref` whose taps are the BASELINE's (radius 3 / step 2, radius 2 / step 1) take the
    column-walking kernel — each pixel looks up only its 8 downward taps and receives the 8 upward ones from the pixels above
    (the range weight is symmetric), neighbours cross lanes with DPP shifts. Both of its table forms (FINE: 4-entry blocks with
    u8 deltas, for gentle tables such as sigmaR = 2; COARSE: the LDS16 form; PLATEAU: the computed part of a STEEP table as it is —
    sigmaR <= 0.0625, the filter's usual range, default 0.02 — and the whole table of 8-bit clips; CUBIC: what lies between, 0.0625 < sigmaR < 0.42,
    a cubic per 128 entries + a correction byte per entry) against the tile kernel and the oracle, byte for
    byte: natural content, full-range white noise (every table entry, every weight handed down differs) and hard edges;
    the filter's default sigmaS = 3 (radius 5, step 2: three tap distances, bilateral_walk36_kernel) in all three table forms;
    strips narrower / wider than a wave's 54 / 58 / 60 output columns, bands that end inside a ring period, planes barely
    larger than the taps."""
    sS, sR = sig
    for shape in [(120, 200), (67, 131), (7, 9), (230, 58), (109, 117), (8, 64)]:
        if content == "natural":
            src = fx.tiled_natural(shape, dtype, 1)
        elif content == "noise":
            src = fx.splitmix64_plane(5 + shape[0], shape, dtype)
        else:
            src = np.where((np.add.outer(np.arange(shape[0]) // 5, np.arange(shape[1]) // 7) & 1) == 0, 0, {np.uint8: 255, np.uint16: 65535}.get(dtype, 1)).astype(dtype)
        import vszip_amd

        try:
            (a,), cfgs = _gpu(dev, [src], sS, sR, algorithm=[2])
        except vszip_amd.VszipError as e:
            assert "plane too small" in str(e)  # planes not larger than the taps are rejected (bilateral.zig(vs):206-209)
            continue
        c = cfgs[0]
        dev.set_option("VSZIP_BILATERAL_NO_FINE", 1)
        (b,), _ = _gpu(dev, [src], sS, sR, algorithm=[2])
        dev.set_option("VSZIP_BILATERAL_NO_FINE", 0)
        dev.set_option("VSZIP_BILATERAL_NO_WALK", 1)
        (t,), _ = _gpu(dev, [src], sS, sR, algorithm=[2])
        dev.set_option("VSZIP_BILATERAL_NO_WALK", 0)
        want = oracle.bilateral_plane(src, c[0], c[1], c[2], c[3], c[4], c[5])
        # (f32 planes, round 3: the 8K RGBS pipeline's Bilateral stage takes the walk kernel too; compared as bit patterns)
        assert np.array_equal(a.view(np.uint8), want.view(np.uint8)), (sig, content, shape, dtype, int((a != want).sum()))
        assert np.array_equal(b.view(np.uint8), want.view(np.uint8)), (sig, content, shape, dtype, "coarse")
        assert np.array_equal(t.view(np.uint8), want.view(np.uint8)), (sig, content, shape, dtype, "tile kernel")


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_large_yuv420_batches_cross_the_launch_table(dev, oracle, dtype):
    """70 frames of YUV 4:2:0 in one call = 210 planes: more than one launch table (192 planes), luma and chroma with
    different radius / step (two launches per table, planes grouped by range-table content), every plane against the
    oracle."""
    shapes = [(64, 96), (32, 48), (32, 48)]
    base = [fx.tiled_natural(s, dtype, i) for i, s in enumerate(shapes)]
    hist = 65536
    cfg = dev.bilateral_cfg([2], [2], yuv=True, ssw=1, ssh=1, hist_len=hist)
    planes, idx = [], []
    for f in range(70):
        for i, p in enumerate(base):
            planes.append(np.ascontiguousarray(np.roll(p, f, axis=1)))
            idx.append(i)
    srcs = [dev.upload(p) for p in planes]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
    dev.bilateral(srcs, dsts, cfg, idx)
    want = {}
    for k in (0, 1, 2, 70 * 3 - 3, 70 * 3 - 2, 70 * 3 - 1, 191, 192, 193, 100):
        c = cfg[idx[k]]
        w = oracle.bilateral_plane(planes[k], c.sigmaS, c.sigmaR, c.algorithm, c.radius, c.step, c.pbficnum)
        assert np.array_equal(dev.download(dsts[k]).view(np.uint8), w.view(np.uint8)), k
    # whole-batch check: away from the left / right borders the filter commutes with the roll that made frame f
    first = [dev.download(dsts[i]) for i in range(3)]
    for k in range(0, 210, 7):
        f, i = divmod(k, 3)
        assert np.array_equal(np.roll(first[i], f, axis=1)[:, 8 + f:-8].view(np.uint8), dev.download(dsts[k])[:, 8 + f:-8].view(np.uint8)), k
    dev.bilateral_free(cfg)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float16])
@pytest.mark.parametrize("sig", [(7, 2), (16, 0.5)], ids=["luma_pbfic_chroma_taps", "all_pbfic_two_layer_counts"])
def test_algorithm1_planes_of_a_batch_share_their_launches(dev, oracle, dtype, sig):
    """Round 4: the algorithm-1 planes of a call are gathered and run as one batch of launches. A YUV 4:2:0 clip at sigmaS = 7 sigmaR = 2 auto-selects PBFIC
    for the luma and the tap kernel for the chroma (sigmaS halves there) — interleaved in the call, 70 frames so that a batch (64 planes) fills and a second
    one follows; at sigmaS = 16 every plane is PBFIC with two different layer counts (two layer-value tables in one launch). Every checked plane against the oracle."""
    sS, sR = sig
    shapes = [(72, 112), (36, 56), (36, 56)]
    base = [fx.tiled_natural(s, dtype, i) for i, s in enumerate(shapes)]
    hist = 256 if dtype == np.uint8 else 65536
    cfg = dev.bilateral_cfg([sS], [sR], yuv=True, ssw=1, ssh=1, hist_len=hist)
    algs = [cfg[i].algorithm for i in range(3)]
    nums = [cfg[i].pbficnum for i in range(3)]
    assert algs[0] == 1 and (algs[1] == 2 if sS == 7 else (algs[1] == 1 and nums[1] != nums[0])), (algs, nums)
    frames = 70
    planes, idx = [], []
    for f in range(frames):
        for i, p in enumerate(base):
            planes.append(np.ascontiguousarray(np.roll(p, f, axis=1)))
            idx.append(i)
    srcs = [dev.upload(p) for p in planes]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
    peak = 255.0 if dtype == np.uint8 else None
    dev.bilateral(srcs, dsts, cfg, idx, peak=peak)
    for k in (0, 1, 2, 3, 3 * 63, 3 * 64, 3 * 64 + 1, 3 * 65 + 2, 3 * frames - 3, 3 * frames - 1, 100):
        c = cfg[idx[k]]
        w = oracle.bilateral_plane(planes[k], c.sigmaS, c.sigmaR, c.algorithm, c.radius, c.step, c.pbficnum)
        assert np.array_equal(dev.download(dsts[k]).view(np.uint8), w.view(np.uint8)), (k, c.algorithm)
    dev.bilateral_free(cfg)
