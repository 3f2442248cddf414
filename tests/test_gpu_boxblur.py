"""GPU parity: vszip_boxblur (HIP, through the C ABI) vs the CPU oracle, bit-exact.

Cases follow the reference's tests/test_boxblur.py: radii sweep, odd/tiny
geometries, stride/offset handling (:122-128), planes independence (:111-121).
"""
import os

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


@pytest.fixture(scope="module")
def dev_shfl():
    import vszip_amd

    d = vszip_amd.Device(0)
    d.set_option("VSZIP_SCAN_MODE", 1)
    d._opt_saved.clear()  # (this context keeps it: conftest's restore leaves it alone)
    yield d
    d.close()


def _run(dev, src, *args, align=32):
    s = dev.upload(src, align)
    d = dev.empty(src.shape[0], src.shape[1], src.dtype, align)
    dev.lib.vszip_dev_memset(dev.ctx, d.ptr, 0xCD, d.nbytes)
    dev.boxblur([s], [d], *args)
    return dev.download(d)


CT_SHAPES = [(320, 640), (319, 639), (47, 53), (270, 480), (1080, 1920), (90, 1500)]


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("r", [1, 2, 3, 5, 8, 9, 13, 16, 17, 22])
def test_ct_int_matches_oracle(dev, oracle, dtype, r):
    for shape in CT_SHAPES:
        if 2 * r >= min(shape):
            continue
        src = fx.splitmix64_plane(1000 + r, shape, dtype)
        got = _run(dev, src, r, 1, r, 1)
        want = oracle.boxblur(src, r, 1, r, 1)
        assert np.array_equal(got, want), (dtype, r, shape, int((got != want).sum()))


@pytest.mark.parametrize("r", [1, 13, 22])
def test_ct_int_shuffle_scan_agrees(dev_shfl, oracle, r):
    src = fx.splitmix64_plane(77, (200, 1000), np.uint16)
    assert np.array_equal(_run(dev_shfl, src, r, 1, r, 1), oracle.boxblur(src, r, 1, r, 1))


def test_ct_int_natural_and_extremes(dev, oracle):
    nat = fx.tiled_natural((540, 960), np.uint16)
    assert np.array_equal(_run(dev, nat, 13, 1, 13, 1), oracle.boxblur(nat, 13, 1, 13, 1))
    for fill in (0, 65535):
        flat = np.full((100, 300), fill, np.uint16)
        assert np.array_equal(_run(dev, flat, 13, 1, 13, 1), oracle.boxblur(flat, 13, 1, 13, 1))


@pytest.mark.parametrize("align", [1, 3, 8, 32])
def test_ct_int_stride_and_alignment(dev, oracle, align):
    """stride != width and rows that are not 16-byte aligned take the scalar load path."""
    src = fx.splitmix64_plane(5, (123, 333), np.uint16)
    assert np.array_equal(_run(dev, src, 13, 1, 13, 1, align=align), oracle.boxblur(src, 13, 1, 13, 1))
    src8 = fx.splitmix64_plane(6, (77, 201), np.uint8)
    assert np.array_equal(_run(dev, src8, 7, 1, 7, 1, align=align), oracle.boxblur(src8, 7, 1, 7, 1))


def test_batch_of_planes_yuv420(dev, oracle):
    """One call over Y,U,V of several frames (mixed plane sizes)."""
    shapes = [(216, 384), (108, 192), (108, 192)] * 3
    srcs = [fx.splitmix64_plane(40 + i, s, np.uint16) for i, s in enumerate(shapes)]
    ds = [dev.upload(a) for a in srcs]
    dd = [dev.empty(a.shape[0], a.shape[1], a.dtype) for a in srcs]
    dev.boxblur(ds, dd, 13, 1, 13, 1)
    for a, d in zip(srcs, dd):
        assert np.array_equal(dev.download(d), oracle.boxblur(a, 13, 1, 13, 1))


def test_errors(dev):
    import vszip_amd

    src = dev.upload(np.zeros((20, 20), np.uint16))
    dst = dev.empty(20, 20, np.uint16)
    with pytest.raises(vszip_amd.VszipError, match="nothing to be performed"):
        dev.boxblur([src], [dst], 0, 1, 0, 1)
    with pytest.raises(vszip_amd.VszipError, match="hradius too large"):
        dev.boxblur([src], [dst], 10, 1, 10, 1)


# ---- CT float and RT paths -------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("r", [1, 2, 7, 13, 22])
def test_ct_float_matches_oracle(dev, oracle, dtype, r):
    for shape in [(120, 200), (67, 131), (300, 500)]:
        if 2 * r >= min(shape):
            continue
        src = fx.splitmix64_plane(50 + r, shape, dtype)
        got = _run(dev, src, r, 1, r, 1)
        want = oracle.boxblur(src, r, 1, r, 1)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dtype, r, shape)


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("r", [1, 3, 4, 5, 8, 12, 13, 16, 17, 18])
def test_ct_float_ring_interior_and_border_strips(dev, oracle, dtype, r):
    """Planes large enough for the register-ring kernel (interior) + the tile kernel (four border strips):
    several column tiles and bands, widths that leave a ragged right strip; r = 18 stays on the tile kernel."""
    for shape in [(333, 777), (150, 1300)]:
        src = fx.splitmix64_plane(90 + r, shape, dtype)
        got = _run(dev, src, r, 1, r, 1)
        want = oracle.boxblur(src, r, 1, r, 1)
        bad = got.view(np.uint8) != want.view(np.uint8)
        assert not bad.any(), (dtype, r, shape, np.argwhere(bad)[:4].tolist())


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_rt_float_pass_chain_kernels(dev, oracle, dtype, monkeypatch):
    """Round 3: float planes with 2 ... 5 passes on an axis run them in ONE launch per axis (a chain of running sums, each stage R samples
    behind its producer: boxblur_rt_float_vchain_kernel / _hchain_kernel). Bit for bit the one-launch-per-pass kernels and the oracle: lines
    barely longer than 2 R + 2, lines shorter than one prefetch group / tile, widths and heights that are not whole tiles, line lengths around
    the multiples of 64 and of the prefetch depth, radii whose ring wraps at every phase, f16 planes (every stage rounds to f16), more than five
    passes (6 = 3 + 3, 7 = 4 + 3, 11 = 5 + 5 + ... chained launches), lines too short for the chain (a launch per pass)."""
    cases = [([(72, 208), (36, 104), (36, 104)], (1, 2, 1, 2)), ([(300, 333)], (2, 3, 2, 2)), ([(70, 1100)], (3, 4, 0, 0)), ([(700, 96)], (0, 0, 4, 2)),
             ([(135, 251), (67, 125)], (5, 3, 5, 3)), ([(233, 130)], (7, 3, 8, 5)), ([(40, 35), (35, 40)], (8, 2, 8, 2)), ([(90, 640)], (1, 5, 1, 4)),
             ([(19, 40)], (2, 2, 2, 2)), ([(18, 18)], (8, 2, 8, 3)), ([(64, 64)], (3, 3, 3, 3)), ([(65, 63)], (4, 2, 2, 5)), ([(128, 129)], (6, 3, 6, 2)),
             ([(81, 257)], (30, 2, 25, 2)), ([(150, 191)], (13, 5, 13, 5)), ([(48, 48)], (2, 6, 2, 6)), ([(16, 300)], (9, 2, 9, 2)), ([(33, 17)], (1, 3, 1, 3)),
             ([(100, 200)], (5, 3, 0, 0)), ([(100, 200)], (0, 0, 5, 3)), ([(60, 90)], (1, 11, 1, 7)), ([(70, 333)], (3, 8, 0, 0)), ([(200, 40)], (0, 0, 2, 12)), ([(97, 1)], (0, 0, 3, 2)), ([(1, 97)], (3, 2, 0, 0))]
    dev.set_option("VSZIP_RT_FCHAIN_ALL", 1)  # (the horizontal chain is for calls of 12 000+ rows by default)
    for shapes, args in cases:
        planes = [fx.splitmix64_plane(31 + i, sh, dtype) if i % 2 == 0 else fx.tiled_natural(sh, dtype, 1) for i, sh in enumerate(shapes)]

        def run():
            srcs = [dev.upload(p) for p in planes]
            dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
            dev.boxblur(srcs, dsts, *args)
            return [dev.download(d) for d in dsts]

        try:
            got = run()
        except Exception as e:  # (a radius the filter refuses for this plane: the same error either way)
            dev.set_option("VSZIP_RT_NO_FCHAIN", 1)
            with pytest.raises(type(e)):
                run()
            dev.set_option("VSZIP_RT_NO_FCHAIN", 0)
            continue
        dev.set_option("VSZIP_RT_NO_FCHAIN", 1)
        per_pass = run()
        dev.set_option("VSZIP_RT_NO_FCHAIN", 0)
        for p, a, b in zip(planes, got, per_pass):
            want = oracle.boxblur(p, *args)
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (shapes, args, int((a != b).sum()), np.argwhere(a != b)[:4].tolist())
            assert np.array_equal(a.view(np.uint8), want.view(np.uint8)), (shapes, args, "oracle", int((a != want).sum()))


def test_ct_float_ring_batch_yuv420(dev, oracle):
    shapes = [(432, 768), (216, 384), (216, 384)] * 2
    srcs = [fx.splitmix64_plane(140 + i, s, np.float32) for i, s in enumerate(shapes)]
    ds = [dev.upload(a) for a in srcs]
    dd = [dev.empty(a.shape[0], a.shape[1], a.dtype) for a in srcs]
    dev.boxblur(ds, dd, 13, 1, 13, 1)
    for a, d in zip(srcs, dd):
        assert np.array_equal(dev.download(d).view(np.uint8), oracle.boxblur(a, 13, 1, 13, 1).view(np.uint8))


def test_ct_ring_batch_wider_than_one_block_table(dev, oracle):
    """192 planes of 40 x 15360 u16 need 12288 (plane, band, column tile) blocks — more than one launch's block table
    (8192): the batch is split over launches instead of failing. Every plane against the oracle's result (one content,
    so one oracle run) — a plane skipped or done twice with another plane's geometry would differ."""
    src = fx.splitmix64_plane(77, (40, 15360), np.uint16)
    want = oracle.boxblur(src, 13, 1, 13, 1)
    s = dev.upload(src)
    dsts = [dev.empty(40, 15360, np.uint16) for _ in range(192)]
    dev.boxblur([s] * 192, dsts, 13, 1, 13, 1)
    for i, d in enumerate(dsts):
        assert np.array_equal(dev.download(d), want), i


RT_CASES = [(23, 1, 23, 1), (40, 1, 40, 1), (4, 1, 9, 1), (9, 1, 4, 1), (5, 3, 5, 3), (5, 1, 5, 2), (5, 2, 5, 1), (0, 0, 7, 1), (7, 1, 0, 0), (6, 2, 3, 3)]


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float16])
@pytest.mark.parametrize("case", RT_CASES, ids=[f"h{c[0]}x{c[1]}_v{c[2]}x{c[3]}" for c in RT_CASES])
def test_rt_matches_oracle(dev, oracle, dtype, case):
    for shape in [(120, 200), (97, 131)]:
        src = fx.splitmix64_plane(70, shape, dtype)
        got = _run(dev, src, *case)
        want = oracle.boxblur(src, *case)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dtype, case, shape)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("r", [23, 30, 63, 64, 100, 250])
def test_rt_horizontal_mirror_extended_rows(dev, oracle, dtype, r, monkeypatch):
    """The horizontal RT pass on rows that are whole lane groups: the prefix runs over the mirror-extended row (halo
    lanes hold reversed real groups), every window a two-term difference. Widths from one chunk to several, with halos that
    fill less / more than a chunk, saturated rows (the 32-bit scale must not overflow), several passes; against the oracle
    and against the clamped-terms kernel (VSZIP_RT_NO_VIRT)."""
    V = 16 // np.dtype(dtype).itemsize
    for w in (2 * r + 1, 512, 1040, 1920 + 2 * V):  # (the filter needs 2 r < width)
        w = (max(w, 2 * r + 1) + V - 1) // V * V
        for passes in (1, 2):
            src = fx.splitmix64_plane(300 + r, (9, w), dtype)
            src[3] = np.iinfo(dtype).max
            src[4, : w // 2] = 0
            got = _run(dev, src, r, passes, 0, 0)
            want = oracle.boxblur(src, r, passes, 0, 0)
            assert np.array_equal(got, want), (dtype, r, w, passes, np.argwhere(got != want)[:4].tolist())
            with dev.variant(VSZIP_RT_NO_VIRT=1) as built:  # (a development variant since round 5)
                if built:
                    assert np.array_equal(got, _run(dev, src, r, passes, 0, 0)), (dtype, r, w, passes)


def test_rt_reference_golden_rgbs(dev):
    """RGBS|full|hpasses=2,hradius=6,vpasses=3,vradius=3 and RGBS|full|hradius=2,vradius=2 (reference goldens)."""
    g = fx.ref_goldens()["exact"]["boxblur"]
    for key, args in (("RGBS|full|hpasses=2,hradius=6,vpasses=3,vradius=3", (6, 2, 3, 3)), ("RGBS|full|hradius=2,vradius=2", (2, 1, 2, 1))):
        for p in range(3):
            st = fx.plane_stats(_run(dev, np.ascontiguousarray(fx.crop_rgbs()[p]), *args))
            for k in ("avg", "min", "max"):
                assert st[k] == pytest.approx(g[key][f"p{p}"][k], rel=1e-6, abs=1e-9)


def test_torch_tensors_on_a_torch_stream(oracle):
    """PyTorch as plumbing: planes are torch CUDA tensors (data_ptr), the kernels are enqueued on a
    torch stream handed over with vszip_ctx_set_stream, and torch sees the result after syncing
    that stream."""
    import torch

    import vszip_amd

    d = vszip_amd.Device(0)
    st = torch.cuda.Stream()
    d.set_stream(st.cuda_stream)
    src_np = fx.splitmix64_plane(31, (270, 480), np.uint16)
    with torch.cuda.stream(st):
        src = torch.from_numpy(src_np.astype(np.int32)).to("cuda").to(torch.int16)  # bit pattern of the u16 samples
        dst = torch.empty_like(src)
        s = d.wrap(src.data_ptr(), 270, 480, src.stride(0), np.uint16)
        o = d.wrap(dst.data_ptr(), 270, 480, dst.stride(0), np.uint16)
        d.boxblur([s], [o], 13, 1, 13, 1)
    st.synchronize()
    got = dst.cpu().numpy().view(np.uint16)
    assert np.array_equal(got, oracle.boxblur(src_np, 13, 1, 13, 1))
    d.close()


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_rt_fused_multipass_equals_per_pass(dev, oracle, dtype, monkeypatch):
    """Round 3: the opt-in fused multi-pass kernel (VSZIP_RT_FUSED=1: a row stays in registers across the passes of an axis, the
    vertical axis runs between two transposes) against the per-pass kernels and the oracle, bit for bit: 2 .. 5 passes, radii up to
    the fused kernel's 119, widths that are / are not whole thread groups, YUV 4:2:0 batches."""
    for shapes, args in [([(72, 208), (36, 104), (36, 104)], (5, 3, 5, 3)), ([(64, 4096)], (13, 5, 0, 0)), ([(300, 96)], (0, 0, 13, 5)),
                         ([(135, 250), (67, 125)], (7, 2, 9, 4)), ([(260, 300)], (119, 2, 100, 3)), ([(40, 33), (33, 40)], (3, 4, 2, 3))]:
        planes = [fx.splitmix64_plane(3 + i, sh, dtype) for i, sh in enumerate(shapes)]
        def run():
            srcs = [dev.upload(p) for p in planes]
            dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
            dev.boxblur(srcs, dsts, *args)
            return [dev.download(d) for d in dsts]

        base = run()
        fx.set_dev_option(dev, "VSZIP_RT_FUSED", 1)
        fused = run()
        fx.set_dev_option(dev, "VSZIP_RT_FUSED", 0)
        for p, a, b in zip(planes, base, fused):
            want = oracle.boxblur(p, *args)
            assert np.array_equal(b, want), (shapes, args, int((b != want).sum()))
            assert np.array_equal(a, want)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_rt_small_radius_multipass_kernels(dev, oracle, dtype, monkeypatch):
    """Round 3: BoxBlur(hradius <= 16, hpasses >= 2) — how scripts approximate a Gaussian — runs ALL horizontal passes in one launch (a wave keeps
    its row in LDS: boxblur_rt_hsmall_kernel) and two vertical passes in one launch as well (a chain of stages with LDS rings, the plane's first
    rows run first for the stages' E_0 constants: boxblur_rt_vsmall_kernel; three and four stages with VSZIP_RT_VSMALL_MAX). Against the oracle
    and the one-launch-per-pass kernels, bit for bit: radii 1 .. 16 (vertical chain: 1 .. 8), widths that are not whole 8-sample groups, planes barely taller
    than the chain's reach, bands that start at the top / inside it / below it, the bottom rows' mirrored windows."""
    cases = [([(72, 208), (36, 104), (36, 104)], (1, 2, 1, 2)), ([(300, 333)], (2, 3, 2, 2)), ([(64, 4096)], (3, 4, 0, 0)), ([(700, 96)], (0, 0, 4, 2)),
             ([(135, 251), (67, 125)], (5, 2, 6, 2)), ([(533, 130)], (7, 3, 8, 2)), ([(40, 35), (35, 40)], (8, 2, 8, 2)), ([(290, 640)], (1, 4, 1, 2)),
             ([(19, 40)], (2, 2, 2, 2)), ([(1080, 520)], (2, 2, 3, 2)),
             # radii 9 .. 16: the horizontal kernel reads two neighbour groups on either side (13 x 5 passes is the reference README's third benchmark)
             ([(72, 300), (36, 150)], (13, 5, 13, 5)), ([(90, 257)], (16, 2, 9, 2)), ([(64, 333)], (11, 3, 0, 0)), ([(50, 35)], (9, 4, 1, 2)), ([(40, 4104)], (12, 2, 0, 0)),
             # the vertical pass chain (3 ... 5 stages, more passes in several chains; radii to 30; lines of 2 R + 2 rows up; widths that are not whole 4-sample groups)
             ([(200, 70)], (0, 0, 2, 3)), ([(333, 131)], (0, 0, 5, 3)), ([(64, 64)], (0, 0, 3, 4)), ([(96, 35)], (0, 0, 1, 5)), ([(130, 200), (65, 100)], (2, 2, 3, 7)),
             ([(62, 48)], (0, 0, 30, 2)), ([(150, 41)], (0, 0, 22, 3)), ([(28, 260)], (0, 0, 13, 2)), ([(300, 5)], (0, 0, 4, 6)), ([(45, 3)], (0, 0, 2, 3)), ([(1080, 36)], (0, 0, 13, 5))]
    dev_variants = fx.has_dev_variants(dev)
    for shapes, args in cases:
        planes = [fx.splitmix64_plane(11 + i, sh, dtype) if i % 2 == 0 else fx.tiled_natural(sh, dtype, 1) for i, sh in enumerate(shapes)]

        def run():
            srcs = [dev.upload(p) for p in planes]
            dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
            dev.boxblur(srcs, dsts, *args)
            return [dev.download(d) for d in dsts]

        got = run()
        vsm = got
        if dev_variants:
            with dev.options(VSZIP_RT_VSMALL=1):  # two vertical stages in one launch (a development variant since round 4)
                vsm = run()
        with dev.options(VSZIP_RT_ICHAIN_ALL=1):  # the vertical pass chain wherever it can run (by default: 3+ passes, batches of 900+ column groups, 8-bit planes from five passes on)
            chain_all = run()
            chain_2 = chain_all
            if dev_variants:
                with dev.options(VSZIP_RT_NO_VSMALL=1):  # ... and for two passes of a small radius too
                    chain_2 = run()
        with dev.options(VSZIP_RT_NO_HSMALL=1, VSZIP_RT_NO_ICHAIN=1):
            per_pass = run()
        for p, a, b, c, c2, v in zip(planes, got, per_pass, chain_all, chain_2, vsm):
            want = oracle.boxblur(p, *args)
            assert np.array_equal(a, want), (shapes, args, int((a != want).sum()))
            assert np.array_equal(v, want), (shapes, args, "two vertical stages", int((v != want).sum()))
            assert np.array_equal(b, want), (shapes, args, "per pass")
            assert np.array_equal(c, want), (shapes, args, "pass chain", int((c != want).sum()), np.argwhere(c != want)[:3].tolist())
            assert np.array_equal(c2, want), (shapes, args, "pass chain, two stages", int((c2 != want).sum()))
    # three and four vertical stages (not the default: measured slower than a launch per pass)
    if not dev_variants:
        return
    fx.set_dev_option(dev, "VSZIP_RT_VSMALL", 1)
    fx.set_dev_option(dev, "VSZIP_RT_VSMALL_MAX", 4)
    for shapes, args in [([(300, 333)], (2, 2, 2, 3)), ([(521, 96), (260, 48)], (0, 0, 3, 4)), ([(70, 520)], (1, 2, 1, 4)), ([(1080, 512)], (0, 0, 2, 3))]:
        planes = [fx.splitmix64_plane(17 + i, sh, dtype) for i, sh in enumerate(shapes)]
        srcs = [dev.upload(p) for p in planes]
        dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
        dev.boxblur(srcs, dsts, *args)
        for p, d in zip(planes, dsts):
            want = oracle.boxblur(p, *args)
            got = dev.download(d)
            assert np.array_equal(got, want), (shapes, args, int((got != want).sum()))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_rt_integer_chain_in_bands(dev, oracle, dtype):
    """Round 4: the integer vertical pass chain cut into bands of rows (boxblur_rt_ichain_kernel: zeroed rings P R rows above a band, the stages' E_0
    constants from a table that a first launch fills, the plane's ends as the mirror-extended source). Exact under any segmentation: against the oracle,
    the unbanded chain and one launch per pass — heights that leave a one-row last band, planes barely taller than the mirror extension needs, bands of
    different counts for luma and chroma, widths that are not whole 4-sample groups, saturated and zero rows, radii to 30, 2 ... 7 passes."""
    cases = [([(1080, 36)], (0, 0, 13, 5)), ([(541, 130), (270, 65), (270, 65)], (0, 0, 5, 3)), ([(400, 70)], (0, 0, 2, 3)), ([(385, 64)], (0, 0, 3, 4)),
             ([(257, 33)], (0, 0, 1, 5)), ([(900, 48)], (0, 0, 30, 2)), ([(700, 41)], (0, 0, 22, 3)), ([(513, 260)], (2, 2, 13, 2)), ([(640, 5)], (0, 0, 4, 6)),
             ([(300, 3)], (0, 0, 2, 7)), ([(1081, 20)], (1, 2, 13, 5)), ([(136, 50)], (0, 0, 9, 3)), ([(2160, 64), (1080, 32)], (0, 0, 5, 3)),
             # round 5: chains cut into nearly equal lengths bounded by the shortest plane's height, P (3 R + 2) <= h / 4: 5 passes as 3 + 2, 4 as 2 + 2, 5 as 2 + 2 + 1
             ([(540, 64)], (0, 0, 13, 5)), ([(540, 72)], (0, 0, 13, 4)), ([(540, 96), (270, 48), (270, 48)], (13, 5, 13, 5)), ([(1080, 40), (540, 20)], (0, 0, 13, 5))]
    for shapes, args in cases:
        planes = [fx.splitmix64_plane(41 + i, sh, dtype) if i % 2 == 0 else fx.tiled_natural(sh, dtype, 1) for i, sh in enumerate(shapes)]
        planes[0][3] = np.iinfo(dtype).max
        planes[0][-2] = 0
        planes[0][planes[0].shape[0] // 2, :] = np.iinfo(dtype).max

        def run():
            srcs = [dev.upload(p) for p in planes]
            dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
            dev.boxblur(srcs, dsts, *args)
            return [dev.download(d) for d in dsts]

        with dev.options(VSZIP_RT_ICHAIN_ALL=1):
            banded = run()
            with dev.options(VSZIP_RT_NO_BANDED=1):
                whole = run()
        with dev.options(VSZIP_RT_NO_ICHAIN=1):
            per_pass = run()
        for p, a, b, c in zip(planes, banded, whole, per_pass):
            want = oracle.boxblur(p, *args)
            assert np.array_equal(a, want), (dtype, shapes, args, "banded", int((a != want).sum()), np.argwhere(a != want)[:4].tolist())
            assert np.array_equal(b, want), (dtype, shapes, args, "whole columns")
            assert np.array_equal(c, want), (dtype, shapes, args, "per pass")


@pytest.mark.parametrize("r", [1, 2, 5, 7, 8, 13, 15, 16, 19, 20, 22])
def test_ct_u8_sixteen_pixels_a_lane(dev, oracle, r):
    """Round 4: 8-bit planes whose widths are whole 16-pixel groups take the ring kernel with 16 pixels a lane (r <= 19; 960-byte row segments instead of 480).
    Against the oracle and the 8-pixel instance (VSZIP_CT_U8_PX8=1), bit for bit: one-tile and multi-tile widths (992 output columns a wave), widths that end
    inside the last wave's mirrored lanes, planes barely taller than a ring period, YUV 4:2:0 batches, a batch with one plane that is NOT whole groups
    (everything falls back to 8 pixels), saturated rows."""
    batches = [[(96, 32)], [(80, 48)], [(70, 992)], [(64, 1008)], [(61, 2000)], [(270, 1920), (135, 960), (135, 960)], [(59, 4096)], [(120, 64), (120, 72)]]
    for shapes in batches:
        if min(min(sh) for sh in shapes) <= 2 * r:  # (the filter refuses 2 r >= the smallest plane dimension)
            continue
        planes = [fx.splitmix64_plane(70 + i + r, sh, np.uint8) if i % 2 == 0 else fx.tiled_natural(sh, np.uint8, 1) for i, sh in enumerate(shapes)]
        planes[0][5] = 255
        planes[0][-1] = 0

        def run():
            srcs = [dev.upload(p) for p in planes]
            dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
            dev.boxblur(srcs, dsts, r, 1, r, 1)
            return [dev.download(d) for d in dsts]

        wide = run()
        with dev.options(VSZIP_CT_U8_PX8=1):
            narrow = run()
        for p, a, b in zip(planes, wide, narrow):
            want = oracle.boxblur(p, r, 1, r, 1)
            assert np.array_equal(a, want), (r, shapes, p.shape, int((a != want).sum()), np.argwhere(a != want)[:4].tolist())
            assert np.array_equal(b, want), (r, shapes, p.shape, "8 pixels a lane")


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("r", [1, 5, 13, 19, 20, 22])
def test_ct_ring_planes_split_by_columns(dev, oracle, dtype, r):
    """Round 4: a plane whose width is not whole pixel groups is split by columns — the tiles whose every lane lies inside the plane go to the fast ring
    kernel (8 or 16 pixels a lane), the last one or two to the general form — instead of sending the whole batch to the general form. Widths around every
    boundary of the split for both lane widths (a tile is all inside iff t * TWO - HL + 64 * PX <= w), batches that mix aligned and unaligned planes
    (1918 x 1078 and 1366 x 768 YUV 4:2:0 clips: 959- and 683-sample chroma), one-sample-over and one-sample-short widths, bit for bit against the oracle
    and against the 8-pixels-a-lane path for 8-bit planes."""
    px = 8
    hl = (r + 1 + px - 1) // px * px
    hr = (r + px - 1) // px * px
    two8 = (64 - hl // px - hr // px) * px
    first8 = 64 * px - hl  # narrowest width with one all-inside tile
    widths = [first8 - 1, first8, first8 + 1, first8 + 7, first8 + two8 - 1, first8 + two8, first8 + two8 + 3, 2 * two8 + 5]
    if dtype == np.uint8:
        hl16 = (r + 1 + 15) // 16 * 16
        hr16 = (r + 15) // 16 * 16
        two16 = (64 - hl16 // 16 - hr16 // 16) * 16
        first16 = 64 * 16 - hl16
        widths += [first16 - 1, first16, first16 + 1, first16 + 9, first16 + two16 + 1]
    batches = [[(64, w)] for w in widths] + [[(270, 1918), (135, 959), (135, 959)], [(96, 1366), (64, 683), (64, 683), (96, 1280)], [(60, 1920), (60, 1913)]]
    for shapes in batches:
        if min(min(sh) for sh in shapes) <= 2 * r:
            continue
        planes = [fx.splitmix64_plane(170 + i + r, sh, dtype) if i % 2 == 0 else fx.tiled_natural(sh, dtype, 1) for i, sh in enumerate(shapes)]

        def run():
            srcs = [dev.upload(p) for p in planes]
            dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
            dev.boxblur(srcs, dsts, r, 1, r, 1)
            return [dev.download(d) for d in dsts]

        got = run()
        for p, a in zip(planes, got):
            want = oracle.boxblur(p, r, 1, r, 1)
            assert np.array_equal(a, want), (r, shapes, p.shape, int((a != want).sum()), np.argwhere(a != want)[:4].tolist())
        if dtype == np.uint8:
            with dev.options(VSZIP_CT_U8_PX8=1):
                narrow = run()
            for a, b in zip(got, narrow):
                assert np.array_equal(a, b), (r, shapes, "8 pixels a lane")


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("r", [1, 2, 5, 8, 13, 16, 19, 20, 22])
def test_horizontal_only_takes_the_ring_kernel(dev, oracle, dtype, r):
    """Round 4: BoxBlur(hradius = r, vradius = 0) on integer planes runs through the compile-time-radius ring kernel with a one-row window — the reference's
    run-time path computes the same closed form for a row. Against the oracle and against the RT row kernel (VSZIP_BOXBLUR_NO_CT_H=1), bit for bit: aligned and
    cropped widths (both lane widths, the column split), one-tile and many-tile rows, planes shorter than the ring kernel takes (they stay with the RT kernel),
    YUV 4:2:0 batches, saturated rows; hpasses = 2 and a vertical pass beside it stay on the RT path and agree as before."""
    batches = [[(64, 1920)], [(70, 1918)], [(61, 992)], [(64, 1009)], [(80, 496)], [(55, 2000)], [(40, 640)], [(270, 1920), (135, 960), (135, 960)], [(96, 1366), (64, 683), (64, 683)],
               [(59, 4096)], [(120, 64), (120, 72)]]
    for shapes in batches:
        if min(sh[1] for sh in shapes) <= 2 * r:
            continue
        planes = [fx.splitmix64_plane(270 + i + r, sh, dtype) if i % 2 == 0 else fx.tiled_natural(sh, dtype, 1) for i, sh in enumerate(shapes)]
        planes[0][3] = np.iinfo(dtype).max
        planes[0][-1] = 0

        def run(*args):
            srcs = [dev.upload(p) for p in planes]
            dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for p in planes]
            dev.boxblur(srcs, dsts, *args)
            return [dev.download(d) for d in dsts]

        got = run(r, 1, 0, 0)
        with dev.options(VSZIP_BOXBLUR_NO_CT_H=1):
            rt = run(r, 1, 0, 0)
        for p, a, b in zip(planes, got, rt):
            want = oracle.boxblur(p, r, 1, 0, 0)
            assert np.array_equal(a, want), (r, shapes, p.shape, int((a != want).sum()), np.argwhere(a != want)[:4].tolist())
            assert np.array_equal(b, want), (r, shapes, p.shape, "RT row kernel")
    p = fx.tiled_natural((96, 600), dtype, 2)
    for args in ((r, 2, 0, 0), (r, 1, 3, 1), (r, 1, 0, 1)):
        if 2 * args[2] >= 96:
            continue
        s, d = dev.upload(p), dev.empty(96, 600, dtype)
        dev.boxblur([s], [d], *args)
        assert np.array_equal(dev.download(d), oracle.boxblur(p, *args)), args
