"""GPU parity: vszip_plane_average / vszip_plane_minmax vs the CPU oracle.
Integer results bit-exact; float sums within 1e-12 relative (reduction order)."""
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


SHAPES = [(320, 640), (319, 639), (7, 13), (1080, 1920)]


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float16])
def test_average_matches_oracle(dev, oracle, dtype):
    for shape in SHAPES:
        a = fx.splitmix64_plane(21, shape, dtype)
        b = fx.splitmix64_plane(22, shape, dtype)
        if np.dtype(dtype).kind == "u":
            a[::3, ::5] = 100
        else:
            a[::3, ::5] = 1.0
        # (list lengths 0 after dropping -1, 1, 2, 3 -> 4, 6 -> 8: the kernel's compile-time sizes; the last: 40+ values, duplicates)
        for excl in ([-1], [100, 7], [1], [100, 7, 250], [100, 7, 3, 9, 250, 1], list(range(90, 130)) + [7, 7, 0, 1]):
            da, db = dev.upload(a), dev.upload(b)
            avg, diff = dev.plane_average([da], excl, [db])
            oavg, odiff = oracle.plane_average(a, excl, b)
            if np.dtype(dtype).kind == "u":
                assert avg[0] == oavg and diff[0] == odiff
            else:
                assert avg[0] == pytest.approx(oavg, rel=1e-12) and diff[0] == pytest.approx(odiff, rel=1e-12)
            avg2, none = dev.plane_average([da], excl)
            assert none is None and (avg2[0] == avg[0])


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float16])
@pytest.mark.parametrize("thr", [(0.0, 0.0), (0.1, 0.1), (0.4, 0.0), (0.0, 0.3), (1.0, 1.0)])
def test_minmax_matches_oracle(dev, oracle, dtype, thr):
    for shape in SHAPES[:3]:
        a = fx.tiled_natural(shape, dtype) if shape[0] > 10 else fx.splitmix64_plane(5, shape, dtype)
        b = fx.splitmix64_plane(23, shape, dtype)
        da, db = dev.upload(a), dev.upload(b)
        mn, mx, df = dev.plane_minmax([da], thr[0], thr[1], [db])
        omn, omx, odf = oracle.plane_minmax(a, thr[0], thr[1], b)
        assert (mn[0], mx[0]) == (omn, omx), (dtype, thr, shape)
        if np.dtype(dtype).kind == "u":
            assert df[0] == odf
        else:
            assert df[0] == pytest.approx(odf, rel=1e-12)


def test_ten_bit_and_batch(dev, oracle):
    planes = [fx.splitmix64_plane(30 + i, s, np.uint16) >> 6 for i, s in enumerate([(216, 384), (108, 192), (108, 192)])]
    ds = [dev.upload(p) for p in planes]
    mn, mx, _ = dev.plane_minmax(ds, 0.05, 0.05, bits=10)
    avg, _ = dev.plane_average(ds, [-1], bits=10)
    for i, p in enumerate(planes):
        o = oracle.plane_minmax(p, 0.05, 0.05, bits=10)
        assert (mn[i], mx[i]) == o[:2]
        assert avg[i] == oracle.plane_average(p, [-1], bits=10)[0]


def test_known_answers(dev):
    """reference tests/test_planeaverage.py:118-147, tests/test_planeminmax.py:99-110,228-236"""
    two = np.concatenate([np.full((32, 64), 1000, np.uint16), np.full((32, 64), 3000, np.uint16)], axis=1)
    d = dev.upload(two)
    assert dev.plane_average([d], [1000])[0][0] == 3000 / 65535
    assert dev.plane_average([d], [1000, 3000])[0][0] == 0.0
    q = dev.upload(np.full((32, 64), 1234, np.uint16))
    assert dev.plane_minmax([q], 1.0, 0.0)[0][0] == 65535
    assert dev.plane_minmax([q], 0.0, 1.0)[1][0] == 0
    p = np.full((32, 64), 200, np.uint8)
    p[:8, :] = 0
    dp = dev.upload(p)
    assert (dev.plane_minmax([dp], 0.2, 0.0)[0][0], dev.plane_minmax([dp], 0.3, 0.0)[0][0]) == (0, 200)


def test_average_u32(dev, oracle):
    """32-bit integer clips: PlaneAverage only, no exclude list (planeaverage.zig(vs):127)."""
    import vszip_amd

    rng = np.random.default_rng(3)
    a = rng.integers(0, 2**32, size=(67, 131), dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2**32, size=(67, 131), dtype=np.uint64).astype(np.uint32)
    avg, diff = dev.plane_average([dev.upload(a)], refs=[dev.upload(b)])
    oa, od = oracle.plane_average(a, ref=b)
    assert (avg[0], diff[0]) == (oa, od)
    full = np.full((5, 9), 0xFFFFFFFF, np.uint32)
    assert dev.plane_average([dev.upload(full)])[0][0] == oracle.plane_average(full)[0]
    with pytest.raises(vszip_amd.VszipError, match="exclude is not supported"):
        dev.plane_average([dev.upload(a)], exclude=[3])


@pytest.mark.parametrize("dtype", [np.uint16, np.float32, np.float16])
def test_minmax_single_read_and_its_second_sweep(dev, oracle, dtype, monkeypatch):
    """Round 3: the opt-in single-read path (VSZIP_MINMAX_SINGLE_READ=1) — a sample of every 16th row predicts a range of values
    around each threshold, one sweep counts what lies below the ranges and builds their histograms; planes whose sample misleads are
    flagged and go through the two histogram sweeps. Same answers as the default two sweeps, and as the oracle."""
    shape = (203, 331)
    nat = fx.tiled_natural(shape, dtype)
    # the sampled rows (8, 24, 40, ...) say "dark", the plane is bright: the prediction misses, the second sweep runs
    lie = fx.tiled_natural(shape, dtype)
    if np.dtype(dtype).kind == "u":
        lie = (lie // 4 + 40000).astype(dtype)
        lie[8::16, :] = (np.arange(shape[1]) % 300).astype(dtype)
    else:
        lie = (lie * 0.2 + 0.7).astype(dtype)
        lie[8::16, :] = (np.arange(shape[1]) % 300 / 4000).astype(dtype)
    flat = np.full(shape, 1234 if np.dtype(dtype).kind == "u" else 0.25, dtype)
    short = fx.splitmix64_plane(9, (5, 77), dtype)  # no sampled row at all
    planes = [nat, lie, flat, short, nat[:, :64].copy()]
    refs = [fx.splitmix64_plane(40 + i, p.shape, dtype) for i, p in enumerate(planes)]
    ds, dr = [dev.upload(p) for p in planes], [dev.upload(r) for r in refs]
    for thr in [(0.0, 0.0), (0.02, 0.02), (0.3, 0.1), (0.06, 0.9), (1.0, 1.0)]:
        fx.set_dev_option(dev, "VSZIP_MINMAX_SINGLE_READ", 0)
        mn, mx, df = dev.plane_minmax(ds, thr[0], thr[1], dr)
        fx.set_dev_option(dev, "VSZIP_MINMAX_SINGLE_READ", 1)
        mn2, mx2, df2 = dev.plane_minmax(ds, thr[0], thr[1], dr)
        mn3, mx3, _ = dev.plane_minmax(ds, thr[0], thr[1])  # (without a reference clip: the other instantiation)
        assert list(mn3) == list(mn2) and list(mx3) == list(mx2), thr
        assert list(mn) == list(mn2) and list(mx) == list(mx2) and list(df) == list(df2), thr
        for i, p in enumerate(planes):
            omn, omx, odf = oracle.plane_minmax(p, thr[0], thr[1], refs[i])
            assert (mn[i], mx[i]) == (omn, omx), (dtype, thr, i)
            assert df[i] == (odf if np.dtype(dtype).kind == "u" else pytest.approx(odf, rel=1e-12))


@pytest.mark.parametrize("dtype", [np.uint16, np.float32, np.float16])
def test_minmax_temporal_prediction_hits_and_misses(dev, oracle, dtype):
    """Round 6: a thresholded call of the same shape as the previous one sweeps its planes ONCE over the value ranges the previous call's answers
    predict (hist_sweep_kernel MODE 1); planes whose answers lie elsewhere - a scene cut - are flagged and take the two histogram sweeps.
    The same planes again (prediction exact), a slightly different frame (prediction close), inverted content (prediction wrong on every plane),
    constant planes, a mix of hits and misses in one call, other thresholds (another shape: no prediction), and VSZIP_MINMAX_NO_PREDICT=1:
    every result equals the oracle's."""
    import vszip_amd

    d = vszip_amd.Device(0)  # a context of its own: the prediction table is per context
    try:
        shapes = [(203, 331), (120, 200), (203, 331), (64, 96)]
        kind = np.dtype(dtype).kind
        frame_a = [fx.tiled_natural(s, dtype, i % 3) for i, s in enumerate(shapes)]
        if kind == "u":
            frame_b = [np.clip(p.astype(np.int32) + 37, 0, 65535).astype(dtype) for p in frame_a]  # the "next frame": a small shift
            frame_c = [(65535 - p).astype(dtype) for p in frame_a]                                  # a scene cut: everything elsewhere
            frame_d = [frame_a[0], frame_c[1], np.full(shapes[2], 777, dtype), frame_b[3]]           # hits, a miss and a constant plane in one call
        else:
            frame_b = [np.clip(p.astype(np.float32) + 0.0006, 0, 1).astype(dtype) for p in frame_a]
            frame_c = [(1.0 - p.astype(np.float32)).astype(dtype) for p in frame_a]
            frame_d = [frame_a[0], frame_c[1], np.full(shapes[2], 0.3, dtype), frame_b[3]]
        refs = [fx.splitmix64_plane(60 + i, s, dtype) for i, s in enumerate(shapes)]
        dr = [d.upload(r) for r in refs]

        def check(frame, thr, with_ref, tag):
            ds = [d.upload(p) for p in frame]
            mn, mx, df = d.plane_minmax(ds, thr[0], thr[1], dr if with_ref else None)
            for i, p in enumerate(frame):
                omn, omx, odf = oracle.plane_minmax(p, thr[0], thr[1], refs[i] if with_ref else None)
                assert (mn[i], mx[i]) == (omn, omx), (tag, dtype, thr, i, mn[i], omn, mx[i], omx)
                if with_ref:
                    assert df[i] == (odf if kind == "u" else pytest.approx(odf, rel=1e-12)), (tag, i)

        for with_ref in (False, True):
            thr = (0.1, 0.1)
            check(frame_a, thr, with_ref, "first call: two sweeps")
            check(frame_a, thr, with_ref, "the same frame: predicted")
            check(frame_b, thr, with_ref, "the next frame: predicted, close")
            check(frame_c, thr, with_ref, "scene cut: every plane flagged")
            check(frame_c, thr, with_ref, "after the cut: predicted again")
            check(frame_d, thr, with_ref, "hits, a miss and a constant plane")
            check(frame_a, (0.02, 0.3), with_ref, "other thresholds: another shape, two sweeps")
            check(frame_b, (0.02, 0.3), with_ref, "predicted under the new thresholds")
            check(frame_a, (1.0, 1.0), with_ref, "nothing qualifies")
            check(frame_b, (1.0, 1.0), with_ref, "nothing qualifies, predicted")
            check(frame_a, (0.999, 0.0), with_ref, "extreme quantile")
            check(frame_c, (0.999, 0.0), with_ref, "extreme quantile, predicted from the wrong end")
            with d.options(VSZIP_MINMAX_NO_PREDICT=1):
                check(frame_a, (0.999, 0.0), with_ref, "prediction off")
    finally:
        d.close()


def test_minmax_temporal_prediction_ranges_at_the_top_of_16_bits(dev, oracle):
    """A predicted range that would reach past 65 535: the packed sweep measures a sample's distance from the range's start modulo 2^16, so the DARK samples of a
    plane that also holds values next to 65 535 would wrap into such a range - the starts are kept at or below 65 536 - 512. Planes that are half 0 ... 300 and
    half 65 535 - (0 ... 300), thresholds that put the maximum (and, inverted, the minimum) into the top levels."""
    import vszip_amd

    d = vszip_amd.Device(0)
    try:
        rng = np.random.default_rng(77)
        shapes = [(120, 256), (96, 200), (64, 96)]

        def frame(k):
            out = []
            for i, (h, w) in enumerate(shapes):
                dark = rng.integers(0, 300, (h, w), dtype=np.int64)
                p = np.where(((np.arange(w)[None, :] + i + k) // 8) % 2 == 0, dark, 65535 - dark)
                out.append(p.astype(np.uint16))
            return out

        for invert in (False, True):
            for thr in ((0.1, 0.1), (0.3, 0.01), (0.45, 0.45), (0.0, 0.2)):
                for k in range(4):  # the first call of a shape runs the two sweeps, the others are predicted
                    f = [(65535 - p).astype(np.uint16) for p in frame(k)] if invert else frame(k)
                    before = d.get_option("VSZIP_STAT_MINMAX_FALLBACKS")
                    mn, mx, _ = d.plane_minmax([d.upload(p) for p in f], thr[0], thr[1])
                    for i, p in enumerate(f):
                        omn, omx, _ = oracle.plane_minmax(p, thr[0], thr[1], None)
                        assert (mn[i], mx[i]) == (omn, omx), (invert, thr, k, i, mn[i], omn, mx[i], omx)
                    # the frames of one sequence are draws from one distribution: their answers lie a few values apart, and a predicted call whose
                    # ranges hold them needs no second look
                    if k >= 1:
                        assert d.get_option("VSZIP_STAT_MINMAX_FALLBACKS") == before, (invert, thr, k)
    finally:
        d.close()


def test_minmax_predictions_of_interleaved_clips(dev, oracle):
    """The thresholded statistics of several clips of a filter graph arrive interleaved on one context: each signature (plane count and sizes, sample type,
    thresholds) keeps its own prediction table (four per plane group, least recently used replaced). Three clips in turn: from their second frames on every
    call is predicted and none needs the two sweeps; a fifth signature evicts the oldest, whose next call is a first call again."""
    import vszip_amd

    d = vszip_amd.Device(0)
    try:
        clips = [([(120, 200), (60, 100), (60, 100)], (0.1, 0.1)), ([(96, 160)], (0.05, 0.2)), ([(120, 200), (60, 100), (60, 100)], (0.3, 0.02))]

        def frame(ci, k):
            return [np.clip(fx.tiled_natural(s, np.uint16, (ci + i) % 3).astype(np.int32) + 11 * k, 0, 65535).astype(np.uint16) for i, s in enumerate(clips[ci][0])]

        def call(ci, k):
            f = frame(ci, k)
            thr = clips[ci][1]
            p0, f0 = d.get_option("VSZIP_STAT_MINMAX_PREDICTED"), d.get_option("VSZIP_STAT_MINMAX_FALLBACKS")
            mn, mx, _ = d.plane_minmax([d.upload(p) for p in f], thr[0], thr[1])
            for i, p in enumerate(f):
                omn, omx, _ = oracle.plane_minmax(p, thr[0], thr[1], None)
                assert (mn[i], mx[i]) == (omn, omx), (ci, k, i)
            return d.get_option("VSZIP_STAT_MINMAX_PREDICTED") - p0, d.get_option("VSZIP_STAT_MINMAX_FALLBACKS") - f0

        for k in range(4):
            for ci in range(3):
                assert call(ci, k) == ((1, 0) if k else (0, 0)), (ci, k)
        # two more signatures: five in all, four tables - clip 0's (the least recently used) goes
        clips.append(([(64, 96)], (0.2, 0.2)))
        clips.append(([(64, 96)], (0.25, 0.2)))
        assert call(3, 0) == (0, 0) and call(4, 0) == (0, 0)
        assert call(0, 4) == (0, 0)   # a first call again
        assert call(0, 5) == (1, 0)
        assert call(2, 4) == (1, 0)   # still there
    finally:
        d.close()


def test_minmax_temporal_prediction_on_10_bit_clips(dev, oracle):
    """10-bit samples in 16-bit words: the histogram is 1 024 values long, a predicted range of 512 covers half of it and may reach past its end"""
    shapes = [(203, 331), (120, 200), (77, 96)]
    f0 = [(fx.tiled_natural(s, np.uint16, i) >> 6).astype(np.uint16) for i, s in enumerate(shapes)]
    f1 = [np.clip(p.astype(np.int32) + 3, 0, 1023).astype(np.uint16) for p in f0]
    f2 = [(1023 - p).astype(np.uint16) for p in f0]
    for thr in ((0.1, 0.1), (0.0, 0.6), (0.9, 0.02)):
        for frame in (f0, f0, f1, f2, f2, f0):
            ds = [dev.upload(p) for p in frame]
            mn, mx, _ = dev.plane_minmax(ds, thr[0], thr[1], bits=10)
            for i, p in enumerate(frame):
                omn, omx, _ = oracle.plane_minmax(p, thr[0], thr[1], bits=10)
                assert (mn[i], mx[i]) == (omn, omx), (thr, i)


def test_minmax_temporal_prediction_across_batches(dev, oracle):
    """more than 192 planes a call: every batch of the queue keeps its own predictions (400 planes = 192 + 192 + 16), two rounds of two frames"""
    shapes = [(96, 160), (48, 80), (37, 53)]
    f0 = [fx.tiled_natural(shapes[i % 3], np.uint16, i % 3) if i % 4 else fx.splitmix64_plane(300 + i, shapes[i % 3], np.uint16) for i in range(400)]
    f1 = [np.roll(p, 3, axis=1) if i % 7 else (65535 - p).astype(np.uint16) for i, p in enumerate(f0)]
    for frame in (f0, f1, f0, f1):
        ds = [dev.upload(p) for p in frame]
        mn, mx, _ = dev.plane_minmax(ds, 0.05, 0.1)
        for i, p in enumerate(frame):
            omn, omx, _ = oracle.plane_minmax(p, 0.05, 0.1)
            assert (mn[i], mx[i]) == (omn, omx), i


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_many_planes_one_call_and_the_async_entry_points(dev, oracle, dtype):
    """Round 4: a call of more than 192 planes is queued as groups back to back with ONE synchronise at the end (results of every group in their own entries of
    the pinned array), and vszip_plane_average_async / _minmax_async write into the caller's pinned array and do not synchronise at all. 400 planes of
    three sizes (three groups: 192 + 192 + 16), every plane against the oracle; the async results equal the synchronous ones to the bit."""
    shapes = [(96, 160), (48, 80), (37, 53)]
    planes = [fx.splitmix64_plane(100 + i, shapes[i % 3], dtype) if i % 5 else fx.tiled_natural(shapes[i % 3], dtype, i % 3) for i in range(400)]
    refs = [fx.splitmix64_plane(900 + i, p.shape, dtype) for i, p in enumerate(planes)]
    ds, dr = [dev.upload(p) for p in planes], [dev.upload(r) for r in refs]
    is_int = np.dtype(dtype).kind == "u"
    avg, diff = dev.plane_average(ds, [-1, 7], dr)
    mn, mx, df = dev.plane_minmax(ds, 0.05, 0.1, dr)
    mn0, mx0, _ = dev.plane_minmax(ds)
    for i, p in enumerate(planes):
        oa, od = oracle.plane_average(p, [-1, 7], refs[i])
        omn, omx, odf = oracle.plane_minmax(p, 0.05, 0.1, refs[i])
        assert (mn[i], mx[i]) == (omn, omx), (dtype, i)
        assert (mn0[i], mx0[i]) == oracle.plane_minmax(p, 0.0, 0.0)[:2], (dtype, i)
        if is_int:
            assert (avg[i], diff[i], df[i]) == (oa, od, odf), (dtype, i)
        else:
            assert avg[i] == pytest.approx(oa, rel=1e-12) and diff[i] == pytest.approx(od, rel=1e-12) and df[i] == pytest.approx(odf, rel=1e-12)
    ra = dev.pinned_array((400, 4), np.float64)
    rm = dev.pinned_array((400, 4), np.float64)
    ra[...] = -1.0
    rm[...] = -1.0
    dev.plane_average_async(ds, ra, [-1, 7], dr)
    dev.plane_minmax_async(ds, rm, 0.05, 0.1, dr)  # queued behind the first: both reuse the context's scratch tables in stream order
    dev.sync()
    assert list(ra[:, 0]) == avg and list(ra[:, 1]) == diff
    assert list(rm[:, 0]) == mn and list(rm[:, 1]) == mx and list(rm[:, 2]) == df
    import vszip_amd

    with pytest.raises(vszip_amd.VszipError, match="pinned"):
        dev.plane_average_async(ds[:2], np.zeros((2, 4)), [-1])


def test_handover_many_planes_many_xcds(dev, oracle):
    """The last-workgroup hand-over of the thresholded PlaneMinMax (hist_sweep_kernel: tables built by all workgroups, read by the one whose ticket
    comes last — agent-scope atomics, no fence: planestats.hip "Hand-over inside a kernel"): 48 planes per launch, each swept by workgroups on
    all eight XCDs, twenty rounds on changing content; every plane of every round against the oracle."""
    shape = (540, 960)
    for rnd in range(20):
        planes = [np.roll(fx.tiled_natural(shape, np.uint16, i % 3), 37 * rnd + 11 * i, axis=1) ^ np.uint16(rnd * 257 + i) for i in range(48)]
        ds = [dev.upload(p) for p in planes]
        thr = [(0.02, 0.02), (0.3, 0.1), (0.06, 0.9)][rnd % 3]
        mn, mx, _ = dev.plane_minmax(ds, *thr)
        for i in range(0, 48, 5 if rnd else 1):
            omn, omx, _ = oracle.plane_minmax(planes[i], thr[0], thr[1], None)
            assert (mn[i], mx[i]) == (omn, omx), (rnd, i, thr)
