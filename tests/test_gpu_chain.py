"""vszip_chain_run: several pixel-filter stages over one plane table in one call, bit-identical to the same
stages called one after another with explicit intermediate planes."""
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


@pytest.mark.parametrize("dtype", [np.uint16, np.uint8, np.float32])
def test_chain_equals_sequential_calls(dev, oracle, dtype):
    shapes = [(120, 208), (60, 104), (60, 104)]
    frames = 2
    src = [fx.tiled_natural(s, dtype, p) for _ in range(frames) for p, s in enumerate(shapes)]
    src = [np.ascontiguousarray(np.roll(a, 3 * i, axis=1)) for i, a in enumerate(src)]
    slots = [0, 1, 2] * frames
    is_f = np.dtype(dtype).kind == "f"
    hist = 65536 if is_f else 1 << (8 * np.dtype(dtype).itemsize)
    peak = float(hist - 1)
    cfg = dev.bilateral_cfg([2], [0.05], yuv=True, ssw=1, ssh=1, hist_len=hist)
    lo = [0.1, 0.2, 0.2] if is_f else [peak * 0.1, peak * 0.3, peak * 0.3]
    hi = [0.9, 0.6, 0.6] if is_f else [peak * 0.9, peak * 0.6, peak * 0.6]
    lo, hi = [float(int(v)) if not is_f else v for v in lo], [float(int(v)) if not is_f else v for v in hi]
    stages = [{"bilateral": cfg, "peak": peak}, {"boxblur": (3, 1, 3, 1), "planes": (True, False, False)}, {"limiter": (lo, hi), "planes": (False, True, True)},
              {"boxblur": (1, 2, 1, 1)}]
    ds = [dev.upload(a) for a in src]
    dd = [dev.empty(a.shape[0], a.shape[1], a.dtype) for a in src]
    dev.chain_run(stages, ds, dd, slots)
    got = [dev.download(d) for d in dd]
    # sequential twin with explicit intermediates
    a = [dev.empty(x.shape[0], x.shape[1], x.dtype) for x in src]
    dev.bilateral(ds, a, cfg, slots, peak=peak)
    b = [dev.empty(x.shape[0], x.shape[1], x.dtype) if s == 0 else a[i] for i, (x, s) in enumerate(zip(src, slots))]
    luma = [i for i, s in enumerate(slots) if s == 0]
    chroma = [i for i, s in enumerate(slots) if s != 0]
    dev.boxblur([a[i] for i in luma], [b[i] for i in luma], 3, 1, 3, 1)
    c = [dev.empty(x.shape[0], x.shape[1], x.dtype) if s != 0 else b[i] for i, (x, s) in enumerate(zip(src, slots))]
    dev.limiter([b[i] for i in chroma], [c[i] for i in chroma], [lo[slots[i]] for i in chroma], [hi[slots[i]] for i in chroma])
    e = [dev.empty(x.shape[0], x.shape[1], x.dtype) for x in src]
    dev.boxblur(c, e, 1, 2, 1, 1)
    want = [dev.download(d) for d in e]
    for i in range(len(src)):
        assert np.array_equal(got[i].view(np.uint8), want[i].view(np.uint8)), (dtype, i)
    # and the first stage against the oracle, as an anchor
    first = dev.download(a[0])
    c0 = cfg[0]
    assert np.array_equal(first, oracle.bilateral_plane(src[0], c0.sigmaS, c0.sigmaR, c0.algorithm, c0.radius, c0.step, c0.pbficnum))
    dev.bilateral_free(cfg)


def test_chain_pass_through_and_errors(dev):
    import vszip_amd

    x = fx.splitmix64_plane(3, (40, 64), np.uint16)
    s, d = dev.upload(x), dev.empty(40, 64, np.uint16)
    dev.chain_run([{"boxblur": (2, 1, 2, 1), "planes": (False, True, True)}], [s], [d], [0])  # nothing filters slot 0: a copy
    assert np.array_equal(dev.download(d), x)
    with pytest.raises(vszip_amd.capi.VszipError):
        dev.chain_run([{"boxblur": (2, 1, 2, 1)}], [s], [d], [5])
