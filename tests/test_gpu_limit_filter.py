"""GPU parity: vszip_limit_filter vs the CPU oracle, bit-exact for every sample type, plus the
reference's goldens (flt = vszip.BoxBlur(2, 2) of the source, as the reference builds them)."""
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


def _run(dev, flts, srcs, refs, dark, bright, elast, align=32):
    df = [dev.upload(np.ascontiguousarray(p), align) for p in flts]
    ds = [dev.upload(np.ascontiguousarray(p), align) for p in srcs]
    dr = [dev.upload(np.ascontiguousarray(p), align) for p in refs] if refs is not None else None
    dd = [dev.empty(p.shape[0], p.shape[1], p.dtype, align) for p in flts]
    dev.limit_filter(df, ds, dd, dark, bright, elast, dr)
    return [dev.download(d) for d in dd]


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float16])
@pytest.mark.parametrize("args", [(4, 4, 2), (16, 2, 4), (8, 16, 1.5), (1, 1, 1), (0, 0, 3)])
@pytest.mark.parametrize("with_ref,shape,align", [(False, (97, 203), 1), (True, (120, 256), 32)])
def test_matches_oracle(dev, oracle, dtype, args, with_ref, shape, align):
    is_float = np.dtype(dtype).kind == "f"
    src = fx.tiled_natural(shape, dtype, 0)
    flt = oracle.boxblur(np.ascontiguousarray(src), 2, 1, 2, 1)
    ref = oracle.boxblur(np.ascontiguousarray(src), 4, 1, 4, 1) if with_ref else None
    bits = 32 if is_float else 8 * np.dtype(dtype).itemsize
    dark = oracle.scale_value_from_8bit(args[0], is_float, bits, False)
    bright = oracle.scale_value_from_8bit(args[1], is_float, bits, True if not is_float else False)
    (got,) = _run(dev, [flt], [src], [ref] if with_ref else None, [dark], [bright], [args[2]], align)
    want = oracle.limit_filter(flt, src, ref, dark, bright, args[2])
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dtype, args, with_ref)


def test_reference_goldens(dev, oracle):
    g = fx.ref_goldens()["exact"]["limitfilter"]
    src = list(fx.crop_rgb24())
    flt = [oracle.boxblur(np.ascontiguousarray(p), 2, 1, 2, 1) for p in src]
    out = _run(dev, flt, src, None, [8.0] * 3, [8.0] * 3, [3.0] * 3)
    for p in range(3):
        st = fx.plane_stats(out[p])
        for k in ("avg", "min", "max"):
            assert st[k] == pytest.approx(g["RGB24|full|bright_thr=8,dark_thr=8,elast=3"][f"p{p}"][k], rel=1e-6, abs=1e-9)
    srcf = [np.ascontiguousarray(p) for p in fx.crop_rgbs()]
    fltf = [oracle.boxblur(p, 2, 1, 2, 1) for p in srcf]
    t = oracle.scale_value_from_8bit(8, True, 32, False)
    out = _run(dev, fltf, srcf, None, [t] * 3, [t] * 3, [3.0] * 3)
    for p in range(3):
        st = fx.plane_stats(out[p])
        for k in ("avg", "min", "max"):
            assert st[k] == pytest.approx(g["RGBS|full|bright_thr=8,dark_thr=8,elast=3"][f"p{p}"][k], rel=1e-6, abs=1e-9)
