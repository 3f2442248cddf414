"""TEST INFRASTRUCTURE ONLY — ctypes/numpy front end of the CPU oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does (see oracle/oracle_common.h).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
import os as _os

_LIB_PATH = Path(_os.environ.get("VSZIP_ORACLE_LIB", _DIR / "liboracle_vszip.so"))  # tests/test_sanitizers.py: the ASan build

U8, U16, F16, F32 = 0, 1, 2, 3
_NP2DT = {np.dtype(np.uint8): U8, np.dtype(np.uint16): U16, np.dtype(np.float16): F16, np.dtype(np.float32): F32, np.dtype(np.uint32): 4}


def build(force: bool = False) -> Path:
    """Compile the oracle with g++ (make). Safe to call repeatedly."""
    srcs = list(_DIR.glob("*.cpp")) + list(_DIR.glob("*.h"))
    stale = (not _LIB_PATH.is_file()) or (
        srcs and max(p.stat().st_mtime for p in srcs) > _LIB_PATH.stat().st_mtime
    )
    if (force or stale) and "VSZIP_ORACLE_LIB" not in _os.environ:
        subprocess.run(["make", "-C", str(_DIR), "-j8"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.is_file():
            build()
        _lib = C.CDLL(str(_LIB_PATH))
        _declare(_lib)
    return _lib


def _declare(l: C.CDLL) -> None:
    vp, i, pd, dbl, f = C.c_void_p, C.c_int, C.c_ssize_t, C.c_double, C.c_float
    pdbl, pint = C.POINTER(C.c_double), C.POINTER(C.c_int)
    l.vszo_boxblur.argtypes = [i, vp, vp, pd, pd, i, i, i, i, i, i]
    l.vszo_boxblur.restype = i
    l.vszo_plane_average.argtypes = [i, vp, vp, pd, pd, i, i, C.POINTER(C.c_int32), i, i, pdbl, pdbl]
    l.vszo_plane_average.restype = i
    l.vszo_plane_minmax.argtypes = [i, vp, vp, pd, pd, i, i, f, f, i, pdbl, pdbl, pdbl]
    l.vszo_plane_minmax.restype = i
    l.vszo_bilateral_gs_lut.argtypes = [vp, i, dbl]
    l.vszo_bilateral_gs_lut.restype = None
    l.vszo_bilateral_gr_lut.argtypes = [vp, i, dbl, dbl]
    l.vszo_bilateral_gr_lut.restype = None
    l.vszo_bilateral_params.argtypes = [pdbl, i, pdbl, pint, pint, i, i, i, pint, pdbl, pint, pint, pint, pint, pint, pint]
    l.vszo_bilateral_params.restype = i
    l.vszo_bilateral_plane.argtypes = [i, vp, vp, vp, pd, pd, pd, i, i, i, i, i, vp, vp, dbl, i, f]
    l.vszo_bilateral_plane.restype = i
    l.vszo_ssimulacra2.argtypes = [C.POINTER(vp), C.POINTER(vp), pd, i, i, pdbl, pdbl]
    l.vszo_ssimulacra2.restype = dbl
    l.vszo_ssim_set_vec.argtypes = [i]
    l.vszo_ssim_set_vec.restype = i
    l.vszo_ssim_to_xyb.argtypes = [C.POINTER(vp), C.POINTER(vp), i, i]
    l.vszo_ssim_to_xyb.restype = None
    l.vszo_ssim_blur.argtypes = [vp, vp, i, i]
    l.vszo_ssim_blur.restype = None
    l.vszo_ssim_downscale.argtypes = [vp, vp, i, i]
    l.vszo_ssim_downscale.restype = None
    l.vszo_xpsnr_wsse.argtypes = [i, C.POINTER(vp), C.POINTER(vp), vp, vp, C.POINTER(C.c_uint64), pint, pint, C.POINTER(pd), i, i, C.c_uint32, i]
    l.vszo_xpsnr_wsse.restype = i
    l.vszo_xpsnr_frame.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, i]
    l.vszo_xpsnr_frame.restype = dbl
    l.vszo_xpsnr_avg.argtypes = [dbl, dbl, C.c_uint64, C.c_uint64, i, C.c_uint64]
    l.vszo_xpsnr_avg.restype = dbl
    l.vszo_eedi3_plane.argtypes = [vp, vp, vp, vp, pd, pd, pd, pd, i, i, i, i, f, f, f, i, i, i, i, f, f, f, i]
    l.vszo_eedi3_plane.restype = i


def dt_of(a: np.ndarray) -> int:
    return _NP2DT[a.dtype]


def _plane(a: np.ndarray):
    """(pointer, stride in elements) of a 2-D array whose rows are contiguous."""
    assert a.ndim == 2 and a.strides[1] == a.itemsize, "rows must be contiguous"
    assert a.strides[0] % a.itemsize == 0
    return a.ctypes.data_as(C.c_void_p), a.strides[0] // a.itemsize


def boxblur(src: np.ndarray, hradius=1, hpasses=1, vradius=1, vpasses=1) -> np.ndarray:
    h, w = src.shape
    dst = np.empty((h, w), dtype=src.dtype)
    sp, ss = _plane(src)
    dp, ds = _plane(dst)
    rc = lib().vszo_boxblur(dt_of(src), sp, dp, ss, ds, w, h, hradius, hpasses, vradius, vpasses)
    assert rc == 0
    return dst


def _bits_of(a: np.ndarray, bits=None) -> int:
    return bits if bits is not None else 8 * a.itemsize


def plane_average(src: np.ndarray, exclude=(), ref: np.ndarray | None = None, bits=None):
    """-> (avg, diff or None); peak = 2^bits - 1 (planeaverage.zig(vs):115)."""
    h, w = src.shape
    sp, ss = _plane(src)
    rp, rs = _plane(ref) if ref is not None else (None, 0)
    ex = (C.c_int32 * max(1, len(exclude)))(*exclude)
    avg, diff = C.c_double(), C.c_double()
    rc = lib().vszo_plane_average(dt_of(src), sp, rp, ss, rs, w, h, ex, len(exclude), _bits_of(src, bits), C.byref(avg), C.byref(diff))
    assert rc == 0
    return avg.value, (diff.value if ref is not None else None)


def plane_minmax(src: np.ndarray, minthr=0.0, maxthr=0.0, ref: np.ndarray | None = None, bits=None):
    """-> (min, max, diff or None)."""
    h, w = src.shape
    sp, ss = _plane(src)
    rp, rs = _plane(ref) if ref is not None else (None, 0)
    mn, mx, df = C.c_double(), C.c_double(), C.c_double()
    rc = lib().vszo_plane_minmax(dt_of(src), sp, rp, ss, rs, w, h, minthr, maxthr, _bits_of(src, bits), C.byref(mn), C.byref(mx), C.byref(df))
    assert rc == 0
    return mn.value, mx.value, (df.value if ref is not None else None)


def bilateral_params(sigmaS=(), sigmaR=(), algorithm=(), pbficnum=(), planes=(True, True, True), yuv=False, ssw=0, ssh=0):
    """bilateralCreate's derivation (bilateral.zig(vs):104-199). Array arguments follow
    hz.getArray: missing entries repeat the previous one; defaults sigmaR .02, algorithm 0, PBFICnum 0."""
    def arr3(vals, default):
        out = []
        for i in range(3):
            out.append(vals[i] if i < len(vals) else (default if i == 0 else out[i - 1]))
        return out
    sS = (C.c_double * 3)(*(list(sigmaS) + [0.0] * 3)[:3])
    sR = (C.c_double * 3)(*arr3(list(sigmaR), 0.02))
    al = (C.c_int * 3)(*arr3(list(algorithm), 0))
    pb = (C.c_int * 3)(*arr3(list(pbficnum), 0))
    pl = (C.c_int * 3)(*[int(bool(x)) for x in planes])
    oS, oP, oA, oB, oR, oT, oM = (C.c_double * 3)(), (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)()
    rc = lib().vszo_bilateral_params(sS, len(sigmaS), sR, al, pb, int(yuv), ssw, ssh, pl, oS, oP, oA, oB, oR, oT, oM)
    if rc != 0:
        raise ValueError(f"bilateral params rejected ({rc})")
    return {"sigmaS": list(oS), "sigmaR": list(sR), "planes": [bool(x) for x in oP], "algorithm": list(oA), "PBFICnum": list(oB),
            "radius": list(oR), "step": list(oT), "samples": list(oM)}


def bilateral_luts(sigmaS: float, sigmaR: float, radius: int, hist_len: int):
    gs = np.empty((radius + 1) ** 2, np.float32)
    gr = np.empty(hist_len, np.float32)
    lib().vszo_bilateral_gs_lut(gs.ctypes.data, radius + 1, sigmaS)
    lib().vszo_bilateral_gr_lut(gr.ctypes.data, hist_len, float(hist_len - 1), sigmaR)
    return gs, gr


def bilateral_plane(src: np.ndarray, sigmaS: float, sigmaR: float, algorithm: int, radius: int, step: int, pbficnum: int = 0,
                    ref: np.ndarray | None = None, bits=None) -> np.ndarray:
    h, w = src.shape
    hist_len = (1 << _bits_of(src, bits)) if src.dtype.kind == "u" else 65536
    gs, gr = bilateral_luts(sigmaS, sigmaR, radius if algorithm == 2 else 0, hist_len)
    dst = np.empty((h, w), src.dtype)
    r = src if ref is None else ref
    sp, ss = _plane(src)
    rp, rs = _plane(r)
    dp, ds = _plane(dst)
    rc = lib().vszo_bilateral_plane(dt_of(src), sp, rp, dp, ss, rs, ds, w, h, algorithm, radius, step, gs.ctypes.data, gr.ctypes.data,
                                    sigmaS, pbficnum, float(hist_len - 1))
    assert rc == 0
    return dst


def _ptr3(planes):
    return (C.c_void_p * 3)(*[p.ctypes.data for p in planes])


def ssimulacra2(ref, dis, want_parts: bool = False):
    """ref / dis: 3 contiguous f32 planes each (linear RGB)."""
    ref = [np.ascontiguousarray(p, np.float32) for p in ref]
    dis = [np.ascontiguousarray(p, np.float32) for p in dis]
    h, w = ref[0].shape
    a = np.zeros((6, 6)); e = np.zeros((6, 12))
    s = lib().vszo_ssimulacra2(_ptr3(ref), _ptr3(dis), w, w, h, a.ctypes.data_as(C.POINTER(C.c_double)), e.ctypes.data_as(C.POINTER(C.c_double)))
    return (s, a, e) if want_parts else s


def ssim_set_vec(v: int) -> int:
    """Test knob: the reference build's vec_size (8: x86_64_v3 / haswell, 16: znver4). Returns the previous value."""
    return lib().vszo_ssim_set_vec(v)


def ssim_to_xyb(rgb):
    rgb = [np.ascontiguousarray(p, np.float32) for p in rgb]
    h, w = rgb[0].shape
    out = [np.empty((h, w), np.float32) for _ in range(3)]
    lib().vszo_ssim_to_xyb(_ptr3(rgb), _ptr3(out), w, h)
    return out


def ssim_blur(p):
    p = np.ascontiguousarray(p, np.float32)
    out = np.empty_like(p)
    lib().vszo_ssim_blur(p.ctypes.data, out.ctypes.data, p.shape[1], p.shape[0])
    return out


def ssim_downscale(p):
    p = np.ascontiguousarray(p, np.float32)
    out = np.empty(((p.shape[0] + 1) // 2, (p.shape[1] + 1) // 2), np.float32)
    lib().vszo_ssim_downscale(p.ctypes.data, out.ctypes.data, p.shape[1], p.shape[0])
    return out


def xpsnr_wsse(org, rec, prv1=None, prv2=None, depth=8, frame_rate=24, temporal=True):
    """org / rec: lists of 1 or 3 planes (u8 or u16); prv1/prv2: previous reference luma or None."""
    n = len(org)
    o = [np.ascontiguousarray(p) for p in org] + [None] * (3 - n)
    r = [np.ascontiguousarray(p) for p in rec] + [None] * (3 - n)
    vp3 = lambda l: (C.c_void_p * 3)(*[(p.ctypes.data if p is not None else None) for p in l])
    w = (C.c_int * 3)(*[(p.shape[1] if p is not None else 0) for p in o])
    hh = (C.c_int * 3)(*[(p.shape[0] if p is not None else 0) for p in o])
    st = (C.c_ssize_t * 3)(*[(p.strides[0] // p.itemsize if p is not None else 0) for p in o])
    out = (C.c_uint64 * 3)()
    p1 = np.ascontiguousarray(prv1) if prv1 is not None else None
    p2 = np.ascontiguousarray(prv2) if prv2 is not None else None
    rc = lib().vszo_xpsnr_wsse(o[0].itemsize, vp3(o), vp3(r), p1.ctypes.data if p1 is not None else None, p2.ctypes.data if p2 is not None else None,
                               out, w, hh, st, depth, n, frame_rate, int(temporal))
    assert rc == 0
    return [int(out[i]) for i in range(n)]


def limiter(src: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """vszip.Limiter on one plane: min(max(lo, x), hi) in the sample type."""
    src = np.ascontiguousarray(src)
    dst = np.empty_like(src)
    sp, ss = _plane(src)
    dp, ds = _plane(dst)
    l = lib()
    l.vszo_limiter.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, C.c_double, C.c_double]
    assert l.vszo_limiter(dt_of(src), sp, dp, ss, ds, src.shape[1], src.shape[0], float(lo), float(hi)) == 0
    return dst


def limiter_default_range(is_float: bool, bits: int, yuv: bool, tv_range: bool):
    """-> (lo[3], hi[3]) of vszip.Limiter without min/max arrays."""
    lo, hi = (C.c_double * 3)(), (C.c_double * 3)()
    l = lib()
    l.vszo_limiter_default_range.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    l.vszo_limiter_default_range.restype = None
    l.vszo_limiter_default_range(int(is_float), bits, int(yuv), int(tv_range), lo, hi)
    return list(lo), list(hi)


def limit_filter(flt: np.ndarray, src: np.ndarray, ref: np.ndarray | None, dark_thr: float, bright_thr: float, elast: float) -> np.ndarray:
    """vszip.LimitFilter on one plane; thresholds already on the clip's scale (scale_value_from_8bit)."""
    flt, src = np.ascontiguousarray(flt), np.ascontiguousarray(src)
    ref = np.ascontiguousarray(ref) if ref is not None else None
    dst = np.empty_like(flt)
    fp, fs = _plane(flt)
    sp, ss = _plane(src)
    rp, rs = _plane(ref) if ref is not None else (None, 0)
    dp, ds = _plane(dst)
    l = lib()
    l.vszo_limit_filter.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_ssize_t] * 4 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    assert l.vszo_limit_filter(dt_of(flt), fp, sp, rp, dp, fs, ss, rs, ds, flt.shape[1], flt.shape[0], dark_thr, bright_thr, elast) == 0
    return dst


def scale_value_from_8bit(value: float, is_float: bool, bits: int, limited: bool) -> float:
    """hz.scaleValue with its default options: an 8-bit-scale threshold carried to the clip's format."""
    l = lib()
    l.vszo_scale_value_from_8bit.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int]
    l.vszo_scale_value_from_8bit.restype = C.c_float
    return float(l.vszo_scale_value_from_8bit(value, int(is_float), bits, int(limited)))


def adaptive_binarize(src: np.ndarray, src2: np.ndarray, c: int = 3) -> np.ndarray:
    """vszip.AdaptiveBinarize on one u8 plane."""
    src, src2 = np.ascontiguousarray(src, np.uint8), np.ascontiguousarray(src2, np.uint8)
    dst = np.empty_like(src)
    l = lib()
    l.vszo_adaptive_binarize.argtypes = [C.c_void_p] * 3 + [C.c_ssize_t] * 3 + [C.c_int] * 3
    assert l.vszo_adaptive_binarize(src.ctypes.data, src2.ctypes.data, dst.ctypes.data, src.strides[0], src2.strides[0], dst.strides[0], src.shape[1], src.shape[0], int(c)) == 0
    return dst


def xpsnr_frame(wsse: int, w: int, h: int, depth: int) -> float:
    return lib().vszo_xpsnr_frame(wsse, w, h, depth)


def eedi3(src: np.ndarray, field: int, dh=False, alpha=0.2, beta=0.25, gamma=20.0, nrad=2, mdis=20, hp=False, vcheck=2,
          vthresh0=32.0, vthresh1=64.0, vthresh2=4.0, sclip=None, mclip=None, horizontal=False) -> np.ndarray:
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape
    dst = np.zeros((h * 2, w) if (dh and not horizontal) else ((h, w * 2) if dh else (h, w)), np.float32)
    sc = np.ascontiguousarray(sclip, np.float32) if sclip is not None else None
    mc = np.ascontiguousarray(mclip, np.uint8) if mclip is not None else None
    rc = lib().vszo_eedi3_plane(src.ctypes.data, dst.ctypes.data, sc.ctypes.data if sc is not None else None, mc.ctypes.data if mc is not None else None,
                                src.shape[1], dst.shape[1], dst.shape[1], w, w, h, field, int(dh), alpha, beta, gamma, nrad, mdis, int(hp), vcheck,
                                vthresh0, vthresh1, vthresh2, int(horizontal))
    assert rc == 0, rc
    return dst
