"""ctypes binding of include/vszip_hip.h (libvszip_hip.so) + small numpy helpers.

This is plumbing for tests, bench.py and the Python host mirror; the product is
the shared library. Loading fails loudly when the HIP library is missing — there
is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "libvszip_hip.so"

U8, U16, F16, F32 = 0, 1, 2, 3
OK, ERR_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_NOMEM = 0, -1, -2, -3, -4
_NP2DT = {np.dtype(np.uint8): U8, np.dtype(np.uint16): U16, np.dtype(np.float16): F16, np.dtype(np.float32): F32, np.dtype(np.uint32): 4}
_DT2NP = {v: k for k, v in _NP2DT.items()}


VARIANTS_ABSENT: dict = {}  # development-variant switches a test asked for that this build does not contain -> how often (Device.variant)


class VszipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


class Plane(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("dst", C.c_void_p), ("ref", C.c_void_p),
        ("src_stride", C.c_ssize_t), ("dst_stride", C.c_ssize_t), ("ref_stride", C.c_ssize_t),
        ("w", C.c_int32), ("h", C.c_int32),
    ]


class BilateralCfg(C.Structure):
    _fields_ = [
        ("sigmaS", C.c_double), ("sigmaR", C.c_double), ("process", C.c_int32), ("algorithm", C.c_int32),
        ("pbficnum", C.c_int32), ("radius", C.c_int32), ("step", C.c_int32), ("samples", C.c_int32),
        ("gs_lut", C.c_void_p), ("gr_lut", C.c_void_p),
    ]


class ChainStage(C.Structure):
    """vszip_chain_stage"""
    _fields_ = [("kind", C.c_int32), ("process", C.c_int32 * 3), ("hradius", C.c_int32), ("hpasses", C.c_int32), ("vradius", C.c_int32), ("vpasses", C.c_int32),
                ("bilateral", C.POINTER(BilateralCfg) * 3), ("peak", C.c_float), ("lo", C.c_double * 3), ("hi", C.c_double * 3)]


STAGE_BOXBLUR, STAGE_BILATERAL, STAGE_LIMITER = 0, 1, 2


class SsimSource(C.Structure):
    """vszip_ssim_source: family (0 RGB, 1 Gray, 2 YUV), dtype, bits, limited, linearize; YUV: ssw, ssh, matrix, chroma_loc, chroma_stride."""
    _fields_ = [("family", C.c_int32), ("dtype", C.c_int32), ("bits", C.c_int32), ("limited", C.c_int32), ("linearize", C.c_int32),
                ("ssw", C.c_int32), ("ssh", C.c_int32), ("matrix", C.c_int32), ("chroma_loc", C.c_int32), ("chroma_stride", C.c_ssize_t)]


CF_RGB, CF_GRAY, CF_YUV = 0, 1, 2


class Eedi3Params(C.Structure):
    _fields_ = [("dh", C.c_int32), ("alpha", C.c_float), ("beta", C.c_float), ("gamma", C.c_float), ("nrad", C.c_int32), ("mdis", C.c_int32),
                ("hp", C.c_int32), ("vcheck", C.c_int32), ("vthresh0", C.c_float), ("vthresh1", C.c_float), ("vthresh2", C.c_float)]


# every symbol include/vszip_hip.h declares: name -> (restype, argtypes)
_vp, _i, _sz, _pd = C.c_void_p, C.c_int, C.c_size_t, C.c_ssize_t
_PP = C.POINTER(Plane)
SYMBOLS = {
    "vszip_abi_version": (_i, []),
    "vszip_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "vszip_ctx_destroy": (None, [_vp]),
    "vszip_ctx_set_stream": (_i, [_vp, _vp]),
    "vszip_ctx_stream": (_vp, [_vp]),
    "vszip_ctx_sync": (_i, [_vp]),
    "vszip_ctx_set_staging": (_i, [_vp, _i]),
    "vszip_ctx_abort": (_i, [_vp]),
    "vszip_ctx_set_option": (_i, [_vp, C.c_char_p, _i]),
    "vszip_ctx_get_option": (_i, [_vp, C.c_char_p, C.POINTER(_i)]),
    "vszip_dev_arena_info": (_i, [_vp, _vp, C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vszip_dev_probe_region": (_i, [_vp, _vp, _sz, _vp, C.POINTER(C.c_double)]),
    "vszip_last_error": (C.c_char_p, [_vp]),
    "vszip_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "vszip_dev_free": (_i, [_vp, _vp]),
    "vszip_dev_memset": (_i, [_vp, _vp, _i, _sz]),
    "vszip_host_alloc_pinned": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "vszip_host_free_pinned": (_i, [_vp, _vp]),
    "vszip_copy_h2d_2d": (_i, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    "vszip_copy_d2h_2d": (_i, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    "vszip_copy_d2d_2d": (_i, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    "vszip_timer_start": (_i, [_vp]),
    "vszip_timer_stop_ms": (_i, [_vp, C.POINTER(C.c_float)]),
    "vszip_probe_enable": (_i, [_vp, _i]),
    "vszip_probe_read": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_i)]),
    "vszip_probe_read_each": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(C.c_float), _i]),
    "vszip_boxblur": (_i, [_vp, _i, _PP, _i, _i, _i, _i, _i]),
    "vszip_bilateral_derive": (_i, [C.POINTER(C.c_double), _i, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), _i, _i, _i,
                                    C.POINTER(C.c_int), C.POINTER(BilateralCfg)]),
    "vszip_bilateral_luts": (_i, [_vp, C.POINTER(BilateralCfg), _i]),
    "vszip_bilateral": (_i, [_vp, _i, _PP, C.POINTER(C.POINTER(BilateralCfg)), _i, C.c_float]),
    "vszip_ssimulacra2": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), _pd, _i, _i, _i, C.POINTER(C.c_double)]),
    "vszip_chain_run": (_i, [_vp, _i, C.POINTER(ChainStage), _i, _PP, C.POINTER(C.c_int), _i]),
    "vszip_resample_table": (_i, [_i, _i, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "vszip_ssimulacra2_src": (_i, [_vp, C.POINTER(SsimSource), C.POINTER(_vp), C.POINTER(_vp), _pd, _i, _i, _i, C.POINTER(C.c_double)]),
    "vszip_to_rgbs_linear": (_i, [_vp, C.POINTER(SsimSource), C.POINTER(_vp), _pd, C.POINTER(_vp), _pd, _i, _i]),
    "vszip_eedi3": (_i, [_vp, _PP, C.POINTER(_vp), C.POINTER(_pd), _i, _i, _i, C.POINTER(Eedi3Params)]),
    "vszip_eedi3_mclip": (_i, [_vp, _PP, C.POINTER(_vp), C.POINTER(_pd), C.POINTER(_vp), C.POINTER(_pd), _i, _i, _i, C.POINTER(Eedi3Params)]),
    "vszip_xpsnr_wsse": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_pd), _i, _i, C.c_uint, _i,
                              C.POINTER(C.c_uint64)]),
    "vszip_xpsnr_wsse_batch": (_i, [_vp, _i, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), C.POINTER(_pd), _i, _i,
                                    C.c_uint, _i, C.POINTER(C.c_uint64)]),
    "vszip_xpsnr_value": (C.c_double, [C.c_uint64, C.c_uint64, C.c_uint64, _i]),
    "vszip_xpsnr_average": (C.c_double, [C.c_double, C.c_double, C.c_uint64, C.c_uint64, _i, C.c_uint64]),
    "vszip_limiter": (_i, [_vp, _i, _PP, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vszip_limit_filter": (_i, [_vp, _i, _PP, C.POINTER(_vp), C.POINTER(_pd), _i, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "vszip_adaptive_binarize": (_i, [_vp, _PP, _i, _i]),
    "vszip_plane_average": (_i, [_vp, _i, _PP, _i, C.POINTER(C.c_int32), _i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vszip_plane_minmax": (_i, [_vp, _i, _PP, _i, C.c_float, C.c_float, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vszip_plane_average_async": (_i, [_vp, _i, _PP, _i, C.POINTER(C.c_int32), _i, _i, _vp]),
    "vszip_plane_minmax_async": (_i, [_vp, _i, _PP, _i, C.c_float, C.c_float, _i, _vp]),
}

_lib = None


def resample_table(src_dim: int, dst_dim: int, shift: float = 0.0):
    """vszip_resample_table (device-free): zimg's Catmull-Rom upscale table of one axis -> (left[dst_dim] int32, coef[dst_dim, 4] f32)."""
    left = np.empty(dst_dim, np.int32)
    coef = np.empty((dst_dim, 4), np.float32)
    rc = load().vszip_resample_table(src_dim, dst_dim, float(shift), left.ctypes.data_as(C.POINTER(C.c_int32)), coef.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != 0:
        raise ValueError(f"vszip_resample_table({src_dim}, {dst_dim}, {shift}) -> {rc}")
    return left, coef


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so.7 (same soname as the system one). Two HIP
    runtimes in one process do not work (the second finds no device), and whichever is loaded
    first wins the soname: when torch is installed but not imported yet, load ITS runtime first, so
    that libvszip_hip.so and a later `import torch` share one — the same state as importing torch
    before this module. Without torch installed the system runtime is used."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")  # does not import torch
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.is_file():
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """dlopen libvszip_hip.so and declare every exported entry point."""
    global _lib
    if _lib is None:
        if not LIB_PATH.is_file():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python vapoursynth-zip_amd/build.py` "
                "(there is no CPU fallback)"
            )
        _share_torch_hip_runtime()
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class DevPlane:
    """A 2-D plane in device memory with a row pitch (in elements)."""

    __slots__ = ("dev", "ptr", "w", "h", "stride", "dtype", "_own")

    def __init__(self, dev: "Device", ptr: int, w: int, h: int, stride: int, dtype, own: bool = True):
        self.dev, self.ptr, self.w, self.h, self.stride, self.dtype, self._own = dev, ptr, w, h, stride, np.dtype(dtype), own

    @property
    def nbytes(self) -> int:
        return self.stride * self.h * self.dtype.itemsize

    def free(self):
        if self._own and self.ptr:
            self.dev.lib.vszip_dev_free(self.dev.ctx, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Device:
    """One context (GPU + stream)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        ctx = C.c_void_p()
        rc = self.lib.vszip_ctx_create(device, C.byref(ctx))
        if rc != OK:
            raise VszipError(rc, f"vszip_ctx_create(device={device}) failed with {rc} (no MI355X visible?)")
        self.ctx = ctx
        self.device = device

    def close(self):
        if self.ctx:
            for hp in getattr(self, "_pinned", []):
                self.lib.vszip_host_free_pinned(self.ctx, C.c_void_p(hp))
            self._pinned = []
            self.lib.vszip_ctx_destroy(self.ctx)
            self.ctx = None

    def check(self, rc: int):
        if rc != OK:
            raise VszipError(rc, self.lib.vszip_last_error(self.ctx).decode())

    def sync(self):
        self.check(self.lib.vszip_ctx_sync(self.ctx))

    def set_staging(self, mode: int):
        """0: host copies go straight from/to the caller's memory; 1: through the context's pinned arena."""
        self.check(self.lib.vszip_ctx_set_staging(self.ctx, mode))

    def set_option(self, name: str, value: int = 1):
        """vszip_ctx_set_option: one switch of csrc/options.inc on this context, by its environment name.
        VszipError with code VSZIP_ERR_UNSUPPORTED for a development variant a default build does not contain."""
        saved = self.__dict__.setdefault("_opt_saved", {})
        if name not in saved:
            try:
                saved[name] = self.get_option(name)
            except VszipError:
                pass
        self.check(self.lib.vszip_ctx_set_option(self.ctx, name.encode(), int(value)))

    def restore_options(self, keep=()):
        """every option set through set_option back to its value before the first set (tests)"""
        for k, v in list(self.__dict__.get("_opt_saved", {}).items()):
            if k not in keep:
                self.lib.vszip_ctx_set_option(self.ctx, k.encode(), int(v))
                del self._opt_saved[k]

    def get_option(self, name: str) -> int:
        v = C.c_int()
        self.check(self.lib.vszip_ctx_get_option(self.ctx, name.encode(), C.byref(v)))
        return v.value

    def variant(self, **kw):
        """like options(), for switches that may be development variants: yields False (and sets nothing) when this build does not
        contain them (VSZIP_ERR_UNSUPPORTED), so a path-agreement test compares the alternative only where it exists"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            try:
                for k in kw:
                    self.get_option(k)
            except VszipError as e:
                if e.code != -3:
                    raise
                for k in kw:  # counted: tests/conftest.py reports how many cross-checks a default build left out
                    VARIANTS_ABSENT[k] = VARIANTS_ABSENT.get(k, 0) + 1
                yield False
                return
            with self.options(**kw):
                yield True

        return scope()

    def options(self, **kw):
        """`with dev.options(VSZIP_RT_NO_ICHAIN=1): ...` — set, run, restore (the path-agreement tests)"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = {k: self.get_option(k) for k in kw}
            try:
                for k, v in kw.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)

        return scope()

    def arena_info(self, ptr: int) -> dict:
        """vszip_dev_arena_info: what the placement search did for a vszip_dev_alloc pointer (candidates = 0: a plain allocation)"""
        nc, rate, ms = C.c_int(), C.c_double(), C.c_double()
        self.check(self.lib.vszip_dev_arena_info(self.ctx, C.c_void_p(ptr), C.byref(nc), C.byref(rate), C.byref(ms)))
        return {"candidates": nc.value, "probe_bytes_per_second": rate.value, "search_ms": ms.value}

    def probe_region(self, ptr: int, nbytes: int, src: int = 0) -> float:
        bps = C.c_double()
        self.check(self.lib.vszip_dev_probe_region(self.ctx, C.c_void_p(ptr), nbytes, C.c_void_p(src) if src else None, C.byref(bps)))
        return bps.value

    def set_stream(self, hip_stream: int):
        """Enqueue on an externally owned hipStream_t (e.g. torch.cuda.Stream().cuda_stream)."""
        self.check(self.lib.vszip_ctx_set_stream(self.ctx, C.c_void_p(hip_stream)))

    def wrap(self, ptr: int, h: int, w: int, stride: int, dtype) -> "DevPlane":
        """A DevPlane view of device memory this context does not own (e.g. a torch tensor's data_ptr)."""
        return DevPlane(self, ptr, w, h, stride, np.dtype(dtype), own=False)

    # -- memory ---------------------------------------------------------------
    def empty(self, h: int, w: int, dtype, align_elems: int = 32) -> DevPlane:
        dtype = np.dtype(dtype)
        stride = -(-w // align_elems) * align_elems
        p = C.c_void_p()
        self.check(self.lib.vszip_dev_alloc(self.ctx, stride * h * dtype.itemsize + 256, C.byref(p)))
        return DevPlane(self, p.value, w, h, stride, dtype)

    def upload(self, a: np.ndarray, align_elems: int = 32) -> DevPlane:
        assert a.ndim == 2 and a.strides[1] == a.itemsize
        d = self.empty(a.shape[0], a.shape[1], a.dtype, align_elems)
        self.check(self.lib.vszip_copy_h2d_2d(self.ctx, d.ptr, d.stride * a.itemsize, a.ctypes.data, a.strides[0], a.shape[1] * a.itemsize, a.shape[0]))
        self.sync()
        return d

    def pinned_array(self, shape, dtype) -> np.ndarray:
        """Host array in page-locked memory (async copies; what a plugin keeps per worker)."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self.check(self.lib.vszip_host_alloc_pinned(self.ctx, n, C.byref(p)))
        buf = (C.c_char * n).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", []) + [p.value]
        return a

    def copy_in(self, d: DevPlane, a: np.ndarray):
        """Async H2D of a host array into an existing device plane (no synchronisation)."""
        self.check(self.lib.vszip_copy_h2d_2d(self.ctx, d.ptr, d.stride * a.itemsize, a.ctypes.data, a.strides[0], a.shape[1] * a.itemsize, a.shape[0]))

    def copy_out(self, a: np.ndarray, d: DevPlane):
        """Async D2H of a device plane into a host array (no synchronisation)."""
        self.check(self.lib.vszip_copy_d2h_2d(self.ctx, a.ctypes.data, a.strides[0], d.ptr, d.stride * d.dtype.itemsize, d.w * d.dtype.itemsize, d.h))

    def download(self, d: DevPlane) -> np.ndarray:
        out = np.empty((d.h, d.w), dtype=d.dtype)
        self.check(self.lib.vszip_copy_d2h_2d(self.ctx, out.ctypes.data, out.strides[0], d.ptr, d.stride * d.dtype.itemsize, d.w * d.dtype.itemsize, d.h))
        self.sync()
        return out

    # -- timing ---------------------------------------------------------------
    def timer_start(self):
        self.check(self.lib.vszip_timer_start(self.ctx))

    def probe_enable(self, on: bool):
        self.check(self.lib.vszip_probe_enable(self.ctx, int(on)))

    def probe_read(self):
        """(summed dominant-kernel ms, launches) since the probe was enabled / last read."""
        ms, n = C.c_double(), C.c_int()
        self.check(self.lib.vszip_probe_read(self.ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def probe_read_each(self, cap: int = 65536):
        """(summed ms, launches, [per-launch ms]) — the per-launch durations in launch order."""
        ms, n = C.c_double(), C.c_int()
        each = (C.c_float * cap)()
        self.check(self.lib.vszip_probe_read_each(self.ctx, C.byref(ms), C.byref(n), each, cap))
        return ms.value, n.value, [each[i] for i in range(min(n.value, cap))]

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        self.check(self.lib.vszip_timer_stop_ms(self.ctx, C.byref(ms)))
        return ms.value

    # -- filters --------------------------------------------------------------
    @staticmethod
    def plane_table(srcs, dsts=None, refs=None):
        n = len(srcs)
        arr = (Plane * n)()
        for i, s in enumerate(srcs):
            arr[i].src, arr[i].src_stride, arr[i].w, arr[i].h = s.ptr, s.stride, s.w, s.h
            if dsts is not None:
                arr[i].dst, arr[i].dst_stride = dsts[i].ptr, dsts[i].stride
            if refs is not None and refs[i] is not None:
                arr[i].ref, arr[i].ref_stride = refs[i].ptr, refs[i].stride
        return arr

    def boxblur_table(self, dtype, table, hradius=1, hpasses=1, vradius=1, vpasses=1):
        self.check(self.lib.vszip_boxblur(self.ctx, _NP2DT[np.dtype(dtype)], table, len(table), hradius, hpasses, vradius, vpasses))

    def boxblur(self, srcs, dsts, hradius=1, hpasses=1, vradius=1, vpasses=1):
        self.boxblur_table(srcs[0].dtype, self.plane_table(srcs, dsts), hradius, hpasses, vradius, vpasses)

    def plane_average(self, srcs, exclude=(), refs=None, bits=None):
        """-> (avg[], diff[] or None) for a list of planes of one sample type."""
        n = len(srcs)
        table = self.plane_table(srcs, None, refs)
        ex = (C.c_int32 * max(1, len(exclude)))(*exclude)
        avg, diff = (C.c_double * n)(), (C.c_double * n)()
        b = bits if bits is not None else 8 * srcs[0].dtype.itemsize
        self.check(self.lib.vszip_plane_average(self.ctx, _NP2DT[srcs[0].dtype], table, n, ex, len(exclude), b, avg, diff))
        return list(avg), (list(diff) if refs is not None else None)

    def plane_average_async(self, srcs, results, exclude=(), refs=None, bits=None):
        """vszip_plane_average_async: `results` is a pinned_array((len(srcs), 4), float64); valid after the next sync()"""
        n = len(srcs)
        table = self.plane_table(srcs, None, refs)
        ex = (C.c_int32 * max(1, len(exclude)))(*exclude)
        b = bits if bits is not None else 8 * srcs[0].dtype.itemsize
        self.check(self.lib.vszip_plane_average_async(self.ctx, _NP2DT[srcs[0].dtype], table, n, ex, len(exclude), b, results.ctypes.data))

    def plane_minmax_async(self, srcs, results, minthr=0.0, maxthr=0.0, refs=None, bits=None):
        """vszip_plane_minmax_async: results[i] = (min, max, diff, -), pinned; valid after the next sync()"""
        n = len(srcs)
        table = self.plane_table(srcs, None, refs)
        b = bits if bits is not None else 8 * srcs[0].dtype.itemsize
        self.check(self.lib.vszip_plane_minmax_async(self.ctx, _NP2DT[srcs[0].dtype], table, n, minthr, maxthr, b, results.ctypes.data))

    def limiter(self, srcs, dsts, lo, hi):
        """dsts[i] = min(max(lo[i], srcs[i]), hi[i]) (vszip.Limiter with the bounds already resolved)."""
        n = len(srcs)
        table = self.plane_table(srcs, dsts)
        self.check(self.lib.vszip_limiter(self.ctx, _NP2DT[srcs[0].dtype], table, n, (C.c_double * n)(*[float(v) for v in lo]), (C.c_double * n)(*[float(v) for v in hi])))

    def prepared_limiter(self, srcs, dsts, lo, hi):
        """-> a callable queueing vszip_limiter on argument blocks built once."""
        n = len(srcs)
        table = self.plane_table(srcs, dsts)
        los, his = (C.c_double * n)(*[float(v) for v in lo]), (C.c_double * n)(*[float(v) for v in hi])
        dt, fn, ctx, check = _NP2DT[srcs[0].dtype], self.lib.vszip_limiter, self.ctx, self.check
        return lambda: check(fn(ctx, dt, table, n, los, his))

    def prepared_limit_filter(self, flts, srcs, dsts, dark_thr, bright_thr, elast):
        """-> a callable queueing vszip_limit_filter (no third clip) on argument blocks built once."""
        n = len(flts)
        table = self.plane_table(flts, dsts, srcs)
        fa = lambda v: (C.c_float * n)(*[float(x) for x in v])
        d, b, e = fa(dark_thr), fa(bright_thr), fa(elast)
        dt, fn, ctx, check = _NP2DT[flts[0].dtype], self.lib.vszip_limit_filter, self.ctx, self.check
        return lambda: check(fn(ctx, dt, table, None, None, n, d, b, e))

    def limit_filter(self, flts, srcs, dsts, dark_thr, bright_thr, elast, refs=None):
        """vszip.LimitFilter per plane; thresholds already on the clip's scale. refs: optional third clip's planes."""
        n = len(flts)
        table = self.plane_table(flts, dsts, srcs)
        fa = lambda v: (C.c_float * n)(*[float(x) for x in v])
        rp = (C.c_void_p * n)(*[p.ptr for p in refs]) if refs is not None else None
        rs = (C.c_ssize_t * n)(*[p.stride for p in refs]) if refs is not None else None
        self.check(self.lib.vszip_limit_filter(self.ctx, _NP2DT[flts[0].dtype], table, rp, rs, n, fa(dark_thr), fa(bright_thr), fa(elast)))

    def adaptive_binarize(self, clips, clips2, dsts, c=3):
        """dsts[i] = 255 where clips2[i] - clips[i] >= c else 0 (u8 planes)."""
        self.check(self.lib.vszip_adaptive_binarize(self.ctx, self.plane_table(clips, dsts, clips2), len(clips), int(c)))

    def prepared_plane_average(self, srcs, exclude=(), refs=None, bits=None):
        """-> a callable running vszip_plane_average on argument blocks built once (a per-frame caller in C pays no Python marshalling either);
        it returns the ctypes result arrays (avg, diff)."""
        n = len(srcs)
        table = self.plane_table(srcs, None, refs)
        ex = (C.c_int32 * max(1, len(exclude)))(*exclude)
        avg, diff = (C.c_double * n)(), (C.c_double * n)()
        b = bits if bits is not None else 8 * srcs[0].dtype.itemsize
        dt, nex, fn, ctx, check = _NP2DT[srcs[0].dtype], len(exclude), self.lib.vszip_plane_average, self.ctx, self.check

        def run():
            check(fn(ctx, dt, table, n, ex, nex, b, avg, diff))
            return avg, diff
        return run

    def prepared_plane_minmax(self, srcs, minthr=0.0, maxthr=0.0, refs=None, bits=None):
        """-> a callable running vszip_plane_minmax on argument blocks built once; it returns the ctypes result arrays (min, max, diff)."""
        n = len(srcs)
        table = self.plane_table(srcs, None, refs)
        mn, mx, df = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        b = bits if bits is not None else 8 * srcs[0].dtype.itemsize
        dt, fn, ctx, check = _NP2DT[srcs[0].dtype], self.lib.vszip_plane_minmax, self.ctx, self.check

        def run():
            check(fn(ctx, dt, table, n, minthr, maxthr, b, mn, mx, df))
            return mn, mx, df
        return run

    def plane_minmax(self, srcs, minthr=0.0, maxthr=0.0, refs=None, bits=None):
        n = len(srcs)
        table = self.plane_table(srcs, None, refs)
        mn, mx, df = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        b = bits if bits is not None else 8 * srcs[0].dtype.itemsize
        self.check(self.lib.vszip_plane_minmax(self.ctx, _NP2DT[srcs[0].dtype], table, n, minthr, maxthr, b, mn, mx, df))
        return list(mn), list(mx), (list(df) if refs is not None else None)

    @staticmethod
    def _arr3(vals, default):
        out = []
        for i in range(3):
            out.append(vals[i] if i < len(vals) else (default if i == 0 else out[i - 1]))
        return out

    def bilateral_cfg(self, sigmaS=(), sigmaR=(), algorithm=(), pbficnum=(), planes=(True, True, True), yuv=False, ssw=0, ssh=0, hist_len=65536):
        """bilateralCreate: derive the 3 per-plane configs and build their LUTs on the device."""
        sS = (C.c_double * 3)(*(list(sigmaS) + [0.0] * 3)[:3])
        sR = (C.c_double * 3)(*self._arr3(list(sigmaR), 0.02))
        al = (C.c_int * 3)(*self._arr3(list(algorithm), 0))
        pb = (C.c_int * 3)(*self._arr3(list(pbficnum), 0))
        pl = (C.c_int * 3)(*[int(bool(x)) for x in planes])
        cfg = (BilateralCfg * 3)()
        rc = self.lib.vszip_bilateral_derive(sS, len(sigmaS), sR, al, pb, int(yuv), ssw, ssh, pl, cfg)
        if rc != OK:
            raise VszipError(rc, "Bilateral: invalid parameters")
        for i in range(3):
            self.check(self.lib.vszip_bilateral_luts(self.ctx, C.byref(cfg[i]), hist_len))
        return cfg

    def bilateral_free(self, cfg):
        for i in range(3):
            for f in ("gs_lut", "gr_lut"):
                p = getattr(cfg[i], f)
                if p:
                    self.lib.vszip_dev_free(self.ctx, p)
                    setattr(cfg[i], f, None)

    def bilateral(self, srcs, dsts, cfg, plane_index, refs=None, peak=None):
        """plane_index[i] = which of the 3 configs plane i uses."""
        n = len(srcs)
        table = self.plane_table(srcs, dsts, refs)
        ptrs = (C.POINTER(BilateralCfg) * n)(*[C.pointer(cfg[k]) for k in plane_index])
        dt = srcs[0].dtype
        if peak is None:
            peak = float((1 << (8 * dt.itemsize)) - 1) if dt.kind == "u" else 65535.0
        self.check(self.lib.vszip_bilateral(self.ctx, _NP2DT[dt], table, ptrs, n, peak))

    def ssimulacra2(self, ref_planes, dis_planes):
        """ref_planes / dis_planes: flat lists of 3*npairs f32 DevPlanes (same geometry). -> scores[npairs]"""
        n = len(ref_planes) // 3
        r = (C.c_void_p * (3 * n))(*[p.ptr for p in ref_planes])
        d = (C.c_void_p * (3 * n))(*[p.ptr for p in dis_planes])
        out = (C.c_double * n)()
        p0 = ref_planes[0]
        self.check(self.lib.vszip_ssimulacra2(self.ctx, r, d, p0.stride, p0.w, p0.h, n, out))
        return list(out)

    def chain_run(self, stages, srcs, dsts, plane_slot):
        """stages: list of dicts — {"boxblur": (hr, hp, vr, vp)} | {"bilateral": cfg, "peak": p} | {"limiter": (lo3, hi3)}, each with
        an optional "planes": (bool, bool, bool). One call: every stage on the resident planes, intermediates owned by the context."""
        arr = (ChainStage * len(stages))()
        for st, d in zip(arr, stages):
            pr = d.get("planes", (True, True, True))
            for k in range(3):
                st.process[k] = int(bool(pr[k]))
            if "boxblur" in d:
                st.kind = STAGE_BOXBLUR
                st.hradius, st.hpasses, st.vradius, st.vpasses = d["boxblur"]
            elif "bilateral" in d:
                st.kind = STAGE_BILATERAL
                for k in range(3):
                    st.bilateral[k] = C.pointer(d["bilateral"][k])
                    st.process[k] = int(bool(pr[k]) and bool(d["bilateral"][k].process))
                st.peak = d["peak"]
            else:
                st.kind = STAGE_LIMITER
                for k in range(3):
                    st.lo[k], st.hi[k] = d["limiter"][0][k], d["limiter"][1][k]
        n = len(srcs)
        slots = (C.c_int * n)(*plane_slot)
        self.check(self.lib.vszip_chain_run(self.ctx, _NP2DT[srcs[0].dtype], arr, len(stages), self.plane_table(srcs, dsts), slots, n))

    @staticmethod
    def ssim_source(family: str, dtype, bits=None, linearize=True, limited=None, ssw=0, ssh=0, matrix=1, chroma_loc=0) -> SsimSource:
        """family "RGB" | "GRAY" | "YUV"; integer samples: full range for RGB, limited for Gray / YUV (zimg's defaults) unless
        given. YUV: log2 subsampling, _Matrix, _ChromaLocation (the chroma row pitch is filled in from the planes)."""
        dt = np.dtype(dtype)
        fam = {"GRAY": CF_GRAY, "YUV": CF_YUV}.get(family.upper(), CF_RGB)
        b = bits if bits is not None else (32 if dt.kind == "f" else 8 * dt.itemsize)
        lim = (fam != CF_RGB) if limited is None else bool(limited)
        return SsimSource(fam, _NP2DT[dt], b, int(lim and dt.kind != "f"), int(bool(linearize)), ssw, ssh, matrix, chroma_loc, 0)

    def ssimulacra2_src(self, fmt: SsimSource, ref_planes, dis_planes):
        """SSIMULACRA2 with the colour pre-stage on the device: ref_planes / dis_planes are flat lists of
        npairs * (3 | 1) source DevPlanes in the clip's own sample type. -> scores[npairs]"""
        per = 1 if fmt.family == CF_GRAY else 3
        n = len(ref_planes) // per
        r = (C.c_void_p * (per * n))(*[p.ptr for p in ref_planes])
        d = (C.c_void_p * (per * n))(*[p.ptr for p in dis_planes])
        out = (C.c_double * n)()
        p0 = ref_planes[0]
        if fmt.family == CF_YUV:
            fmt.chroma_stride = ref_planes[1].stride
            assert all(p.stride == (p0.stride if i % 3 == 0 else fmt.chroma_stride) for i, p in enumerate(list(ref_planes) + list(dis_planes)))
        self.check(self.lib.vszip_ssimulacra2_src(self.ctx, C.byref(fmt), r, d, p0.stride, p0.w, p0.h, n, out))
        return list(out)

    def to_rgbs_linear(self, fmt: SsimSource, planes):
        """One frame's source planes -> [R, G, B] linear-light f32 DevPlanes (hz.toRGBS + sRGBtoLinearRGB)."""
        p0 = planes[0]
        dst = [self.empty(p0.h, p0.w, np.float32) for _ in range(3)]
        per = 1 if fmt.family == CF_GRAY else 3
        s = (C.c_void_p * per)(*[p.ptr for p in planes[:per]])
        d = (C.c_void_p * 3)(*[p.ptr for p in dst])
        if fmt.family == CF_YUV:
            fmt.chroma_stride = planes[1].stride
        self.check(self.lib.vszip_to_rgbs_linear(self.ctx, C.byref(fmt), s, p0.stride, d, dst[0].stride, p0.w, p0.h))
        self.sync()
        return dst

    def eedi3(self, srcs, field, dh=False, alpha=0.2, beta=0.25, gamma=20.0, nrad=2, mdis=20, hp=False, vcheck=2,
              vthresh0=32.0, vthresh1=64.0, vthresh2=4.0, sclips=None, horizontal=False, mclips=None):
        """srcs: f32 DevPlanes; mclips: optional u8 DevPlanes with the geometry of srcs. Returns the output DevPlanes."""
        dsts = []
        for s in srcs:
            if horizontal:
                dsts.append(self.empty(s.h, s.w * 2 if dh else s.w, np.float32))
            else:
                dsts.append(self.empty(s.h * 2 if dh else s.h, s.w, np.float32))
        n = len(srcs)
        table = self.plane_table(srcs, dsts)
        prm = Eedi3Params(int(dh), alpha, beta, gamma, nrad, mdis, int(hp), vcheck, vthresh0, vthresh1, vthresh2)
        if sclips is not None:
            sp = (C.c_void_p * n)(*[(s.ptr if s is not None else None) for s in sclips])
            ss = (C.c_ssize_t * n)(*[(s.stride if s is not None else 0) for s in sclips])
        else:
            sp, ss = None, None
        if mclips is not None:
            mp = (C.c_void_p * n)(*[(m.ptr if m is not None else None) for m in mclips])
            ms = (C.c_ssize_t * n)(*[(m.stride if m is not None else 0) for m in mclips])
            self.check(self.lib.vszip_eedi3_mclip(self.ctx, table, sp, ss, mp, ms, n, field, int(horizontal), C.byref(prm)))
        else:
            self.check(self.lib.vszip_eedi3(self.ctx, table, sp, ss, n, field, int(horizontal), C.byref(prm)))
        return dsts

    def xpsnr_wsse(self, org, rec, prev1=None, prev2=None, depth=8, frame_rate=24, temporal=True):
        """org / rec: lists of 1 or 3 DevPlanes (u8/u16). -> [wsse64 per plane]"""
        n = len(org)
        vp3 = lambda l: (C.c_void_p * 3)(*([p.ptr for p in l] + [None] * (3 - n)))
        w = (C.c_int * 3)(*([p.w for p in org] + [0] * (3 - n)))
        h = (C.c_int * 3)(*([p.h for p in org] + [0] * (3 - n)))
        st = (C.c_ssize_t * 3)(*([p.stride for p in org] + [0] * (3 - n)))
        out = (C.c_uint64 * 3)()
        self.check(self.lib.vszip_xpsnr_wsse(self.ctx, org[0].dtype.itemsize, vp3(org), vp3(rec), prev1.ptr if prev1 else None, prev2.ptr if prev2 else None,
                                             w, h, st, depth, n, frame_rate, int(temporal), out))
        return [int(out[i]) for i in range(n)]

    def xpsnr_batch_call(self, orgs, recs, prev1s=None, prev2s=None, depth=8, frame_rate=24, temporal=True):
        """The argument marshalling of xpsnr_wsse_batch done once: returns run() -> [[wsse64 per plane] per
        frame] (a host that keeps its frame pointers in C arrays pays none of it per call)."""
        nf, n = len(orgs), len(orgs[0])
        flat = lambda ll: (C.c_void_p * (nf * n))(*[p.ptr for l in ll for p in l])
        prev = lambda l: (C.c_void_p * nf)(*[(p.ptr if p is not None else None) for p in l]) if l is not None else None
        o0 = orgs[0]
        w = (C.c_int * 3)(*([p.w for p in o0] + [0] * (3 - n)))
        h = (C.c_int * 3)(*([p.h for p in o0] + [0] * (3 - n)))
        st = (C.c_ssize_t * 3)(*([p.stride for p in o0] + [0] * (3 - n)))
        out = (C.c_uint64 * (3 * nf))()
        fo, fr, p1, p2 = flat(orgs), flat(recs), prev(prev1s), prev(prev2s)
        bps, keep = o0[0].dtype.itemsize, (orgs, recs, prev1s, prev2s)

        def run(_keep=keep):
            self.check(self.lib.vszip_xpsnr_wsse_batch(self.ctx, bps, nf, fo, fr, p1, p2, w, h, st, depth, n, frame_rate, int(temporal), out))
            return [[int(out[3 * f + i]) for i in range(n)] for f in range(nf)]

        return run

    def xpsnr_wsse_batch(self, orgs, recs, prev1s=None, prev2s=None, depth=8, frame_rate=24, temporal=True):
        """orgs / recs: one list of 1 or 3 DevPlanes per frame (same geometry); prev1s / prev2s: per
        frame the luma DevPlane of frames n-1 / n-2 or None. -> [[wsse64 per plane] per frame]"""
        return self.xpsnr_batch_call(orgs, recs, prev1s, prev2s, depth, frame_rate, temporal)()
