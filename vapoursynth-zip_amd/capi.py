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
_NP2DT = {np.dtype(np.uint8): U8, np.dtype(np.uint16): U16, np.dtype(np.float16): F16, np.dtype(np.float32): F32}
_DT2NP = {v: k for k, v in _NP2DT.items()}


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
    "vszip_boxblur": (_i, [_vp, _i, _PP, _i, _i, _i, _i, _i]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen libvszip_hip.so and declare every exported entry point."""
    global _lib
    if _lib is None:
        if not LIB_PATH.is_file():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python vapoursynth-zip_amd/build.py` "
                "(there is no CPU fallback)"
            )
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
            self.lib.vszip_ctx_destroy(self.ctx)
            self.ctx = None

    def check(self, rc: int):
        if rc != OK:
            raise VszipError(rc, self.lib.vszip_last_error(self.ctx).decode())

    def sync(self):
        self.check(self.lib.vszip_ctx_sync(self.ctx))

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

    def download(self, d: DevPlane) -> np.ndarray:
        out = np.empty((d.h, d.w), dtype=d.dtype)
        self.check(self.lib.vszip_copy_d2h_2d(self.ctx, out.ctypes.data, out.strides[0], d.ptr, d.stride * d.dtype.itemsize, d.w * d.dtype.itemsize, d.h))
        self.sync()
        return out

    # -- timing ---------------------------------------------------------------
    def timer_start(self):
        self.check(self.lib.vszip_timer_start(self.ctx))

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
