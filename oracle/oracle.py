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
_LIB_PATH = _DIR / "liboracle_vszip.so"

U8, U16, F16, F32 = 0, 1, 2, 3
_NP2DT = {np.dtype(np.uint8): U8, np.dtype(np.uint16): U16, np.dtype(np.float16): F16, np.dtype(np.float32): F32}


def build(force: bool = False) -> Path:
    """Compile the oracle with g++ (make). Safe to call repeatedly."""
    srcs = list(_DIR.glob("*.cpp")) + list(_DIR.glob("*.h"))
    stale = (not _LIB_PATH.is_file()) or (
        srcs and max(p.stat().st_mtime for p in srcs) > _LIB_PATH.stat().st_mtime
    )
    if force or stale:
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
    vp, i, pd = C.c_void_p, C.c_int, C.c_ssize_t
    l.vszo_boxblur.argtypes = [i, vp, vp, pd, pd, i, i, i, i, i, i]
    l.vszo_boxblur.restype = i


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
