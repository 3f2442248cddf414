"""Python face of the VapourSynth-free test host (tests/fakevs/fakevs.cpp): a tiny `vs`-like
API — clips built from numpy planes, `core.vszip.<Filter>(clip, ...)`, `clip.get_frame(n)` — so
the plugin-boundary tests read like the reference's tests/test_*.py."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
ROOT = _DIR.parents[1]
import os as _os

# tests/test_sanitizers.py points these at the ASan/UBSan builds of tests/sanitize/_build
LIB = Path(_os.environ.get("VSZIP_FAKEVS_LIB", _DIR / "libfakevs.so"))
PLUGIN = Path(_os.environ.get("VSZIP_PLUGIN_LIB", ROOT / "vapoursynth-zip_amd" / "libvszip.so"))

GRAY, RGB, YUV = 1, 2, 3
INTEGER, FLOAT = 0, 1


def fmt_id(cf, st, bits, ssw=0, ssh=0):
    return (cf << 28) | (st << 24) | (bits << 16) | (ssw << 8) | ssh


GRAY8, GRAY16, GRAY32, GRAYH, GRAYS = fmt_id(GRAY, INTEGER, 8), fmt_id(GRAY, INTEGER, 16), fmt_id(GRAY, INTEGER, 32), fmt_id(GRAY, FLOAT, 16), fmt_id(GRAY, FLOAT, 32)
YUV420P8, YUV420P10, YUV420P16, YUV444P16 = fmt_id(YUV, INTEGER, 8, 1, 1), fmt_id(YUV, INTEGER, 10, 1, 1), fmt_id(YUV, INTEGER, 16, 1, 1), fmt_id(YUV, INTEGER, 16)
YUV420PS, YUV444PS = fmt_id(YUV, FLOAT, 32, 1, 1), fmt_id(YUV, FLOAT, 32)
RGB24, RGB48, RGBH, RGBS = fmt_id(RGB, INTEGER, 8), fmt_id(RGB, INTEGER, 16), fmt_id(RGB, FLOAT, 16), fmt_id(RGB, FLOAT, 32)


class Error(RuntimeError):
    pass


def _np_dtype(fid):
    st, bits = (fid >> 24) & 0xF, (fid >> 16) & 0xFF
    if st == FLOAT:
        return np.float16 if bits == 16 else np.float32
    return np.uint8 if bits <= 8 else (np.uint16 if bits <= 16 else np.uint32)


_lib = None


def lib():
    global _lib
    if _lib is None:
        from vszip_amd.capi import _share_torch_hip_runtime

        if "VSZIP_PLUGIN_LIB" not in _os.environ:  # (the sanitizer build links a GPU-less stub)
            _share_torch_hip_runtime()  # one HIP runtime per process (see capi.py)
        l = C.CDLL(str(LIB))
        vp, i, i64 = C.c_void_p, C.c_int, C.c_int64
        l.fakevs_load_plugin.argtypes = [C.c_char_p, C.c_char_p, i]
        l.fakevs_source.restype = vp
        l.fakevs_source.argtypes = [C.c_uint32, i, i, i, i64, i64, i, i]
        l.fakevs_source_frame.restype = vp
        l.fakevs_pull.argtypes = [vp, i, i, i, C.POINTER(C.c_double)]
        l.fakevs_pull_warm.argtypes = [vp, i, i, i, i, C.POINTER(C.c_double)]
        l.fakevs_source_frame.argtypes = [vp, i]
        l.fakevs_node_free.argtypes = [vp]
        l.fakevs_node_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(C.c_uint32), C.POINTER(i64), C.POINTER(i64)]
        l.fakevs_map_new.restype = vp
        l.fakevs_map_free.argtypes = [vp]
        l.fakevs_map_set_int.argtypes = [vp, C.c_char_p, i64]
        l.fakevs_map_set_float.argtypes = [vp, C.c_char_p, C.c_double]
        l.fakevs_map_set_data.argtypes = [vp, C.c_char_p, C.c_char_p]
        l.fakevs_map_set_node.argtypes = [vp, C.c_char_p, vp]
        l.fakevs_map_set_empty.argtypes = [vp, C.c_char_p, i]
        l.fakevs_invoke.restype = vp
        l.fakevs_invoke.argtypes = [C.c_char_p, C.c_char_p, vp]
        l.fakevs_map_error.restype = C.c_char_p
        l.fakevs_map_error.argtypes = [vp]
        l.fakevs_map_node.restype = vp
        l.fakevs_map_node.argtypes = [vp, C.c_char_p]
        l.fakevs_get_frame.restype = vp
        l.fakevs_get_frame.argtypes = [vp, i, C.c_char_p, i]
        l.fakevs_frame_free.argtypes = [vp]
        l.fakevs_frame_plane.restype = C.POINTER(C.c_uint8)
        l.fakevs_frame_plane.argtypes = [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(C.c_ssize_t)]
        l.fakevs_frame_num_planes.argtypes = [vp]
        l.fakevs_frame_prop_count.argtypes = [vp, C.c_char_p]
        l.fakevs_frame_prop_type.argtypes = [vp, C.c_char_p]
        l.fakevs_frame_prop_int.restype = i64
        l.fakevs_frame_prop_int.argtypes = [vp, C.c_char_p, i]
        l.fakevs_frame_prop_float.restype = C.c_double
        l.fakevs_frame_prop_float.argtypes = [vp, C.c_char_p, i]
        l.fakevs_frame_set_prop_int.argtypes = [vp, C.c_char_p, i64]
        l.fakevs_frame_num_props.argtypes = [vp]
        l.fakevs_frame_prop_key.restype = C.c_char_p
        l.fakevs_frame_prop_key.argtypes = [vp, i]
        l.fakevs_plugin_info.argtypes = [C.c_char_p, C.c_char_p, i, C.POINTER(i), C.POINTER(i)]
        l.fakevs_function_args.restype = C.c_char_p
        l.fakevs_function_args.argtypes = [C.c_char_p, C.c_char_p]
        err = C.create_string_buffer(512)
        if l.fakevs_load_plugin(str(PLUGIN).encode(), err, 512) != 0:
            raise ImportError(f"cannot load {PLUGIN}: {err.value.decode()}")
        _lib = l
    return _lib


def _plane_array(frame_ptr, p, dtype):
    w, h, st = C.c_int(), C.c_int(), C.c_ssize_t()
    ptr = lib().fakevs_frame_plane(frame_ptr, p, C.byref(w), C.byref(h), C.byref(st))
    isz = np.dtype(dtype).itemsize
    buf = np.ctypeslib.as_array(ptr, shape=(h.value * st.value,))
    return np.lib.stride_tricks.as_strided(buf.view(dtype), shape=(h.value, w.value), strides=(st.value, isz))


class Frame:
    def __init__(self, ptr, dtype):
        self.ptr, self.dtype = ptr, dtype
        self.planes = [np.array(_plane_array(ptr, p, dtype)) for p in range(lib().fakevs_frame_num_planes(ptr))]
        self.props = {}
        for i in range(lib().fakevs_frame_num_props(ptr)):
            k = lib().fakevs_frame_prop_key(ptr, i)
            n = lib().fakevs_frame_prop_count(ptr, k)
            t = lib().fakevs_frame_prop_type(ptr, k)
            vals = [(lib().fakevs_frame_prop_int(ptr, k, j) if t == 1 else lib().fakevs_frame_prop_float(ptr, k, j)) for j in range(n)] if t in (1, 2) else []
            self.props[k.decode()] = vals[0] if len(vals) == 1 else vals
        lib().fakevs_frame_free(ptr)

    def __getitem__(self, p):
        return self.planes[p]

    def __len__(self):
        return len(self.planes)


class Clip:
    def __init__(self, node):
        self.node = node
        w, h, n, f, a, b = C.c_int(), C.c_int(), C.c_int(), C.c_uint32(), C.c_int64(), C.c_int64()
        lib().fakevs_node_info(node, C.byref(w), C.byref(h), C.byref(n), C.byref(f), C.byref(a), C.byref(b))
        self.width, self.height, self.num_frames, self.format_id = w.value, h.value, n.value, f.value
        self.fps = (a.value, b.value)
        self.vszip = _Namespace(self)

    def get_frame(self, n=0) -> Frame:
        err = C.create_string_buffer(1024)
        ptr = lib().fakevs_get_frame(self.node, n, err, 1024)
        if not ptr:
            raise Error(err.value.decode())
        return Frame(ptr, _np_dtype(self.format_id))

    def pull(self, count, threads, first=0, warm_per_thread=0):
        """`threads` workers fetch `count` frames (numbers modulo the clip length) and drop them,
        like VapourSynth's output loop under fmParallel. -> wall seconds. warm_per_thread > 0:
        every worker first fetches that many frames untimed (VapourSynth's workers outlive a run,
        so per-thread filter state is warm in steady state)."""
        sec = C.c_double()
        failed = lib().fakevs_pull_warm(self.node, first, count, threads, warm_per_thread, C.byref(sec))
        if failed:
            raise Error(f"{failed} frames failed")
        return sec.value

    def __del__(self):
        try:
            lib().fakevs_node_free(self.node)
        except Exception:
            pass


def source(frames, format_id, fps=(24, 1), extra_stride=0, offset=0, props=None) -> Clip:
    """frames: list of frames, each a list of 2-D numpy planes. extra_stride / offset (bytes)
    emulate cropped clips whose plane pointers are offset and whose stride exceeds the width."""
    h, w = frames[0][0].shape
    node = lib().fakevs_source(format_id, w, h, len(frames), fps[0], fps[1], extra_stride, offset)
    dt = _np_dtype(format_id)
    for i, planes in enumerate(frames):
        fp = lib().fakevs_source_frame(node, i)
        for p, a in enumerate(planes):
            _plane_array(fp, p, dt)[...] = np.asarray(a, dt)
        for k, v in (props or {}).items():
            lib().fakevs_frame_set_prop_int(fp, k.encode(), int(v))
    return Clip(node)


def core_standins(on: bool):
    """Register (or remove) the test host's stand-ins for std.SetFrameProps and resize.Point — the
    two core functions the plugin delegates depth conversions to (fakevs.cpp explains what they
    are and are not)."""
    lib().fakevs_enable_core_standins(1 if on else 0)


def fusion_stats():
    """(getFrame calls that ran fused upstream vszip stages, stages run) since the plugin was loaded."""
    lib()  # the plugin is loaded by now
    pl = C.CDLL(str(PLUGIN))
    a, b = C.c_long(), C.c_long()
    pl.vszip_plugin_fusion_stats(C.byref(a), C.byref(b))
    return a.value, b.value


def upload_stats():
    """(host planes the plugin copied to a device, uploads it answered with a copy the same getFrame had made) since it was loaded."""
    lib()
    pl = C.CDLL(str(PLUGIN))
    a, b = C.c_long(), C.c_long()
    pl.vszip_plugin_upload_stats(C.byref(a), C.byref(b))
    return a.value, b.value


def standin_log():
    out, buf, i = [], C.create_string_buffer(256), 0
    while lib().fakevs_standin_log(i, buf, 256):
        out.append(buf.value.decode())
        i += 1
    return out


def blank(format_id, w, h, color, length=1, fps=(24, 1)) -> Clip:
    """std.BlankClip"""
    dt = _np_dtype(format_id)
    ssw, ssh = (format_id >> 8) & 0xFF, format_id & 0xFF
    np_ = 1 if (format_id >> 28) == GRAY else 3
    color = list(color) if isinstance(color, (list, tuple)) else [color] * np_
    planes = [np.full((h >> (ssh if p else 0), w >> (ssw if p else 0)), color[p], dt) for p in range(np_)]
    return source([planes] * length, format_id, fps)


def invoke(fn: str, **kwargs) -> Clip:
    l = lib()
    m = l.fakevs_map_new()
    keep = []
    try:
        for k, v in kwargs.items():
            if v is None:
                continue
            kb = k.encode()
            vals = v if isinstance(v, (list, tuple)) else [v]
            if isinstance(v, (list, tuple)) and not v:
                l.fakevs_map_set_empty(m, kb, 1)
            for x in vals:
                if isinstance(x, Clip):
                    l.fakevs_map_set_node(m, kb, x.node)
                    keep.append(x)
                elif isinstance(x, bool) or isinstance(x, (int, np.integer)):
                    l.fakevs_map_set_int(m, kb, int(x))
                elif isinstance(x, (float, np.floating)):
                    l.fakevs_map_set_float(m, kb, float(x))
                elif isinstance(x, str):
                    l.fakevs_map_set_data(m, kb, x.encode())
                else:
                    raise TypeError(f"{k}: {type(x)}")
        out = l.fakevs_invoke(b"vszip", fn.encode(), m)
        err = l.fakevs_map_error(out)
        if err:
            msg = err.decode()
            l.fakevs_map_free(out)
            raise Error(msg)
        node = l.fakevs_map_node(out, b"clip")
        l.fakevs_map_free(out)
        return Clip(node)
    finally:
        l.fakevs_map_free(m)


class _Namespace:
    def __init__(self, clip=None):
        self._clip = clip

    def __getattr__(self, fn):
        def call(*args, **kw):
            names = signature(fn)
            if self._clip is not None:
                kw = {names[0]: self._clip, **kw}  # clip.vszip.F(...): the clip is the function's first argument
            for name, a in zip([n for n in names if n not in kw], args):
                kw[name] = a
            return invoke(fn, **kw)

        return call


def signature(fn: str):
    s = lib().fakevs_function_args(b"vszip", fn.encode())
    if s is None:
        raise Error(f"no function {fn}")
    return [item.split(":")[0] for item in s.decode().split(";") if item]


def signature_string(fn: str) -> str:
    return lib().fakevs_function_args(b"vszip", fn.encode()).decode()


def plugin_info():
    idb = C.create_string_buffer(128)
    v, n = C.c_int(), C.c_int()
    assert lib().fakevs_plugin_info(b"vszip", idb, 128, C.byref(v), C.byref(n)) == 0
    return idb.value.decode(), v.value, n.value


class _Core:
    vszip = _Namespace()


core = _Core()
