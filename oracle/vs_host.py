"""TEST INFRASTRUCTURE ONLY — numpy restatement of the HOST-side steps the reference's tests
and wrappers lean on but which live outside /root/reference: VapourSynth's `std.BoxBlur` and the
zimg conversions behind `resize.*` (depth conversion, gray -> RGB, sRGB -> linear light).

Third-party code, absent from the reference tree and from this image:
  * VapourSynth core (`pyproject.toml:13`: VapourSynth>=75) — `std.BoxBlur`, used by the
    reference's tests to build distorted / companion clips (tests/test_ssimulacra2.py:19-26,
    tests/test_adaptive_binarize.py:12-14, tests/test_planeaverage.py:71-73);
  * zimg (version unpinned, whatever the VapourSynth build carries) — `resize.Bicubic(format=RGBS,
    matrix_in=...)` in `hz.toRGBS` (src/helper.zig:225-243) and `resize.Bicubic(transfer=LINEAR)`
    in `sRGBtoLinearRGB` (src/vapoursynth/ssimulacra2.zig:132-162).
Neither source is available here, so every rule below is restated from the published behaviour
and PINNED BY THE REFERENCE'S OWN GOLDENS (tests/test_oracle_vs_host.py):
  * integer std.BoxBlur = horizontal then vertical running box, replicated edges,
    dst = (sum + 2r) / (2r+1)  [sic: the rounding term is 2r, not r] — reproduces all six
    `adaptive_binarize.json` GRAY8|full keys and the RGB24 ones to the pixel count, and the
    `planeaverage.json` YUV420P8 `|ref3` luma diff to every digit;
  * float std.BoxBlur = the same walk with an f32 running sum times f32(1/(2r+1)), replicated
    edges — `planeaverage.json` `RGBS|...|ref3`;
  * zimg integer -> float: (v - offset) * f32(1/range), full range for RGB, limited for Gray/YUV;
  * zimg sRGB -> linear with VapourSynth's default approximate_gamma=1: a 2^16+1-entry table over
    [-0.5, 1.5] indexed by rint(x * 32768 + 16384), entries from the sRGB EOTF with zimg's
    constants (alpha 1.055010718947587, beta 0.003041282560128) — with it the oracle meets
    `ssimulacra2.json` `RGBS|full|dist=blur1` to 7e-6 relative (the textbook EOTF: 8e-4),
    and the RGB24 / GRAY8 keys to < 1e-4.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

SRGB_ALPHA = 1.055010718947587
SRGB_BETA = 0.003041282560128


# ---------------------------------------------------------------------------------------
# std.BoxBlur (VapourSynth core)
# ---------------------------------------------------------------------------------------
def _rep_index(n: int, r: int) -> np.ndarray:
    return np.clip(np.arange(-r, n + r), 0, n - 1)


def _box_rows_int(a: np.ndarray, r: int) -> np.ndarray:
    h, w = a.shape
    p = a[:, _rep_index(w, r)].astype(np.int64)
    c = np.concatenate([np.zeros((h, 1), np.int64), np.cumsum(p, 1)], 1)
    s = c[:, 2 * r + 1:] - c[:, :-(2 * r + 1)]
    return ((s + 2 * r) // (2 * r + 1)).astype(a.dtype)


def _box_rows_float(a: np.ndarray, r: int) -> np.ndarray:
    """f32 running sum along each row: acc starts as r copies of the first sample plus the first r
    samples; per output: acc += entering, dst = acc * div, acc -= leaving (edges replicated)."""
    h, w = a.shape
    div = np.float32(1.0 / (2 * r + 1))
    src = a.astype(np.float32)
    acc = np.float32(r) * src[:, 0]
    for x in range(r):
        acc = acc + src[:, min(x, w - 1)]
    out = np.empty_like(src)
    for x in range(w):
        acc = acc + src[:, min(x + r, w - 1)]
        out[:, x] = acc * div
        acc = acc - src[:, max(x - r, 0)]
    return out


def std_boxblur(plane: np.ndarray, hradius: int = 1, vradius: int = 1) -> np.ndarray:
    """VapourSynth `std.BoxBlur(hradius, vradius)`, one pass per axis, on one plane."""
    fn = _box_rows_int if plane.dtype.kind == "u" else _box_rows_float
    out = np.ascontiguousarray(plane)
    if hradius > 0:
        out = fn(out, hradius)
    if vradius > 0:
        out = np.ascontiguousarray(fn(np.ascontiguousarray(out.T), vradius).T)
    return out


# ---------------------------------------------------------------------------------------
# zimg: depth conversion, gray -> RGB, sRGB -> linear
# ---------------------------------------------------------------------------------------
def int_to_float(plane: np.ndarray, bits: int, limited: bool, chroma: bool = False) -> np.ndarray:
    """zimg integer -> f32 as its x86 kernels do it: fma(v, f32(1 / range), f32(-offset / range))."""
    if limited:
        off = (128 if chroma else 16) << (bits - 8)
        rng = (224 if chroma else 219) << (bits - 8)
    else:
        off = (1 << (bits - 1)) if chroma else 0
        rng = (1 << bits) - 1
    if off == 0:
        return (plane.astype(np.float32) * np.float32(1.0 / rng)).astype(np.float32)
    return fma32(plane.astype(np.float32), np.float32(1.0 / rng), np.float32(-off / rng))


def srgb_eotf(x: np.ndarray) -> np.ndarray:
    """zimg's sRGB EOTF: negative input is clamped to 0 first (as its transfer functions do) — found
    in round 3 through the YUV goldens, whose decoded RGB leaves [0, 1]: with a linear extension
    below zero `ssimulacra2.json` `YUV420P8|full|dist=blur1` is 9e-4 off, with the clamp 2e-5."""
    x = np.maximum(x.astype(np.float64), 0.0)
    lin = np.where(x < 12.92 * SRGB_BETA, x / 12.92, ((x + (SRGB_ALPHA - 1.0)) / SRGB_ALPHA) ** 2.4)
    return lin.astype(np.float32)


_LUT = None


def srgb_to_linear_lut() -> np.ndarray:
    """The 65537-entry table of zimg's approximate-gamma path: entry i = EOTF(i / 65536 * 2 - 0.5)."""
    global _LUT
    if _LUT is None:
        x = (np.arange(65537, dtype=np.float32) / np.float32(65536) * np.float32(2) - np.float32(0.5)).astype(np.float32)
        _LUT = srgb_eotf(x)
    return _LUT


def srgb_to_linear(x: np.ndarray) -> np.ndarray:
    idx = np.rint(x.astype(np.float32) * np.float32(32768) + np.float32(16384)).astype(np.int64)
    return srgb_to_linear_lut()[np.clip(idx, 0, 65536)]


def to_rgbs(planes, family: str, bits: int = 32) -> list:
    """`hz.toRGBS` for the inputs that need no resampler: RGB24/48 (full range), RGBS (as is),
    Gray8/16 (limited range; R = G = B = Y through any YUV -> RGB matrix since U = V = 0)."""
    if family == "RGBS":
        return [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
    if family == "RGB":
        return [int_to_float(p, bits, False) for p in planes]
    if family == "GRAY":
        y = planes[0].astype(np.float32) if planes[0].dtype.kind == "f" else int_to_float(planes[0], bits, True)
        return [y, y, y]
    raise ValueError(family)


def to_linear_rgbs(planes, family: str, bits: int = 32) -> list:
    """toRGBS followed by sRGBtoLinearRGB — what vszip.SSIMULACRA2 feeds its kernel."""
    return [srgb_to_linear(p) for p in to_rgbs(planes, family, bits)]


# ---------------------------------------------------------------------------------------
# zimg: resampler + colour matrix (round 3) — what `resize.Bilinear(format=YUV..., matrix=1)`
# (the reference's fixture path, tests/conftest.py:88-102) and `resize.Bicubic(format=RGBS,
# matrix_in=...)` (`hz.toRGBS`, src/helper.zig:225-243) do.  zimg's source is not in this image
# (third party, version unpinned: whatever VapourSynth >= 75 ships); the rules below restate its
# published algorithm (graph order, filter construction, x86 FMA kernels' accumulation order) and
# were SELECTED BY THE REFERENCE'S OWN GOLDENS among the plausible variants
# (tools/zimg_variant_search.py keeps the search): with them
#   * GRAYS avg (planeaverage.json `GRAYS|full|exclude=[-1]`) matches to every digit and the GRAY16 /
#     YUV420P16 luma sum exactly  -> the matrix is an FMA chain  c0*R, fma(c1,G,.), fma(c2,B,.);
#   * the U and V plane sums of YUV420P8 and YUV420P16 (planeaverage.json) match exactly, as do the
#     f32 extremes of YUV420PS (planeminmax.json)  -> chroma 4:4:4 -> 4:2:0 is vertical first, then
#     horizontal, taps accumulated in two interleaved FMA accumulators (even / odd taps) that are
#     added at the end; float -> integer is fma(x, range, offset), round half to even.
# ---------------------------------------------------------------------------------------
def fma32(a, b, c) -> np.ndarray:
    """Correctly rounded f32 fused multiply-add on arrays: the exact product (48 bits) fits f64;
    the f64 sum is corrected where it lands exactly between two f32 values (double rounding)."""
    a = np.asarray(a, np.float32).astype(np.float64)
    b = np.asarray(b, np.float32).astype(np.float64)
    c = np.asarray(c, np.float32).astype(np.float64)
    p = a * b
    s = p + c
    bb = s - p
    e = (p - (s - bb)) + (c - bb)  # TwoSum error of s = p + c
    r = s.astype(np.float32)
    r64 = r.astype(np.float64)
    d = s - r64
    if np.any((d != 0) & (e != 0)):
        other = np.where(d > 0, np.nextafter(r, np.float32(np.inf)), np.nextafter(r, np.float32(-np.inf))).astype(np.float32)
        tie = (d != 0) & (np.abs(other.astype(np.float64) - s) == np.abs(d)) & (e != 0)
        hi = np.maximum(r, other)
        lo = np.minimum(r, other)
        r = np.where(tie, np.where(e > 0, hi, lo), r).astype(np.float32)
    return r


def _mul32(a, b):
    return (np.asarray(a, np.float32) * np.asarray(b, np.float32)).astype(np.float32)


def _add32(a, b):
    return (np.asarray(a, np.float32) + np.asarray(b, np.float32)).astype(np.float32)


_KR_KB = {1: (0.2126, 0.0722), 6: (0.299, 0.114), 5: (0.299, 0.114), 9: (0.2627, 0.0593)}  # _Matrix -> (Kr, Kb)


def rgb_to_yuv_matrix(matrix: int) -> np.ndarray:
    """zimg's non-constant-luminance RGB -> YUV matrix (f64)."""
    kr, kb = _KR_KB[matrix]
    kg = 1.0 - kr - kb
    us = 1.0 / (2.0 - 2.0 * kb)
    vs = 1.0 / (2.0 - 2.0 * kr)
    return np.array([[kr, kg, kb], [-kr * us, -kg * us, (1.0 - kb) * us], [(1.0 - kr) * vs, -kg * vs, -kb * vs]], np.float64)


def yuv_to_rgb_matrix(matrix: int) -> np.ndarray:
    """Inverse of the above by cofactors in f64, as zimg forms it."""
    m = rgb_to_yuv_matrix(matrix)
    det = (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0])
           + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))
    inv = np.empty((3, 3), np.float64)
    for i in range(3):
        for j in range(3):
            r = [k for k in range(3) if k != j]
            c = [k for k in range(3) if k != i]
            minor = m[r[0], c[0]] * m[r[1], c[1]] - m[r[0], c[1]] * m[r[1], c[0]]
            inv[i, j] = ((-1) ** (i + j)) * minor / det
    return inv


def apply_matrix(m64: np.ndarray, a, b, c) -> list:
    """zimg's x86 matrix kernel on three f32 planes: out_i = fma(m_i2, c, fma(m_i1, b, m_i0 * a))."""
    m = m64.astype(np.float32)
    return [fma32(m[i, 2], c, fma32(m[i, 1], b, _mul32(m[i, 0], a))) for i in range(3)]


def _bilinear(x):
    return np.maximum(1.0 - np.abs(x), 0.0)


def _bicubic(x, b=0.0, c=0.5):
    x = np.abs(x)
    p0 = (6.0 - 2.0 * b) / 6.0
    p2 = (-18.0 + 12.0 * b + 6.0 * c) / 6.0
    p3 = (12.0 - 9.0 * b - 6.0 * c) / 6.0
    q0 = (8.0 * b + 24.0 * c) / 6.0
    q1 = (-12.0 * b - 48.0 * c) / 6.0
    q2 = (6.0 * b + 30.0 * c) / 6.0
    q3 = (-b - 6.0 * c) / 6.0
    return np.where(x < 1.0, p0 + p2 * x * x + p3 * x * x * x, np.where(x < 2.0, q0 + q1 * x + q2 * x * x + q3 * x * x * x, 0.0))


def _point(x):
    return np.ones_like(np.asarray(x, np.float64))


_FILTERS = {"point": (_point, 0.0), "bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


def zimg_filter(kind: str, src_dim: int, dst_dim: int, shift: float = 0.0):
    """zimg's filter table for one axis: per output sample `left` and `width` coefficients (f64,
    normalised), taps outside the line folded back by reflection about the edge (edge sample
    repeated), rows trimmed to their non-zero span and widened to the table's common width."""
    f, support = _FILTERS[kind]
    scale = dst_dim / src_dim
    step = min(scale, 1.0)
    support = support / step
    fsize = max(int(np.ceil(support * 2)), 1)
    m = np.zeros((dst_dim, src_dim), np.float64)
    for i in range(dst_dim):
        pos = (i + 0.5) / scale + shift
        begin = np.floor(pos - fsize / 2.0 + 0.5) + 0.5
        xs = begin + np.arange(fsize)
        w = f((xs - pos) * step)
        total = w.sum()
        for xpos, wk in zip(xs, w):
            if xpos < 0.0:
                real = -xpos
            elif xpos >= src_dim:
                real = 2.0 * src_dim - xpos
            else:
                real = xpos
            real = min(max(real, 0.0), np.nextafter(float(src_dim), -np.inf))
            m[i, int(np.floor(real))] += wk / total
    nz = m != 0.0
    first = np.where(nz.any(1), nz.argmax(1), 0)
    last = np.where(nz.any(1), src_dim - 1 - nz[:, ::-1].argmax(1), 0)
    width = int((last - first + 1).max())
    left = np.minimum(first, src_dim - width).astype(np.int64)
    coef = np.stack([m[i, left[i]:left[i] + width] for i in range(dst_dim)])
    return left, coef


def _accumulate_two(taps, coefs) -> np.ndarray:
    """zimg's AVX2 f32 resize kernels: taps alternate between two accumulators (k even / k odd),
    each an FMA chain started by a plain product, summed at the end; the vertical kernel handles 8
    taps per sweep and re-enters with the stored partial result as accumulator 0."""
    n = len(taps)
    out = None
    for base in range(0, n, 8):
        a0 = _mul32(coefs[base], taps[base]) if out is None else fma32(coefs[base], taps[base], out)
        a1 = None
        for k in range(base + 1, min(base + 8, n)):
            if (k - base) % 2 == 0:
                a0 = fma32(coefs[k], taps[k], a0)
            else:
                a1 = _mul32(coefs[k], taps[k]) if a1 is None else fma32(coefs[k], taps[k], a1)
        out = a0 if a1 is None else _add32(a0, a1)
    return out


def _accumulate_two_h(taps, coefs) -> np.ndarray:
    """Horizontal kernel: the same two accumulators run over the whole tap list (no re-entry)."""
    a0 = _mul32(coefs[0], taps[0])
    a1 = None
    for k in range(1, len(taps)):
        if k % 2 == 0:
            a0 = fma32(coefs[k], taps[k], a0)
        else:
            a1 = _mul32(coefs[k], taps[k]) if a1 is None else fma32(coefs[k], taps[k], a1)
    return a0 if a1 is None else _add32(a0, a1)


def _resize_axis_f32(p: np.ndarray, axis: int, kind: str, dst_dim: int, shift: float) -> np.ndarray:
    src_dim = p.shape[axis]
    left, coef = zimg_filter(kind, src_dim, dst_dim, shift)
    c32 = coef.astype(np.float32)
    q = p if axis == 0 else p.T
    taps = [q[left + k] for k in range(coef.shape[1])]  # [dst_dim, other]
    cs = [c32[:, k][:, None] for k in range(coef.shape[1])]
    out = _accumulate_two(taps, cs) if axis == 0 else _accumulate_two_h(taps, cs)
    return np.ascontiguousarray(out if axis == 0 else out.T)


def _resize_axis_u16(p: np.ndarray, axis: int, kind: str, dst_dim: int, shift: float, peak: int) -> np.ndarray:
    """zimg's 16-bit integer kernels: coefficients quantised to 1/16384 with error diffusion along
    the row, the remainder given to the largest one; dst = clamp((sum + 8192) >> 14)."""
    src_dim = p.shape[axis]
    left, coef = zimg_filter(kind, src_dim, dst_dim, shift)
    ci = np.zeros(coef.shape, np.int64)
    for i in range(coef.shape[0]):
        err = 0.0
        big, big_k, tot = 0, 0, 0
        for k in range(coef.shape[1]):
            want = coef[i, k] * 16384.0 - err
            v = int(np.rint(want))
            err = v - want
            if abs(v) > big:
                big, big_k = abs(v), k
            tot += v
            ci[i, k] = v
        ci[i, big_k] += 16384 - tot
    q = (p if axis == 0 else p.T).astype(np.int64)
    acc = np.zeros((dst_dim, q.shape[1]), np.int64)
    for k in range(coef.shape[1]):
        acc += ci[:, k][:, None] * q[left + k]
    out = np.clip((acc + 8192) >> 14, 0, peak).astype(np.uint16)
    return np.ascontiguousarray(out if axis == 0 else out.T)


def _h_first(xscale: float, yscale: float) -> bool:
    """zimg's pass-order heuristic (cost of a horizontal pass = 2 x a vertical one)."""
    h_cost = max(xscale, 1.0) * 2.0 + xscale * max(yscale, 1.0)
    v_cost = max(yscale, 1.0) + yscale * max(xscale, 1.0) * 2.0
    return h_cost < v_cost


def resize_plane(p: np.ndarray, dst_w: int, dst_h: int, kind: str, shift_w: float = 0.0, shift_h: float = 0.0, peak: int = 65535) -> np.ndarray:
    """One plane through zimg's two-pass resize (f32 planes: FMA kernels; u16 planes: integer)."""
    h, w = p.shape
    steps = []
    if dst_w != w or shift_w != 0.0:
        steps.append((1, dst_w, shift_w))
    if dst_h != h or shift_h != 0.0:
        steps.append((0, dst_h, shift_h))
    if len(steps) == 2 and not _h_first(dst_w / w, dst_h / h):
        steps.reverse()
    for axis, n, sh in steps:
        p = _resize_axis_f32(p, axis, kind, n, sh) if p.dtype == np.float32 else _resize_axis_u16(p, axis, kind, n, sh, peak)
    return p


def chroma_offset(loc: int, ss: int, vertical: bool) -> float:
    """Position of a sited chroma sample relative to the centre of its 2^ss luma samples, in luma samples.
    `_ChromaLocation`: 0 left, 1 center, 2 top-left, 3 top, 4 bottom-left, 5 bottom.
    zimg's rule (graphbuilder's chroma shift: raw siting shift -/+0.5, scaled by 1 / 2^ss into chroma samples):
    half a luma sample for every ss > 0 — NOT the geometric (2^ss - 1) / 2. The two agree at ss = 1, which is all
    the reference's goldens cover (YUV420P8 / P16); zimg's source is not in the reference tree, so ss = 2 is
    restated from its published behaviour and unpinned."""
    if ss == 0:
        return 0.0
    edge = -0.5
    if vertical:
        return edge if loc in (2, 3) else (-edge if loc in (4, 5) else 0.0)
    return edge if loc in (0, 2, 4) else 0.0


def float_to_int(x: np.ndarray, bits: int, limited: bool, chroma: bool = False) -> np.ndarray:
    """zimg f32 -> integer without dither: fma(x, range, offset), round half to even, clamp."""
    if limited:
        off = (128 if chroma else 16) << (bits - 8)
        rng = (224 if chroma else 219) << (bits - 8)
    else:
        off = (1 << (bits - 1)) if chroma else 0
        rng = (1 << bits) - 1
    y = np.rint(fma32(x, np.float32(rng), np.float32(off)))
    return np.clip(y, 0, (1 << bits) - 1).astype(np.uint8 if bits <= 8 else np.uint16)


int_to_float_fma = int_to_float


def rgb24_to_yuv(rgb, bits: int = 8, ssw: int = 1, ssh: int = 1, matrix: int = 1, sample: str = "int", loc: int = 0, gray: bool = False,
                 kind: str = "bilinear") -> list:
    """`resize.Bilinear(format=YUV4xxP<bits> | YUV4xxPS | GRAY*, matrix=1)` of an RGB24 clip — the
    reference's fixture conversion (tests/conftest.py:88-102).  u8 -> f32 (full range), matrix,
    chroma to the target subsampling with the bilinear kernel widened by the scale, f32 -> integer
    (limited range).  `sample`: "int" | "f32" | "f16"; `kind`: "bilinear", or "point" for the
    reference's temporal fixture (resize.Point, tests/conftest.py:151-168: one tap, the sample the
    sited position falls into — column 2i, row 2i + 1 for 4:2:0)."""
    r, g, b = (int_to_float(np.asarray(rgb[i]), 8, False) for i in range(3))
    y, u, v = apply_matrix(rgb_to_yuv_matrix(matrix), r, g, b)
    planes = [y]
    if not gray:
        h, w = y.shape
        cw, ch = (w + (1 << ssw) - 1) >> ssw, (h + (1 << ssh) - 1) >> ssh
        for c in (u, v):
            if ssw or ssh:
                c = resize_plane(c, cw, ch, kind, chroma_offset(loc, ssw, False), chroma_offset(loc, ssh, True))
            planes.append(c)
    if sample == "f32":
        return planes
    if sample == "f16":
        return [p.astype(np.float16) for p in planes]
    return [float_to_int(p, bits, True, chroma=i > 0) for i, p in enumerate(planes)]


def yuv_to_rgbs(planes, bits: int = 8, ssw: int = 1, ssh: int = 1, matrix: int | None = None, loc: int = 0, limited: bool = True) -> list:
    """`hz.toRGBS` on a YUV clip (src/helper.zig:225-243): `resize.Bicubic(format=RGBS,
    matrix_in = 1 if height > 650 else 6)`, VapourSynth's Bicubic defaults b = 0, c = 0.5.
    integer -> f32, chroma to 4:4:4 (Catmull-Rom, horizontal pass first when both axes double),
    YUV -> RGB matrix.  Float clips skip the depth step."""
    y = planes[0]
    h, w = y.shape
    if matrix is None:
        matrix = 1 if h > 650 else 6
    out = []
    for i, p in enumerate(planes):
        if p.dtype.kind == "u":
            p = int_to_float_fma(p, bits, limited, chroma=i > 0)
        else:
            p = np.ascontiguousarray(p, dtype=np.float32)
        if i > 0 and (ssw or ssh):
            p = resize_plane(p, w, h, "bicubic", -chroma_offset(loc, ssw, False) / (1 << ssw), -chroma_offset(loc, ssh, True) / (1 << ssh))
        out.append(p)
    return apply_matrix(yuv_to_rgb_matrix(matrix), *out)


def yuv_to_linear_rgbs(planes, bits: int = 8, ssw: int = 1, ssh: int = 1, matrix: int | None = None, loc: int = 0) -> list:
    return [srgb_to_linear(p) for p in yuv_to_rgbs(planes, bits, ssw, ssh, matrix, loc)]


def resize_yuv_int(planes, bits: int, dst_w: int, dst_h: int, ssw: int = 1, ssh: int = 1, kind: str = "bicubic", loc: int = 0) -> list:
    """`clip.resize.Bicubic(w, h)` on an integer YUV clip (the reference's `dist=resize` distortion,
    tests/test_ssimulacra2.py:20-21): 8-bit planes are widened to 16 bits (<< 8), resized with the
    16-bit integer kernels, and narrowed again (round half to even); chroma keeps its siting."""
    out = []
    for i, p in enumerate(planes):
        sw, sh = (ssw, ssh) if i > 0 else (0, 0)
        h, w = p.shape
        tw, th = (dst_w + (1 << sw) - 1) >> sw, (dst_h + (1 << sh) - 1) >> sh
        q = p.astype(np.uint16) << (16 - bits) if bits < 16 else p
        # a sited chroma plane keeps its position relative to luma: shift = off_out/2^ss... in source samples
        offw, offh = chroma_offset(loc, sw, False), chroma_offset(loc, sh, True)
        sx = (offw * (w / tw) - offw) / (1 << sw) if sw else 0.0
        sy = (offh * (h / th) - offh) / (1 << sh) if sh else 0.0
        q = resize_plane(q, tw, th, kind, sx, sy)
        if bits < 16:
            q = np.clip(np.rint(q.astype(np.float64) / (1 << (16 - bits))), 0, (1 << bits) - 1).astype(np.uint8 if bits <= 8 else np.uint16)
        out.append(q)
    return out
