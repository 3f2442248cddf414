// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of vszip.BoxBlur.
//
// Follows (vszip v19.0.0):
//   src/vapoursynth/boxblur.zig:85-113   RT dispatch (H passes, then V passes)
//   src/vapoursynth/boxblur.zig:188      CT/RT path choice
//   src/filters/boxblur_comptime.zig     CT path: hvBlur, mirrorRows, col*, hBlurInt, vBlurFloat, hBlurFloat
//   src/filters/boxblur_runtime.zig      RT path: blurInt, blurFloat, blur_passes, hblur, vblur/vSweep*
#include "oracle_common.h"

namespace {

// ---------------------------------------------------------------------------
// CT path (hradius == vradius in [1,22], one pass per axis): vertical first,
// then horizontal, row by row.  boxblur_comptime.zig:10-46
// ---------------------------------------------------------------------------

// boxblur_comptime.zig:50-70 — the ksize source rows feeding output row i.
// Top: reflect-101 (no edge duplication). Bottom: overshoot mirrors about the
// CURRENT row i, not about the edge.
static inline int ct_tap_row(int k, int i, int radius, int ih) {
    const int dist_from_bottom = ih - 1 - i;
    if (k < radius) {
        return (i < radius - k) ? std::min(radius - k - i, ih - 1) : (i - radius + k);
    }
    return (dist_from_bottom < k - radius) ? (i - std::min(k - radius - dist_from_bottom, i)) : (i - radius + k);
}

// boxblur_comptime.zig:130-159
template <typename T>
static void ct_hblur_int(const T* srcp, T* dstp, uint32_t w, uint32_t ksize, uint64_t inv) {
    const uint32_t radius = ksize >> 1;
    uint64_t sum = srcp[radius];
    const uint64_t inv2 = inv >> 16;
    for (uint32_t x = 0; x < radius; ++x) sum += (uint32_t)srcp[x] << 1;
    sum = (sum * inv + (1ull << 31)) >> 16;

    uint32_t x = 0;
    for (; x <= radius; ++x) {
        sum += (uint32_t)srcp[radius + x] * inv2;
        sum -= (uint32_t)srcp[radius - x] * inv2;
        dstp[x] = (T)(sum >> 16);
    }
    for (; x < w - radius; ++x) {
        sum += (uint32_t)srcp[radius + x] * inv2;
        sum -= (uint32_t)srcp[x - radius - 1] * inv2;
        dstp[x] = (T)(sum >> 16);
    }
    for (; x < w; ++x) {
        sum += (uint32_t)srcp[2 * w - radius - x - 1] * inv2;
        sum -= (uint32_t)srcp[x - radius - 1] * inv2;
        dstp[x] = (T)(sum >> 16);
    }
}

// boxblur_comptime.zig:192-263 — every output column sums `div * tap` in tap
// order; the column index of tap k uses the same asymmetric mirror as the rows.
template <typename T>
static void ct_hblur_float(const T* srcp, T* dstp, int w, int ksize, float div) {
    const int radius = ksize >> 1;
    for (int j = 0; j < w; ++j) {
        float sum = 0.0f;
        for (int k = 0; k < ksize; ++k) {
            const int idx = ct_tap_row(k, j, radius, w);
            sum += div * px_traits<T>::to_f32(srcp[idx]);
        }
        dstp[j] = px_traits<T>::from_f32(sum);
    }
}

template <typename T>
static void ct_hvblur_int(uint32_t radius, const T* src, T* dst, ptrdiff_t sstride, ptrdiff_t dstride, uint32_t w, uint32_t h) {
    const uint32_t ksize = 2 * radius + 1;
    const int ih = (int)h;
    std::vector<T> tmp(w);
    std::vector<uint32_t> col(w);
    const uint64_t inv = ((1ull << 32) + radius) / ksize;  // :28
    for (int i = 0; i < ih; ++i) {
        // :31-36 — interior rows slide the column window by one row (colUpdate
        // :72-89), edge rows recompute it from the mirrored taps (colRecompute :91-112).
        if (i > (int)radius && i + (int)radius < ih) {
            const T* add_row = src + (ptrdiff_t)(i + (int)radius) * sstride;
            const T* sub_row = src + (ptrdiff_t)(i - (int)radius - 1) * sstride;
            for (uint32_t j = 0; j < w; ++j) {
                col[j] += add_row[j];
                col[j] -= sub_row[j];
            }
        } else {
            for (uint32_t j = 0; j < w; ++j) col[j] = 0;
            for (uint32_t k = 0; k < ksize; ++k) {
                const T* row = src + (ptrdiff_t)ct_tap_row((int)k, i, (int)radius, ih) * sstride;
                for (uint32_t j = 0; j < w; ++j) col[j] += row[j];
            }
        }
        for (uint32_t j = 0; j < w; ++j) tmp[j] = (T)(((uint64_t)col[j] * inv + (1ull << 31)) >> 32);  // :114-128
        ct_hblur_int<T>(tmp.data(), dst + (ptrdiff_t)i * dstride, w, ksize, inv);
    }
}

template <typename T>
static void ct_hvblur_float(uint32_t radius, const T* src, T* dst, ptrdiff_t sstride, ptrdiff_t dstride, uint32_t w, uint32_t h) {
    const int ksize = (int)(2 * radius + 1);
    const int ih = (int)h;
    const float div = 1.0f / (float)ksize;  // :39
    std::vector<T> tmp(w);
    for (int i = 0; i < ih; ++i) {
        // vBlurFloat :161-190 — acc = acc + div * v over taps in order, f32.
        for (uint32_t j = 0; j < w; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < ksize; ++k) {
                const float v = px_traits<T>::to_f32(src[(ptrdiff_t)ct_tap_row(k, i, (int)radius, ih) * sstride + j]);
                acc = acc + div * v;
            }
            tmp[j] = px_traits<T>::from_f32(acc);
        }
        ct_hblur_float<T>(tmp.data(), dst + (ptrdiff_t)i * dstride, (int)w, ksize, div);
    }
}

// ---------------------------------------------------------------------------
// RT path: 1-D running box with symmetric edge-duplicating mirror at both ends.
// ---------------------------------------------------------------------------

// boxblur_runtime.zig:10-41
template <typename T>
static void rt_blur_int(const T* srcp, ptrdiff_t sstep, T* dstp, ptrdiff_t dstep, uint32_t len, uint32_t radius) {
    const uint32_t ksize = (radius << 1) + 1;
    const uint64_t inv = ((1ull << 32) + radius) / ksize;
    uint64_t sum = srcp[(ptrdiff_t)radius * sstep];
    const uint64_t inv2 = inv >> 16;
    for (uint32_t x = 0; x < radius; ++x) sum += (uint32_t)srcp[(ptrdiff_t)x * sstep] << 1;
    sum = (sum * inv + (1ull << 31)) >> 16;
    uint32_t x = 0;
    for (; x <= radius; ++x) {
        sum += srcp[(ptrdiff_t)(radius + x) * sstep] * inv2;
        sum -= srcp[(ptrdiff_t)(radius - x) * sstep] * inv2;
        dstp[(ptrdiff_t)x * dstep] = (T)(sum >> 16);
    }
    for (; x < len - radius; ++x) {
        sum += srcp[(ptrdiff_t)(radius + x) * sstep] * inv2;
        sum -= srcp[(ptrdiff_t)(x - radius - 1) * sstep] * inv2;
        dstp[(ptrdiff_t)x * dstep] = (T)(sum >> 16);
    }
    for (; x < len; ++x) {
        sum += srcp[(ptrdiff_t)(2 * len - radius - x - 1) * sstep] * inv2;
        sum -= srcp[(ptrdiff_t)(x - radius - 1) * sstep] * inv2;
        dstp[(ptrdiff_t)x * dstep] = (T)(sum >> 16);
    }
}

// boxblur_runtime.zig:43-79 — a RUNNING f32 sum: rounding depends on order.
template <typename T>
static void rt_blur_float(const T* srcp, ptrdiff_t sstep, T* dstp, ptrdiff_t dstep, uint32_t len, uint32_t radius) {
    const float ksize = (float)(radius * 2 + 1);
    const float div = 1.0f / ksize;
    float sum = px_traits<T>::to_f32(srcp[(ptrdiff_t)radius * sstep]);
    for (uint32_t x = 0; x < radius; ++x) {
        const float srcv = px_traits<T>::to_f32(srcp[(ptrdiff_t)x * sstep]);
        sum += srcv * 2;
    }
    sum = sum * div;
    uint32_t x = 0;
    for (; x <= radius; ++x) {
        const float s1 = px_traits<T>::to_f32(srcp[(ptrdiff_t)(radius + x) * sstep]);
        const float s2 = px_traits<T>::to_f32(srcp[(ptrdiff_t)(radius - x) * sstep]);
        sum += (s1 - s2) * div;
        dstp[(ptrdiff_t)x * dstep] = px_traits<T>::from_f32(sum);
    }
    for (; x < len - radius; ++x) {
        const float s1 = px_traits<T>::to_f32(srcp[(ptrdiff_t)(radius + x) * sstep]);
        const float s2 = px_traits<T>::to_f32(srcp[(ptrdiff_t)(x - radius - 1) * sstep]);
        sum += (s1 - s2) * div;
        dstp[(ptrdiff_t)x * dstep] = px_traits<T>::from_f32(sum);
    }
    for (; x < len; ++x) {
        const float s1 = px_traits<T>::to_f32(srcp[(ptrdiff_t)(2 * len - radius - x - 1) * sstep]);
        const float s2 = px_traits<T>::to_f32(srcp[(ptrdiff_t)(x - radius - 1) * sstep]);
        sum += (s1 - s2) * div;
        dstp[(ptrdiff_t)x * dstep] = px_traits<T>::from_f32(sum);
    }
}

template <typename T>
static inline void rt_blur_1d(const T* s, ptrdiff_t ss, T* d, ptrdiff_t ds, uint32_t len, uint32_t radius) {
    if constexpr (px_traits<T>::is_int)
        rt_blur_int<T>(s, ss, d, ds, len, radius);
    else
        rt_blur_float<T>(s, ss, d, ds, len, radius);
}

// One full-plane pass along one axis, out of place (the reference's per-row
// tmp ping-pong `blur_passes` :81-119, `hblur` :121, `vblur`/`vSweep*`
// :283-416 and `hvBlurFused` :153-274 all reduce to repeated application of
// the same 1-D operator; vSweep keeps one running sum per column with the
// identical op sequence, the fused variant is documented bit-identical).
template <typename T>
static void rt_pass(const T* src, ptrdiff_t sstride, T* dst, ptrdiff_t dstride, uint32_t w, uint32_t h, uint32_t radius, bool vertical) {
    if (!vertical) {
        for (uint32_t y = 0; y < h; ++y) rt_blur_1d<T>(src + (ptrdiff_t)y * sstride, 1, dst + (ptrdiff_t)y * dstride, 1, w, radius);
    } else {
        for (uint32_t x = 0; x < w; ++x) rt_blur_1d<T>(src + x, sstride, dst + x, dstride, h, radius);
    }
}

template <typename T>
static void boxblur_plane(const T* src, T* dst, ptrdiff_t sstride, ptrdiff_t dstride, uint32_t w, uint32_t h, uint32_t hradius, int hpasses, uint32_t vradius, int vpasses) {
    // boxblur.zig:188 — note that the CT path ignores hpasses/vpasses == 0.
    const bool use_rt = (hradius != vradius) || (hradius > 22) || (hpasses > 1) || (vpasses > 1);
    if (!use_rt) {
        if constexpr (px_traits<T>::is_int)
            ct_hvblur_int<T>(hradius, src, dst, sstride, dstride, w, h);
        else
            ct_hvblur_float<T>(hradius, src, dst, sstride, dstride, w, h);
        return;
    }
    // boxblur.zig:85-112 — horizontal passes first, then vertical passes.
    const bool hb = (hradius > 0) && (hpasses > 0);
    const bool vb = (vradius > 0) && (vpasses > 0);
    std::vector<T> a((size_t)w * h), b((size_t)w * h);
    const T* cur = src;
    ptrdiff_t cs = sstride;
    T* bufs[2] = {a.data(), b.data()};
    int which = 0;
    const int total = (hb ? hpasses : 0) + (vb ? vpasses : 0);
    int done = 0;
    auto step = [&](uint32_t radius, bool vertical) {
        ++done;
        T* out = (done == total) ? dst : bufs[which];
        const ptrdiff_t os = (done == total) ? dstride : (ptrdiff_t)w;
        rt_pass<T>(cur, cs, out, os, w, h, radius, vertical);
        cur = out;
        cs = os;
        which ^= 1;
    };
    for (int p = 0; hb && p < hpasses; ++p) step(hradius, false);
    for (int p = 0; vb && p < vpasses; ++p) step(vradius, true);
    if (total == 0) {
        for (uint32_t y = 0; y < h; ++y) std::memcpy(dst + (ptrdiff_t)y * dstride, src + (ptrdiff_t)y * sstride, sizeof(T) * w);
    }
}

}  // namespace

// Strides are in ELEMENTS of the plane's sample type (getDimensions2, boxblur.zig:46).
VSZO_API int vszo_boxblur(int dtype, const void* src, void* dst, ptrdiff_t src_stride, ptrdiff_t dst_stride, int w, int h, int hradius, int hpasses, int vradius, int vpasses) {
    switch (dtype) {
        case VSZO_U8:
            boxblur_plane<uint8_t>((const uint8_t*)src, (uint8_t*)dst, src_stride, dst_stride, w, h, hradius, hpasses, vradius, vpasses);
            return 0;
        case VSZO_U16:
            boxblur_plane<uint16_t>((const uint16_t*)src, (uint16_t*)dst, src_stride, dst_stride, w, h, hradius, hpasses, vradius, vpasses);
            return 0;
        case VSZO_F16:
            boxblur_plane<half_t>((const half_t*)src, (half_t*)dst, src_stride, dst_stride, w, h, hradius, hpasses, vradius, vpasses);
            return 0;
        case VSZO_F32:
            boxblur_plane<float>((const float*)src, (float*)dst, src_stride, dst_stride, w, h, hradius, hpasses, vradius, vpasses);
            return 0;
    }
    return -1;
}
