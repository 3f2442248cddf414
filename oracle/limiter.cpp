// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of vszip.Limiter.
//
// Follows (vszip v19.0.0):
//   src/vapoursynth/limiter.zig:28-96     LimiterRT / Limiter getFrame: dst = min(max(lo, x), hi) per plane
//   src/vapoursynth/limiter.zig:107-221   limiterCreate: min/max parsing, validation order, path selection
//   src/filters/limiter.zig:66-91         the comptime range tables (full / yuv / rgb per depth, yuvf / rgbf)
#include <algorithm>
#include <cmath>
#include <cstring>

#include "oracle_common.h"

namespace {

// Zig's @max/@min on floats return the non-NaN operand (llvm.maxnum/minnum); std::fmax/fmin do the same.
template <typename T>
static void limit_plane(const T* src, T* dst, ptrdiff_t sstride, ptrdiff_t dstride, int w, int h, double lo_d, double hi_d) {
    for (int y = 0; y < h; ++y) {
        const T* s = src + (ptrdiff_t)y * sstride;
        T* d = dst + (ptrdiff_t)y * dstride;
        if constexpr (px_traits<T>::is_int) {
            const T lo = (T)lo_d, hi = (T)hi_d;
            for (int x = 0; x < w; ++x) d[x] = std::min(std::max(lo, s[x]), hi);
        } else if constexpr (std::is_same<T, half_t>::value) {
            // the bounds are rounded to f16 first (@floatCast / comptime_float -> f16), the compare is on f16 values
            const float lo = half_to_float(float_to_half((float)lo_d)), hi = half_to_float(float_to_half((float)hi_d));
            for (int x = 0; x < w; ++x) d[x] = float_to_half(std::fmin(std::fmax(lo, half_to_float(s[x])), hi));
        } else {
            const float lo = (float)lo_d, hi = (float)hi_d;
            for (int x = 0; x < w; ++x) d[x] = std::fmin(std::fmax(lo, s[x]), hi);
        }
    }
}

}  // namespace

// One plane. lo / hi are the bounds as the wrapper holds them (u32 for integer clips, f32 for float clips), passed as f64.
VSZO_API int vszo_limiter(int dtype, const void* src, void* dst, ptrdiff_t sstride, ptrdiff_t dstride, int w, int h, double lo, double hi) {
    switch (dtype) {
        case VSZO_U8: limit_plane<uint8_t>((const uint8_t*)src, (uint8_t*)dst, sstride, dstride, w, h, lo, hi); break;
        case VSZO_U16: limit_plane<uint16_t>((const uint16_t*)src, (uint16_t*)dst, sstride, dstride, w, h, lo, hi); break;
        case VSZO_U32: limit_plane<uint32_t>((const uint32_t*)src, (uint32_t*)dst, sstride, dstride, w, h, lo, hi); break;
        case VSZO_F16: limit_plane<half_t>((const half_t*)src, (half_t*)dst, sstride, dstride, w, h, lo, hi); break;
        case VSZO_F32: limit_plane<float>((const float*)src, (float*)dst, sstride, dstride, w, h, lo, hi); break;
        default: return -1;
    }
    return 0;
}

// The bounds limiterCreate ends up with when no min/max arrays are given (src/filters/limiter.zig:66-91):
// integer clips full range [0, 2^bits - 1], or with tv_range [16, 235 | 240] << (bits - 8) (chroma 240 only
// for YUV clips that are not masks); float clips [0, 1] / chroma [-0.5, 0.5] for YUV non-mask clips, tv_range or not.
VSZO_API void vszo_limiter_default_range(int is_float, int bits, int yuv, int tv_range, double lo3[3], double hi3[3]) {
    for (int p = 0; p < 3; ++p) {
        if (is_float) {
            lo3[p] = (yuv && p > 0) ? -0.5 : 0.0;
            hi3[p] = (yuv && p > 0) ? 0.5 : 1.0;
        } else if (tv_range) {
            lo3[p] = (double)(16ull << (bits - 8));
            hi3[p] = (double)(((yuv && p > 0) ? 240ull : 235ull) << (bits - 8));
        } else {
            lo3[p] = 0.0;
            hi3[p] = (double)((1ull << bits) - 1);
        }
    }
}
