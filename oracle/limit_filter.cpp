// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of vszip.LimitFilter.
//
// Follows (vszip v19.0.0):
//   src/filters/limit_filter.zig:3-34        process: soft limit of (flt - ref) with dark / bright thresholds and elasticity
//   src/vapoursynth/limit_filter.zig:82-125  limitFilterCreate: getArray defaults / bounds, thresholds scaled from the 8-bit scale
//   src/helper.zig:281-336                   getLowestValue / getPeakValue / scaleValue
#include <cmath>

#include "oracle_common.h"

namespace {

template <typename T>
static void limit_filter_plane(const T* flt, const T* src, const T* ref, T* dst, ptrdiff_t fs, ptrdiff_t ss, ptrdiff_t rs, ptrdiff_t ds, int w, int h, float dark_thr,
                               float bright_thr, float elast) {
    for (int y = 0; y < h; ++y) {
        const T *f = flt + y * fs, *s = src + y * ss, *r = ref + y * rs;
        T* d = dst + y * ds;
        for (int x = 0; x < w; ++x) {
            const float sf = px_traits<T>::to_f32(s[x]), ff = px_traits<T>::to_f32(f[x]), rf = px_traits<T>::to_f32(r[x]);
            const float diff_signed = ff - rf, diff_abs = std::fabs(diff_signed);
            const float thr1 = diff_signed > 0 ? bright_thr : dark_thr;
            const float thr2 = thr1 * elast;
            float out;
            if (diff_abs <= thr1)
                out = ff;
            else if (diff_abs >= thr2)
                out = sf;
            else
                out = sf + (ff - sf) * (thr2 - diff_abs) / (thr2 - thr1);  // :28, plain f32 operations in this order
            if constexpr (px_traits<T>::is_int)
                d[x] = (T)std::trunc(out + 0.5f);
            else
                d[x] = px_traits<T>::from_f32(out);
        }
    }
}

}  // namespace

VSZO_API int vszo_limit_filter(int dtype, const void* flt, const void* src, const void* ref, void* dst, ptrdiff_t fs, ptrdiff_t ss, ptrdiff_t rs, ptrdiff_t ds, int w,
                               int h, float dark_thr, float bright_thr, float elast) {
    if (!ref) {
        ref = src;
        rs = ss;
    }
    switch (dtype) {
        case VSZO_U8: limit_filter_plane<uint8_t>((const uint8_t*)flt, (const uint8_t*)src, (const uint8_t*)ref, (uint8_t*)dst, fs, ss, rs, ds, w, h, dark_thr, bright_thr, elast); break;
        case VSZO_U16: limit_filter_plane<uint16_t>((const uint16_t*)flt, (const uint16_t*)src, (const uint16_t*)ref, (uint16_t*)dst, fs, ss, rs, ds, w, h, dark_thr, bright_thr, elast); break;
        case VSZO_F16: limit_filter_plane<half_t>((const half_t*)flt, (const half_t*)src, (const half_t*)ref, (half_t*)dst, fs, ss, rs, ds, w, h, dark_thr, bright_thr, elast); break;
        case VSZO_F32: limit_filter_plane<float>((const float*)flt, (const float*)src, (const float*)ref, (float*)dst, fs, ss, rs, ds, w, h, dark_thr, bright_thr, elast); break;
        default: return -1;
    }
    return 0;
}

// hz.scaleValue(value, clip, zapi, .{}) :312-336 with the default options (depth_in 8, integer, luma):
// a threshold on the 8-bit scale carried to the clip's depth / sample type. `limited` is the clip's
// colour range (frame 0's prop, else RGB -> full, others -> limited).
VSZO_API float vszo_scale_value_from_8bit(float value, int is_float, int bits, int limited) {
    if (!is_float && bits == 8) return value;
    auto lowest = [&](bool flt, int b) -> float { return flt ? 0.0f : (limited ? (float)(16 << (b - 8)) : 0.0f); };
    auto peak = [&](bool flt, int b) -> float { return flt ? 1.0f : (limited ? (float)(235 << (b - 8)) : (float)((1 << b) - 1)); };
    // fmt_in = the clip's format with 8 bits, integer
    float out = value * ((peak(is_float, bits) - lowest(is_float, bits)) / (peak(false, 8) - lowest(false, 8)));
    if (!is_float) out = std::fmax(std::fmin(std::round(out), (float)((1 << bits) - 1)), 0.0f);
    return out;
}
