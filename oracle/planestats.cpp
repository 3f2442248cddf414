// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of
// vszip.PlaneAverage and vszip.PlaneMinMax.
//
// Follows (vszip v19.0.0):
//   src/filters/planeaverage.zig:16-84    result / average / averageRef
//   src/filters/planeminmax.zig:11-139    minMaxImpl (histogram + thresholds), minMaxNoThr(Ref), absDiff
//   src/vapoursynth/planeaverage.zig:110-137, planeminmax.zig:134-147  peak / exclude conversion
#include <algorithm>
#include <type_traits>

#include "oracle_common.h"

namespace {

template <typename T>
static inline double as_f64(T v) {
    if constexpr (std::is_same<T, half_t>::value)
        return (double)half_to_float(v);
    else
        return (double)v;
}

// planeaverage.zig:26-84. Integer planes: exact u64 sums; float planes: f64 sums in
// raster order. `exclude` values are compared for equality in the sample type
// (ints: the i32 list; floats: the list converted with @floatFromInt).
template <typename T>
static void plane_average(const T* src, const T* ref, ptrdiff_t sstride, ptrdiff_t rstride, int w, int h, const int32_t* excl, int nexcl, float peak, double* avg, double* diff) {
    constexpr bool is_int = px_traits<T>::is_int;
    const uint32_t total0 = (uint32_t)w * (uint32_t)h;
    uint32_t total = total0;
    uint64_t iacc = 0, idiff = 0;
    double facc = 0.0, fdiff = 0.0;
    for (int y = 0; y < h; ++y) {
        const T* s = src + (ptrdiff_t)y * sstride;
        const T* r = ref ? ref + (ptrdiff_t)y * rstride : nullptr;
        for (int x = 0; x < w; ++x) {
            bool found = false;
            if constexpr (is_int) {
                for (int e = 0; e < nexcl; ++e)
                    if ((int32_t)s[x] == excl[e]) { found = true; break; }
            } else {
                const float v = px_traits<T>::to_f32(s[x]);
                for (int e = 0; e < nexcl; ++e)
                    if (v == (float)excl[e]) { found = true; break; }
            }
            if (found) {
                total -= 1;
            } else {
                if constexpr (is_int) iacc += s[x]; else facc += as_f64(s[x]);
            }
            if (r) {
                if constexpr (is_int) {
                    idiff += (s[x] > r[x]) ? (uint64_t)(s[x] - r[x]) : (uint64_t)(r[x] - s[x]);  // hz.absDiff
                } else {
                    // hz.absDiff in T (helper.zig:124-126), widened for the f64 accumulator
                    const float a = px_traits<T>::to_f32(s[x]), b = px_traits<T>::to_f32(r[x]);
                    float d = (a > b) ? (a - b) : (b - a);
                    if constexpr (std::is_same<T, half_t>::value) d = half_to_float(float_to_half(d));
                    fdiff += (double)d;
                }
            }
        }
    }
    // result(): planeaverage.zig:16-24
    const double totalf = (double)total;
    if (total == 0)
        *avg = 0.0;
    else if constexpr (is_int)
        *avg = (double)iacc / totalf / (double)peak;
    else
        *avg = facc / totalf;
    if (ref) {
        const double t0 = (double)total0;
        if constexpr (is_int)
            *diff = (double)idiff / t0 / (double)peak;
        else
            *diff = fdiff / t0;
    }
}

// planeminmax.zig:11-70 (thresholded) and :80-133 (exact).
template <typename T>
static void plane_minmax(const T* src, const T* ref, ptrdiff_t sstride, ptrdiff_t rstride, int w, int h, float minthr, float maxthr, int bits, double* omin, double* omax, double* odiff) {
    constexpr bool is_int = px_traits<T>::is_int;
    const double total = (double)((uint32_t)w * (uint32_t)h);
    const uint32_t hist_size = is_int ? (1u << bits) : 65536u;  // planeminmax.zig(vs):147
    const uint16_t peak = (uint16_t)(hist_size - 1);
    const float peakf = (float)peak;
    double diffacc = 0.0;
    const bool no_thr = (maxthr == 0.0f) && (minthr == 0.0f);  // planeminmax.zig(vs):160
    if (no_thr) {
        float fmin = INFINITY, fmax = -INFINITY;
        uint32_t imin = 0xFFFFFFFFu, imax = 0;
        for (int y = 0; y < h; ++y) {
            const T* s = src + (ptrdiff_t)y * sstride;
            const T* r = ref ? ref + (ptrdiff_t)y * rstride : nullptr;
            for (int x = 0; x < w; ++x) {
                if constexpr (is_int) {
                    imin = std::min<uint32_t>(imin, s[x]);
                    imax = std::max<uint32_t>(imax, s[x]);
                    if (r) diffacc += std::fabs((double)s[x] - (double)r[x]);
                } else {
                    const float v = px_traits<T>::to_f32(s[x]);
                    fmin = std::fmin(fmin, v);
                    fmax = std::fmax(fmax, v);
                    if (r) {
                        float d = std::fabs(v - px_traits<T>::to_f32(r[x]));  // @abs(v - j) in T
                        if constexpr (std::is_same<T, half_t>::value) d = half_to_float(float_to_half(d));
                        diffacc += (double)d;
                    }
                }
            }
        }
        if constexpr (is_int) {
            *omin = (double)imin;
            *omax = (double)imax;
        } else {
            *omin = (double)fmin;
            *omax = (double)fmax;
        }
        if (ref) *odiff = is_int ? diffacc / total / (double)peakf : diffacc / total;
        return;
    }
    std::vector<uint32_t> hist(65536, 0);
    for (int y = 0; y < h; ++y) {
        const T* s = src + (ptrdiff_t)y * sstride;
        const T* r = ref ? ref + (ptrdiff_t)y * rstride : nullptr;
        for (int x = 0; x < w; ++x) {
            uint32_t idx;
            if constexpr (is_int) {
                idx = s[x];
                if (r) diffacc += std::fabs((double)s[x] - (double)r[x]);
            } else {
                const float v = px_traits<T>::to_f32(s[x]);
                // math.lossyCast(u16, v * 65535.0 + 0.5): saturating, truncating, NaN -> 0
                const float t = v * 65535.0f + 0.5f;
                idx = (t != t) ? 0u : (t <= 0.0f ? 0u : (t >= 65535.0f ? 65535u : (uint32_t)t));
                if (r) {
                    float d = std::fabs(v - px_traits<T>::to_f32(r[x]));
                    if constexpr (std::is_same<T, half_t>::value) d = half_to_float(float_to_half(d));
                    diffacc += (double)d;
                }
            }
            hist[idx] += 1;
        }
    }
    const uint32_t totalmin = (uint32_t)std::trunc(total * (double)minthr);
    const uint32_t totalmax = (uint32_t)std::trunc(total * (double)maxthr);
    uint32_t count = 0;
    uint32_t retmin = peak;
    for (uint32_t u = 0; u < hist_size; ++u) {
        count += hist[u];
        if (count > totalmin) { retmin = u; break; }
    }
    count = 0;
    uint32_t retmax = 0;
    for (int i = (int)peak; i >= 0; --i) {
        count += hist[(uint32_t)i];
        if (count > totalmax) { retmax = (uint32_t)i; break; }
    }
    if constexpr (is_int) {
        *omin = (double)retmin;
        *omax = (double)retmax;
    } else {
        *omin = (double)((float)retmin / 65535.0f);
        *omax = (double)((float)retmax / 65535.0f);
    }
    if (ref) *odiff = is_int ? diffacc / total / (double)peakf : diffacc / total;
}

}  // namespace

// peak = (1 << bitsPerSample) - 1 as f32 (planeaverage.zig(vs):115); strides in elements.
VSZO_API int vszo_plane_average(int dtype, const void* src, const void* ref, ptrdiff_t sstride, ptrdiff_t rstride, int w, int h, const int32_t* excl, int nexcl, int bits, double* avg, double* diff) {
    const float peak = (float)(((uint64_t)1 << bits) - 1);
    switch (dtype) {
        case VSZO_U8: plane_average<uint8_t>((const uint8_t*)src, (const uint8_t*)ref, sstride, rstride, w, h, excl, nexcl, peak, avg, diff); return 0;
        case VSZO_U16: plane_average<uint16_t>((const uint16_t*)src, (const uint16_t*)ref, sstride, rstride, w, h, excl, nexcl, peak, avg, diff); return 0;
        case VSZO_F16: plane_average<half_t>((const half_t*)src, (const half_t*)ref, sstride, rstride, w, h, excl, nexcl, peak, avg, diff); return 0;
        case VSZO_F32: plane_average<float>((const float*)src, (const float*)ref, sstride, rstride, w, h, excl, nexcl, peak, avg, diff); return 0;
        case VSZO_U32:  // planeaverage.zig(vs):127 rejects exclude for 32-bit integer clips
            if (nexcl > 0) return -1;
            plane_average<uint32_t>((const uint32_t*)src, (const uint32_t*)ref, sstride, rstride, w, h, excl, 0, peak, avg, diff);
            return 0;
    }
    return -1;
}

VSZO_API int vszo_plane_minmax(int dtype, const void* src, const void* ref, ptrdiff_t sstride, ptrdiff_t rstride, int w, int h, float minthr, float maxthr, int bits, double* omin, double* omax, double* odiff) {
    switch (dtype) {
        case VSZO_U8: plane_minmax<uint8_t>((const uint8_t*)src, (const uint8_t*)ref, sstride, rstride, w, h, minthr, maxthr, bits, omin, omax, odiff); return 0;
        case VSZO_U16: plane_minmax<uint16_t>((const uint16_t*)src, (const uint16_t*)ref, sstride, rstride, w, h, minthr, maxthr, bits, omin, omax, odiff); return 0;
        case VSZO_F16: plane_minmax<half_t>((const half_t*)src, (const half_t*)ref, sstride, rstride, w, h, minthr, maxthr, bits, omin, omax, odiff); return 0;
        case VSZO_F32: plane_minmax<float>((const float*)src, (const float*)ref, sstride, rstride, w, h, minthr, maxthr, bits, omin, omax, odiff); return 0;
    }
    return -1;
}
