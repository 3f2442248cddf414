// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of vszip.Bilateral.
//
// Follows (vszip v19.0.0):
//   src/vapoursynth/bilateral.zig:100-231   parameter derivation (sigmaS per plane, PBFICnum,
//                                           radius/step/samples, algorithm choice), LUT allocation
//   src/filters/bilateral.zig:15-36         rangeIndex / valOf / finalize
//   src/filters/bilateral.zig:91-171        pbfic (algorithm 1)
//   src/filters/bilateral.zig:178-304       truncated (algorithm 2): interior, pixel, edges
//   src/filters/bilateral.zig:306-334       spatial / range LUT generation
//   src/filters/bilateral.zig:336-431       recursive Gaussian (parameters, vertical, horizontal)
#include <algorithm>
#include <type_traits>

#include "oracle_common.h"

namespace {

// bilateral.zig:15-22
template <typename T>
static inline uint32_t range_index(T a, T b) {
    if constexpr (px_traits<T>::is_int) {
        return a > b ? (uint32_t)(a - b) : (uint32_t)(b - a);  // hz.absDiff
    } else {
        float ad = std::fabs(px_traits<T>::to_f32(a) - px_traits<T>::to_f32(b));
        if constexpr (std::is_same<T, half_t>::value) ad = half_to_float(float_to_half(ad));  // |a-b| in f16
        return (uint32_t)std::trunc(std::fmin(1.0f, ad) * 65535.0f + 0.5f);
    }
}

// bilateral.zig:30-36
template <typename T>
static inline T finalize(float sum, float wsum, float peak) {
    if constexpr (px_traits<T>::is_int) {
        float v = sum / wsum + 0.5f;
        v = v < 0.0f ? 0.0f : (v > peak ? peak : v);  // math.clamp
        return (T)std::trunc(v);
    } else {
        return px_traits<T>::from_f32(sum / wsum);
    }
}

// bilateral.zig:178-304 — one formula covers interior (:184-233) and edge bands
// (:266-304): the edges clamp coordinates (replicate), which is the identity inside.
// Only the 4 diagonal-quadrant taps (+-xx, +-yy), xx,yy in {1, 1+step, ...} <= radius,
// are sampled; axis taps never are.
template <typename T>
static void truncated(const T* src, const T* ref, T* dst, const float* gs, const float* gr, ptrdiff_t sstride, ptrdiff_t rstride, ptrdiff_t dstride, int w, int h, int radius, int step, float peak) {
    const int radius2 = radius + 1;
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            const T cx = ref[(ptrdiff_t)y * rstride + x];
            float wsum = gs[0] * gr[0];
            float sum = px_traits<T>::to_f32(src[(ptrdiff_t)y * sstride + x]) * wsum;
            for (int yy = 1; yy < radius2; yy += step) {
                const int ya = std::max(y - yy, 0), yb = std::min(y + yy, h - 1);
                for (int xx = 1; xx < radius2; xx += step) {
                    const int xa = std::min(x + xx, w - 1), xb = std::max(x - xx, 0);
                    const float swei = gs[yy * radius2 + xx];
                    const float rw1 = gr[range_index<T>(cx, ref[(ptrdiff_t)ya * rstride + xa])];
                    const float rw2 = gr[range_index<T>(cx, ref[(ptrdiff_t)yb * rstride + xa])];
                    const float rw3 = gr[range_index<T>(cx, ref[(ptrdiff_t)ya * rstride + xb])];
                    const float rw4 = gr[range_index<T>(cx, ref[(ptrdiff_t)yb * rstride + xb])];
                    wsum += swei * (rw1 + rw2 + rw3 + rw4);
                    const float s1 = px_traits<T>::to_f32(src[(ptrdiff_t)ya * sstride + xa]);
                    const float s2 = px_traits<T>::to_f32(src[(ptrdiff_t)yb * sstride + xa]);
                    const float s3 = px_traits<T>::to_f32(src[(ptrdiff_t)ya * sstride + xb]);
                    const float s4 = px_traits<T>::to_f32(src[(ptrdiff_t)yb * sstride + xb]);
                    sum += swei * (s1 * rw1 + s2 * rw2 + s3 * rw3 + s4 * rw4);
                }
            }
            dst[(ptrdiff_t)y * dstride + x] = finalize<T>(sum, wsum, peak);
        }
    }
}

// bilateral.zig:350-364
static void rg_params(double sigma, float* b, float* b1, float* b2, float* b3) {
    const double q = (sigma < 2.5) ? (3.97156 - 4.14554 * std::sqrt(1 - 0.26891 * sigma)) : 0.98711 * sigma - 0.96330;
    const double den = 1.57825 + 2.44413 * q + 1.4281 * q * q + 0.422205 * q * q * q;
    const double n1 = 2.44413 * q + 2.85619 * q * q + 1.26661 * q * q * q;
    const double n2 = -(1.4281 * q * q + 1.26661 * q * q * q);
    const double n3 = 0.422205 * q * q * q;
    *b = (float)(1 - (n1 + n2 + n3) / den);
    *b1 = (float)(n1 / den);
    *b2 = (float)(n2 / den);
    *b3 = (float)(n3 / den);
}

// bilateral.zig:366-409 (in place: output == input)
static void rg_vertical(float* io, int height, int width, int stride, float b, float b1, float b2, float b3) {
    for (int j = 0; j < height; ++j) {
        size_t x0 = (size_t)stride * j;
        size_t x1 = j < 1 ? x0 : x0 - stride;
        size_t x2 = j < 2 ? x1 : x1 - stride;
        size_t x3 = j < 3 ? x2 : x2 - stride;
        for (int i = 0; i < width; ++i, ++x0, ++x1, ++x2, ++x3) io[x0] = b * io[x0] + b1 * io[x1] + b2 * io[x2] + b3 * io[x3];
    }
    for (int j = height - 1; j >= 0; --j) {
        size_t x0 = (size_t)stride * j;
        size_t x1 = j >= height - 1 ? x0 : x0 + stride;
        size_t x2 = j >= height - 2 ? x1 : x1 + stride;
        size_t x3 = j >= height - 3 ? x2 : x2 + stride;
        for (int i = 0; i < width; ++i, ++x0, ++x1, ++x2, ++x3) io[x0] = b * io[x0] + b1 * io[x1] + b2 * io[x2] + b3 * io[x3];
    }
}

// bilateral.zig:411-431 (in place)
static void rg_horizontal(float* io, int height, int width, int stride, float b, float b1, float b2, float b3) {
    for (int j = 0; j < height; ++j) {
        const size_t lower = (size_t)stride * j, upper = lower + width;
        size_t i = lower;
        float p1 = io[i], p2 = p1, p3 = p2;
        io[i] = p3;
        ++i;
        for (; i < upper; ++i) {
            const float p0 = b * io[i] + b1 * p1 + b2 * p2 + b3 * p3;
            p3 = p2;
            p2 = p1;
            p1 = p0;
            io[i] = p0;
        }
        --i;
        p1 = io[i];
        p2 = p1;
        p3 = p2;
        if (i == lower) continue;
        --i;
        for (;;) {
            const float p0 = b * io[i] + b1 * p1 + b2 * p2 + b3 * p3;
            p3 = p2;
            p2 = p1;
            p1 = p0;
            io[i] = p0;
            if (i == lower) break;
            --i;
        }
    }
}

// bilateral.zig:91-171 — planes processed with a dense (stride == width) float scratch.
template <typename T>
static void pbfic(const T* src, const T* ref, T* dst, const float* gr, ptrdiff_t sstride, ptrdiff_t rstride, ptrdiff_t dstride, int w, int h, double sigmaS, uint32_t num, float peak) {
    constexpr bool is_float = !px_traits<T>::is_int;
    const size_t pcount = (size_t)w * h;
    std::vector<T> pk(num);
    if constexpr (is_float) {
        // denom and k/denom are computed in T (f16 for half clips)
        for (uint32_t k = 0; k < num; ++k) {
            if constexpr (std::is_same<T, half_t>::value) {
                const float denom = half_to_float(float_to_half((float)(num - 1)));
                pk[k] = float_to_half(half_to_float(float_to_half((float)k)) / denom);
            } else {
                pk[k] = (float)k / (float)(num - 1);
            }
        }
    } else {
        const float numf = (float)num;
        for (uint32_t k = 0; k < num; ++k) {
            const float v = peak * (float)k / (numf - 1) + 0.5f;
            pk[k] = (T)v;  // math.lossyCast: truncation (values are in range)
        }
    }
    float b, b1, b2, b3;
    rg_params(sigmaS, &b, &b1, &b2, &b3);
    std::vector<float> layers((size_t)num * pcount), wk(pcount), jk(pcount);
    for (uint32_t k = 0; k < num; ++k) {
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const size_t i = (size_t)y * w + x;
                wk[i] = gr[range_index<T>(pk[k], ref[(ptrdiff_t)y * rstride + x])];
                jk[i] = wk[i] * px_traits<T>::to_f32(src[(ptrdiff_t)y * sstride + x]);
            }
        rg_horizontal(wk.data(), h, w, w, b, b1, b2, b3);
        rg_vertical(wk.data(), h, w, w, b, b1, b2, b3);
        rg_horizontal(jk.data(), h, w, w, b, b1, b2, b3);
        rg_vertical(jk.data(), h, w, w, b, b1, b2, b3);
        float* lk = layers.data() + (size_t)k * pcount;
        for (size_t i = 0; i < pcount; ++i) lk[i] = (wk[i] == 0) ? 0.0f : (jk[i] / wk[i]);
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const T rv = ref[(ptrdiff_t)y * rstride + x];
            const float rf = px_traits<T>::to_f32(rv);
            uint32_t k = 0;
            for (; k < num - 2; ++k) {
                const float lo = px_traits<T>::to_f32(pk[k]), hi = px_traits<T>::to_f32(pk[k + 1]);
                if (rf < hi && rf >= lo) break;
            }
            const float p0f = px_traits<T>::to_f32(pk[k]), p1f = px_traits<T>::to_f32(pk[k + 1]);
            const float lo = layers[(size_t)k * pcount + i], hi = layers[(size_t)(k + 1) * pcount + i];
            const float vf = ((p1f - rf) * lo + (rf - p0f) * hi) / (p1f - p0f);
            dst[(ptrdiff_t)y * dstride + x] = finalize<T>(vf, 1.0f, peak);
        }
}

}  // namespace

// bilateral.zig:306-314
VSZO_API void vszo_bilateral_gs_lut(float* gs, int upper, double sigmaS) {
    for (int y = 0; y < upper; ++y)
        for (int x = 0; x < upper; ++x) gs[y * upper + x] = (float)std::exp((double)(x * x + y * y) / (sigmaS * sigmaS * -2.0));
}

// bilateral.zig:316-334 (+ normalizedGaussianFunction :336-339)
VSZO_API void vszo_bilateral_gr_lut(float* gr, int len, double range, double sigmaR) {
    const uint32_t upper = (uint32_t)std::trunc(std::min(range, sigmaR * 8.0 * range + 0.5));
    uint32_t i = 0;
    for (; i <= upper && (int)i < len; ++i) {
        const double j = (double)i / range;
        const double x = j / sigmaR;
        gr[i] = (float)(std::exp(x * x / -2) / (std::sqrt(2.0 * M_PI) * sigmaR));
    }
    if ((int)i < len) {
        const float up = gr[upper];
        for (; (int)i < len; ++i) gr[i] = up;
    }
}

// Per-plane parameter derivation, bilateral.zig(vs):104-199. Inputs: the user arrays
// already expanded to 3 entries (hz.getArray semantics are the caller's job) except
// sigmaS, whose chroma default depends on subsampling (:104-124) and is done here.
// n_sigmaS = number of sigmaS values the user gave (0..3).
VSZO_API int vszo_bilateral_params(const double* sigmaS_in, int n_sigmaS, const double* sigmaR, const int* algorithm_in, const int* pbfic_in,
                                   int is_yuv, int ssw, int ssh, const int* planes_in,
                                   double* sigmaS, int* planes, int* algorithm, int* pbficnum, int* radius, int* step, int* samples) {
    for (int i = 0; i < 3; ++i) {
        if (i < n_sigmaS)
            sigmaS[i] = sigmaS_in[i];
        else if (i == 0)
            sigmaS[0] = 3;
        else if (i == 1 && is_yuv && ssh != 0 && ssw != 0)
            sigmaS[1] = sigmaS[0] / std::sqrt((double)((1u << ssh) * (1u << ssw)));
        else
            sigmaS[i] = sigmaS[i - 1];
        if (sigmaS[i] < 0) return -1;
    }
    for (int i = 0; i < 3; ++i) {
        planes[i] = planes_in[i];
        if (sigmaS[i] == 0 || sigmaR[i] == 0) planes[i] = 0;
        algorithm[i] = algorithm_in[i];
        pbficnum[i] = pbfic_in[i];
        radius[i] = step[i] = samples[i] = 0;
    }
    for (int i = 0; i < 3; ++i)
        if (pbficnum[i] == 1) return -2;
    for (int i = 0; i < 3; ++i) {
        if (planes[i] && pbficnum[i] == 0) {
            if (sigmaR[i] >= 0.08)
                pbficnum[i] = 4;
            else if (sigmaR[i] >= 0.015)
                pbficnum[i] = std::min(16, (int)std::trunc(4 * 0.08 / sigmaR[i] + 0.5));
            else
                pbficnum[i] = std::min(32, (int)std::trunc(16 * 0.015 / sigmaR[i] + 0.5));
            if (i > 0 && is_yuv && (pbficnum[i] % 2 == 0) && pbficnum[i] < 256) pbficnum[i] += 1;
        }
    }
    for (int i = 0; i < 3; ++i) {
        if (!planes[i]) continue;
        const int orad = std::max((int)std::trunc(sigmaS[i] * 2 + 0.5), 1);
        step[i] = orad < 4 ? 1 : (orad < 8 ? 2 : 3);
        samples[i] = 1;
        radius[i] = 1 + (samples[i] - 1) * step[i];
        while (orad * 2 > radius[i] * 3) {
            samples[i] += 1;
            radius[i] = 1 + (samples[i] - 1) * step[i];
            if (radius[i] >= orad && samples[i] > 2) {
                samples[i] -= 1;
                radius[i] = 1 + (samples[i] - 1) * step[i];
                break;
            }
        }
    }
    for (int i = 0; i < 3; ++i) {
        if (planes[i] && algorithm[i] <= 0) {
            algorithm[i] = (step[i] == 1) ? 2 : ((sigmaR[i] < 0.08 && samples[i] < 5) ? 2 : ((4 * samples[i] * samples[i] <= 15 * pbficnum[i]) ? 2 : 1));
        }
    }
    return 0;
}

// One plane. `peak` = hist_len - 1 (float clips: 65535), bilateral.zig(vs):101-102.
// gs (algorithm 2) has (radius+1)^2 entries; gr has hist_len entries. ref may equal src.
VSZO_API int vszo_bilateral_plane(int dtype, const void* src, const void* ref, void* dst, ptrdiff_t sstride, ptrdiff_t rstride, ptrdiff_t dstride,
                                  int w, int h, int algorithm, int radius, int step, const float* gs, const float* gr, double sigmaS, int pbficnum, float peak) {
#define RUN(T)                                                                                                                          \
    if (algorithm == 1)                                                                                                                 \
        pbfic<T>((const T*)src, (const T*)ref, (T*)dst, gr, sstride, rstride, dstride, w, h, sigmaS, (uint32_t)pbficnum, peak);           \
    else                                                                                                                                \
        truncated<T>((const T*)src, (const T*)ref, (T*)dst, gs, gr, sstride, rstride, dstride, w, h, radius, step, peak);                 \
    return 0;
    switch (dtype) {
        case VSZO_U8: RUN(uint8_t)
        case VSZO_U16: RUN(uint16_t)
        case VSZO_F16: RUN(half_t)
        case VSZO_F32: RUN(float)
    }
#undef RUN
    return -1;
}
