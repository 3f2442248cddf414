// vszip.Limiter on gfx950: dst = min(max(lo, x), hi) per plane (LimiterRT / Limiter getFrame,
// src/vapoursynth/limiter.zig:28-96; the bounds — explicit min/max arrays or the comptime tables of
// src/filters/limiter.zig:66-91 — are resolved by the wrapper). A pure streaming kernel: 16 bytes per
// lane per access, one launch for a whole table of planes; HBM roofline = every byte read once and
// written once.
#include <algorithm>

#include "common.hpp"

#ifndef VSZIP_STREAM_PLAIN_LOADS
#define VSZIP_STREAM_LOAD(p) __builtin_nontemporal_load(p)  // every sample is read once
#else
#define VSZIP_STREAM_LOAD(p) (*(p))
#endif

namespace {

constexpr int kMaxPlanesL = 192;  // planes per launch (64 YUV frames are ONE launch since round 4: four 48-plane launches paid four ramps and tails)
// Rows per workgroup, measured on 16 4K YUV420P16 frames (tools/ab_stream.sh, round 2): 1 row 0.42 of the HBM
// peak (a workgroup's fixed cost — plane lookup, two half-filled passes over a 480-vector row — dominates),
// 2 rows + non-temporal loads 0.68, 4 rows 0.67. (A pure copy gains from short-lived workgroups in address
// order, profiles/r02_membw.md; with per-workgroup set-up in the way the gain is a few percent.)
#ifndef VSZIP_STREAM_ROWS
#define VSZIP_STREAM_ROWS 2
#endif
constexpr int kRowsPerBlock = VSZIP_STREAM_ROWS;

struct LPlane {
    const void *src;
    void *dst;
    int sstride, dstride, w, h;
    int block0;
    float lo_f, hi_f;      // float clips (already rounded to the sample type's precision by the host for f16)
    uint32_t lo_u, hi_u;   // integer clips
};

struct LParams {
    LPlane p[kMaxPlanesL];
    int nplanes;
};

template <typename T>
struct LOps;
template <>
struct LOps<uint8_t> {
    static __device__ __forceinline__ uint8_t f(uint8_t v, const LPlane &pl) { return (uint8_t)min(max((uint32_t)v, pl.lo_u), pl.hi_u); }
};
template <>
struct LOps<uint16_t> {
    static __device__ __forceinline__ uint16_t f(uint16_t v, const LPlane &pl) { return (uint16_t)min(max((uint32_t)v, pl.lo_u), pl.hi_u); }
};
template <>
struct LOps<uint32_t> {
    static __device__ __forceinline__ uint32_t f(uint32_t v, const LPlane &pl) { return min(max(v, pl.lo_u), pl.hi_u); }
};
template <>
struct LOps<float> {
    // @max / @min return the non-NaN operand (maxnum / minnum): fmaxf / fminf
    static __device__ __forceinline__ float f(float v, const LPlane &pl) { return fminf(fmaxf(pl.lo_f, v), pl.hi_f); }
};
template <>
struct LOps<_Float16> {
    static __device__ __forceinline__ _Float16 f(_Float16 v, const LPlane &pl) { return (_Float16)fminf(fmaxf(pl.lo_f, (float)v), pl.hi_f); }
};

template <typename T>
__global__ __launch_bounds__(256) void limiter_kernel(const LParams prm) {
    constexpr int V = 16 / sizeof(T);
    typedef T VecT __attribute__((ext_vector_type(V)));
    int pi = 0;
    const int b = blockIdx.x;

    {  // block0 ascends: eight scalar steps for 192 planes (the linear scan was part of every workgroup's fixed cost)
        int hi = prm.nplanes - 1;
        while (pi < hi) {
            const int mid = (pi + hi + 1) >> 1;
            if (b >= prm.p[mid].block0)
                pi = mid;
            else
                hi = mid - 1;
        }
    }
    const LPlane pl = prm.p[pi];
    const int y0 = (b - pl.block0) * kRowsPerBlock;
    const T *src = static_cast<const T *>(pl.src);
    T *dst = static_cast<T *>(pl.dst);
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)((size_t)pl.sstride * sizeof(T)) |
                       (uintptr_t)((size_t)pl.dstride * sizeof(T))) & 15) == 0;
    for (int r = 0; r < kRowsPerBlock; ++r) {
        const int y = y0 + r;
        if (y >= pl.h) break;
        const T *s = src + (size_t)y * pl.sstride;
        T *d = dst + (size_t)y * pl.dstride;
        int x = 0;
        if (vec) {
            const int nv = pl.w / V;
            for (int i = threadIdx.x; i < nv; i += 256) {
                VecT v = VSZIP_STREAM_LOAD(reinterpret_cast<const VecT *>(s) + i);
#pragma unroll
                for (int k = 0; k < V; ++k) v[k] = LOps<T>::f(v[k], pl);
                __builtin_nontemporal_store(v, reinterpret_cast<VecT *>(d) + i);
            }
            x = nv * V;
        }
        for (int i = x + threadIdx.x; i < pl.w; i += 256) d[i] = LOps<T>::f(s[i], pl);
    }
}

template <typename T>
int run(vszip_ctx *ctx, const vszip_plane *planes, int nplanes, const double *lo, const double *hi) {
    for (int done = 0; done < nplanes;) {
        LParams prm;
        int n = 0, blocks = 0;
        for (; done + n < nplanes && n < kMaxPlanesL; ++n) {
            const vszip_plane &s = planes[done + n];
            if (!s.src || !s.dst || s.w <= 0 || s.h <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "Limiter: bad plane %d", done + n);
            if (lo[done + n] > hi[done + n]) return vszip_set_error(ctx, VSZIP_ERR_ARG, "Limiter: min value must be less than or equal to max value.");
            LPlane &d = prm.p[n];
            d.src = s.src;
            d.dst = s.dst;
            d.sstride = (int)s.src_stride;
            d.dstride = (int)s.dst_stride;
            d.w = s.w;
            d.h = s.h;
            d.block0 = blocks;
            d.lo_u = (uint32_t)std::max(0.0, lo[done + n]);
            d.hi_u = (uint32_t)std::max(0.0, hi[done + n]);
            float lf = (float)lo[done + n], hf = (float)hi[done + n];
            if (std::is_same<T, _Float16>::value) {  // the bounds are f16 values (@floatCast / comptime_float -> f16)
                lf = (float)(_Float16)lf;
                hf = (float)(_Float16)hf;
            }
            d.lo_f = lf;
            d.hi_f = hf;
            blocks += (s.h + kRowsPerBlock - 1) / kRowsPerBlock;
        }
        prm.nplanes = n;
        {
            vszip_probe_scope probe(ctx);
            hipLaunchKernelGGL((limiter_kernel<T>), dim3(blocks), dim3(256), 0, ctx->stream, prm);
        }
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += n;
    }
    return VSZIP_OK;
}

}  // namespace

VSZIP_EXPORT int vszip_limiter(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, const double *lo, const double *hi) {
    if (!ctx || !planes || !lo || !hi || nplanes <= 0) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    switch (dtype) {
        case VSZIP_U8: return run<uint8_t>(ctx, planes, nplanes, lo, hi);
        case VSZIP_U16: return run<uint16_t>(ctx, planes, nplanes, lo, hi);
        case VSZIP_U32: return run<uint32_t>(ctx, planes, nplanes, lo, hi);
        case VSZIP_F16: return run<_Float16>(ctx, planes, nplanes, lo, hi);
        case VSZIP_F32: return run<float>(ctx, planes, nplanes, lo, hi);
    }
    return vszip_set_error(ctx, VSZIP_ERR_ARG, "Limiter: not supported Int format.");
}
