// vszip.LimitFilter on gfx950: process() of src/filters/limit_filter.zig:3-34 — per pixel, the
// filtered sample is kept where |flt - ref| <= thr, the source sample is restored where it exceeds
// thr * elast, and blended in between — in plain f32 operations in the reference's order
// (-ffp-contract=off, IEEE division). Three streams in, one out; one launch per table of planes.
#include <algorithm>

#include "common.hpp"

#ifndef VSZIP_STREAM_PLAIN_LOADS
#define VSZIP_STREAM_LOAD(p) __builtin_nontemporal_load(p)  // every sample is read once
#else
#define VSZIP_STREAM_LOAD(p) (*(p))
#endif

namespace {

constexpr int kMaxPlanesLF = 192;  // planes per launch (64 YUV frames are ONE launch since round 4: four 48-plane launches paid four ramps and tails)
// Rows per workgroup, measured on 16 4K YUV420P16 frames (tools/ab_stream.sh, round 2): 1 row 0.42 of the HBM
// peak (a workgroup's fixed cost — plane lookup, two half-filled passes over a 480-vector row — dominates),
// 2 rows + non-temporal loads 0.68, 4 rows 0.67. (A pure copy gains from short-lived workgroups in address
// order, profiles/r02_membw.md; with per-workgroup set-up in the way the gain is a few percent.)
#ifndef VSZIP_STREAM_ROWS
#define VSZIP_STREAM_ROWS 2
#endif
constexpr int kRowsPerBlockLF = VSZIP_STREAM_ROWS;

struct LFPlane {
    const void *flt, *src, *ref;
    void *dst;
    int fstride, sstride, rstride, dstride, w, h;
    int block0;
    float dark_thr, bright_thr, elast;
};

struct LFParams {
    LFPlane p[kMaxPlanesLF];
    int nplanes;
};

template <typename T>
struct LFSmp {
    static constexpr bool is_int = true;
    static __device__ __forceinline__ float f(T v) { return (float)v; }
};
template <>
struct LFSmp<float> {
    static constexpr bool is_int = false;
    static __device__ __forceinline__ float f(float v) { return v; }
};
template <>
struct LFSmp<_Float16> {
    static constexpr bool is_int = false;
    static __device__ __forceinline__ float f(_Float16 v) { return (float)v; }
};

template <typename T>
__device__ __forceinline__ T limit_px(T fv, T sv, T rv, const LFPlane &pl) {
    const float sf = LFSmp<T>::f(sv), ff = LFSmp<T>::f(fv), rf = LFSmp<T>::f(rv);
    const float diff_signed = ff - rf, diff_abs = fabsf(diff_signed);
    const float thr1 = diff_signed > 0 ? pl.bright_thr : pl.dark_thr;
    const float thr2 = thr1 * pl.elast;
    float out;
    if (diff_abs <= thr1)
        out = ff;
    else if (diff_abs >= thr2)
        out = sf;
    else
        out = sf + __fdiv_rn((ff - sf) * (thr2 - diff_abs), thr2 - thr1);  // :28
    if constexpr (LFSmp<T>::is_int)
        return (T)truncf(out + 0.5f);
    else
        return (T)out;
}

template <typename T>
__global__ __launch_bounds__(256) void limit_filter_kernel(const LFParams prm) {
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
    const LFPlane pl = prm.p[pi];
    const int y0 = (b - pl.block0) * kRowsPerBlockLF;
    const T *flt = static_cast<const T *>(pl.flt), *src = static_cast<const T *>(pl.src), *ref = static_cast<const T *>(pl.ref);
    T *dst = static_cast<T *>(pl.dst);
    const bool vec = ((reinterpret_cast<uintptr_t>(flt) | reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(dst) |
                       (uintptr_t)((size_t)pl.fstride * sizeof(T)) | (uintptr_t)((size_t)pl.sstride * sizeof(T)) | (uintptr_t)((size_t)pl.rstride * sizeof(T)) |
                       (uintptr_t)((size_t)pl.dstride * sizeof(T))) & 15) == 0;
    for (int r = 0; r < kRowsPerBlockLF; ++r) {
        const int y = y0 + r;
        if (y >= pl.h) break;
        const T *f = flt + (size_t)y * pl.fstride, *s = src + (size_t)y * pl.sstride, *q = ref + (size_t)y * pl.rstride;
        T *d = dst + (size_t)y * pl.dstride;
        int x = 0;
        if (vec) {
            const int nv = pl.w / V;
            for (int i = threadIdx.x; i < nv; i += 256) {
                const VecT fv = VSZIP_STREAM_LOAD(reinterpret_cast<const VecT *>(f) + i), sv = VSZIP_STREAM_LOAD(reinterpret_cast<const VecT *>(s) + i);
                const VecT rv = q == s ? sv : VSZIP_STREAM_LOAD(reinterpret_cast<const VecT *>(q) + i);
                VecT o;
#pragma unroll
                for (int k = 0; k < V; ++k) o[k] = limit_px<T>(fv[k], sv[k], rv[k], pl);
                __builtin_nontemporal_store(o, reinterpret_cast<VecT *>(d) + i);
            }
            x = nv * V;
        }
        for (int i = x + threadIdx.x; i < pl.w; i += 256) d[i] = limit_px<T>(f[i], s[i], q[i], pl);
    }
}

template <typename T>
int run(vszip_ctx *ctx, const vszip_plane *planes, const void *const *refs, const ptrdiff_t *ref_strides, int nplanes, const float *dark, const float *bright,
        const float *elast) {
    for (int done = 0; done < nplanes;) {
        LFParams prm;
        int n = 0, blocks = 0;
        for (; done + n < nplanes && n < kMaxPlanesLF; ++n) {
            const int i = done + n;
            const vszip_plane &s = planes[i];
            if (!s.src || !s.ref || !s.dst || s.w <= 0 || s.h <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "LimitFilter: bad plane %d", i);
            LFPlane &d = prm.p[n];
            d.flt = s.src;
            d.src = s.ref;
            d.fstride = (int)s.src_stride;
            d.sstride = (int)s.ref_stride;
            const bool has_ref = refs && refs[i];
            d.ref = has_ref ? refs[i] : s.ref;
            d.rstride = has_ref ? (int)ref_strides[i] : (int)s.ref_stride;
            d.dst = s.dst;
            d.dstride = (int)s.dst_stride;
            d.w = s.w;
            d.h = s.h;
            d.block0 = blocks;
            d.dark_thr = dark[i];
            d.bright_thr = bright[i];
            d.elast = elast[i];
            blocks += (s.h + kRowsPerBlockLF - 1) / kRowsPerBlockLF;
        }
        prm.nplanes = n;
        {
            vszip_probe_scope probe(ctx);
            hipLaunchKernelGGL((limit_filter_kernel<T>), dim3(blocks), dim3(256), 0, ctx->stream, prm);
        }
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += n;
    }
    return VSZIP_OK;
}

}  // namespace

VSZIP_EXPORT int vszip_limit_filter(vszip_ctx *ctx, int dtype, const vszip_plane *planes, const void *const *refs, const ptrdiff_t *ref_strides, int nplanes,
                                    const float *dark_thr, const float *bright_thr, const float *elast) {
    if (!ctx || !planes || !dark_thr || !bright_thr || !elast || nplanes <= 0 || (refs && !ref_strides)) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    switch (dtype) {
        case VSZIP_U8: return run<uint8_t>(ctx, planes, refs, ref_strides, nplanes, dark_thr, bright_thr, elast);
        case VSZIP_U16: return run<uint16_t>(ctx, planes, refs, ref_strides, nplanes, dark_thr, bright_thr, elast);
        case VSZIP_F16: return run<_Float16>(ctx, planes, refs, ref_strides, nplanes, dark_thr, bright_thr, elast);
        case VSZIP_F32: return run<float>(ctx, planes, refs, ref_strides, nplanes, dark_thr, bright_thr, elast);
    }
    return vszip_set_error(ctx, VSZIP_ERR_ARG, "LimitFilter: not supported Int format.");
}
