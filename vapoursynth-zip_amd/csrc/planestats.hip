// vszip.PlaneAverage / vszip.PlaneMinMax on gfx950.
//
// Replaces src/filters/planeaverage.zig:26-84 (average / averageRef) and
// src/filters/planeminmax.zig:11-133 (minMaxImpl, minMaxNoThr(Ref)).
//
// One 256-thread workgroup reduces a band of rows of one plane: 16-byte row loads,
// per-thread accumulation (exact u64 for integer samples, f64 for float samples),
// wave shuffles, one partial per block; a second tiny kernel folds the partials of
// each plane in block order, so results are run-to-run reproducible.
// Thresholded PlaneMinMax needs the reference's 65 536-bin histogram (256 KiB — more
// than a CU's LDS): it is taken as a two-level radix instead — a 256-bin LDS histogram
// of the high byte, the two buckets the thresholds fall into are located on the device,
// and a second sweep histograms the low byte inside those two buckets. Integer results
// are exact; float sums differ from the reference's sequential f64 order by rounding only.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPlanesPS = 48;

struct PSPlane {
    const void *src;
    const void *ref;
    int sstride, rstride;  // elements
    int w, h;
    int block0;
    int nblocks;
};

constexpr int kMaxExclude = 256;  // distinct exclude values per call (the list travels in the kernel argument)

struct PSParams {
    PSPlane p[kMaxPlanesPS];
    int nplanes;
    int rows_per_block;
    int32_t excl[kMaxExclude];
    int nexcl;
    double *partial;     // [total_blocks][4]: avg: sum, count, diff ; minmax: min, max, diff
    uint32_t *hist;      // [nplanes][2][256]
    uint32_t *bucket;    // [nplanes][8]: lo bucket, count below it, hi bucket, count above it
    double *result;      // [nplanes][4]
    float minthr, maxthr;
    float peak;
    int hist_size;
};

template <typename T>
struct Smp;
template <>
struct Smp<uint8_t> {
    static constexpr bool is_int = true;
    using Acc = uint64_t;
    static __device__ __forceinline__ uint32_t idx(uint8_t v) { return v; }
    static __device__ __forceinline__ float f(uint8_t v) { return (float)v; }
};
template <>
struct Smp<uint16_t> {
    static constexpr bool is_int = true;
    using Acc = uint64_t;
    static __device__ __forceinline__ uint32_t idx(uint16_t v) { return v; }
    static __device__ __forceinline__ float f(uint16_t v) { return (float)v; }
};
// planeminmax.zig:26 — math.lossyCast(u16, v * 65535 + 0.5): truncating, saturating, NaN -> 0
__device__ __forceinline__ uint32_t float_bin(float v) {
    const float t = v * 65535.0f + 0.5f;
    return (t != t) ? 0u : (t <= 0.0f ? 0u : (t >= 65535.0f ? 65535u : (uint32_t)t));
}
template <>
struct Smp<uint32_t> {  // PlaneAverage only
    static constexpr bool is_int = true;
    using Acc = uint64_t;
    static __device__ __forceinline__ uint32_t idx(uint32_t v) { return v; }
    static __device__ __forceinline__ float f(uint32_t v) { return (float)v; }
};
template <>
struct Smp<float> {
    static constexpr bool is_int = false;
    using Acc = double;
    static __device__ __forceinline__ uint32_t idx(float v) { return float_bin(v); }
    static __device__ __forceinline__ float f(float v) { return v; }
};
template <>
struct Smp<_Float16> {
    static constexpr bool is_int = false;
    using Acc = double;
    static __device__ __forceinline__ uint32_t idx(_Float16 v) { return float_bin((float)v); }
    static __device__ __forceinline__ float f(_Float16 v) { return (float)v; }
};

template <typename A>
__device__ __forceinline__ A block_sum(A v, A *sh) {
    v = wave_reduce_sum(v);
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wv] = v;
    __syncthreads();
    A r = sh[0];
    for (int i = 1; i < kThreads / 64; ++i) r += sh[i];
    return r;
}

// One row through fn(v, j): 16-byte vector loads when the row(s) are 16-byte aligned (every
// VapourSynth row is), scalar loads for the tail and for unaligned rows.
template <typename T, bool REF, typename F>
__device__ __forceinline__ void row_apply(const T *s, const T *r, int w, F &&fn) {
    constexpr int V = 16 / (int)sizeof(T);
    const bool vec = ((reinterpret_cast<uintptr_t>(s) & 15) == 0) && (!REF || (reinterpret_cast<uintptr_t>(r) & 15) == 0);
    int x0 = 0;
    if (vec) {
        const int nv = w / V;
        for (int i = threadIdx.x; i < nv; i += kThreads) {
            union {
                uint4 q;
                T e[V];
            } a, b;
            a.q = reinterpret_cast<const uint4 *>(s)[i];
            if constexpr (REF) b.q = reinterpret_cast<const uint4 *>(r)[i];
#pragma unroll
            for (int k = 0; k < V; ++k) fn(a.e[k], REF ? b.e[k] : a.e[k]);
        }
        x0 = nv * V;
    }
    for (int x = x0 + (int)threadIdx.x; x < w; x += kThreads) fn(s[x], REF ? r[x] : s[x]);
}

__device__ __forceinline__ int find_plane(const PSParams &prm, int b) {
    int pi = 0;
    for (int i = 1; i < prm.nplanes; ++i)
        if (b >= prm.p[i].block0) pi = i;
    return pi;
}

// ---- PlaneAverage ---------------------------------------------------------------
// NEX: compile-time size of the exclude list (0, 1 or 8 entries; a list shorter than NEX is
// padded with copies of its first entry by the host) — the common exclude=[-1] on an integer
// clip can never match and compiles to no compare at all. NEX < 0: a longer list, walked at run
// time from the kernel argument (scalar loads).
template <typename T, bool REF, int NEX>
__global__ __launch_bounds__(kThreads) void average_kernel(const PSParams prm) {
    using S = Smp<T>;
    using Acc = typename S::Acc;
    __shared__ Acc sh[8];
    __shared__ uint32_t shc[8];
    const int b = blockIdx.x;
    const PSPlane pl = prm.p[find_plane(prm, b)];
    const int y0 = (b - pl.block0) * prm.rows_per_block;
    const int y1 = min(y0 + prm.rows_per_block, pl.h);
    const T *src = static_cast<const T *>(pl.src);
    const T *ref = static_cast<const T *>(pl.ref);
    Acc acc = 0, dacc = 0;
    uint32_t cnt = 0;
    int32_t ex[NEX > 0 ? NEX : 1];
#pragma unroll
    for (int e = 0; e < NEX; ++e) ex[e] = prm.excl[e];
    for (int y = y0; y < y1; ++y) {
        const T *s = src + (size_t)y * pl.sstride;
        const T *r = REF ? ref + (size_t)y * pl.rstride : nullptr;
        row_apply<T, REF>(s, r, pl.w, [&](T v, T j) {
            bool found = false;
#pragma unroll
            for (int e = 0; e < NEX; ++e) {
                if constexpr (S::is_int)
                    found = found || ((int32_t)v == ex[e]);
                else
                    found = found || (S::f(v) == (float)ex[e]);
            }
            if constexpr (NEX < 0) {
                for (int e = 0; e < prm.nexcl; ++e) {
                    if constexpr (S::is_int)
                        found = found || ((int32_t)v == prm.excl[e]);
                    else
                        found = found || (S::f(v) == (float)prm.excl[e]);
                }
            }
            if (!found) {
                if constexpr (S::is_int) acc += v; else acc += (double)S::f(v);
                ++cnt;
            }
            if constexpr (REF) {
                if constexpr (S::is_int) {
                    dacc += v > j ? (uint64_t)(v - j) : (uint64_t)(j - v);
                } else {
                    const T d = v > j ? (T)(v - j) : (T)(j - v);  // hz.absDiff in T
                    dacc += (double)S::f(d);
                }
            }
        });
    }
    const Acc tot = block_sum<Acc>(acc, sh);
    const Acc dtot = REF ? block_sum<Acc>(dacc, sh) : Acc(0);
    const uint32_t ctot = block_sum<uint32_t>(cnt, shc);
    if (threadIdx.x == 0) {
        double *o = prm.partial + (size_t)b * 4;
        if constexpr (S::is_int) {
            // exact integers carried bit-for-bit through the f64 slots
            reinterpret_cast<uint64_t *>(o)[0] = tot;
            reinterpret_cast<uint64_t *>(o)[2] = dtot;
        } else {
            o[0] = tot;
            o[2] = dtot;
        }
        o[1] = (double)ctot;
    }
}

template <bool IS_INT>
__global__ __launch_bounds__(64) void average_final_kernel(const PSParams prm) {
    const int pi = blockIdx.x, lane = threadIdx.x;
    const PSPlane pl = prm.p[pi];
    double *res = prm.result + (size_t)pi * 4;
    double total = 0;
    if constexpr (IS_INT) {
        unsigned long long s = 0, d = 0;
        for (int b = lane; b < pl.nblocks; b += 64) {
            const double *o = prm.partial + (size_t)(pl.block0 + b) * 4;
            s += reinterpret_cast<const unsigned long long *>(o)[0];
            d += reinterpret_cast<const unsigned long long *>(o)[2];
            total += o[1];  // pixel counts: exact in f64 in any order
        }
        s = wave_reduce_sum(s);
        d = wave_reduce_sum(d);
        total = wave_reduce_sum(total);
        if (lane != 0) return;
        // result(): planeaverage.zig:16-24
        res[0] = total == 0 ? 0.0 : (double)s / total / (double)prm.peak;
        res[1] = (double)d / (double)((uint32_t)pl.w * (uint32_t)pl.h) / (double)prm.peak;
    } else {
        double s = 0, d = 0;
        for (int b = lane; b < pl.nblocks; b += 64) {
            const double *o = prm.partial + (size_t)(pl.block0 + b) * 4;
            s += o[0];
            d += o[2];
            total += o[1];
        }
        s = wave_reduce_sum(s);  // fixed tree: reproducible
        d = wave_reduce_sum(d);
        total = wave_reduce_sum(total);
        if (lane != 0) return;
        res[0] = total == 0 ? 0.0 : s / total;
        res[1] = d / (double)((uint32_t)pl.w * (uint32_t)pl.h);
    }
}

// ---- PlaneMinMax, exact (minthr == maxthr == 0) ----------------------------------------
template <typename T, bool REF>
__global__ __launch_bounds__(kThreads) void minmax_kernel(const PSParams prm) {
    using S = Smp<T>;
    __shared__ double shd[8];
    __shared__ float shmin[8], shmax[8];
    const int b = blockIdx.x;
    const PSPlane pl = prm.p[find_plane(prm, b)];
    const int y0 = (b - pl.block0) * prm.rows_per_block;
    const int y1 = min(y0 + prm.rows_per_block, pl.h);
    const T *src = static_cast<const T *>(pl.src);
    const T *ref = static_cast<const T *>(pl.ref);
    float mn = INFINITY, mx = -INFINITY;  // u8/u16 are exact in f32
    double dacc = 0;
    for (int y = y0; y < y1; ++y) {
        const T *s = src + (size_t)y * pl.sstride;
        const T *r = REF ? ref + (size_t)y * pl.rstride : nullptr;
        row_apply<T, REF>(s, r, pl.w, [&](T sv, T rv) {
            const float v = S::f(sv);
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
            if constexpr (REF) {
                if constexpr (S::is_int)
                    dacc += fabs((double)v - (double)S::f(rv));  // planeminmax.zig:135-139
                else
                    dacc += (double)S::f((T)fabsf((float)(T)(sv - rv)));  // @abs(v - j) in T
            }
        });
    }
    for (int d = 32; d >= 1; d >>= 1) {
        mn = fminf(mn, __shfl_down(mn, d, 64));
        mx = fmaxf(mx, __shfl_down(mx, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        shmin[threadIdx.x >> 6] = mn;
        shmax[threadIdx.x >> 6] = mx;
    }
    const double dtot = REF ? block_sum<double>(dacc, shd) : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kThreads / 64; ++i) {
            mn = fminf(mn, shmin[i]);
            mx = fmaxf(mx, shmax[i]);
        }
        double *o = prm.partial + (size_t)b * 4;
        o[0] = mn;
        o[1] = mx;
        o[2] = dtot;
    }
}

__global__ __launch_bounds__(64) void minmax_final_kernel(const PSParams prm, int is_int) {
    const int pi = blockIdx.x, lane = threadIdx.x;
    const PSPlane pl = prm.p[pi];
    double mn = INFINITY, mx = -INFINITY, d = 0;
    for (int b = lane; b < pl.nblocks; b += 64) {
        const double *o = prm.partial + (size_t)(pl.block0 + b) * 4;
        mn = fmin(mn, o[0]);
        mx = fmax(mx, o[1]);
        d += o[2];
    }
    for (int k = 32; k >= 1; k >>= 1) {
        mn = fmin(mn, __shfl_down(mn, k, 64));
        mx = fmax(mx, __shfl_down(mx, k, 64));
    }
    d = wave_reduce_sum(d);
    if (lane != 0) return;
    double *res = prm.result + (size_t)pi * 4;
    const double total = (double)((uint32_t)pl.w * (uint32_t)pl.h);
    res[0] = mn;
    res[1] = mx;
    res[2] = is_int ? d / total / (double)prm.peak : d / total;
}

// ---- PlaneMinMax, thresholded: two-level radix histogram ----------------------------------
// LEVEL 0: histogram of idx >> 8 (u8: idx itself, in level 0 only), + the abs-diff sum.
// LEVEL 1: histogram of idx & 255 restricted to the low and the high threshold bucket.
template <typename T, bool REF, int LEVEL>
__global__ __launch_bounds__(kThreads) void hist_kernel(const PSParams prm) {
    using S = Smp<T>;
    __shared__ uint32_t h0[256], h1[256];
    __shared__ double shd[8];
    const int b = blockIdx.x;
    const int pi = find_plane(prm, b);
    const PSPlane pl = prm.p[pi];
    h0[threadIdx.x] = 0;
    h1[threadIdx.x] = 0;
    __syncthreads();
    const int y0 = (b - pl.block0) * prm.rows_per_block;
    const int y1 = min(y0 + prm.rows_per_block, pl.h);
    const T *src = static_cast<const T *>(pl.src);
    const T *ref = static_cast<const T *>(pl.ref);
    constexpr bool wide = sizeof(T) > 1;
    uint32_t blo = 0, bhi = 0;
    if (LEVEL == 1) {
        blo = prm.bucket[pi * 8 + 0];
        bhi = prm.bucket[pi * 8 + 2];
    }
    double dacc = 0;
    for (int y = y0; y < y1; ++y) {
        const T *s = src + (size_t)y * pl.sstride;
        const T *r = (REF && LEVEL == 0) ? ref + (size_t)y * pl.rstride : nullptr;
        if (LEVEL == 0) {
            row_apply<T, REF>(s, r, pl.w, [&](T sv, T rv) {
                const uint32_t idx = S::idx(sv);
                atomicAdd(&h0[wide ? (idx >> 8) : idx], 1u);
                if constexpr (REF) {
                    if constexpr (S::is_int)
                        dacc += fabs((double)S::f(sv) - (double)S::f(rv));
                    else
                        dacc += (double)S::f((T)fabsf((float)(T)(sv - rv)));
                }
            });
        } else {
            row_apply<T, false>(s, s, pl.w, [&](T sv, T) {
                const uint32_t idx = S::idx(sv);
                if ((idx >> 8) == blo) atomicAdd(&h0[idx & 255u], 1u);
                if ((idx >> 8) == bhi) atomicAdd(&h1[idx & 255u], 1u);
            });
        }
    }
    __syncthreads();
    uint32_t *g = prm.hist + (size_t)pi * 512;
    if (h0[threadIdx.x]) atomicAdd(&g[threadIdx.x], h0[threadIdx.x]);
    if (LEVEL == 1 && h1[threadIdx.x]) atomicAdd(&g[256 + threadIdx.x], h1[threadIdx.x]);
    if (LEVEL == 0) {
        const double dtot = REF ? block_sum<double>(dacc, shd) : 0.0;
        if (threadIdx.x == 0) prm.partial[(size_t)b * 4 + 2] = dtot;
    }
}

// After level 0: locate the buckets (planeminmax.zig:43-57: count > trunc(total * thr)
// scanning up from 0 for the minimum, down from the peak for the maximum).
// One wave per plane: the 256-bin table is scanned with a wave prefix sum (4 bins per lane) instead
// of a serial loop; "first bin whose running count exceeds the threshold" = the lowest lane/bin
// whose inclusive prefix does.
__device__ __forceinline__ void scan_bins(const uint32_t *g, int nb, bool from_top, uint32_t thr, uint32_t *bucket, uint32_t *below) {
    // lane l owns bins 4l..4l+3 in scan order (ascending, or descending from nb-1)
    const int lane = threadIdx.x;
    uint32_t v[4], loc[4];
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = lane * 4 + k;
        const int bin = from_top ? nb - 1 - idx : idx;
        v[k] = idx < nb ? g[bin] : 0u;
        run += v[k];
        loc[k] = run;
    }
    const uint32_t incl = wave_incl_scan_shfl(run);
    const uint32_t base = incl - run;
    // first scan position whose inclusive count exceeds thr
    int pos = 1 << 30;
#pragma unroll
    for (int k = 3; k >= 0; --k)
        if (lane * 4 + k < nb && base + loc[k] > thr) pos = lane * 4 + k;
    for (int d = 32; d >= 1; d >>= 1) pos = min(pos, __shfl_xor(pos, d, 64));
    const uint32_t tot = __shfl(incl, 63, 64);
    if (pos == (1 << 30)) {  // wave-uniform: no bin qualifies
        if (lane == 0) {
            *bucket = 0xffffffffu;
            *below = tot;
        }
        return;
    }
    // count of the bins before `pos` in scan order
    const int pl = pos >> 2, pk = pos & 3;
    const uint32_t bl = __shfl(base, pl, 64);
    const uint32_t l0 = __shfl(loc[0], pl, 64), l1 = __shfl(loc[1], pl, 64), l2 = __shfl(loc[2], pl, 64);
    const uint32_t before = bl + (pk == 0 ? 0u : (pk == 1 ? l0 : (pk == 2 ? l1 : l2)));
    if (lane == 0) {
        *bucket = (uint32_t)(from_top ? nb - 1 - pos : pos);
        *below = before;
    }
}

__global__ __launch_bounds__(64) void bucket_kernel(const PSParams prm, int wide) {
    const int pi = blockIdx.x;
    const PSPlane pl = prm.p[pi];
    uint32_t *g = prm.hist + (size_t)pi * 512;
    const double total = (double)((uint32_t)pl.w * (uint32_t)pl.h);
    const uint32_t totalmin = (uint32_t)trunc(total * (double)prm.minthr);
    const uint32_t totalmax = (uint32_t)trunc(total * (double)prm.maxthr);
    const int nb = wide ? (prm.hist_size >> 8) : prm.hist_size;  // <= 256
    uint32_t *bk = prm.bucket + pi * 8;
    // planeminmax.zig:43-57: count > trunc(total * thr), scanning up from 0 / down from the peak
    scan_bins(g, nb, false, totalmin, &bk[0], &bk[1]);
    scan_bins(g, nb, true, totalmax, &bk[2], &bk[3]);
    __syncthreads();
    if (wide)
        for (int u = threadIdx.x; u < 512; u += 64) g[u] = 0u;  // level 1 reuses the table
}

__global__ __launch_bounds__(64) void thr_final_kernel(const PSParams prm, int wide, int is_int) {
    const int pi = blockIdx.x;
    const PSPlane pl = prm.p[pi];
    double d = 0;
    for (int b = threadIdx.x; b < pl.nblocks; b += 64) d += prm.partial[(size_t)(pl.block0 + b) * 4 + 2];
    d = wave_reduce_sum(d);
    if (threadIdx.x != 0) return;
    const uint32_t *g = prm.hist + (size_t)pi * 512;
    const uint32_t *bk = prm.bucket + pi * 8;
    const double total = (double)((uint32_t)pl.w * (uint32_t)pl.h);
    const uint32_t totalmin = (uint32_t)trunc(total * (double)prm.minthr);
    const uint32_t totalmax = (uint32_t)trunc(total * (double)prm.maxthr);
    const uint32_t peak = (uint32_t)prm.hist_size - 1;
    uint32_t retmin = peak, retmax = 0;  // the reference's `else` values when no bin qualifies
    if (wide) {
        if (bk[0] != 0xffffffffu) {
            uint32_t count = bk[1];
            for (int u = 0; u < 256; ++u) {
                count += g[u];
                if (count > totalmin) {
                    retmin = (bk[0] << 8) | (uint32_t)u;
                    break;
                }
            }
        }
        if (bk[2] != 0xffffffffu) {
            uint32_t count = bk[3];
            for (int u = 255; u >= 0; --u) {
                count += g[256 + u];
                if (count > totalmax) {
                    retmax = (bk[2] << 8) | (uint32_t)u;
                    break;
                }
            }
        }
    } else {
        if (bk[0] != 0xffffffffu) retmin = bk[0];
        if (bk[2] != 0xffffffffu) retmax = bk[2];
    }
    double *res = prm.result + (size_t)pi * 4;
    if (is_int) {
        res[0] = retmin;
        res[1] = retmax;
        res[2] = d / total / (double)prm.peak;
    } else {
        res[0] = (double)((float)retmin / 65535.0f);  // planeminmax.zig:63-64
        res[1] = (double)((float)retmax / 65535.0f);
        res[2] = d / total;
    }
}

struct Launch {
    PSParams prm;
    int total_blocks = 0;
};

int prepare(vszip_ctx *ctx, const vszip_plane *planes, int nplanes, bool need_ref, Launch &L) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    if (nplanes > kMaxPlanesPS) return vszip_set_error(ctx, VSZIP_ERR_ARG, "at most %d planes per call", kMaxPlanesPS);
    PSParams &prm = L.prm;
    prm.nplanes = nplanes;
    long px = 0;
    for (int i = 0; i < nplanes; ++i) px += (long)planes[i].w * planes[i].h;
    int rows = 8;
    for (int i = 0; i < nplanes; ++i) {
        const vszip_plane &s = planes[i];
        if (!s.src || s.w <= 0 || s.h <= 0 || (need_ref && !s.ref)) return vszip_set_error(ctx, VSZIP_ERR_ARG, "bad plane %d", i);
    }
    prm.rows_per_block = rows;
    int blocks = 0;
    for (int i = 0; i < nplanes; ++i) {
        const vszip_plane &s = planes[i];
        PSPlane &d = prm.p[i];
        d.src = s.src;
        d.ref = s.ref;
        d.sstride = (int)s.src_stride;
        d.rstride = (int)s.ref_stride;
        d.w = s.w;
        d.h = s.h;
        d.block0 = blocks;
        d.nblocks = (s.h + rows - 1) / rows;
        blocks += d.nblocks;
    }
    L.total_blocks = blocks;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t need = (size_t)blocks * 4 * sizeof(double) + (size_t)nplanes * (512 + 8) * sizeof(uint32_t) + (size_t)nplanes * 4 * sizeof(double) + 256;
    int rc = vszip_ensure_scratch(ctx, need);
    if (rc != VSZIP_OK) return rc;
    rc = vszip_ensure_scalars(ctx, (size_t)nplanes * 4 * sizeof(double));
    if (rc != VSZIP_OK) return rc;
    char *p = static_cast<char *>(ctx->scratch);
    prm.partial = reinterpret_cast<double *>(p);
    p += (size_t)blocks * 4 * sizeof(double);
    p += (size_t)nplanes * 4 * sizeof(double);
    // the final kernels write the per-plane results straight into the pinned host buffer
    // (device-visible): the call ends with a synchronise, no copy command
    VSZIP_HIP_CHECK(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&prm.result), ctx->scalars_host, 0));
    prm.hist = reinterpret_cast<uint32_t *>(p);
    p += (size_t)nplanes * 512 * sizeof(uint32_t);
    prm.bucket = reinterpret_cast<uint32_t *>(p);
    return VSZIP_OK;
}

int fetch(vszip_ctx *ctx, const Launch &L, double *r0, double *r1, double *r2) {
    VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const double *h = static_cast<const double *>(ctx->scalars_host);
    for (int i = 0; i < L.prm.nplanes; ++i) {
        if (r0) r0[i] = h[i * 4 + 0];
        if (r1) r1[i] = h[i * 4 + 1];
        if (r2) r2[i] = h[i * 4 + 2];
    }
    return VSZIP_OK;
}

template <typename T>
int run_average(vszip_ctx *ctx, Launch &L, bool ref) {
    // exclude values an integer sample can never take are dropped (the usual exclude=[-1]);
    // the list is then padded to the kernel's compile-time size with copies of its first entry
    int n = 0;
    for (int i = 0; i < L.prm.nexcl; ++i)
        if (!Smp<T>::is_int || (L.prm.excl[i] >= 0 && L.prm.excl[i] <= (int32_t)L.prm.peak)) L.prm.excl[n++] = L.prm.excl[i];
    L.prm.nexcl = n;
    const int nex = n == 0 ? 0 : (n == 1 ? 1 : (n <= 8 ? 8 : -1));
    for (int i = n; i < nex; ++i) L.prm.excl[i] = L.prm.excl[0];
    const dim3 grid(L.total_blocks), block(kThreads);
#define VSZIP_AVG_LAUNCH(NEX)                                                                        \
    do {                                                                                             \
        if (ref)                                                                                     \
            hipLaunchKernelGGL((average_kernel<T, true, NEX>), grid, block, 0, ctx->stream, L.prm);  \
        else                                                                                         \
            hipLaunchKernelGGL((average_kernel<T, false, NEX>), grid, block, 0, ctx->stream, L.prm); \
    } while (0)
    {
        vszip_probe_scope probe(ctx);  // the plane reader (the final kernels and the copy-back are not it)
        if (nex == 0)
            VSZIP_AVG_LAUNCH(0);
        else if (nex == 1)
            VSZIP_AVG_LAUNCH(1);
        else if (nex == 8)
            VSZIP_AVG_LAUNCH(8);
        else
            VSZIP_AVG_LAUNCH(-1);
    }
#undef VSZIP_AVG_LAUNCH
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    if (Smp<T>::is_int)
        hipLaunchKernelGGL((average_final_kernel<true>), dim3(L.prm.nplanes), dim3(64), 0, ctx->stream, L.prm);
    else
        hipLaunchKernelGGL((average_final_kernel<false>), dim3(L.prm.nplanes), dim3(64), 0, ctx->stream, L.prm);
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    return VSZIP_OK;
}

template <typename T, bool REF>
int run_minmax_t(vszip_ctx *ctx, Launch &L, bool no_thr) {
    constexpr int is_int = Smp<T>::is_int ? 1 : 0;
    constexpr int wide = sizeof(T) > 1 ? 1 : 0;
    if (no_thr) {
        {
            vszip_probe_scope probe(ctx);
            hipLaunchKernelGGL((minmax_kernel<T, REF>), dim3(L.total_blocks), dim3(kThreads), 0, ctx->stream, L.prm);
        }
        hipLaunchKernelGGL(minmax_final_kernel, dim3(L.prm.nplanes), dim3(64), 0, ctx->stream, L.prm, is_int);
    } else {
        VSZIP_HIP_CHECK(ctx, hipMemsetAsync(L.prm.hist, 0, (size_t)L.prm.nplanes * 512 * sizeof(uint32_t), ctx->stream));
        {
            vszip_probe_scope probe(ctx);
            hipLaunchKernelGGL((hist_kernel<T, REF, 0>), dim3(L.total_blocks), dim3(kThreads), 0, ctx->stream, L.prm);
        }
        hipLaunchKernelGGL(bucket_kernel, dim3(L.prm.nplanes), dim3(64), 0, ctx->stream, L.prm, wide);
        if (wide) hipLaunchKernelGGL((hist_kernel<T, false, 1>), dim3(L.total_blocks), dim3(kThreads), 0, ctx->stream, L.prm);
        hipLaunchKernelGGL(thr_final_kernel, dim3(L.prm.nplanes), dim3(64), 0, ctx->stream, L.prm, wide, is_int);
    }
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    return VSZIP_OK;
}

template <typename T>
int run_minmax(vszip_ctx *ctx, Launch &L, bool ref, bool no_thr) {
    return ref ? run_minmax_t<T, true>(ctx, L, no_thr) : run_minmax_t<T, false>(ctx, L, no_thr);
}

}  // namespace

static int plane_average_batch(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, const int32_t *exclude, int nexclude, int bits_per_sample,
                               double *avg, double *diff) {
    if (nexclude < 0 || (nexclude > 0 && !exclude)) return VSZIP_ERR_ARG;
    const bool ref = planes && nplanes > 0 && planes[0].ref != nullptr;
    Launch L;
    int rc = prepare(ctx, planes, nplanes, ref, L);
    if (rc != VSZIP_OK) return rc;
    L.prm.nexcl = 0;
    for (int i = 0; i < nexclude; ++i) {  // distinct values only: membership is all the kernel tests
        bool seen = false;
        for (int k = 0; k < L.prm.nexcl; ++k) seen = seen || L.prm.excl[k] == exclude[i];
        if (seen) continue;
        if (L.prm.nexcl == kMaxExclude) return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "PlaneAverage: more than %d distinct exclude values", kMaxExclude);
        L.prm.excl[L.prm.nexcl++] = exclude[i];
    }
    L.prm.peak = (float)(((uint64_t)1 << bits_per_sample) - 1);  // planeaverage.zig(vs):115
    switch (dtype) {
        case VSZIP_U8: rc = run_average<uint8_t>(ctx, L, ref); break;
        case VSZIP_U16: rc = run_average<uint16_t>(ctx, L, ref); break;
        case VSZIP_F16: rc = run_average<_Float16>(ctx, L, ref); break;
        case VSZIP_F32: rc = run_average<float>(ctx, L, ref); break;
        case VSZIP_U32:
            if (nexclude > 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "PlaneAverage: exclude is not supported for 32-bit integer clips.");
            rc = run_average<uint32_t>(ctx, L, ref);
            break;
        default: return vszip_set_error(ctx, VSZIP_ERR_ARG, "PlaneAverage: not supported Int format.");
    }
    if (rc != VSZIP_OK) return rc;
    return fetch(ctx, L, avg, ref ? diff : nullptr, nullptr);
}

static int plane_minmax_batch(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, float minthr, float maxthr, int bits_per_sample,
                              double *vmin, double *vmax, double *diff) {
    if (minthr < 0 || minthr > 1) return vszip_set_error(ctx, VSZIP_ERR_ARG, "PlaneMinMax: minthr should be a float between 0.0 and 1.0");
    if (maxthr < 0 || maxthr > 1) return vszip_set_error(ctx, VSZIP_ERR_ARG, "PlaneMinMax: maxthr should be a float between 0.0 and 1.0");
    const bool ref = planes && nplanes > 0 && planes[0].ref != nullptr;
    Launch L;
    int rc = prepare(ctx, planes, nplanes, ref, L);
    if (rc != VSZIP_OK) return rc;
    const bool is_float = dtype == VSZIP_F16 || dtype == VSZIP_F32;
    L.prm.minthr = minthr;
    L.prm.maxthr = maxthr;
    L.prm.hist_size = is_float ? 65536 : (1 << bits_per_sample);  // planeminmax.zig(vs):147
    L.prm.peak = (float)(L.prm.hist_size - 1);
    const bool no_thr = (maxthr == 0.0f) && (minthr == 0.0f);
    switch (dtype) {
        case VSZIP_U8: rc = run_minmax<uint8_t>(ctx, L, ref, no_thr); break;
        case VSZIP_U16: rc = run_minmax<uint16_t>(ctx, L, ref, no_thr); break;
        case VSZIP_F16: rc = run_minmax<_Float16>(ctx, L, ref, no_thr); break;
        case VSZIP_F32: rc = run_minmax<float>(ctx, L, ref, no_thr); break;
        default: return vszip_set_error(ctx, VSZIP_ERR_ARG, "PlaneMinMax: not supported Int format.");
    }
    if (rc != VSZIP_OK) return rc;
    return fetch(ctx, L, vmin, vmax, ref ? diff : nullptr);
}

// Any number of planes per call: batches of kMaxPlanesPS (the per-plane table travels in the kernel argument).
VSZIP_EXPORT int vszip_plane_average(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, const int32_t *exclude, int nexclude, int bits_per_sample,
                                     double *avg, double *diff) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    for (int o = 0; o < nplanes; o += kMaxPlanesPS) {
        const int rc = plane_average_batch(ctx, dtype, planes + o, std::min(kMaxPlanesPS, nplanes - o), exclude, nexclude, bits_per_sample, avg ? avg + o : nullptr,
                                           diff ? diff + o : nullptr);
        if (rc != VSZIP_OK) return rc;
    }
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_plane_minmax(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, float minthr, float maxthr, int bits_per_sample,
                                    double *vmin, double *vmax, double *diff) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    for (int o = 0; o < nplanes; o += kMaxPlanesPS) {
        const int rc = plane_minmax_batch(ctx, dtype, planes + o, std::min(kMaxPlanesPS, nplanes - o), minthr, maxthr, bits_per_sample, vmin ? vmin + o : nullptr,
                                          vmax ? vmax + o : nullptr, diff ? diff + o : nullptr);
        if (rc != VSZIP_OK) return rc;
    }
    return VSZIP_OK;
}
