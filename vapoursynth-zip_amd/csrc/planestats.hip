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
#include <cstring>
#include <functional>
#include <type_traits>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPlanesPS = 192;  // planes per launch (the per-plane table travels in the kernel argument: 7.7 KB): 64 YUV frames are ONE launch (round 4; 48 before)
constexpr int kHistWords = 4096, kBucketWords = 16;
// thresholded PlaneMinMax, single-read sweep: the candidate ranges (see hist_sweep_kernel MODE 1). Round 6, tools/minmax_thr_timing.py (64 x 4K YUV420P16 a call,
// k frames/s on noise / the test picture / the picture x 257; two sweeps: 93 on all three): 1280 values x 4 copies (round 3's, sized for a row sample's error)
// 134 / 93 / 93; 768 x 4: 131 / 111 / 111; 512 x 4: 137 / 122 / 124; 512 x 8: 137 / 132 / 135; 256 x 8: 135 / 133 / 137 - on pictures the samples INSIDE a range
// (and their queues on equal values) are what the sweep pays for, and the previous frame's answer needs no wide range: +-256 values (one 8-bit level).
#ifndef VSZIP_MM_RANGE  // (sweeps)
#define VSZIP_MM_RANGE 512
#endif
#ifndef VSZIP_MM_RCOPIES
#define VSZIP_MM_RCOPIES 8
#endif
constexpr int kRange = VSZIP_MM_RANGE, kRangeCopies = VSZIP_MM_RCOPIES;  // values a candidate range spans (a multiple of 256), interleaved copies of its bins
static_assert(kRange % 256 == 0 && kRange >= 256, "range_scan walks 256-bin chunks");
[[maybe_unused]] constexpr int kSampleStep = 16;  // thresholded PlaneMinMax: table sizes; every 16th row is sampled

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
    uint32_t *hist;      // [nplanes][kHistWords]: [0..255] high byte (first sweep) / low byte of the low bucket (second sweep), [256..511] low byte of the
                         // high bucket, [512..] round 3: the histograms of the two candidate RANGES of the single-read path
    uint32_t *bucket;    // [nplanes][kBucketWords]: see locate_buckets
    uint32_t *shist;     // [nplanes][256]: high-byte histogram of the row SAMPLE (single-read path)
    uint32_t *pred;      // [nplanes][2] or NULL: the candidate ranges' first values for the NEXT call on the same planes, written with every result (round 6, see run_minmax_t)
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

__device__ __forceinline__ int find_plane(const PSParams &prm, int b) {  // (block0 ascends: eight scalar steps for 192 planes)
    int lo = 0, hi = prm.nplanes - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b >= prm.p[mid].block0)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

// ---- PlaneAverage ---------------------------------------------------------------
// NEX: compile-time size of the exclude list (0, 1, 2, 4 or 8 entries; a list shorter than NEX is
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
// second sweep (16-bit / float): histogram of idx & 255 restricted to the low and the high threshold bucket. kb workgroups per plane, rows
// interleaved; the workgroups of a plane the single-read path answered return at once.
constexpr int kRefineCopies = 8;  // (same-address queues: 8-bit pictures carried in 16 bits put a whole bucket into ONE low-byte bin)
template <typename T>
__global__ __launch_bounds__(kThreads) void hist_refine_kernel(const PSParams prm, int kb) {
    using S = Smp<T>;
    const int pi = blockIdx.x / kb, part = blockIdx.x - pi * kb;
    const uint32_t *bk = prm.bucket + pi * kBucketWords;
    if (!bk[4]) return;  // workgroup-uniform
    __shared__ uint32_t h0[256 * kRefineCopies], h1[256 * kRefineCopies];
    const PSPlane pl = prm.p[pi];
    for (int i = threadIdx.x; i < 256 * kRefineCopies; i += kThreads) {
        h0[i] = 0;
        h1[i] = 0;
    }
    __syncthreads();
    const uint32_t blo = bk[0], bhi = bk[2];
    const int copy = threadIdx.x & (kRefineCopies - 1);
    auto one = [&](T sv) {
        const uint32_t idx = S::idx(sv);
        if ((idx >> 8) == blo) atomicAdd(&h0[(idx & 255u) * kRefineCopies + copy], 1u);
        if ((idx >> 8) == bhi) atomicAdd(&h1[(idx & 255u) * kRefineCopies + copy], 1u);
    };
    // This workgroup's band of rows, its 16-byte vectors flattened over the rows with two loads in flight, like the first sweep (round 5: a row
    // at a time, every kb-th row, one load a thread and row in flight, it ran at 4.3 TB/s where the first sweep reaches 5.5)
    constexpr int V = 16 / (int)sizeof(T);
    const int rpb = (pl.h + kb - 1) / kb, y0 = part * rpb, nrows = min(y0 + rpb, pl.h) - y0;
    if (nrows > 0) {
        const T *src = static_cast<const T *>(pl.src) + (size_t)y0 * pl.sstride;
        const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0 && ((size_t)pl.sstride * sizeof(T)) % 16 == 0;
        const int nv = vec ? pl.w / V : 0, tid = threadIdx.x;
        if (nv > 0) {
            union Vec {
                uint4 q;
                T e[V];
            };
            const int qs = kThreads / nv, rs = kThreads - qs * nv;
            int ry = tid / nv, vx = tid - ry * nv;
            auto advance = [&]() {
                ry += qs;
                vx += rs;
                if (vx >= nv) {
                    vx -= nv;
                    ++ry;
                }
            };
            while (ry < nrows) {
                Vec a0, a1;
                a0.q = reinterpret_cast<const uint4 *>(src + (size_t)ry * pl.sstride)[vx];
                advance();
                const bool two = ry < nrows;
                if (two) {
                    a1.q = reinterpret_cast<const uint4 *>(src + (size_t)ry * pl.sstride)[vx];
                    advance();
                }
#pragma unroll
                for (int k = 0; k < V; ++k) one(a0.e[k]);
                if (two) {
#pragma unroll
                    for (int k = 0; k < V; ++k) one(a1.e[k]);
                }
            }
        }
        const int x0 = nv * V, tw = pl.w - x0;  // the columns past the last whole vector (all of them on unaligned planes)
        if (tw > 0) {
            const int total = nrows * tw;
            for (int i = tid; i < total; i += kThreads) {
                const int ry = i / tw;
                one(src[(size_t)ry * pl.sstride + x0 + i - ry * tw]);
            }
        }
    }
    __syncthreads();
    uint32_t *g = prm.hist + (size_t)pi * kHistWords;
    uint32_t c0 = 0, c1 = 0;
#pragma unroll
    for (int k = 0; k < kRefineCopies; ++k) {
        c0 += h0[threadIdx.x * kRefineCopies + k];
        c1 += h1[threadIdx.x * kRefineCopies + k];
    }
    if (c0) atomicAdd(&g[threadIdx.x], c0);
    if (c1) atomicAdd(&g[256 + threadIdx.x], c1);
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
        v[k] = idx < nb ? __hip_atomic_load(&g[bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;  // (may be called in the kernel that built the table)
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

// ---- the steps after a sweep, one wave per plane (device functions: the sweep runs them in the last workgroup to finish a plane) ----
// bucket words of a plane: [0] lo bucket, [1] count below it, [2] hi bucket, [3] count above it, [4] the histogram sweeps are needed,
// [7] sweep ticket, [8] / [9] first value of the low / high candidate range (single-read path), [10] quantile-sweep ticket,
// [11] result written, [12] samples below the low range, [13] samples below the END of the high range, [15] sample ticket.
// st[] holds words 0..4 in LDS for the wave that runs the steps.
//
// Hand-over inside a kernel (tables built by all workgroups, read by the last one) without a device-scope fence: on this chip such a
// fence writes back and invalidates the XCD's L2, and one per workgroup made the sweep 20x slower. Everything the last workgroup
// reads was written with agent-scope atomics (histogram adds, atomic stores of the partial sums), each wave waits for its own to
// complete before the workgroup barrier that precedes the ticket add, and the reads are agent-scope atomic loads. Values handed to a
// LATER kernel (bucket words, zeroed tables, results) are plain stores: the kernel boundary orders them.
// THE RULE: inside the last workgroup's steps (locate_buckets, finish_plane, write_result, range_scan, scan_bins) every read of
// hist[], bucket[] words 12 / 13 and partial[] MUST be __hip_atomic_load(..., __HIP_MEMORY_SCOPE_AGENT) — a plain load may hit a stale line
// of this XCD's L2. The words the steps WRITE and a later kernel reads (bk[0..4], bk[8], bk[9], bk[11], results) are plain.
// tests/test_gpu_planestats.py::test_handover_many_planes_many_xcds runs enough planes that every plane's last workgroup
// reads tables built on all eight XCDs, many times over, and compares every plane with the oracle.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// locate the two buckets in the high-byte (u8: the only) histogram
__device__ __forceinline__ void locate_buckets(const PSParams &prm, int pi, int wide, uint32_t *st /* LDS, 8 words */) {
    const PSPlane pl = prm.p[pi];
    uint32_t *g = prm.hist + (size_t)pi * kHistWords;
    const double total = (double)((uint32_t)pl.w * (uint32_t)pl.h);
    const uint32_t totalmin = (uint32_t)trunc(total * (double)prm.minthr);
    const uint32_t totalmax = (uint32_t)trunc(total * (double)prm.maxthr);
    const int nb = wide ? (prm.hist_size >> 8) : prm.hist_size;  // <= 256
    uint32_t *bk = prm.bucket + pi * kBucketWords;
    // planeminmax.zig:43-57: count > trunc(total * thr), scanning up from 0 / down from the peak
    scan_bins(g, nb, false, totalmin, &st[0], &st[1]);
    scan_bins(g, nb, true, totalmax, &st[2], &st[3]);
    if (threadIdx.x == 0) st[4] = wide ? 1u : 0u;
    wave_lds_fence();
    if (threadIdx.x < 5) bk[threadIdx.x] = st[threadIdx.x];
    if (wide)
        for (int u = threadIdx.x; u < 512; u += 64) g[u] = 0u;  // the second sweep reuses the table
}

__device__ __forceinline__ void write_result(const PSParams &prm, int pi, int is_int, uint32_t retmin, uint32_t retmax) {
    const PSPlane pl = prm.p[pi];
    double d = 0;
    for (int b = threadIdx.x; b < pl.nblocks; b += 64) d += __hip_atomic_load(&prm.partial[(size_t)(pl.block0 + b) * 4 + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    d = wave_reduce_sum(d);
    if (threadIdx.x != 0) return;
    const double total = (double)((uint32_t)pl.w * (uint32_t)pl.h);
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
    prm.bucket[pi * kBucketWords + 11] = 1u;
    if (prm.pred) {  // where the next frame's thresholds are looked for first: a range of kRange values centred on this frame's answers
        // (... and ending at or below 65 536: the packed sweep measures distances from a range's start modulo 2^16 - a range reaching past the top would
        // collect the plane's darkest samples, and its plane would be flagged on every call)
        prm.pred[pi * 2] = (uint32_t)min(max((int)retmin - kRange / 2, 0), 65536 - kRange);
        prm.pred[pi * 2 + 1] = (uint32_t)min(max((int)retmax - kRange / 2, 0), 65536 - kRange);
    }
}

// the plane's results from the bucket words (st[], LDS) and the low-byte histograms of the second sweep (planeminmax.zig:43-64)
__device__ __forceinline__ void finish_plane(const PSParams &prm, int pi, int wide, int is_int, const uint32_t *st, uint32_t *fine /* LDS, 4 words */) {
    const PSPlane pl = prm.p[pi];
    const uint32_t *g = prm.hist + (size_t)pi * kHistWords;
    const double total = (double)((uint32_t)pl.w * (uint32_t)pl.h);
    const uint32_t totalmin = (uint32_t)trunc(total * (double)prm.minthr);
    const uint32_t totalmax = (uint32_t)trunc(total * (double)prm.maxthr);
    const uint32_t peak = (uint32_t)prm.hist_size - 1;
    uint32_t retmin = peak, retmax = 0;  // the reference's `else` values when no bin qualifies
    if (wide) {
        // "count (starting at the bins before the bucket) + bins of the bucket > thr": the same wave scan with the threshold reduced by that start
        if (threadIdx.x < 4) fine[threadIdx.x] = 0xffffffffu;
        wave_lds_fence();
        if (st[0] != 0xffffffffu) scan_bins(g, 256, false, totalmin - st[1], &fine[0], &fine[1]);  // (wave-uniform conditions)
        if (st[2] != 0xffffffffu) scan_bins(g + 256, 256, true, totalmax - st[3], &fine[2], &fine[3]);
        wave_lds_fence();
        if (fine[0] != 0xffffffffu) retmin = (st[0] << 8) | fine[0];
        if (fine[2] != 0xffffffffu) retmax = (st[2] << 8) | fine[2];
    } else {
        if (st[0] != 0xffffffffu) retmin = st[0];
        if (st[2] != 0xffffffffu) retmax = st[2];
    }
    write_result(prm, pi, is_int, retmin, retmax);
}

__global__ __launch_bounds__(64) void thr_final_kernel(const PSParams prm, int wide, int is_int) {
    __shared__ uint32_t st[8], fine[4];
    const uint32_t *bk = prm.bucket + blockIdx.x * kBucketWords;
    if (bk[11]) return;  // a sweep's last workgroup has written this plane
    if (threadIdx.x < 8) st[threadIdx.x] = bk[threadIdx.x];
    wave_lds_fence();
    finish_plane(prm, blockIdx.x, wide, is_int, st, fine);
}

// ---- round 3: ONE read of the plane for 16-bit / float clips ------------------------------------------------------------------
// The two-level radix reads every plane twice: the low-byte histograms it needs are those of the two buckets the thresholds fall
// into, known only after a full sweep. A row SAMPLE (every 16th row: 6 % of the bytes) predicts them instead: its high-byte
// histogram is scanned like the real one (thresholds scaled to the sample), and each threshold gets a candidate RANGE of values, the
// predicted bucket and two on either side (5 x 256 values). The single sweep then needs no full histogram at all — per plane
//   * how many samples lie below the low range, and how many below the end of the high range (two compares and adds per sample,
//     no LDS atomic: the plain histogram sweep is bound by its one ds_add per sample),
//   * the exact histograms of the two ranges (LDS atomics for the few samples inside them),
// and the last workgroup to finish the plane finds the two answers inside the ranges by the reference's own rule (the running count
// starts at the samples below the range). When an answer lies outside its range — the sampled quantile was more than two 8-bit
// levels off — the plane is flagged and goes through the two histogram sweeps; their workgroups return at once for every other plane.
// Results are identical either way (tests/test_gpu_planestats.py::test_minmax_single_read_and_its_second_sweep forces both).
// LDS histogram adds that survive real pictures: neighbouring samples of a picture share their high byte, so 64 lanes x 8 samples
// would queue on one or two LDS words (measured: the histogram sweep took 3x as long on the test picture as on noise). The table is
// kept in kCopies interleaved copies (lane & 15: sixteen neighbouring banks), which divides the queue by sixteen. (Merging the runs
// of equal bins inside a lane first was tried and is worse: at ~16 VALU instructions per sample a sweep runs at the memory rate, and
// the merge alone costs 8.)
constexpr int kCopies = 16;
static_assert(512 + 2 * kRange <= kHistWords, "range tables");

#ifdef VSZIP_DEV_VARIANTS
constexpr int kSampleBlocks = 32;  // workgroups per plane in the sample pass
template <typename T>
__global__ __launch_bounds__(kThreads) void hist_sample_kernel(const PSParams prm) {
    using S = Smp<T>;
    constexpr int V = 16 / (int)sizeof(T);
    __shared__ uint32_t h[256 * kCopies];
    __shared__ uint32_t tmp[4];
    __shared__ int last;
    const int pi = blockIdx.x / kSampleBlocks, part = blockIdx.x % kSampleBlocks;
    const PSPlane pl = prm.p[pi];
    for (int i = threadIdx.x; i < 256 * kCopies; i += kThreads) h[i] = 0;
    __syncthreads();
    const T *src = static_cast<const T *>(pl.src);
    const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0 && ((size_t)pl.sstride * sizeof(T)) % 16 == 0;
    const int nv = vec ? pl.w / V : 0;
    const int copy = threadIdx.x & (kCopies - 1);
    for (int y = kSampleStep / 2 + kSampleStep * part; y < pl.h; y += kSampleStep * kSampleBlocks) {
        const T *s = src + (size_t)y * pl.sstride;
        for (int i = threadIdx.x; i < nv; i += kThreads) {
            union {
                uint4 q;
                T e[V];
            } a;
            a.q = reinterpret_cast<const uint4 *>(s)[i];
#pragma unroll
            for (int k = 0; k < V; ++k) atomicAdd(&h[(S::idx(a.e[k]) >> 8) * kCopies + copy], 1u);
        }
        for (int x = nv * V + (int)threadIdx.x; x < pl.w; x += kThreads) atomicAdd(&h[(S::idx(s[x]) >> 8) * kCopies + copy], 1u);
    }
    __syncthreads();
    uint32_t *g = prm.shist + (size_t)pi * 256;
    uint32_t *bk = prm.bucket + pi * kBucketWords;
    {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < kCopies; ++k) c += h[threadIdx.x * kCopies + k];
        if (c) atomicAdd(&g[threadIdx.x], c);
    }
    // the plane's last workgroup turns the sample histogram into the candidate ranges (no fence: see "Hand-over inside a kernel")
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&bk[15], 1u) == (uint32_t)kSampleBlocks - 1;
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    const int rows = pl.h > kSampleStep / 2 ? (pl.h - kSampleStep / 2 - 1) / kSampleStep + 1 : 0;
    const double total = (double)((uint32_t)pl.w * (uint32_t)rows);
    const int nb = prm.hist_size >> 8;
    scan_bins(g, nb, false, (uint32_t)trunc(total * (double)prm.minthr), &tmp[0], &tmp[1]);
    scan_bins(g, nb, true, (uint32_t)trunc(total * (double)prm.maxthr), &tmp[2], &tmp[3]);
    wave_lds_fence();
    if (threadIdx.x == 0) {
        // Nothing in the sample exceeds a threshold (thr = 1, or a plane too short to be sampled): an empty range that everything lies below
        // (low side) / whose end everything lies at or above (high side); the sweep's last workgroup then either knows that no value
        // qualifies or flags the plane.
        bk[8] = tmp[0] == 0xffffffffu ? 0x10000u : (uint32_t)max(((int)tmp[0] << 8) + 128 - kRange / 2, 0);
        bk[9] = tmp[2] == 0xffffffffu ? (uint32_t)-kRange : (uint32_t)max(((int)tmp[2] << 8) + 128 - kRange / 2, 0);
    }
}

#endif  // VSZIP_DEV_VARIANTS

// One threshold of the single-read path in the sweep's last workgroup: the reference's scan with the running count started at `start`
// (the samples before the range in scan order), over the range's kRange bins in chunks of 256. found: 0 = the answer lies outside.
__device__ __forceinline__ uint32_t range_scan(const uint32_t *tab, bool from_top, uint32_t start, uint32_t thr, uint32_t *fine, int *found) {
    uint32_t run = start, ans = 0;
    *found = 0;
    if (start > thr) return 0;  // the count already exceeds the threshold before the range: the answer lies before it
    for (int c = 0; c < kRange / 256; ++c) {
        const int chunk = from_top ? kRange / 256 - 1 - c : c;
        wave_lds_fence();
        scan_bins(tab + chunk * 256, 256, from_top, thr - run, &fine[0], &fine[1]);
        wave_lds_fence();
        if (fine[0] != 0xffffffffu) {
            ans = (uint32_t)chunk * 256u + fine[0];
            *found = 1;
            break;
        }
        run += fine[1];  // the chunk's total
    }
    return ans;
}

// Round 6, the packed form of the single-read sweep for 16-bit integer planes (VSZIP_MM_NO_PACKED restores the per-sample form for sweeps).
// The per-sample form spends ~10 VALU instructions a sample (two differences, two range tests, two sign bits, their sums) and its time does not
// depend on the content: it is issue bound. Here a DWORD (two samples) goes through
//   t = v_pk_sub_u16(p, start x 0x10001)          the samples' distances from a range's start, modulo 2^16: in the range <=> t < kRange
//   sum_t = v_sad_u16(t, 0, sum_t), sum_p likewise  |t - 0| + |t' - 0| + sum: one instruction adds two 16-bit halves to a 32-bit sum
// and "samples below the start" is not counted at all: the sum of t over n samples is sum(p) - n start + 65536 x (samples below the start), exactly.
// The test reads a half in place (v_cmp_gt_u16 with an SDWA half select) and the rare add computes its address with one v_mad_u16 (half select through
// op_sel): 4.5 instructions a sample in front of the adds.
typedef unsigned short mm_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mm_pk_sub_u16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(mm_us2, a) - __builtin_bit_cast(mm_us2, b)); }
// one half of t against the range, and the add where it lies inside. `addr` is a register whose upper half stays zero (v_mad_u16 writes the lower).
template <int HALF, int OFF>
__device__ __forceinline__ void mm_range_add(uint32_t t, uint32_t range, uint32_t copy4, uint32_t &addr, uint32_t one) {
    static_assert(kRange * kRangeCopies * 4 <= 65536, "v_mad_u16 computes the byte offset inside a range's table");
    uint64_t m, sv;
    if constexpr (HALF == 0)
        asm volatile(
            "v_cmp_gt_u16_sdwa %[m], %[rng], %[t] src0_sel:DWORD src1_sel:WORD_0\n\t"
            "s_and_saveexec_b64 %[sv], %[m]\n\t"
            "s_cbranch_execz .Lmm_skip_%=\n\t"
            "v_mad_u16 %[a], %[t], %[mul], %[c4] op_sel:[0,0,0,0]\n\t"
            "ds_add_u32 %[a], %[one] offset:%[off]\n"
            ".Lmm_skip_%=:\n\t"
            "s_or_b64 exec, exec, %[sv]"
            : [m] "=&s"(m), [sv] "=&s"(sv), [a] "+v"(addr)
            : [rng] "s"(range), [t] "v"(t), [c4] "v"(copy4), [one] "v"(one), [mul] "i"(kRangeCopies * 4), [off] "i"(OFF)
            : "memory");
    else
        asm volatile(
            "v_cmp_gt_u16_sdwa %[m], %[rng], %[t] src0_sel:DWORD src1_sel:WORD_1\n\t"
            "s_and_saveexec_b64 %[sv], %[m]\n\t"
            "s_cbranch_execz .Lmm_skip_%=\n\t"
            "v_mad_u16 %[a], %[t], %[mul], %[c4] op_sel:[1,0,0,0]\n\t"
            "ds_add_u32 %[a], %[one] offset:%[off]\n"
            ".Lmm_skip_%=:\n\t"
            "s_or_b64 exec, exec, %[sv]"
            : [m] "=&s"(m), [sv] "=&s"(sv), [a] "+v"(addr)
            : [rng] "s"(range), [t] "v"(t), [c4] "v"(copy4), [one] "v"(one), [mul] "i"(kRangeCopies * 4), [off] "i"(OFF)
            : "memory");
}

// The sweeps. MODE 0: histogram of idx >> 8 (u8: idx itself — one level is all it needs) for the two-level radix; with `flagged` only
// the planes the single-read path could not answer. MODE 1: the single-read path's counts and range histograms. Both: the abs-diff sum.
// Persistent workgroups of 512 threads, each over a contiguous range of the call's 8-row units (flushing its tables where the range
// crosses into the next plane): the tables reach global memory once per workgroup and plane, not once per unit. The last workgroup to
// finish a plane runs the steps that follow the sweep.
constexpr int kSweepThreads = 512;  // 2 waves per SIMD and workgroup; the plain variants hold 3 workgroups per CU, the ones with a reference clip 2
template <typename T, bool REF, int MODE>
__global__ __launch_bounds__(kSweepThreads) void hist_sweep_kernel(const PSParams prm, int units_total, int flagged) {
    using S = Smp<T>;
    constexpr int V = 16 / (int)sizeof(T);
    constexpr bool wide = sizeof(T) > 1;
    constexpr int kTab = MODE == 0 ? 256 * kCopies : 2 * kRange * kRangeCopies;
#ifdef VSZIP_MM_NO_PACKED
    constexpr bool kPacked = false;
#else
    constexpr bool kPacked = MODE == 1 && !REF && std::is_same<T, uint16_t>::value;  // see mm_range_add
#endif
    __shared__ uint32_t tab[kTab];  // MODE 0: the histogram's copies; MODE 1: the low range's bins, then the high range's (kRangeCopies interleaved copies each)
    __shared__ double shd[kSweepThreads / 64];
    __shared__ uint32_t shc[2][kSweepThreads / 64];
    __shared__ uint32_t st[8], fine[4];
    __shared__ int last;
    const int tid = threadIdx.x, copy = tid & (kCopies - 1), rcopy = tid & (kRangeCopies - 1);
    const int u1 = (int)((long)(blockIdx.x + 1) * units_total / gridDim.x);
    int u = (int)((long)blockIdx.x * units_total / gridDim.x);
    while (u < u1) {  // one turn per plane the range touches (workgroup-uniform)
        const int pi = find_plane(prm, u);
        const PSPlane pl = prm.p[pi];
        const int ue = min(u1, pl.block0 + pl.nblocks);
        uint32_t *bk = prm.bucket + pi * kBucketWords;
        if (MODE == 0 && flagged && !bk[4]) {
            u = ue;
            continue;
        }
        for (int i = tid; i < kTab; i += kSweepThreads) tab[i] = 0;
        __syncthreads();
        // MODE 1: where the two ranges start - the previous call's answers (PSParams::pred; the plane's last workgroup replaces them only after every
        // workgroup of the plane has handed in its ticket, i.e. has read them) or, in the dev build's sampled variant, what the sample pass left in the bucket
        const uint32_t lo_start = MODE == 1 ? (prm.pred ? prm.pred[pi * 2] : bk[8]) : 0u, hi_start = MODE == 1 ? (prm.pred ? prm.pred[pi * 2 + 1] : bk[9]) : 0u;
        const int y0 = (u - pl.block0) * prm.rows_per_block;
        const int nrows = min((ue - pl.block0) * prm.rows_per_block, pl.h) - y0;
        const T *src = static_cast<const T *>(pl.src) + (size_t)y0 * pl.sstride;
        const T *ref = REF ? static_cast<const T *>(pl.ref) + (size_t)y0 * pl.rstride : nullptr;
        const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0 && ((size_t)pl.sstride * sizeof(T)) % 16 == 0 &&
                         (!REF || ((reinterpret_cast<uintptr_t>(ref) & 15) == 0 && ((size_t)pl.rstride * sizeof(T)) % 16 == 0));
        const int nv = vec ? pl.w / V : 0;
        double dacc = 0;
        // MODE 1: samples below the low range / below the END of the high range (kPacked: below its START - the flush adds the samples inside)
        uint32_t below_lo = 0, below_hi_end = 0;
        auto one = [&](T sv, T rv) {
            const uint32_t idx = S::idx(sv);
            if constexpr (MODE == 0) {
                atomicAdd(&tab[(wide ? idx >> 8 : idx) * kCopies + copy], 1u);
            } else {
                // idx and the range starts are below 2^17: the sign bit of the difference is the borrow
                const uint32_t dl = idx - lo_start, eh = idx - hi_start - (uint32_t)kRange;
                below_lo += dl >> 31;
                below_hi_end += (kPacked ? idx - hi_start : eh) >> 31;
                // (round 6, measured and not kept: ONE unconditional add a sample, out-of-range samples into a word of the thread's own - 80 k frames/s
                // against 92-134 k: the add itself is what costs, and the branches skip it for most samples)
                if (dl < (uint32_t)kRange) atomicAdd(&tab[dl * kRangeCopies + rcopy], 1u);
                if (eh + (uint32_t)kRange < (uint32_t)kRange) atomicAdd(&tab[(eh + 2u * kRange) * kRangeCopies + rcopy], 1u);
            }
            if constexpr (REF) {
                if constexpr (S::is_int)
                    dacc += fabs((double)S::f(sv) - (double)S::f(rv));
                else
                    dacc += (double)S::f((T)fabsf((float)(T)(sv - rv)));
            }
        };
        if (nv > 0) {
            // the range's 16-byte vectors, rows flattened (a 4K row has 480 of them, fewer than the workgroup has lanes); a lane's next vector
            // lies kSweepThreads further: (ry, vx) advance by a quotient and a remainder instead of a division per step. Two loads in flight.
            union Vec {
                uint4 q;
                T e[V];
            };
            const int qs = kSweepThreads / nv, rs = kSweepThreads - qs * nv;
            int ry = tid / nv, vx = tid - ry * nv;
            auto advance = [&]() {
                ry += qs;
                vx += rs;
                if (vx >= nv) {
                    vx -= nv;
                    ++ry;
                }
            };
            auto fetch = [&](Vec &a, Vec &b) {
                a.q = reinterpret_cast<const uint4 *>(src + (size_t)ry * pl.sstride)[vx];
                if constexpr (REF) b.q = reinterpret_cast<const uint4 *>(ref + (size_t)ry * pl.rstride)[vx];
            };
            auto tally = [&](const Vec &a, const Vec &b) {
#pragma unroll
                for (int k = 0; k < V; ++k) one(a.e[k], REF ? b.e[k] : a.e[k]);
            };
            // The loop over a lane's vectors (not workgroup-uniform: no barriers inside): a pair is tallied while the NEXT pair's loads are in flight - four
            // 16-byte loads a lane instead of two (24 waves a CU x 64 lanes x 32 bytes in flight were fewer than the memory system's latency x rate).
            auto pipelined = [&](auto &&tl, auto &&after) {
#ifdef VSZIP_MM_NO_PIPE  // (sweeps: load a pair, tally it)
                constexpr bool kPipe = false;
#else
                constexpr bool kPipe = !REF;  // (with a reference clip twice the registers: one workgroup a CU less)
#endif
                if constexpr (!kPipe) {
                while (ry < nrows) {
                    Vec a0, b0, a1, b1;
                    fetch(a0, b0);
                    advance();
                    const bool two = ry < nrows;
                    if (two) {
                        fetch(a1, b1);
                        advance();
                    }
                    tl(a0, b0);
                    if (two) tl(a1, b1);
                    after();
                }
                } else {
                // Every load is issued whether or not its vector exists (past the end it re-reads the range's last row): conditional loads have no static
                // count, and the compiler then waits for ALL of them (vmcnt(0)) in front of the tally.
                Vec a0, b0, a1, b1, n0, m0, n1, m1;
                auto fetch_any = [&](Vec &a, Vec &b) {
                    const int r = min(ry, nrows - 1);
                    a.q = reinterpret_cast<const uint4 *>(src + (size_t)r * pl.sstride)[vx];
                    if constexpr (REF) b.q = reinterpret_cast<const uint4 *>(ref + (size_t)r * pl.rstride)[vx];
                };
#ifdef VSZIP_MM_FETCH4  // (sweeps: four loads, then their tallies)
                while (ry < nrows) {
                    const bool h0 = true;
                    fetch_any(a0, b0);
                    advance();
                    const bool h1 = ry < nrows;
                    fetch_any(a1, b1);
                    advance();
                    const bool g0 = ry < nrows;
                    fetch_any(n0, m0);
                    advance();
                    const bool g1 = ry < nrows;
                    fetch_any(n1, m1);
                    advance();
                    if (h0) tl(a0, b0);
                    if (h1) tl(a1, b1);
                    if (g0) tl(n0, m0);
                    if (g1) tl(n1, m1);
                    after();
                }
#else
                bool h0 = ry < nrows;
                fetch_any(a0, b0);
                advance();
                bool h1 = ry < nrows;
                fetch_any(a1, b1);
                advance();
                while (h0) {  // (two turns a trip, the register sets changing roles: a copy at the end of a turn would wait for the loads it copies)
                    const bool g0 = ry < nrows;
                    fetch_any(n0, m0);
                    advance();
                    const bool g1 = ry < nrows;
                    fetch_any(n1, m1);
                    advance();
                    tl(a0, b0);
                    if (h1) tl(a1, b1);
                    after();
                    if (!g0) break;
                    h0 = ry < nrows;
                    fetch_any(a0, b0);
                    advance();
                    h1 = ry < nrows;
                    fetch_any(a1, b1);
                    advance();
                    tl(n0, m0);
                    if (g1) tl(n1, m1);
                    after();
                }
#endif
                }
            };
            // ranges that end past 65 536 (the dev build's row sample leaves such starts, and its "nothing qualifies" markers) take the per-sample form: modulo 2^16
            // the plane's darkest samples would fall into them
            const bool packed = kPacked && lo_start <= 65536u - kRange && hi_start <= 65536u - kRange;
            if (packed) {
                const uint32_t lo2 = lo_start * 0x10001u, hi2 = hi_start * 0x10001u, copy4 = (uint32_t)rcopy * 4u, one1 = 1u;
                uint32_t sum_p = 0, sum_l = 0, sum_h = 0, nvec = 0, addr = 0;
                auto fold = [&]() {  // 32-bit sums hold 65 536 samples of 16 bits
                    const uint64_t n = (uint64_t)nvec * V;
                    below_lo += (uint32_t)(((uint64_t)sum_l + n * lo_start - sum_p) >> 16);
                    below_hi_end += (uint32_t)(((uint64_t)sum_h + n * hi_start - sum_p) >> 16);
                    sum_p = sum_l = sum_h = nvec = 0;
                };
                auto tally2 = [&](uint32_t pd) {
                    const uint32_t tl = mm_pk_sub_u16(pd, lo2), th = mm_pk_sub_u16(pd, hi2);
                    sum_p = __builtin_amdgcn_sad_u16(pd, 0u, sum_p);
                    sum_l = __builtin_amdgcn_sad_u16(tl, 0u, sum_l);
                    sum_h = __builtin_amdgcn_sad_u16(th, 0u, sum_h);
                    mm_range_add<0, 0>(tl, (uint32_t)kRange, copy4, addr, one1);
                    mm_range_add<0, kRange * kRangeCopies * 4>(th, (uint32_t)kRange, copy4, addr, one1);
                    mm_range_add<1, 0>(tl, (uint32_t)kRange, copy4, addr, one1);
                    mm_range_add<1, kRange * kRangeCopies * 4>(th, (uint32_t)kRange, copy4, addr, one1);
                };
                auto tally4 = [&](const uint4 &q) {
                    tally2(q.x);
                    tally2(q.y);
                    tally2(q.z);
                    tally2(q.w);
                    ++nvec;
                };
                pipelined([&](const Vec &a, const Vec &) { tally4(a.q); }, [&]() {
                    if (nvec >= 8000u) fold();
                });
                fold();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the adds above are invisible to the compiler's counters
            } else
                pipelined(tally, []() {});
        }
        const int x0 = nv * V, tw = pl.w - x0;  // the columns past the last whole vector (all of them on unaligned planes)
        if (tw > 0) {
            const int total = nrows * tw;
            for (int i = tid; i < total; i += kSweepThreads) {
                const int ry = i / tw, x = x0 + i - ry * tw;
                one(src[(size_t)ry * pl.sstride + x], REF ? ref[(size_t)ry * pl.rstride + x] : T(0));
            }
        }
        __syncthreads();
        uint32_t *g = prm.hist + (size_t)pi * kHistWords;
        if constexpr (MODE == 0) {
            if (tid < 256) {
                uint32_t c = 0;
#pragma unroll
                for (int k = 0; k < kCopies; ++k) c += tab[tid * kCopies + k];
                if (c) atomicAdd(&g[tid], c);
            }
        } else {
            for (int i = tid; i < 2 * kRange; i += kSweepThreads) {
                uint32_t c = 0;
#pragma unroll
                for (int k = 0; k < kRangeCopies; ++k) c += tab[i * kRangeCopies + k];
                if (c) atomicAdd(&g[512 + i], c);
                if (kPacked && i >= kRange) below_hi_end += c;  // below the high range's start + inside it = below its end
            }
            const uint32_t wl = wave_reduce_sum(below_lo), wh = wave_reduce_sum(below_hi_end);
            if ((tid & 63) == 0) {
                shc[0][tid >> 6] = wl;
                shc[1][tid >> 6] = wh;
            }
        }
        // the abs-diff sum of the range goes into its first unit's slot, the other units' slots are zero
        if constexpr (REF) {
            const double w = wave_reduce_sum(dacc);
            if ((tid & 63) == 0) shd[tid >> 6] = w;
        }
        __syncthreads();
        if (tid == 0) {
            if constexpr (MODE == 1) {
                uint32_t cl = 0, ch = 0;
                for (int i = 0; i < kSweepThreads / 64; ++i) {
                    cl += shc[0][i];
                    ch += shc[1][i];
                }
                if (cl) atomicAdd(&bk[12], cl);
                if (ch) atomicAdd(&bk[13], ch);
            }
        }
        if (!(MODE == 0 && flagged)) {  // (the flagged planes' sums were written by the single sweep)
            if (tid == 0) {
                double dtot = 0;
                if constexpr (REF)
                    for (int i = 0; i < kSweepThreads / 64; ++i) dtot += shd[i];
                __hip_atomic_store(&prm.partial[(size_t)u * 4 + 2], dtot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int k = u + 1 + tid; k < ue; k += kSweepThreads) __hip_atomic_store(&prm.partial[(size_t)k * 4 + 2], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ticket: units of the plane done so far (no fence: see "Hand-over inside a kernel")
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) last = atomicAdd(&bk[MODE == 0 ? 7 : 10], (uint32_t)(ue - u)) + (uint32_t)(ue - u) == (uint32_t)pl.nblocks;
        __syncthreads();
        if (last && tid < 64) {
            if constexpr (MODE == 0) {
                locate_buckets(prm, pi, wide ? 1 : 0, st);
                if (!wide) finish_plane(prm, pi, 0, S::is_int ? 1 : 0, st, fine);
            } else {
                const uint32_t total = (uint32_t)pl.w * (uint32_t)pl.h;
                const uint32_t totalmin = (uint32_t)trunc((double)total * (double)prm.minthr);
                const uint32_t totalmax = (uint32_t)trunc((double)total * (double)prm.maxthr);
                const uint32_t nlo = __hip_atomic_load(&bk[12], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t nhi = total - __hip_atomic_load(&bk[13], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // samples at or above the high range's end
                const uint32_t peak = (uint32_t)prm.hist_size - 1;
                uint32_t retmin = peak, retmax = 0;  // the reference's `else` values: the count never exceeds a threshold >= total
                int okl = 1, okh = 1;
                if (totalmin < total) retmin = lo_start + range_scan(g + 512, false, nlo, totalmin, fine, &okl);
                if (totalmax < total) retmax = hi_start + range_scan(g + 512 + kRange, true, nhi, totalmax, fine, &okh);
                if (okl && okh)
                    write_result(prm, pi, S::is_int ? 1 : 0, retmin, retmax);
                else if (tid == 0) {
                    bk[4] = 1u;  // the two histogram sweeps answer this plane
                    prm.result[(size_t)pi * 4 + 3] = 1.0;  // (the result's spare word: a caller that defers those sweeps looks here after its synchronise)
                }
            }
        }
        __syncthreads();
        u = ue;
    }
}

struct Launch {
    PSParams prm;
    int total_blocks = 0;
};

int prepare(vszip_ctx *ctx, const vszip_plane *planes, int nplanes, bool need_ref, Launch &L, double *result_dev) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    if (nplanes > kMaxPlanesPS) return vszip_set_error(ctx, VSZIP_ERR_ARG, "at most %d planes per call", kMaxPlanesPS);
    PSParams &prm = L.prm;
    prm.nplanes = nplanes;
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
    const size_t need = (size_t)blocks * 4 * sizeof(double) + (size_t)nplanes * (kHistWords + kBucketWords + 256) * sizeof(uint32_t) + (size_t)nplanes * 4 * sizeof(double) + 256;
    int rc = vszip_ensure_scratch(ctx, need);
    if (rc != VSZIP_OK) return rc;
    char *p = static_cast<char *>(ctx->scratch);
    prm.partial = reinterpret_cast<double *>(p);
    p += (size_t)blocks * 4 * sizeof(double);
    p += (size_t)nplanes * 4 * sizeof(double);
    // the final kernels write the per-plane results ([plane][4] doubles) straight into pinned host memory (device-visible):
    // the context's own buffer (the calls that return values: one synchronise at the end, no copy command) or the caller's
    // (the _async calls: no synchronise at all)
    prm.result = result_dev;
    prm.hist = reinterpret_cast<uint32_t *>(p);
    p += (size_t)nplanes * kHistWords * sizeof(uint32_t);
    prm.bucket = reinterpret_cast<uint32_t *>(p);
    p += (size_t)nplanes * kBucketWords * sizeof(uint32_t);
    prm.shist = reinterpret_cast<uint32_t *>(p);
    prm.pred = nullptr;
    return VSZIP_OK;
}

// the context's pinned result buffer for `nplanes` planes: host and device views
int result_buffer(vszip_ctx *ctx, int nplanes, double **host, double **dev) {
    const int rc = vszip_ensure_scalars(ctx, (size_t)nplanes * 4 * sizeof(double));
    if (rc != VSZIP_OK) return rc;
    *host = static_cast<double *>(ctx->scalars_host);
    VSZIP_HIP_CHECK(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(dev), ctx->scalars_host, 0));
    return VSZIP_OK;
}
// a caller's pinned array as the kernels see it
int caller_buffer(vszip_ctx *ctx, double *pinned, double **dev) {
    if (!pinned) return vszip_set_error(ctx, VSZIP_ERR_ARG, "the result array is NULL");
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(dev), pinned, 0) != hipSuccess) {
        (void)hipGetLastError();
        return vszip_set_error(ctx, VSZIP_ERR_ARG, "the result array must be pinned host memory (vszip_host_alloc_pinned)");
    }
    return VSZIP_OK;
}
int fetch(vszip_ctx *ctx, int nplanes, const double *h, double *r0, double *r1, double *r2) {
    for (int i = 0; i < nplanes; ++i) {
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
    const int nex = n == 0 ? 0 : (n == 1 ? 1 : (n == 2 ? 2 : (n <= 4 ? 4 : (n <= 8 ? 8 : -1))));  // (2 and 4 since round 4: exclude=[16, 235] paid eight compares a sample)
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
        else if (nex == 2)
            VSZIP_AVG_LAUNCH(2);
        else if (nex == 4)
            VSZIP_AVG_LAUNCH(4);
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

// Thresholded calls on 16-bit / float planes, round 6: TEMPORAL prediction. The two-level radix reads every plane twice because the buckets the
// thresholds fall into are known only after a full sweep; round 3's single-read sweep (MODE 1: exact histograms of two candidate ranges of
// 5 x 256 values + the counts below them, flagged planes fall back to the two sweeps) needed a row-sample pass to predict the ranges and lost
// on pictures. A clip's next frame is the better predictor: every result leaves "a range of kRange = 512 values centred on its answer" in a per-context
// table (PSParams::pred), and a call of the same shape as the previous one (plane count and sizes, sample type, thresholds) sweeps ONCE over
// those ranges. A scene cut, another clip or a first call cost what they always did — the flagged planes' two sweeps — and the results are the
// reference's either way (tests/test_gpu_planestats.py::test_minmax_temporal_prediction*). VSZIP_MINMAX_NO_PREDICT=1: always two sweeps.
constexpr int kPredSlots = 4;
static uint64_t minmax_signature(const Launch &L, int dtype_size, bool ref) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix((uint64_t)L.prm.nplanes);
    mix((uint64_t)dtype_size);
    mix(ref ? 2 : 1);
    mix((uint64_t)L.prm.hist_size);
    uint32_t a, b;
    std::memcpy(&a, &L.prm.minthr, 4);
    std::memcpy(&b, &L.prm.maxthr, 4);
    mix(((uint64_t)a << 32) | b);
    for (int i = 0; i < L.prm.nplanes; ++i) mix(((uint64_t)(uint32_t)L.prm.p[i].w << 32) | (uint32_t)L.prm.p[i].h);
    return h ? h : 1;
}

template <typename T, bool REF>
int run_minmax_t(vszip_ctx *ctx, Launch &L, bool no_thr, int batch, std::function<int()> *deferred) {
    constexpr int is_int = Smp<T>::is_int ? 1 : 0;
    constexpr int wide = sizeof(T) > 1 ? 1 : 0;
    bool predicted = false;
    if (wide && !no_thr && !ctx->opt.minmax_no_predict && batch >= 0) {
        // kPredSlots tables per plane group, by signature, least recently used replaced: the thresholded statistics of several clips (or one clip under
        // several thresholds) of a filter graph arrive interleaved on one context, and each keeps its own predictions
        const size_t group = (size_t)batch * kPredSlots, need = (group + kPredSlots) * kMaxPlanesPS;
        if (ctx->minmax_pred_planes < need) {  // grow-only, old predictions kept (a group's tables do not move when later groups are added)
            const size_t cap = std::max<size_t>(need, 2 * ctx->minmax_pred_planes);
            void *np = nullptr;
            VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (vszip_hip_malloc(ctx, &np, cap * 2 * sizeof(uint32_t)) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "PlaneMinMax: prediction table allocation failed");
            if (ctx->minmax_pred) {
                VSZIP_HIP_CHECK(ctx, hipMemcpy(np, ctx->minmax_pred, ctx->minmax_pred_planes * 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice));
                (void)hipFree(ctx->minmax_pred);
            }
            ctx->minmax_pred = np;
            ctx->minmax_pred_planes = cap;
        }
        if (ctx->minmax_sig.size() < group + kPredSlots) {
            ctx->minmax_sig.resize(group + kPredSlots, 0);
            ctx->minmax_used.resize(group + kPredSlots, 0);
        }
        const uint64_t sig = minmax_signature(L, (int)sizeof(T), REF);
        size_t slot = group;
        for (size_t k = group; k < group + kPredSlots; ++k) {
            if (ctx->minmax_sig[k] == sig) {
                slot = k;
                predicted = true;
                break;
            }
            if (ctx->minmax_used[k] < ctx->minmax_used[slot]) slot = k;
        }
        ctx->minmax_sig[slot] = sig;  // (this call leaves predictions for the next one of its signature, whichever way it runs)
        ctx->minmax_used[slot] = ++ctx->minmax_tick;
        L.prm.pred = static_cast<uint32_t *>(ctx->minmax_pred) + slot * kMaxPlanesPS * 2;
        if (predicted) ++ctx->minmax_predicted;
    } else {
        L.prm.pred = nullptr;
    }
    if (no_thr) {
        {
            vszip_probe_scope probe(ctx);
            hipLaunchKernelGGL((minmax_kernel<T, REF>), dim3(L.total_blocks), dim3(kThreads), 0, ctx->stream, L.prm);
        }
        hipLaunchKernelGGL(minmax_final_kernel, dim3(L.prm.nplanes), dim3(64), 0, ctx->stream, L.prm, is_int);
    } else {
        // hist, bucket and shist are contiguous: one memset
        VSZIP_HIP_CHECK(ctx, hipMemsetAsync(L.prm.hist, 0, (size_t)L.prm.nplanes * (kHistWords + kBucketWords + 256) * sizeof(uint32_t), ctx->stream));
        const int sampled = wide && ctx->opt.minmax_single_read ? 1 : 0;  // (development variant: the ranges from a row-sample pass, round 3)
        const int single = sampled || predicted ? 1 : 0;
        const int grid = std::min(L.total_blocks, (REF ? 2 : 3) * 256), grid0 = std::min(L.total_blocks, 3 * 256);
        if constexpr (wide != 0) {
            if (single) {
                Launch S = L;  // (the sampled variant's sweep reads its range starts from the bucket, the predicted one from PSParams::pred)
#ifdef VSZIP_DEV_VARIANTS
                if (sampled) {
                    S.prm.pred = nullptr;
                    hipLaunchKernelGGL((hist_sample_kernel<T>), dim3(L.prm.nplanes * kSampleBlocks), dim3(kThreads), 0, ctx->stream, L.prm);
                }
#endif
                {
                    vszip_probe_scope probe(ctx);
                    hipLaunchKernelGGL((hist_sweep_kernel<T, REF, 1>), dim3(grid), dim3(kSweepThreads), 0, ctx->stream, S.prm, L.total_blocks, 0);
                }
                // The planes the single sweep flagged (an answer outside its range) go through the two-level radix; every other plane's workgroups return at
                // once - three launches that do nothing in a clip's steady state. A caller that synchronises anyway (vszip_plane_minmax, one group of planes)
                // takes them as a closure instead and runs it only if the flags in the results' spare words say so.
                auto fallback = [ctx, L, grid0]() -> int {
                    hipLaunchKernelGGL((hist_sweep_kernel<T, false, 0>), dim3(std::min(grid0, 256)), dim3(kSweepThreads), 0, ctx->stream, L.prm, L.total_blocks, 1);
                    const int kb = std::min(128, std::max(8, 4096 / L.prm.nplanes));
                    hipLaunchKernelGGL((hist_refine_kernel<T>), dim3(L.prm.nplanes * kb), dim3(kThreads), 0, ctx->stream, L.prm, kb);
                    hipLaunchKernelGGL(thr_final_kernel, dim3(L.prm.nplanes), dim3(64), 0, ctx->stream, L.prm, wide, is_int);
                    VSZIP_HIP_CHECK(ctx, hipGetLastError());
                    return VSZIP_OK;
                };
                if (deferred && !sampled) {
                    *deferred = fallback;
                    VSZIP_HIP_CHECK(ctx, hipGetLastError());
                    return VSZIP_OK;
                }
                return fallback();
            }
        }
        {
            (void)grid0;
            (void)single;
            vszip_probe_scope probe(ctx);
            hipLaunchKernelGGL((hist_sweep_kernel<T, REF, 0>), dim3(grid), dim3(kSweepThreads), 0, ctx->stream, L.prm, L.total_blocks, 0);
        }
        if constexpr (wide != 0) {
            const int kb = std::min(128, std::max(8, 4096 / L.prm.nplanes));  // workgroups per plane of the second sweep
            hipLaunchKernelGGL((hist_refine_kernel<T>), dim3(L.prm.nplanes * kb), dim3(kThreads), 0, ctx->stream, L.prm, kb);
        }
        hipLaunchKernelGGL(thr_final_kernel, dim3(L.prm.nplanes), dim3(64), 0, ctx->stream, L.prm, wide, is_int);  // (planes not written yet)
    }
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    return VSZIP_OK;
}

template <typename T>
int run_minmax(vszip_ctx *ctx, Launch &L, bool ref, bool no_thr, int batch, std::function<int()> *deferred) {
    return ref ? run_minmax_t<T, true>(ctx, L, no_thr, batch, deferred) : run_minmax_t<T, false>(ctx, L, no_thr, batch, deferred);
}

}  // namespace

void vszip_planestats_release(vszip_ctx *ctx) {
    if (ctx->minmax_pred) (void)hipFree(ctx->minmax_pred);
    ctx->minmax_pred = nullptr;
    ctx->minmax_pred_planes = 0;
    ctx->minmax_sig.clear();
    ctx->minmax_used.clear();
}

static int plane_average_batch(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, const int32_t *exclude, int nexclude, int bits_per_sample,
                               double *result_dev) {
    if (nexclude < 0 || (nexclude > 0 && !exclude)) return VSZIP_ERR_ARG;
    const bool ref = planes && nplanes > 0 && planes[0].ref != nullptr;
    Launch L;
    int rc = prepare(ctx, planes, nplanes, ref, L, result_dev);
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
    return rc;
}

static int plane_minmax_batch(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, float minthr, float maxthr, int bits_per_sample,
                              double *result_dev, int batch, std::function<int()> *deferred = nullptr) {
    if (minthr < 0 || minthr > 1) return vszip_set_error(ctx, VSZIP_ERR_ARG, "PlaneMinMax: minthr should be a float between 0.0 and 1.0");
    if (maxthr < 0 || maxthr > 1) return vszip_set_error(ctx, VSZIP_ERR_ARG, "PlaneMinMax: maxthr should be a float between 0.0 and 1.0");
    const bool ref = planes && nplanes > 0 && planes[0].ref != nullptr;
    Launch L;
    int rc = prepare(ctx, planes, nplanes, ref, L, result_dev);
    if (rc != VSZIP_OK) return rc;
    const bool is_float = dtype == VSZIP_F16 || dtype == VSZIP_F32;
    L.prm.minthr = minthr;
    L.prm.maxthr = maxthr;
    L.prm.hist_size = is_float ? 65536 : (1 << bits_per_sample);  // planeminmax.zig(vs):147
    L.prm.peak = (float)(L.prm.hist_size - 1);
    const bool no_thr = (maxthr == 0.0f) && (minthr == 0.0f);
    switch (dtype) {
        case VSZIP_U8: rc = run_minmax<uint8_t>(ctx, L, ref, no_thr, batch, deferred); break;
        case VSZIP_U16: rc = run_minmax<uint16_t>(ctx, L, ref, no_thr, batch, deferred); break;
        case VSZIP_F16: rc = run_minmax<_Float16>(ctx, L, ref, no_thr, batch, deferred); break;
        case VSZIP_F32: rc = run_minmax<float>(ctx, L, ref, no_thr, batch, deferred); break;
        default: return vszip_set_error(ctx, VSZIP_ERR_ARG, "PlaneMinMax: not supported Int format.");
    }
    return rc;
}

// Any number of planes per call: batches of kMaxPlanesPS (the per-plane table travels in the kernel argument).
// Every group of kMaxPlanesPS planes is queued without waiting for the one before (their results land in different entries of the
// result array; the scratch tables are reused in stream order): a 64-frame call is four groups and ONE synchronise.
static int plane_average_queue(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, const int32_t *exclude, int nexclude, int bits_per_sample, double *result_dev) {
    for (int o = 0; o < nplanes; o += kMaxPlanesPS) {
        const int rc = plane_average_batch(ctx, dtype, planes + o, std::min(kMaxPlanesPS, nplanes - o), exclude, nexclude, bits_per_sample, result_dev + (size_t)o * 4);
        if (rc != VSZIP_OK) return rc;
    }
    return VSZIP_OK;
}
static int plane_minmax_queue(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, float minthr, float maxthr, int bits_per_sample, double *result_dev,
                              std::function<int()> *deferred = nullptr) {
    if (nplanes > kMaxPlanesPS) deferred = nullptr;  // (the groups share the scratch tables in stream order: a later group's sweep wipes what a deferred fallback needs)
    for (int o = 0; o < nplanes; o += kMaxPlanesPS) {
        const int rc = plane_minmax_batch(ctx, dtype, planes + o, std::min(kMaxPlanesPS, nplanes - o), minthr, maxthr, bits_per_sample, result_dev + (size_t)o * 4, o / kMaxPlanesPS, deferred);
        if (rc != VSZIP_OK) return rc;
    }
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_plane_average(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, const int32_t *exclude, int nexclude, int bits_per_sample,
                                     double *avg, double *diff) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    double *host = nullptr, *dev = nullptr;
    int rc = result_buffer(ctx, nplanes, &host, &dev);
    if (rc != VSZIP_OK) return rc;
    rc = plane_average_queue(ctx, dtype, planes, nplanes, exclude, nexclude, bits_per_sample, dev);
    if (rc != VSZIP_OK) return rc;
    VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return fetch(ctx, nplanes, host, avg, planes[0].ref ? diff : nullptr, nullptr);
}

VSZIP_EXPORT int vszip_plane_average_async(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, const int32_t *exclude, int nexclude, int bits_per_sample,
                                           double *pinned_results) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    double *dev = nullptr;
    const int rc = caller_buffer(ctx, pinned_results, &dev);
    if (rc != VSZIP_OK) return rc;
    return plane_average_queue(ctx, dtype, planes, nplanes, exclude, nexclude, bits_per_sample, dev);
}

VSZIP_EXPORT int vszip_plane_minmax(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, float minthr, float maxthr, int bits_per_sample,
                                    double *vmin, double *vmax, double *diff) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    double *host = nullptr, *dev = nullptr;
    int rc = result_buffer(ctx, nplanes, &host, &dev);
    if (rc != VSZIP_OK) return rc;
    for (int i = 0; i < nplanes; ++i) host[(size_t)i * 4 + 3] = 0.0;  // (a sweep that cannot answer a plane sets this word)
    std::function<int()> fallback;
    rc = plane_minmax_queue(ctx, dtype, planes, nplanes, minthr, maxthr, bits_per_sample, dev, &fallback);
    if (rc != VSZIP_OK) return rc;
    VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (fallback) {
        bool flagged = false;
        for (int i = 0; i < nplanes; ++i) flagged = flagged || host[(size_t)i * 4 + 3] != 0.0;
        if (flagged) {
            ++ctx->minmax_fallbacks;
            rc = fallback();
            if (rc != VSZIP_OK) return rc;
            VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return fetch(ctx, nplanes, host, vmin, vmax, planes[0].ref ? diff : nullptr);
}

VSZIP_EXPORT int vszip_plane_minmax_async(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, float minthr, float maxthr, int bits_per_sample,
                                          double *pinned_results) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    double *dev = nullptr;
    const int rc = caller_buffer(ctx, pinned_results, &dev);
    if (rc != VSZIP_OK) return rc;
    return plane_minmax_queue(ctx, dtype, planes, nplanes, minthr, maxthr, bits_per_sample, dev);
}
