// vszip.XPSNR on gfx950: the per-frame kernel getWSSE (src/filters/xpsnr.zig:376-524).
//
// The pixel work is exact integer arithmetic per XPSNR block: squared error
// (calcSquaredError :214-251), spatial activity (spatialAct :174-212, or highds :28-64 on
// the 2x-decimated grid for frames larger than 2048x1152) and temporal activity
// (tempDiff1/2 :111-170, diff1st/2nd :66-109), reduced to u64 sums per XPSNR block. The few
// hundred per-block sums are copied back and the f64 weighting (calcSquaredErrorAndWeight
// :315-357, the <=640x480 minimum smoothing and the weighted sums, getWSSE :437-521) runs on the
// host in the reference's sequential block order, so wsse64 is identical to the reference's.
//
// Strip kernel (the path every VapourSynth frame takes): ONE launch covers the luma and chroma
// planes of a whole batch of frames. A wave owns a strip of 64 lanes x 4 pixels x 8/16 rows inside one
// block row; a lane reads its 4 pixels with one 4/8-byte load (+ two small halo loads on the
// luma), the activity filters run separably down the rows (each row is read once per strip, the
// row loop is fully unrolled so that its loads are all in flight together), and the lanes of one
// XPSNR block are folded with a segmented wave reduction before one u64 atomic per block and
// strip. Integer sums are order independent, so the atomics do not affect the result.
// The older one-workgroup-per-block kernels below remain for planes that do not meet the strip
// kernel's alignment rules (odd sizes, rows not a multiple of 4 samples).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.hpp"

namespace {

struct XArgs {
    const void *org, *rec, *p1, *p2;  // luma planes (p1/p2 may be NULL)
    int stride, w, h;
    int b, w_blk, h_blk;
    int b_val;       // 2: highds + 2x2 temporal blocks; 1: 3x3 Laplacian + per-pixel temporal
    int temporal;    // 0 none, 1 first order, 2 second order
    uint64_t *out;   // [blocks][3]: sse, saAct, taAct
};

template <typename A>
__device__ __forceinline__ A block_sum256(A v, A *sh) {
    v = wave_reduce_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

template <typename T>
__global__ __launch_bounds__(256) void xpsnr_luma_kernel(const XArgs a) {
    __shared__ uint64_t sh[4];
    const int bx = blockIdx.x, by = blockIdx.y;
    const int ox = bx * a.b, oy = by * a.b;
    const int bw = min(a.b, a.w - ox), bh = min(a.b, a.h - oy);
    const T *org = static_cast<const T *>(a.org), *rec = static_cast<const T *>(a.rec);
    const T *p1 = static_cast<const T *>(a.p1), *p2 = static_cast<const T *>(a.p2);
    const ptrdiff_t o = a.stride;
    const T *o_m0 = org + (ptrdiff_t)oy * o + ox;
    const T *r_m0 = rec + (ptrdiff_t)oy * o + ox;
    const int bv = a.b_val;
    const int x_act = ox > 0 ? 0 : bv, y_act = oy > 0 ? 0 : bv;
    const int w_act = (ox + bw < a.w) ? bw : bw - bv, h_act = (oy + bh < a.h) ? bh : bh - bv;
    const bool act = !(w_act <= x_act || h_act <= y_act);

    uint64_t sse = 0, sa = 0, ta = 0;
    for (int i = threadIdx.x; i < bw * bh; i += 256) {
        const int y = i / bw, x = i - y * bw;
        const int64_t e = (int64_t)o_m0[(ptrdiff_t)y * o + x] - (int64_t)r_m0[(ptrdiff_t)y * o + x];
        sse += (uint64_t)(e * e);
        if (!act) continue;
        if (bv == 1) {
            if (x >= x_act && x < w_act && y >= y_act && y < h_act) {  // spatialAct :174-212
                const T *rc = o_m0 + (ptrdiff_t)y * o + x;
                const int32_t f = 12 * (int32_t)rc[0] - 2 * ((int32_t)rc[-1] + (int32_t)rc[1] + (int32_t)rc[-o] + (int32_t)rc[o]) -
                                  ((int32_t)rc[-o - 1] + (int32_t)rc[-o + 1] + (int32_t)rc[o - 1] + (int32_t)rc[o + 1]);
                sa += (uint64_t)abs(f);
            }
            if (a.temporal) {  // tempDiff1/2 :111-170
                int32_t t = (int32_t)o_m0[(ptrdiff_t)y * o + x];
                const ptrdiff_t q = (ptrdiff_t)(oy + y) * o + ox + x;
                if (a.temporal == 1) {
                    if (p1) t -= (int32_t)p1[q];
                } else {
                    if (p1) t -= 2 * (int32_t)p1[q];
                    if (p1 && p2) t += (int32_t)p2[q];
                }
                ta += 2ull * (uint64_t)abs(t);
            }
        } else if (((x | y) & 1) == 0) {
            if (w_act > 12 && x >= x_act && x < w_act && y >= y_act && y < h_act) {  // highds :28-64
                const T *p = o_m0 + (ptrdiff_t)y * o + x;
                auto g = [&](ptrdiff_t d) -> int32_t { return (int32_t)p[d]; };
                const int32_t f = 12 * (g(0) + g(1) + g(o) + g(o + 1)) - 3 * (g(-o) + g(-o + 1) + g(2 * o) + g(2 * o + 1)) -
                                  3 * (g(-1) + g(2) + g(o - 1) + g(o + 2)) - 2 * (g(-o - 1) + g(-o + 2) + g(2 * o - 1) + g(2 * o + 2)) -
                                  (g(-2 * o - 1) + g(-2 * o) + g(-2 * o + 1) + g(-2 * o + 2) + g(3 * o - 1) + g(3 * o) + g(3 * o + 1) + g(3 * o + 2) +
                                   g(-o - 2) + g(-2) + g(o - 2) + g(2 * o - 2) + g(-o + 3) + g(3) + g(o + 3) + g(2 * o + 3));
                sa += (uint64_t)abs(f);
            }
            if (a.temporal) {  // diff1st / diff2nd :66-109 (2x2 block sums over the whole block)
                auto s4 = [&](const T *b) -> int32_t {
                    const T *q = b + (ptrdiff_t)(oy + y) * o + ox + x;
                    return (int32_t)q[0] + (int32_t)q[1] + (int32_t)q[o] + (int32_t)q[o + 1];
                };
                int32_t t = s4(org);
                if (a.temporal == 1) {
                    if (p1) t -= s4(p1);
                } else {
                    if (p1) t -= 2 * s4(p1);
                    if (p1 && p2) t += s4(p2);
                }
                ta += 2ull * (uint64_t)abs(t);
            }
        }
    }
    sse = block_sum256<uint64_t>(sse, sh);
    sa = block_sum256<uint64_t>(sa, sh);
    ta = block_sum256<uint64_t>(ta, sh);
    if (threadIdx.x == 0) {
        uint64_t *q = a.out + ((size_t)by * a.w_blk + bx) * 3;
        q[0] = sse;
        q[1] = sa;
        q[2] = ta;
    }
}

struct CArgs {
    const void *org, *rec;
    int stride, w, h, bx, by, nbx;
    uint64_t *out;
};

template <typename T>
__global__ __launch_bounds__(256) void xpsnr_sse_kernel(const CArgs a) {
    __shared__ uint64_t sh[4];
    const int ox = blockIdx.x * a.bx, oy = blockIdx.y * a.by;
    const int bw = min(a.bx, a.w - ox), bh = min(a.by, a.h - oy);
    const T *org = static_cast<const T *>(a.org) + (ptrdiff_t)oy * a.stride + ox;
    const T *rec = static_cast<const T *>(a.rec) + (ptrdiff_t)oy * a.stride + ox;
    uint64_t sse = 0;
    for (int i = threadIdx.x; i < bw * bh; i += 256) {
        const int y = i / bw, x = i - y * bw;
        const int64_t e = (int64_t)org[(ptrdiff_t)y * a.stride + x] - (int64_t)rec[(ptrdiff_t)y * a.stride + x];
        sse += (uint64_t)(e * e);
    }
    sse = block_sum256<uint64_t>(sse, sh);
    if (threadIdx.x == 0) a.out[(size_t)blockIdx.y * a.nbx + blockIdx.x] = sse;
}

// ---- strip kernel -----------------------------------------------------------------------------

struct XFrame {
    const void *org[3], *rec[3], *p1, *p2;  // p1/p2: luma of frames n-1 / n-2 (NULL: absent)
};
constexpr int kInlineFrames = 8;  // batches up to this size travel in the kernel argument

struct XGeo {     // strip decomposition of one plane
    int w, h, stride;  // samples
    int bx, by;        // XPSNR block size on this plane
    int nbx;           // blocks across
    int nsx;           // strips across (64 * vec samples each)
    int segs;          // row segments per block row
    int nstrips;       // nsx * segs * block rows
    int vec;           // samples per lane: 8, 4, 2 or 1 on the SSE-only path, 8 or 4 on luma with activity sums
    int out_off;       // u64 index of this plane's block sums in a frame's result
};

struct XStripArgs {
    XFrame inl[kInlineFrames];
    const XFrame *tab;  // non-NULL: the frames come from this device table instead of `inl`
    uint64_t *out;
    unsigned out_per_frame;
    XGeo g[3];
    int ncomp;
    int luma_act;  // plane 0 carries the activity sums (b >= 4)
    int bv;        // 1 or 2 (see XArgs::b_val)
    int tmode;     // 0 none, 1 first order, 2 second order
    int packed;    // 8-bit samples: the packed-arithmetic strip (VSZIP_XPSNR_UNPACKED=1 selects the generic one)
};

constexpr int kRowsBv1 = 8, kRowsBv2 = 8, kRowsSse = 16;

// N samples as one register-sized load (the unit every strip load is made of) and their unpacking
template <typename T, int N>
struct Raw {
    using type = uint32_t;
};
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <>
struct Raw<uint16_t, 4> {
    using type = u32x2;
};
template <>
struct Raw<uint8_t, 8> {
    using type = u32x2;
};
template <typename T>
using GP = const T __attribute__((address_space(1))) *;  // a plane pointer, known to be global memory
template <typename T, int N>
__device__ __forceinline__ typename Raw<T, N>::type load_raw(GP<T> p) {
    if constexpr (N == 1)
        return (uint32_t)p[0];
    else if constexpr (sizeof(T) * N == 2)
        return (uint32_t) * (GP<uint16_t>)p;
    else if constexpr (sizeof(T) * N == 4)
        return *(GP<uint32_t>)p;
    else
        return *(GP<u32x2>)p;
}
template <typename T, int N>
__device__ __forceinline__ void unpack(typename Raw<T, N>::type q, int v[N]) {
    if constexpr (N == 1) {
        v[0] = (int)q;
    } else if constexpr (sizeof(T) == 1) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = (q >> (8 * j)) & 0xff;
    } else if constexpr (N == 2) {
        v[0] = q & 0xffff;
        v[1] = q >> 16;
    } else {
        v[0] = q.x & 0xffff;
        v[1] = q.x >> 16;
        v[2] = q.y & 0xffff;
        v[3] = q.y >> 16;
    }
}

// Sum v over runs of equal `key` (runs are contiguous in lane order); valid in the first lane of
// each run.
template <typename A>
__device__ __forceinline__ A seg_reduce(A v, int key, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int k2 = __shfl_down(key, d, 64);
        const A v2 = __shfl_down(v, d, 64);
        if (lane + d < 64 && k2 == key) v += v2;
    }
    return v;
}

__device__ __forceinline__ void add_u64(uint64_t *p, uint64_t v) { atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v); }

// All strip functions are written in two phases: every load of the strip first, into register
// arrays, with clamped (always valid) addresses and no branch in between, so that they are all in
// flight together; then the arithmetic. Row validity is applied in the second phase.

// The BV samples either side of a lane's 4: its neighbours' centre loads, moved across lanes
// (wave_shr / wave_shl); only lane 0 (left) and lane 63 (right) need memory, and they share ONE load
// instruction (`edge`: lane 0's left samples or lane 63's right samples, StripHalo::x). Round 5: the two
// sub-dword halo loads a row were a third of the kernel's load instructions and 18 % of its time.
template <typename T, int BV>
struct StripHalo {
    static constexpr int kBits = 8 * (int)sizeof(T) * BV;
    int x;       // where this lane's edge samples start (lanes 0 and 63 only)
    bool edge;
    __device__ __forceinline__ StripHalo(int lane, int xlft, int xrgt) : x(lane == 0 ? xlft : xrgt), edge(lane == 0 || lane == 63) {}
    __device__ __forceinline__ static uint32_t first_word(uint32_t q) { return q; }
    __device__ __forceinline__ static uint32_t last_word(uint32_t q) { return q; }
    __device__ __forceinline__ static uint32_t first_word(u32x2 q) { return q.x; }
    __device__ __forceinline__ static uint32_t last_word(u32x2 q) { return q.y; }
    template <typename Q>
    __device__ __forceinline__ static void get(Q centre, uint32_t edge_samples, int lane, uint32_t &left, uint32_t &right) {
        const uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)last_word(centre), 0x138, 0xf, 0xf, false);   // wave_shr:1
        const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)first_word(centre), 0x130, 0xf, 0xf, false);  // wave_shl:1
        left = lane == 0 ? edge_samples : l >> (32 - kBits);
        right = lane == 63 ? edge_samples : (kBits == 32 ? r : r & ((1u << (kBits & 31)) - 1u));
    }
};

__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }

// SSE of a strip of an SSE-only plane (chroma; every plane when b < 4): calcSquaredError :214-251
template <typename T, int VEC>
__device__ __forceinline__ void sse_strip(const XGeo &g, GP<T> org, GP<T> rec, uint64_t *out, int sx, int brow, int seg, int lane) {
    const int ys = brow * g.by + seg * kRowsSse, ye = min(min(ys + kRowsSse, brow * g.by + g.by), g.h);
    if (ys >= ye) return;
    const int x0 = (sx * 64 + lane) * VEC;
    const bool valid = x0 < g.w;
    const int xl = valid ? x0 : 0;
    typename Raw<T, VEC>::type qo[kRowsSse], qr[kRowsSse];
#pragma unroll
    for (int i = 0; i < kRowsSse; ++i) {
        const ptrdiff_t q = (ptrdiff_t)min(ys + i, g.h - 1) * g.stride + xl;
        qo[i] = load_raw<T, VEC>(org + q);
        qr[i] = load_raw<T, VEC>(rec + q);
    }
    uint64_t sse = 0;
#pragma unroll
    for (int i = 0; i < kRowsSse; ++i) {
        if (ys + i < ye) {
            int o[VEC], r[VEC];
            unpack<T, VEC>(qo[i], o);
            unpack<T, VEC>(qr[i], r);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const uint32_t ue = (uint32_t)abs(o[j] - r[j]);
                if (x0 + j < g.w) sse += (uint64_t)(ue * ue);
            }
        }
    }
    const int key = valid ? x0 / g.bx : -1;
    sse = seg_reduce(sse, key, lane);
    const int up = __shfl_up(key, 1, 64);
    if (valid && (lane == 0 || up != key)) add_u64(out + g.out_off + (size_t)brow * g.nbx + key, sse);
}

// The same strip for 8-bit samples, 8 of them a lane, kept packed: SSE = sum(o*o) + sum(r*r) - 2 sum(o*r) from three
// v_dot4_u32_u8 a group of 4 (each sum of a strip stays below 2^24).
__device__ __forceinline__ void sse_strip_u8x8(const XGeo &g, GP<uint8_t> org, GP<uint8_t> rec, uint64_t *out, int sx, int brow, int seg, int lane) {
    const int ys = brow * g.by + seg * kRowsSse, ye = min(min(ys + kRowsSse, brow * g.by + g.by), g.h);
    if (ys >= ye) return;
    const int x0 = (sx * 64 + lane) * 8;
    const bool valid = x0 < g.w;
    const int xl = valid ? x0 : 0;
    u32x2 qo[kRowsSse], qr[kRowsSse];
#pragma unroll
    for (int i = 0; i < kRowsSse; ++i) {
        const ptrdiff_t q = (ptrdiff_t)min(ys + i, g.h - 1) * g.stride + xl;
        qo[i] = load_raw<uint8_t, 8>(org + q);
        qr[i] = load_raw<uint8_t, 8>(rec + q);
    }
    uint32_t vm[2];  // bytes of each group of 4 inside the plane
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int n = !valid ? 0 : min(max(g.w - (x0 + 4 * k), 0), 4);
        vm[k] = n >= 4 ? 0xffffffffu : (1u << (8 * n)) - 1u;
    }
    uint32_t oo = 0, rr = 0, orr = 0;
#pragma unroll
    for (int i = 0; i < kRowsSse; ++i) {
        if (ys + i < ye) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t o = qo[i][k] & vm[k], r = qr[i][k] & vm[k];
                oo = dot4(o, o, oo);
                rr = dot4(r, r, rr);
                orr = dot4(o, r, orr);
            }
        }
    }
    uint64_t sse = (uint64_t)(oo + rr - 2u * orr);
    const int key = valid ? x0 / g.bx : -1;
    sse = seg_reduce(sse, key, lane);
    const int up = __shfl_up(key, 1, 64);
    if (valid && (lane == 0 || up != key)) add_u64(out + g.out_off + (size_t)brow * g.nbx + key, sse);
}

// Geometry of one luma strip and of this lane's PX samples in it (PX == 8 only where a block is whole 8-sample groups: one key a lane).
template <int BV, int PX = 4>
struct LumaLane {
    int w, h, oy, ys, ye, x0, xc, xlft, xrgt, key, y_act, h_act;
    bool valid;
    bool mx[PX];  // sample j lies in the activity columns of its block
    __device__ __forceinline__ bool init(const XGeo &g, int RS, int sx, int brow, int seg, int lane) {
        const int b = g.bx;
        w = g.w;
        h = g.h;
        oy = brow * b;
        const int bh = min(b, h - oy);
        ys = oy + seg * RS;
        ye = min(ys + RS, oy + bh);
        if (ys >= ye) return false;
        x0 = (sx * 64 + lane) * PX;
        valid = x0 < w;
        xc = valid ? x0 : 0;             // centre load position
        xlft = max(xc - BV, 0);          // left halo (BV samples ending at xc-1)
        xrgt = min(xc + PX, w - BV);     // right halo (BV samples from xc+PX)
        key = valid ? x0 / b : -1;
        const int ox = key * b, bw = min(b, w - ox);
        // calcSquaredErrorAndWeight :283-286
        const int x_act = ox > 0 ? 0 : BV;
        y_act = oy > 0 ? 0 : BV;
        const int w_act = (ox + bw < w) ? bw : bw - BV;
        h_act = (oy + bh < h) ? bh : bh - BV;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int xr = x0 + j - ox;
            mx[j] = valid && x0 + j < w && xr >= x_act && xr < w_act && (BV == 1 || w_act > 12);
        }
        return true;
    }
    __device__ __forceinline__ bool act_row(int y) const { return y - oy >= y_act && y - oy < h_act; }
};

// One luma strip: SSE + spatial activity + temporal activity of a 256 x kRows window of one block
// row. BV == 1: the 3x3 filter of spatialAct :174-212 as rows [-1,-2,-1] / [-2,12,-2] / [-1,-2,-1];
// BV == 2: the 6x6 filter of highds :28-64 on even positions as rows A,B,C,C,B,A with
// A = [0,-1,-1,-1,-1,0], B = [-1,-2,-3,-3,-2,-1], C = [-1,-3,12,12,-3,-1].
template <typename T, int BV>
__device__ __forceinline__ void luma_strip(const XStripArgs &a, GP<T> org, GP<T> rec, GP<T> p1, GP<T> p2, uint64_t *out, int sx, int brow, int seg,
                                           int lane) {
    constexpr int RS = BV == 1 ? kRowsBv1 : kRowsBv2, NR = RS + 2 * BV;
    const XGeo &g = a.g[0];
    LumaLane<BV> L;
    if (!L.init(g, RS, sx, brow, seg, lane)) return;
    const ptrdiff_t o = g.stride;
    const int w = L.w, h = L.h, ys = L.ys, ye = L.ye, x0 = L.x0;
    const bool valid = L.valid;
    const int tmode = a.tmode;
    const bool has1 = p1 != nullptr, has2 = p1 != nullptr && p2 != nullptr;
    const int c1 = tmode == 1 ? 1 : 2;

    typename Raw<T, 4>::type qc[NR], qr[RS], q1[RS], q2[RS];
    uint32_t qe[NR];
    const StripHalo<T, BV> halo(lane, L.xlft, L.xrgt);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        qc[i] = load_raw<T, 4>(org + (ptrdiff_t)min(max(ys - BV + i, 0), h - 1) * o + L.xc);
        qe[i] = 0;
    }
    if (halo.edge) {
#pragma unroll
        for (int i = 0; i < NR; ++i) qe[i] = load_raw<T, BV>(org + (ptrdiff_t)min(max(ys - BV + i, 0), h - 1) * o + halo.x);
    }
#pragma unroll
    for (int i = 0; i < RS; ++i) qr[i] = load_raw<T, 4>(rec + (ptrdiff_t)min(ys + i, h - 1) * o + L.xc);
    if (tmode && has1) {
#pragma unroll
        for (int i = 0; i < RS; ++i) q1[i] = load_raw<T, 4>(p1 + (ptrdiff_t)min(ys + i, h - 1) * o + L.xc);
    }
    if (tmode == 2 && has2) {
#pragma unroll
        for (int i = 0; i < RS; ++i) q2[i] = load_raw<T, 4>(p2 + (ptrdiff_t)min(ys + i, h - 1) * o + L.xc);
    }

    uint64_t sse = 0;
    uint32_t sa = 0, ta = 0;
    if constexpr (BV == 1) {
        int P[4] = {0, 0, 0, 0}, Pn[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int y = ys - 1 + i;
            int v[6];
            unpack<T, 4>(qc[i], v + 1);
            uint32_t lft, rgt;
            StripHalo<T, BV>::get(qc[i], qe[i], lane, lft, rgt);
            v[0] = (int)lft;
            v[5] = (int)rgt;
            const bool emit = i >= 2 && y - 1 < ye && L.act_row(y - 1);  // the row completed by this iteration
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int h1 = -(v[j] + 2 * v[j + 1] + v[j + 2]);
                const int h0 = 12 * v[j + 1] - 2 * (v[j] + v[j + 2]);
                const int f = P[j] + h1;
                if (emit && L.mx[j]) sa += (uint32_t)abs(f);
                P[j] = Pn[j] + h0;
                Pn[j] = h1;
            }
            if (i >= 1 && i <= RS && y < ye) {
                int r[4], t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
                unpack<T, 4>(qr[i - 1], r);
                if (tmode && has1) unpack<T, 4>(q1[i - 1], t1);
                if (tmode == 2 && has2) unpack<T, 4>(q2[i - 1], t2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!(valid && x0 + j < w)) continue;
                    const uint32_t ue = (uint32_t)abs(v[j + 1] - r[j]);
                    sse += (uint64_t)(ue * ue);
                    if (tmode) ta += 2u * (uint32_t)abs(v[j + 1] - c1 * t1[j] + t2[j]);  // tempDiff1/2 :111-170
                }
            }
        }
    } else {
        int Pp[2] = {0, 0}, Pn[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < NR / 2; ++i) {
            const int y = ys - 2 + 2 * i;  // rows y, y+1
            int A[2][2], B[2][2], C[2][2], s4[2] = {0, 0};
            int ctr[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                int v[8];
                uint32_t lft, rgt;
                StripHalo<T, BV>::get(qc[2 * i + rr], qe[2 * i + rr], lane, lft, rgt);
                unpack<T, 4>(qc[2 * i + rr], v + 2);
                unpack<T, 2>(lft, v);
                unpack<T, 2>(rgt, v + 6);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int *u = v + 2 * q;
                    A[rr][q] = -(u[1] + u[2] + u[3] + u[4]);
                    B[rr][q] = -(u[0] + u[5]) - 2 * (u[1] + u[4]) - 3 * (u[2] + u[3]);
                    C[rr][q] = -(u[0] + u[5]) - 3 * (u[1] + u[4]) + 12 * (u[2] + u[3]);
                    s4[q] += u[2] + u[3];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) ctr[rr][j] = v[2 + j];
            }
            const bool emit = i >= 2 && y - 2 < ye && L.act_row(y - 2);  // the row pair completed by this iteration
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int f = Pp[q] + B[0][q] + A[1][q];
                if (emit && L.mx[2 * q]) sa += (uint32_t)abs(f);
                Pp[q] = Pn[q] + C[0][q] + C[1][q];
                Pn[q] = A[0][q] + B[1][q];
            }
            if (i >= 1 && i <= RS / 2 && y < ye) {
                int s1[2] = {0, 0}, s2[2] = {0, 0};
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int k = 2 * (i - 1) + rr;
                    int r[4], t[4];
                    unpack<T, 4>(qr[k], r);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t ue = (uint32_t)abs(ctr[rr][j] - r[j]);
                        if (valid && x0 + j < w) sse += (uint64_t)(ue * ue);
                    }
                    if (tmode && has1) {
                        unpack<T, 4>(q1[k], t);
                        s1[0] += t[0] + t[1];
                        s1[1] += t[2] + t[3];
                    }
                    if (tmode == 2 && has2) {
                        unpack<T, 4>(q2[k], t);
                        s2[0] += t[0] + t[1];
                        s2[1] += t[2] + t[3];
                    }
                }
                if (tmode) {  // diff1st / diff2nd :66-109
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (valid && x0 + 2 * q < w) ta += 2u * (uint32_t)abs(s4[q] - c1 * s1[q] + s2[q]);
                }
            }
        }
    }
    sse = seg_reduce(sse, L.key, lane);
    sa = seg_reduce(sa, L.key, lane);
    ta = seg_reduce(ta, L.key, lane);
    const int up = __shfl_up(L.key, 1, 64);
    if (valid && (lane == 0 || up != L.key)) {
        uint64_t *q = out + ((size_t)brow * g.nbx + L.key) * 3;
        add_u64(q, sse);
        if (sa) add_u64(q + 1, sa);
        if (ta) add_u64(q + 2, ta);
    }
}

// The same strip for 8-bit samples with the pixels kept packed: a lane's 4 pixels stay in one
// register, the horizontal filters are v_dot4_u32_u8 against constant weight bytes (BV == 2) or
// packed 16-bit arithmetic on pixel pairs (BV == 1), SSE is sum(o*o) + sum(r*r) - 2 sum(o*r) from
// three dot products, the first-order temporal difference is v_sad_u8. Every intermediate is an
// exact integer (|f| <= 3060 fits i16; a strip's dot-product sums stay below 2^23), so the sums
// are the same numbers as luma_strip's.
typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s as_v2s(uint32_t v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ uint32_t abs_dot2(v2s v, uint32_t mask, uint32_t acc) {  // acc + |v.lo| * mask.lo + |v.hi| * mask.hi
    const v2s a = __builtin_elementwise_max(v, -v);
    return __builtin_amdgcn_udot2(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, mask), acc, false);
}

template <int BV, int NG>  // NG groups of 4 pixels a lane: one 4- or 8-byte load a row and plane
__device__ __forceinline__ void luma_strip_u8(const XStripArgs &a, GP<uint8_t> org, GP<uint8_t> rec, GP<uint8_t> p1, GP<uint8_t> p2, uint64_t *out, int sx,
                                              int brow, int seg, int lane) {
    constexpr int RS = BV == 1 ? kRowsBv1 : kRowsBv2, NR = RS + 2 * BV, PX = 4 * NG, kHaloBits = 8 * BV;
    using Q = typename Raw<uint8_t, PX>::type;
    const XGeo &g = a.g[0];
    LumaLane<BV, PX> L;
    if (!L.init(g, RS, sx, brow, seg, lane)) return;
    const ptrdiff_t o = g.stride;
    const int w = L.w, h = L.h, ys = L.ys, ye = L.ye, x0 = L.x0;
    const bool valid = L.valid;
    uint32_t vm[NG];  // bytes of each group inside the plane (w is even)
#pragma unroll
    for (int k = 0; k < NG; ++k) vm[k] = !valid || x0 + 4 * k >= w ? 0u : (x0 + 4 * k + 3 < w ? 0xffffffffu : 0x0000ffffu);
    const int tmode = a.tmode;
#ifdef VSZIP_XPSNR_TIMING_NOTEMP  // timing only (profiles/r05_notes.md 8): results are wrong
    const bool has1 = false, has2 = false;
#else
    const bool has1 = p1 != nullptr, has2 = p1 != nullptr && p2 != nullptr;
#endif

    Q qc[NR], qr[RS], q1[RS], q2[RS];
    uint32_t qe[NR];
    const StripHalo<uint8_t, BV> halo(lane, L.xlft, L.xrgt);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        qc[i] = load_raw<uint8_t, PX>(org + (ptrdiff_t)min(max(ys - BV + i, 0), h - 1) * o + L.xc);
        qe[i] = 0;
    }
#ifndef VSZIP_XPSNR_TIMING_NOEDGE  // (defined: timing only, results are wrong)
    if (halo.edge)
#else
    if (false)
#endif
    {
#pragma unroll
        for (int i = 0; i < NR; ++i) qe[i] = load_raw<uint8_t, BV>(org + (ptrdiff_t)min(max(ys - BV + i, 0), h - 1) * o + halo.x);
    }
#pragma unroll
    for (int i = 0; i < RS; ++i) {
        qr[i] = load_raw<uint8_t, PX>(rec + (ptrdiff_t)min(ys + i, h - 1) * o + L.xc);
        q1[i] = q2[i] = Q{};
    }
    if (tmode && has1) {
#pragma unroll
        for (int i = 0; i < RS; ++i) q1[i] = load_raw<uint8_t, PX>(p1 + (ptrdiff_t)min(ys + i, h - 1) * o + L.xc);
    }
    if (tmode == 2 && has2) {
#pragma unroll
        for (int i = 0; i < RS; ++i) q2[i] = load_raw<uint8_t, PX>(p2 + (ptrdiff_t)min(ys + i, h - 1) * o + L.xc);
    }
    // group k of a row: its 4 centre bytes and the BV samples either side (the lane's other group or the halo)
    auto word = [](const Q &q, int k) -> uint32_t {
        if constexpr (NG == 1)
            return q;
        else
            return k == 0 ? q.x : q.y;
    };
    auto sides = [&](const Q &q, uint32_t edge, int k, uint32_t &lft, uint32_t &rgt) {
        uint32_t hl, hr;
        StripHalo<uint8_t, BV>::get(q, edge, lane, hl, hr);
        lft = k == 0 ? hl : word(q, k - 1) >> (32 - kHaloBits);
        rgt = k == NG - 1 ? hr : word(q, k + 1) & ((1u << kHaloBits) - 1u);
    };

    uint32_t oo = 0, rr2 = 0, orr = 0, sa = 0, ta = 0;
    if constexpr (BV == 1) {
        uint32_t m01[NG], m23[NG], t01[NG], t23[NG];
        v2s P01[NG], P23[NG], N01[NG], N23[NG];  // P: h0(y-1) - h1n(y-2); N: h1n(y-1), h1n = l + 2c + r
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            m01[k] = (L.mx[4 * k] ? 1u : 0u) | (L.mx[4 * k + 1] ? 0x10000u : 0u);
            m23[k] = (L.mx[4 * k + 2] ? 1u : 0u) | (L.mx[4 * k + 3] ? 0x10000u : 0u);
            t01[k] = (vm[k] & 0xffu ? 1u : 0u) | (vm[k] & 0xff00u ? 0x10000u : 0u);
            t23[k] = (vm[k] & 0xff0000u ? 1u : 0u) | (vm[k] & 0xff000000u ? 0x10000u : 0u);
            P01[k] = P23[k] = N01[k] = N23[k] = (v2s){0, 0};
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int y = ys - 1 + i;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const uint32_t C = word(qc[i], k);
                uint32_t lft, rgt;
                sides(qc[i], qe[i], k, lft, rgt);
                // pixel pairs as packed u16: centre (c0,c1) (c2,c3), left neighbours (l,c0) (c1,c2), right neighbours (c1,c2) (c3,r)
                const v2s c01 = as_v2s(__builtin_amdgcn_perm(0u, C, 0x0c010c00u)), c23 = as_v2s(__builtin_amdgcn_perm(0u, C, 0x0c030c02u));
                const v2s l01 = as_v2s(__builtin_amdgcn_perm(lft, C, 0x0c000c04u)), mid = as_v2s(__builtin_amdgcn_perm(0u, C, 0x0c020c01u));
                const v2s r23 = as_v2s(__builtin_amdgcn_perm(rgt, C, 0x0c040c03u));
                const v2s s01 = l01 + mid, s23 = mid + r23;
                const v2s n01 = c01 * (v2s){2, 2} + s01, n23 = c23 * (v2s){2, 2} + s23;
                const v2s h01 = c01 * (v2s){12, 12} - s01 * (v2s){2, 2}, h23 = c23 * (v2s){12, 12} - s23 * (v2s){2, 2};
                if (i >= 2 && y - 1 < ye && L.act_row(y - 1)) {
                    sa = abs_dot2(P01[k] - n01, m01[k], sa);
                    sa = abs_dot2(P23[k] - n23, m23[k], sa);
                }
                P01[k] = h01 - N01[k];
                P23[k] = h23 - N23[k];
                N01[k] = n01;
                N23[k] = n23;
                if (i >= 1 && i <= RS && y < ye) {
                    const uint32_t Cm = C & vm[k], Rm = word(qr[i - 1], k) & vm[k], T1 = word(q1[i - 1], k), T2 = word(q2[i - 1], k);
                    oo = dot4(Cm, Cm, oo);
                    rr2 = dot4(Rm, Rm, rr2);
                    orr = dot4(Cm, Rm, orr);
                    if (tmode == 1) {
                        ta = __builtin_amdgcn_sad_u8(Cm, T1 & vm[k], ta);  // tempDiff1 :111-140
                    } else if (tmode == 2) {                                // tempDiff2 :142-170: |o - 2 p1 + p2|
                        const v2s a01 = as_v2s(__builtin_amdgcn_perm(0u, T1, 0x0c010c00u)), a23 = as_v2s(__builtin_amdgcn_perm(0u, T1, 0x0c030c02u));
                        const v2s b01 = as_v2s(__builtin_amdgcn_perm(0u, T2, 0x0c010c00u)), b23 = as_v2s(__builtin_amdgcn_perm(0u, T2, 0x0c030c02u));
                        ta = abs_dot2(c01 - a01 * (v2s){2, 2} + b01, t01[k], ta);
                        ta = abs_dot2(c23 - a23 * (v2s){2, 2} + b23, t23[k], ta);
                    }
                }
            }
        }
    } else {
        int Pp[NG][2], Pn[NG][2];
        bool tq[NG][2];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            Pp[k][0] = Pp[k][1] = Pn[k][0] = Pn[k][1] = 0;
            tq[k][0] = valid && x0 + 4 * k < w;
            tq[k][1] = valid && x0 + 4 * k + 2 < w;
        }
#pragma unroll
        for (int i = 0; i < NR / 2; ++i) {
            const int y = ys - 2 + 2 * i;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                uint32_t An[2][2], Bn[2][2], s4[2] = {0, 0};
                int Cs[2][2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t C = word(qc[2 * i + r], k);
                    uint32_t LL, RR;
                    sides(qc[2 * i + r], qe[2 * i + r], k, LL, RR);
                    // position 0 sees (LL.b0, LL.b1, C.b0..b3), position 1 (C.b0..b3, RR.b0, RR.b1); weights are per byte, lowest first
                    An[r][0] = dot4(C, 0x00010101u, dot4(LL, 0x00000100u, 0));
                    Bn[r][0] = dot4(C, 0x01020303u, dot4(LL, 0x00000201u, 0));
                    Cs[r][0] = (int)dot4(C, 0x00000c0cu, 0) - (int)dot4(C, 0x01030000u, dot4(LL, 0x00000301u, 0));
                    An[r][1] = dot4(C, 0x01010100u, dot4(RR, 0x00000001u, 0));
                    Bn[r][1] = dot4(C, 0x03030201u, dot4(RR, 0x00000102u, 0));
                    Cs[r][1] = (int)dot4(C, 0x0c0c0000u, 0) - (int)dot4(C, 0x00000301u, dot4(RR, 0x00000103u, 0));
                    s4[0] = dot4(C, 0x00000101u, s4[0]);
                    s4[1] = dot4(C, 0x01010000u, s4[1]);
                }
                const bool emit = i >= 2 && y - 2 < ye && L.act_row(y - 2);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int f = Pp[k][q] - (int)Bn[0][q] - (int)An[1][q];
                    if (emit && L.mx[4 * k + 2 * q]) sa += (uint32_t)abs(f);
                    Pp[k][q] = Pn[k][q] + Cs[0][q] + Cs[1][q];
                    Pn[k][q] = -(int)An[0][q] - (int)Bn[1][q];
                }
                if (i >= 1 && i <= RS / 2 && y < ye) {
                    uint32_t s1[2] = {0, 0}, s2[2] = {0, 0};
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int kk = 2 * (i - 1) + r;
                        const uint32_t Cm = word(qc[2 * i + r], k) & vm[k], Rm = word(qr[kk], k) & vm[k], T1 = word(q1[kk], k), T2 = word(q2[kk], k);
                        oo = dot4(Cm, Cm, oo);
                        rr2 = dot4(Rm, Rm, rr2);
                        orr = dot4(Cm, Rm, orr);
                        s1[0] = dot4(T1, 0x00000101u, s1[0]);
                        s1[1] = dot4(T1, 0x01010000u, s1[1]);
                        s2[0] = dot4(T2, 0x00000101u, s2[0]);
                        s2[1] = dot4(T2, 0x01010000u, s2[1]);
                    }
                    if (tmode) {  // diff1st / diff2nd :66-109
                        const int c1 = tmode == 1 ? 1 : 2;
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            if (tq[k][q]) ta += (uint32_t)abs((int)s4[q] - c1 * (int)s1[q] + (int)s2[q]);
                    }
                }
            }
        }
    }
    uint64_t sse = (uint64_t)(oo + rr2 - 2u * orr);
    ta *= 2u;
    sse = seg_reduce(sse, L.key, lane);
    sa = seg_reduce(sa, L.key, lane);
    ta = seg_reduce(ta, L.key, lane);
    const int up = __shfl_up(L.key, 1, 64);
    if (valid && (lane == 0 || up != L.key)) {
        uint64_t *q = out + ((size_t)brow * g.nbx + L.key) * 3;
        add_u64(q, sse);
        if (sa) add_u64(q + 1, sa);
        if (ta) add_u64(q + 2, ta);
    }
}

// ---- f64 weighting on the device --------------------------------------------------------------
// getWSSE :437-521 for one frame per workgroup, from the block sums the kernels above leave in
// device memory. IEEE f64 add / mul / div / sqrt are correctly rounded on gfx950 as on the host and
// contraction is off, so evaluating the reference's expressions in the reference's order gives the
// reference's bits: the per-block weights are independent (all threads), the <=640x480 smoothing
// and the three weighted sums are sequential recurrences (one thread each, in block order).
constexpr int kWeighMaxBlocks = 1536;  // 4 f64 arrays of this length in LDS

struct WArgs {
    uint64_t *sums;        // [frame][total]; left zeroed
    uint64_t *wsse;        // [frame][3]
    unsigned total;
    int w, h, b, w_blk, n_luma, b_val, temporal, num_comps, small;  // small: wh <= 640*480
    unsigned coff[3], cn[3];
    double sf, avg_act;
};

__global__ __launch_bounds__(256) void xpsnr_weigh_kernel(const WArgs a) {
    __shared__ double W[kWeighMaxBlocks], P[3][kWeighMaxBlocks];
    uint64_t *res = a.sums + (size_t)blockIdx.x * a.total;
    uint64_t *out = a.wsse + (size_t)blockIdx.x * 3;
    const int tid = threadIdx.x;
    if (a.b < 4) {
        if (tid < 3) out[tid] = tid < a.num_comps ? res[a.coff[tid]] : 0;
        if (tid < a.num_comps) res[a.coff[tid]] = 0;
        return;
    }
    for (int idx = tid; idx < a.n_luma; idx += 256) {  // calcSquaredErrorAndWeight :268-357
        const int x = (idx % a.w_blk) * a.b, y = (idx / a.w_blk) * a.b;
        const int bw = min(a.b, a.w - x), bh = min(a.b, a.h - y);
        const int x_act = x > 0 ? 0 : a.b_val, y_act = y > 0 ? 0 : a.b_val;
        const int w_act = (x + bw < a.w) ? bw : bw - a.b_val, h_act = (y + bh < a.h) ? bh : bh - a.b_val;
        const uint64_t *q = res + (size_t)idx * 3;
        double ms_act = 1.0;
        if (!(w_act <= x_act || h_act <= y_act)) {
            ms_act = (double)q[1] / ((double)(w_act - x_act) * (double)(h_act - y_act));
            if (a.temporal) ms_act += (double)q[2] / ((double)bw * (double)bh);
            if (ms_act < a.sf) ms_act = a.sf;
            ms_act *= ms_act;
        }
        W[idx] = 1.0 / sqrt(ms_act);
    }
    __syncthreads();
    if (a.small && tid == 0) {  // :450-467, in block order
        const int w_blk = a.w_blk;
        for (int idx = 0; idx < a.n_luma; ++idx) {
            const int x = (idx % w_blk) * a.b, y = (idx / w_blk) * a.b;
            double prev;
            if (x == 0)
                prev = idx > 1 ? W[idx - 2] : 0;
            else
                prev = x > a.b ? fmax(W[idx - 2], W[idx]) : W[idx];
            if (idx > w_blk) prev = fmax(prev, W[idx - 1 - w_blk]);
            if (idx > 0 && W[idx - 1] > prev) W[idx - 1] = prev;
            if ((x + a.b >= a.w) && (y + a.b >= a.h) && (idx > w_blk)) {
                prev = fmax(W[idx - 1], W[idx - w_blk]);
                if (W[idx] > prev) W[idx] = prev;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < a.n_luma; idx += 256) {
        P[0][idx] = (double)res[(size_t)idx * 3] * W[idx];
        res[(size_t)idx * 3] = res[(size_t)idx * 3 + 1] = res[(size_t)idx * 3 + 2] = 0;
        for (int c = 1; c < a.num_comps; ++c)
            if (idx < (int)a.cn[c]) {
                P[c][idx] = (double)res[a.coff[c] + idx] * W[idx];
                res[a.coff[c] + idx] = 0;
            }
    }
    __syncthreads();
    const int c = tid >> 6;
    if ((tid & 63) == 0 && c < 3) {
        uint64_t v = 0;
        if (c < a.num_comps) {
            const int n = c == 0 ? a.n_luma : (int)a.cn[c];
            // a sequential f64 chain: the operands are fetched eight at a time ahead of the adds, so a
            // step costs one add latency, not an LDS round trip
            double acc = 0.0;
            int i = 0;
            for (; i + 8 <= n; i += 8) {
                double v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = P[c][i + k];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += v[k];
            }
            for (; i < n; ++i) acc += P[c][i];
            if (acc > 0.0) {
                const double t = acc * a.avg_act + 0.5;
                v = c == 0 ? (uint64_t)trunc(t) : (uint64_t)(t < 0 ? 0 : t);
            }
        }
        out[c] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256, 4) void xpsnr_strip_kernel(const XStripArgs a) {
    const int lane = threadIdx.x & 63;
    int sid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the strip geometry in SGPRs
    int c = 0;
    while (c < a.ncomp && sid >= a.g[c].nstrips) sid -= a.g[c++].nstrips;
    if (c >= a.ncomp) return;
    const XGeo &g = a.g[c];
    const void *org, *rec, *p1, *p2;
    if (a.tab) {
        const XFrame *f = a.tab + blockIdx.y;
        org = f->org[c], rec = f->rec[c], p1 = f->p1, p2 = f->p2;
    } else {
        const XFrame &f = a.inl[blockIdx.y];
        org = f.org[c], rec = f.rec[c], p1 = f.p1, p2 = f.p2;
    }
    uint64_t *out = a.out + (size_t)blockIdx.y * a.out_per_frame;
    const int sx = sid % g.nsx, t = sid / g.nsx;
    const int brow = t / g.segs, seg = t - brow * g.segs;
    // the plane pointers come out of a table: type them as global memory (global_load instead of
    // flat_load, which would also tie up the LDS counter)
    const GP<T> o = (GP<T>)org, r = (GP<T>)rec, t1 = (GP<T>)p1, t2 = (GP<T>)p2;
    if (c == 0 && a.luma_act) {
        if constexpr (sizeof(T) == 1) {
            if (a.packed == 0) {
                if (a.bv == 1)
                    luma_strip<T, 1>(a, o, r, t1, t2, out, sx, brow, seg, lane);
                else
                    luma_strip<T, 2>(a, o, r, t1, t2, out, sx, brow, seg, lane);
            } else if (g.vec == 8) {
                if (a.bv == 1)
                    luma_strip_u8<1, 2>(a, o, r, t1, t2, out, sx, brow, seg, lane);
                else
                    luma_strip_u8<2, 2>(a, o, r, t1, t2, out, sx, brow, seg, lane);
            } else if (a.bv == 1) {
                luma_strip_u8<1, 1>(a, o, r, t1, t2, out, sx, brow, seg, lane);
            } else {
                luma_strip_u8<2, 1>(a, o, r, t1, t2, out, sx, brow, seg, lane);
            }
        } else if (a.bv == 1) {
            luma_strip<T, 1>(a, o, r, t1, t2, out, sx, brow, seg, lane);
        } else {
            luma_strip<T, 2>(a, o, r, t1, t2, out, sx, brow, seg, lane);
        }
    } else if (g.vec == 8) {
        if constexpr (sizeof(T) == 1) sse_strip_u8x8(g, o, r, out, sx, brow, seg, lane);
    } else if (g.vec == 4) {
        sse_strip<T, 4>(g, o, r, out, sx, brow, seg, lane);
    } else if (g.vec == 2) {
        sse_strip<T, 2>(g, o, r, out, sx, brow, seg, lane);
    } else {
        sse_strip<T, 1>(g, o, r, out, sx, brow, seg, lane);
    }
}

}  // namespace

// Geometry of one call: block sizes and the layout of a frame's device result.
namespace {

struct XPlan {
    uint32_t w, h, wh, b, w_blk, h_blk;
    double avg_act;
    size_t n_luma, total;
    size_t coff[3], cn[3];
    uint32_t cbx[3], cby[3], cnbx[3], cnby[3];
    int b_val, tmode;
};

int xpsnr_plan(vszip_ctx *ctx, const int *width3, const int *height3, int depth, int num_comps, unsigned frame_rate, int temporal, XPlan &p) {
    p.w = (uint32_t)width3[0];
    p.h = (uint32_t)height3[0];
    p.wh = p.w * p.h;
    // getWSSE :389-398
    const double r = (double)p.wh / (3840.0 * 2160.0);
    const double bq = 32.0 * std::sqrt(r) + 0.5;
    p.b = (uint32_t)(bq < 0 ? 0 : bq) * 4;
    p.w_blk = p.b >= 4 ? (p.w + p.b - 1) / p.b : 0;
    p.h_blk = p.b >= 4 ? (p.h + p.b - 1) / p.b : 0;
    const uint32_t sft = 1u << (2 * depth - 9);
    p.avg_act = std::sqrt(16.0 * (double)sft / std::sqrt(std::max(0.00001, r)));
    // device result layout: luma [w_blk*h_blk][3], then per chroma plane its block SSEs
    p.n_luma = (size_t)p.w_blk * p.h_blk;
    p.total = p.n_luma * 3;
    for (int c = 0; c < 3; ++c) p.coff[c] = p.cn[c] = p.cbx[c] = p.cby[c] = p.cnbx[c] = p.cnby[c] = 0;
    for (int c = 0; c < num_comps; ++c) {
        const uint32_t wp = (uint32_t)width3[c], hp = (uint32_t)height3[c];
        if (p.b < 4) {
            p.cbx[c] = wp;
            p.cby[c] = hp;
        } else if (c > 0) {
            p.cbx[c] = (p.b * wp) / p.w;
            p.cby[c] = (p.b * hp) / p.h;
        } else {
            continue;
        }
        if (p.cbx[c] == 0 || p.cby[c] == 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "XPSNR : plane %d too small", c);
        p.cnbx[c] = (wp + p.cbx[c] - 1) / p.cbx[c];
        p.cnby[c] = (hp + p.cby[c] - 1) / p.cby[c];
        p.coff[c] = p.total;
        p.cn[c] = (size_t)p.cnbx[c] * p.cnby[c];
        p.total += p.cn[c];
    }
    p.b_val = ((uint64_t)p.w * p.h > 2048ull * 1152ull) ? 2 : 1;  // calcSquaredErrorAndWeight :279
    p.tmode = temporal ? (frame_rate < 32 ? 1 : 2) : 0;
    return VSZIP_OK;
}

// The f64 weighting of one frame's block sums, in the reference's block order (getWSSE :437-521).
void xpsnr_weigh(const XPlan &p, const uint64_t *res, int depth, int num_comps, int temporal, uint64_t *wsse3) {
    const uint32_t w = p.w, h = p.h, b = p.b, w_blk = p.w_blk;
    const int b_val = p.b_val;
    wsse3[0] = wsse3[1] = wsse3[2] = 0;
    std::vector<double> weights(p.n_luma);
    if (b >= 4) {
        double wsse_luma = 0.0;
        std::vector<double> sse_luma(p.n_luma);
        size_t idx = 0;
        for (uint32_t y = 0; y < h; y += b) {
            const uint32_t bh = (y + b > h) ? (h - y) : b;
            for (uint32_t x = 0; x < w; x += b, ++idx) {
                const uint32_t bw = (x + b > w) ? (w - x) : b;
                const uint64_t *q = res + idx * 3;
                // calcSquaredErrorAndWeight :268-357
                const int x_act = x > 0 ? 0 : b_val, y_act = y > 0 ? 0 : b_val;
                const int w_act = (x + bw < w) ? (int)bw : (int)bw - b_val, h_act = (y + bh < h) ? (int)bh : (int)bh - b_val;
                double ms_act = 1.0, ms_act_prev = 0.0;
                sse_luma[idx] = (double)q[0];
                if (!(w_act <= x_act || h_act <= y_act)) {
                    ms_act = (double)q[1] / ((double)(w_act - x_act) * (double)(h_act - y_act));
                    if (temporal) ms_act += (double)q[2] / ((double)bw * (double)bh);
                    const double sf = (double)((size_t)1 << (depth - 6));
                    if (ms_act < sf) ms_act = sf;
                    ms_act *= ms_act;
                }
                weights[idx] = 1.0 / std::sqrt(ms_act);
                if (p.wh <= 640u * 480u) {  // :450-467
                    if (x == 0)
                        ms_act_prev = idx > 1 ? weights[idx - 2] : 0;
                    else
                        ms_act_prev = x > b ? std::max(weights[idx - 2], weights[idx]) : weights[idx];
                    if (idx > w_blk) ms_act_prev = std::max(ms_act_prev, weights[idx - 1 - w_blk]);
                    if (idx > 0 && weights[idx - 1] > ms_act_prev) weights[idx - 1] = ms_act_prev;
                    if ((x + b >= w) && (y + b >= h) && (idx > w_blk)) {
                        ms_act_prev = std::max(weights[idx - 1], weights[idx - w_blk]);
                        if (weights[idx] > ms_act_prev) weights[idx] = ms_act_prev;
                    }
                }
            }
        }
        for (size_t i = 0; i < idx; ++i) wsse_luma += sse_luma[i] * weights[i];
        wsse3[0] = wsse_luma <= 0.0 ? 0 : (uint64_t)std::trunc(wsse_luma * p.avg_act + 0.5);
    }
    for (int c = 0; c < num_comps; ++c) {
        if (b < 4) {
            wsse3[c] = res[p.coff[c]];
        } else if (c > 0) {
            double wsse_chroma = 0.0;
            for (size_t i = 0; i < p.cn[c]; ++i) wsse_chroma += (double)res[p.coff[c] + i] * weights[i];
            const double v = wsse_chroma * p.avg_act + 0.5;
            wsse3[c] = wsse_chroma <= 0.0 ? 0 : (uint64_t)(v < 0 ? 0 : v);
        }
    }
}

// One workgroup per block: the launches of one frame into dev[0..total).
template <typename T>
void launch_block_kernels(vszip_ctx *ctx, const XPlan &p, const void *const *org3, const void *const *rec3, const void *prev1, const void *prev2,
                          const int *width3, const int *height3, const ptrdiff_t *stride3, int num_comps, int temporal, uint64_t *dev) {
    if (p.b >= 4) {
        XArgs a;
        a.org = org3[0];
        a.rec = rec3[0];
        a.p1 = temporal ? prev1 : nullptr;
        a.p2 = (temporal && p.tmode == 2) ? prev2 : nullptr;
        a.stride = (int)stride3[0];
        a.w = (int)p.w;
        a.h = (int)p.h;
        a.b = (int)p.b;
        a.w_blk = (int)p.w_blk;
        a.h_blk = (int)p.h_blk;
        a.b_val = p.b_val;
        a.temporal = p.tmode;
        a.out = dev;
        hipLaunchKernelGGL((xpsnr_luma_kernel<T>), dim3(p.w_blk, p.h_blk), dim3(256), 0, ctx->stream, a);
    }
    for (int c = 0; c < num_comps; ++c) {
        if (p.cn[c] == 0) continue;
        CArgs a;
        a.org = org3[c];
        a.rec = rec3[c];
        a.stride = (int)stride3[c];
        a.w = width3[c];
        a.h = height3[c];
        a.bx = (int)p.cbx[c];
        a.by = (int)p.cby[c];
        a.nbx = (int)p.cnbx[c];
        a.out = dev + p.coff[c];
        hipLaunchKernelGGL((xpsnr_sse_kernel<T>), dim3(p.cnbx[c], p.cnby[c]), dim3(256), 0, ctx->stream, a);
    }
}

bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

VSZIP_EXPORT int vszip_xpsnr_wsse_batch(vszip_ctx *ctx, int bytes_per_sample, int nframes, const void *const *org3, const void *const *rec3,
                                        const void *const *prev1, const void *const *prev2, const int *width3, const int *height3, const ptrdiff_t *stride3,
                                        int depth, int num_comps, unsigned frame_rate, int temporal, uint64_t *wsse3) {
    if (!ctx || !org3 || !rec3 || !width3 || !height3 || !stride3 || !wsse3 || num_comps < 1 || num_comps > 3 || nframes < 0) return VSZIP_ERR_ARG;
    if (bytes_per_sample != 1 && bytes_per_sample != 2) return vszip_set_error(ctx, VSZIP_ERR_ARG, "XPSNR : only supports 8 or 10 bit clips");
    if (nframes == 0) return VSZIP_OK;
    if (nframes > 65535) return vszip_set_error(ctx, VSZIP_ERR_ARG, "XPSNR : at most 65535 frames per call");
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XPlan p;
    int rc = xpsnr_plan(ctx, width3, height3, depth, num_comps, frame_rate, temporal, p);
    if (rc != VSZIP_OK) return rc;

    // strip kernel eligibility: even planes, rows that hold whole 4-sample groups, aligned bases
    const size_t bps = (size_t)bytes_per_sample;
    XStripArgs sa;
    bool strips = (p.w % 2 == 0) && (p.h % 2 == 0) && !ctx->opt.xpsnr_blocks;
    size_t tab_bytes = 0;
    if (strips) {
        sa.ncomp = num_comps;
        sa.luma_act = p.b >= 4;
        sa.bv = p.b_val;
        sa.tmode = p.tmode;
        sa.packed = !ctx->opt.xpsnr_unpacked;
        for (int c = 0; c < num_comps && strips; ++c) {
            XGeo &g = sa.g[c];
            g.w = width3[c];
            g.h = height3[c];
            g.stride = (int)stride3[c];
            const bool luma = c == 0 && sa.luma_act;
            g.bx = luma ? (int)p.b : (int)p.cbx[c];
            g.by = luma ? (int)p.b : (int)p.cby[c];
            g.nbx = luma ? (int)p.w_blk : (int)p.cnbx[c];
            g.out_off = luma ? 0 : (int)p.coff[c];
            auto fits = [&](int v) {
                if (g.bx % v || stride3[c] % v || stride3[c] < (ptrdiff_t)((g.w + v - 1) / v * v)) return false;
                for (int f = 0; f < nframes; ++f) {
                    if (!aligned_to(org3[f * num_comps + c], v * bps) || !aligned_to(rec3[f * num_comps + c], v * bps)) return false;
                    if (luma && ((prev1 && !aligned_to(prev1[f], v * bps)) || (prev2 && !aligned_to(prev2[f], v * bps)))) return false;
                }
                return true;
            };
            g.vec = luma ? (bps == 1 && sa.packed && fits(8) ? 8 : 4) : (bps == 1 && sa.packed && fits(8) ? 8 : fits(4) ? 4 : fits(2) ? 2 : 1);  // luma: 8 samples a lane where a block is whole 8-sample groups
            if (luma && !fits(4)) strips = false;
            const int rs = luma ? (p.b_val == 1 ? kRowsBv1 : kRowsBv2) : kRowsSse;
            const int nby = (g.h + g.by - 1) / g.by;
            g.nsx = (g.w + 64 * g.vec - 1) / (64 * g.vec);
            g.segs = (std::min(g.by, g.h) + rs - 1) / rs;
            g.nstrips = g.nsx * g.segs * nby;
        }
        if (nframes > kInlineFrames) tab_bytes = ((size_t)nframes * sizeof(XFrame) + 255) & ~(size_t)255;
    }

    const size_t res_bytes = (size_t)nframes * p.total * sizeof(uint64_t);
    const size_t wsse_bytes = (size_t)nframes * 3 * sizeof(uint64_t);
    bool dev_weigh = p.n_luma <= (size_t)kWeighMaxBlocks && !ctx->opt.xpsnr_host_weigh;
    for (int c = 1; c < num_comps; ++c) dev_weigh = dev_weigh && (p.b < 4 || p.cn[c] <= p.n_luma);
    if (tab_bytes) {
        rc = vszip_ensure_scratch(ctx, tab_bytes);
        if (rc != VSZIP_OK) return rc;
    }
    rc = vszip_ensure_scalars(ctx, std::max(dev_weigh ? wsse_bytes : res_bytes, tab_bytes));
    if (rc != VSZIP_OK) return rc;
    // The block sums live in a buffer of their own: the strip kernel accumulates into it with
    // atomics, the weighting kernel zeroes what it consumed, so a steady stream of calls never
    // pays for a memset.
    if (res_bytes > ctx->xpsnr_sums_bytes) {
        if (ctx->xpsnr_sums) {
            VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->xpsnr_sums);
            ctx->xpsnr_sums = nullptr;
            ctx->xpsnr_sums_bytes = 0;
        }
        const size_t want = std::max<size_t>(res_bytes * 2, 1 << 20);
        if (vszip_hip_malloc(ctx, &ctx->xpsnr_sums, want) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "XPSNR : block sum buffer allocation failed");
        ctx->xpsnr_sums_bytes = want;
        ctx->xpsnr_clean = false;
    }
    uint64_t *dev = static_cast<uint64_t *>(ctx->xpsnr_sums);
    if (!ctx->xpsnr_clean) {
        VSZIP_HIP_CHECK(ctx, hipMemsetAsync(dev, 0, ctx->xpsnr_sums_bytes, ctx->stream));
        ctx->xpsnr_clean = true;
    }

    if (strips) {
        auto fill = [&](XFrame &x, int f) {
            for (int c = 0; c < 3; ++c) {
                x.org[c] = c < num_comps ? org3[f * num_comps + c] : nullptr;
                x.rec[c] = c < num_comps ? rec3[f * num_comps + c] : nullptr;
            }
            x.p1 = (temporal && prev1) ? prev1[f] : nullptr;
            x.p2 = (temporal && p.tmode == 2 && prev2) ? prev2[f] : nullptr;
        };
        sa.tab = nullptr;
        if (nframes > kInlineFrames) {
            XFrame *host = static_cast<XFrame *>(ctx->scalars_host);  // pinned
            for (int f = 0; f < nframes; ++f) fill(host[f], f);
            XFrame *tab_dev = static_cast<XFrame *>(ctx->scratch);
            VSZIP_HIP_CHECK(ctx, hipMemcpyAsync(tab_dev, host, (size_t)nframes * sizeof(XFrame), hipMemcpyHostToDevice, ctx->stream));
            sa.tab = tab_dev;
        } else {
            for (int f = 0; f < kInlineFrames; ++f) fill(sa.inl[f], std::min(f, nframes - 1));
        }
        sa.out = dev;
        sa.out_per_frame = (unsigned)p.total;
        int nstrips = 0;
        for (int c = 0; c < num_comps; ++c) nstrips += sa.g[c].nstrips;
        ctx->xpsnr_clean = false;  // until the weighting kernel has consumed the sums
        {
            vszip_probe_scope probe(ctx);
            if (bytes_per_sample == 1)
                hipLaunchKernelGGL((xpsnr_strip_kernel<uint8_t>), dim3((nstrips + 3) / 4, nframes), dim3(256), 0, ctx->stream, sa);
            else
                hipLaunchKernelGGL((xpsnr_strip_kernel<uint16_t>), dim3((nstrips + 3) / 4, nframes), dim3(256), 0, ctx->stream, sa);
        }
    } else {
        for (int f = 0; f < nframes; ++f) {
            const void *q1 = prev1 ? prev1[f] : nullptr, *q2 = prev2 ? prev2[f] : nullptr;
            if (bytes_per_sample == 1)
                launch_block_kernels<uint8_t>(ctx, p, org3 + f * num_comps, rec3 + f * num_comps, q1, q2, width3, height3, stride3, num_comps, temporal, dev + f * p.total);
            else
                launch_block_kernels<uint16_t>(ctx, p, org3 + f * num_comps, rec3 + f * num_comps, q1, q2, width3, height3, stride3, num_comps, temporal, dev + f * p.total);
        }
    }
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    if (dev_weigh) {
        WArgs wa;
        wa.sums = dev;
        // results go straight into the pinned host buffer (device-visible): no copy command
        VSZIP_HIP_CHECK(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&wa.wsse), ctx->scalars_host, 0));
        wa.total = (unsigned)p.total;
        wa.w = (int)p.w;
        wa.h = (int)p.h;
        wa.b = (int)p.b;
        wa.w_blk = (int)p.w_blk;
        wa.n_luma = (int)p.n_luma;
        wa.b_val = p.b_val;
        wa.temporal = temporal ? 1 : 0;
        wa.num_comps = num_comps;
        wa.small = p.wh <= 640u * 480u;
        for (int c = 0; c < 3; ++c) {
            wa.coff[c] = (unsigned)p.coff[c];
            wa.cn[c] = (unsigned)p.cn[c];
        }
        wa.sf = (double)((size_t)1 << (depth - 6));
        wa.avg_act = p.avg_act;
        hipLaunchKernelGGL(xpsnr_weigh_kernel, dim3(nframes), dim3(256), 0, ctx->stream, wa);
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        ctx->xpsnr_clean = true;
        VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        std::copy_n(static_cast<const uint64_t *>(ctx->scalars_host), (size_t)nframes * 3, wsse3);
        return VSZIP_OK;
    }
    ctx->xpsnr_clean = false;
    VSZIP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->scalars_host, dev, res_bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t *res = static_cast<const uint64_t *>(ctx->scalars_host);
    for (int f = 0; f < nframes; ++f) xpsnr_weigh(p, res + (size_t)f * p.total, depth, num_comps, temporal, wsse3 + (size_t)f * 3);
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_xpsnr_wsse(vszip_ctx *ctx, int bytes_per_sample, const void *const *org3, const void *const *rec3, const void *prev1, const void *prev2,
                                  const int *width3, const int *height3, const ptrdiff_t *stride3, int depth, int num_comps, unsigned frame_rate, int temporal,
                                  uint64_t *wsse3) {
    return vszip_xpsnr_wsse_batch(ctx, bytes_per_sample, 1, org3, rec3, &prev1, &prev2, width3, height3, stride3, depth, num_comps, frame_rate, temporal, wsse3);
}

// (vszip_xpsnr_value / vszip_xpsnr_average are device-free: host_params.cpp)
