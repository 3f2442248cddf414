// vszip.BoxBlur on gfx950.
//
// Replaces src/filters/boxblur_comptime.zig (CT path: hradius == vradius in
// [1,22], one pass per axis) and src/filters/boxblur_runtime.zig (RT path), as
// dispatched by src/vapoursynth/boxblur.zig:85-113,188-209.
//
// CT integer path (the BASELINE r=13 YUV420P16 case) — one wave streams a
// column tile down a band of rows:
//   * lane l owns 8 adjacent columns (one 16-byte load per row for u16);
//     vertical column sums live in registers and slide by one row per step
//     (+ entering row, - leaving row), which is exact integer arithmetic and so
//     identical to the reference's colUpdate/colRecompute (:72-112);
//   * tmp = (col*inv + 2^31) >> 32 is evaluated as mulhi(col + r, ceil(2^32/k))
//     (identical for every reachable col; tests/test_oracle_boxblur.py proves it
//     exhaustively);
//   * the horizontal 16.16 running sum of hBlurInt (:130-159) has the closed
//     form  dst[x] = (inv2*E_x + 32768 + ((E_0*invlo) >> 16)) >> 16  with E_x
//     the edge-duplicating mirrored window sum of tmp, E_0 the one at x = 0,
//     inv2 = inv >> 16 and invlo = inv & 0xffff.  E_x comes from a wave-wide
//     prefix sum of tmp (in-lane adds + a DPP scan of the lane totals) parked in
//     2 KiB of LDS: E_x = P[x+r] - P[x-r-1].  The first lanes of every wave own
//     plane columns [0, r] so that E_0 is available to every tile.
//   * virtual columns left of 0 / right of w-1 are loaded from their mirrored
//     source column, so tile edges need no special window arithmetic.
// HBM traffic is one read + one write of the plane; the halo rows/columns
// re-read by neighbouring tiles are L2 / Infinity-Cache hits.
//
// Two kernels implement it:
//   boxblur_ct_ring_kernel  (16-byte aligned planes — every VapourSynth frame):
//     the 2r+1 window rows stay PACKED IN REGISTERS (a ring of 2r+1+D uint4 per
//     lane, the row loop unrolled over one ring period so that every slot index
//     is a compile-time constant), so each source row crosses the fabric once
//     per band; rows are prefetched D steps ahead; E_0's term is kept per wave
//     from plane columns 0..r (two 1-pixel loads per step), which frees two
//     lanes and makes 480 output columns per wave (3840 = 8 tiles, 1920 = 4,
//     960 = 2); blockIdx is remapped so that neighbouring tiles/bands run on
//     one XCD and find their halo in that XCD's L2.
//   boxblur_ct_int_kernel   (any alignment / stride): window rows re-read from
//     cache, first lanes own columns [0, r] for E_0.
#pragma once
#include <cmath>
#include <cstdlib>
#include <utility>

#include "common.hpp"

namespace {

// cache-policy bits of the row loads / stores (buffer instruction aux operand). Stores carry the
// non-temporal hint (aux 2 = nt on gfx94x/gfx950): the output is never read back, and keeping it
// out of L2 leaves the cache to the halo rows — measured 724 -> 696 us per 64-frame launch, A/B
// five times inside one run. The same hint on the loads costs 15 % (the halo re-reads miss).
#ifndef VSZIP_ST_AUX
#define VSZIP_ST_AUX 2
#endif
#ifndef VSZIP_LD_AUX
#define VSZIP_LD_AUX 0
#endif
#ifndef VSZIP_ST8_AUX
#define VSZIP_ST8_AUX 0
#endif
#ifndef VSZIP_ST16_AUX
#define VSZIP_ST16_AUX 2
#endif
constexpr int kStoreAux = VSZIP_ST_AUX, kLoadAux = VSZIP_LD_AUX, kStoreAux8 = VSZIP_ST8_AUX, kStoreAux16 = VSZIP_ST16_AUX;  // (16: 8-bit planes, 16 pixels a lane)
typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t U32x2 __attribute__((ext_vector_type(2)));

constexpr int kMaxPlanes = 48;  // planes per launch (kernel-argument table)
constexpr int PX = 8;           // pixels per lane per row

struct BBPlane {
    const void *src;
    void *dst;
    int sstride, dstride;  // elements
    int w, h;
    int block0;  // first block index of this plane
    int ntx;     // column tiles
    int nbands;  // row bands
    int nperiods;  // ring kernel: ring periods covering the plane, ceil(h / NR), split evenly over the bands
    int cx0;       // ring kernel: plane column of this launch's first tile (a plane may be split between two launches by columns)
};

struct BBParams {
    BBPlane p[kMaxPlanes];
    int nplanes;
    int band_rows;
    int nblocks;
};

// Ring kernel launch table: up to 64 4K YUV frames per launch (longer bands, fewer re-read halo
// rows). The block -> plane map replaces a linear scan of the plane table by every wave.
constexpr int kRingMaxPlanes = 192;
constexpr int kRingMaxBlocks = 8192;
struct RingParams {
    BBPlane p[kRingMaxPlanes];
    int nplanes;
    int nblocks;
    uint8_t plane_of_block[kRingMaxBlocks];
};

// boxblur_comptime.zig:50-70 — source row of tap k for output row i.
__device__ __forceinline__ int ct_tap_row(int k, int i, int radius, int ih) {
    const int dist_from_bottom = ih - 1 - i;
    if (k < radius) return (i < radius - k) ? min(radius - k - i, ih - 1) : (i - radius + k);
    return (dist_from_bottom < k - radius) ? (i - min(k - radius - dist_from_bottom, i)) : (i - radius + k);
}

// Eight pixels of one lane, still packed as loaded (kept packed while the load is
// in flight so that no s_waitcnt lands before the row's arithmetic).
template <typename T>
struct Raw8;
template <>
struct Raw8<uint16_t> {
    uint4 q;
};
template <>
struct Raw8<uint8_t> {
    uint2 q;
};

__device__ __forceinline__ void unpack8(const Raw8<uint16_t> &r, uint32_t v[PX]) {
    v[0] = r.q.x & 0xffffu; v[1] = r.q.x >> 16;
    v[2] = r.q.y & 0xffffu; v[3] = r.q.y >> 16;
    v[4] = r.q.z & 0xffffu; v[5] = r.q.z >> 16;
    v[6] = r.q.w & 0xffffu; v[7] = r.q.w >> 16;
}

__device__ __forceinline__ void unpack8(const Raw8<uint8_t> &r, uint32_t v[PX]) {
    v[0] = r.q.x & 0xffu; v[1] = (r.q.x >> 8) & 0xffu; v[2] = (r.q.x >> 16) & 0xffu; v[3] = r.q.x >> 24;
    v[4] = r.q.y & 0xffu; v[5] = (r.q.y >> 8) & 0xffu; v[6] = (r.q.y >> 16) & 0xffu; v[7] = r.q.y >> 24;
}

__device__ __forceinline__ void pack8(const uint32_t v[PX], Raw8<uint16_t> &r) {
    r.q.x = v[0] | (v[1] << 16);
    r.q.y = v[2] | (v[3] << 16);
    r.q.z = v[4] | (v[5] << 16);
    r.q.w = v[6] | (v[7] << 16);
}

__device__ __forceinline__ void pack8(const uint32_t v[PX], Raw8<uint8_t> &r) {
    r.q.x = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    r.q.y = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
}

// The ring kernel's lane load: PXN pixels, packed. 8 pixels a lane for 16-bit planes (16 bytes) and for 8-bit planes of any width (8 bytes); 16 pixels a
// lane for 8-bit planes whose widths are whole 16-pixel groups (round 4): a wave's row segment is then 960 bytes like a 16-bit plane's — with 480-byte
// segments the 8-bit kernel's memory side alone took 425 us per 64-frame 4K launch whatever it computed (profiles/r04_notes.md section 3).
template <typename T, int PXN>
struct RawN : Raw8<T> {};
template <>
struct RawN<uint8_t, 16> {
    uint4 q;
};
template <typename T>
__device__ __forceinline__ void unpackN(const RawN<T, 8> &r, uint32_t v[8]) {
    unpack8(static_cast<const Raw8<T> &>(r), v);
}
__device__ __forceinline__ void unpackN(const RawN<uint8_t, 16> &r, uint32_t v[16]) {
    const uint32_t dw[4] = {r.q.x, r.q.y, r.q.z, r.q.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (dw[k >> 2] >> (8 * (k & 3))) & 0xffu;
}

// Edge-duplicating mirror of a virtual column (hBlurInt's implicit padding:
// index -k -> k-1, index w-1+k -> w-k), clamped for halo columns nobody reads.
__device__ __forceinline__ int mirror_col(int c, int w) {
    c = c < 0 ? -c - 1 : c;
    c = c >= w ? 2 * w - 1 - c : c;
    return min(max(c, 0), w - 1);
}

template <typename T>
__device__ __forceinline__ Raw8<T> load8(const T *row, int vx0, int w, bool fast) {
    Raw8<T> r;
    if (fast) {
        r.q = *reinterpret_cast<const decltype(r.q) *>(row + vx0);
    } else {
        uint32_t v[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) v[k] = row[mirror_col(vx0 + k, w)];
        pack8(v, r);
    }
    return r;
}

template <typename T>
__device__ __forceinline__ void store8(T *row, int x0, int w, bool fast, const uint32_t o[PX]);

template <>
__device__ __forceinline__ void store8<uint16_t>(uint16_t *row, int x0, int w, bool fast, const uint32_t o[PX]) {
    if (fast) {
        uint4 q;
        q.x = (o[0] & 0xffffu) | (o[1] << 16);
        q.y = (o[2] & 0xffffu) | (o[3] << 16);
        q.z = (o[4] & 0xffffu) | (o[5] << 16);
        q.w = (o[6] & 0xffffu) | (o[7] << 16);
        *reinterpret_cast<uint4 *>(row + x0) = q;
    } else {
#pragma unroll
        for (int k = 0; k < PX; ++k)
            if (x0 + k < w) row[x0 + k] = (uint16_t)o[k];
    }
}

template <>
__device__ __forceinline__ void store8<uint8_t>(uint8_t *row, int x0, int w, bool fast, const uint32_t o[PX]) {
    if (fast) {
        uint2 q;
        q.x = (o[0] & 0xffu) | ((o[1] & 0xffu) << 8) | ((o[2] & 0xffu) << 16) | (o[3] << 24);
        q.y = (o[4] & 0xffu) | ((o[5] & 0xffu) << 8) | ((o[6] & 0xffu) << 16) | (o[7] << 24);
        *reinterpret_cast<uint2 *>(row + x0) = q;
    } else {
#pragma unroll
        for (int k = 0; k < PX; ++k)
            if (x0 + k < w) row[x0 + k] = (uint8_t)o[k];
    }
}

template <int R>
struct CtGeom {
    static constexpr int K = 2 * R + 1;
    static constexpr int NE = (R + 1 + PX - 1) / PX;   // lanes owning plane columns [0, 8*NE) for E_0
    static constexpr int HL = NE * PX;                 // left halo  (>= R + 1)
    static constexpr int HR = ((R + PX - 1) / PX) * PX; // right halo (>= R)
    static constexpr int OUT_LANES = 64 - NE - HL / PX - HR / PX;
    static constexpr int TWO = OUT_LANES * PX;         // output columns per wave tile
};

// Wave-level ordering of LDS traffic (single-wave workgroups: no s_barrier, no
// vmcnt drain — global prefetches stay in flight across it).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int R, bool DPP>
__global__ __launch_bounds__(64) void boxblur_ct_int_kernel(const BBParams prm) {
    using G = CtGeom<R>;
    constexpr uint32_t K = G::K;
    constexpr uint32_t MAGIC = (uint32_t)(((1ull << 32) + K - 1) / K);  // ceil(2^32 / k)
    constexpr uint64_t INV = ((1ull << 32) + R) / K;                    // boxblur_comptime.zig:28
    constexpr uint32_t INV2 = (uint32_t)(INV >> 16);
    constexpr uint32_t INVLO = (uint32_t)(INV & 0xffffu);

    __shared__ __attribute__((aligned(16))) uint32_t P[64 * PX];

    // block -> (plane, column tile, row band)
    int pi = 0;
    const int b = blockIdx.x;
#pragma unroll 1
    for (int i = 1; i < prm.nplanes; ++i)
        if (b >= prm.p[i].block0) pi = i;
    const BBPlane pl = prm.p[pi];
    const int lb = b - pl.block0;
    const int tx = lb % pl.ntx;
    const int by = lb / pl.ntx;
    const int w = pl.w, h = pl.h;
    const int y0 = by * prm.band_rows;
    const int y1 = min(y0 + prm.band_rows, h);
    const T *src = static_cast<const T *>(pl.src);
    T *dst = static_cast<T *>(pl.dst);

    const int lane = threadIdx.x;
    const int X0 = tx * G::TWO;
    const int vx0 = lane < G::NE ? lane * PX : X0 - G::HL + (lane - G::NE) * PX;
    constexpr int VB = sizeof(T) * PX;  // bytes per lane-load
    const bool src_al = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)((size_t)pl.sstride * sizeof(T))) & (VB - 1)) == 0;
    const bool dst_al = ((reinterpret_cast<uintptr_t>(dst) | (uintptr_t)((size_t)pl.dstride * sizeof(T))) & (VB - 1)) == 0;
    const bool in_fast = src_al && vx0 >= 0 && vx0 + PX <= w;
    const bool is_out = lane >= G::NE + G::HL / PX && lane < G::NE + G::HL / PX + G::OUT_LANES && vx0 < w;
    const bool out_fast = dst_al && vx0 + PX <= w;

    // column sums of the first row of the band: the ksize mirrored taps (:50-70, :91-112)
    uint32_t col[PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) col[k] = 0;
#pragma unroll 1
    for (int k = 0; k < (int)K; ++k) {
        uint32_t v[PX];
        unpack8(load8<T>(src + (size_t)ct_tap_row(k, y0, R, h) * pl.sstride, vx0, w, in_fast), v);
#pragma unroll
        for (int j = 0; j < PX; ++j) col[j] += v[j];
    }

    const int ci = lane * PX;  // this lane's first index into P
#pragma unroll 1
    for (int i = y0; i < y1; ++i) {
        // prefetch the rows that slide the window to output row i+1:
        //   entering row (i+1)+r, or (i+1)-1 once the window hangs over the bottom edge;
        //   leaving  row (i+1)-r-1, or r-(i+1)+1 while the window hangs over the top edge.
        Raw8<T> an, sn;
        const bool more = i + 1 < y1;
        if (more) {
            const int n = i + 1;
            const int ar = (n + R < h) ? n + R : n - 1;
            const int sr = (n <= R) ? R - n + 1 : n - R - 1;
            an = load8<T>(src + (size_t)ar * pl.sstride, vx0, w, in_fast);
            sn = load8<T>(src + (size_t)sr * pl.sstride, vx0, w, in_fast);
        }

        // vertical mean, rounded (:114-128), then in-lane inclusive prefix
        uint32_t p[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) p[k] = __umulhi(col[k] + R, MAGIC);
#pragma unroll
        for (int k = 1; k < PX; ++k) p[k] += p[k - 1];
        const uint32_t incl = DPP ? wave_incl_scan_dpp(p[PX - 1]) : wave_incl_scan_shfl(p[PX - 1]);
        const uint32_t base = incl - p[PX - 1];
#pragma unroll
        for (int k = 0; k < PX; ++k) p[k] += base;
        *reinterpret_cast<uint4 *>(&P[ci]) = make_uint4(p[0], p[1], p[2], p[3]);
        *reinterpret_cast<uint4 *>(&P[ci + 4]) = make_uint4(p[4], p[5], p[6], p[7]);
        wave_lds_fence();

        if (is_out) {
            // E_0 = tmp[r] + 2*sum_{x<r} tmp[x]  (:131-137)
            const uint32_t e0 = P[R] + P[R - 1];
            const uint32_t krow = 32768u + (uint32_t)(((uint64_t)e0 * INVLO) >> 16);
            uint32_t o[PX];
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                const uint32_t e = P[ci + k + R] - P[ci + k - R - 1];
                o[k] = (uint32_t)(((uint64_t)e * INV2 + krow) >> 16);
            }
            store8<T>(dst + (size_t)i * pl.dstride, vx0, w, out_fast, o);
        }
        wave_lds_fence();

        if (more) {
            uint32_t a[PX], sb[PX];
            unpack8(an, a);
            unpack8(sn, sb);
#pragma unroll
            for (int k = 0; k < PX; ++k) col[k] += a[k] - sb[k];
        }
    }
}


// ---------------------------------------------------------------------------
// Ring kernel
// ---------------------------------------------------------------------------

// RV: the vertical radius — R, or 0 for a horizontal-only blur (round 4: the window is then the row itself, the ring only its prefetch)
template <int R, int SLOT_VGPRS = 4, int PXN = 8, int RV = R>
struct RingGeom {
    static constexpr int K = 2 * RV + 1;  // rows of the (vertical) window
    static constexpr int HL = ((R + 1 + PXN - 1) / PXN) * PXN;  // left halo  (>= R + 1)
    static constexpr int HR = ((R + PXN - 1) / PXN) * PXN;      // right halo (>= R)
#ifdef VSZIP_RING_OUT_LANES  // (sweeps: fewer output lanes per wave, e.g. 56 = whole 128-byte lines per row segment)
    static constexpr int OUT_LANES = VSZIP_RING_OUT_LANES < 64 - HL / PXN - HR / PXN ? VSZIP_RING_OUT_LANES : 64 - HL / PXN - HR / PXN;
#else
    static constexpr int OUT_LANES = 64 - HL / PXN - HR / PXN;
#endif
    static constexpr int TWO = OUT_LANES * PXN;
    // Rows prefetched ahead of their first use: D is odd so that NR is even (the LDS double
    // buffer alternates with the slot). The ring takes NR * SLOT_VGPRS registers, the rest of a
    // step about 56 (r=13 u16, D=1: 112 + 55 = 167 <= 168 -> 3 waves per SIMD out of the 512
    // VGPRs a SIMD lane has). A shallow prefetch (D = 1) is taken when it buys a wave per SIMD —
    // the other waves then hide the latency; otherwise D is 3, 5 or 7, whichever makes NR a
    // multiple of KL, the prefetch ring of the K_row column pixels (statically indexed, S % KL).
    static constexpr int est_vgprs(int d) { return (2 * RV + 1 + d) * SLOT_VGPRS + (PXN == 16 ? 104 : 56); }
    static constexpr int tier(int v) { return v <= 128 ? 4 : (v <= 168 ? 3 : 2); }
    static constexpr int D3 = (2 * RV + 4) % 3 == 0 ? 3 : ((2 * RV + 6) % 3 == 0 ? 5 : 7);
#ifdef VSZIP_RING_D
    static constexpr int D = VSZIP_RING_D | 1;
    static constexpr int KL = (2 * RV + 1 + D) % 3 == 0 ? 3 : 2;
#else
    static constexpr bool SHALLOW = tier(est_vgprs(1)) > tier(est_vgprs(D3));
    static constexpr int D = SHALLOW ? 1 : D3;
    static constexpr int KL = SHALLOW ? 2 : 3;
#endif
    static constexpr int NR = 2 * RV + 1 + D;   // ring slots
#ifdef VSZIP_RING_WPE
    static constexpr int WPE = VSZIP_RING_WPE;
#else
    // 8-bit planes (2 VGPRs a slot) fit four waves per SIMD at every radius — and run 4-10 % faster with THREE (168 VGPRs, 3 072 waves per launch instead
    // of 4 096: r = 2 / 5 / 13 / 20 on 64 4K YUV420P8 frames 415 -> 387, 413 -> 384, 452 -> 412, 447 -> 440 us, interleaved A/B in one process,
    // gpurun_out/r4_u8_ab*.txt; two waves lose again: 451 / 486 us at r = 13 / 20). The prefetch depth does not matter (D = 1 ... 9: 452 ... 429 us at four waves).
    static constexpr int WPE = (SLOT_VGPRS == 2 && PXN == 8) ? (tier(est_vgprs(D)) > 3 ? 3 : tier(est_vgprs(D))) : tier(est_vgprs(D));
#endif
    static_assert(NR % KL == 0, "K-column ring must divide the period");
    static constexpr uint32_t MAGIC = K > 1 ? (uint32_t)(((1ull << 32) + K - 1) / K) : 0u;  // ceil(2^32 / k): mulhi(n, MAGIC) == n / k (k = 1: no division)
};

// All per-wave state of the ring kernel. step<S>() is instantiated once per ring
// slot so that every ring index is a compile-time constant and the ring stays in
// VGPRs (a runtime-indexed array would be demoted to scratch). A ring period is
// straight-line code: bands are whole periods (the last band of a plane is
// shifted up to end at the last row and recomputes a few rows of its neighbour,
// writing identical values), so no step is conditional and every row load lands
// directly in its slot, D steps before its first use.
//
// GENERAL = false needs w % 8 == 0: lanes whose columns fall outside the plane hold
// the edge-duplicating mirror image (hBlurInt's implicit padding, :139-158) of a
// real 8-pixel group, pixel order reversed in registers, so that the window sum is
// the same compile-time-offset prefix difference everywhere. GENERAL = true takes
// any width: plane edges are evaluated as prefix differences over real columns.
template <typename T, int R, bool GENERAL, int PXN = 8, bool HONLY = false>
struct RingWave {
    static_assert(PXN == 8 || (PXN == 16 && sizeof(T) == 1 && !GENERAL), "16 pixels a lane: 8-bit planes, whole 16-pixel groups");
    static constexpr int RV = HONLY ? 0 : R;  // vertical radius
    using G = RingGeom<R, (int)(sizeof(T) * PXN / 4), PXN, RV>;
    using RawT = RawN<T, PXN>;
    using Vec = decltype(RawT{}.q);
    static constexpr int NR = G::NR;
    static constexpr uint64_t INV = ((1ull << 32) + R) / (uint64_t)(2 * R + 1);  // (the horizontal window)
    static constexpr uint32_t INV2 = (uint32_t)(INV >> 16);

    RawT ring[NR];
    uint32_t col[PXN];
    // K_row = 32768 + ((E_0 * invlo) >> 16) with E_0 = tmp[r] + 2*sum_{x<r} tmp[x] (hBlurInt's
    // start value, boxblur_comptime.zig:131-137) needs the vertical means of plane columns
    // 0..r only. Every wave keeps them itself: lane c < 32 slides the column sum of plane column
    // min(c, r) down its band (one 1-pixel load each for the entering and the leaving row,
    // prefetched KL steps ahead), a DPP row reduction gives E_0, the scalar unit the rest.
    uint32_t kcol;        // column sum (+r) of plane column min(lane, r)
    uint32_t kwgt;        // 2 for lanes < r, 1 for lane r, 0 above
    uint32_t kvo;         // byte offset of that column in a row
    uint32_t kn[G::KL], ko[G::KL];  // prefetched entering / leaving pixels
    uint32_t kn_off, ko_off;        // wave-uniform: row offsets of the next prefetch
    uint32_t kr_prev;     // wave-uniform: K_row of the previous row
    uint32_t *P;
    const char *srcb;   // wave-uniform plane bases
    char *dstb;
    uint32_t coff;      // byte offset of this lane's (real) column group in a row
    uint32_t doff;      // byte offset of this lane's output group in a row
    uint32_t next_off;  // wave-uniform: byte offset of the source row the next refill reads
    uint32_t out_off;   // wave-uniform: byte offset of the destination row the next emit writes
    uint32_t srow, drow;  // row pitches in bytes
    int w, h, lane, c0;
    bool rev, ld_ok, is_out, out_full;
    bool plain;    // wave-uniform: every row this band touches is an interior source row
    // Raw buffer descriptors of the two planes: the row offset rides in an SGPR (soffset) and
    // a lane is switched off by giving it an out-of-range voffset (the hardware drops the
    // access), so every step issues exactly one load and one store with no divergent
    // branch around them — the s_waitcnt vmcnt() counts stay exact and D rows deep.
    __amdgpu_buffer_rsrc_t rs, rd;
    uint32_t sdoff;  // store voffset: doff for output lanes, kOOB for halo lanes
    uint32_t psel;   // v_perm selector: identity, or "reversed half of the mirror partner" (mirrored lanes)
    static constexpr uint32_t kOOB = 0xfffffff0u;

    __device__ __forceinline__ RawT fetch_off(uint32_t row_off) const {
        RawT t;
        if constexpr (sizeof(T) * PXN == 16) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, coff, row_off, sizeof(T) == 2 ? kLoadAux : 0);
            t.q = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, coff, row_off, 0);
            t.q = make_uint2(v[0], v[1]);
        }
        return t;
    }
    __device__ __forceinline__ uint32_t fetch_px(uint32_t row_off) const {
        if constexpr (sizeof(T) == 2)
            return __builtin_amdgcn_raw_buffer_load_b16(rs, kvo, row_off, 0);
        else
            return __builtin_amdgcn_raw_buffer_load_b8(rs, kvo, row_off, 0);
    }
    // E_0 of the current row from the K columns: lanes 0..r hold weight * tmp, the rest 0
    __device__ __forceinline__ uint32_t krow_now() const {
        constexpr uint32_t INVLO = (uint32_t)(INV & 0xffffu);
        uint32_t v = __umul24(G::K > 1 ? __umulhi(kcol, G::MAGIC) : kcol, kwgt);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
        uint32_t e0;
        if constexpr (R < 16) {
            e0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
        } else {
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
            e0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
        }
        return 32768u + (uint32_t)(((uint64_t)e0 * INVLO) >> 16);
    }

    // Mirrored lanes hold the 8 pixels of their mirror image; put them in plane order
    // (reversed) once, when the row enters the window: one v_perm per dword with a
    // per-lane selector, identity for ordinary lanes.
    __device__ __forceinline__ void fix_order(RawT &r) const {
        if constexpr (GENERAL) return;
        if constexpr (sizeof(T) * PXN == 16) {  // (16-bit x 8 and 8-bit x 16 alike: the selector reverses halves / bytes of the partner dword)
            const uint4 o = r.q;
            r.q.x = __builtin_amdgcn_perm(o.w, o.x, psel);
            r.q.y = __builtin_amdgcn_perm(o.z, o.y, psel);
            r.q.z = __builtin_amdgcn_perm(o.y, o.z, psel);
            r.q.w = __builtin_amdgcn_perm(o.x, o.w, psel);
        } else {
            const uint2 o = r.q;
            r.q.x = __builtin_amdgcn_perm(o.y, o.x, psel);
            r.q.y = __builtin_amdgcn_perm(o.x, o.y, psel);
        }
    }
    // Virtual row v of the sliding window -> source row. Above the top edge: |v|
    // (reflect-101, boxblur_comptime.zig:56-59). Past the bottom edge the window
    // mirrors about the CURRENT row (:61-66): sliding from row i to i+1 then adds
    // row i, not row i+1+r, i.e. virtual row v >= h stands for source row v-r-1.
    // With that map the ring's "entering row" slot is always the right one.
    __device__ __forceinline__ uint32_t row_off(int v) const {
        const int r = v < 0 ? -v : (v < h ? v : v - RV - 1);
        return (uint32_t)min(r, h - 1) * srow;
    }

    template <int J>
    __device__ __forceinline__ void fill(int y0) {
        ring[J] = fetch_off(row_off(y0 - RV + J));
    }
    template <int J>
    __device__ __forceinline__ void accum() {
        uint32_t v[PXN];
        fix_order(ring[J]);
        unpackN(ring[J], v);
#pragma unroll
        for (int k = 0; k < PXN; ++k) col[k] += v[k];
    }
    // the D prefetched rows first: loads return in order, so by the time the window rows
    // have been summed the first steps of the band find their entering rows in place
    template <int... J>
    __device__ __forceinline__ void fill_all(int y0, std::integer_sequence<int, J...>) {
        (fill<(J + G::K) % NR>(y0), ...);
    }
    template <int... J>
    __device__ __forceinline__ void accum_all(std::integer_sequence<int, J...>) {
        (accum<J>(), ...);
    }

    // LDS layout of the prefix: column c of the wave tile lives at (c % 8) * 64 + c / 8,
    // i.e. [pixel-in-lane][lane]. For a fixed pixel index the 64 lanes touch 64
    // consecutive dwords, so every read and write is bank-conflict free (the natural
    // [lane][pixel] layout is an 8-way conflict: lane stride = 8 dwords over 32 banks).
    static __device__ __forceinline__ int pidx(int c) { return (c & (PXN - 1)) * 64 + c / PXN; }

    // Window sums E_x of one row from the prefix parked in LDS.
    __device__ __forceinline__ void window_sums(const uint32_t *Pb, uint32_t e[PXN]) const {
        if constexpr (!GENERAL) {
#pragma unroll
            for (int k = 0; k < PXN; ++k) {
                // compile-time offsets from the lane's own slot: (k+r) and (k-r-1) split
                // into pixel-in-lane and lane displacement
                const int hi = k + R, lo = k - R - 1;
                const int hi_px = hi & (PXN - 1), hi_ln = hi / PXN;
                const int lo_px = lo & (PXN - 1), lo_ln = (lo - lo_px) / PXN;  // floor
                e[k] = Pb[hi_px * 64 + lane + hi_ln] - Pb[lo_px * 64 + lane + lo_ln];
            }
        } else {
            // Plane edges as prefix differences over real columns, Q(c) = sum_{j<=c} tmp[j], Q(-1) = 0:
            //   E_x = Q(min(x+r, w-1)) - Q(x-r-1) + Q(w-1) - Q(min(2w-2-x-r, w-1)) + Q(r-x-1)
            // (the mirrored terms cancel by themselves away from the edges).
            const int off = lane * PXN - c0;  // tile index of plane column c is c + off
            auto lds = [&](int idx) { return Pb[pidx(min(max(idx, 0), 64 * PXN - 1))]; };
            const uint32_t qw = lds(w - 1 + off);
#pragma unroll
            for (int k = 0; k < PXN; ++k) {
                const int x = c0 + k;
                const int lo = x - R - 1, ml = R - x - 1;
                uint32_t v = lds(min(x + R, w - 1) + off) + qw - lds(min(2 * w - 2 - x - R, w - 1) + off);
                v -= lo >= 0 ? lds(lo + off) : 0u;
                v += ml >= 0 ? lds(ml + off) : 0u;
                e[k] = v;
            }
        }
    }

    // dst[x] = (inv2*E_x + K_row) >> 16 (hBlurInt :130-159 in closed form). The 16.16
    // sum is the running mean + 0.5 and never exceeds 65535.5 * 65536, so it fits 32
    // bits: one 24-bit multiply-add per pixel, the result is the high half.
    __device__ __forceinline__ void emit_row(const uint32_t e[PXN], uint32_t row_off_bytes, uint32_t kr, bool live) const {
        if constexpr (GENERAL) {
            if (!live || !is_out) return;
        }
        uint32_t t[PXN];
#pragma unroll
        for (int k = 0; k < PXN; ++k) t[k] = __umul24(e[k], INV2) + kr;
        if constexpr (!GENERAL) {
#ifdef VSZIP_DIAG_NO_STORE  // (timing diagnostics only: one lane of the wave stores)
            const uint32_t vo = (live && lane == 8) ? sdoff : kOOB;
#else
            const uint32_t vo = live ? sdoff : kOOB;
#endif
            if constexpr (sizeof(T) == 2) {
                U32x4 v;
                v.x = __builtin_amdgcn_perm(t[1], t[0], 0x07060302u);
                v.y = __builtin_amdgcn_perm(t[3], t[2], 0x07060302u);
                v.z = __builtin_amdgcn_perm(t[5], t[4], 0x07060302u);
                v.w = __builtin_amdgcn_perm(t[7], t[6], 0x07060302u);
                __builtin_amdgcn_raw_buffer_store_b128(v, rd, vo, row_off_bytes, kStoreAux);
            } else if constexpr (PXN == 16) {
                uint32_t dw[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t a0 = __builtin_amdgcn_perm(t[4 * g + 1], t[4 * g], 0x0c0c0602u);  // bytes: t0.2, t1.2, 0, 0
                    const uint32_t a1 = __builtin_amdgcn_perm(t[4 * g + 3], t[4 * g + 2], 0x0c0c0602u);
                    dw[g] = a0 | (a1 << 16);
                }
                U32x4 v;
                v.x = dw[0];
                v.y = dw[1];
                v.z = dw[2];
                v.w = dw[3];
                __builtin_amdgcn_raw_buffer_store_b128(v, rd, vo, row_off_bytes, kStoreAux16);
            } else {
                U32x2 v;
                const uint32_t a0 = __builtin_amdgcn_perm(t[1], t[0], 0x0c0c0602u);  // bytes: t0.2, t1.2, 0, 0
                const uint32_t a1 = __builtin_amdgcn_perm(t[3], t[2], 0x0c0c0602u);
                const uint32_t a2 = __builtin_amdgcn_perm(t[5], t[4], 0x0c0c0602u);
                const uint32_t a3 = __builtin_amdgcn_perm(t[7], t[6], 0x0c0c0602u);
                v.x = a0 | (a1 << 16);
                v.y = a2 | (a3 << 16);
                __builtin_amdgcn_raw_buffer_store_b64(v, rd, vo, row_off_bytes, kStoreAux8);
            }
        } else {
            T *q = reinterpret_cast<T *>(dstb + (row_off_bytes + doff));
            if (out_full) {
                if constexpr (sizeof(T) == 2) {
                    uint4 v;
                    v.x = __builtin_amdgcn_perm(t[1], t[0], 0x07060302u);
                    v.y = __builtin_amdgcn_perm(t[3], t[2], 0x07060302u);
                    v.z = __builtin_amdgcn_perm(t[5], t[4], 0x07060302u);
                    v.w = __builtin_amdgcn_perm(t[7], t[6], 0x07060302u);
                    *reinterpret_cast<uint4 *>(q) = v;
                } else {
                    uint2 v;
                    const uint32_t a0 = __builtin_amdgcn_perm(t[1], t[0], 0x0c0c0602u);
                    const uint32_t a1 = __builtin_amdgcn_perm(t[3], t[2], 0x0c0c0602u);
                    const uint32_t a2 = __builtin_amdgcn_perm(t[5], t[4], 0x0c0c0602u);
                    const uint32_t a3 = __builtin_amdgcn_perm(t[7], t[6], 0x0c0c0602u);
                    v.x = a0 | (a1 << 16);
                    v.y = a2 | (a3 << 16);
                    *reinterpret_cast<uint2 *>(q) = v;
                }
            } else {
#pragma unroll
                for (int k = 0; k < PXN; ++k)
                    if (c0 + k < w) q[k] = (T)(t[k] >> 16);
            }
        }
    }

    // One row step, software-pipelined by one row so that two independent
    // dependency chains are in flight per wave:
    //   A(i):   column sums -> rounded vertical mean -> wave prefix -> LDS buffer S&1
    //   B(i-1): window sums of the previous row from LDS buffer (S&1)^1 -> store
    // NR is even, so the buffer parity is a compile-time property of the slot.
    template <int S>
    __device__ __forceinline__ void step(int i, int y0) {
        uint32_t *Pa = P + (S & 1) * (64 * PXN);
        const uint32_t *Pb = P + ((S & 1) ^ 1) * (64 * PXN);
        // Steps are scheduled one at a time: without the fence the scheduler hoists the first
        // use of a prefetched row (fix_order) to just behind its load, and the s_waitcnt that
        // comes with it collapses the D-row prefetch distance to about one row.
        __builtin_amdgcn_sched_barrier(0);

        // B (first half): LDS reads of the previous row's prefix. The very first step of a
        // band has no previous row: it reads stale LDS and its store is switched off.
        uint32_t e[PXN];
        const bool have_prev = S > 0 || i > y0;
#ifdef VSZIP_DIAG_NO_HORIZONTAL  // (timing diagnostics only: no prefix, no LDS, no window sums)
#pragma unroll
        for (int k = 0; k < PXN; ++k) e[k] = col[k];
#else
        if (!GENERAL || have_prev) window_sums(Pb, e);
#endif

        // A: vertical mean, rounded (:114-128; col carries the +r), wave-wide inclusive prefix
        uint32_t p[PXN];
#pragma unroll
        for (int k = 0; k < PXN; ++k) {
            if constexpr (G::K == 1) {
                p[k] = col[k];  // (horizontal only: the "vertical mean" of one row)
            } else if constexpr (sizeof(T) == 1) {
                // 8-bit samples: col <= 255 k + r < 2^14, so floor(col / k) == (col * ceil(2^19 / k)) >> 19 exactly
                // (error term col * (M k - 2^19) < 2^19 since M k - 2^19 < k) and the product fits 32 bits: a
                // full-rate 24-bit multiply and a shift instead of the quarter-rate v_mul_hi_u32
                constexpr uint32_t M19 = (uint32_t)(((1u << 19) + G::K - 1) / G::K);
                static_assert((uint64_t)(255u * G::K + RV) * (M19 * G::K - (1u << 19)) < (1u << 19), "8-bit divide-by-k shortcut must be exact");
                p[k] = __umul24(col[k], M19) >> 19;
            } else {
                p[k] = __umulhi(col[k], G::MAGIC);
            }
        }
        if constexpr (GENERAL) {
            const uint32_t m = ld_ok ? 0xffffffffu : 0u;  // lanes outside the plane hold zeros
#pragma unroll
            for (int k = 0; k < PXN; ++k) p[k] &= m;
        }
#ifdef VSZIP_DIAG_NO_HORIZONTAL
        (void)Pa;
        kr_prev += p[0] + p[7];
#else
#pragma unroll
        for (int k = 1; k < PXN; ++k) p[k] += p[k - 1];
        const uint32_t incl = wave_incl_scan_dpp(p[PXN - 1]);
        const uint32_t base = incl - p[PXN - 1];
#pragma unroll
        for (int k = 0; k < PXN; ++k) Pa[k * 64 + lane] = p[k] + base;
#endif

        // slide the window to row i+1: entering row i+1+r (or row i once the window
        // hangs over the bottom edge, :61-66), leaving row i-r; refill the freed slot
        // with row i+1+r+D (|v| above the top edge, clamped below the bottom edge).
        {
            constexpr int E = (S + 1 + 2 * RV) % NR;
            uint32_t a[PXN], sb[PXN];
            fix_order(ring[E]);
            unpackN(ring[E], a);
            unpackN(ring[S], sb);
#pragma unroll
            for (int k = 0; k < PXN; ++k) col[k] += a[k] - sb[k];
#ifdef VSZIP_DIAG_NO_LOAD  // (timing diagnostics only: the ring keeps what the band's fill put there)
            if (next_off == 0xffffffffu) ring[S] = fetch_off(next_off);
#else
            ring[S] = fetch_off(next_off);
#endif
            // advance the refill row: plain bands just step down one row

        }
        // K_row of this row (used by the next step's store), then slide the K columns
        const uint32_t kr_cur = krow_now();
        {
            constexpr int J = S % G::KL;
            kcol += kn[J] - ko[J];
#ifndef VSZIP_DIAG_NO_KCOL  // (timing diagnostics only: wrong results)
            kn[J] = fetch_px(kn_off);
            ko[J] = fetch_px(ko_off);
#endif

        }

        // B (second half): scale and store the previous row
        emit_row(e, out_off, kr_prev, have_prev);
        out_off += have_prev ? drow : 0u;
        kr_prev = kr_cur;
        // advance the three prefetch rows. Bands that touch no plane edge just step down one
        // row; the mirror arithmetic (about 20 scalar instructions) sits behind a wave-uniform
        // branch that holds no memory instruction, so the vmcnt bookkeeping stays exact.
        if (plain) {
            next_off += srow;
            kn_off += srow;
            ko_off += srow;
        } else {
            next_off = row_off(i + 2 + RV + G::D);
            kn_off = row_off(i + 2 + RV + G::KL);
            ko_off = row_off(i + 1 - RV + G::KL);
        }
#ifndef VSZIP_DIAG_NO_FENCE  // (timing diagnostics only: wrong results)
        wave_lds_fence();
#endif
    }

    template <int... S>
    __device__ __forceinline__ void period(int i0, int y0, std::integer_sequence<int, S...>) {
        (step<S>(i0 + S, y0), ...);
    }
};

template <typename T, int R, bool GENERAL, int PXN = 8, bool HONLY = false>
__global__ __launch_bounds__(64, (GENERAL ? 2 : RingGeom<R, (int)(sizeof(T) * PXN / 4), PXN, (HONLY ? 0 : R)>::WPE)) void boxblur_ct_ring_kernel(const RingParams prm) {
    using W = RingWave<T, R, GENERAL, PXN, HONLY>;
    using G = typename W::G;
    constexpr int RV = W::RV;
    static_assert(G::NR % 2 == 0 && G::NR <= 64, "ring period must be even and fit a wave");
    __shared__ __attribute__((aligned(16))) uint32_t P[2 * 64 * PXN];

    // XCD-aware remap: blocks b and b+8 share an XCD, so give every XCD one
    // contiguous chunk of the (plane, band, tile) list — neighbours share an L2.
    const int chunk = (prm.nblocks + 7) >> 3;
    const int b = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (b >= prm.nblocks) return;
    const int pi = prm.plane_of_block[b];
    const BBPlane pl = prm.p[pi];
    const int lb = b - pl.block0;
    const int tx = lb % pl.ntx;
    const int by = lb / pl.ntx;
    // band `by` owns ring periods [by*P/nb, (by+1)*P/nb); the last band is shifted up to end
    // at the last row (it recomputes < NR rows of its neighbour, writing identical values)
    const int p0 = (int)((long)by * pl.nperiods / pl.nbands);
    const int p1 = (int)((long)(by + 1) * pl.nperiods / pl.nbands);
    const int band_rows = (p1 - p0) * G::NR;
    const int y0 = min(p0 * G::NR, pl.h - band_rows);
    const int w = pl.w;

    W st;
    st.P = P;
    st.w = w;
    st.h = pl.h;
    st.srow = (uint32_t)pl.sstride * (uint32_t)sizeof(T);
    st.drow = (uint32_t)pl.dstride * (uint32_t)sizeof(T);
    st.srcb = static_cast<const char *>(pl.src);
    st.dstb = static_cast<char *>(pl.dst);
    const int lane = threadIdx.x;
    st.lane = lane;
    const int vc0 = pl.cx0 + tx * G::TWO - G::HL + lane * PXN;  // first (virtual) plane column of this lane
    st.c0 = vc0;
    st.is_out = lane >= G::HL / PXN && lane < G::HL / PXN + G::OUT_LANES && vc0 < w;
    st.doff = (uint32_t)(max(vc0, 0) * (int)sizeof(T));
    st.out_full = vc0 + PXN <= w;
    if constexpr (!GENERAL) {
        // real 8-pixel group behind this lane: itself, or its edge-duplicating mirror image
        int g = vc0;
        st.rev = false;
        if (vc0 < 0) {
            g = -vc0 - PXN;
            st.rev = true;
        } else if (vc0 >= w) {
            g = 2 * w - vc0 - PXN;
            st.rev = true;
        }
        g = min(max(g, 0), w - PXN);
        st.coff = (uint32_t)(g * (int)sizeof(T));
        st.ld_ok = true;
    } else {
        st.rev = false;
        st.ld_ok = vc0 >= 0 && vc0 < w;  // [w, stride) is readable padding
        st.coff = (uint32_t)(min(max(vc0, 0), ((w - 1) / PXN) * PXN) * (int)sizeof(T));
    }

    st.psel = st.rev ? (sizeof(T) == 2 ? 0x05040706u : 0x04050607u) : 0x03020100u;  // reversed: the partner dword's halves (16 bit) / bytes (8 bit) backwards
    st.sdoff = st.is_out ? st.doff : W::kOOB;
    // descriptors: raw (stride 0), 32-bit data format; num_records = plane bytes (ring_ok keeps it < 4 GiB)
    st.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(pl.src), 0, (int)((uint32_t)pl.h * st.srow), 0x00020000);
    st.rd = __builtin_amdgcn_make_buffer_rsrc(pl.dst, 0, (int)((uint32_t)pl.h * st.drow), 0x00020000);
    st.plain = (y0 - RV >= 0) && (y0 + band_rows + RV + G::D + 2 < pl.h);

    st.fill_all(y0, std::make_integer_sequence<int, G::NR>{});
#pragma unroll
    for (int k = 0; k < PXN; ++k) st.col[k] = RV;  // the rounding term of (col + r) / k rides along
    st.accum_all(std::make_integer_sequence<int, (int)G::K>{});
    // K columns: window sum of rows y0-r .. y0+r at plane column min(lane, r), then the first KL
    // entering / leaving pixels
    {
        const int kc = min(lane, R);
        st.kvo = (uint32_t)(kc * (int)sizeof(T));
        st.kwgt = lane < R ? 2u : (lane == R ? 1u : 0u);
        uint32_t acc = RV;
#pragma unroll
        for (int k = 0; k < (int)G::K; ++k) acc += st.fetch_px(st.row_off(y0 - RV + k));
        st.kcol = acc;
#pragma unroll
        for (int j = 0; j < G::KL; ++j) {
            st.kn[j] = st.fetch_px(st.row_off(y0 + j + 1 + RV));
            st.ko[j] = st.fetch_px(st.row_off(y0 + j - RV));
        }
        st.kn_off = st.row_off(y0 + G::KL + 1 + RV);
        st.ko_off = st.row_off(y0 + G::KL - RV);
        st.kr_prev = 0;
    }
    st.next_off = st.row_off(y0 + 1 + RV + G::D);  // window rows y0-r .. y0+r+D are in the ring
    st.out_off = (uint32_t)y0 * st.drow;

    const int y1 = y0 + band_rows;
#pragma unroll 1
    for (int i0 = y0; i0 < y1; i0 += G::NR) st.period(i0, y0, std::make_integer_sequence<int, G::NR>{});
    {
        uint32_t e[PXN];  // last row of the band: slot NR-1 wrote the odd buffer
        st.window_sums(P + 64 * PXN, e);
        st.emit_row(e, st.out_off, st.kr_prev, true);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

template <typename T, int R>
int launch_ct_int(vszip_ctx *ctx, const vszip_plane *planes, int nplanes) {
    using G = CtGeom<R>;
    int done = 0;
    while (done < nplanes) {
        BBParams prm;
        const int n = std::min(kMaxPlanes, nplanes - done);
        prm.nplanes = n;
        // rows per band: enough bands to fill the chip, few enough that the
        // (2r+1)-row warm-up of every band stays a small fraction of the work
        long total_px = 0;
        for (int i = 0; i < n; ++i) total_px += (long)planes[done + i].w * planes[done + i].h;
        int band = 64;
        while (band > 16 && total_px / ((long)G::TWO * band) < 4096) band >>= 1;
        prm.band_rows = band;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            const vszip_plane &s = planes[done + i];
            BBPlane &d = prm.p[i];
            d.src = s.src;
            d.dst = s.dst;
            d.sstride = (int)s.src_stride;
            d.dstride = (int)s.dst_stride;
            d.w = s.w;
            d.h = s.h;
            d.block0 = blocks;
            d.ntx = (s.w + G::TWO - 1) / G::TWO;
            d.nbands = (s.h + band - 1) / band;
            blocks += d.ntx * d.nbands;
        }
        if (ctx->scan_mode == 1)
            hipLaunchKernelGGL((boxblur_ct_int_kernel<T, R, false>), dim3(blocks), dim3(64), 0, ctx->stream, prm);
        else
            hipLaunchKernelGGL((boxblur_ct_int_kernel<T, R, true>), dim3(blocks), dim3(64), 0, ctx->stream, prm);
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += n;
    }
    return VSZIP_OK;
}

// columns [cx0, cx0 + ntx * TWO) of a plane (ntx == 0: nothing of this plane in this launch)
struct RingSpan {
    int cx0, ntx;
};

// Band length of a launch: every plane is cut into bands of about `target` ring periods (returned; *waves_out = the waves of that plan).
template <typename T, int R, int PXN, bool HONLY = false>
int ring_plan(const vszip_plane *planes, const RingSpan *spans, int n, double *waves_ret) {
    using G = typename RingWave<T, R, false, PXN, HONLY>::G;
    // (the planes of a batch are a few geometries repeated — luma and chroma of every frame: the search below runs over the distinct ones, weighted)
    struct Geo {
        int h, ntx, count;
    };
    std::vector<Geo> geos;
    for (int i = 0; i < n; ++i) {
        bool found = false;
        for (Geo &g : geos)
            if (g.h == planes[i].h && g.ntx == spans[i].ntx) {
                ++g.count;
                found = true;
                break;
            }
        if (!found) geos.push_back({planes[i].h, spans[i].ntx, 1});
    }
        // The choice
        // trades re-read halo rows (2r+D per band: shorter bands = more traffic) against how well
        // the waves fill the chip's wave slots over time: with W waves of up to max_len periods
        // running in ceil(W / slots) generations, the fraction of slot-time doing work is
        //     eff = total_work / (slots * generations * max_len),
        // and measured launch times follow  total_work * (1 + 0.3 * (1 - eff))  within a few percent
        // (tools/sweep_periods.sh: e.g. 64 4K frames — 19 periods: 4+2 equal bands, 3072 waves =
        // every slot, 603 us; 26: 635 us; 16: 1.17 generations, 710 us; 39: half the slots, 694 us).
        auto bands_for = [](int h, int P, int target) { return std::max((h % G::NR != 0 && P >= 2) ? 2 : 1, (P + target / 2) / target); };
        const double halo_p = (double)(G::K - 1 + G::D) / G::NR;  // warm-up rows of a band, in periods
        const double slots = 256.0 * 4 * G::WPE;
        double waves_out = 0;
        auto cost_for = [&](int target) {
            double work = 0, waves = 0, max_len = 0;
            for (const Geo &g : geos) {
                const int P = (g.h + G::NR - 1) / G::NR;
                const int nb = bands_for(g.h, P, target);
                const double ntx = (double)g.ntx * g.count;
                work += ntx * (P + nb * halo_p);
                waves += ntx * nb;
                max_len = std::max(max_len, (double)((P + nb - 1) / nb) + halo_p);
            }
            waves_out = waves;
            const double gens = std::ceil(waves / slots);
            const double eff = std::min(1.0, work / (slots * gens * max_len));
            // a launch that cannot fill the chip ends with its longest wave (a lone wave steps about
            // three times as fast as one of a full chip): small inputs get many short bands
            return std::max(work / slots * (1.0 + 0.3 * (1.0 - eff)), 0.3 * max_len);
        };
        // (small launches run out of the 256 MiB Infinity Cache and reward occupancy more than the
        // model says: never go below 60 % of the slots when shorter bands can fill them)
        int target = 1;
        double best = cost_for(1);
        const double min_waves = std::min(0.6 * slots, waves_out);
        for (int t = 2; t <= 96; ++t) {
            const double c = cost_for(t);
            if (waves_out >= min_waves && c < best) {
                best = c;
                target = t;
            }
        }
        *waves_ret = (cost_for(target), waves_out);
        return target;
}


// One kernel instance over the given column spans of the planes. GEN = false is the fast form: every lane of every tile it is given must hold a whole pixel
// group inside the plane or its left mirror image (the caller's rule); GEN = true takes any width and any tile.
template <typename T, int R, int PXN, bool GEN, bool HONLY = false>
int launch_ct_ring_spans(vszip_ctx *ctx, const vszip_plane *all_planes, const RingSpan *all_spans, int nall) {
    using G = typename RingWave<T, R, false, PXN, HONLY>::G;
    std::vector<vszip_plane> planes_v;
    std::vector<RingSpan> spans_v;
    for (int i = 0; i < nall; ++i)
        if (all_spans[i].ntx > 0) {
            planes_v.push_back(all_planes[i]);
            spans_v.push_back(all_spans[i]);
        }
    const vszip_plane *planes = planes_v.data();
    const RingSpan *spans = spans_v.data();
    const int nplanes = (int)planes_v.size();
    int done = 0;
    while (done < nplanes) {
        RingParams prm;
        const int n = std::min(kRingMaxPlanes, nplanes - done);
        auto bands_for = [](int h, int P, int target) { return std::max((h % G::NR != 0 && P >= 2) ? 2 : 1, (P + target / 2) / target); };
        double waves_out = 0;
        int target = ring_plan<T, R, PXN, HONLY>(planes + done, spans + done, n, &waves_out);
        if (ctx->opt.ring_periods > 0) target = ctx->opt.ring_periods;  // development sweep knob (-DVSZIP_DEV_VARIANTS)
#ifdef VSZIP_RING_PERIODS_FIXED  // (tools/variant.sh sweeps: one translation unit, no option)
        target = VSZIP_RING_PERIODS_FIXED;
#endif
#ifdef VSZIP_RING_PLAN_DEBUG  // (tools/variant.sh: what the band planner chose)
        fprintf(stderr, "ring plan PX%d GEN%d r=%d: n=%d target=%d waves=%.0f slots=%d NR=%d | plane0 %dx%d ntx=%d\n", PXN, (int)GEN, R, n, target, waves_out, 256 * 4 * G::WPE, G::NR,
                planes[done].w, planes[done].h, spans[done].ntx);
#endif
        VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        int blocks = 0, fit = n;
        for (int i = 0; i < n; ++i) {
            const vszip_plane &s = planes[done + i];
            BBPlane &d = prm.p[i];
            d.src = s.src;
            d.dst = s.dst;
            d.sstride = (int)s.src_stride;
            d.dstride = (int)s.dst_stride;
            d.w = s.w;
            d.h = s.h;
            d.nperiods = (s.h + G::NR - 1) / G::NR;
            d.nbands = bands_for(s.h, d.nperiods, target);
            d.ntx = spans[done + i].ntx;
            d.cx0 = spans[done + i].cx0;
            d.block0 = blocks;
            if (blocks + d.ntx * d.nbands > kRingMaxBlocks) {  // very wide planes: the rest of the batch goes into the next launch
                if (i == 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: launch table overflow (%d blocks)", blocks + d.ntx * d.nbands);
                fit = i;
                break;
            }
            for (int k = 0; k < d.ntx * d.nbands; ++k) prm.plane_of_block[blocks + k] = (uint8_t)i;
            blocks += d.ntx * d.nbands;
        }
        prm.nplanes = fit;
        prm.nblocks = blocks;
        const dim3 grid(((blocks + 7) / 8) * 8);
        {
            vszip_probe_scope probe(ctx);
            hipLaunchKernelGGL((boxblur_ct_ring_kernel<T, R, GEN, PXN, HONLY>), grid, dim3(64), 0, ctx->stream, prm);
        }
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += fit;
    }
    return VSZIP_OK;
}

// The ring kernel needs 16-byte (u8: 8-byte) aligned rows and a row pitch that
// covers whole lane groups; VapourSynth frames always satisfy this.
template <typename T>
bool ring_ok(const vszip_plane *planes, int nplanes) {
    constexpr uintptr_t VB = sizeof(T) * PX;
    for (int i = 0; i < nplanes; ++i) {
        const vszip_plane &p = planes[i];
        if ((reinterpret_cast<uintptr_t>(p.src) | reinterpret_cast<uintptr_t>(p.dst) | (uintptr_t)(p.src_stride * sizeof(T)) | (uintptr_t)(p.dst_stride * sizeof(T))) & (VB - 1)) return false;
        if (p.src_stride < ((p.w + PX - 1) / PX) * PX) return false;
        if (p.src_stride < 24) return false;       // K_row kernel reads columns [0, HL)
        if (p.h < 2 * 22 + 1 + 8) return false;    // a band is at least one ring period
        if ((uint64_t)p.src_stride * p.h * sizeof(T) >= (1ull << 32) || (uint64_t)p.dst_stride * p.h * sizeof(T) >= (1ull << 32)) return false;  // 32-bit row offsets
    }
    return true;
}

// 8-bit planes take 16 pixels a lane when their rows are 16-byte aligned (every VapourSynth frame) and no plane is narrower than 32 samples.
// Up to r = 19: beyond, the ring (2r + 1 + D slots of four VGPRs) leaves no room for two waves a SIMD and the instance spills
// (64 4K YUV420P8 frames, 16 against 8 pixels a lane, tools/u8_px16_sweep.py: r = 1 ... 7 -10 ... -12 % of the time, 8 ... 16 -4 ... -8 %, 17 ... 19 -11 ... -13 %,
// 20 ... 22 +12 ... +17 %)
constexpr int kRing16MaxR = 19;
// A wave of the 16-pixel form puts out 992 (r <= 15) or 960 columns, one of the 8-pixel form 480 / 464: widths that fill the wide tiles badly lose more to
// the empty lanes than the wider loads return — r = 13, 16 against 8 pixels a lane (tools/u8_px_by_size.py, Gpixel/s): 1280 x 720 941 / 1128, 2560 x 1440
// 1516 / 1607, 4096 x 2160 1341 / 1637, but 1080p 1724 / 1613 and 4K 1815 / 1687. From r = 8 on the wide form is taken only where its tiles are at least
// 0.93 as full as the narrow ones (small radii keep it everywhere: r = 2 wins 2-21 % at every size).
template <typename T, int R, bool HONLY = false>
bool ring16_ok(const vszip_ctx *ctx, const vszip_plane *planes, int nplanes) {
    if (sizeof(T) != 1 || ctx->opt.ct_u8_px8) return false;
    using G16 = RingGeom<R, 4, 16, (HONLY ? 0 : R)>;
    using G8 = RingGeom<R, 2, 8, (HONLY ? 0 : R)>;
    double px = 0, t16 = 0, t8 = 0;
    for (int i = 0; i < nplanes; ++i) {
        const vszip_plane &p = planes[i];
        if (p.w < 32) return false;
        if ((reinterpret_cast<uintptr_t>(p.src) | reinterpret_cast<uintptr_t>(p.dst) | (uintptr_t)p.src_stride | (uintptr_t)p.dst_stride) & 15) return false;
        px += (double)p.w * p.h;
        t16 += (double)((p.w + G16::TWO - 1) / G16::TWO) * G16::TWO * p.h;
        t8 += (double)((p.w + G8::TWO - 1) / G8::TWO) * G8::TWO * p.h;
    }
    if (R >= 8 && px / t16 < 0.93 * (px / t8)) return false;
    // ... and only where its band plan fills its wave slots as the narrow form's does: two waves a SIMD with a third of the slots empty have nothing to hide their
    // latencies behind (1920 x 1088: 62-69 % of 2 048 slots, 22 % slower than 1920 x 1080 at 87 %, while the narrow form fills 2 816 of its 3 072 either way;
    // tools/u8_height_probe.py). The plans cost a few microseconds each: ring_plan searches over the batch's DISTINCT geometries — over all 192 planes it took
    // 50-100 us of host time a call, which launches of 160-300 us (1080p batches) did not hide: every such leg ran 7 % below what the device delivers.
    if (R >= 8) {
        const int n = std::min(nplanes, kRingMaxPlanes);
        std::vector<RingSpan> s16(n), s8(n);
        for (int i = 0; i < n; ++i) {
            s16[i] = {0, (planes[i].w + G16::TWO - 1) / G16::TWO};
            s8[i] = {0, (planes[i].w + G8::TWO - 1) / G8::TWO};
        }
        double w16 = 0, w8 = 0;
        ring_plan<T, R, 16, HONLY>(planes, s16.data(), n, &w16);
        ring_plan<T, R, 8, HONLY>(planes, s8.data(), n, &w8);
        const double f16 = w16 / (256.0 * 4 * G16::WPE), f8 = w8 / (256.0 * 4 * G8::WPE);
        if (f16 < 0.8 && f8 > f16 + 0.05) return false;
    }
    return true;
}

// The ring kernel over a batch of planes. The fast form needs whole pixel groups (8 or 16 samples) up to the right edge; a plane whose width is not such a
// multiple — cropped clips, 1366 x 768, 854 x 480, the 959-sample chroma of a 1918-wide clip — used to send the WHOLE batch to the general form (about half
// the rate, profiles/r04_cliff_sweep_before.txt). Round 4: such a plane is split by columns — the tiles whose every lane lies inside the plane (or in its left
// mirror image) go to the fast form with the aligned planes, the last one or two tiles to the general form in a second launch.
template <typename T, int R, int PXF, bool HONLY = false>
int run_ct_ring(vszip_ctx *ctx, const vszip_plane *planes, int nplanes) {
    using GF = typename RingWave<T, R, false, PXF, HONLY>::G;
    using GG = typename RingWave<T, R, true, 8, HONLY>::G;
    std::vector<RingSpan> fast(nplanes), rest(nplanes);
    bool any_rest = false, any_fast = false;
    for (int i = 0; i < nplanes; ++i) {
        const int w = planes[i].w;
        if (w % PXF == 0) {
            fast[i] = {0, (w + GF::TWO - 1) / GF::TWO};
            rest[i] = {0, 0};
        } else {
            const int room = w + GF::HL - 64 * PXF;  // tile t is all inside iff t * TWO - HL + 64 * PXF <= w
            const int nsafe = room >= 0 ? room / GF::TWO + 1 : 0;
            fast[i] = {0, nsafe};
            const int x = nsafe * GF::TWO;
            rest[i] = {x, (w - x + GG::TWO - 1) / GG::TWO};
        }
        any_fast = any_fast || fast[i].ntx > 0;
        any_rest = any_rest || rest[i].ntx > 0;
    }
    if (any_fast) {
        const int rc = launch_ct_ring_spans<T, R, PXF, false, HONLY>(ctx, planes, fast.data(), nplanes);
        if (rc != VSZIP_OK) return rc;
    }
    if (any_rest) return launch_ct_ring_spans<T, R, 8, true, HONLY>(ctx, planes, rest.data(), nplanes);
    return VSZIP_OK;
}

// Development builds (-DVSZIP_DEV_R=13) instantiate a single radius to keep the
// edit-compile-measure loop short; release builds carry all 22.
#ifdef VSZIP_DEV_R
#define VSZIP_R_ENABLED(R) ((R) == VSZIP_DEV_R)
#else
#define VSZIP_R_ENABLED(R) true
#endif

template <typename T, int R, int RLO>
struct CtIntDispatch {
    static int run(vszip_ctx *ctx, int r, const vszip_plane *planes, int nplanes) {
        if constexpr (VSZIP_R_ENABLED(R)) if (r == R) {
            if (ctx->scan_mode == 0 && ring_ok<T>(planes, nplanes)) {
                if constexpr (sizeof(T) == 1 && R <= kRing16MaxR) {
                    if (ring16_ok<T, R>(ctx, planes, nplanes)) return run_ct_ring<T, R, 16>(ctx, planes, nplanes);
                }
                return run_ct_ring<T, R, 8>(ctx, planes, nplanes);
            }
            return launch_ct_int<T, R>(ctx, planes, nplanes);
        }
        if constexpr (R > RLO)
            return CtIntDispatch<T, R - 1, RLO>::run(ctx, r, planes, nplanes);
        else
            return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: CT radius not built (development build?)");
    }
    // Round 4: a horizontal-only blur of radius r (one pass) through the ring kernel with a one-row window. The reference sends it down its run-time-radius path
    // (the radii differ, boxblur.zig:188), whose integer row pass is the same closed form as the compile-time one (blurInt :24-40 / hBlurInt :130-159: same start
    // value, same 16.16 constants, same edge-duplicating mirror), so the results are the RT kernels' bit for bit (tests/test_gpu_boxblur.py) — at the fused
    // kernel's rate instead of the row-at-a-time kernel's (8-bit 1080p: 0.82 -> 1.8 Tpx/s). VSZIP_ERR_UNSUPPORTED: not for these planes, take the RT path.
    static int run_h(vszip_ctx *ctx, int r, const vszip_plane *planes, int nplanes) {
        if constexpr (VSZIP_R_ENABLED(R)) if (r == R) {
            if (ctx->scan_mode != 0 || !ring_ok<T>(planes, nplanes)) return VSZIP_ERR_UNSUPPORTED;
            if constexpr (sizeof(T) == 1) {
                if (ring16_ok<T, R, true>(ctx, planes, nplanes)) return run_ct_ring<T, R, 16, true>(ctx, planes, nplanes);
            }
            return run_ct_ring<T, R, 8, true>(ctx, planes, nplanes);
        }
        if constexpr (R > RLO)
            return CtIntDispatch<T, R - 1, RLO>::run_h(ctx, r, planes, nplanes);
        else
            return VSZIP_ERR_UNSUPPORTED;
    }
};

}  // namespace
