// vszip.BoxBlur, CT float path (f32 / f16): boxblur_comptime.zig:161-263 (vBlurFloat /
// hBlurFloat): both axes accumulate `acc + div * tap` over the 2r+1 taps IN TAP ORDER with the
// asymmetric mirror of mirrorRows (:50-70); unfused f32 (-ffp-contract=off), so results are
// bit-identical to the reference's.
//
// boxblur_ct_float_kernel<T, R>: a 256-thread workgroup stages a 64 x 32 output tile plus its
// halo in LDS, runs the vertical taps into a second LDS tile (stored as T, like the reference's
// tmp row) and then the horizontal taps. The radius is a template parameter: the tap loops
// unroll, and tiles that touch no plane border use compile-time tap offsets (the mirror index
// arithmetic costs more than the taps themselves) and are register blocked (8 outputs per
// work item share their 8 + 2r loads).
#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kMaxPlanesF = 192;  // planes per launch (round 4: 48 made a 64-frame 1080p call four launches, each planned as if it were the only one)

struct FPlane {
    const void *src;
    void *dst;
    int sstride, dstride, w, h;
    int rx0, ry0, rx1, ry1;  // the rectangle of the plane this entry tiles (the whole plane, or one border strip beside the ring kernel's interior)
    int block0, nbx;
};
struct FParams {
    FPlane p[kMaxPlanesF];
    int nplanes;
};

// boxblur_comptime.zig:50-70 — index of tap k for output index i (rows and columns alike)
__device__ __forceinline__ int ct_tap(int k, int i, int radius, int n) {
    const int dist_from_end = n - 1 - i;
    if (k < radius) return (i < radius - k) ? min(radius - k - i, n - 1) : (i - radius + k);
    return (dist_from_end < k - radius) ? (i - min(k - radius - dist_from_end, i)) : (i - radius + k);
}

constexpr int FTW = 64, FTH = 32;

template <typename T, int R>
__global__ __launch_bounds__(256) void boxblur_ct_float_kernel(const FParams prm) {
    constexpr int K = 2 * R + 1, IW = FTW + 2 * R, IH = FTH + 2 * R;
    __shared__ float tile[IH][IW + 1];
    __shared__ float vt[FTH][IW + 1];
    const int b = blockIdx.x;
    int pi = 0;  // the last plane whose first block is not beyond b
    for (int lo = 1, hi = prm.nplanes - 1; lo <= hi;) {
        const int mid = (lo + hi) >> 1;
        if (b >= prm.p[mid].block0) {
            pi = mid;
            lo = mid + 1;
        } else {
            hi = mid - 1;
        }
    }
    const FPlane pl = prm.p[pi];
    const int lb = b - pl.block0;
    const int w = pl.w, h = pl.h;
    const int x0 = pl.rx0 + (lb % pl.nbx) * FTW, y0 = pl.ry0 + (lb / pl.nbx) * FTH;
    const T *src = static_cast<const T *>(pl.src);
    T *dst = static_cast<T *>(pl.dst);
    const float div = 1.0f / (float)K;  // :39
    const int tid = threadIdx.x;
    const bool interior = x0 >= R && y0 >= R && x0 + FTW + R <= w && y0 + FTH + R <= h && x0 + FTW <= pl.rx1 && y0 + FTH <= pl.ry1;
    if (interior) {
        for (int i = tid; i < IH * IW; i += 256) {
            const int r = i / IW, c = i - r * IW;
            tile[r][c] = (float)src[(size_t)(y0 - R + r) * pl.sstride + x0 - R + c];
        }
        __syncthreads();
        // register blocked: a work item owns 8 consecutive outputs of a column (vertical pass) or of
        // a row (horizontal pass) and loads its 8 + 2r inputs once; each output still accumulates
        // its 2r+1 taps in tap order, unfused, so the arithmetic is the reference's
        constexpr int NB = 8;
        for (int i = tid; i < (FTH / NB) * IW; i += 256) {  // vBlurFloat :161-190
            const int c = i % IW, r0 = (i / IW) * NB;
            float v[NB + 2 * R];
#pragma unroll
            for (int j = 0; j < NB + 2 * R; ++j) v[j] = tile[r0 + j][c];
#pragma unroll
            for (int o = 0; o < NB; ++o) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) acc = acc + div * v[o + k];
                vt[r0 + o][c] = (float)(T)acc;  // tmp row is stored as T
            }
        }
        __syncthreads();
        {  // hBlurFloat :192-263: 32 rows x 8 strips of 8 outputs = 256 work items
            const int r = tid / (FTW / NB), c0 = (tid % (FTW / NB)) * NB;
            float v[NB + 2 * R];
#pragma unroll
            for (int j = 0; j < NB + 2 * R; ++j) v[j] = vt[r][c0 + j];
            T *drow = dst + (size_t)(y0 + r) * pl.dstride + x0 + c0;
#pragma unroll
            for (int o = 0; o < NB; ++o) {
                float sum = 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) sum += div * v[o + k];
                drow[o] = (T)sum;
            }
        }
        return;
    }
    // border tiles (and partial ones): only the rows / columns the tile's outputs tap are staged, and the
    // mirrored tap indices (boxblur_comptime.zig:50-70) are tabulated once per tile
    __shared__ short rtap[FTH][K], ctap[FTW][K];
    const int th = min(FTH, pl.ry1 - y0), tw = min(FTW, pl.rx1 - x0);
    const int cx0 = max(x0 - R, 0), cy0 = max(y0 - R, 0);
    const int cw = min(x0 + tw + R, w) - cx0, ch = min(y0 + th + R, h) - cy0;
    for (int i0 = tid; i0 < ch * cw; i0 += 4 * 256) {  // 4 loads in flight per work item
        T v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * 256, ch * cw - 1);
            const int r = i / cw, c = i - r * cw;
            v[u] = src[(size_t)(cy0 + r) * pl.sstride + cx0 + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * 256, ch * cw - 1);
            const int r = i / cw, c = i - r * cw;
            tile[r][c] = (float)v[u];
        }
    }
    // an axis whose taps meet no border on this tile runs the register-blocked pass with compile-time offsets
    // (a strip beside the ring kernel's interior mirrors on one axis only)
    constexpr int NBE = 8;
    const bool vplain = y0 >= R && y0 + th + R <= h && th == FTH;
    const bool hplain = x0 >= R && x0 + tw + R <= w && tw == FTW;
    if (!vplain)
        for (int i = tid; i < th * K; i += 256) rtap[i / K][i % K] = (short)(ct_tap(i % K, y0 + i / K, R, h) - cy0);
    if (!hplain)
        for (int i = tid; i < tw * K; i += 256) ctap[i / K][i % K] = (short)(ct_tap(i % K, x0 + i / K, R, w) - cx0);
    __syncthreads();
    if (vplain) {  // tile rows are y0 - R .. y0 + FTH + R - 1
        for (int i = tid; i < (FTH / NBE) * cw; i += 256) {
            const int c = i % cw, r0 = (i / cw) * NBE;
            float v[NBE + 2 * R];
#pragma unroll
            for (int j = 0; j < NBE + 2 * R; ++j) v[j] = tile[r0 + j][c];
#pragma unroll
            for (int o = 0; o < NBE; ++o) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) acc = acc + div * v[o + k];
                vt[r0 + o][c] = (float)(T)acc;
            }
        }
    } else {
        for (int i = tid; i < th * cw; i += 256) {
            const int r = i / cw, c = i - r * cw;
            float acc = 0.0f;
#pragma unroll 9
            for (int k = 0; k < K; ++k) acc = acc + div * tile[rtap[r][k]][c];
            vt[r][c] = (float)(T)acc;
        }
    }
    __syncthreads();
    if (hplain) {  // vt columns are x0 - R .. x0 + FTW + R - 1
        for (int i = tid; i < th * (FTW / NBE); i += 256) {
            const int r = i / (FTW / NBE), c0 = (i % (FTW / NBE)) * NBE;
            float v[NBE + 2 * R];
#pragma unroll
            for (int j = 0; j < NBE + 2 * R; ++j) v[j] = vt[r][c0 + j];
            T *drow = dst + (size_t)(y0 + r) * pl.dstride + x0 + c0;
#pragma unroll
            for (int o = 0; o < NBE; ++o) {
                float sum = 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) sum += div * v[o + k];
                drow[o] = (T)sum;
            }
        }
    } else {
        for (int i = tid; i < th * tw; i += 256) {
            const int r = i / tw, c = i - r * tw;
            float sum = 0.0f;
#pragma unroll 9
            for (int k = 0; k < K; ++k) sum += div * vt[r][ctap[c][k]];
            dst[(size_t)(y0 + r) * pl.dstride + x0 + c] = (T)sum;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// boxblur_ctf_ring_kernel<T, R>: the part of a plane whose taps meet no border (rows [ya, yb), whole column
// tiles whose halo is real data). One wave walks a 256-column tile down a band of rows, 4 pixels per lane
// (one 16-byte load per row for f32):
//   * the 2r+1 window rows stay in registers as PRODUCTS div * v (the reference multiplies every tap by div
//     before adding, :176; the product is the same for every output row that taps it), in a ring of
//     2r+1+D slots whose row loop is unrolled over one ring period, so every slot index is a compile-time
//     constant; rows are fetched D steps ahead into the slot that just left the window;
//   * each step sums its 2r+1 slots IN TAP ORDER (unfused adds: bit-identical to vBlurFloat :161-190),
//     rounds to T like the reference's tmp row, and parks div * tmp in 1 KiB of LDS laid out
//     [pixel-in-lane][lane] (conflict-free both ways);
//   * every output lane reads its 4 + 2r neighbours back and sums them in tap order (hBlurFloat :192-263).
// The rest of the plane (four border strips) goes through boxblur_ct_float_kernel.
// ---------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float;

#ifndef VSZIP_FR_ST_AUX
#define VSZIP_FR_ST_AUX 2
#endif
#ifndef VSZIP_FRD
#define VSZIP_FRD 7
#endif
constexpr int kFRD = VSZIP_FRD;    // rows in flight ahead of the window
constexpr int kFRPx = 4;           // pixels per lane and row
constexpr uint32_t kFROob = 0xfffffff0u;  // store offset of a lane / row that must not be written (the buffer descriptor drops it)
constexpr int kFRPix = 256, kFRRowB = 128;  // LDS: [pixel-in-lane] blocks of 256 floats, row A at +0, row B at +128, 8 floats of slack before lane 0
constexpr int kFRLds = kFRPx * kFRPix + 16;
constexpr int kFRMaxR = 22;         // larger radii: the ring (2r+1+D rows x 4 registers) no longer fits the register file with useful occupancy

template <int R>
struct FRGeom {
    static constexpr int K = 2 * R + 1, NR = K + kFRD;  // NR even: rows are processed in pairs
    static_assert((K + kFRD) % 2 == 0, "ring period must be even");
    static constexpr int HL = ((R + kFRPx - 1) / kFRPx) * kFRPx;  // halo columns each side, whole lanes
    static constexpr int OUT_LANES = 64 - 2 * (HL / kFRPx);
    static constexpr int TWO = OUT_LANES * kFRPx;                 // output columns per tile
};

struct FRPlane {
    const void *src;
    void *dst;
    int sstride, dstride;
    int ya, yb;          // rows of the main loops [ya, yb): yb = h - R, or h - NB when the kernel also does the bottom rows
    int h;
    int ylast;           // > 0: first row of the last band, which ends on a ring-period boundary and then does the bottom NB rows
    int xb;              // interior columns [0, xb)
    int ntx, nbands;
    int block0;
};
struct FRParams {
    FRPlane p[kMaxPlanesF];
    int nplanes, nblocks;
};

template <typename T>
struct FRow;
template <>
struct FRow<float> {
#ifdef VSZIP_FR_LD_NT
    static __device__ __forceinline__ void load(v4f &slot, const float *p) { slot = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)); }
#else
    static __device__ __forceinline__ void load(v4f &slot, const float *p) { slot = *reinterpret_cast<const v4f *>(p); }
#endif
    static __device__ __forceinline__ void to_products(v4f &slot, float div) { slot = div * slot; }
    typedef v4u raw_t;
    static __device__ __forceinline__ raw_t pack(v4f v) { return __builtin_bit_cast(v4u, v); }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rd, uint32_t vo, uint32_t so, raw_t v) {
        __builtin_amdgcn_raw_buffer_store_b128(v, rd, vo, so, VSZIP_FR_ST_AUX);  // aux 2 = nt
    }
    static __device__ __forceinline__ v4f narrow(v4f v) { return v; }
};
template <>
struct FRow<_Float16> {
    // the 8 loaded bytes wait in the slot's first two registers until the row enters the window
    static __device__ __forceinline__ void load(v4f &slot, const _Float16 *p) {
        const v2u r = *reinterpret_cast<const v2u *>(p);
        slot.x = __uint_as_float(r.x);
        slot.y = __uint_as_float(r.y);
    }
    static __device__ __forceinline__ void to_products(v4f &slot, float div) {
        v2u r = {__float_as_uint(slot.x), __float_as_uint(slot.y)};
        const v4h h = __builtin_bit_cast(v4h, r);
        slot = div * v4f{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    }
    typedef v2u raw_t;
    static __device__ __forceinline__ raw_t pack(v4f v) {
        const v4h h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        return __builtin_bit_cast(v2u, h);
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rd, uint32_t vo, uint32_t so, raw_t v) {
        __builtin_amdgcn_raw_buffer_store_b64(v, rd, vo, so, VSZIP_FR_ST_AUX);
    }
    static __device__ __forceinline__ v4f narrow(v4f v) {  // tmp row is stored as T (:186)
        return v4f{(float)(_Float16)v.x, (float)(_Float16)v.y, (float)(_Float16)v.z, (float)(_Float16)v.w};
    }
};

// The wave's state. Ring slots are addressed by template parameters (a period of NR steps is a fold
// expression), so every slot index is a compile-time constant whatever the optimizer's unroll budget.
template <typename T, int R>
struct FRWave {
    using G = FRGeom<R>;
    static constexpr int K = G::K, NR = G::NR, PXL = kFRPx;
    v4f ring[NR];  // slot of row rho = (rho - (y0 - R)) mod NR
    const T *sp;
    size_t ss;
    __amdgpu_buffer_rsrc_t rd;  // stores go through a buffer descriptor: halo lanes carry an out-of-range offset and the
    uint32_t sdoff, drow;       // hardware drops them, so a row is ONE unconditional store (no branch for the wait counters to be pessimistic about)
    float *pw, *lds0;  // this lane's LDS slot of row A, pixel 0; the LDS row itself (lane 0's slot)
    // first column tile of a plane: columns left of 0 are reflect-101 padding (:204-206) and the vertical pass is per
    // column, so their LDS entries are copies of this row's own entries of columns 1 .. HL - halo lanes fetch them
    // there (msrc[e]: LDS index of the mirror image of this lane's e-th column; mirror_lane: a halo lane of tile 0)
    bool left_tile, mirror_lane;
    int msrc[PXL];
    // LDS address of each neighbour lane's column block, kept opaque to the optimizer: with the lane offset folded
    // into the instruction it pairs reads of neighbouring lanes instead of rows A / B
    static constexpr int NBH = (R + PXL - 1) / PXL + 1;
    const lds_float *nb[2 * NBH + 1];
    float div;
    int last_row;
    int ystop;  // rows from here on are computed by the main loop's pairs but not stored

    template <int J>
    __device__ __forceinline__ void preload(int y0) {
        FRow<T>::load(ring[J], sp + (size_t)min(abs(y0 - R + J), last_row) * ss);  // rows above the plane: reflect-101 (:54)
    }
    template <int... J>
    __device__ __forceinline__ void preload_all(int y0, std::integer_sequence<int, J...>) {
        (preload<J>(y0), ...);
    }
    template <int... J>
    __device__ __forceinline__ void products_all(std::integer_sequence<int, J...>) {
        (FRow<T>::to_products(ring[J], div), ...);
    }
    template <int S, int... Kk>
    __device__ __forceinline__ v4f window_sum(std::integer_sequence<int, Kk...>) {
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
        ((acc = acc + ring[(S + Kk) % NR]), ...);  // tap order, unfused (:176)
        return acc;
    }
    // div * tmp of one output row goes to LDS row A (par 0) or B (par 1)
    __device__ __forceinline__ void put_row(v4f acc, int par) {
        const v4f ph = div * FRow<T>::narrow(acc);
        float *p = pw + par * kFRRowB;
        p[0 * kFRPix] = ph.x;
        p[1 * kFRPix] = ph.y;
        p[2 * kFRPix] = ph.z;
        p[3 * kFRPix] = ph.w;
    }
    // vertical pass of output row y + S (window = slots S .. S + K - 1 mod NR)
    template <int S>
    __device__ __forceinline__ void vstep(int y) {
        FRow<T>::to_products(ring[(S + K - 1) % NR], div);  // row y + S + R arrived D steps ago
        FRow<T>::load(ring[(S + NR - 1) % NR], sp + (size_t)min(y + S + R + kFRD, last_row) * ss);
        put_row(window_sum<S>(std::make_integer_sequence<int, K>{}), S & 1);
    }
    // Two output rows per horizontal pass (rows ya, ya + 1, already parked in LDS rows A / B): the packed
    // accumulators pair ROW A with ROW B of one column, and that pair (qA[j], qB[j]) is two LDS words 128 floats
    // apart - one two-address read, already in an aligned register pair (pairing neighbouring columns of one row needs
    // 45 register moves per row instead). Rows from `ystop` on are computed but not stored.
    __device__ __forceinline__ void hpass(int ya_row) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (left_tile) {  // (uniform)
            if (mirror_lane) {
                float a[PXL], bb[PXL];
#pragma unroll
                for (int e = 0; e < PXL; ++e) {
                    a[e] = lds0[msrc[e]];
                    bb[e] = lds0[msrc[e] + kFRRowB];
                }
#pragma unroll
                for (int e = 0; e < PXL; ++e) {
                    pw[e * kFRPix] = a[e];
                    pw[e * kFRPix + kFRRowB] = bb[e];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        v2f q[PXL + 2 * R];  // q[j] = column j - R relative to this lane's first one, rows A and B
#pragma unroll
        for (int j = 0; j < PXL + 2 * R; ++j) {
            const int c = j - R;
            const int k = ((c % PXL) + PXL) % PXL;         // pixel-in-lane of that column
            const lds_float *p = nb[(c - k) / PXL + NBH];  // ... of its owner lane
            q[j] = v2f{p[k * kFRPix], p[k * kFRPix + kFRRowB]};
        }
        v2f o[PXL];
#pragma unroll
        for (int e = 0; e < PXL; ++e) {
            o[e] = v2f{0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < K; ++k) o[e] = o[e] + q[e + k];  // tap order, unfused (:232)
        }
        // gfx950 reads the data registers of a 16-byte buffer store over the following cycles; a VALU write to them needs
        // 2 wait states, and the compiler's hazard recogniser does not provide them for stores whose soffset is an SGPR
        // (it assumes the hazard away there) - the last lanes of every 16 then store whatever was written next. So: both
        // rows' data are assembled in registers of their own FIRST (the empty asm pins them), then the two stores issue
        // back to back, then real wait states, and nothing is scheduled across.
        typename FRow<T>::raw_t ta = FRow<T>::pack(v4f{o[0].x, o[1].x, o[2].x, o[3].x}), tb = FRow<T>::pack(v4f{o[0].y, o[1].y, o[2].y, o[3].y});
        uint32_t va = ya_row < ystop ? sdoff : kFROob, vb = ya_row + 1 < ystop ? sdoff : kFROob;  // (the second offset's select landed in the first store's data register)
        asm volatile("" : "+v"(ta), "+v"(tb), "+v"(va), "+v"(vb));
        __builtin_amdgcn_sched_barrier(0);
        FRow<T>::store(rd, va, (uint32_t)ya_row * drow, ta);
        FRow<T>::store(rd, vb, (uint32_t)(ya_row + 1) * drow, tb);
        asm volatile("s_nop 3");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);  // a row pair is a scheduling region: the row loads stay D steps ahead of their use
    }
    template <int S>
    __device__ __forceinline__ bool step2(int y, int yend) {
        if (y + S >= yend) return false;  // (uniform) a band may end inside a ring period
        vstep<S>(y);
        vstep<S + 1>(y);
        hpass(y + S);
        return true;
    }

    // The last NB rows of the plane (NB = R or R + 1, even): taps past the bottom edge mirror about the CURRENT row
    // (boxblur_comptime.zig:61-66), so output row h - NB + J with d = NB - 1 - J rows below it sums rows i - R .. h - 1
    // and then i - 1, i - 2, .., i - (R - d). All of them sit in the ring when the band's main loop has ended on a
    // period boundary (row y_f = h - NB - 1 at step NR - 1: row y_f - R + k in slot (NR - 1 + k) mod NR), which makes
    // every slot of every sequence a compile-time constant.
    static constexpr int NB = (R % 2 == 0) ? R : R + 1;
    template <int J, int Kk>
    static constexpr int bottom_slot() {
        constexpr int d = NB - 1 - J;
        return Kk <= R + d ? (J + Kk) % NR : (R + J - (Kk - R - d) + NR) % NR;
    }
    template <int J, int... Kk>
    __device__ __forceinline__ v4f bottom_sum(std::integer_sequence<int, Kk...>) {
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
        ((acc = acc + ring[bottom_slot<J, Kk>()]), ...);
        return acc;
    }
    template <int P2>
    __device__ __forceinline__ void bottom_pair(int yb0) {
        put_row(bottom_sum<2 * P2>(std::make_integer_sequence<int, K>{}), 0);
        put_row(bottom_sum<2 * P2 + 1>(std::make_integer_sequence<int, K>{}), 1);
        hpass(yb0 + 2 * P2);
    }
    template <int... P2>
    __device__ __forceinline__ void bottom_rows(int yb0, std::integer_sequence<int, P2...>) {
        if constexpr (NB == R + 1) FRow<T>::to_products(ring[(NR - 1 + K) % NR], div);  // row h - 1 was fetched ahead of the last window
        (bottom_pair<P2>(yb0), ...);
    }
    template <int... S2>
    __device__ __forceinline__ void period(int y, int yend, std::integer_sequence<int, S2...>) {
        (void)(step2<2 * S2>(y, yend) && ...);
    }
};

template <typename T, int R>
__global__ __launch_bounds__(64) void boxblur_ctf_ring_kernel(const FRParams prm) {
    using W = FRWave<T, R>;
    using G = FRGeom<R>;
    constexpr int K = G::K, NR = G::NR, PXL = kFRPx;
    __shared__ float P[kFRLds];  // div * tmp of the current two rows, [pixel-in-lane][row A | row B][lane]
    const int chunk = (prm.nblocks + 7) >> 3;  // XCD-aware: blocks b and b+8 share an XCD
    const int b = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (b >= prm.nblocks) return;
    int pi = 0;  // the last plane whose first block is not beyond b
#pragma unroll 1
    for (int lo = 1, hi = prm.nplanes - 1; lo <= hi;) {
        const int mid = (lo + hi) >> 1;
        if (b >= prm.p[mid].block0) {
            pi = mid;
            lo = mid + 1;
        } else {
            hi = mid - 1;
        }
    }
    const FRPlane &pl = prm.p[pi];
    const int lb = b - pl.block0;
    const int tx = lb % pl.ntx, by = lb / pl.ntx;
    // Bands split the plane's row PAIRS evenly. Without the bottom rows (ylast == 0) an odd row count makes the last band
    // compute row yb as well - it belongs to the bottom strip, whose kernel runs afterwards. With them, the last band
    // is [ylast, yb) - a whole number of ring periods - and the others share [ya, ylast) (an odd count there makes the band
    // before the last one compute row ylast too: the same bits as the last band's).
    const bool bottom = pl.ylast > 0;
    const int nb_even = bottom ? pl.nbands - 1 : pl.nbands;
    const int y_even_end = bottom ? pl.ylast : pl.yb;
    const int npairs = (y_even_end - pl.ya + 1) / 2;
    const bool last_band = bottom && by == pl.nbands - 1;
    const int y0 = last_band ? pl.ylast : pl.ya + 2 * (int)((long)by * npairs / nb_even);
    const int y1 = last_band ? pl.yb : pl.ya + 2 * (int)((long)(by + 1) * npairs / nb_even);
    const int lane = threadIdx.x;
    // this lane's first column; the last tile is shifted left to end at xb, the first one starts HL columns left of
    // the plane (its halo lanes hold the reflect-101 images of columns 1 .. HL)
    const int cx = min(tx * G::TWO, pl.xb - G::TWO) - G::HL + lane * PXL;
    W st;
    st.left_tile = tx == 0;
    st.mirror_lane = cx < 0;
#pragma unroll
    for (int e = 0; e < PXL; ++e) {
        const int j = -(cx + e);            // mirror image of virtual column cx + e (reflect-101: -j -> j)
        const int pcol = max(j, 0) + G::HL;  // its position in the tile
        st.msrc[e] = 8 + (pcol % PXL) * kFRPix + pcol / PXL;
    }
    st.lds0 = P;
    st.sp = static_cast<const T *>(pl.src) + max(cx, 0);  // (halo lanes of the first tile load columns they overwrite in LDS)
    st.ss = (size_t)pl.sstride;
    st.drow = (uint32_t)pl.dstride * (uint32_t)sizeof(T);
    st.rd = __builtin_amdgcn_make_buffer_rsrc(pl.dst, 0, (int)((uint32_t)pl.h * st.drow), 0x00020000);
    const bool is_out = lane >= G::HL / PXL && lane < G::HL / PXL + G::OUT_LANES;
    st.sdoff = is_out ? (uint32_t)cx * (uint32_t)sizeof(T) : kFROob;  // (output lanes: cx >= 0)
    st.div = 1.0f / (float)K;  // :39
    st.last_row = pl.h - 1;  // rows fetched past it are never summed
    st.ystop = bottom ? pl.yb : 0x7fffffff;  // (with the bottom rows in the kernel a main-loop pair must not write into them)
    st.pw = P + 8 + lane;
#pragma unroll
    for (int i = 0; i < 2 * W::NBH + 1; ++i) {
        const lds_float *b = (const lds_float *)P + 8 + lane + (i - W::NBH);
        asm volatile("" : "+v"(b));
        st.nb[i] = b;
    }
    st.preload_all(y0, std::make_integer_sequence<int, NR - 1>{});
    st.products_all(std::make_integer_sequence<int, K - 1>{});
#pragma unroll 1
    for (int y = y0; y < y1; y += NR) st.period(y, y1, std::make_integer_sequence<int, NR / 2>{});
    if (last_band) {  // (uniform)
        st.ystop = pl.h;
        st.bottom_rows(pl.yb, std::make_integer_sequence<int, W::NB / 2>{});
    }
}

template <typename T, int R>
struct FloatDispatch {
    static void launch(vszip_ctx *ctx, int r, int blocks, const FParams &prm) {
        if (r == R)
            hipLaunchKernelGGL((boxblur_ct_float_kernel<T, R>), dim3(blocks), dim3(256), 0, ctx->stream, prm);
        else
            FloatDispatch<T, R - 1>::launch(ctx, r, blocks, prm);
    }
};
template <typename T>
struct FloatDispatch<T, 0> {
    static void launch(vszip_ctx *, int, int, const FParams &) {}
};

template <typename T, int R>
struct RingDispatch {
    static void launch(vszip_ctx *ctx, int r, int blocks, const FRParams &prm) {
        if (r == R)
            hipLaunchKernelGGL((boxblur_ctf_ring_kernel<T, R>), dim3(((blocks + 7) / 8) * 8), dim3(64), 0, ctx->stream, prm);
        else
            RingDispatch<T, R - 1>::launch(ctx, r, blocks, prm);
    }
    static void geom(int r, int &k, int &nr, int &hl, int &two) {
        if (r == R) {
            k = FRGeom<R>::K, nr = FRGeom<R>::NR, hl = FRGeom<R>::HL, two = FRGeom<R>::TWO;
        } else {
            RingDispatch<T, R - 1>::geom(r, k, nr, hl, two);
        }
    }
    // waves one resident round holds: what the register count of this radius leaves per CU, times the CUs
    static int round_waves(int r, int cus) {
        if (r != R) return RingDispatch<T, R - 1>::round_waves(r, cus);
        static int per_cu = 0;  // (benign race: every thread computes the same value)
        if (!per_cu) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, boxblur_ctf_ring_kernel<T, R>, 64, 0) != hipSuccess || n < 1) n = 8;
            per_cu = n;
        }
        return per_cu * cus;
    }
};
template <typename T>
struct RingDispatch<T, 0> {
    static void launch(vszip_ctx *, int, int, const FRParams &) {}
    static void geom(int, int &, int &, int &, int &) {}
    static int round_waves(int, int) { return 2048; }
};

// Interior of a plane for the ring kernel: columns [0, xb) (xb the last multiple of 4 that leaves HL columns of
// real halo), rows [R, h - R); false when the plane is too small or not 16-byte (f16: 8-byte) aligned - the tile
// kernel then takes all of it.
template <typename T>
bool ring_interior(const vszip_ctx *ctx, const vszip_plane &s, int radius, int nr, int hl, int two, int &ntx, int &xb) {
    constexpr uintptr_t VB = sizeof(T) * kFRPx;
    if (radius > kFRMaxR || ctx->opt.boxblur_no_float_ring) return false;
    if ((reinterpret_cast<uintptr_t>(s.src) | reinterpret_cast<uintptr_t>(s.dst) | (uintptr_t)(s.src_stride * sizeof(T)) | (uintptr_t)(s.dst_stride * sizeof(T))) & (VB - 1)) return false;
    xb = ((s.w - hl) / kFRPx) * kFRPx;
    ntx = (xb + two - 1) / two;
    if ((uint64_t)s.dst_stride * s.h * sizeof(T) >= (1ull << 32)) return false;  // 32-bit store offsets
    return xb >= two + hl && s.h - 2 * radius >= nr;
}

template <typename T>
int run_ct_float(vszip_ctx *ctx, const vszip_plane *planes, int nplanes, int radius) {
    if (radius < 1 || radius > 22) return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: CT float radius out of range");
    int K = 0, NR = 0, HL = 0, TWO = 0;
    RingDispatch<T, kFRMaxR>::geom(radius, K, NR, HL, TWO);
    vszip_probe_scope probe(ctx);
    // 1. interiors: one ring launch per kMaxPlanesF planes
    // bands per plane: the call's waves must fit ONE resident round (r = 13: 2 waves per SIMD = 2048; one wave
    // more starts a second round and the launch takes twice as long). Greedy: the planes with
    // the most rows per band get one more band each while the round has room; a band is at least one ring period
    // (every band re-reads 2r rows).
    const long round = RingDispatch<T, kFRMaxR>::round_waves(radius, ctx->num_cus > 0 ? ctx->num_cus : 256);
    std::vector<int> nb(nplanes, 0), ntxs(nplanes, 0), xbs(nplanes, 0);
    for (int i = 0; i < nplanes; ++i)
        if (ring_interior<T>(ctx, planes[i], radius, NR, HL, TWO, ntxs[i], xbs[i])) nb[i] = 1;
    // (planned per LAUNCH: the ring planes are taken kMaxPlanesF at a time below, and each launch is a resident round of its own)
    for (int g0 = 0; g0 < nplanes;) {
        int g1 = g0, cnt = 0;
        long waves = 0;
        for (; g1 < nplanes && cnt < kMaxPlanesF; ++g1)
            if (nb[g1]) {
                ++cnt;
                waves += ntxs[g1];
            }
        for (;;) {
            int worst = 0;
            for (int i = g0; i < g1; ++i)
                if (nb[i]) worst = std::max(worst, (planes[i].h - radius + nb[i] - 1) / nb[i]);
            if (worst == 0 || worst < 2 * NR) break;  // (worst == 0: no plane takes the ring kernel)
            long extra = 0;
            for (int i = g0; i < g1; ++i)
                if (nb[i] && (planes[i].h - radius + nb[i] - 1) / nb[i] == worst) extra += ntxs[i];
            if (waves + extra > round) break;
            for (int i = g0; i < g1; ++i)
                if (nb[i] && (planes[i].h - radius + nb[i] - 1) / nb[i] == worst) ++nb[i];
            waves += extra;
        }
        g0 = g1;
    }
    // Bottom rows inside the ring kernel: needs a last band of a whole number of ring periods ending at h - NB and at
    // least one band before it (otherwise the bottom strip goes through the tile kernel as the right one does)
    const int NB = (radius % 2 == 0) ? radius : radius + 1;
    std::vector<int> ylasts(nplanes, 0);
    if (!ctx->opt.boxblur_float_bottom_strip)
        for (int i = 0; i < nplanes; ++i) {
            if (nb[i] < 2) continue;
            const int main_rows = planes[i].h - NB;
            int m = (std::max(planes[i].h / nb[i] - NB, NR) + NR / 2) / NR;  // the last band also does the NB bottom rows: about as many rows as the others in all
            m = std::max(1, std::min(m, (main_rows - 2 * (nb[i] - 1)) / NR));
            if (m < 1 || main_rows - m * NR < 2 * (nb[i] - 1)) continue;
            ylasts[i] = main_rows - m * NR;
        }
    for (int done = 0; done < nplanes;) {
        FRParams prm;
        int n = 0, blocks = 0;
        for (; done < nplanes && n < kMaxPlanesF; ++done) {
            const vszip_plane &s = planes[done];
            if (!nb[done]) continue;
            FRPlane &d = prm.p[n++];
            d.src = s.src;
            d.dst = s.dst;
            d.sstride = (int)s.src_stride;
            d.dstride = (int)s.dst_stride;
            d.h = s.h;
            d.ya = 0;  // (the ring kernel mirrors the top rows itself)
            d.yb = s.h - radius;
            d.ylast = ylasts[done];
            if (d.ylast > 0) d.yb = s.h - NB;  // ... and the bottom rows, behind a last band that is a whole number of ring periods
            d.xb = xbs[done];
            d.ntx = ntxs[done];
            d.nbands = nb[done];
            d.block0 = blocks;
            blocks += d.ntx * d.nbands;
        }
        if (n) {
            prm.nplanes = n;
            prm.nblocks = blocks;
            RingDispatch<T, kFRMaxR>::launch(ctx, radius, blocks, prm);
            VSZIP_HIP_CHECK(ctx, hipGetLastError());
        }
    }
    // 2. everything else (border strips, or whole planes) through the tile kernel
    FParams prm;
    int n = 0, blocks = 0;
    auto flush = [&]() {
        if (!n) return;
        prm.nplanes = n;
        FloatDispatch<T, 22>::launch(ctx, radius, blocks, prm);
        n = blocks = 0;
    };
    auto add_rect = [&](const vszip_plane &s, int rx0, int ry0, int rx1, int ry1) {
        if (rx1 <= rx0 || ry1 <= ry0) return;
        if (n == kMaxPlanesF) flush();
        FPlane &d = prm.p[n++];
        d.src = s.src;
        d.dst = s.dst;
        d.sstride = (int)s.src_stride;
        d.dstride = (int)s.dst_stride;
        d.w = s.w;
        d.h = s.h;
        d.rx0 = rx0, d.ry0 = ry0, d.rx1 = rx1, d.ry1 = ry1;
        d.block0 = blocks;
        d.nbx = (rx1 - rx0 + FTW - 1) / FTW;
        blocks += d.nbx * ((ry1 - ry0 + FTH - 1) / FTH);
    };
    for (int i = 0; i < nplanes; ++i) {
        const vszip_plane &s = planes[i];
        int ntx, xb;
        if (ring_interior<T>(ctx, s, radius, NR, HL, TWO, ntx, xb)) {
            if (ylasts[i] > 0) {
                add_rect(s, xb, 0, s.w, s.h);  // right (the ring kernel does the top, left and bottom edges itself)
            } else {
                const int yb = s.h - radius;
                add_rect(s, 0, yb, s.w, s.h);  // bottom
                add_rect(s, xb, 0, s.w, yb);   // right
            }
        } else {
            add_rect(s, 0, 0, s.w, s.h);
        }
    }
    flush();
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    return VSZIP_OK;
}

}  // namespace

int vszip_bb_ct_float(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, int radius) {
    if (dtype == VSZIP_F16) return run_ct_float<_Float16>(ctx, planes, nplanes, radius);
    return run_ct_float<float>(ctx, planes, nplanes, radius);
}
