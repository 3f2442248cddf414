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
#include "common.hpp"

namespace {

constexpr int kMaxPlanes = 48;  // planes per launch (kernel-argument table)
constexpr int PX = 8;           // pixels per lane per row

struct BBPlane {
    const void *src;
    void *dst;
    int sstride, dstride;  // elements
    int w, h;
    int block0;  // first block index of this plane
    int ntx;     // column tiles
};

struct BBParams {
    BBPlane p[kMaxPlanes];
    int nplanes;
    int band_rows;
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
            blocks += d.ntx * ((s.h + band - 1) / band);
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

template <typename T, int R>
struct CtIntDispatch {
    static int run(vszip_ctx *ctx, int r, const vszip_plane *planes, int nplanes) {
        if (r == R) return launch_ct_int<T, R>(ctx, planes, nplanes);
        return CtIntDispatch<T, R - 1>::run(ctx, r, planes, nplanes);
    }
};
template <typename T>
struct CtIntDispatch<T, 0> {
    static int run(vszip_ctx *ctx, int, const vszip_plane *, int) { return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: bad CT radius"); }
};

}  // namespace

VSZIP_EXPORT int vszip_boxblur(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, int hradius, int hpasses, int vradius, int vpasses) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    if (hradius < 0 || vradius < 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: negative radius");
    const bool vblur = (vradius > 0) && (vpasses > 0);
    const bool hblur = (hradius > 0) && (hpasses > 0);
    if (!vblur && !hblur) return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: nothing to be performed");  // boxblur.zig:152
    const bool use_rt = (hradius != vradius) || (hradius > 22) || (hpasses > 1) || (vpasses > 1);         // boxblur.zig:188
    for (int i = 0; i < nplanes; ++i) {
        const vszip_plane &p = planes[i];
        if (!p.src || !p.dst || p.w <= 0 || p.h <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: bad plane %d", i);
        // boxblur.zig:158-179; the CT path additionally needs both (the reference reads
        // out of bounds there when a pass count is 0 and the plane is tiny)
        if ((hblur || !use_rt) && (2L * hradius >= p.w))
            return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: hradius too large; 2*hradius must be < the (smallest processed) plane width.");
        if ((vblur || !use_rt) && (2L * vradius >= p.h))
            return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: vradius too large; 2*vradius must be < the (smallest processed) plane height.");
    }
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!use_rt) {
        switch (dtype) {
            case VSZIP_U8: return CtIntDispatch<uint8_t, 22>::run(ctx, hradius, planes, nplanes);
            case VSZIP_U16: return CtIntDispatch<uint16_t, 22>::run(ctx, hradius, planes, nplanes);
            default: return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "BoxBlur: CT float path not built yet");
        }
    }
    return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "BoxBlur: RT path not built yet");
}
