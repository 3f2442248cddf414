// vszip.BoxBlur — the paths besides the CT integer kernel:
//   * (CT float lives in boxblur_ctf.hip)
//   * RT integer (any radius, passes, hradius != vradius): boxblur_runtime.zig:10-41 blurInt,
//     one launch per pass and axis. The 16.16 running sum has the closed form
//         dst[x] = (inv2*E_x + 32768 + ((E_0*invlo) >> 16)) >> 16
//     with E_x the edge-duplicating mirrored window sum. Horizontal: one workgroup per row,
//     block-wide prefix sum of the row in LDS, E_x as prefix differences. Vertical: one thread
//     per column walks its rows with a sliding window sum (lanes = columns, coalesced).
//   * RT float: boxblur_runtime.zig:43-79 blurFloat keeps a RUNNING f32 sum whose rounding
//     depends on the visiting order, so every row (horizontal) / column (vertical) is walked
//     sequentially by one thread in exactly the reference's order: bit-identical, at the price
//     of parallelism = rows (columns) x planes. This path is not on any headline config.
#include <cstdlib>
#include <vector>

#include "common.hpp"

// boxblur.hip: one integer row pass of r <= 22 through the ring kernel (VSZIP_ERR_UNSUPPORTED: not for these planes)
int vszip_bb_ct_row_pass(vszip_ctx *ctx, int dtype, int r, const vszip_plane *planes, int nplanes);

namespace {

constexpr int kMaxPlanesRT = 192;  // planes per launch (round 4: 64 YUV frames are one launch per pass; 48 before)

struct RPlane {
    const void *src;
    void *dst;
    int sstride, dstride, w, h;
    int block0;
    int aux;  // banded integer chain: this plane's first entry in the E_0 table (u32 units)
};
struct RParams {
    RPlane p[kMaxPlanesRT];
    int nplanes;
    int radius;
    int keep;  // the output is the next pass's input: plain stores (it stays in L2 / the Infinity Cache); 0: streamed out with the nt hint
};

template <typename P>
__device__ __forceinline__ int rt_find(const P &prm, int b) {  // block0 ascends: eight scalar steps for 192 planes (a row-per-workgroup kernel pays this per row)
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

template <typename T>
__device__ __forceinline__ float ldf(const T *p) {
    return (float)*p;
}

// boxblur_comptime.zig:50-70 — index of tap k for output index i (rows and columns alike)
__device__ __forceinline__ int ct_tap(int k, int i, int radius, int n) {
    const int dist_from_end = n - 1 - i;
    if (k < radius) return (i < radius - k) ? min(radius - k - i, n - 1) : (i - radius + k);
    return (dist_from_end < k - radius) ? (i - min(k - radius - dist_from_end, i)) : (i - radius + k);
}

// ---- RT integer -----------------------------------------------------------------------------
// Horizontal: one workgroup per row. LDS holds the inclusive prefix of the row.
template <typename T>
__global__ __launch_bounds__(256) void boxblur_rt_hint_kernel(const RParams prm) {
    extern __shared__ __attribute__((aligned(16))) uint32_t P[];
    __shared__ uint32_t wsum[4];
    const int b = blockIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int y = b - pl.block0;
    const int w = pl.w, R = prm.radius;
    const T *s = static_cast<const T *>(pl.src) + (size_t)y * pl.sstride;
    T *d = static_cast<T *>(pl.dst) + (size_t)y * pl.dstride;
    const int tid = threadIdx.x;
    const int per = (w + 255) / 256;
    const int lo = min(tid * per, w), hi = min(lo + per, w);
    uint32_t run = 0;
    for (int x = lo; x < hi; ++x) {
        run += s[x];
        P[x] = run;
    }
    // block-wide exclusive scan of the 256 chunk totals
    uint32_t incl = wave_incl_scan_shfl(run);
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - run;
    for (int i = 0; i < (tid >> 6); ++i) base += wsum[i];
    for (int x = lo; x < hi; ++x) P[x] += base;
    __syncthreads();
    const uint32_t ksize = 2u * (uint32_t)R + 1u;
    const uint64_t inv = ((1ull << 32) + (uint64_t)R) / ksize;
    const uint32_t inv2 = (uint32_t)(inv >> 16), invlo = (uint32_t)(inv & 0xffffu);
    auto Q = [&](int c) -> uint32_t { return c < 0 ? 0u : P[min(c, w - 1)]; };
    const uint32_t e0 = Q(R) + Q(R - 1);  // srcp[r] + 2*sum_{x<r} srcp[x]
    const uint32_t kr = 32768u + (uint32_t)(((uint64_t)e0 * invlo) >> 16);
    for (int x = tid; x < w; x += 256) {
        // blurInt :24-40: taps left of 0 mirror as -k -> k-1, right of w-1 as w-1+k -> w-k
        uint32_t e = Q(min(x + R, w - 1)) - Q(x - R - 1);
        if (x - R - 1 < -1) e += Q(R - x - 1);
        if (x + R > w - 1) e += Q(w - 1) - Q(2 * w - 2 - x - R);
        d[x] = (T)(((uint64_t)e * inv2 + kr) >> 16);
    }
}

// Vertical: one thread per column, sliding window down the rows.
template <typename T>
__global__ __launch_bounds__(64) void boxblur_rt_vint_kernel(const RParams prm) {
    const int b = blockIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int x = (b - pl.block0) * 64 + threadIdx.x;
    if (x >= pl.w) return;
    const int len = pl.h, R = prm.radius;
    const T *s = static_cast<const T *>(pl.src) + x;
    T *d = static_cast<T *>(pl.dst) + x;
    const size_t ss = pl.sstride, ds = pl.dstride;
    const uint32_t ksize = 2u * (uint32_t)R + 1u;
    const uint64_t inv = ((1ull << 32) + (uint64_t)R) / ksize;
    const uint64_t inv2 = inv >> 16;
    // blurInt verbatim :10-41 (a u64 16.16 running sum is exact, so it IS the closed form)
    uint64_t sum = s[(size_t)R * ss];
    for (int i = 0; i < R; ++i) sum += (uint32_t)s[(size_t)i * ss] << 1;
    sum = (sum * inv + (1ull << 31)) >> 16;
    int i = 0;
    for (; i <= R; ++i) {
        sum += s[(size_t)(R + i) * ss] * inv2;
        sum -= s[(size_t)(R - i) * ss] * inv2;
        d[(size_t)i * ds] = (T)(sum >> 16);
    }
    for (; i < len - R; ++i) {
        sum += s[(size_t)(R + i) * ss] * inv2;
        sum -= s[(size_t)(i - R - 1) * ss] * inv2;
        d[(size_t)i * ds] = (T)(sum >> 16);
    }
    for (; i < len; ++i) {
        sum += s[(size_t)(2 * len - R - i - 1) * ss] * inv2;
        sum -= s[(size_t)(i - R - 1) * ss] * inv2;
        d[(size_t)i * ds] = (T)(sum >> 16);
    }
}


// ---- RT integer, vectorised (rows 16-byte aligned: every VapourSynth frame) ---------------------
// The 16.16 running sum of blurInt (:10-41) has the closed form
//     dst[i] = (inv2 * E_i + 32768 + ((E_0 * invlo) >> 16)) >> 16,
// E_i the edge-duplicating mirrored window sum (index j < 0 -> -j-1, j >= len -> 2*len-1-j), so a
// line can be cut anywhere: only E at the cut and the line's E_0 are needed.
template <typename T>
struct RtVec {
    static constexpr int V = 16 / (int)sizeof(T);
    // raw(): the 16-byte load alone, so that a caller can put several rows in flight before it
    // unpacks the first (a vertical walk is otherwise one global round trip per row)
    static __device__ __forceinline__ uint4 raw(const T *p) { return *reinterpret_cast<const uint4 *>(p); }
    static __device__ __forceinline__ void load(const T *p, uint32_t v[V]) { unpack(raw(p), v); }
    static __device__ __forceinline__ void unpack(const uint4 q, uint32_t v[V]) {
        const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if constexpr (sizeof(T) == 2)
                v[k] = (d[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
            else
                v[k] = (d[k >> 2] >> ((k & 3) * 8)) & 0xffu;
        }
    }
    static __device__ __forceinline__ void store(T *p, const uint32_t v[V], int n, int keep = 0) {
        if (n >= V) {
            uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < V; ++k) {
                if constexpr (sizeof(T) == 2)
                    d[k >> 1] |= v[k] << ((k & 1) * 16);
                else
                    d[k >> 2] |= v[k] << ((k & 3) * 8);
            }
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 q = {d[0], d[1], d[2], d[3]};
            if (keep)
                *reinterpret_cast<u32x4 *>(p) = q;  // an intermediate pass: the next pass reads it back from cache
            else
                __builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(p));  // streamed output: keep it out of L2
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k)
                if (k < n) p[k] = (T)v[k];
        }
    }
    // eight values (hsmall: a lane's group of eight samples, for 8-bit clips half a 16-byte vector)
    static __device__ __forceinline__ void store8(T *p, const uint32_t v[8], int n, int keep = 0) {
        if (n >= 8) {
            if constexpr (sizeof(T) == 2) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 q = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
                if (keep)
                    *reinterpret_cast<u32x4 *>(p) = q;
                else
                    __builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(p));
            } else {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 q = {v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24), v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24)};
                if (keep)
                    *reinterpret_cast<u32x2 *>(p) = q;
                else
                    __builtin_nontemporal_store(q, reinterpret_cast<u32x2 *>(p));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < n) p[k] = (T)v[k];
        }
    }
};

struct RVParams {
    RPlane p[kMaxPlanesRT];
    int ncg[kMaxPlanesRT];  // column groups (64 lanes x V columns) per plane
    int nplanes, radius, band;
    int keep;
};

// Vertical: lane = V adjacent columns, one wave per (column group, band of rows). E_0 and the
// window sum at the top of the band are summed directly, then the window slides (one entering and
// one leaving row vector per output row).
// RING (round 3): the leaving row is the one that entered 2R + 1 steps earlier. Re-reading it from memory made the pass fetch 2.2x its
// input from HBM (PMC: 444 MB per 199 MB launch — 3 000 waves x (2R + 1) KiB outlive the L2s); the wave now keeps its last 2R + 2 raw row
// vectors in LDS ([slot = row mod D][lane], 16 bytes per lane: conflict free, and private to the lane — no barrier) and takes the
// leaving row from there. D KiB per wave: used up to R = 8 (18 KiB: eight waves a CU) — at R = 13 (28 KiB: five waves) the README's 5 + 5-pass
// bench lost 11 % to it (22.1 k -> 19.6 k fps in an interleaved A/B, tools/rt_ring_ab.py) — else the re-reading form.
constexpr int kVRingMaxR = 8;
template <typename T, bool RING>
__global__ __launch_bounds__(64) void boxblur_rt_vband_kernel(const RVParams prm) {
    using X = RtVec<T>;
    constexpr int V = X::V;
    extern __shared__ __attribute__((aligned(16))) uint4 vring[];
    const int D = 2 * prm.radius + 2;
    auto slot = [&](int row) -> uint4 & { return vring[(row % D) * 64 + (int)threadIdx.x]; };
    const int b = blockIdx.x;
    const int pi = rt_find(prm, b);
    const RPlane pl = prm.p[pi];
    const int lb = b - pl.block0, ncg = prm.ncg[pi];
    const int x0 = ((lb % ncg) * 64 + (int)threadIdx.x) * V;
    if (x0 >= pl.w) return;
    const int len = pl.h, R = prm.radius;
    const int y0 = (lb / ncg) * prm.band, y1 = min(y0 + prm.band, len);
    const T *s = static_cast<const T *>(pl.src) + x0;
    T *d = static_cast<T *>(pl.dst) + x0;
    const size_t ss = pl.sstride, ds = pl.dstride;
    const int nst = min(V, pl.w - x0);
    const uint32_t ksize = 2u * (uint32_t)R + 1u;
    const uint64_t inv = ((1ull << 32) + (uint64_t)R) / ksize;
    const uint32_t inv2 = (uint32_t)(inv >> 16), invlo = (uint32_t)(inv & 0xffffu);
    auto mrow = [&](int j) { return j < 0 ? -j - 1 : (j >= len ? 2 * len - 1 - j : j); };

    // Rows are fetched kPf at a time (raw 16-byte loads, all issued before the first is unpacked): the
    // walk down a band is a chain of dependent adds, not of memory round trips.
    constexpr int kPf = 8;
    uint32_t e0[V], e[V], kk[V], t[V];
#pragma unroll
    for (int k = 0; k < V; ++k) e0[k] = 0;
    for (int r0 = 0; r0 <= R; r0 += kPf) {  // E_0 = s[R] + 2 * sum_{i<R} s[i]
        uint4 q[kPf];
#pragma unroll
        for (int u = 0; u < kPf; ++u) q[u] = X::raw(s + (size_t)min(r0 + u, R) * ss);
#pragma unroll
        for (int u = 0; u < kPf; ++u) {
            const int r = r0 + u;
            if (r <= R) {
                X::unpack(q[u], t);
#pragma unroll
                for (int k = 0; k < V; ++k) e0[k] += r < R ? 2u * t[k] : t[k];
                if constexpr (RING)
                    if (y0 == 0) slot(r) = q[u];  // the first band's window is rows 0 .. R (and their mirror images)
            }
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) kk[k] = 32768u + (uint32_t)(((uint64_t)e0[k] * invlo) >> 16);
    if (y0 == 0) {
#pragma unroll
        for (int k = 0; k < V; ++k) e[k] = e0[k];
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) e[k] = 0;
        for (int j0 = y0 - R; j0 <= y0 + R; j0 += kPf) {
            uint4 q[kPf];
#pragma unroll
            for (int u = 0; u < kPf; ++u) q[u] = X::raw(s + (size_t)mrow(min(j0 + u, y0 + R)) * ss);
#pragma unroll
            for (int u = 0; u < kPf; ++u) {
                if (j0 + u <= y0 + R) {
                    X::unpack(q[u], t);
#pragma unroll
                    for (int k = 0; k < V; ++k) e[k] += t[k];
                    if constexpr (RING) slot(mrow(j0 + u)) = q[u];  // (a mirrored row lands on the slot of the row it mirrors: same data)
                }
            }
        }
    }
    const bool mul24 = (uint64_t)(sizeof(T) == 1 ? 255u : 65535u) * ksize < (1u << 24);
    constexpr int kPm = RING ? 8 : 4;  // output rows per chunk of the sliding part: 8 loads in flight either way
    // RING: software pipeline — the next chunk's entering rows are requested before the current chunk is worked through (the leaving rows
    // come from LDS, so a chunk costs kPm loads and the registers hold two chunks)
    uint4 qn[kPm];
    if constexpr (RING) {
#pragma unroll
        for (int u = 0; u < kPm; ++u) qn[u] = X::raw(s + (size_t)mrow(min(y0 + u, y1 - 1) + 1 + R) * ss);
    }
    for (int i0 = y0; i0 < y1; i0 += kPm) {
        uint4 qa[kPm], qc[kPm];
#pragma unroll
        for (int u = 0; u < kPm; ++u) {
            const int i = min(i0 + u, y1 - 1);
            if constexpr (RING) {
                qa[u] = qn[u];
            } else {
                qa[u] = X::raw(s + (size_t)mrow(i + 1 + R) * ss);
                qc[u] = X::raw(s + (size_t)mrow(i - R) * ss);
            }
        }
        if constexpr (RING) {
            if (i0 + kPm < y1) {
#pragma unroll
                for (int u = 0; u < kPm; ++u) qn[u] = X::raw(s + (size_t)mrow(min(i0 + kPm + u, y1 - 1) + 1 + R) * ss);
            }
        }
#pragma unroll
        for (int u = 0; u < kPm; ++u) {
            const int i = i0 + u;
            if (i < y1) {
                uint32_t o[V], a[V], c[V];
                if constexpr (RING) {
                    qc[u] = slot(mrow(i - R));                   // read the leaving row first:
                    if (i + 1 + R < len) slot(i + 1 + R) = qa[u];  // the entering one may take a slot only rows before it used (D = 2R + 2)
                }
                X::unpack(qa[u], a);
                X::unpack(qc[u], c);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    // the 16.16 value is the window mean + 0.5: 32 bits hold it (a 24-bit multiply while the window sum is below 2^24)
                    o[k] = ((mul24 ? (uint32_t)__umul24(e[k], inv2) : e[k] * inv2) + kk[k]) >> 16;
                    e[k] += a[k] - c[k];
                }
                X::store(d + (size_t)i * ds, o, nst, prm.keep);
            }
        }
    }
}

// Horizontal: one wave per row. Pass 1 builds the inclusive prefix of the row in LDS (in-lane
// prefix + DPP wave scan + a running carry), laid out [chunk][pixel-in-lane][lane] so that both
// passes are bank-conflict free; pass 2 forms E_x as prefix differences and stores.
template <typename T>
__global__ __launch_bounds__(64) void boxblur_rt_hrow_kernel(const RParams prm) {
    using X = RtVec<T>;
    constexpr int V = X::V, CH = 64 * V;
    extern __shared__ __attribute__((aligned(16))) uint32_t P[];
    const int b = blockIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int y = b - pl.block0;
    const int w = pl.w, R = prm.radius;
    const T *s = static_cast<const T *>(pl.src) + (size_t)y * pl.sstride;
    T *d = static_cast<T *>(pl.dst) + (size_t)y * pl.dstride;
    const int lane = threadIdx.x;
    const int nch = (w + CH - 1) / CH;
    auto pidx = [&](uint32_t c) { return (c / (uint32_t)CH) * (uint32_t)CH + (c % (uint32_t)V) * 64u + (c % (uint32_t)CH) / (uint32_t)V; };  // unsigned: shifts and masks
    uint32_t carry = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const int x0 = ch * CH + lane * V;
        uint32_t v[V];
        if (x0 < w) {
            X::load(s + x0, v);  // [w, stride) is readable padding; masked below
#pragma unroll
            for (int k = 0; k < V; ++k)
                if (x0 + k >= w) v[k] = 0;
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = 0;
        }
#pragma unroll
        for (int k = 1; k < V; ++k) v[k] += v[k - 1];
        const uint32_t incl = wave_incl_scan_dpp(v[V - 1]);
        const uint32_t base = carry + incl - v[V - 1];
#pragma unroll
        for (int k = 0; k < V; ++k) P[ch * CH + k * 64 + lane] = v[k] + base;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    __syncthreads();
    const uint32_t ksize = 2u * (uint32_t)R + 1u;
    const uint64_t inv = ((1ull << 32) + (uint64_t)R) / ksize;
    const uint32_t inv2 = (uint32_t)(inv >> 16), invlo = (uint32_t)(inv & 0xffffu);
    auto Q = [&](int c) -> uint32_t { return c < 0 ? 0u : P[pidx((uint32_t)min(c, w - 1))]; };
    const uint32_t e0 = Q(R) + Q(R - 1);  // srcp[r] + 2*sum_{x<r} srcp[x]
    const uint32_t kr = 32768u + (uint32_t)(((uint64_t)e0 * invlo) >> 16);
    for (int ch = 0; ch < nch; ++ch) {
        const int x0 = ch * CH + lane * V;
        // chunks whose windows stay inside the row need no mirror terms (wave-uniform test)
        const bool inner = ch * CH - R - 1 >= 0 && ch * CH + CH - 1 + R <= w - 1;
        if (x0 >= w) break;
        uint32_t o[V];
        if (inner) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const uint32_t x = (uint32_t)(x0 + k);
                const uint32_t e = P[pidx(x + (uint32_t)R)] - P[pidx(x - (uint32_t)R - 1u)];
                o[k] = (uint32_t)(((uint64_t)e * inv2 + kr) >> 16);
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const int x = x0 + k;
                // blurInt :24-40: taps left of 0 mirror as -k -> k-1, right of w-1 as w-1+k -> w-k
                uint32_t e = Q(min(x + R, w - 1)) - Q(x - R - 1);
                if (x - R - 1 < -1) e += Q(R - x - 1);
                if (x + R > w - 1) e += Q(w - 1) - Q(2 * w - 2 - x - R);
                o[k] = (uint32_t)(((uint64_t)e * inv2 + kr) >> 16);
            }
        }
        X::store(d + x0, o, min(V, w - x0), prm.keep);
    }
}


// Horizontal, single pass (radius < one chunk of 64 * V columns): the prefix lives in a ring of
// four chunks in LDS instead of the whole row — 8 KiB (u16) per wave instead of 4 bytes per
// column, so more waves per CU — and the row is read once: chunk c+1 is prefixed while chunk c
// is emitted from the chunks c-1, c, c+1 that its windows (and the mirror terms at the row ends)
// can reach. Four slots, not three: the slot of a column is then two bits of its index, and the
// LDS address of "column x + r" / "column x - r - 1" of a lane is a per-lane constant that moves by
// one chunk per step — an add and a mask per access instead of a division by three.
//
// VIRT (every row a whole number of lane groups, both halos narrower than the row): the prefix runs over the
// VIRTUAL row [mirror image of columns 0..HL-1 | the row | mirror image of the last HR columns] (blurInt's
// implicit padding: -k -> k-1, w-1+k -> w-k; a halo lane loads the real group it mirrors and reverses it in
// registers), so every window — at the row ends too — is the same two-term prefix difference. Without it the two end
// chunks of a row (a third of a 4K frame's chunks) evaluate five clamped terms per pixel: 190 VALU instructions per
// chunk on average against about 100.
template <typename T, bool VIRT>
__global__ __launch_bounds__(64) void boxblur_rt_hring_kernel(const RParams prm) {
    using X = RtVec<T>;
    constexpr int V = X::V, CH = 64 * V;
    constexpr uint32_t kRing = 4u * CH;
    constexpr int LV = V == 8 ? 3 : 4, LCH = LV + 6;
    static_assert((1 << LV) == V && (1 << LCH) == CH, "chunk geometry");
    __shared__ __attribute__((aligned(16))) uint32_t P[kRing];
    const int b = blockIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int y = b - pl.block0;
    const int w = pl.w, R = prm.radius;
    const T *s = static_cast<const T *>(pl.src) + (size_t)y * pl.sstride;
    T *d = static_cast<T *>(pl.dst) + (size_t)y * pl.dstride;
    const int lane = threadIdx.x;
    // virtual row: HL mirrored columns, the row, HR mirrored columns (both 0 without VIRT)
    const int HL = VIRT ? ((R + V) / V) * V : 0, HR = VIRT ? ((R + V - 1) / V) * V : 0;
    const int vw = HL + w + HR;
    const int nch = (vw + CH - 1) / CH;
    // column c -> [slot = chunk & 3][pixel-in-lane][lane]: conflict-free for a fixed pixel index
    auto pidx = [&](uint32_t c) { return (c & (kRing - 1u) & ~(uint32_t)(CH - 1)) | ((c & (uint32_t)(V - 1)) << 6) | ((c & (uint32_t)(CH - 1)) >> LV); };
    uint32_t carry = 0;
    // the row is fetched one chunk ahead of its prefix (a load issued and scanned in the same step
    // would put a global round trip on every chunk of the row's chain)
    auto fetch = [&](int ch) -> uint4 {
        const int v0 = ch * CH + lane * V;
        if (ch >= nch || v0 >= vw) return make_uint4(0, 0, 0, 0);  // (without VIRT: [w, stride) is readable padding; masked below)
        int g = v0 - HL;
        if constexpr (VIRT) g = g < 0 ? -g - V : (g >= w ? 2 * w - g - V : g);
        return X::raw(s + g);
    };
    auto prefix_chunk = [&](int ch, uint4 q) {
        const int x0 = ch * CH + lane * V;
        if constexpr (VIRT) {
            if (ch * CH < HL || ch * CH + CH > HL + w) {  // (wave-uniform: the chunk holds halo lanes)
                const int c0 = x0 - HL;
                if (c0 < 0 || c0 >= w) {  // mirrored lanes: pixel order reversed
                    const uint4 o = q;
                    if constexpr (sizeof(T) == 2) {
                        q.x = __builtin_amdgcn_alignbit(o.w, o.w, 16);
                        q.y = __builtin_amdgcn_alignbit(o.z, o.z, 16);
                        q.z = __builtin_amdgcn_alignbit(o.y, o.y, 16);
                        q.w = __builtin_amdgcn_alignbit(o.x, o.x, 16);
                    } else {
                        q.x = __builtin_amdgcn_perm(o.w, o.w, 0x00010203u);
                        q.y = __builtin_amdgcn_perm(o.z, o.z, 0x00010203u);
                        q.z = __builtin_amdgcn_perm(o.y, o.y, 0x00010203u);
                        q.w = __builtin_amdgcn_perm(o.x, o.x, 0x00010203u);
                    }
                }
            }
        }
        uint32_t v[V];
        X::unpack(q, v);
        if constexpr (!VIRT) {
            if (ch * CH + CH > w) {  // (wave-uniform: only the last chunk holds columns past the row)
#pragma unroll
                for (int k = 0; k < V; ++k)
                    if (x0 + k >= w) v[k] = 0;
            }
        }
#pragma unroll
        for (int k = 1; k < V; ++k) v[k] += v[k - 1];
        const uint32_t incl = wave_incl_scan_dpp(v[V - 1]);
        const uint32_t base = carry + incl - v[V - 1];
        uint32_t *slot = P + (ch & 3) * CH;
#pragma unroll
        for (int k = 0; k < V; ++k) slot[k * 64 + lane] = v[k] + base;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    };
    const uint4 q0 = fetch(0), q1 = fetch(1);
    uint4 qn = fetch(2);  // raw samples of the chunk the loop scans next (fetching four chunks ahead instead of one: -3 %)
    prefix_chunk(0, q0);
    if (nch > 1) prefix_chunk(1, q1);
    vszip_wave_fence();
    const uint32_t ksize = 2u * (uint32_t)R + 1u;
    const uint64_t inv = ((1ull << 32) + (uint64_t)R) / ksize;
    const uint32_t inv2 = (uint32_t)(inv >> 16), invlo = (uint32_t)(inv & 0xffffu);
    auto Q = [&](int c) -> uint32_t { return c < 0 ? 0u : P[pidx((uint32_t)min(c, w - 1))]; };
    // srcp[r] + 2*sum_{x<r} srcp[x] (r < CH: chunks 0 / 1); real prefix Q(c) = virtual prefix at c + HL minus the left halo's sum
    uint32_t e0;
    if constexpr (VIRT)
        e0 = P[pidx((uint32_t)(R + HL))] + P[pidx((uint32_t)(R - 1 + HL))] - 2u * P[pidx((uint32_t)(HL - 1))];
    else
        e0 = Q(R) + Q(R - 1);
    const uint32_t kr = 32768u + (uint32_t)(((uint64_t)e0 * invlo) >> 16);
    // (e * inv2 + kr) >> 16 in 32 bits: the 16.16 value is the window mean + 0.5, at most 65535.5 * 65536 (blurInt
    // :24-40); a 24-bit multiply serves while the window sum itself stays below 2^24
    constexpr uint32_t kPeak = sizeof(T) == 1 ? 255u : 65535u;
    const bool mul24 = (uint64_t)kPeak * ksize < (1u << 24);
    auto scale = [&](uint32_t e) -> uint32_t { return ((mul24 ? (uint32_t)__umul24(e, inv2) : e * inv2) + kr) >> 16; };
    // LDS addresses of this lane's window ends in chunk 0's frame; chunk ch adds ch * CH, modulo the ring
    uint32_t a_hi[V], a_lo[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        a_hi[k] = pidx((uint32_t)(lane * V + k + R));
        a_lo[k] = pidx((uint32_t)(lane * V + k - R - 1) + kRing);
    }
    for (int ch = 0; ch < nch; ++ch) {
        if (ch >= 1 && ch + 1 < nch) {  // chunk ch+1 takes the slot of chunk ch-3, which no window reaches any more
            const uint4 cur = qn;
            qn = fetch(ch + 2);
            prefix_chunk(ch + 1, cur);
            vszip_wave_fence();
        }
        const int x0 = ch * CH + lane * V - HL;  // real column of this lane's group
        const bool inner = VIRT || (ch * CH - R - 1 >= 0 && ch * CH + CH - 1 + R <= w - 1);
        if (x0 >= 0 && x0 < w) {
            uint32_t o[V];
            if (inner) {
                const uint32_t adv = (uint32_t)ch << LCH;
#pragma unroll
                for (int k = 0; k < V; ++k) o[k] = scale(P[(a_hi[k] + adv) & (kRing - 1u)] - P[(a_lo[k] + adv) & (kRing - 1u)]);
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const int x = x0 + k;
                    // blurInt :24-40: taps left of 0 mirror as -k -> k-1, right of w-1 as w-1+k -> w-k
                    uint32_t e = Q(min(x + R, w - 1)) - Q(x - R - 1);
                    if (x - R - 1 < -1) e += Q(R - x - 1);
                    if (x + R > w - 1) e += Q(w - 1) - Q(2 * w - 2 - x - R);
                    o[k] = scale(e);
                }
            }
            X::store(d + x0, o, min(V, w - x0), prm.keep);
        }
        vszip_wave_fence();  // the next iteration overwrites a ring slot
    }
}

#ifndef VSZIP_RTF_PF
#define VSZIP_RTF_PF 32
#endif
#ifndef VSZIP_RTF_LW
#define VSZIP_RTF_LW 64
#endif
#ifndef VSZIP_RTF_RB
#define VSZIP_RTF_RB 16
#endif
// One-wave workgroups: LDS operations of a wave execute in order, so a hand-over between lanes needs the compiler to keep the order and nothing else —
// __syncthreads() would also wait for every load and store in flight (s_waitcnt vmcnt(0)): the end of all prefetching.
__device__ __forceinline__ void fc_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- RT float: blurFloat (:43-79) — a running f32 sum per line, sequential by definition ------
// Index of the sample entering / leaving the window at output x (the three loops of blurFloat :55-78
// as one pair of functions: x <= R, R < x < len - R, x >= len - R).
__device__ __forceinline__ int rtf_in(int x, int len, int R) { return (x > R && x >= len - R) ? 2 * len - R - x - 1 : R + x; }
__device__ __forceinline__ int rtf_out(int x, int R) { return x <= R ? R - x : x - R - 1; }

// Vertical float pass: lane = column (coalesced), the running sum walks down the rows. The samples
// of kPf output rows (entering and leaving) are loaded before the first of them is added, so a step
// of the chain is one f32 add, not a global round trip.
template <typename T>
__global__ __launch_bounds__(64) void boxblur_rt_float_v_kernel(const RParams prm) {
    const int b = blockIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    // kLW columns per wave. Measured on 8 4K YUV420PS frames, 3+3 passes of r = 5 (k fps): 64 -> 5.9, 32 -> 5.2,
    // 16 -> 3.6; a variant with 16-byte loads (4 columns per lane, a quarter of the waves) 5.6: the pass is bound by
    // the number of row requests in flight, not by a wave's chain.
    constexpr int kLW = VSZIP_RTF_LW;
    const int i = (b - pl.block0) * kLW + threadIdx.x;
    if ((int)threadIdx.x >= kLW || i >= pl.w) return;
    const T *s = static_cast<const T *>(pl.src) + i;
    T *d = static_cast<T *>(pl.dst) + i;
    const size_t ss = pl.sstride, ds = pl.dstride;
    const int len = pl.h, R = prm.radius;
    const float div = 1.0f / (float)(R * 2 + 1);
    // columns are the only parallelism a running sum leaves (one wave per 64 columns: about one wave per SIMD on
    // 8 4K frames), so the chain must never wait on memory: 2 x kPf rows (entering + leaving) are in flight per
    // wave while the previous kPf are summed - 8 rows left the launch latency bound at 1.2 TB/s
    constexpr int kPf = VSZIP_RTF_PF;
    float sum = 0.0f;
    for (int x0 = 0; x0 <= R; x0 += kPf) {  // :47-49 — sum = s[R] + 2 * s[0] + 2 * s[1] + ..., in that order
        float q[kPf];
#pragma unroll
        for (int u = 0; u < kPf; ++u) q[u] = (float)s[(size_t)(x0 + u == 0 ? R : min(x0 + u - 1, R - 1 >= 0 ? R - 1 : 0)) * ss];
#pragma unroll
        for (int u = 0; u < kPf; ++u) {
            const int k = x0 + u;  // term k: k == 0 is s[R], k >= 1 is s[k-1] * 2
            if (k == 0)
                sum = q[u];
            else if (k <= R)
                sum += q[u] * 2;
        }
    }
    sum = sum * div;
    // two groups in flight: the samples of rows x0 + kPf .. are requested before rows x0 .. are summed
    float qi[kPf], qo[kPf], ni[kPf], no[kPf];
    auto fetch = [&](int x0, float *fi, float *fo) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kPf; ++u) {
            const int x = min(x0 + u, len - 1);
            fi[u] = (float)s[(size_t)rtf_in(x, len, R) * ss];
            fo[u] = (float)s[(size_t)rtf_out(x, R) * ss];
        }
    };
    fetch(0, qi, qo);
    for (int x0 = 0; x0 < len; x0 += kPf) {
        fetch(x0 + kPf, ni, no);
#pragma unroll
        for (int u = 0; u < kPf; ++u) {
            if (x0 + u < len) {
                sum += (qi[u] - qo[u]) * div;
                d[(size_t)(x0 + u) * ds] = (T)sum;
            }
        }
#pragma unroll
        for (int u = 0; u < kPf; ++u) {
            qi[u] = ni[u];
            qo[u] = no[u];
        }
    }
}

// Horizontal float pass: the running sum of a row is sequential, so lanes are rows — but a lane walking
// its own row touches 64 cache lines per step. Instead the entering and the leaving samples of 64
// output columns x 64 rows go through two LDS tiles (loaded with lane = column: coalesced), the chain
// runs with lane = row on LDS, the results go back through the first tile; the next chunk's tiles are
// fetched into registers while the current chunk is summed.
template <typename T>
__global__ __launch_bounds__(64) void boxblur_rt_float_h_kernel(const RParams prm) {
    // RB rows per wave: rows are the only parallelism of this pass, and a wave's time is the latency of its chain
    // (LDS round trips, not lane work), so fewer rows per wave = more waves in flight = proportionally faster until
    // HBM saturates (64 rows: 544 waves on 8 4K frames, half the SIMDs idle)
    constexpr int RB = VSZIP_RTF_RB;
    __shared__ float tin[RB][65], tout[RB][65];
    const int b = blockIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int lane = threadIdx.x;
    const int y0 = (b - pl.block0) * RB;
    const int rows = min(RB, pl.h - y0), len = pl.w, R = prm.radius;
    const T *s = static_cast<const T *>(pl.src) + (size_t)y0 * pl.sstride;
    T *d = static_cast<T *>(pl.dst) + (size_t)y0 * pl.dstride;
    const size_t ss = pl.sstride, ds = pl.dstride;
    const float div = 1.0f / (float)(R * 2 + 1);
    // the initial sum (:47-50), lane = row, sequential over the first R + 1 samples of the row
    float sum = 0.0f;
    if (lane < rows) {
        const T *row = s + (size_t)lane * ss;
        sum = (float)row[R];
        for (int x = 0; x < R; ++x) sum += (float)row[x] * 2;
        sum = sum * div;
    }
    const int nchunk = (len + 63) / 64;
    float ni[RB], no[RB];
    auto fetch = [&](int c) __attribute__((always_inline)) {
        const int x = min(c * 64 + lane, len - 1);
        const int ci = rtf_in(x, len, R), co = rtf_out(x, R);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const T *row = s + (size_t)min(r, rows - 1) * ss;
            ni[r] = (float)row[ci];
            no[r] = (float)row[co];
        }
    };
    fetch(0);
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            tin[r][lane] = ni[r];
            tout[r][lane] = no[r];
        }
        fc_wave_sync();
        if (c + 1 < nchunk) fetch(c + 1);
        const int x0 = c * 64, cw = min(64, len - x0);
        if (lane < rows) {
            int k = 0;
            for (; k + 8 <= cw; k += 8) {
                float a[8], o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a[u] = tin[lane][k + u];
                    o[u] = tout[lane][k + u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    sum += (a[u] - o[u]) * div;
                    tin[lane][k + u] = sum;
                }
            }
            for (; k < cw; ++k) {
                sum += (tin[lane][k] - tout[lane][k]) * div;
                tin[lane][k] = sum;
            }
        }
        fc_wave_sync();
        if (lane < cw) {
#pragma unroll 8
            for (int r = 0; r < rows; ++r) d[(size_t)r * ds + x0 + lane] = (T)tin[r][lane];
        }
        fc_wave_sync();
    }
}

// ---- RT float, several passes along one axis in ONE kernel ------------------------------------------------------------------------------
// blurFloat's running sum is sequential along its line, but the PASSES of blur_passes (boxblur_runtime.zig:81-119) pipeline: stage k's output x
// needs stage k-1's outputs x - R - 1 ... x + R only, so P stages run in lock step, stage k trailing stage k-1 by R samples, each with its own
// running sum and the last 2 R + 2 outputs of its producer in a lane-private LDS ring (slot = sample index % D). A line is then read once and
// written once for all P passes. At tick t the source sample t enters ring 0 and stage k = 1 ... P produces its output x = t - k R:
//   x == 0          the initial sum of :47-50 from its producer's samples 0 ... R (all there: the producer is at R)
//   every x         sum += (in - out) * div with the three index ranges of :55-78 (rtf_in / rtf_out above)
// Away from the line ends the entering sample is what the previous stage produced in this very tick (a register) and the leaving one sits in the
// slot after the one just written (`fast`); the first (P + 1) R and the last P R ticks take `tick`, which indexes the rings by sample number.
// Every stage performs the per-pass kernel's operations on the per-pass kernel's values (an f16 plane's stage output is rounded to f16 before the
// next stage sees it), so the bits are those of P launches. Lines of at least 2 R + 2 samples.
// What a stage computes, float (blurFloat :43-79) or integer (blurInt :10-41 in the closed form of the per-pass kernels above: dst = (inv2 * E + 32768 +
// ((E_0 * invlo) >> 16)) >> 16 with E the window sum; E_0, the stage's first window sum, is what its initial sum is). E: a sample in LDS, V: in registers, S: the sum.
template <typename T>
struct FcArithF {
    using E = float;
    using V = float;
    using S = float;
    float div;
    __device__ __forceinline__ void setup(int R) { div = 1.0f / (float)(R * 2 + 1); }
    __device__ __forceinline__ void start(S &sum, uint32_t &, V s) const { sum = s * div; }
    __device__ __forceinline__ V step(S &sum, uint32_t, V a, V o) const {
        sum += (a - o) * div;
        return (float)(T)sum;
    }
};
template <typename T>
struct FcArithI {
    using E = uint16_t;
    using V = uint32_t;
    using S = uint32_t;
    uint32_t inv2, invlo;
    __device__ __forceinline__ void setup(int R) {
        const uint32_t ksize = 2u * (uint32_t)R + 1u;
        const uint64_t inv = ((1ull << 32) + (uint64_t)R) / ksize;
        inv2 = (uint32_t)(inv >> 16);
        invlo = (uint32_t)(inv & 0xffffu);
    }
    __device__ __forceinline__ void start(S &sum, uint32_t &kk, V s) const {
        sum = s;
        kk = 32768u + (uint32_t)(((uint64_t)s * invlo) >> 16);
    }
    __device__ __forceinline__ V step(S &sum, uint32_t kk, V a, V o) const {
        sum += a - o;
        // the window sum of a chain radius is below 2^24 (R <= 127) and inv2 below 2^16: one full-rate 24-bit multiply-add (left to itself the compiler forms
        // v_mad_u64_u32 out of the multiply and the add: a quarter-rate 64-bit instruction per stage and tick)
        uint32_t m;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(m) : "v"(sum), "v"(inv2), "v"(kk));
        return m >> 16;
    }
};
template <typename T>
using FcArith = typename std::conditional<std::is_integral<T>::value, FcArithI<T>, FcArithF<T>>::type;

template <typename T, int P, int LS /* lanes per ring row */>
struct FChain {
    using A = FcArith<T>;
    using E = typename A::E;
    using V = typename A::V;
    A ar;
    typename A::S sum[P];
    uint32_t kk[P];  // (integer stages: 32768 + ((E_0 * invlo) >> 16))
    V carry[P];      // what the source (0) / stage k produced at the previous tick: the next stage's entering sample at this one
    E *ring;         // this lane's column of [D][P][LS]: sample y of ring j (0: the source, k: stage k's output) sits in slot (y + j L) % D, so that at
                     // tick t every ring's newest sample t - j L is in slot t % D and every stage's leaving sample in slot (t + 1) % D
    int D, R, L, len;
    __device__ __forceinline__ void init(E *lane_ring, int radius, int length) {
        ring = lane_ring;
        R = radius;
        L = radius + 1;
        D = 2 * radius + 3;
        len = length;
        ar.setup(radius);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            sum[k] = 0;
            carry[k] = 0;
            kk[k] = 0;
        }
    }
    __device__ __forceinline__ int ticks() const { return len + P * L; }
    __device__ __forceinline__ int lag() const { return P * L; }
    // sample y of ring j at tick t (c = t % D): the ring's newest sample t - j L is in slot c, y sits t - j L - y slots before it — 0 ... D - 1 for every sample a tick touches
    __device__ __forceinline__ E *at(int j, int y, int t, int c) const {
        int sl = c - (t - j * L - y);
        if (sl < 0) sl += D;
        return ring + (sl * P + j) * LS;
    }
    // One tick near a line end (any stage may be starting, mirroring or finished); c = t % D. All ring reads of the tick come before its writes, as in `fast`: a stage's
    // inputs were produced at earlier ticks.
    __device__ __forceinline__ bool tick(const int t, const int c, const V v, V &out) {
        V a[P], o[P];
        bool act[P];
        const E *pn = ring + (c + 1 == D ? 0 : c + 1) * (P * LS);  // the slot of every ring's leaving sample, for the stages that are away from their ends
#pragma unroll
        for (int k = 1; k <= P; ++k) {
            const int x = t - k * L;
            act[k - 1] = x >= 0 && x < len;
            a[k - 1] = o[k - 1] = 0;
            if (act[k - 1]) {
                if (x > R && x < len - R) {  // (the stages start L ticks apart and an end lasts R + 1 ticks: at most one stage of a tick is not here)
                    a[k - 1] = carry[k - 1];
                    o[k - 1] = (V)pn[(k - 1) * LS];
                } else {
                    if (x == 0) {
                        V s = (V)*at(k - 1, R, t, c);
                        for (int j = 0; j < R; ++j) s += (V)*at(k - 1, j, t, c) * 2;
                        ar.start(sum[k - 1], kk[k - 1], s);
                    }
                    a[k - 1] = (V)*at(k - 1, rtf_in(x, len, R), t, c);
                    o[k - 1] = (V)*at(k - 1, rtf_out(x, R), t, c);
                }
            }
        }
        E *pc = ring + c * (P * LS);  // every ring's newest sample: what this tick writes
        if (t < len) pc[0] = (E)v;
        carry[0] = v;
        bool has = false;
#pragma unroll
        for (int k = 1; k <= P; ++k) {
            if (act[k - 1]) {
                const V r = ar.step(sum[k - 1], kk[k - 1], a[k - 1], o[k - 1]);
                if (k < P) {
                    pc[k * LS] = (E)r;
                    carry[k] = r;
                } else {
                    out = r;
                    has = true;
                }
            }
        }
        return has;
    }
    // U ticks t ... t + U - 1 away from the line ends; c = t % D. Two things make a tick cheap. (1) The leaving samples of all U ticks are read before
    // the first tick writes: they were produced 2 R + 2 ticks before their tick, so none is written inside a group of U <= 4 (the compiler cannot
    // move a ring read above the ring write of the stage before on its own: it cannot know the slots differ), and one slot address per tick serves
    // all rings (the layout above; the ring index is an immediate offset). (2) A stage trails its producer by R + 1, not R: its entering sample is what
    // the producer made at the PREVIOUS tick (`carry`), so the P stages of a tick do not depend on one another — a wave is alone on its SIMD here
    // (columns / rows are the only parallelism) and P dependent sub-mul-add triples per tick were most of its time.
    template <int U>
    __device__ __forceinline__ void fast(const int c, const V *v, V *out) {
        if (c + U < D) {  // no slot of the group wraps: one address, the slots are immediate offsets as well
            E *pw[U + 1];
            E *base = ring + c * (P * LS);
#pragma unroll
            for (int u = 0; u <= U; ++u) pw[u] = base + u * (P * LS);
            fast_at<U>(pw, v, out);
        } else {
            E *pw[U + 1];
#pragma unroll
            for (int u = 0; u <= U; ++u) {
                int sl = c + u;
                if (sl >= D) sl -= D;
                pw[u] = ring + sl * (P * LS);
            }
            fast_at<U>(pw, v, out);
        }
    }
    template <int U>
    __device__ __forceinline__ void fast_at(E *const *pw, const V *v, V *out) {
        V o[U][P];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < P; ++k) o[u][k] = (V)pw[u + 1][k * LS];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            V na[P + 1];
            na[0] = v[u];
#pragma unroll
            for (int k = 1; k <= P; ++k) na[k] = ar.step(sum[k - 1], kk[k - 1], carry[k - 1], o[u][k - 1]);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                pw[u][k * LS] = (E)na[k];
                carry[k] = na[k];
            }
            out[u] = na[P];
        }
    }
    // every stage away from its line's ends at ticks t ... t + u - 1
    __device__ __forceinline__ bool interior(int t, int u) const { return t > P * L + R && t + u <= len; }
    __device__ __forceinline__ int first_interior() const { return P * L + R + 1; }
};

constexpr int kFcPf = 32;
// Vertical: lane = column, the plane's rows are the ticks. Columns are the only parallelism a running sum leaves (about one wave per SIMD on 8 4K frames),
// so what counts is a wave's own time per tick, and a wave hides the memory latency itself: three register sets of kFcPf rows rotate — one is summed
// while two are in flight. One load instruction fetches FOUR rows (16 lanes x 4 samples each) and one store writes four, through LDS both ways: with a
// load and a store per row the 64 memory operations s_waitcnt can count (vmcnt is 6 bits, stores included) were one group — no lookahead at all.
template <typename T, int P>
__global__ __launch_bounds__(64) void boxblur_rt_float_vchain_kernel(const RParams prm) {
    using CH = FChain<T, P, 64>;
    using E = typename CH::E;  // float planes: f32 in LDS; integer planes (the same chain with blurInt's closed form as the stage): u16
    using V = typename CH::V;
    extern __shared__ __attribute__((aligned(16))) unsigned char fc_lds_raw[];  // [D][P][64] rings, [kFcPf][64] parked rows (the chain takes its group's rows by a running index), [8][64] outputs
    E *fc_lds = reinterpret_cast<E *>(fc_lds_raw);
    typedef T Raw __attribute__((ext_vector_type(4)));
    typedef E f32x4 __attribute__((ext_vector_type(4)));  // (four LDS samples)
    const int b = blockIdx.x, lane = threadIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int i0 = (b - pl.block0) * 64, i = i0 + lane;
    const bool ok = i < pl.w;
    const int qrow = lane >> 4, qcol = 4 * (lane & 15);
    const int vcol = min(i0 + qcol, ((pl.w + 3) & ~3) - 4);  // (a lane group past the row's end re-reads the last one: its columns are never stored)
    const T *sv = static_cast<const T *>(pl.src) + vcol;
    T *d = static_cast<T *>(pl.dst) + min(i, pl.w - 1);
    T *dv = static_cast<T *>(pl.dst) + i0 + qcol;
    const size_t ss = pl.sstride, ds = pl.dstride;
    const int len = pl.h;
    CH ch;
    ch.init(fc_lds + lane, prm.radius, len);
    const int D = ch.D;
    E *park = fc_lds + P * D * 64, *otile = park + kFcPf * 64;
    const int total = ch.ticks(), lag = ch.lag();
    auto fetch = [&](int t0, Raw *f) __attribute__((always_inline)) {
        if (t0 < len) {
#pragma unroll
            for (int j = 0; j < kFcPf / 4; ++j) f[j] = *reinterpret_cast<const Raw *>(sv + (size_t)min(t0 + 4 * j + qrow, len - 1) * ss);
        }
    };
    const int tlo = ch.first_interior();
    const bool deep = 2 * prm.radius + 2 >= 8;  // groups of 8 ticks: their leaving samples are all older than the group
    auto run = [&](int t0, const Raw *q) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < kFcPf / 4; ++j) *reinterpret_cast<f32x4 *>(park + (4 * j + qrow) * 64 + qcol) = __builtin_convertvector(q[j], f32x4);
        fc_wave_sync();
        const int ne = min(kFcPf, total - t0);
        const int e1 = min(ne, max(0, tlo - t0)), e2 = min(ne, len - t0);  // [e1, e2): every stage away from the line's ends
        int u0 = 0;
        int c = t0 % D;
        for (; u0 < e1; ++u0) {
            V r;
            if (ch.tick(t0 + u0, c, (V)park[u0 * 64 + lane], r) && ok) d[(size_t)(t0 + u0 - lag) * ds] = (T)r;
            c = c + 1 == D ? 0 : c + 1;
        }
        if (deep) {
            for (; u0 + 8 <= e2; u0 += 8) {
                V v[8], r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (V)park[(u0 + u) * 64 + lane];
                ch.template fast<8>(c, v, r);
#pragma unroll
                for (int u = 0; u < 8; ++u) otile[u * 64 + lane] = (E)r[u];
                fc_wave_sync();
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 o4 = *reinterpret_cast<const f32x4 *>(otile + (4 * h + qrow) * 64 + qcol);
                    const Raw w4 = __builtin_convertvector(o4, Raw);
                    T *dp = dv + (size_t)(t0 + u0 - lag + 4 * h + qrow) * ds;
                    if (i0 + qcol + 3 < pl.w) {
                        *reinterpret_cast<Raw *>(dp) = w4;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (i0 + qcol + e < pl.w) dp[e] = w4[e];
                    }
                }
                fc_wave_sync();
                c += 8;
                if (c >= D) c -= D;
            }
        }
        for (; u0 + 4 <= e2; u0 += 4) {
            V v[4], r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (V)park[(u0 + u) * 64 + lane];
            ch.template fast<4>(c, v, r);
#pragma unroll
            for (int u = 0; u < 4; ++u) otile[u * 64 + lane] = (E)r[u];
            fc_wave_sync();
            const f32x4 o4 = *reinterpret_cast<const f32x4 *>(otile + qrow * 64 + qcol);
            const Raw w4 = __builtin_convertvector(o4, Raw);
            T *dp = dv + (size_t)(t0 + u0 - lag + qrow) * ds;
            if (i0 + qcol + 3 < pl.w) {
                *reinterpret_cast<Raw *>(dp) = w4;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i0 + qcol + e < pl.w) dp[e] = w4[e];
            }
            fc_wave_sync();
            c += 4;
            if (c >= D) c -= D;
        }
        for (; u0 < e2; ++u0) {
            V v = (V)park[u0 * 64 + lane], r;
            ch.template fast<1>(c, &v, &r);
            if (ok) d[(size_t)(t0 + u0 - lag) * ds] = (T)r;
            c = c + 1 == D ? 0 : c + 1;
        }
        for (; u0 < ne; ++u0) {
            V r;
            if (ch.tick(t0 + u0, c, (V)park[u0 * 64 + lane], r) && ok) d[(size_t)(t0 + u0 - lag) * ds] = (T)r;
            c = c + 1 == D ? 0 : c + 1;
        }
        fc_wave_sync();
    };
    Raw b0[kFcPf / 4], b1[kFcPf / 4], b2[kFcPf / 4];
    fetch(0, b0);
    fetch(kFcPf, b1);
    for (int t0 = 0; t0 < total; t0 += 3 * kFcPf) {
        fetch(t0 + 2 * kFcPf, b2);
        run(t0, b0);
        if (t0 + kFcPf >= total) break;
        fetch(t0 + 3 * kFcPf, b0);
        run(t0 + kFcPf, b1);
        if (t0 + 2 * kFcPf >= total) break;
        fetch(t0 + 4 * kFcPf, b1);
        run(t0 + 2 * kFcPf, b2);
    }
}

// ---- the integer vertical chain in BANDS (round 4) -----------------------------------------------------------------------------------------
// The chain above walks a whole column per lane: a 16-frame 1080p call is 960 waves on 1 024 SIMDs, each alone with its latencies (~30 ns per stage
// and tick), and the P (2R + 1) ticks at a plane's top and bottom run the generic `tick` (0.44 us each). blurInt's closed form (:10-41) makes the
// passes EXACT under any segmentation of the rows:
//   * a stage's output is a function of its input's window SUM and of one constant per line, K = 32768 + ((E_0 * invlo) >> 16), E_0 = the stage's
//     first window sum. With K known a stage needs no start: its running sum is a sliding sum of whatever sits in its ring, so a band can begin with
//     zeroed rings and sums P R rows above its first output row — what entered before the true samples (zeros, and the outputs computed from them)
//     leaves every window again, bit for bit (integer adds): stage k is exact k (2R + 1) ticks after the band's first source row, the last stage at
//     the band's first output row.
//   * the plane's ends need no special ticks either: blurInt's padding is the edge-duplicating mirror (-k -> k - 1, len - 1 + k -> len - k) of EVERY
//     stage's input, and a symmetric window over a mirrored input gives a mirrored output (same sum, same K), so feeding the chain the mirror-extended
//     SOURCE rows (P R virtual rows on either side) makes every stage's virtual outputs the mirror of its real ones — exactly the padding the next
//     stage wants.
// So: boxblur_rt_ichain_kernel<T, P, 1> runs the plane's first P (R + 1) + 1 ticks through the generic chain once per column and leaves the P
// constants in a table (no stores); <T, P, 2> runs one band of rows per wave — every tick a fast one — with the constants from the table. Waves
// per call: column groups x bands (4 096 aimed at), against column groups alone.
template <typename T, int P, int MODE /* 1: E_0 constants of the P stages, 2: one band of output rows */>
__global__ __launch_bounds__(64) void boxblur_rt_ichain_kernel(const RParams prm, uint32_t *__restrict__ kk_tab, const int band_rows) {
    using CH = FChain<T, P, 64>;
    using E = typename CH::E;
    using V = typename CH::V;
    extern __shared__ __attribute__((aligned(16))) unsigned char fc_lds_raw[];  // [D][P][64] rings, [kFcPf][64] parked rows, [8][64] outputs
    E *fc_lds = reinterpret_cast<E *>(fc_lds_raw);
    typedef T Raw __attribute__((ext_vector_type(4)));
    typedef E e4 __attribute__((ext_vector_type(4)));
    const int b = blockIdx.x, lane = threadIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int ncg = (pl.w + 63) / 64;
    const int lb = b - pl.block0;
    const int cg = MODE == 2 ? lb % ncg : lb, band = MODE == 2 ? lb / ncg : 0;
    const int i0 = cg * 64, i = i0 + lane;
    const int qrow = lane >> 4, qcol = 4 * (lane & 15);
    const int vcol = min(i0 + qcol, ((pl.w + 3) & ~3) - 4);
    const T *sv = static_cast<const T *>(pl.src) + vcol;
    T *dv = static_cast<T *>(pl.dst) + i0 + qcol;
    const size_t ss = pl.sstride, ds = pl.dstride;
    const int len = pl.h, R = prm.radius;
    const int wpad = ncg * 64;
    uint32_t *kt = kk_tab + pl.aux + i;  // stage k's constant of column i: kt[k * wpad]
    CH ch;
    ch.init(fc_lds + lane, R, len);
    const int D = ch.D, L = ch.L;
    E *park = fc_lds + P * D * 64, *otile = park + kFcPf * 64;
    // source row of (virtual) tick t: blurInt's mirror on either side
    auto src_row = [&](int t) __attribute__((always_inline)) {
        int r = t < 0 ? -t - 1 : t;
        r = r >= len ? 2 * len - 1 - r : r;
        return min(max(r, 0), len - 1);
    };
    if constexpr (MODE == 1) {
        const int total = P * L + 1;  // stage P starts (and its constant is known) at tick P L
        auto fetch = [&](int t0, Raw *f) __attribute__((always_inline)) {
            if (t0 < min(len, total)) {
#pragma unroll
                for (int j = 0; j < kFcPf / 4; ++j) f[j] = *reinterpret_cast<const Raw *>(sv + (size_t)min(t0 + 4 * j + qrow, len - 1) * ss);
            }
        };
        auto run1 = [&](int tb, const Raw *q) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < kFcPf / 4; ++j) *reinterpret_cast<e4 *>(park + (4 * j + qrow) * 64 + qcol) = __builtin_convertvector(q[j], e4);
            fc_wave_sync();
            const int ne = min(kFcPf, total - tb);
            int c = tb % D;
            for (int u = 0; u < ne; ++u) {
                V r;
                (void)ch.tick(tb + u, c, (V)park[u * 64 + lane], r);
                c = c + 1 == D ? 0 : c + 1;
            }
            fc_wave_sync();
        };
        Raw b0[kFcPf / 4], b1[kFcPf / 4];
        fetch(0, b0);
        for (int t0 = 0; t0 < total; t0 += 2 * kFcPf) {
            fetch(t0 + kFcPf, b1);
            run1(t0, b0);
            if (t0 + kFcPf >= total) break;
            fetch(t0 + 2 * kFcPf, b0);
            run1(t0 + kFcPf, b1);
        }
#pragma unroll
        for (int k = 0; k < P; ++k) kt[(size_t)k * wpad] = ch.kk[k];
    } else {
        const int nb = (len + band_rows - 1) / band_rows;
        const int y0 = band * band_rows, y1 = band == nb - 1 ? len : y0 + band_rows;
        const int lag = P * L;
        const int ts = y0 - P * R, te = y1 + lag;  // ticks [ts, te): source rows from P R above the band; the last stage's row y1 - 1 leaves at tick y1 - 1 + lag
        // zeroed rings, sums and carries + the constants: every tick is a fast one
        for (int k = 0; k < P * D; ++k) fc_lds[k * 64 + lane] = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) ch.kk[k] = kt[(size_t)k * wpad];
        fc_wave_sync();
        auto fetch = [&](int t0, Raw *f) __attribute__((always_inline)) {
            if (t0 < te) {
#pragma unroll
                for (int j = 0; j < kFcPf / 4; ++j) f[j] = *reinterpret_cast<const Raw *>(sv + (size_t)src_row(t0 + 4 * j + qrow) * ss);
            }
        };
        const bool deep = 2 * R + 2 >= 8;
        const bool full4 = i0 + qcol + 3 < pl.w;
        auto put4 = [&](int row, const e4 o4) __attribute__((always_inline)) {
            if (row < y0 || row >= y1) return;
            const Raw w4 = __builtin_convertvector(o4, Raw);
            T *dp = dv + (size_t)row * ds;
            if (full4) {
                if (prm.keep)
                    *reinterpret_cast<Raw *>(dp) = w4;
                else
                    __builtin_nontemporal_store(w4, reinterpret_cast<Raw *>(dp));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i0 + qcol + e < pl.w) dp[e] = w4[e];
            }
        };
        auto run = [&](int t0, const Raw *q) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < kFcPf / 4; ++j) *reinterpret_cast<e4 *>(park + (4 * j + qrow) * 64 + qcol) = __builtin_convertvector(q[j], e4);
            fc_wave_sync();
            const int ne = min(kFcPf, te - t0);
            int c = (t0 - ts) % D;  // (any phase: the rings only need consistent slots)
            int u0 = 0;
            if (deep) {
                for (; u0 + 8 <= ne; u0 += 8) {
                    V v[8], r[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = (V)park[(u0 + u) * 64 + lane];
                    ch.template fast<8>(c, v, r);
#pragma unroll
                    for (int u = 0; u < 8; ++u) otile[u * 64 + lane] = (E)r[u];
                    fc_wave_sync();
#pragma unroll
                    for (int h = 0; h < 2; ++h) put4(t0 + u0 - lag + 4 * h + qrow, *reinterpret_cast<const e4 *>(otile + (4 * h + qrow) * 64 + qcol));
                    fc_wave_sync();
                    c += 8;
                    if (c >= D) c -= D;
                }
            }
            for (; u0 + 4 <= ne; u0 += 4) {
                V v[4], r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (V)park[(u0 + u) * 64 + lane];
                ch.template fast<4>(c, v, r);
#pragma unroll
                for (int u = 0; u < 4; ++u) otile[u * 64 + lane] = (E)r[u];
                fc_wave_sync();
                put4(t0 + u0 - lag + qrow, *reinterpret_cast<const e4 *>(otile + qrow * 64 + qcol));
                fc_wave_sync();
                c += 4;
                if (c >= D) c -= D;
            }
            for (; u0 < ne; ++u0) {
                V v = (V)park[u0 * 64 + lane], r;
                ch.template fast<1>(c, &v, &r);
                const int row = t0 + u0 - lag;
                if (row >= y0 && row < y1 && i < pl.w) static_cast<T *>(pl.dst)[(size_t)row * ds + i] = (T)r;
                c = c + 1 == D ? 0 : c + 1;
            }
            fc_wave_sync();
        };
        Raw b0[kFcPf / 4], b1[kFcPf / 4], b2[kFcPf / 4];
        fetch(ts, b0);
        fetch(ts + kFcPf, b1);
        for (int t0 = ts; t0 < te; t0 += 3 * kFcPf) {
            fetch(t0 + 2 * kFcPf, b2);
            run(t0, b0);
            if (t0 + kFcPf >= te) break;
            fetch(t0 + 3 * kFcPf, b0);
            run(t0 + kFcPf, b1);
            if (t0 + 2 * kFcPf >= te) break;
            fetch(t0 + 4 * kFcPf, b1);
            run(t0 + 2 * kFcPf, b2);
        }
    }
}

// Horizontal: lane = row for the chain, lane = column for memory: 64 source columns of the wave's 64 rows enter through one LDS tile, the last stage's
// outputs leave through another, flushed whenever its 64 columns are complete. All 64 lanes carry a row (the per-pass kernel takes 16 rows a wave for
// more waves; here a wave's time is the line's ticks times the cost of a tick whatever its row count, and a SIMD runs two waves no faster than one).
constexpr int kFcRB = 64;
template <typename T, int P>
__global__ __launch_bounds__(64) void boxblur_rt_float_hchain_kernel(const RParams prm) {
    constexpr int RB = kFcRB;
    extern __shared__ float fc_lds[];  // [D][P][RB] rings
    __shared__ float tin[RB][65], tout[RB][65];
    const int b = blockIdx.x, lane = threadIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int y0 = (b - pl.block0) * RB;
    const int rows = min(RB, pl.h - y0), len = pl.w;
    const T *s = static_cast<const T *>(pl.src) + (size_t)y0 * pl.sstride;
    T *d = static_cast<T *>(pl.dst) + (size_t)y0 * pl.dstride;
    const size_t ss = pl.sstride, ds = pl.dstride;
    FChain<T, P, RB> ch;
    ch.init(fc_lds + lane, prm.radius, len);  // (float planes only: E = V = float)
    const int D = ch.D;
    const int total = ch.ticks(), lag = ch.lag(), tlo = ch.first_interior();
    const bool deep = 2 * prm.radius + 2 >= 8;  // groups of 8 ticks: their leaving samples are all older than the group
    const int nchunk = (total + 63) / 64;
    float ni[RB];
    // Memory side of a 64 x 64 tile. WIDE (round 5: whole blocks of f32 planes with 16-byte aligned rows, whole chunks): a lane moves 4 columns
    // of rows rg, rg + 4, ... with 16-byte loads and stores - 16 instructions a tile each way instead of 64 (timing only, no loads and stores
    // at all: 427 -> 307 us a launch). Both LDS sides stay conflict free at the pitch of 65 floats (bank = row + column).
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int rg = lane >> 4, c4 = (lane & 15) * 4;
    const bool wide_in = sizeof(T) == 4 && rows == RB && ((reinterpret_cast<uintptr_t>(s) | (ss * sizeof(T))) & 15) == 0;
    const bool wide_out = sizeof(T) == 4 && rows == RB && ((reinterpret_cast<uintptr_t>(d) | (ds * sizeof(T))) & 15) == 0;
    bool parked_wide = false;  // how the chunk in ni[] was fetched
    auto fetch = [&](int c) __attribute__((always_inline)) {
        parked_wide = wide_in && (c + 1) * 64 <= len;
        if (parked_wide) {
            const T *p = s + (size_t)rg * ss + c * 64 + c4;
#pragma unroll
            for (int k = 0; k < RB / 4; ++k) {
                const v4 q = *reinterpret_cast<const v4 *>(p + (size_t)(4 * k) * ss);
                ni[4 * k] = q.x, ni[4 * k + 1] = q.y, ni[4 * k + 2] = q.z, ni[4 * k + 3] = q.w;
            }
            return;
        }
        const int x = min(c * 64 + lane, len - 1);
#pragma unroll
        for (int r = 0; r < RB; ++r) ni[r] = (float)s[(size_t)min(r, rows - 1) * ss + x];
    };
    fetch(0);
    for (int c = 0; c < nchunk; ++c) {
        if (parked_wide) {
#pragma unroll
            for (int k = 0; k < RB / 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) tin[4 * k + rg][c4 + j] = ni[4 * k + j];
        } else {
#pragma unroll
            for (int r = 0; r < RB; ++r) tin[r][lane] = ni[r];
        }
        fc_wave_sync();
        if ((c + 1) * 64 < len) fetch(c + 1);
        const int t0 = c * 64, kmax = min(64, total - t0);
        int k = 0;
        while (k < kmax) {
            const int xp = t0 + k - lag;                      // the column the last stage produces at tick t0 + k
            const int kend = min(kmax, k + 64 - (xp & 63));  // up to the end of its 64-column tile
            if (lane < rows) {
                int kk = k;
                const int e1 = min(kend, max(k, tlo - t0)), e2 = min(kend, len - t0);  // [e1, e2): every stage away from the line's ends
                int cc = (t0 + kk) % D;
                for (; kk < e1; ++kk) {
                    float r;
                    if (ch.tick(t0 + kk, cc, tin[lane][kk], r)) tout[lane][(t0 + kk - lag) & 63] = r;
                    cc = cc + 1 == D ? 0 : cc + 1;
                }
                if (deep) {
                    for (; kk + 8 <= e2; kk += 8) {
                        float v[8], r[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = tin[lane][kk + u];
                        ch.template fast<8>(cc, v, r);
                        float *to = &tout[lane][(t0 + kk - lag) & 63];
#pragma unroll
                        for (int u = 0; u < 8; ++u) to[u] = r[u];
                        cc += 8;
                        if (cc >= D) cc -= D;
                    }
                }
                for (; kk + 4 <= e2; kk += 4) {
                    float v[4], r[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = tin[lane][kk + u];
                    ch.template fast<4>(cc, v, r);
                    float *to = &tout[lane][(t0 + kk - lag) & 63];  // (a segment ends with its output tile: the four do not wrap)
#pragma unroll
                    for (int u = 0; u < 4; ++u) to[u] = r[u];
                    cc += 4;
                    if (cc >= D) cc -= D;
                }
                for (; kk < e2; ++kk) {
                    float v = tin[lane][kk], r;
                    ch.template fast<1>(cc, &v, &r);
                    tout[lane][(t0 + kk - lag) & 63] = r;
                    cc = cc + 1 == D ? 0 : cc + 1;
                }
                for (; kk < kend; ++kk) {
                    float r;
                    if (ch.tick(t0 + kk, cc, tin[lane][kk], r)) tout[lane][(t0 + kk - lag) & 63] = r;
                    cc = cc + 1 == D ? 0 : cc + 1;
                }
            }
            fc_wave_sync();
            const int xl = t0 + kend - 1 - lag;
            if (xl >= 0 && ((xl & 63) == 63 || xl == len - 1)) {
                const int x = (xl & ~63) + lane;
                if (wide_out && (xl & 63) == 63) {
                    T *q = d + (size_t)rg * ds + (xl & ~63) + c4;
#pragma unroll
                    for (int k = 0; k < RB / 4; ++k) {
                        const float *t = &tout[4 * k + rg][c4];
                        *reinterpret_cast<v4 *>(q + (size_t)(4 * k) * ds) = v4{t[0], t[1], t[2], t[3]};
                    }
                } else if (x < len) {
#pragma unroll 8
                    for (int r = 0; r < rows; ++r) d[(size_t)r * ds + x] = (T)tout[r][lane];
                }
                fc_wave_sync();
            }
            k = kend;
        }
    }
}

#ifdef VSZIP_DEV_VARIANTS  // measured slower than what the default build runs (options.inc)
// ---------------------------------------------------------------------------------------------
// Round 3 — several integer passes along one axis in ONE kernel (blur_passes, boxblur_runtime.zig:81-119): a 256-thread
// workgroup owns a row; a thread keeps its 16 contiguous samples in REGISTERS across the passes. Per pass: in-thread
// prefix, DPP wave scan, the four wave totals through LDS, the row's prefix parked in LDS laid out [sample-in-thread][thread]
// (conflict free both ways), then every output is the difference of two prefix entries plus the two mirror terms of blurInt
// (:24-40) at the row ends — the closed form of the single-pass kernels, so the bits are theirs. HBM sees the row once
// in and once out whatever the number of passes (the per-pass kernels: one round trip per pass). Two barriers per pass; the
// prefix is double buffered. Rows up to 4096 samples; the vertical passes run through it on transposed planes.
// ---------------------------------------------------------------------------------------------
constexpr int kHmNT = 256, kHmEPT = 16, kHmMaxW = kHmNT * kHmEPT;
constexpr int kHmMaxR = 119;                                   // halo entries fit the padding of the prefix array
constexpr int kHmCols = kHmNT + (2 * kHmMaxR + 2 + 15) / 16;   // columns of the [16][cols] prefix array: (w + 2 R + 2) / 16 entries per row

template <typename T>
__global__ __launch_bounds__(kHmNT) void boxblur_rt_hmulti_kernel(const RParams prm, const int passes) {
    // The prefix of the MIRROR-EXTENDED row (blurInt's taps left of 0 mirror as -k -> k-1, right of w-1 as w-1+k -> w-k, :24-40), minus
    // the left halo's sum (which cancels in every difference): PV(j) = Preal(j) inside the row, -Preal(m - 2) at j = -m, and
    // 2 Preal(w-1) - Preal(w-1-m) at j = w-1+m. With it every output is ONE difference PV(x + R) - PV(x - R - 1): no thread
    // takes a border path. Entry j lives at [(j + R + 1) & 15][(j + R + 1) >> 4]: a thread's 16 entries and a wave's reads of
    // "x + R" / "x - R - 1" are both conflict free.
    __shared__ uint32_t P[2][16 * kHmCols];
    __shared__ uint32_t wtot[2][4];
    const int b = blockIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int y = b - pl.block0;
    const int w = pl.w, R = prm.radius;
    const T *s = static_cast<const T *>(pl.src) + (size_t)y * pl.sstride;
    T *d = static_cast<T *>(pl.dst) + (size_t)y * pl.dstride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = tid * kHmEPT;
    uint32_t v[kHmEPT];
    {
        // 16 samples = one (u8) or two (u16) 16-byte loads; [w, stride) is readable padding, masked here
        constexpr int V = RtVec<T>::V;
#pragma unroll
        for (int g = 0; g < kHmEPT / V; ++g) {
            uint32_t t[V];
            if (x0 + g * V < w) {
                RtVec<T>::load(s + x0 + g * V, t);
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) t[k] = 0;
            }
#pragma unroll
            for (int k = 0; k < V; ++k) v[g * V + k] = (x0 + g * V + k < w) ? t[k] : 0u;
        }
    }
    const uint32_t ksize = 2u * (uint32_t)R + 1u;
    const uint64_t inv = ((1ull << 32) + (uint64_t)R) / ksize;
    const uint32_t inv2 = (uint32_t)(inv >> 16), invlo = (uint32_t)(inv & 0xffffu);
    const uint32_t off = (uint32_t)R + 1u;  // virtual index = real index + off
    auto slot = [&](uint32_t jv) { return (jv & 15u) * (uint32_t)kHmCols + (jv >> 4); };
    // per-thread constants of the two reads of output k: virtual indices x0 + k + 2 R + 1 and x0 + k
    for (int pass = 0; pass < passes; ++pass) {
        uint32_t *Pb = P[pass & 1];
#pragma unroll
        for (int k = 1; k < kHmEPT; ++k) v[k] += v[k - 1];  // in place: the thread's inclusive prefix
        const uint32_t incl = wave_incl_scan_dpp(v[kHmEPT - 1]);
        if (lane == 63) wtot[pass & 1][wave] = incl;
        __syncthreads();
        uint32_t base = incl - v[kHmEPT - 1];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < wave) base += wtot[pass & 1][q];
        if (x0 < w) {
#pragma unroll
            for (int k = 0; k < kHmEPT; ++k) Pb[slot((uint32_t)(x0 + k) + off)] = v[k] + base;  // (entries past w - 1 repeat Preal(w-1): harmless, rewritten below)
        }
        __syncthreads();
        // the 2 R + 1 halo entries, one per thread: left j = -m (m = 1 .. R + 1), right j = w - 1 + m (m = 1 .. R)
        {
            const uint32_t last = Pb[slot((uint32_t)(w - 1) + off)];
            if (tid <= R) {
                const int m = tid + 1;
                Pb[slot(off - (uint32_t)m)] = m >= 2 ? 0u - Pb[slot((uint32_t)(m - 2) + off)] : 0u;
            } else if (tid <= 2 * R) {
                const int m = tid - R;
                Pb[slot((uint32_t)(w - 1 + m) + off)] = 2u * last - Pb[slot((uint32_t)(w - 1 - m) + off)];
            }
        }
        __syncthreads();
        const uint32_t e0 = Pb[slot((uint32_t)R + off)] + Pb[slot((uint32_t)(R - 1) + off)];  // srcp[r] + 2 * sum_{x<r} srcp[x]
        const uint32_t kr = 32768u + (uint32_t)(((uint64_t)e0 * invlo) >> 16);
#pragma unroll
        for (int k = 0; k < kHmEPT; ++k) {
            const uint32_t x = (uint32_t)(x0 + k);
            const uint32_t e = Pb[slot(x + (uint32_t)R + off)] - Pb[slot(x + off - (uint32_t)R - 1u)];
            // the 16.16 value is the window mean + 0.5: 32 bits hold it (proved exhaustively in tests/test_oracle_boxblur.py)
            v[k] = (x0 + k < w) ? (e * inv2 + kr) >> 16 : 0u;
        }
    }
    {
        constexpr int V = RtVec<T>::V;
#pragma unroll
        for (int g = 0; g < kHmEPT / V; ++g)
            if (x0 + g * V < w) RtVec<T>::store(d + x0 + g * V, v + g * V, min(V, w - (x0 + g * V)), prm.keep);
    }
}

// 64 x 64 tiles through LDS: dst[x][y] = src[y][x] (integer planes; the vertical passes of the fused path)
#endif  // VSZIP_DEV_VARIANTS
// ---------------------------------------------------------------------------------------------
// Several HORIZONTAL passes of a SMALL radius in one launch (round 3): BoxBlur(hradius = 1..8, hpasses >= 2) is how scripts approximate a
// Gaussian, and each pass used to be a kernel of its own — the same ~100 us per 8 4K frames whatever the radius. One wave owns a row, keeps it
// in LDS as u16 (two buffers, ping-pong, R mirrored samples on either side so that every window is the same sum) and runs the passes on it:
// a lane takes 8 adjacent samples at a time (three 16-byte LDS reads: the vector and its neighbours), sums the first window directly and
// slides it — blurInt's closed form (:10-41, see RtVec above): dst[x] = (inv2 * E_x + 32768 + ((E_0 * invlo) >> 16)) >> 16 with E_0 read from
// the row's first R + 1 samples of the pass's own input, which are right there. One read and one write of the plane for all passes.
// ---------------------------------------------------------------------------------------------
constexpr int kHsMaxR = 16, kHsMaxW = 8192, kHsHalo = 16;  // (radius 13 x 5 passes is the reference README's third benchmark)
[[maybe_unused]] constexpr int kVsMaxR = 8;
template <typename T, int R>
__global__ __launch_bounds__(64) void boxblur_rt_hsmall_kernel(const RParams prm, const int npass, const int pitch /* u16 elements per LDS buffer */) {
    extern __shared__ __attribute__((aligned(16))) uint16_t hs[];
    const int b = blockIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int y = b - pl.block0, w = pl.w, lane = (int)threadIdx.x;
    const T *s = static_cast<const T *>(pl.src) + (size_t)y * pl.sstride;
    T *d = static_cast<T *>(pl.dst) + (size_t)y * pl.dstride;
    constexpr uint32_t ksize = 2u * R + 1u;
    constexpr uint64_t inv = ((1ull << 32) + (uint64_t)R) / ksize;
    constexpr uint32_t inv2 = (uint32_t)(inv >> 16), invlo = (uint32_t)(inv & 0xffffu);
    constexpr int H = kHsHalo, G = (R + 7) / 8, NT = 8 * (2 * G + 1);  // groups of halo reach; samples a lane unpacks per group of outputs
    const int nv = (w + 7) / 8;  // 8-sample groups of the row; sample x lives at element H + x of a buffer
    // stage the row (u16 in LDS whatever the clip's sample size)
    if constexpr (sizeof(T) == 1) {
        // 8-bit rows: the whole groups four lane-steps at a time with all four loads in flight, unpacked afterwards (64 x 1080p, 2 passes of
        // r = 13: 190 -> 166 us; the same batching made 16-bit rows SLOWER, 103 -> 144 us, so they keep the plain loop below)
        const int nfull = w / 8;
        for (int v0 = 0; v0 < nfull; v0 += 256) {
            uint2 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + 64 * u + lane;
                if (v < nfull) q[u] = *reinterpret_cast<const uint2 *>(s + 8 * v);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + 64 * u + lane;
                if (v >= nfull) continue;
                uint32_t e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = ((k < 4 ? q[u].x : q[u].y) >> (8 * (k & 3))) & 0xffu;
                *reinterpret_cast<uint4 *>(hs + H + 8 * v) = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
            }
        }
        if (nfull < nv && lane == 0) {  // the row's last, partial group
            const int x0 = 8 * nfull;
            uint32_t e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = x0 + k < w ? (uint32_t)s[x0 + k] : 0u;
            *reinterpret_cast<uint4 *>(hs + H + x0) = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        }
    } else {
        for (int v = lane; v < nv; v += 64) {
            const int x0 = 8 * v;
            if (x0 + 8 <= w) {
                *reinterpret_cast<uint4 *>(hs + H + x0) = *reinterpret_cast<const uint4 *>(s + x0);
                continue;
            }
            uint32_t e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = x0 + k < w ? (uint32_t)s[x0 + k] : 0u;
            *reinterpret_cast<uint4 *>(hs + H + x0) = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        }
    }
    vszip_wave_fence();
    uint16_t *cur = hs, *nxt = hs + pitch;
    for (int pass = 0; pass < npass; ++pass) {
        // blurInt's implicit padding: -k -> k - 1, w - 1 + k -> w - k
        if (lane < R) {
            cur[H - 1 - lane] = cur[H + min(lane, w - 1)];
            cur[H + w + lane] = cur[H + max(w - 1 - lane, 0)];
        }
        vszip_wave_fence();
        // E_0 = s[R] + 2 * sum_{i<R} s[i]: lane i reads sample i, one wave scan adds them (read one after another by every lane, as first written,
        // these R + 1 dependent LDS round trips and their address arithmetic were 100 of a pass's ~530 instructions - half of a chroma row's).
        // Measured and not kept: 16 outputs a lane and step instead of 8 (the halo groups unpacked and the first window summed once per 16) -
        // 323 us against 263 for 5 passes of r = 13 on 32 x 1080p (two-way conflicts of the 16-byte reads at a 32-byte lane stride, 56 unpacked samples live).
        uint32_t e0;
        {
            uint32_t x = lane <= R ? (uint32_t)cur[H + min(lane, w - 1)] : 0u;
            x = lane < R ? 2u * x : x;
            e0 = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_dpp(x), 63);
        }
        const uint32_t kr = 32768u + (uint32_t)(((uint64_t)e0 * invlo) >> 16);
        const bool last = pass == npass - 1;
        for (int v = lane; v < nv; v += 64) {
            const int x0 = 8 * v;
            uint32_t t[NT];  // t[8 G + i] = sample x0 + i: the group and G groups on either side (three 16-byte reads for R <= 8, five beyond)
            // 16-byte reads, spelled out: left to itself the compiler narrows the five reads to the 18 dwords the windows touch and issues them as
            // 8-byte ds_read2_b32 pairs at a 16-byte lane stride - four lanes a bank: 68 % of this kernel's LDS cycles were bank conflicts and the
            // LDS pipe, not the VALU, was what bounded it (profiles/r05_notes.md section 2). ds_read_b128 at a 16-byte lane stride is conflict free.
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 q[2 * G + 1];
            {
                const uint32_t a = (uint32_t)reinterpret_cast<uintptr_t>(cur + H + x0 - 8 * G);  // (the low half of a shared pointer is its LDS offset)
                if constexpr (G == 1)
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2])
                                 : "v"(a)
                                 : "memory");
                else
                    asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\tds_read_b128 %3, %5 offset:48\n\t"
                                 "ds_read_b128 %4, %5 offset:64\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[2 * G])
                                 : "v"(a)
                                 : "memory");
            }
#pragma unroll
            for (int g = 0; g < 2 * G + 1; ++g) {
                const uint32_t dw[4] = {q[g].x, q[g].y, q[g].z, q[g].w};
#pragma unroll
                for (int k = 0; k < 8; ++k) t[8 * g + k] = (dw[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
            }
            uint32_t e = 0, o[8];
#pragma unroll
            for (int k = -R; k <= R; ++k) e += t[8 * G + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                o[k] = (__umul24(e, inv2) + kr) >> 16;  // (e < 2^22 here: the 24-bit multiply is exact, the 16.16 value fits 32 bits)
                if (k < 7) e += t[8 * G + k + 1 + R] - t[8 * G + k - R];
            }
            if (!last) {
                uint4 q;
                q.x = o[0] | (o[1] << 16);
                q.y = o[2] | (o[3] << 16);
                q.z = o[4] | (o[5] << 16);
                q.w = o[6] | (o[7] << 16);
                *reinterpret_cast<uint4 *>(nxt + H + x0) = q;
            } else {
                RtVec<T>::store8(d + x0, o, min(8, w - x0), prm.keep);
            }
        }
        vszip_wave_fence();
        uint16_t *sw = cur;
        cur = nxt;
        nxt = sw;
    }
}

#ifdef VSZIP_DEV_VARIANTS  // measured slower than what the default build runs (options.inc)
// ---------------------------------------------------------------------------------------------
// ... and several VERTICAL passes of a small radius in one launch: boxblur_rt_vsmall_kernel<T, P>. A wave owns 64 x 8 columns and a band of
// output rows and pushes every input row through a chain of P stages; stage k keeps the last 2R + 2 rows of ITS input in an LDS ring (lane-
// private: no barrier), slides its window sum and hands each output row straight to stage k + 1 — the last stage stores. Every stage is
// blurInt's closed form on a column (see RtVec above): E_0 = s[R] + 2 * sum_{i<R} s[i] at the top, then E += entering - leaving, rows beyond
// the plane mirrored (-k -> k - 1, len - 1 + k -> len - k) out of the ring; dst[i] = (inv2 * E_i + 32768 + ((E_0 * invlo) >> 16)) >> 16, the
// E_0 of a stage being that of ITS input's first rows. A band that does not start at the top first runs the plane's first P * R + 1 rows
// through the chain for those constants, then warms the windows up from P * R rows above its first output row.
// ---------------------------------------------------------------------------------------------
template <typename T, int P>
struct VsCtx {
    uint32_t e[P][8], kk[P][8];
    int cnt[P], pos[P];  // rows received; ring slot of the newest row (-1: none yet)
    uint4 *ring;  // LDS, this lane's column of [stage][slot][lane]
    T *d;
    size_t ds;
    int R, D, len, y0, y1, nst, keep;
    uint32_t inv2, invlo;
    bool top;        // the chain runs from the plane's first row (E_0 sums, constants computed on the way)
    bool store_on;   // the last stage's rows go to memory
};
__device__ __forceinline__ void vs_unpack(const uint4 q, uint32_t v[8]) {
    const uint32_t dw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (dw[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
}
__device__ __forceinline__ uint4 vs_pack(const uint32_t v[8]) { return make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)); }

// a row of stage K's input arrives (rows arrive in order; r = its index): compile-time recursion down the chain
template <typename T, int P, int K>
__device__ __forceinline__ void vs_feed(VsCtx<T, P> &c, const uint32_t (&x)[8], const int r) {
    const uint4 q = vs_pack(x);
    const int R = c.R, D = c.D, len = c.len;
    auto slot = [&](int row) -> uint4 & { return c.ring[((size_t)K * D + (row % D)) * 64]; };  // (a division: only for the mirrored rows at the plane's ends)
    // rows arrive in order: the newest row's slot advances by one, and the slot after it holds the row 2R + 1 older — the one that leaves
    int ps = c.pos[K] < 0 ? r % D : (c.pos[K] + 1 == D ? 0 : c.pos[K] + 1);
    c.pos[K] = ps;
    c.ring[((size_t)K * D + ps) * 64] = q;
    const int pl = ps + 1 == D ? 0 : ps + 1;
    const int cnt = ++c.cnt[K];
    const int need = c.top ? R + 1 : 2 * R + 1;
    if (cnt <= need) {
        const uint32_t wgt = (c.top && cnt <= R) ? 2u : 1u;
#pragma unroll
        for (int i = 0; i < 8; ++i) c.e[K][i] += wgt * x[i];
        if (cnt < need) return;
        if (c.top) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c.kk[K][i] = 32768u + (uint32_t)(((uint64_t)c.e[K][i] * c.invlo) >> 16);
        }
    } else {
        const int lr = r - 2 * R - 1;  // the row that leaves the window (mirrored at the top)
        uint32_t l[8];
        vs_unpack(lr < 0 ? slot(-lr - 1) : c.ring[((size_t)K * D + pl) * 64], l);
#pragma unroll
        for (int i = 0; i < 8; ++i) c.e[K][i] += x[i] - l[i];
    }
    // the output row r - R, and after the input's last row the R rows that remain (entering rows mirrored at the bottom)
    for (int i = r - R;; ++i) {
        uint32_t o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__umul24(c.e[K][j], c.inv2) + c.kk[K][j]) >> 16;  // (radius <= 8: the window sum is below 2^21)
        if constexpr (K + 1 < P) {
            vs_feed<T, P, K + 1>(c, o, i);
        } else {
            if (c.store_on && i >= c.y0 && i < c.y1) RtVec<T>::store8(c.d + (size_t)i * c.ds, o, c.nst, c.keep);
        }
        if (r != len - 1 || i == len - 1) break;
        uint32_t a[8], l[8];
        vs_unpack(slot(2 * len - 1 - (i + 1 + R)), a);
        vs_unpack(slot(i - R), l);
#pragma unroll
        for (int j = 0; j < 8; ++j) c.e[K][j] += a[j] - l[j];
    }
}

template <typename T, int P>
__global__ __launch_bounds__(64) void boxblur_rt_vsmall_kernel(const RVParams prm) {
    extern __shared__ __attribute__((aligned(16))) uint4 vsr[];  // [stage][slot][lane]
    int pi = 0;
    const int b = blockIdx.x;
    for (int i = 1; i < prm.nplanes; ++i)
        if (b >= prm.p[i].block0) pi = i;
    const RPlane pl = prm.p[pi];
    const int lb = b - pl.block0, ncg = prm.ncg[pi], lane = (int)threadIdx.x;
    const int x0 = ((lb % ncg) * 64 + lane) * 8;
    if (x0 >= pl.w) return;
    const int len = pl.h, R = prm.radius;
    const T *s = static_cast<const T *>(pl.src) + x0;
    const size_t ss = pl.sstride;
    VsCtx<T, P> c;
    c.ring = vsr + lane;
    c.d = static_cast<T *>(pl.dst) + x0;
    c.ds = pl.dstride;
    c.R = R;
    c.D = 2 * R + 2;
    c.len = len;
    // bands of prm.band rows, the last one takes the remainder as well (half a band to a band and a half): a band must not start within R rows
    // of the bottom — its windows' warm-up sums real rows only
    const int nb = max(1, (len + prm.band / 2) / prm.band), bi = lb / ncg;
    c.y0 = bi * prm.band;
    c.y1 = bi == nb - 1 ? len : c.y0 + prm.band;
    c.nst = min(8, pl.w - x0);
    c.keep = prm.keep;
    {
        const uint32_t ksize = 2u * (uint32_t)R + 1u;
        const uint64_t inv = ((1ull << 32) + (uint64_t)R) / ksize;
        c.inv2 = (uint32_t)(inv >> 16);
        c.invlo = (uint32_t)(inv & 0xffffu);
    }
    auto load = [&](int row) -> uint4 {  // 8 samples of an input row, as packed u16
        if constexpr (sizeof(T) == 2) {
            return *reinterpret_cast<const uint4 *>(s + (size_t)row * ss);
        } else {
            const uint2 q = *reinterpret_cast<const uint2 *>(s + (size_t)row * ss);
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ((i < 4 ? q.x : q.y) >> (8 * (i & 3))) & 0xffu;
            return vs_pack(v);
        }
    };
    auto reset = [&]() {
#pragma unroll
        for (int k = 0; k < P; ++k) {
            c.cnt[k] = 0;
            c.pos[k] = -1;
#pragma unroll
            for (int i = 0; i < 8; ++i) c.e[k][i] = 0;
        }
    };
    auto run = [&](int r0, int r1) {  // input rows r0 .. r1 through the chain, eight loads in flight
        constexpr int kPf = 8;
        uint4 *stg = vsr + (size_t)P * c.D * 64 + lane;  // eight rows parked in LDS behind the rings: the chain below is one copy of code, not eight
#pragma unroll 1
        for (int rb = r0; rb <= r1; rb += kPf) {
            {
                uint4 q[kPf];
#pragma unroll
                for (int u = 0; u < kPf; ++u) q[u] = load(min(rb + u, r1));
#pragma unroll
                for (int u = 0; u < kPf; ++u) stg[u * 64] = q[u];
            }
#pragma unroll 1
            for (int u = 0; u < kPf && rb + u <= r1; ++u) {
                uint32_t x[8];
                vs_unpack(stg[u * 64], x);
                vs_feed<T, P, 0>(c, x, rb + u);
            }
        }
    };
    const int reach = P * R;
    reset();
#pragma unroll
    for (int k = 0; k < P; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) c.kk[k][i] = 0;
    c.top = true;
    c.store_on = false;
    if (c.y0 - reach > 0) {
        // the constants first: the plane's top rows until the last stage has seen its E_0 (its first output needs input rows 0 .. P * R)
        run(0, min(reach, len - 1));
        reset();
        c.top = false;
        c.store_on = true;
        run(c.y0 - reach, min(c.y1 - 1 + reach, len - 1));
    } else {
        c.store_on = true;
        run(0, min(c.y1 - 1 + reach, len - 1));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rt_transpose_kernel(const RParams prm) {
    __shared__ T tile[64][64 + 2];
    const int b = blockIdx.x;
    const RPlane pl = prm.p[rt_find(prm, b)];
    const int lb = b - pl.block0;
    const int nbx = (pl.w + 63) / 64;
    const int bx = (lb % nbx) * 64, by = (lb / nbx) * 64;
    const T *s = static_cast<const T *>(pl.src);
    T *d = static_cast<T *>(pl.dst);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4)
        if (by + r < pl.h && bx + tx < pl.w) tile[r][tx] = s[(size_t)(by + r) * pl.sstride + bx + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4)
        if (bx + r < pl.w && by + tx < pl.h) d[(size_t)(bx + r) * pl.dstride + by + tx] = tile[tx][r];
}

#endif  // VSZIP_DEV_VARIANTS
// all horizontal passes of a small radius in one launch (boxblur_rt_hsmall_kernel); false: the planes do not qualify
template <typename T>
bool hsmall_ok(const vszip_ctx *ctx, const std::vector<RPlane> &pl, int radius, int npass) {
    if (!std::is_integral<T>::value || sizeof(T) > 2 || npass < 2 || radius < 1 || radius > kHsMaxR || ctx->opt.rt_no_hsmall) return false;
    for (const RPlane &q : pl) {
        const uintptr_t bits = reinterpret_cast<uintptr_t>(q.src) | reinterpret_cast<uintptr_t>(q.dst) | (uintptr_t)((size_t)q.sstride * sizeof(T)) | (uintptr_t)((size_t)q.dstride * sizeof(T));
        if ((bits & 15) != 0 || q.w <= 2 * radius || q.w > kHsMaxW || q.sstride < ((q.w + 7) / 8) * 8 || q.dstride < ((q.w + 7) / 8) * 8) return false;
    }
    return true;
}
template <typename T>
int launch_hsmall(vszip_ctx *ctx, const std::vector<RPlane> &pl, int radius, int npass, bool keep) {
    size_t done = 0;
    while (done < pl.size()) {
        RParams prm;
        const int n = (int)std::min<size_t>(kMaxPlanesRT, pl.size() - done);
        prm.nplanes = n;
        prm.radius = radius;
        prm.keep = keep ? 1 : 0;
        int blocks = 0, maxw = 0;
        for (int i = 0; i < n; ++i) {
            prm.p[i] = pl[done + i];
            prm.p[i].block0 = blocks;
            blocks += prm.p[i].h;
            maxw = std::max(maxw, prm.p[i].w);
        }
        const int pitch = ((maxw + 7) / 8) * 8 + 2 * kHsHalo + 16;  // halo room on the left, the row in whole groups, halo room on the right, slack for the last group's reads
        const size_t lds = (size_t)2 * pitch * sizeof(uint16_t);
#define VSZIP_HS(RR) case RR: hipLaunchKernelGGL((boxblur_rt_hsmall_kernel<T, RR>), dim3(blocks), dim3(64), lds, ctx->stream, prm, npass, pitch); break
        switch (radius) {
            VSZIP_HS(1); VSZIP_HS(2); VSZIP_HS(3); VSZIP_HS(4); VSZIP_HS(5); VSZIP_HS(6); VSZIP_HS(7); VSZIP_HS(8);
            VSZIP_HS(9); VSZIP_HS(10); VSZIP_HS(11); VSZIP_HS(12); VSZIP_HS(13); VSZIP_HS(14); VSZIP_HS(15);
            default: hipLaunchKernelGGL((boxblur_rt_hsmall_kernel<T, 16>), dim3(blocks), dim3(64), lds, ctx->stream, prm, npass, pitch); break;
        }
#undef VSZIP_HS
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += n;
    }
    return VSZIP_OK;
}

#ifdef VSZIP_DEV_VARIANTS  // measured slower than what the default build runs (options.inc)
// all vertical passes of a small radius in one launch (boxblur_rt_vsmall_kernel); false: the planes do not qualify
template <typename T>
bool vsmall_ok(const vszip_ctx *ctx, const std::vector<RPlane> &pl, int radius, int npass) {
    // Measured (tools/boxblur_radii_probe.py, 1080p, 64 frames per call, against one launch per pass): two passes 93.3 k -> 107.7 k fps (u16), 105 k -> 108 k
    // (u8); three passes 68 k -> 66.5 k (u16), 79 k -> 65 k (u8) — the chain's rings and registers leave 6 waves a CU at three stages. Hence two passes
    // only (VSZIP_RT_VSMALL_MAX=3 / 4 for experiments).
    // End of round 3: OPT-IN (VSZIP_RT_VSMALL=1). The per-pass kernel's LDS ring with its software pipeline (later that round) overtook it: two vertical passes, 1080p, 64 / 16 / 4
    // frames per call, two launches against this kernel: u8 r = 1 179 k / 191 k / 60 k fps against 156 k / 160 k / 48 k, u16 147 k / 167 k / 68 k against 153 k / 160 k / 49 k; from r = 3 on
    // this kernel falls to 64 - 105 k (its rings leave few waves a CU) where two launches stay at 137 - 180 k (tools/boxblur_radii_probe.py).
    if (!ctx->opt.rt_vsmall) return false;
    const int max_pass = std::min(4, std::max(2, (int)ctx->opt.rt_vsmall_max));
    if (!std::is_integral<T>::value || sizeof(T) > 2 || npass < 2 || npass > max_pass || radius < 1 || radius > kVsMaxR || ctx->opt.rt_no_vsmall) return false;
    if ((size_t)npass * (2 * radius + 2) * 1024 > 40 * 1024) return false;  // the rings: four waves a CU at least
    for (const RPlane &q : pl) {
        const uintptr_t bits = reinterpret_cast<uintptr_t>(q.src) | reinterpret_cast<uintptr_t>(q.dst) | (uintptr_t)((size_t)q.sstride * sizeof(T)) | (uintptr_t)((size_t)q.dstride * sizeof(T));
        if ((bits & 15) != 0 || q.h <= 2 * npass * radius + 2 || q.sstride < ((q.w + 7) / 8) * 8 || q.dstride < ((q.w + 7) / 8) * 8) return false;
    }
    return true;
}
template <typename T>
int launch_vsmall(vszip_ctx *ctx, const std::vector<RPlane> &pl, int radius, int npass, bool keep) {
    size_t done = 0;
    while (done < pl.size()) {
        RVParams vp;
        const int n = (int)std::min<size_t>(kMaxPlanesRT, pl.size() - done);
        vp.nplanes = n;
        vp.radius = radius;
        vp.keep = keep ? 1 : 0;
        long colgroups = 0;
        int maxh = 0;
        for (int i = 0; i < n; ++i) {
            colgroups += (pl[done + i].w + 511) / 512;
            maxh = std::max(maxh, pl[done + i].h);
        }
        // bands: a band pays 2 * passes * radius + 1 extra rows (constants, warm-up); shorter ones only while the launch would not fill the chip
        int band = 256;
        while (band > 64 && colgroups * ((maxh + band - 1) / band) < 2048) band /= 2;
        vp.band = band;
        int vb = 0;
        for (int i = 0; i < n; ++i) {
            vp.p[i] = pl[done + i];
            vp.p[i].block0 = vb;
            vp.ncg[i] = (vp.p[i].w + 511) / 512;
            vb += vp.ncg[i] * std::max(1, (vp.p[i].h + band / 2) / band);  // (the last band takes the remainder: see the kernel)
        }
        const size_t lds = ((size_t)npass * (2 * radius + 2) + 8) * 64 * sizeof(uint4);  // the stages' rings + eight parked input rows
        if (npass == 2)
            hipLaunchKernelGGL((boxblur_rt_vsmall_kernel<T, 2>), dim3(vb), dim3(64), lds, ctx->stream, vp);
        else if (npass == 3)
            hipLaunchKernelGGL((boxblur_rt_vsmall_kernel<T, 3>), dim3(vb), dim3(64), lds, ctx->stream, vp);
        else
            hipLaunchKernelGGL((boxblur_rt_vsmall_kernel<T, 4>), dim3(vb), dim3(64), lds, ctx->stream, vp);
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += n;
    }
    return VSZIP_OK;
}

#else
template <typename T>
bool vsmall_ok(const vszip_ctx *, const std::vector<RPlane> &, int, int) { return false; }
template <typename T>
int launch_vsmall(vszip_ctx *ctx, const std::vector<RPlane> &, int, int, bool) { return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "boxblur_rt_vsmall_kernel is a development variant"); }
#endif  // VSZIP_DEV_VARIANTS
template <typename T>
int launch_pass(vszip_ctx *ctx, const std::vector<RPlane> &pl, int radius, bool vertical, bool keep) {
    constexpr bool is_int = std::is_integral<T>::value;
    // Round 4: an integer ROW pass of r <= 22 goes through the compile-time-radius ring kernel with a one-row window (boxblur_ct.hpp, CtIntDispatch::run_h:
    // the same closed form, at the fused kernel's rate) wherever that kernel takes the planes; VSZIP_BOXBLUR_NO_CT_H=1 keeps the kernels below
    if constexpr (is_int) {
        if (!vertical && radius >= 1 && radius <= 22 && !ctx->opt.boxblur_no_ct_h) {
            std::vector<vszip_plane> vp(pl.size());
            for (size_t i = 0; i < pl.size(); ++i) {
                vp[i] = vszip_plane{};
                vp[i].src = pl[i].src;
                vp[i].dst = pl[i].dst;
                vp[i].src_stride = pl[i].sstride;
                vp[i].dst_stride = pl[i].dstride;
                vp[i].w = pl[i].w;
                vp[i].h = pl[i].h;
            }
            const int rc = vszip_bb_ct_row_pass(ctx, sizeof(T) == 1 ? VSZIP_U8 : VSZIP_U16, radius, vp.data(), (int)vp.size());
            if (rc != VSZIP_ERR_UNSUPPORTED) return rc;
        }
    }
    size_t done = 0;
    while (done < pl.size()) {
        RParams prm;
        const int n = (int)std::min<size_t>(kMaxPlanesRT, pl.size() - done);
        prm.nplanes = n;
        prm.radius = radius;
        prm.keep = keep ? 1 : 0;
        int blocks = 0, maxw = 0;
        for (int i = 0; i < n; ++i) {
            prm.p[i] = pl[done + i];
            prm.p[i].block0 = blocks;
            if (is_int && !vertical)
                blocks += prm.p[i].h;
            else if (!is_int && !vertical)
                blocks += (prm.p[i].h + VSZIP_RTF_RB - 1) / VSZIP_RTF_RB;
            else if (!is_int)
                blocks += (prm.p[i].w + VSZIP_RTF_LW - 1) / VSZIP_RTF_LW;
            else
                blocks += ((vertical ? prm.p[i].w : prm.p[i].h) + 63) / 64;
            maxw = std::max(maxw, prm.p[i].w);
        }
        if constexpr (is_int) {
            bool aligned = true;
            for (int i = 0; i < n; ++i) {
                const RPlane &q = prm.p[i];
                aligned = aligned && (((reinterpret_cast<uintptr_t>(q.src) | reinterpret_cast<uintptr_t>(q.dst) | (uintptr_t)((size_t)q.sstride * sizeof(T)) |
                                        (uintptr_t)((size_t)q.dstride * sizeof(T))) & 15) == 0) && q.sstride >= ((q.w + RtVec<T>::V - 1) / RtVec<T>::V) * RtVec<T>::V;
            }
            if (aligned && !vertical) {
                constexpr int CH = 64 * RtVec<T>::V;
                if (radius < CH - 1 && !ctx->opt.rt_hrow) {
                    // virtual (mirror-extended) rows when every row is whole lane groups and wider than its halos
                    bool virt = !ctx->opt.rt_no_virt;
                    for (int i = 0; i < n; ++i) virt = virt && prm.p[i].w % RtVec<T>::V == 0 && prm.p[i].w >= radius + 1 + RtVec<T>::V;
                    if (virt)
                        hipLaunchKernelGGL((boxblur_rt_hring_kernel<T, true>), dim3(blocks), dim3(64), 0, ctx->stream, prm);
                    else
                        hipLaunchKernelGGL((boxblur_rt_hring_kernel<T, false>), dim3(blocks), dim3(64), 0, ctx->stream, prm);
                } else {
                    const size_t lds = (size_t)((maxw + CH - 1) / CH) * CH * sizeof(uint32_t);
                    if (lds > 64000)  // only the whole-row-prefix kernels keep a row in LDS; the ring kernel above serves any width
                        return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "BoxBlur: a horizontal radius of %d on rows longer than 15000 samples is not built", radius);
                    hipLaunchKernelGGL((boxblur_rt_hrow_kernel<T>), dim3(blocks), dim3(64), lds, ctx->stream, prm);
                }
            } else if (aligned) {
                RVParams vp;
                vp.nplanes = n;
                vp.radius = radius;
                vp.keep = keep ? 1 : 0;
                // bands: enough waves to fill the chip, long enough that the 3r+2 warm-up rows stay a fraction
                long colgroups = 0;
                int maxh = 0;
                for (int i = 0; i < n; ++i) {
                    colgroups += (prm.p[i].w + 64 * RtVec<T>::V - 1) / (64 * RtVec<T>::V);
                    maxh = std::max(maxh, prm.p[i].h);
                }
                int band = std::max(64, 4 * radius);
                while (band < maxh && colgroups * ((maxh + band - 1) / band) > 16384) band *= 2;
                if (ctx->opt.rt_vband > 0) band = std::max(8, (int)ctx->opt.rt_vband);  // development sweep knob (-DVSZIP_DEV_VARIANTS)
                vp.band = band;
                int vb = 0;
                for (int i = 0; i < n; ++i) {
                    vp.p[i] = prm.p[i];
                    vp.p[i].block0 = vb;
                    vp.ncg[i] = (prm.p[i].w + 64 * RtVec<T>::V - 1) / (64 * RtVec<T>::V);
                    vb += vp.ncg[i] * ((prm.p[i].h + band - 1) / band);
                }
                if (radius <= (ctx->opt.rt_vring_maxr > 0 ? (int)ctx->opt.rt_vring_maxr : kVRingMaxR) && !ctx->opt.rt_no_vring)
                    hipLaunchKernelGGL((boxblur_rt_vband_kernel<T, true>), dim3(vb), dim3(64), (size_t)(2 * radius + 2) * 64 * sizeof(uint4), ctx->stream, vp);
                else
                    hipLaunchKernelGGL((boxblur_rt_vband_kernel<T, false>), dim3(vb), dim3(64), 0, ctx->stream, vp);
            } else if (!vertical) {
                if ((size_t)maxw * sizeof(uint32_t) > 64000)
                    return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "BoxBlur: rows longer than 16000 samples need 16-byte aligned planes on the RT integer path");
                hipLaunchKernelGGL((boxblur_rt_hint_kernel<T>), dim3(blocks), dim3(256), (size_t)maxw * sizeof(uint32_t), ctx->stream, prm);
            } else {
                hipLaunchKernelGGL((boxblur_rt_vint_kernel<T>), dim3(blocks), dim3(64), 0, ctx->stream, prm);
            }
        } else {
            if (vertical)
                hipLaunchKernelGGL((boxblur_rt_float_v_kernel<T>), dim3(blocks), dim3(64), 0, ctx->stream, prm);
            else
                hipLaunchKernelGGL((boxblur_rt_float_h_kernel<T>), dim3(blocks), dim3(64), 0, ctx->stream, prm);
        }
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += n;
    }
    return VSZIP_OK;
}

// The float chain kernels: 2 ... 5 passes, lines of at least 2 R + 2 samples, rings within one workgroup's LDS.
constexpr int kFcMaxPass = 5;
template <typename T>
bool fchain_ok(const vszip_ctx *ctx, const std::vector<RPlane> &pl, int radius, int npass, bool vertical, int bands = 1) {
    constexpr bool is_int = std::is_integral<T>::value;
    if ((is_int ? ctx->opt.rt_no_ichain : ctx->opt.rt_no_fchain) || npass < 2 || npass > kFcMaxPass) return false;
    if (is_int && (!vertical || sizeof(T) > 2 || radius > 127)) return false;  // integer planes: the vertical chain only (the horizontal passes have boxblur_rt_hsmall_kernel)
    // 8-bit planes: the per-pass kernel moves 16 samples a lane and wins up to four passes and at larger radii (1080p, 64 frames per call, chain against a launch per pass:
    // r = 2 x 3 passes -11 %, 13 x 2 -19 %, 2 x 4 even, 3 x 5 +16 %; 16-bit planes: +12 ... +55 % throughout — tools/boxblur_radii_probe.py). VSZIP_RT_ICHAIN_ALL=1: every case (tests).
    if (is_int && bands < 2 && sizeof(T) == 1 && !(npass >= 5 && radius <= 8) && !ctx->opt.rt_ichain_all) return false;
    if (is_int && bands < 2 && !ctx->opt.rt_ichain_all) {
        // Two passes: two launches of the per-pass kernel are as fast (u16, r >= 3: the chain +3 ... 8 % at 64 frames) and do not mind small calls. And the chain is as slow as its
        // longest column (a wave per 64 columns, ~0.1 us a tick): with fewer waves than SIMDs — a plugin context submits ONE frame per call — the per-pass kernels, which
        // cut a plane into bands, are several times faster (4 1080p frames, three passes: ~110 us against 3 x 25 us).
        if (npass < 3) return false;
        long waves = 0;
        for (const RPlane &q : pl) waves += (q.w + 63) / 64;
        if (waves < 900) return false;
    }
    if (!is_int && !vertical && !ctx->opt.rt_fchain_all) {
        // The horizontal chain carries 64 rows a wave and costs what a row's ticks cost however few waves there are; the per-pass kernel (16 rows a wave) is faster on small calls:
        // 4K YUV420PS, 3 passes of r = 5, 1 / 2 / 4 frames per call: chain 2.5 k / 5.0 k / 9.7 k fps, a launch per pass 3.0 k / 5.7 k / 9.2 k. (The vertical chain wins from one frame on.)
        long rows = 0;
        for (const RPlane &q : pl) rows += q.h;
        if (rows < 12000) return false;
    }
    const size_t esz = sizeof(typename FcArith<T>::E);
    const size_t lds = vertical ? ((size_t)npass * (2 * radius + 3) + kFcPf + 8) * 64 * esz : (size_t)npass * (2 * radius + 3) * kFcRB * sizeof(float);
    if (lds > (vertical ? 48 : 30) * 1024) return false;  // (the horizontal kernel also holds two 64 x 64 tiles)
    for (const RPlane &q : pl) {
        if ((vertical ? q.h : q.w) < 2 * radius + 2) return false;
        if (vertical) {  // rows go four samples a lane
            const uintptr_t bits = reinterpret_cast<uintptr_t>(q.src) | reinterpret_cast<uintptr_t>(q.dst) | (uintptr_t)((size_t)q.sstride * sizeof(T)) | (uintptr_t)((size_t)q.dstride * sizeof(T));
            if ((bits & (4 * sizeof(T) - 1)) != 0 || q.sstride < ((q.w + 3) & ~3) || q.w < 4) return false;
        }
    }
    return true;
}

template <typename T>
int launch_fchain(vszip_ctx *ctx, const std::vector<RPlane> &pl, int radius, int npass, bool vertical, bool keep) {
    size_t done = 0;
    while (done < pl.size()) {
        RParams prm;
        const int n = (int)std::min<size_t>(kMaxPlanesRT, pl.size() - done);
        prm.nplanes = n;
        prm.radius = radius;
        prm.keep = keep ? 1 : 0;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            prm.p[i] = pl[done + i];
            prm.p[i].block0 = blocks;
            blocks += vertical ? (prm.p[i].w + 63) / 64 : (prm.p[i].h + kFcRB - 1) / kFcRB;
        }
        const int D = 2 * radius + 3;
        const size_t lds = vertical ? ((size_t)npass * D + kFcPf + 8) * 64 * sizeof(typename FcArith<T>::E) : (size_t)npass * D * kFcRB * sizeof(float);
#define VSZIP_FC_LAUNCH(PP)                                                                                                        \
    case PP:                                                                                                                       \
        if (vertical) {                                                                                                            \
            hipLaunchKernelGGL((boxblur_rt_float_vchain_kernel<T, PP>), dim3(blocks), dim3(64), lds, ctx->stream, prm);           \
        } else {                                                                                                                   \
            if constexpr (!std::is_integral<T>::value)                                                                             \
                hipLaunchKernelGGL((boxblur_rt_float_hchain_kernel<T, PP>), dim3(blocks), dim3(64), lds, ctx->stream, prm);       \
        }                                                                                                                          \
        break;
        switch (npass) {
            VSZIP_FC_LAUNCH(2)
            VSZIP_FC_LAUNCH(3)
            VSZIP_FC_LAUNCH(4)
            VSZIP_FC_LAUNCH(5)
        }
#undef VSZIP_FC_LAUNCH
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += n;
    }
    return VSZIP_OK;
}

// The banded integer vertical chain (boxblur_rt_ichain_kernel): the planes qualify when the unbanded chain does, are tall enough for the mirror
// extension (P (R + 1) + 1 rows) and for at least two bands of 2 P (2R + 1) rows — below that a band's warm-up is more than its rows.
// How many bands: a cost model fitted to tools/rt_band_sweep.py (gpurun_out/r4_rt_band_sweep.txt: 12 workloads x 8 band counts). A wave's time is its ticks — the band's rows
// + P (2R + 1) warm-up + P (R + 1) drain — times the tick's latency, which grows with the waves sharing a SIMD (about +36 % per extra wave: LDS hand-overs and the
// loop-carried sums leave gaps that a second wave fills only partly); the rings' LDS bounds the resident waves (five stages of r = 13: 23.7 KB a wave, six waves a CU),
// beyond which waves queue. T(nb) = max(longest wave, all wave-ticks / resident waves) x slow(waves per SIMD); the band count with the smallest T wins, 1 = no bands
// (the callers then choose between the whole-column chain and a launch per pass as before).
template <typename T>
int ichain_band_rows(const vszip_ctx *ctx, const std::vector<RPlane> &pl, int radius, int npass) {
    if (!std::is_integral<T>::value || sizeof(T) > 2 || ctx->opt.rt_no_banded) return 0;
    int maxh = 0, minh = 1 << 30;
    for (const RPlane &q : pl) {
        maxh = std::max(maxh, q.h);
        minh = std::min(minh, q.h);
    }
    if (minh < npass * (radius + 1) + 1) return 0;  // (the mirror extension reflects once)
    if (ctx->opt.rt_ichain_bands > 0) {  // (tests and sweeps: any band count)
        const int nb = std::min(ctx->opt.rt_ichain_bands, std::max(1, maxh / 8));
        return nb < 2 ? 0 : (maxh + nb - 1) / nb;
    }
    const int warm = npass * (2 * radius + 1), drain = npass * (radius + 1);
    const size_t lds = ((size_t)npass * (2 * radius + 3) + kFcPf + 8) * 64 * sizeof(uint16_t);
    const double cap = 256.0 * std::min<double>(16.0, std::floor(160.0 * 1024 / (double)lds));
    auto cost = [&](int nb) {
        const int br = (maxh + nb - 1) / nb;
        double total = 0, waves = 0, longest = 0;
        for (const RPlane &q : pl) {
            const int ncg = (q.w + 63) / 64;
            if (nb == 1) {  // whole columns: the generic ticks at both ends cost about three fast ones
                const double t = q.h + drain + 2.0 * warm;
                total += ncg * t;
                waves += ncg;
                longest = std::max(longest, t);
                continue;
            }
            const int nbp = (q.h + br - 1) / br;
            for (int b = 0; b < nbp; ++b) {
                const double t = std::min(br, q.h - b * br) + warm + drain;
                total += ncg * t;
                longest = std::max(longest, t);
            }
            waves += (double)ncg * nbp;
        }
        // (a grid search over the five constants against the sweep's 12 x 9 measurements: the model's pick is the measured optimum on nine workloads and
        // within 4 % of it on the other three — profiles/r04_rt_band_sweep.txt, tools/rt_band_fit.py)
        const double conc = std::min(waves, cap), w = conc / 1024.0;
        const double slow = 1.0 + 0.2 * std::max(0.0, w - 0.75);
        double t = std::max(longest, total / conc) * slow;
        if (waves > conc) t += 0.3 * longest * slow;  // queued waves: the last ones run beside idle slots
        return t + (nb > 1 ? 1.0 * drain : 0.0);      // (+ the launch that fills the table of constants)
    };
    int best_nb = 1;
    double best = cost(1);
    for (int nb : {2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32}) {
        if ((maxh + nb - 1) / nb < 32) break;
        const double c = cost(nb);
        if (c < best) {
            best = c;
            best_nb = nb;
        }
    }
    return best_nb < 2 ? 0 : (maxh + best_nb - 1) / best_nb;
}
template <typename T>
size_t ichain_table_bytes(const std::vector<RPlane> &pl) {
    size_t e = 0;
    for (const RPlane &q : pl) e += (size_t)((q.w + 63) & ~63) * kFcMaxPass;
    return e * sizeof(uint32_t);
}
template <typename T>
int launch_ichain_banded(vszip_ctx *ctx, const std::vector<RPlane> &pl, int radius, int npass, bool keep, int band_rows, uint32_t *kk_tab) {
    if constexpr (std::is_integral<T>::value && sizeof(T) <= 2) {
        size_t done = 0, kk_off = 0;
        while (done < pl.size()) {
            RParams prm, pe0;
            const int n = (int)std::min<size_t>(kMaxPlanesRT, pl.size() - done);
            prm.nplanes = pe0.nplanes = n;
            prm.radius = pe0.radius = radius;
            prm.keep = pe0.keep = keep ? 1 : 0;
            int blocks = 0, blocks0 = 0;
            for (int i = 0; i < n; ++i) {
                prm.p[i] = pl[done + i];
                const int ncg = (prm.p[i].w + 63) / 64;
                prm.p[i].aux = (int)kk_off;
                kk_off += (size_t)ncg * 64 * kFcMaxPass;
                pe0.p[i] = prm.p[i];
                prm.p[i].block0 = blocks;
                pe0.p[i].block0 = blocks0;
                blocks += ncg * ((prm.p[i].h + band_rows - 1) / band_rows);
                blocks0 += ncg;
            }
            const int D = 2 * radius + 3;
            const size_t lds = ((size_t)npass * D + kFcPf + 8) * 64 * sizeof(uint16_t);
#define VSZIP_IC_LAUNCH(PP)                                                                                                              \
    case PP:                                                                                                                             \
        hipLaunchKernelGGL((boxblur_rt_ichain_kernel<T, PP, 1>), dim3(blocks0), dim3(64), lds, ctx->stream, pe0, kk_tab, band_rows);     \
        hipLaunchKernelGGL((boxblur_rt_ichain_kernel<T, PP, 2>), dim3(blocks), dim3(64), lds, ctx->stream, prm, kk_tab, band_rows);      \
        break;
            switch (npass) {
                VSZIP_IC_LAUNCH(2)
                VSZIP_IC_LAUNCH(3)
                VSZIP_IC_LAUNCH(4)
                VSZIP_IC_LAUNCH(5)
            }
#undef VSZIP_IC_LAUNCH
            VSZIP_HIP_CHECK(ctx, hipGetLastError());
            done += n;
        }
        return VSZIP_OK;
    } else {
        return VSZIP_ERR_ARG;
    }
}

template <typename T>
int run_rt(vszip_ctx *ctx, const vszip_plane *planes, int nplanes, int hradius, int hpasses, int vradius, int vpasses) {
    // boxblur.zig:85-112: hpasses horizontal passes, then vpasses vertical passes
    const bool hb = hradius > 0 && hpasses > 0, vb = vradius > 0 && vpasses > 0;
    const int total = (hb ? hpasses : 0) + (vb ? vpasses : 0);
    size_t elems = 0;
    std::vector<size_t> off(nplanes);
    for (int i = 0; i < nplanes; ++i) {
        off[i] = elems;
        elems += (size_t)((planes[i].w + 63) & ~63) * planes[i].h;
    }
#ifdef VSZIP_DEV_VARIANTS  // measured slower than what the default build runs (options.inc)
    // Integer planes with >= 2 passes on an axis, OPT-IN (VSZIP_RT_FUSED=1): the fused kernel keeps a row in registers across that
    // axis's passes (horizontal directly; vertical, from 3 passes on, between two transposes: four HBM round trips for any 5 + 5
    // instead of ten). Bit-exact (tests/test_gpu_boxblur.py::test_rt_fused_multipass_equals_per_pass) — and measured SLOWER than
    // the per-pass kernels, which is why it is not the default: 5 fused passes of r = 13 on 16 1080p YUV420P16 frames take 511 us
    // against 5 x 65 us separately. The per-pass kernels are not HBM-bound but issue-bound (about 12 instructions per sample and
    // pass at 4.3 TB/s effective); a kernel that keeps the row on chip needs about 19 (the whole row's prefix has to go through
    // LDS with three barriers per pass), so saving the round trips buys nothing. The README's 5 + 5-pass benchmark therefore stays
    // at ten round trips (profiles/r03_notes.md).
    if constexpr (std::is_integral<T>::value) {
        bool ok = ctx->opt.rt_fused && total > 1;
        const bool fuse_h = hb && hpasses >= 2, fuse_v = vb && vpasses >= 3;
        ok = ok && (fuse_h || fuse_v);
        for (int i = 0; i < nplanes && ok; ++i) {
            const vszip_plane &q = planes[i];
            ok = (((reinterpret_cast<uintptr_t>(q.src) | reinterpret_cast<uintptr_t>(q.dst) | (uintptr_t)((size_t)q.src_stride * sizeof(T)) | (uintptr_t)((size_t)q.dst_stride * sizeof(T))) & 15) == 0) &&
                 (size_t)q.src_stride >= (size_t)((q.w + RtVec<T>::V - 1) / RtVec<T>::V) * RtVec<T>::V;
            if (fuse_h) ok = ok && q.w <= kHmMaxW && 2 * hradius < q.w && hradius <= kHmMaxR;
            if (fuse_v) ok = ok && q.h <= kHmMaxW && 2 * vradius < q.h && vradius <= kHmMaxR;
        }
        if (ok) {
            // scratch: two buffers that hold a plane set in either orientation (rows padded to 64 samples)
            std::vector<size_t> poff(nplanes);
            size_t pe = 0;
            for (int i = 0; i < nplanes; ++i) {
                poff[i] = pe;
                pe += std::max((size_t)((planes[i].w + 63) & ~63) * planes[i].h, (size_t)((planes[i].h + 63) & ~63) * planes[i].w);
            }
            int rc = vszip_ensure_scratch(ctx, 2 * pe * sizeof(T) + 256);
            if (rc != VSZIP_OK) return rc;
            T *buf[2] = {static_cast<T *>(ctx->scratch), static_cast<T *>(ctx->scratch) + pe};
            struct Cur { const void *ptr; int stride, w, h; };
            std::vector<Cur> cur(nplanes);
            for (int i = 0; i < nplanes; ++i) cur[i] = Cur{planes[i].src, (int)planes[i].src_stride, planes[i].w, planes[i].h};
            int which = 0;
            // one step over all planes: kind 0 = fused passes along the rows, 1 = transpose; `to_dst`: into the caller's planes
            auto step = [&](int kind, int radius, int npass, bool to_dst) -> int {
                size_t done = 0;
                while (done < (size_t)nplanes) {
                    RParams prm;
                    const int n = (int)std::min<size_t>(kMaxPlanesRT, nplanes - done);
                    prm.nplanes = n;
                    prm.radius = radius;
                    prm.keep = to_dst ? 0 : 1;
                    int blocks = 0;
                    for (int i = 0; i < n; ++i) {
                        const int gi = (int)done + i;
                        RPlane &r = prm.p[i];
                        r.src = cur[gi].ptr;
                        r.sstride = cur[gi].stride;
                        r.w = cur[gi].w;
                        r.h = cur[gi].h;
                        const int ow = kind == 1 ? cur[gi].h : cur[gi].w;
                        if (to_dst) {
                            r.dst = planes[gi].dst;
                            r.dstride = (int)planes[gi].dst_stride;
                        } else {
                            r.dst = buf[which] + poff[gi];
                            r.dstride = (ow + 63) & ~63;
                        }
                        r.block0 = blocks;
                        blocks += kind == 1 ? ((r.w + 63) / 64) * ((r.h + 63) / 64) : r.h;
                    }
                    if (kind == 1)
                        hipLaunchKernelGGL((rt_transpose_kernel<T>), dim3(blocks), dim3(256), 0, ctx->stream, prm);
                    else
                        hipLaunchKernelGGL((boxblur_rt_hmulti_kernel<T>), dim3(blocks), dim3(kHmNT), 0, ctx->stream, prm, npass);
                    VSZIP_HIP_CHECK(ctx, hipGetLastError());
                    for (int i = 0; i < n; ++i) {
                        const int gi = (int)done + i;
                        const RPlane &r = prm.p[i];
                        cur[gi] = kind == 1 ? Cur{r.dst, r.dstride, r.h, r.w} : Cur{r.dst, r.dstride, r.w, r.h};
                    }
                    done += n;
                }
                which ^= 1;
                return VSZIP_OK;
            };
            // the per-pass kernels on the current planes (an axis the fused kernel does not take), into scratch or the caller's planes
            auto per_pass = [&](int radius, int npass, bool vertical, bool last_axis) -> int {
                for (int p = 0; p < npass; ++p) {
                    const bool to_dst = last_axis && p == npass - 1;
                    std::vector<RPlane> v(nplanes);
                    for (int i = 0; i < nplanes; ++i) {
                        v[i].src = cur[i].ptr;
                        v[i].sstride = cur[i].stride;
                        v[i].w = cur[i].w;
                        v[i].h = cur[i].h;
                        v[i].dst = to_dst ? planes[i].dst : static_cast<void *>(buf[which] + poff[i]);
                        v[i].dstride = to_dst ? (int)planes[i].dst_stride : ((cur[i].w + 63) & ~63);
                    }
                    const int prc = launch_pass<T>(ctx, v, radius, vertical, !to_dst);
                    if (prc != VSZIP_OK) return prc;
                    for (int i = 0; i < nplanes; ++i) cur[i] = Cur{v[i].dst, v[i].dstride, v[i].w, v[i].h};
                    which ^= 1;
                }
                return VSZIP_OK;
            };
            if (hb) {
                rc = fuse_h ? step(0, hradius, hpasses, !vb) : per_pass(hradius, hpasses, false, !vb);
                if (rc != VSZIP_OK) return rc;
            }
            if (vb) {
                if (fuse_v) {
                    if ((rc = step(1, 0, 0, false)) != VSZIP_OK) return rc;
                    if ((rc = step(0, vradius, vpasses, false)) != VSZIP_OK) return rc;
                    if ((rc = step(1, 0, 0, true)) != VSZIP_OK) return rc;
                } else if ((rc = per_pass(vradius, vpasses, true, true)) != VSZIP_OK) {
                    return rc;
                }
            }
            return VSZIP_OK;
        }
    }
#endif  // VSZIP_DEV_VARIANTS
    // Several passes: the planes go through ALL passes in groups small enough that a pass's output is still in the Infinity
    // Cache (256 MiB, memory side) when the next pass reads it — a line stays resident while everything touched between its two
    // uses fits (MI355X_MICROARCH.md, Infinity Cache): about three group sizes here, so groups of <= 48 MB. HBM then sees the
    // first read and the last write; the passes in between run cache to cache (their stores drop the nt hint). One pass, or
    // VSZIP_RT_GROUP_MB=0: one group.
    // (measured, round 3: 4K YUV420P16 3 + 3 passes of r = 5, groups of 24 / 48 / 96 MB: 8.4 k / 12.8 k / 13.6 k fps against 13.5 k in one
    // group — the passes are not HBM-bound enough for residency to beat the smaller launches; the default is ONE group)
    size_t group_bytes = ~(size_t)0;
    if (ctx->opt.rt_group_mb > 0) group_bytes = (size_t)ctx->opt.rt_group_mb << 20;  // (-DVSZIP_DEV_VARIANTS)
    int g0 = 0;
    while (g0 < nplanes) {
        int g1 = g0;
        size_t bytes = 0, elems_g = 0;
        while (g1 < nplanes && (g1 == g0 || bytes + (size_t)planes[g1].w * planes[g1].h * sizeof(T) <= group_bytes)) {
            bytes += (size_t)planes[g1].w * planes[g1].h * sizeof(T);
            elems_g += (size_t)((planes[g1].w + 63) & ~63) * planes[g1].h;
            ++g1;
        }
        const int ng = g1 - g0;
        T *scratch[2] = {nullptr, nullptr};
        uint32_t *kk_tab = nullptr;  // the banded integer chain's per-column constants
        std::vector<size_t> goff(ng);
        {
            size_t e = 0;
            for (int i = 0; i < ng; ++i) {
                goff[i] = e;
                e += (size_t)((planes[g0 + i].w + 63) & ~63) * planes[g0 + i].h;
            }
        }
        if (total > 1) {
            // (grow-only scratch sized for the largest group seen; every group reuses the same two buffers)
            const size_t planes_bytes = (2 * elems_g * sizeof(T) + 255) & ~(size_t)255;
            size_t tab_bytes = 0;
            if (std::is_integral<T>::value && vb && vpasses >= 2)
                for (int i = 0; i < ng; ++i) tab_bytes += (size_t)((planes[g0 + i].w + 63) & ~63) * kFcMaxPass * sizeof(uint32_t);
            int rc = vszip_ensure_scratch(ctx, planes_bytes + tab_bytes + 256);
            if (rc != VSZIP_OK) return rc;
            scratch[0] = static_cast<T *>(ctx->scratch);
            scratch[1] = scratch[0] + elems_g;
            kk_tab = reinterpret_cast<uint32_t *>(static_cast<char *>(ctx->scratch) + planes_bytes);
        }
        std::vector<RPlane> cur(ng);
        for (int i = 0; i < ng; ++i) {
            cur[i].src = planes[g0 + i].src;
            cur[i].sstride = (int)planes[g0 + i].src_stride;
            cur[i].w = planes[g0 + i].w;
            cur[i].h = planes[g0 + i].h;
        }
        int which = 0;
        for (int p = 0; p < total; ++p) {
            const bool vertical = p >= (hb ? hpasses : 0);
            // every horizontal pass of a small radius at once (round 3): the step then stands for hpasses passes
            int span = 1;
            if constexpr (!std::is_integral<T>::value) {
                // float: what is left of this axis's passes in chains of up to kFcMaxPass stages (6 = 3 + 3 rather than 5 + 1: a lone pass is a launch of its own)
                const int rem = vertical ? total - p : hpasses - p;
                if (rem >= 2) span = rem <= kFcMaxPass ? rem : (rem - kFcMaxPass == 1 ? kFcMaxPass - 1 : kFcMaxPass);
            } else if constexpr (sizeof(T) <= 2) {
                if (p == 0 && hb && hpasses >= 2) span = hpasses;
                if (vertical) {
                    // What is left of the vertical passes, in chains of nearly equal length. A chain of P stages costs every row band P (3 R + 2) rows of
                    // warm-up and drain and P (2 R + 3) ring rows of LDS, so short planes take shorter chains and pay another trip through memory instead:
                    // at most a quarter of the shortest plane's rows may be warm-up (r = 13 on 1080p 4:2:0, 540-row chroma: 5 passes as 3 + 2 - 292 us
                    // against 343 for one chain of 5 on 32 frames, 4 as 2 + 2 - 218 against 234; 4K keeps its chain of 5: 246 against 289;
                    // tools/rt_pass_scaling.py)
                    const int rem = total - p;
                    int minh = 1 << 30;
                    for (int i = 0; i < ng; ++i) minh = std::min(minh, cur[i].h);
                    const int pmax = std::max(2, std::min(kFcMaxPass, minh / (4 * (3 * vradius + 2))));
                    if (rem >= 2) {
                        const int nchains = (rem + pmax - 1) / pmax;
                        span = (rem + nchains - 1) / nchains;
                    }
                }
            }
            const bool last = p + span - 1 == total - 1;
            for (int i = 0; i < ng; ++i) {
                if (last) {
                    cur[i].dst = planes[g0 + i].dst;
                    cur[i].dstride = (int)planes[g0 + i].dst_stride;
                } else {
                    cur[i].dst = scratch[which] + goff[i];
                    cur[i].dstride = (planes[g0 + i].w + 63) & ~63;
                }
            }
            int rc;
            if (!std::is_integral<T>::value && span > 1 && fchain_ok<T>(ctx, cur, vertical ? vradius : hradius, span, vertical)) {
                if constexpr (!std::is_integral<T>::value) rc = launch_fchain<T>(ctx, cur, vertical ? vradius : hradius, span, vertical, !last);
                else rc = VSZIP_ERR_ARG;
                p += span - 1;
            } else if (const int br = (std::is_integral<T>::value && span > 1 && vertical && kk_tab && !vsmall_ok<T>(ctx, cur, vradius, span)) ? ichain_band_rows<T>(ctx, cur, vradius, span) : 0;
                       br > 0 && fchain_ok<T>(ctx, cur, vradius, span, true, (cur[0].h + br - 1) / br)) {
                // integer planes: the chain in bands of rows (exact under any segmentation: see boxblur_rt_ichain_kernel)
                rc = launch_ichain_banded<T>(ctx, cur, vradius, span, !last, br, kk_tab);
                p += span - 1;
            } else if (std::is_integral<T>::value && span > 1 && vertical && !vsmall_ok<T>(ctx, cur, vradius, span) && fchain_ok<T>(ctx, cur, vradius, span, true)) {
                // integer planes, vertical passes the two-stage small-radius kernel does not take: the pass chain (one column a lane, the stages' rings in LDS)
                if constexpr (std::is_integral<T>::value && sizeof(T) <= 2) rc = launch_fchain<T>(ctx, cur, vradius, span, true, !last);
                else rc = VSZIP_ERR_ARG;
                p += span - 1;
            } else if (span > 1 && !vertical && hsmall_ok<T>(ctx, cur, hradius, hpasses)) {
                if constexpr (std::is_integral<T>::value && sizeof(T) <= 2) rc = launch_hsmall<T>(ctx, cur, hradius, hpasses, !last);
                else rc = VSZIP_ERR_ARG;
                p += span - 1;
            } else if (span > 1 && vertical && vsmall_ok<T>(ctx, cur, vradius, span)) {
                if constexpr (std::is_integral<T>::value && sizeof(T) <= 2) rc = launch_vsmall<T>(ctx, cur, vradius, span, !last);
                else rc = VSZIP_ERR_ARG;
                p += span - 1;
            } else {
                if (span > 1) {  // (the planes do not qualify: back to one pass per launch; this one is not the last)
                    span = 1;
                    for (int i = 0; i < ng; ++i) {
                        if (total > 1) {
                            cur[i].dst = scratch[which] + goff[i];
                            cur[i].dstride = (planes[g0 + i].w + 63) & ~63;
                        }
                    }
                }
                const bool last1 = p == total - 1;
                rc = launch_pass<T>(ctx, cur, vertical ? vradius : hradius, vertical, !last1);
            }
            if (rc != VSZIP_OK) return rc;
            for (int i = 0; i < ng; ++i) {
                cur[i].src = cur[i].dst;
                cur[i].sstride = cur[i].dstride;
            }
            which ^= 1;
        }
        g0 = g1;
    }
    return VSZIP_OK;
}

}  // namespace

int vszip_bb_rt(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, int hradius, int hpasses, int vradius, int vpasses) {
    switch (dtype) {
        case VSZIP_U8: return run_rt<uint8_t>(ctx, planes, nplanes, hradius, hpasses, vradius, vpasses);
        case VSZIP_U16: return run_rt<uint16_t>(ctx, planes, nplanes, hradius, hpasses, vradius, vpasses);
        case VSZIP_F16: return run_rt<_Float16>(ctx, planes, nplanes, hradius, hpasses, vradius, vpasses);
        case VSZIP_F32: return run_rt<float>(ctx, planes, nplanes, hradius, hpasses, vradius, vpasses);
    }
    return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: not supported Int format.");
}
