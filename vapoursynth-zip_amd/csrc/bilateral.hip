// vszip.Bilateral on gfx950.
//
// Replaces filter.bilateral (src/filters/bilateral.zig:81-89) -> truncated (:178-304,
// algorithm 2: truncated spatial window with sub-sampling) and the LUT generators
// (:306-339); parameter derivation follows src/vapoursynth/bilateral.zig:104-231.
// Algorithm 1 (PBFIC, :91-171) keeps every range layer resident in HBM; see below.
//
// The (radius/step)^2 diagonal-quadrant taps are accumulated in exactly the
// reference's order with unfused f32 multiplies and adds (-ffp-contract=off) and an
// IEEE division, so integer outputs are bit-exact and float outputs identical.
//
// bilateral_tiled_kernel (radius <= 16): a 256-thread workgroup stages a 64 x 32
// output tile plus its halo in LDS with replicate padding baked in (truncatedEdges'
// coordinate clamp, :281-289, becomes the identity), then each thread walks 8 rows.
// The kernel is bound by the texture-address unit — one vector-memory instruction per
// tap per wave — so moving the 16 (32 with a joint `ref` clip) pixel taps per pixel to
// LDS reads leaves only the range-LUT gathers on that path. The range LUT (256 KiB
// for 16-bit and float input, too large for LDS) is gathered through L1/L2, where
// natural content keeps the low-difference end hot.
// bilateral_truncated_kernel (any radius): one thread per pixel, taps as cached loads.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <utility>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kMaxPlanesBL = 192;  // planes per launch (64 YUV frames): the persistent kernels pay their table load and their tail once per launch

struct BLPlane {
    const void *src, *ref;
    void *dst;
    int sstride, rstride, dstride;
    int w, h;
    int block0, nbx;
    int radius, step;
    const float *gs, *gr;
};

struct BLParams {
    BLPlane p[kMaxPlanesBL];
    int nplanes;
    float peak;
    int lut_len, lut_offset;  // tiled kernel with the range LUT in LDS: entries, byte offset behind the tiles
    int lut_upper;            // walk kernel, PLATEAU form: the table is constant from this entry on (gaussianFunctionRangeLUTGeneration's `upper`)
};

template <typename T>
struct BSmp;
template <>
struct BSmp<uint8_t> {
    static constexpr bool is_int = true;
    static __device__ __forceinline__ float f(uint8_t v) { return (float)v; }
    static __device__ __forceinline__ uint32_t ridx(uint8_t a, uint8_t b) { return a > b ? (uint32_t)(a - b) : (uint32_t)(b - a); }
};
template <>
struct BSmp<uint16_t> {
    static constexpr bool is_int = true;
    static __device__ __forceinline__ float f(uint16_t v) { return (float)v; }
    // |a - b| in one instruction: v_sad_u16 sums the absolute differences of the two 16-bit halves (upper halves are 0)
    static __device__ __forceinline__ uint32_t ridx(uint16_t a, uint16_t b) { return __builtin_amdgcn_sad_u16((uint32_t)a, (uint32_t)b, 0u); }
};
template <>
struct BSmp<float> {
    static constexpr bool is_int = false;
    static __device__ __forceinline__ float f(float v) { return v; }
    // bilateral.zig:15-22
    static __device__ __forceinline__ uint32_t ridx(float a, float b) { return (uint32_t)truncf(fminf(1.0f, fabsf(a - b)) * 65535.0f + 0.5f); }
};
template <>
struct BSmp<_Float16> {
    static constexpr bool is_int = false;
    static __device__ __forceinline__ float f(_Float16 v) { return (float)v; }
    static __device__ __forceinline__ uint32_t ridx(_Float16 a, _Float16 b) {
        const _Float16 d = a - b;  // |a - b| in f16, widened
        return (uint32_t)truncf(fminf(1.0f, fabsf((float)d)) * 65535.0f + 0.5f);
    }
};

constexpr int kBX = 64, kBY = 4;  // one wave per row segment
constexpr int kTileH = 32;        // tiled kernel: output rows per workgroup (8 per thread)
constexpr int kTileMaxR = 16;     // largest radius the tiled kernel stages

// LDSLUT: the range LUT has at most kLdsLutMax entries (8/10/12-bit clips) and is staged in LDS
// behind the tiles, which takes the gathers off the texture-address path as well.
constexpr int kLdsLutMax = 4096;

template <typename T, bool JOINT, bool LDSLUT>
__global__ __launch_bounds__(kBX *kBY) void bilateral_tiled_kernel(const BLParams prm) {
    using S = BSmp<T>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int pi = 0;
    const int b = blockIdx.x;
#pragma unroll 1
    for (int i = 1; i < prm.nplanes; ++i)
        if (b >= prm.p[i].block0) pi = i;
    const BLPlane pl = prm.p[pi];
    const int lb = b - pl.block0;
    const int x0 = (lb % pl.nbx) * kBX, y0 = (lb / pl.nbx) * kTileH;
    const int r = pl.radius;
    const int tw = kBX + 2 * r, th = kTileH + 2 * r;
    const T *src = static_cast<const T *>(pl.src);
    const T *ref = static_cast<const T *>(pl.ref);
    T *tr = reinterpret_cast<T *>(smem);
    T *ts = JOINT ? tr + th * tw : tr;
    const int w1 = pl.w - 1, h1 = pl.h - 1;
    const int lx = (int)threadIdx.x, tyi = (int)threadIdx.y;
    const float *gr = pl.gr;
    if constexpr (LDSLUT) {
        float *lut = reinterpret_cast<float *>(smem + prm.lut_offset);
        for (int i = tyi * kBX + lx; i < prm.lut_len; i += kBX * kBY) lut[i] = pl.gr[i];
        gr = lut;
    }

    // stage: tile row t holds source row clamp(y0 - r + t), tile column c source column clamp(x0 - r + c)
    for (int t = tyi; t < th; t += kBY) {
        const int gy = min(max(y0 - r + t, 0), h1);
        const T *rr = ref + (size_t)gy * pl.rstride;
        const T *sr = src + (size_t)gy * pl.sstride;
        for (int c = lx; c < tw; c += kBX) {
            const int gx = min(max(x0 - r + c, 0), w1);
            tr[t * tw + c] = rr[gx];
            if constexpr (JOINT) ts[t * tw + c] = sr[gx];
        }
    }
    __syncthreads();

    const int x = x0 + lx;
    if (x >= pl.w) return;
    T *dst = static_cast<T *>(pl.dst);
    const float *gs = pl.gs;
    const int radius2 = r + 1, step = pl.step;
    const float w0 = gs[0] * gr[0];
#pragma unroll 1
    for (int k = 0; k < kTileH / kBY; ++k) {
        const int ly = tyi + kBY * k;  // a wave owns one row of 64 pixels at a time
        const int y = y0 + ly;
        if (y >= pl.h) break;
        const int c0 = (ly + r) * tw + lx + r;
        const T cx = tr[c0];
        float wsum = w0;
        float sum = S::f(ts[c0]) * wsum;
        for (int yy = 1; yy < radius2; yy += step) {
            const int oa = c0 - yy * tw, ob = c0 + yy * tw;
            for (int xx = 1; xx < radius2; xx += step) {
                const float swei = gs[yy * radius2 + xx];
                const T ra1 = tr[oa + xx], ra2 = tr[ob + xx], ra3 = tr[oa - xx], ra4 = tr[ob - xx];
                const float rw1 = gr[S::ridx(cx, ra1)];
                const float rw2 = gr[S::ridx(cx, ra2)];
                const float rw3 = gr[S::ridx(cx, ra3)];
                const float rw4 = gr[S::ridx(cx, ra4)];
                wsum += swei * (rw1 + rw2 + rw3 + rw4);
                if constexpr (JOINT)
                    sum += swei * (S::f(ts[oa + xx]) * rw1 + S::f(ts[ob + xx]) * rw2 + S::f(ts[oa - xx]) * rw3 + S::f(ts[ob - xx]) * rw4);
                else
                    sum += swei * (S::f(ra1) * rw1 + S::f(ra2) * rw2 + S::f(ra3) * rw3 + S::f(ra4) * rw4);
            }
        }
        const float q = __fdiv_rn(sum, wsum);
        if constexpr (S::is_int) {
            const float v = fminf(fmaxf(q + 0.5f, 0.0f), prm.peak);  // finalize :30-36
            dst[(size_t)y * pl.dstride + x] = (T)truncf(v);
        } else {
            dst[(size_t)y * pl.dstride + x] = (T)q;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 16-bit / float clips: the WHOLE 65536-entry range LUT in LDS, in an exact compressed form.
// gr_lut is monotone non-increasing in the index, and so are the bit patterns of its (non-negative) f32
// entries: bits(gr[i]) = base[i >> 6] - delta[i] with base = the block's first entry (u32 x 1024 = 4 KiB) and
// delta a u16 (128 KiB) reproduces every bit whenever no 64-entry block spans more than 65535 ulps —
// vszip_bilateral_luts verifies all 65536 entries and registers the table only then (steep small-sigmaR tables
// keep the gathered path). 132 KiB of a CU's 160 KiB: one persistent 1024-thread workgroup per CU walks the
// tiles. Measured (tools/lds_gather_probe.hip, per 64-lane lookup and CU): 38 / 68 / 198 cycles for the gather
// from L1/L2 on natural / edgy / uniform index distributions against 14 / 15 / 18 for the two LDS reads.
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
// The LDS pipeline is this kernel's limit, so part of the lookups goes to the f32 table in global memory: the
// vector-memory pipeline (L1/L2) runs beside it. How many pays depends on the content - smooth pictures keep the
// touched part of the table in L1, heavy grain does not. Measured (1080p YUV420P16, k fps) at 0 / 2 / 3 / 4 / 6 / 8
// of every 16 lookups: natural 37.7 / 40.0 / 41.4 / 42.5 / 39.2 / 33.2; + grain sigma 1500 LSB 35.8 / 38.2 / 38.9 /
// 34.4 / 25.8 / 20.6; full-range white noise 33.8 / 35.6 / 27.1 / 21.4 / 15.1 / 11.6. 3 wins or ties on everything
// a camera produces; a per-wave choice between 4 and 2 from the previous pass's gradients was slower than either
// (two copies of the loop, 39.2 natural).
#ifndef VSZIP_L16_SPLIT
#define VSZIP_L16_SPLIT 3
#endif
constexpr int kL16Split = VSZIP_L16_SPLIT;
constexpr int kL16Rows = 16;                       // thread rows of the workgroup: 64 x 16 = 1024 threads
// output rows per tile: 64 (4 per wave) for 2-byte samples; f32 tiles, and the tile pairs of a joint (`ref`) clip,
// are half as high so that two buffers of them still fit beside the table (2 x 70 x 38 x 4 B at radius 3)
template <typename T, bool JOINT>
constexpr int kL16TileH = (sizeof(T) == 4 || JOINT) ? 32 : 64;
constexpr int kL16LutBytes = 4096 + 131072;        // base + delta
constexpr int kL16MaxLds = 160 * 1024;
constexpr int kWalkFineBytes = 65536 + 65536;      // the 4-entry-block form of the table: base4 u32 x 16384 + delta8 u8 x 65536
// staged samples per thread and clip: tiles up to 1024 * kL16MaxStage samples
template <typename T, bool JOINT>
constexpr int kL16MaxStage = (sizeof(T) == 4 || JOINT) ? 4 : 6;

// CR / CS > 0: every plane of the launch has this radius / step (the BASELINE's luma 3 / 2 and chroma 2 / 1):
// the tap loops unroll and every LDS offset becomes an immediate; 0 = read them from the plane table.
// PLAT (round 3): the table is a steep one kept AS IT IS up to prm.lut_upper, where the reference stops computing it (see WalkLut<2> below) —
// for the tap shapes the walk kernel does not have (the filter's default sigmaS = 3: radius 5) and for `ref` clips.
template <typename T, bool JOINT, int CR, int CS, bool PLAT = false>
__global__ __launch_bounds__(kBX *kL16Rows) void bilateral_lds16_kernel(const BLParams prm, const int nblocks) {
    using S = BSmp<T>;
    // the LUT is a STATIC allocation (addresses fold into the ds_read offsets); the tile buffers follow it
    __shared__ __attribute__((aligned(16))) struct { uint32_t base[1024]; uint16_t delta[65536]; } slut;
    uint32_t *sbase = slut.base;
    uint16_t *sdelta = slut.delta;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *tiles = reinterpret_cast<T *>(smem);
    const int lx = (int)threadIdx.x, tyi = (int)threadIdx.y, tid = tyi * kBX + lx;
    constexpr int NT = kBX * kL16Rows;
    const float *splain = reinterpret_cast<const float *>(&slut);  // PLAT: the same 132 KiB hold up to 32 768 plain entries
    const uint32_t upper = (uint32_t)prm.lut_upper;
    if constexpr (PLAT) {
        const uint4 *g = reinterpret_cast<const uint4 *>(prm.p[0].gr);
        for (int i = tid; i < ((int)upper + 4) / 4; i += NT) reinterpret_cast<uint4 *>(&slut)[i] = g[i];
    } else {
        // the planes of one launch share one range LUT (the host groups them by table)
        const uint32_t *gb = reinterpret_cast<const uint32_t *>(prm.p[0].gr + 65536);
        const uint4 *gd = reinterpret_cast<const uint4 *>(gb + 1024);
        sbase[tid] = gb[tid];
        for (int i = tid; i < 65536 / 8; i += NT) reinterpret_cast<uint4 *>(sdelta)[i] = gd[i];
    }
    auto lut = [&](uint32_t i) {
        if constexpr (PLAT)
            return splain[min(i, upper)];
        else
            return __uint_as_float(sbase[i >> 6] - (uint32_t)sdelta[i]);
    };
    // Only ONE workgroup fits a CU, so nothing else hides a tile's global loads: the next tile is fetched into
    // registers while this one is filtered, and the two LDS tile buffers alternate (one barrier per tile).
    const int tile_elems_max = prm.lut_offset;  // (reused field) samples of the largest tile of the launch, per clip
    T pre_r[kL16MaxStage<T, JOINT>], pre_s[JOINT ? kL16MaxStage<T, JOINT> : 1];
    int pi = 0;
    auto plane_of = [&](int b) {
        while (pi + 1 < prm.nplanes && b >= prm.p[pi + 1].block0) ++pi;  // tiles are visited in increasing order
        return pi;
    };
    auto fetch = [&](int b) {
        const BLPlane &pl = prm.p[plane_of(b)];
        const int lb = b - pl.block0;
        const int x0 = (lb % pl.nbx) * kBX, y0 = (lb / pl.nbx) * kL16TileH<T, JOINT>;
        const int r = CR > 0 ? CR : pl.radius, tw = kBX + 2 * r, n = tw * (kL16TileH<T, JOINT> + 2 * r);
        const T *src = static_cast<const T *>(pl.src), *ref = static_cast<const T *>(pl.ref);
#pragma unroll
        for (int k = 0; k < kL16MaxStage<T, JOINT>; ++k) {
            const int i = min(tid + k * NT, n - 1);
            const int t = i / tw, c = i - t * tw;
            const int gy = min(max(y0 - r + t, 0), pl.h - 1), gx = min(max(x0 - r + c, 0), pl.w - 1);
            pre_r[k] = ref[(size_t)gy * pl.rstride + gx];
            if constexpr (JOINT) pre_s[k] = src[(size_t)gy * pl.sstride + gx];
        }
    };
    int b = blockIdx.x, buf = 0;
    if (b < nblocks) fetch(b);
#pragma unroll 1
    for (; b < nblocks; b += gridDim.x, buf ^= 1) {
        const BLPlane &pl = prm.p[plane_of(b)];
        const int lb = b - pl.block0;
        const int x0 = (lb % pl.nbx) * kBX, y0 = (lb / pl.nbx) * kL16TileH<T, JOINT>;
        const int r = CR > 0 ? CR : pl.radius;
        const int tw = kBX + 2 * r, th = kL16TileH<T, JOINT> + 2 * r, n = tw * th;
        T *tr = tiles + (size_t)buf * tile_elems_max * (JOINT ? 2 : 1);
        T *ts = JOINT ? tr + tile_elems_max : tr;
#pragma unroll
        for (int k = 0; k < kL16MaxStage<T, JOINT>; ++k) {
            const int i = tid + k * NT;
            if (i < n) {
                tr[i] = pre_r[k];
                if constexpr (JOINT) ts[i] = pre_s[k];
            }
        }
        __syncthreads();  // (also: the LUT is in place the first time; the other buffer's readers finished a tile ago)
        const int pi_here = pi;
        if (b + (int)gridDim.x < nblocks) fetch(b + gridDim.x);  // in flight while this tile is filtered
        const BLPlane &pc = prm.p[pi_here];
        const int x = x0 + lx;
        if (x < pc.w) {
            T *dst = static_cast<T *>(pc.dst);
            const float *gs = pc.gs;
            const int radius2 = r + 1, step = CS > 0 ? CS : pc.step;
            const float w0 = gs[0] * lut(0);
            [[maybe_unused]] const float *__restrict__ grg = prm.p[0].gr;
            // one (yy, xx) quadrant set: 4 diagonal taps, the reference's operation order (:236-264)
            auto taps = [&](const T cx, int c0, int yy, int xx, float swei, float &wsum, float &sum) {
                const int oa = c0 - yy * tw, ob = c0 + yy * tw;
                const T ra1 = tr[oa + xx], ra2 = tr[ob + xx], ra3 = tr[oa - xx], ra4 = tr[ob - xx];
                const float rw1 = lut(S::ridx(cx, ra1));
                const float rw2 = lut(S::ridx(cx, ra2));
                const float rw3 = lut(S::ridx(cx, ra3));
                const float rw4 = lut(S::ridx(cx, ra4));
                wsum += swei * (rw1 + rw2 + rw3 + rw4);
                if constexpr (JOINT)
                    sum += swei * (S::f(ts[oa + xx]) * rw1 + S::f(ts[ob + xx]) * rw2 + S::f(ts[oa - xx]) * rw3 + S::f(ts[ob - xx]) * rw4);
                else
                    sum += swei * (S::f(ra1) * rw1 + S::f(ra2) * rw2 + S::f(ra3) * rw3 + S::f(ra4) * rw4);
            };
            float sw[CR > 0 ? (CR + 1) * (CR + 1) : 1];
            if constexpr (CR > 0) {
#pragma unroll
                for (int i = 0; i < (CR + 1) * (CR + 1); ++i) sw[i] = gs[i];
            }
            auto finish = [&](float sum, float wsum, int y) {
                const float q = __fdiv_rn(sum, wsum);
                if constexpr (S::is_int) {
                    const float v = fminf(fmaxf(q + 0.5f, 0.0f), prm.peak);  // finalize :30-36
                    dst[(size_t)y * pc.dstride + x] = (T)truncf(v);
                } else {
                    dst[(size_t)y * pc.dstride + x] = (T)q;
                }
            };
            if constexpr (CR > 0 && CS > 0) {
                // Two output rows per pass: the lookups stay per sample, the weight arithmetic of the pair runs as
                // packed f32 (v_pk_mul / v_pk_add: IEEE per element, so the reference's operation order holds).
                auto taps2 = [&](const T cxa, const T cxb, int ca, int cb, int yy, int xx, int ng, float swei, v2f &wsum, v2f &sum) {
                    const int oa = -yy * tw, ob = yy * tw;
                    const T a1 = tr[ca + oa + xx], a2 = tr[ca + ob + xx], a3 = tr[ca + oa - xx], a4 = tr[ca + ob - xx];
                    const T b1 = tr[cb + oa + xx], b2 = tr[cb + ob + xx], b3 = tr[cb + oa - xx], b4 = tr[cb + ob - xx];
                    // ng of the 4 lookups go to the f32 table in global memory (L1/L2): the vector-memory pipeline
                    // runs beside the LDS one, which is this kernel's limit
                    auto look = [&](uint32_t i, bool global) { return global ? grg[i] : lut(i); };
                    const v2f rw1 = {look(S::ridx(cxa, a1), ng > 3), look(S::ridx(cxb, b1), ng > 3)};
                    const v2f rw2 = {look(S::ridx(cxa, a2), ng > 2), look(S::ridx(cxb, b2), ng > 2)};
                    const v2f rw3 = {look(S::ridx(cxa, a3), ng > 1), look(S::ridx(cxb, b3), ng > 1)};
                    const v2f rw4 = {look(S::ridx(cxa, a4), ng > 0), look(S::ridx(cxb, b4), ng > 0)};
                    const v2f sv = {swei, swei};
                    wsum += sv * (rw1 + rw2 + rw3 + rw4);
                    v2f f1, f2, f3, f4;
                    if constexpr (JOINT) {
                        f1 = v2f{S::f(ts[ca + oa + xx]), S::f(ts[cb + oa + xx])};
                        f2 = v2f{S::f(ts[ca + ob + xx]), S::f(ts[cb + ob + xx])};
                        f3 = v2f{S::f(ts[ca + oa - xx]), S::f(ts[cb + oa - xx])};
                        f4 = v2f{S::f(ts[ca + ob - xx]), S::f(ts[cb + ob - xx])};
                    } else {
                        f1 = v2f{S::f(a1), S::f(b1)};
                        f2 = v2f{S::f(a2), S::f(b2)};
                        f3 = v2f{S::f(a3), S::f(b3)};
                        f4 = v2f{S::f(a4), S::f(b4)};
                    }
                    sum += sv * (f1 * rw1 + f2 * rw2 + f3 * rw3 + f4 * rw4);
                };
#pragma unroll 1
                for (int k = 0; k < kL16TileH<T, JOINT> / kL16Rows; k += 2) {
                    const int lya = tyi + kL16Rows * k, lyb = lya + kL16Rows;
                    const int ya = y0 + lya, yb = y0 + lyb;
                    if (ya >= pc.h) break;
                    const int ca = (lya + r) * tw + lx + r, cb = (lyb + r) * tw + lx + r;
                    const T cxa = tr[ca], cxb = tr[cb];
                    v2f wsum = {w0, w0};
                    v2f sum = v2f{S::f(ts[ca]), S::f(ts[cb])} * wsum;
                    int call = 0;
#pragma unroll
                    for (int yy = 1; yy <= CR; yy += CS)
#pragma unroll
                        for (int xx = 1; xx <= CR; xx += CS, ++call) {
                            const int ng = (kL16Split * (call + 1)) / 4 - (kL16Split * call) / 4;  // spread evenly over the tap sets
                            taps2(cxa, cxb, ca, cb, yy, xx, ng, sw[yy * (CR + 1) + xx], wsum, sum);
                        }
                    finish(sum.x, wsum.x, ya);
                    if (yb < pc.h) finish(sum.y, wsum.y, yb);
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < kL16TileH<T, JOINT> / kL16Rows; ++k) {
                    const int ly = tyi + kL16Rows * k;
                    const int y = y0 + ly;
                    if (y >= pc.h) break;
                    const int c0 = (ly + r) * tw + lx + r;
                    const T cx = tr[c0];
                    float wsum = w0;
                    float sum = S::f(ts[c0]) * wsum;
                    for (int yy = 1; yy < radius2; yy += step)
                        for (int xx = 1; xx < radius2; xx += step) taps(cx, c0, yy, xx, gs[yy * radius2 + xx], wsum, sum);
                    finish(sum, wsum, y);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 3 — 16-bit clips, no `ref`, compile-time taps: bilateral_walk16_kernel<CR, CS>.
// The LDS16 kernel above is bound by the LDS pipeline: 16 taps x {tile read, base[i >> 6], delta[i]} = 45 lane
// operations per pixel. Two facts remove two thirds of them:
//   (1) the range weight is symmetric — gr[|c - n|] for the tap c -> n is the one for n -> c — so a pixel looks up
//       only its 8 DOWNWARD taps and receives the 8 upward ones from the pixels above, which computed them as their
//       downward taps yy rows earlier: same f32 bits, and the accumulation order swei * (rw1 + rw2 + rw3 + rw4) is
//       untouched, so the result stays bit-exact;
//   (2) a wave that walks DOWN a strip of columns (lane = column) keeps the 2 CR + 1 rows it needs in registers:
//       neighbour samples and the handed-down weights cross lanes with DPP wave shifts (VALU), not through LDS.
// LDS traffic is the 16 table reads per pixel and nothing else (no tile, no barrier after the table is loaded);
// the 1024-thread workgroup (one per CU: the packed table leaves room for no second one) is 16 independent waves.
// A wave covers 64 columns of which the inner 64 - 2 CR are outputs (the outer ones only feed their neighbours);
// columns and rows outside the plane are the clamped samples of truncatedEdges (:281-289), held by the halo lanes
// and the rows above / below, which makes the handed-down weights of the border pixels come out right by the same
// rule (a virtual pixel's downward tap onto a real one is that pixel's upward tap onto the clamped position).
// Rings of R rows per column set (own column and the four tap columns) and of yy rows per handed-down weight are
// register arrays addressed by template parameters: the row loop is unrolled over one ring period R.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float walk_shr1(float v) {  // lane i takes lane i - 1 (wave_shr:1; lane 0 reads 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float walk_shl1(float v) {  // lane i takes lane i + 1 (wave_shl:1; lane 63 reads 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
// {value of lane i - T0, value of lane i - T1} / {lane i + T0, lane i + T1}: T0 = 1, T1 = 2 or 3, chained single shifts
// The T1-lane hop: T1 chained DPP shifts are T1 half-rate VALU instructions (4 cycles each on gfx950, tools/ubench/valu_rate.hip); one
// ds_bpermute_b32 moves the value any distance through the LDS crossbar, which has room beside the table reads (no bank conflicts, no storage).
// It wraps around the wave where the DPP shift feeds zeros: either way only halo lanes see the difference (see above).
// Measured (tools/bil_ab.py, 1080p): the PLATEAU table form (one LDS read a lookup) gains 7 % (8-bit sigmaS = 2; the filter's defaults on 16-bit),
// the packed forms (two reads a lookup, the LDS pipeline 65 % busy and two thirds of that bank conflicts) lose 2-3 %: the choice follows the form.
#ifdef VSZIP_WALK_NO_BPERM  // (sweeps: the round-3 form everywhere)
template <int FORM>
constexpr bool kWalkBperm = false;
#else
template <int FORM>
constexpr bool kWalkBperm = FORM == 2;
#endif
template <int T1, bool BP>
__device__ __forceinline__ v2f walk_left_pair(float a, float b, int addr_l) {  // a travels T0 = 1 lane, b travels T1 lanes; addr_l = 4 * ((lane - T1) & 63)
    float y;
    if constexpr (BP) {
        y = __int_as_float(__builtin_amdgcn_ds_bpermute(addr_l, __float_as_int(b)));
    } else {
        y = walk_shr1(b);
#pragma unroll
        for (int k = 1; k < T1; ++k) y = walk_shr1(y);
    }
    return v2f{walk_shr1(a), y};
}
template <int T1, bool BP>
__device__ __forceinline__ v2f walk_right_pair(float a, float b, int addr_r) {
    float y;
    if constexpr (BP) {
        y = __int_as_float(__builtin_amdgcn_ds_bpermute(addr_r, __float_as_int(b)));
    } else {
        y = walk_shl1(b);
#pragma unroll
        for (int k = 1; k < T1; ++k) y = walk_shl1(y);
    }
    return v2f{walk_shl1(a), y};
}

#ifndef VSZIP_WALK_BAND
#define VSZIP_WALK_BAND 270
#endif
constexpr int kWalkBand = VSZIP_WALK_BAND;  // output rows per strip: a multiple of both ring periods, and 1080 / 540 / 2160 rows split into whole bands.
// 270: a strip pays its 2 CR rows of ring warm-up once per band — 108 / 180 / 270 / 360 / 540 rows measured 69.4 / 68-69 / 72.3 / 62 / 54.9 k fps at 1080p
// (interleaved A/B on one device; longer bands leave too few strips to balance 4096 waves)

template <int CR, int CS>
struct WalkState {
    static constexpr int T0 = 1, T1 = 1 + CS;      // the two tap distances per axis (CR = 3, CS = 2: 1, 3; CR = 2, CS = 1: 1, 2)
    static constexpr int R = CR == 3 ? 9 : 6;      // ring period: >= 2 CR + 1 rows, a multiple of T0 and T1
    float sc[R];                                   // the lane's own column, as f32 (exact)
    v2f sl[R], sr[R];                              // the tap columns as pairs over the tap distance: {x - T0, x - T1}, {x + T0, x + T1}
    v2f ha0[T0], hb0[T0];                          // handed-down weights of the yy = T0 taps: a = from the pixels at x - xx (rw3), b = from x + xx (rw1)
    v2f ha1[T1], hb1[T1];                          // ... of the yy = T1 taps
    int addr_l, addr_r;                            // ds_bpermute addresses of the lanes T1 to the left / right
};

// The table in LDS. COARSE (0) = base[i >> 6] (u32 x 1024) - delta[i] (u16), any table that packs; FINE (1) = base4[i >> 2] (u32 x 16384) -
// delta8[i] (u8), gentle tables only — its byte offsets are i & ~3 and i: one VALU instruction of address arithmetic per lookup instead of three.
// PLATEAU (2, round 3): the table as it is, up to the entry where the reference stops computing it (`upper` = trunc(sigmaR * 8 * 65535 + 0.5),
// bilateral.zig:316-334: every entry beyond repeats gr[upper]) — the STEEP tables of the filter's usual sigmaR (0.01 ... 0.06; default 0.02),
// which neither packed form holds and which had to be gathered through L2: one LDS read and a v_min per lookup, up to 32 768 entries.
constexpr int kWalkPlateauMax = 32768;
template <int FORM>
struct WalkLut;
template <>
struct WalkLut<0> {
    uint32_t base[1024];
    uint16_t delta[65536];
    __device__ __forceinline__ float at(uint32_t i, uint32_t) const { return __uint_as_float(base[i >> 6] - (uint32_t)delta[i]); }
};
template <>
struct WalkLut<1> {
    uint32_t base[16384];
    uint8_t delta[65536];
    __device__ __forceinline__ float at(uint32_t i, uint32_t) const { return __uint_as_float(base[i >> 2] - (uint32_t)delta[i]); }
};
template <>
struct WalkLut<2> {
    float t[kWalkPlateauMax];
    __device__ __forceinline__ float at(uint32_t i, uint32_t upper) const { return t[min(i, upper)]; }
};
// CUBIC (3, round 3): what lies between — 0.0625 < sigmaR < 0.42: more than 32 768 computed entries, too steep for 16-bit deltas. The table is smooth
// (a Gaussian), so it is restated as a cubic per 128 entries, evaluated with three FMAs exactly as the host evaluates it when it builds the form, plus a
// signed byte per entry with the difference of the bit patterns, bits(table[i]) - bits(cubic(i)): the sum IS the entry (the host checks every entry up to
// `upper` and that every difference fits the byte). 8 + 64 KiB of LDS, a 16-byte and a 1-byte read and five VALU instructions more per lookup than PLATEAU.
constexpr int kCubicSeg = 512, kCubicBytes = kCubicSeg * 16 + 65536;
template <>
struct WalkLut<3> {
    v4f coef[kCubicSeg];
    int8_t corr[65536];
    __device__ __forceinline__ float at(uint32_t i, uint32_t upper) const {
        const uint32_t j = min(i, upper);
        const v4f c = coef[j >> 7];
        const float t = (float)(j & 127u);
        const float a = fmaf(fmaf(fmaf(c.w, t, c.z), t, c.y), t, c.x);
        return __int_as_float(__float_as_int(a) + (int)corr[j]);
    }
};

template <int CR, int CS, int P, int FINE, typename T>
__device__ __forceinline__ void walk_step(WalkState<CR, CS> &st, const WalkLut<FINE> &tab, const uint32_t upper, const v2f (&sw)[2], float w0, float peak,
                                           T &pend, const T *__restrict__ nextp, T *__restrict__ dstp) {
    using W = WalkState<CR, CS>;
    constexpr int R = W::R, T0 = W::T0, T1 = W::T1;
    // 1: the row loaded a step ago enters the rings (slot P), with its shifted copies
    const float v = (float)pend;
    st.sc[P] = v;
    st.sl[P] = walk_left_pair<T1, kWalkBperm<FINE>>(v, v, st.addr_l);
    st.sr[P] = walk_right_pair<T1, kWalkBperm<FINE>>(v, v, st.addr_r);
    // 2: the next row's load is in flight during the arithmetic
    pend = *nextp;
    // 3: the row CR above the newest one. Pairs run over the tap distance xx (v_pk_* : IEEE per element, the reference's order)
    constexpr int C = (P - CR + R) % R;
    const float c = st.sc[C];
    auto lut = [&](float n) {
        if constexpr (!std::is_same<T, float>::value)
            return tab.at((uint32_t)fabsf(c - n), upper);  // |c - n|: exact in f32 for 8- / 16-bit samples
        else
            return tab.at((uint32_t)truncf(fminf(1.0f, fabsf(c - n)) * 65535.0f + 0.5f), upper);  // bilateral.zig:15-22 (|c - n| == |n - c|: the same entry both ways)
    };
    float wsum = w0, sum = c * w0;
    auto sets = [&](const v2f swv, const v2f n1, const v2f n2, const v2f n3, const v2f n4, v2f &ha, v2f &hb) {
        const v2f rw2 = {lut(n2.x), lut(n2.y)}, rw4 = {lut(n4.x), lut(n4.y)};  // the downward taps (+yy, +xx), (+yy, -xx)
        const v2f rw1 = hb, rw3 = ha;  // handed down: (-yy, +xx) is the (+yy, -xx) tap of the pixel at x + xx, (-yy, -xx) the (+yy, +xx) tap of the one at x - xx
        ha = walk_left_pair<T1, kWalkBperm<FINE>>(rw2.x, rw2.y, st.addr_l);   // what the pixels yy rows below take from their left / right neighbours
        hb = walk_right_pair<T1, kWalkBperm<FINE>>(rw4.x, rw4.y, st.addr_r);
        const v2f wi = swv * (rw1 + rw2 + rw3 + rw4);
        const v2f si = swv * (n1 * rw1 + n2 * rw2 + n3 * rw3 + n4 * rw4);
        wsum += wi.x;
        sum += si.x;
        wsum += wi.y;
        sum += si.y;
    };
    {
        constexpr int up = (C - T0 + R) % R, dn = (C + T0) % R, sl = P % T0;
        sets(sw[0], st.sr[up], st.sr[dn], st.sl[up], st.sl[dn], st.ha0[sl], st.hb0[sl]);
    }
    {
        constexpr int up = (C - T1 + R) % R, dn = (C + T1) % R, sl = P % T1;
        sets(sw[1], st.sr[up], st.sr[dn], st.sl[up], st.sl[dn], st.ha1[sl], st.hb1[sl]);
    }
    // No branch anywhere in a step (a period of R steps is ONE basic block: the scheduler may start a step's loads and table
    // reads under the previous step's arithmetic): rows and lanes that produce no output store to a dummy line instead.
    const float q = __fdiv_rn(sum, wsum);
    if constexpr (!std::is_same<T, float>::value)
        *dstp = (T)truncf(fminf(fmaxf(q + 0.5f, 0.0f), peak));  // finalize :30-36
    else
        *dstp = q;
}

template <int CR, int CS, int FINE, typename T, int... P>
__device__ __forceinline__ void walk_period(WalkState<CR, CS> &st, const WalkLut<FINE> &tab, const uint32_t upper, const v2f (&sw)[2], float w0, float peak, T &pend,
                                             const T *__restrict__ colp, int rstride, int h, int ys, int t0, int y0, int y1, bool lane_out, T *__restrict__ dcol,
                                             int dstride, T *__restrict__ dummy, std::integer_sequence<int, P...>) {
    // step t handles the new row ys + t and the output row ys + t - CR; rows are clamped into the plane (replicate padding)
    (walk_step<CR, CS, P, FINE, T>(st, tab, upper, sw, w0, peak, pend, colp + (size_t)min(max(ys + t0 + P + 1, 0), h - 1) * rstride,
                          (lane_out && ys + t0 + P - CR >= y0 && ys + t0 + P - CR < y1) ? dcol + (size_t)(ys + t0 + P - CR) * dstride : dummy),
     ...);
}

// (T last: 8- / 16-bit integer samples — 8-bit clips and 10- / 12-bit ones in 16-bit containers through the PLATEAU form, their whole table —
// or, the 8K RGBS pipeline's Bilateral stage, f32 samples: the same walk, the table index and the finish of the float path)
template <int CR, int CS, int FINE, typename T = uint16_t>
__global__ __launch_bounds__(1024) void bilateral_walk16_kernel(const BLParams prm, const int nstrips, int *__restrict__ next_strip) {
    using W = WalkState<CR, CS>;
    __shared__ __attribute__((aligned(16))) WalkLut<FINE> slut;
    const int tid = (int)threadIdx.x;
    const uint32_t upper = (uint32_t)prm.lut_upper;
    if constexpr (FINE == 2) {
        // the f32 table itself, up to its plateau
        const uint4 *g = reinterpret_cast<const uint4 *>(prm.p[0].gr);
        for (int i = tid; i < ((int)upper + 4) / 4; i += 1024) reinterpret_cast<uint4 *>(&slut)[i] = g[i];
    } else if constexpr (FINE == 3) {
        const uint4 *g = reinterpret_cast<const uint4 *>(prm.p[0].gr + 65536);  // the cubic form lies behind the f32 table
        for (int i = tid; i < kCubicBytes / 16; i += 1024) reinterpret_cast<uint4 *>(&slut)[i] = g[i];
    } else {
        // the packed forms lie behind the f32 table: COARSE (4 + 128 KiB), then FINE (64 + 64 KiB)
        const uint4 *g = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(prm.p[0].gr + 65536) + (FINE ? kL16LutBytes : 0));
        for (int i = tid; i < (int)(sizeof(WalkLut<FINE>) / 16); i += 1024) reinterpret_cast<uint4 *>(&slut)[i] = g[i];
    }
    __syncthreads();  // the only barrier: from here on the 16 waves are independent
    const int lane = tid & 63;
    constexpr int WOUT = 64 - 2 * CR;
    // strips are handed out dynamically (one atomic per strip, lane 0): a plane's strips differ in length (the last band) and
    // 19 584 luma strips do not divide by 4 096 waves; the counter is zeroed by the host before the launch
    auto grab = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(next_strip, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    T *dummy = reinterpret_cast<T *>(next_strip + 64) + lane;  // 256 B behind the counter: what the halo lanes / rows "store"
    int pi = 0;
#pragma unroll 1
    for (int sidx = grab(); sidx < nstrips; sidx = grab()) {
        while (pi + 1 < prm.nplanes && sidx >= prm.p[pi + 1].block0) ++pi;  // a wave's strips come in increasing order
        const BLPlane &pl = prm.p[pi];
        const float *gs = pl.gs;  // (planes of one launch share the range table, radius and step — not necessarily sigmaS)
        const v2f sw[2] = {{gs[W::T0 * (CR + 1) + W::T0], gs[W::T0 * (CR + 1) + W::T1]}, {gs[W::T1 * (CR + 1) + W::T0], gs[W::T1 * (CR + 1) + W::T1]}};
        const float w0 = gs[0] * slut.at(0, upper);
        const int ls = sidx - pl.block0;
        const int X0 = (ls % pl.nbx) * WOUT, y0 = (ls / pl.nbx) * kWalkBand, y1 = min(y0 + kWalkBand, pl.h);
        const int col = X0 - CR + lane;
        const T *colp = static_cast<const T *>(pl.src) + min(max(col, 0), pl.w - 1);
        T *dcol = static_cast<T *>(pl.dst) + min(max(col, 0), pl.w - 1);
        const bool lane_out = lane >= CR && lane < 64 - CR && col < pl.w;
        W st;
        st.addr_l = ((lane - W::T1) & 63) << 2;
        st.addr_r = ((lane + W::T1) & 63) << 2;
#pragma unroll
        for (int b = 0; b < W::R; ++b) {
            st.sc[b] = 0.0f;
            st.sl[b] = st.sr[b] = v2f{0.0f, 0.0f};
        }
#pragma unroll
        for (int b = 0; b < W::T0; ++b) st.ha0[b] = st.hb0[b] = v2f{0.0f, 0.0f};
#pragma unroll
        for (int b = 0; b < W::T1; ++b) st.ha1[b] = st.hb1[b] = v2f{0.0f, 0.0f};
        const int ys = y0 - 2 * CR;  // first row needed: the weights handed to row y0 come from rows y0 - CR .., which tap rows down to y0 - 2 CR
        T pend = colp[(size_t)min(max(ys, 0), pl.h - 1) * pl.sstride];
        const int steps = (y1 - y0) + 3 * CR;
#pragma unroll 1
        for (int t0 = 0; t0 < steps; t0 += W::R)
            walk_period<CR, CS, FINE, T>(st, slut, upper, sw, w0, prm.peak, pend, colp, pl.sstride, pl.h, ys, t0, y0, y1, lane_out, dcol, pl.dstride, dummy, std::make_integer_sequence<int, W::R>{});
    }
}

// ---------------------------------------------------------------------------------------------
// The same walk for the filter's DEFAULT spatial sigma (sigmaS = 3: radius 5, step 2 — tap distances 1, 3, 5, nine (yy, xx) sets, 36
// taps of which a pixel looks up 18): bilateral_walk36_kernel. The third distance rides beside the packed pair as a scalar; rings of 15
// rows (a multiple of all three distances, >= 2 * 5 + 1); 512-thread workgroups — the state is about 200 registers.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ float walk_shr(float v) {
#pragma unroll
    for (int k = 0; k < N; ++k) v = walk_shr1(v);
    return v;
}
template <int N>
__device__ __forceinline__ float walk_shl(float v) {
#pragma unroll
    for (int k = 0; k < N; ++k) v = walk_shl1(v);
    return v;
}
// a hop of N lanes: one ds_bpermute_b32 (see walk_left_pair) or N chained DPP shifts
template <int N, bool BP>
__device__ __forceinline__ float walk_hop_l(float v, int addr) {
    if constexpr (BP && N > 1)
        return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
    else
        return walk_shr<N>(v);
}
template <int N, bool BP>
__device__ __forceinline__ float walk_hop_r(float v, int addr) {
    if constexpr (BP && N > 1)
        return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
    else
        return walk_shl<N>(v);
}
struct Walk3State {
    static constexpr int CR = 5, T0 = 1, T1 = 3, T2 = 5, R = 15;
    int al3, ar3, al5, ar5;  // ds_bpermute addresses of the lanes 3 / 5 to the left / right
    float sc[R];
    v2f sl[R], sr[R];   // {x -+ 1, x -+ 3}
    float slz[R], srz[R];  // x -+ 5
    v2f ha0[T0], hb0[T0], ha1[T1], hb1[T1], ha2[T2], hb2[T2];
    float haz0[T0], hbz0[T0], haz1[T1], hbz1[T1], haz2[T2], hbz2[T2];
};

template <int P, int FORM, typename T>
__device__ __forceinline__ void walk3_step(Walk3State &st, const WalkLut<FORM> &tab, const uint32_t upper, const v2f (&sw)[3], const float (&swz)[3], float w0, float peak,
                                            T &pend, const T *__restrict__ nextp, T *__restrict__ dstp) {
    using W = Walk3State;
    constexpr int R = W::R, CR = W::CR, T0 = W::T0, T1 = W::T1, T2 = W::T2;
    const float v = (float)pend;
    st.sc[P] = v;
    {
        const float l1 = walk_shr1(v), r1 = walk_shl1(v);
        float l3, r3, l5, r5;
        constexpr bool BP = kWalkBperm<FORM>;
        if constexpr (BP) {
            l3 = walk_hop_l<T1, BP>(v, st.al3), r3 = walk_hop_r<T1, BP>(v, st.ar3), l5 = walk_hop_l<T2, BP>(v, st.al5), r5 = walk_hop_r<T2, BP>(v, st.ar5);
        } else {
            l3 = walk_shr<2>(l1), r3 = walk_shl<2>(r1), l5 = walk_shr<2>(l3), r5 = walk_shl<2>(r3);
        }
        st.sl[P] = v2f{l1, l3};
        st.sr[P] = v2f{r1, r3};
        st.slz[P] = l5;
        st.srz[P] = r5;
    }
    pend = *nextp;
    constexpr int C = (P - CR + R) % R;
    const float c = st.sc[C];
    auto lut = [&](float n) {
        if constexpr (!std::is_same<T, float>::value)
            return tab.at((uint32_t)fabsf(c - n), upper);
        else
            return tab.at((uint32_t)truncf(fminf(1.0f, fabsf(c - n)) * 65535.0f + 0.5f), upper);
    };
    float wsum = w0, sum = c * w0;
    // one yy: the three xx sets in the reference's order (1, 3, 5); handed-down weights as in walk_step
    auto sets = [&](const v2f swv, const float swq, const v2f n1, const v2f n2, const v2f n3, const v2f n4, const float n1z, const float n2z, const float n3z, const float n4z, v2f &ha,
                    v2f &hb, float &haz, float &hbz) {
        const v2f rw2 = {lut(n2.x), lut(n2.y)}, rw4 = {lut(n4.x), lut(n4.y)};
        const float rw2z = lut(n2z), rw4z = lut(n4z);
        const v2f rw1 = hb, rw3 = ha;
        const float rw1z = hbz, rw3z = haz;
        ha = v2f{walk_shr1(rw2.x), walk_hop_l<T1, kWalkBperm<FORM>>(rw2.y, st.al3)};
        hb = v2f{walk_shl1(rw4.x), walk_hop_r<T1, kWalkBperm<FORM>>(rw4.y, st.ar3)};
        haz = walk_hop_l<T2, kWalkBperm<FORM>>(rw2z, st.al5);
        hbz = walk_hop_r<T2, kWalkBperm<FORM>>(rw4z, st.ar5);
        const v2f wi = swv * (rw1 + rw2 + rw3 + rw4);
        const v2f si = swv * (n1 * rw1 + n2 * rw2 + n3 * rw3 + n4 * rw4);
        const float wiz = swq * (rw1z + rw2z + rw3z + rw4z);
        const float siz = swq * (n1z * rw1z + n2z * rw2z + n3z * rw3z + n4z * rw4z);
        wsum += wi.x;
        sum += si.x;
        wsum += wi.y;
        sum += si.y;
        wsum += wiz;
        sum += siz;
    };
    {
        constexpr int up = (C - T0 + R) % R, dn = (C + T0) % R, sl = P % T0;
        sets(sw[0], swz[0], st.sr[up], st.sr[dn], st.sl[up], st.sl[dn], st.srz[up], st.srz[dn], st.slz[up], st.slz[dn], st.ha0[sl], st.hb0[sl], st.haz0[sl], st.hbz0[sl]);
    }
    {
        constexpr int up = (C - T1 + R) % R, dn = (C + T1) % R, sl = P % T1;
        sets(sw[1], swz[1], st.sr[up], st.sr[dn], st.sl[up], st.sl[dn], st.srz[up], st.srz[dn], st.slz[up], st.slz[dn], st.ha1[sl], st.hb1[sl], st.haz1[sl], st.hbz1[sl]);
    }
    {
        constexpr int up = (C - T2 + R) % R, dn = (C + T2) % R, sl = P % T2;
        sets(sw[2], swz[2], st.sr[up], st.sr[dn], st.sl[up], st.sl[dn], st.srz[up], st.srz[dn], st.slz[up], st.slz[dn], st.ha2[sl], st.hb2[sl], st.haz2[sl], st.hbz2[sl]);
    }
    const float q = __fdiv_rn(sum, wsum);
    if constexpr (!std::is_same<T, float>::value)
        *dstp = (T)truncf(fminf(fmaxf(q + 0.5f, 0.0f), peak));
    else
        *dstp = q;
}

template <int FORM, typename T, int... P>
__device__ __forceinline__ void walk3_period(Walk3State &st, const WalkLut<FORM> &tab, const uint32_t upper, const v2f (&sw)[3], const float (&swz)[3], float w0, float peak, T &pend,
                                              const T *__restrict__ colp, int rstride, int h, int ys, int t0, int y0, int y1, bool lane_out, T *__restrict__ dcol, int dstride,
                                              T *__restrict__ dummy, std::integer_sequence<int, P...>) {
    constexpr int CR = Walk3State::CR;
    (walk3_step<P, FORM, T>(st, tab, upper, sw, swz, w0, peak, pend, colp + (size_t)min(max(ys + t0 + P + 1, 0), h - 1) * rstride,
                            (lane_out && ys + t0 + P - CR >= y0 && ys + t0 + P - CR < y1) ? dcol + (size_t)(ys + t0 + P - CR) * dstride : dummy),
     ...);
}

template <int FORM, typename T>
__global__ __launch_bounds__(512) void bilateral_walk36_kernel(const BLParams prm, const int nstrips, int *__restrict__ next_strip) {
    using W = Walk3State;
    constexpr int CR = W::CR;
    __shared__ __attribute__((aligned(16))) WalkLut<FORM> slut;
    const int tid = (int)threadIdx.x;
    const uint32_t upper = (uint32_t)prm.lut_upper;
    if constexpr (FORM == 2) {
        const uint4 *g = reinterpret_cast<const uint4 *>(prm.p[0].gr);
        for (int i = tid; i < ((int)upper + 4) / 4; i += 512) reinterpret_cast<uint4 *>(&slut)[i] = g[i];
    } else if constexpr (FORM == 3) {
        const uint4 *g = reinterpret_cast<const uint4 *>(prm.p[0].gr + 65536);
        for (int i = tid; i < kCubicBytes / 16; i += 512) reinterpret_cast<uint4 *>(&slut)[i] = g[i];
    } else {
        const uint4 *g = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(prm.p[0].gr + 65536) + (FORM ? kL16LutBytes : 0));
        for (int i = tid; i < (int)(sizeof(WalkLut<FORM>) / 16); i += 512) reinterpret_cast<uint4 *>(&slut)[i] = g[i];
    }
    __syncthreads();
    const int lane = tid & 63;
    constexpr int WOUT = 64 - 2 * CR;
    auto grab = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(next_strip, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    T *dummy = reinterpret_cast<T *>(next_strip + 64) + lane;
    int pi = 0;
#pragma unroll 1
    for (int sidx = grab(); sidx < nstrips; sidx = grab()) {
        while (pi + 1 < prm.nplanes && sidx >= prm.p[pi + 1].block0) ++pi;
        const BLPlane &pl = prm.p[pi];
        const float *gs = pl.gs;
        constexpr int G = CR + 1;
        const v2f sw[3] = {{gs[1 * G + 1], gs[1 * G + 3]}, {gs[3 * G + 1], gs[3 * G + 3]}, {gs[5 * G + 1], gs[5 * G + 3]}};
        const float swz[3] = {gs[1 * G + 5], gs[3 * G + 5], gs[5 * G + 5]};
        const float w0 = gs[0] * slut.at(0, upper);
        const int ls = sidx - pl.block0;
        const int X0 = (ls % pl.nbx) * WOUT, y0 = (ls / pl.nbx) * kWalkBand, y1 = min(y0 + kWalkBand, pl.h);
        const int col = X0 - CR + lane;
        const T *colp = static_cast<const T *>(pl.src) + min(max(col, 0), pl.w - 1);
        T *dcol = static_cast<T *>(pl.dst) + min(max(col, 0), pl.w - 1);
        const bool lane_out = lane >= CR && lane < 64 - CR && col < pl.w;
        W st;
        st.al3 = ((lane - 3) & 63) << 2, st.ar3 = ((lane + 3) & 63) << 2, st.al5 = ((lane - 5) & 63) << 2, st.ar5 = ((lane + 5) & 63) << 2;
#pragma unroll
        for (int b = 0; b < W::R; ++b) {
            st.sc[b] = 0.0f;
            st.sl[b] = st.sr[b] = v2f{0.0f, 0.0f};
            st.slz[b] = st.srz[b] = 0.0f;
        }
#pragma unroll
        for (int b = 0; b < W::T0; ++b) {
            st.ha0[b] = st.hb0[b] = v2f{0.0f, 0.0f};
            st.haz0[b] = st.hbz0[b] = 0.0f;
        }
#pragma unroll
        for (int b = 0; b < W::T1; ++b) {
            st.ha1[b] = st.hb1[b] = v2f{0.0f, 0.0f};
            st.haz1[b] = st.hbz1[b] = 0.0f;
        }
#pragma unroll
        for (int b = 0; b < W::T2; ++b) {
            st.ha2[b] = st.hb2[b] = v2f{0.0f, 0.0f};
            st.haz2[b] = st.hbz2[b] = 0.0f;
        }
        const int ys = y0 - 2 * CR;
        T pend = colp[(size_t)min(max(ys, 0), pl.h - 1) * pl.sstride];
        const int steps = (y1 - y0) + 3 * CR;
#pragma unroll 1
        for (int t0 = 0; t0 < steps; t0 += W::R)
            walk3_period<FORM, T>(st, slut, upper, sw, swz, w0, prm.peak, pend, colp, pl.sstride, pl.h, ys, t0, y0, y1, lane_out, dcol, pl.dstride, dummy, std::make_integer_sequence<int, W::R>{});
    }
}

template <typename T>
__global__ __launch_bounds__(kBX *kBY) void bilateral_truncated_kernel(const BLParams prm) {
    using S = BSmp<T>;
    int pi = 0;
    const int b = blockIdx.x;
#pragma unroll 1
    for (int i = 1; i < prm.nplanes; ++i)
        if (b >= prm.p[i].block0) pi = i;
    const BLPlane pl = prm.p[pi];
    const int lb = b - pl.block0;
    const int x = (lb % pl.nbx) * kBX + (int)threadIdx.x;
    const int y = (lb / pl.nbx) * kBY + (int)threadIdx.y;
    if (x >= pl.w || y >= pl.h) return;
    const T *src = static_cast<const T *>(pl.src);
    const T *ref = static_cast<const T *>(pl.ref);
    T *dst = static_cast<T *>(pl.dst);
    const float *gs = pl.gs, *gr = pl.gr;
    const int radius2 = pl.radius + 1, step = pl.step;
    const int w1 = pl.w - 1, h1 = pl.h - 1;

    const T cx = ref[(size_t)y * pl.rstride + x];
    float wsum = gs[0] * gr[0];
    float sum = S::f(src[(size_t)y * pl.sstride + x]) * wsum;
    for (int yy = 1; yy < radius2; yy += step) {
        // replicate padding (truncatedEdges :281-289); the identity away from the border
        const int ya = max(y - yy, 0), yb = min(y + yy, h1);
        const T *la = src + (size_t)ya * pl.sstride, *lb_ = src + (size_t)yb * pl.sstride;
        const T *lar = ref + (size_t)ya * pl.rstride, *lbr = ref + (size_t)yb * pl.rstride;
        for (int xx = 1; xx < radius2; xx += step) {
            const int xa = min(x + xx, w1), xb = max(x - xx, 0);
            const float swei = gs[yy * radius2 + xx];
            const float rw1 = gr[S::ridx(cx, lar[xa])];
            const float rw2 = gr[S::ridx(cx, lbr[xa])];
            const float rw3 = gr[S::ridx(cx, lar[xb])];
            const float rw4 = gr[S::ridx(cx, lbr[xb])];
            wsum += swei * (rw1 + rw2 + rw3 + rw4);
            sum += swei * (S::f(la[xa]) * rw1 + S::f(lb_[xa]) * rw2 + S::f(la[xb]) * rw3 + S::f(lb_[xb]) * rw4);
        }
    }
    const float q = __fdiv_rn(sum, wsum);
    if constexpr (S::is_int) {
        const float v = fminf(fmaxf(q + 0.5f, 0.0f), prm.peak);  // finalize :30-36
        dst[(size_t)y * pl.dstride + x] = (T)truncf(v);
    } else {
        dst[(size_t)y * pl.dstride + x] = (T)q;
    }
}


// ---------------------------------------------------------------------------
// Algorithm 1: PBFIC (Yang's O(1) bilateral), bilateral.zig:91-171 + the recursive
// Gaussian of :336-431. All PBFICnum range layers of a plane are processed at once —
// layer k has its own W_k / J_k f32 planes (dense, stride == width) in the context
// scratch; 288 GB of HBM make that affordable (4K luma, 32 layers: 2.1 GB) and it is
// what gives the IIR passes their parallelism: the recursion is sequential along a
// line (kept exactly in the reference's operation order), so a pass has only
// lines x planes independent chains.
//   pbfic_wj_kernel   W_k = gr[|pk - ref|], J_k = W_k * src                       (:118-131)
//   pbfic_rg_h_kernel causal + anticausal 3rd-order IIR along x, one wave per 64 rows,
//                     64x64 tiles staged through LDS so HBM sees coalesced rows       (:411-431)
//   pbfic_rg_v_kernel the same along y, one thread per column                       (:366-409)
//   pbfic_out_kernel  layer = J/W, linear interpolation between the two layers that
//                     bracket the pixel's ref value, finalize                       (:133-171)
// ---------------------------------------------------------------------------
// Round 4: a BATCH of planes per set of launches. One plane alone gives the IIR passes lines x 2 PBFICnum chains (1080p luma, four layers: 136 waves of the
// horizontal pass, 64 blocks of the vertical one) and the per-plane entry ran four launches and a stream synchronise for each plane of a call: 1.2 k 1080p frames/s
// at sigmaS = 7 next to 24 k at sigmaS = 5 (tools/bil_sigma_sweep.py). The planes of a call — as many as the scratch budget holds — now share the four launches;
// a block finds its plane by binary search over the per-kernel block offsets, the layer values travel in the kernel arguments (no copy, no synchronise).
constexpr int kMaxPB = 64;                       // planes per batch
constexpr size_t kPBScratchBudget = 6ull << 30;  // W / J layer planes of a batch (a 1080p YUV frame with four layers: 100 MB)
struct PBPlane {
    const void *src, *ref;
    void *dst;
    const float *gr;
    size_t wj_off;  // floats from PBBatch::wj: [2 * num] planes of w*h floats: W_0, J_0, W_1, J_1, ...
    int sstride, rstride, dstride, w, h;
    int num, pkt;   // layers; which of the batch's layer-value tables
    float b, b1, b2, b3;
    int b0[4];      // first block of this plane in the wj / rg_h / rg_v / out launch
};
struct PBBatch {
    PBPlane p[kMaxPB];
    uint32_t pk[3][256];  // layer values, the bits of T (planes of one call have at most three configurations)
    float *wj;
    float peak;
    int n;
};

template <int KRN>
__device__ __forceinline__ int pb_find(const PBBatch &a, int b) {  // the last plane whose first block (of kernel KRN) is not beyond b
    int pi = 0;
    for (int lo = 1, hi = a.n - 1; lo <= hi;) {
        const int mid = (lo + hi) >> 1;
        if (b >= a.p[mid].b0[KRN]) {
            pi = mid;
            lo = mid + 1;
        } else {
            hi = mid - 1;
        }
    }
    return pi;
}
template <typename T>
__device__ __forceinline__ T pb_layer(const PBBatch &a, const PBPlane &pl, int k) {
    const uint32_t bits = a.pk[pl.pkt][k];
    T v;
    __builtin_memcpy(&v, &bits, sizeof(T));
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void pbfic_wj_kernel(const PBBatch a) {
    using S = BSmp<T>;
    const PBPlane &pl = a.p[pb_find<0>(a, (int)blockIdx.x)];
    const int lb = (int)blockIdx.x - pl.b0[0], nbx = (pl.w + 255) / 256;
    const int x = (lb % nbx) * 256 + threadIdx.x, y = (lb / nbx) % pl.h, k = lb / (nbx * pl.h);
    if (x >= pl.w) return;
    const T rv = static_cast<const T *>(pl.ref)[(size_t)y * pl.rstride + x];
    const T sv = static_cast<const T *>(pl.src)[(size_t)y * pl.sstride + x];
    const T pk = pb_layer<T>(a, pl, k);
    const float wv = pl.gr[S::ridx(pk, rv)];
    const size_t plane = (size_t)pl.w * pl.h, i = (size_t)y * pl.w + x;
    float *wj = a.wj + pl.wj_off;
    wj[(size_t)(2 * k) * plane + i] = wv;
    wj[(size_t)(2 * k + 1) * plane + i] = wv * S::f(sv);
}

// One wave per 64 rows (lane = row in the recursion, lane = column while a 64 x 64 tile moves between
// HBM and LDS, so HBM sees coalesced rows). The recursion along a row is inherently sequential; what
// the kernel hides is everything else: the next tile is fetched into registers while the current one
// is filtered, so the chain never waits for a global round trip, and a tile's LDS reads are issued
// eight steps ahead of the adds.
// DIR +1: causal pass, left to right (:413-424), the first sample passes through;
// DIR -1: anticausal pass, right to left (:425-431), the last sample passes through.
template <int DIR>
__device__ __forceinline__ void pbfic_h_pass(float (*tile)[65], float *__restrict__ io, int y0, int rows, int w, int lane, float b, float b1, float b2, float b3) {
    const int nchunk = (w + 63) / 64;
    const int c_first = DIR > 0 ? 0 : nchunk - 1, c_last = DIR > 0 ? nchunk - 1 : 0;
    float p1 = 0, p2 = 0, p3 = 0;
    float nxt[64];
    {
        const int x0 = c_first * 64, cw = min(64, w - x0);
#pragma unroll
        for (int r = 0; r < 64; ++r) nxt[r] = lane < cw ? io[(size_t)(y0 + min(r, rows - 1)) * w + x0 + lane] : 0.0f;
    }
    for (int c = c_first;; c += DIR) {
#pragma unroll
        for (int r = 0; r < 64; ++r) tile[r][lane] = nxt[r];
        __syncthreads();
        if (c != c_last) {  // the next tile: in flight while this one is filtered (rows past the plane repeat the last one)
            const int x0 = (c + DIR) * 64, cw = min(64, w - x0);
#pragma unroll
            for (int r = 0; r < 64; ++r) nxt[r] = lane < cw ? io[(size_t)(y0 + min(r, rows - 1)) * w + x0 + lane] : 0.0f;
        }
        const int x0 = c * 64, cw = min(64, w - x0);
        if (lane < rows) {
            float *row = &tile[lane][0];
            int i = DIR > 0 ? 0 : cw - 1;
            if (c == c_first) {  // the pass's first sample
                p1 = p2 = p3 = row[i];
                i += DIR;
            }
            const int n = DIR > 0 ? cw - i : i + 1;
            int k = 0;
            for (; k + 8 <= n; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = row[i + DIR * (k + u)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float o = b * v[u] + b1 * p1 + b2 * p2 + b3 * p3;  // :417-423 / :427-430, the reference's operation order
                    p3 = p2;
                    p2 = p1;
                    p1 = o;
                    row[i + DIR * (k + u)] = o;
                }
            }
            for (; k < n; ++k) {
                const float o = b * row[i + DIR * k] + b1 * p1 + b2 * p2 + b3 * p3;
                p3 = p2;
                p2 = p1;
                p1 = o;
                row[i + DIR * k] = o;
            }
        }
        __syncthreads();
        if (lane < cw) {
#pragma unroll 8
            for (int r = 0; r < rows; ++r) io[(size_t)(y0 + r) * w + x0 + lane] = tile[r][lane];
        }
        __syncthreads();
        if (c == c_last) break;
    }
}

__global__ __launch_bounds__(64) void pbfic_rg_h_kernel(const PBBatch a) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    const PBPlane &pl = a.p[pb_find<1>(a, (int)blockIdx.x)];
    const int lb = (int)blockIdx.x - pl.b0[1], nby = (pl.h + 63) / 64;
    const int y0 = (lb % nby) * 64;
    float *io = a.wj + pl.wj_off + (size_t)(lb / nby) * (size_t)pl.w * pl.h;
    const int rows = min(64, pl.h - y0);
    pbfic_h_pass<1>(tile, io, y0, rows, pl.w, lane, pl.b, pl.b1, pl.b2, pl.b3);
    pbfic_h_pass<-1>(tile, io, y0, rows, pl.w, lane, pl.b, pl.b1, pl.b2, pl.b3);
}

__global__ __launch_bounds__(256) void pbfic_rg_v_kernel(const PBBatch a) {
    const PBPlane &pl = a.p[pb_find<2>(a, (int)blockIdx.x)];
    const int lb = (int)blockIdx.x - pl.b0[2], nbx = (pl.w + 255) / 256;
    const int x = (lb % nbx) * 256 + threadIdx.x;
    if (x >= pl.w) return;
    const size_t plane = (size_t)pl.w * pl.h;
    float *io = a.wj + pl.wj_off + (size_t)(lb / nbx) * plane + x;
    const int w = pl.w, h = pl.h;
    const float b = pl.b, b1 = pl.b1, b2 = pl.b2, b3 = pl.b3;
    // :368-387 — rows 0..2 reuse the nearest already-filtered row for the missing taps
    float p1, p2, p3;
    {
        const float v = io[0];
        const float o = b * v + b1 * v + b2 * v + b3 * v;
        io[0] = o;
        p1 = p2 = p3 = o;
    }
    constexpr int U = 8;
    for (int j = 1; j < h; j += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (j + u < h) ? io[(size_t)(j + u) * w] : 0.0f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j + u < h) {
                const float o = b * v[u] + b1 * p1 + b2 * p2 + b3 * p3;
                p3 = p2;
                p2 = p1;
                p1 = o;
                io[(size_t)(j + u) * w] = o;
            }
        }
    }
    // :388-408 — bottom to top
    {
        const float v = io[(size_t)(h - 1) * w];
        const float o = b * v + b1 * v + b2 * v + b3 * v;
        io[(size_t)(h - 1) * w] = o;
        p1 = p2 = p3 = o;
    }
    for (int j = h - 2; j >= 0; j -= U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (j - u >= 0) ? io[(size_t)(j - u) * w] : 0.0f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j - u >= 0) {
                const float o = b * v[u] + b1 * p1 + b2 * p2 + b3 * p3;
                p3 = p2;
                p2 = p1;
                p1 = o;
                io[(size_t)(j - u) * w] = o;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pbfic_out_kernel(const PBBatch a) {
    using S = BSmp<T>;
    const PBPlane &pl = a.p[pb_find<3>(a, (int)blockIdx.x)];
    const int lb = (int)blockIdx.x - pl.b0[3], nbx = (pl.w + 255) / 256;
    const int x = (lb % nbx) * 256 + threadIdx.x, y = lb / nbx;
    if (x >= pl.w) return;
    const float rf = S::f(static_cast<const T *>(pl.ref)[(size_t)y * pl.rstride + x]);
    int k = 0;
    for (; k < pl.num - 2; ++k)
        if (rf < S::f(pb_layer<T>(a, pl, k + 1)) && rf >= S::f(pb_layer<T>(a, pl, k))) break;
    const float p0f = S::f(pb_layer<T>(a, pl, k)), p1f = S::f(pb_layer<T>(a, pl, k + 1));
    const size_t plane = (size_t)pl.w * pl.h, i = (size_t)y * pl.w + x;
    const float *wj = a.wj + pl.wj_off;
    const float w0 = wj[(size_t)(2 * k) * plane + i], j0 = wj[(size_t)(2 * k + 1) * plane + i];
    const float w1 = wj[(size_t)(2 * k + 2) * plane + i], j1 = wj[(size_t)(2 * k + 3) * plane + i];
    const float lo = (w0 == 0.0f) ? 0.0f : __fdiv_rn(j0, w0);
    const float hi = (w1 == 0.0f) ? 0.0f : __fdiv_rn(j1, w1);
    const float vf = __fdiv_rn((p1f - rf) * lo + (rf - p0f) * hi, p1f - p0f);
    T *dst = static_cast<T *>(pl.dst);
    if constexpr (S::is_int) {
        const float v = fminf(fmaxf(__fdiv_rn(vf, 1.0f) + 0.5f, 0.0f), a.peak);
        dst[(size_t)y * pl.dstride + x] = (T)truncf(v);
    } else {
        dst[(size_t)y * pl.dstride + x] = (T)vf;
    }
}

// bilateral.zig:350-364
void rg_params(double sigma, float *b, float *b1, float *b2, float *b3) {
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

// `count` consecutive algorithm-1 planes of a call, in batches the scratch budget holds
template <typename T>
int run_pbfic(vszip_ctx *ctx, const vszip_plane *planes, const vszip_bilateral_cfg *const *cfgs, int count, float peak) {
    for (int done = 0; done < count;) {
        PBBatch a;
        a.peak = peak;
        int n = 0, ntab = 0;
        const vszip_bilateral_cfg *tabs[3] = {nullptr, nullptr, nullptr};
        size_t floats = 0;
        long blocks[4] = {0, 0, 0, 0};
        for (; done + n < count && n < kMaxPB; ++n) {
            const vszip_plane &s = planes[done + n];
            const vszip_bilateral_cfg &c = *cfgs[done + n];
            const int num = c.pbficnum;
            if (!s.src || !s.dst || s.w <= 0 || s.h <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "Bilateral: bad plane");
            if (!c.gr_lut) return vszip_set_error(ctx, VSZIP_ERR_ARG, "Bilateral: LUTs missing (vszip_bilateral_luts)");
            if (num < 2 || num > 256) return vszip_set_error(ctx, VSZIP_ERR_ARG, "Bilateral: PBFICnum %d out of range", num);
            const size_t plane = (size_t)s.w * s.h, need = (size_t)2 * num * plane;
            if (n > 0 && (floats + need) * sizeof(float) > kPBScratchBudget) break;
            int t = 0;
            while (t < ntab && tabs[t]->pbficnum != num) ++t;
            if (t == ntab) {
                if (ntab == 3) break;  // (a fourth layer count: the next batch)
                tabs[ntab++] = &c;
                // layer values, in T exactly as :96-116 (integer: trunc(peak*k/(num-1) + .5); float: k/(num-1) in T)
                for (int k = 0; k < num; ++k) {
                    T v;
                    if constexpr (BSmp<T>::is_int) {
                        const float f = peak * (float)k / ((float)num - 1) + 0.5f;
                        v = (T)f;
                    } else {
                        v = (T)((T)(float)k / (T)(float)(num - 1));
                    }
                    uint32_t bits = 0;
                    std::memcpy(&bits, &v, sizeof(T));
                    a.pk[t][k] = bits;
                }
            }
            PBPlane &d = a.p[n];
            d.src = s.src;
            d.ref = s.ref ? s.ref : s.src;
            d.dst = s.dst;
            d.gr = c.gr_lut;
            d.wj_off = floats;
            d.sstride = (int)s.src_stride;
            d.rstride = s.ref ? (int)s.ref_stride : (int)s.src_stride;
            d.dstride = (int)s.dst_stride;
            d.w = s.w;
            d.h = s.h;
            d.num = num;
            d.pkt = t;
            rg_params(c.sigmaS, &d.b, &d.b1, &d.b2, &d.b3);
            const long nbx = (s.w + 255) / 256, nby = (s.h + 63) / 64;
            const long nb[4] = {nbx * s.h * num, nby * 2 * num, nbx * 2 * num, nbx * s.h};
            for (int q = 0; q < 4; ++q) {
                d.b0[q] = (int)blocks[q];
                blocks[q] += nb[q];
            }
            if (blocks[0] > 0x7fffffffL) return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "Bilateral: algorithm 1 batch too large");
            floats += need;
        }
        a.n = n;
        const int rc = vszip_ensure_scratch(ctx, floats * sizeof(float));
        if (rc != VSZIP_OK) return rc;
        a.wj = static_cast<float *>(ctx->scratch);
        hipLaunchKernelGGL((pbfic_wj_kernel<T>), dim3((unsigned)blocks[0]), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(pbfic_rg_h_kernel, dim3((unsigned)blocks[1]), dim3(64), 0, ctx->stream, a);
        hipLaunchKernelGGL(pbfic_rg_v_kernel, dim3((unsigned)blocks[2]), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL((pbfic_out_kernel<T>), dim3((unsigned)blocks[3]), dim3(256), 0, ctx->stream, a);
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += n;
    }
    return VSZIP_OK;
}

// Range LUTs whose packed form (see bilateral_lds16_kernel) was verified exact and uploaded behind the f32 table
// in the same device allocation: keyed by the gr_lut pointer, dropped when the pointer is freed.
struct PackedLuts {
    std::mutex mu;
    std::map<const void *, uint64_t> exact;  // pointer -> content key (the bits of sigmaR: length and peak are fixed for these tables)
    std::set<const void *> fine;             // ... whose 4-entry-block form (bilateral_walk16_kernel, FINE) is exact as well and stored behind the first
    std::map<const void *, std::pair<uint64_t, uint32_t>> plateau;  // 65536-entry tables that are constant from `upper` <= kWalkPlateauMax - 1 on: pointer -> (content key, upper)
    std::map<const void *, std::pair<uint64_t, uint32_t>> cubic;    // tables stored in the CUBIC form behind the f32 table: pointer -> (content key, upper)
};
PackedLuts &packed_luts() {
    static PackedLuts *p = new PackedLuts();
    return *p;
}
bool lut_is_packed(const void *gr, uint64_t *key = nullptr, bool *fine = nullptr) {
    PackedLuts &p = packed_luts();
    std::lock_guard<std::mutex> lk(p.mu);
    const auto it = p.exact.find(gr);
    if (it == p.exact.end()) return false;
    if (key) *key = it->second;
    if (fine) *fine = p.fine.count(gr) != 0;
    return true;
}

// the LDS form of a table no packed form holds: 2 = PLATEAU, 3 = CUBIC, 0 = none
int lut_plateau(const void *gr, uint64_t *key, uint32_t *upper) {
    PackedLuts &p = packed_luts();
    std::lock_guard<std::mutex> lk(p.mu);
    auto it = p.plateau.find(gr);
    int form = 2;
    if (it == p.plateau.end()) {
        it = p.cubic.find(gr);
        form = 3;
        if (it == p.cubic.end()) return 0;
    }
    *key = it->second.first;
    *upper = it->second.second;
    return form;
}

template <typename T, bool JOINT, int CR, int CS, bool PLAT = false>
int launch_lds16_k(vszip_ctx *ctx, const BLParams &prm, int blocks, size_t lds) {
    const dim3 grid(std::min(blocks, 256)), block(kBX, kL16Rows);
    VSZIP_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(bilateral_lds16_kernel<T, JOINT, CR, CS, PLAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((bilateral_lds16_kernel<T, JOINT, CR, CS, PLAT>), grid, block, lds, ctx->stream, prm, blocks);
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    return VSZIP_OK;
}

template <typename T>
int launch_lds16(vszip_ctx *ctx, BLParams prm, int blocks, bool joint, int max_radius, bool plateau = false) {
    const int tile_elems = (kBX + 2 * max_radius) * ((joint ? kL16TileH<T, true> : kL16TileH<T, false>) + 2 * max_radius);
    prm.lut_offset = tile_elems;  // (the field is free in this kernel: samples of the largest tile, per clip)
    const size_t lds = (size_t)2 * tile_elems * sizeof(T) * (joint ? 2 : 1);  // dynamic part; the LUT is static
    if (plateau) {  // (prm.lut_upper set by the caller) run-time taps only: the walk kernel has the compile-time shapes
        vszip_probe_scope probe(ctx);
        return joint ? launch_lds16_k<T, true, 0, 0, true>(ctx, prm, blocks, lds) : launch_lds16_k<T, false, 0, 0, true>(ctx, prm, blocks, lds);
    }
    int r = prm.p[0].radius, st = prm.p[0].step;
    for (int i = 1; i < prm.nplanes; ++i)
        if (prm.p[i].radius != r || prm.p[i].step != st) r = st = 0;
    vszip_probe_scope probe(ctx);
    if (!joint && r == 3 && st == 2) return launch_lds16_k<T, false, 3, 2>(ctx, prm, blocks, lds);  // sigmaS = 2 luma (BASELINE)
    if (!joint && r == 2 && st == 1) return launch_lds16_k<T, false, 2, 1>(ctx, prm, blocks, lds);  // sigmaS = 1: its 4:2:0 chroma
    if (!joint && r == 3 && st == 1) return launch_lds16_k<T, false, 3, 1>(ctx, prm, blocks, lds);  // sigmaS = 1.5
    if (joint && r == 3 && st == 2) return launch_lds16_k<T, true, 3, 2>(ctx, prm, blocks, lds);
    if (joint && r == 2 && st == 1) return launch_lds16_k<T, true, 2, 1>(ctx, prm, blocks, lds);
    return joint ? launch_lds16_k<T, true, 0, 0>(ctx, prm, blocks, lds) : launch_lds16_k<T, false, 0, 0>(ctx, prm, blocks, lds);
}

// The column-walking kernel: strips of 64 - 2 CR output columns x kWalkBand rows; prm.p[i].block0 / nbx are re-based on strips.
template <int CR, int CS, typename T = uint16_t>
int launch_walk16(vszip_ctx *ctx, BLParams prm, int form) {  // form: 0 COARSE, 1 FINE, 2 PLATEAU (prm.lut_upper)
    int strips = 0;
    for (int i = 0; i < prm.nplanes; ++i) {
        BLPlane &p = prm.p[i];
        p.block0 = strips;
        p.nbx = (p.w + (64 - 2 * CR) - 1) / (64 - 2 * CR);
        strips += p.nbx * ((p.h + kWalkBand - 1) / kWalkBand);
    }
    const int grid = std::min((strips + 15) / 16, 256);
    int rcs = vszip_ensure_scratch(ctx, 512);  // the strip counter, and the dummy store line 256 B behind it
    if (rcs != VSZIP_OK) return rcs;
    int *counter = static_cast<int *>(ctx->scratch);
    VSZIP_HIP_CHECK(ctx, hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));
    vszip_probe_scope probe(ctx);
    if (form == 3)
        hipLaunchKernelGGL((bilateral_walk16_kernel<CR, CS, 3, T>), dim3(grid), dim3(1024), 0, ctx->stream, prm, strips, counter);
    else if (form == 2)
        hipLaunchKernelGGL((bilateral_walk16_kernel<CR, CS, 2, T>), dim3(grid), dim3(1024), 0, ctx->stream, prm, strips, counter);
    else if (form == 1)
        hipLaunchKernelGGL((bilateral_walk16_kernel<CR, CS, 1, T>), dim3(grid), dim3(1024), 0, ctx->stream, prm, strips, counter);
    else
        hipLaunchKernelGGL((bilateral_walk16_kernel<CR, CS, 0, T>), dim3(grid), dim3(1024), 0, ctx->stream, prm, strips, counter);
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    return VSZIP_OK;
}

// ... and the three-distance walk (radius 5, step 2): 54 output columns per wave, 8 waves per workgroup
template <typename T>
int launch_walk36(vszip_ctx *ctx, BLParams prm, int form) {
    constexpr int CR = 5;
    int strips = 0;
    for (int i = 0; i < prm.nplanes; ++i) {
        BLPlane &p = prm.p[i];
        p.block0 = strips;
        p.nbx = (p.w + (64 - 2 * CR) - 1) / (64 - 2 * CR);
        strips += p.nbx * ((p.h + kWalkBand - 1) / kWalkBand);
    }
    const int grid = std::min((strips + 7) / 8, 256);
    int rcs = vszip_ensure_scratch(ctx, 512);
    if (rcs != VSZIP_OK) return rcs;
    int *counter = static_cast<int *>(ctx->scratch);
    VSZIP_HIP_CHECK(ctx, hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));
    vszip_probe_scope probe(ctx);
    if (form == 3)
        hipLaunchKernelGGL((bilateral_walk36_kernel<3, T>), dim3(grid), dim3(512), 0, ctx->stream, prm, strips, counter);
    else if (form == 2)
        hipLaunchKernelGGL((bilateral_walk36_kernel<2, T>), dim3(grid), dim3(512), 0, ctx->stream, prm, strips, counter);
    else if (form == 1)
        hipLaunchKernelGGL((bilateral_walk36_kernel<1, T>), dim3(grid), dim3(512), 0, ctx->stream, prm, strips, counter);
    else
        hipLaunchKernelGGL((bilateral_walk36_kernel<0, T>), dim3(grid), dim3(512), 0, ctx->stream, prm, strips, counter);
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    return VSZIP_OK;
}

template <typename T>
int launch_truncated(vszip_ctx *ctx, const BLParams &prm, int blocks, bool tiled, bool joint, int max_radius) {
    {
        vszip_probe_scope probe(ctx);
        if (tiled) {
            size_t lds = (size_t)(kBX + 2 * max_radius) * (kTileH + 2 * max_radius) * sizeof(T) * (joint ? 2 : 1);
            const bool ldslut = BSmp<T>::is_int && prm.lut_len <= kLdsLutMax;
            BLParams q = prm;
            if (ldslut) {
                lds = (lds + 15) & ~(size_t)15;
                q.lut_offset = (int)lds;
                lds += (size_t)q.lut_len * sizeof(float);
            }
            const dim3 grid(blocks), block(kBX, kBY);
            if (joint && ldslut)
                hipLaunchKernelGGL((bilateral_tiled_kernel<T, true, true>), grid, block, lds, ctx->stream, q);
            else if (joint)
                hipLaunchKernelGGL((bilateral_tiled_kernel<T, true, false>), grid, block, lds, ctx->stream, q);
            else if (ldslut)
                hipLaunchKernelGGL((bilateral_tiled_kernel<T, false, true>), grid, block, lds, ctx->stream, q);
            else
                hipLaunchKernelGGL((bilateral_tiled_kernel<T, false, false>), grid, block, lds, ctx->stream, q);
        } else {
            hipLaunchKernelGGL((bilateral_truncated_kernel<T>), dim3(blocks), dim3(kBX, kBY), 0, ctx->stream, prm);
        }
    }
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    return VSZIP_OK;
}

}  // namespace

// (vszip_bilateral_derive — bilateralCreate's per-plane derivation — is device-free: host_params.cpp)

// vszip_dev_free forgets a packed range LUT with its allocation
void vszip_bilateral_forget_lut(const void *dptr) {
    PackedLuts &pl = packed_luts();
    std::lock_guard<std::mutex> lk(pl.mu);
    pl.exact.erase(dptr);
    pl.fine.erase(dptr);
    pl.plateau.erase(dptr);
    pl.cubic.erase(dptr);
}

// LUTs exactly as bilateral.zig:306-339 computes them (f64 exp on the host, cast to f32),
// uploaded to device memory owned by the caller (vszip_dev_free).
VSZIP_EXPORT int vszip_bilateral_luts(vszip_ctx *ctx, vszip_bilateral_cfg *cfg, int hist_len) {
    if (!ctx || !cfg || hist_len <= 0) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    cfg->gs_lut = cfg->gr_lut = nullptr;
    if (!cfg->process) return VSZIP_OK;
    const double range = (double)(hist_len - 1);
    std::vector<float> gr((size_t)hist_len);
    const uint32_t upper = (uint32_t)std::trunc(std::min(range, cfg->sigmaR * 8.0 * range + 0.5));
    uint32_t i = 0;
    for (; i <= upper && (int)i < hist_len; ++i) {
        const double x = ((double)i / range) / cfg->sigmaR;
        gr[i] = (float)(std::exp(x * x / -2) / (std::sqrt(2.0 * M_PI) * cfg->sigmaR));
    }
    for (const float up = gr[upper]; (int)i < hist_len; ++i) gr[i] = up;
    // 65536-entry tables: the exact packed form for bilateral_lds16_kernel (base[i >> 6] - delta[i]), kept behind
    // the f32 table in the same allocation when every entry round-trips
    std::vector<uint32_t> pbase;
    std::vector<uint16_t> pdelta;
    bool packed = hist_len == 65536 && !ctx->opt.bilateral_no_lds16;
    if (packed) {
        pbase.resize(1024);
        pdelta.resize(65536);
        for (int b = 0; b < 1024 && packed; ++b) {
            uint32_t hi;
            std::memcpy(&hi, &gr[(size_t)b * 64], 4);
            pbase[b] = hi;
            for (int j = 0; j < 64; ++j) {
                uint32_t v;
                std::memcpy(&v, &gr[(size_t)b * 64 + j], 4);
                if (v > hi || hi - v > 65535u || (v >> 31)) {
                    packed = false;
                    break;
                }
                pdelta[(size_t)b * 64 + j] = (uint16_t)(hi - v);
            }
        }
    }
    // ... and the FINE form for gentle tables (sigmaR around 1.5 and up, the BASELINE's 2 among them): blocks of FOUR entries,
    // bits(gr[i]) = base4[i >> 2] - delta8[i] with a u8 delta (64 + 64 KiB) — the byte offsets of both reads are i itself and
    // i & ~3, no shift: two VALU instructions fewer per lookup in bilateral_walk16_kernel
    std::vector<uint32_t> fbase;
    std::vector<uint8_t> fdelta;
    bool fine = packed;
    if (fine) {
        fbase.resize(16384);
        fdelta.resize(65536);
        for (int b = 0; b < 16384 && fine; ++b) {
            uint32_t hi;
            std::memcpy(&hi, &gr[(size_t)b * 4], 4);
            fbase[b] = hi;
            for (int j = 0; j < 4; ++j) {
                uint32_t v;
                std::memcpy(&v, &gr[(size_t)b * 4 + j], 4);
                if (v > hi || hi - v > 255u) {
                    fine = false;
                    break;
                }
                fdelta[(size_t)b * 4 + j] = (uint8_t)(hi - v);
            }
        }
    }
    // ... and the CUBIC form for what neither a packed form nor the PLATEAU form holds (walk kernels, WalkLut<3>): least-squares cubic per 128 entries,
    // evaluated here with the kernel's three FMAs, + the difference of the bit patterns per entry
    std::vector<float> ccoef;
    std::vector<int8_t> ccorr;
    bool cubic = hist_len == 65536 && !packed && upper + 1 > (uint32_t)kWalkPlateauMax && !ctx->opt.bilateral_no_cubic && !ctx->opt.bilateral_no_lds16;
    if (cubic) {
        ccoef.assign((size_t)kCubicSeg * 4, 0.0f);
        ccorr.assign(65536, 0);
        for (int sg = 0; sg < kCubicSeg && cubic; ++sg) {
            double A[4][5] = {};
            int n = 0;
            for (int t = 0; t < 128; ++t) {
                const uint32_t idx = (uint32_t)sg * 128 + t;
                if (idx > upper) break;
                const double z = t / 128.0, y = (double)gr[idx];
                double pw[7] = {1, z, z * z, z * z * z, 0, 0, 0};
                pw[4] = pw[3] * z;
                pw[5] = pw[4] * z;
                pw[6] = pw[5] * z;
                for (int a = 0; a < 4; ++a) {
                    for (int b = 0; b < 4; ++b) A[a][b] += pw[a + b];
                    A[a][4] += pw[a] * y;
                }
                ++n;
            }
            if (n == 0) break;
            const int deg = n >= 8 ? 4 : (n >= 2 ? 2 : 1);
            for (int a = 0; a < deg; ++a) {  // Gauss-Jordan with pivoting
                int pv = a;
                for (int b = a + 1; b < deg; ++b)
                    if (std::fabs(A[b][a]) > std::fabs(A[pv][a])) pv = b;
                for (int c = 0; c < 5; ++c) std::swap(A[a][c], A[pv][c]);
                for (int b = 0; b < deg; ++b) {
                    if (b == a) continue;
                    const double f = A[b][a] / A[a][a];
                    for (int c = a; c < 5; ++c) A[b][c] -= f * A[a][c];
                }
            }
            double cz[4] = {0, 0, 0, 0};
            for (int a = 0; a < deg; ++a) cz[a] = A[a][4] / A[a][a];
            float *c = &ccoef[(size_t)sg * 4];
            c[0] = (float)cz[0];
            c[1] = (float)(cz[1] / 128.0);
            c[2] = (float)(cz[2] / (128.0 * 128.0));
            c[3] = (float)(cz[3] / (128.0 * 128.0 * 128.0));
            for (int t = 0; t < 128; ++t) {
                const uint32_t idx = (uint32_t)sg * 128 + t;
                if (idx > upper) break;
                const float tt = (float)t;
                const float a = std::fmaf(std::fmaf(std::fmaf(c[3], tt, c[2]), tt, c[1]), tt, c[0]);
                int32_t ba, bt;
                std::memcpy(&ba, &a, 4);
                std::memcpy(&bt, &gr[idx], 4);
                const long df = (long)bt - (long)ba;
                if (df < -128 || df > 127 || !(a > 0.0f)) {
                    cubic = false;
                    break;
                }
                ccorr[idx] = (int8_t)df;
            }
        }
    }
    void *d = nullptr;
    const size_t gr_bytes = gr.size() * sizeof(float);
    if (vszip_hip_malloc(ctx, &d, gr_bytes + (packed ? (size_t)kL16LutBytes : 0) + (fine ? (size_t)kWalkFineBytes : 0) + (cubic ? (size_t)kCubicBytes : 0)) != hipSuccess)
        return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "Bilateral: range LUT allocation failed");
    auto upload = [&](size_t off, const void *src, size_t n) { return hipMemcpy(static_cast<char *>(d) + off, src, n, hipMemcpyHostToDevice) == hipSuccess; };
    bool up_ok = upload(0, gr.data(), gr_bytes);
    if (packed) up_ok = up_ok && upload(gr_bytes, pbase.data(), 4096) && upload(gr_bytes + 4096, pdelta.data(), 131072);
    if (fine) up_ok = up_ok && upload(gr_bytes + kL16LutBytes, fbase.data(), 65536) && upload(gr_bytes + kL16LutBytes + 65536, fdelta.data(), 65536);
    if (cubic) up_ok = up_ok && upload(gr_bytes, ccoef.data(), (size_t)kCubicSeg * 16) && upload(gr_bytes + (size_t)kCubicSeg * 16, ccorr.data(), 65536);  // (never beside a packed form)
    if (!up_ok) {
        (void)hipGetLastError();
        (void)hipFree(d);  // (the allocation leaked on a failed upload)
        return vszip_set_error(ctx, VSZIP_ERR_HIP, "Bilateral: range LUT upload failed");
    }
    {
        PackedLuts &pl = packed_luts();
        std::lock_guard<std::mutex> lk(pl.mu);
        // the registry is keyed by the raw pointer: an address reused after a free that bypassed vszip_dev_free must not keep a stale entry
        pl.exact.erase(d);
        pl.fine.erase(d);
        pl.plateau.erase(d);
        pl.cubic.erase(d);
        if (cubic) {
            uint64_t key;
            std::memcpy(&key, &cfg->sigmaR, sizeof key);
            pl.cubic[d] = {key, upper};
        }
        if (hist_len % 4 == 0 && upper + 1 <= (uint32_t)kWalkPlateauMax && !ctx->opt.bilateral_no_lds16) {  // (8- / 10- / 12-bit clips: the whole table is that short)
            uint64_t key;
            std::memcpy(&key, &cfg->sigmaR, sizeof key);
            pl.plateau[d] = {key, upper};
        }
        if (packed) {
            uint64_t key;
            std::memcpy(&key, &cfg->sigmaR, sizeof key);
            pl.exact[d] = key;
            if (fine) pl.fine.insert(d);
        }
    }
    cfg->gr_lut = static_cast<float *>(d);
    if (cfg->algorithm == 2) {
        const int up2 = cfg->radius + 1;
        std::vector<float> gs((size_t)up2 * up2);
        for (int y = 0; y < up2; ++y)
            for (int x = 0; x < up2; ++x) gs[(size_t)y * up2 + x] = (float)std::exp((double)(x * x + y * y) / (cfg->sigmaS * cfg->sigmaS * -2.0));
        if (vszip_hip_malloc(ctx, &d, gs.size() * sizeof(float)) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "Bilateral: spatial LUT allocation failed");
        VSZIP_HIP_CHECK(ctx, hipMemcpy(d, gs.data(), gs.size() * sizeof(float), hipMemcpyHostToDevice));
        cfg->gs_lut = static_cast<float *>(d);
    }
    return VSZIP_OK;
}

static int bilateral_alg2(vszip_ctx *ctx, int dtype, const vszip_plane *planes, const vszip_bilateral_cfg *const *cfgs, int nplanes, float peak);

VSZIP_EXPORT int vszip_bilateral(vszip_ctx *ctx, int dtype, const vszip_plane *planes, const vszip_bilateral_cfg *const *cfgs, int nplanes, float peak) {
    if (!ctx || !planes || !cfgs || nplanes <= 0) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // The algorithm-1 planes of the call — typically the luma of every frame: sigmaS is halved for subsampled chroma, which then stays with algorithm 2 —
    // are independent of the others: they are gathered and share their launches (run_pbfic), the rest keeps its order.
    int n1 = 0;
    for (int i = 0; i < nplanes; ++i) n1 += (cfgs[i] && cfgs[i]->process && cfgs[i]->algorithm == 1) ? 1 : 0;
    if (n1 == 0) return bilateral_alg2(ctx, dtype, planes, cfgs, nplanes, peak);
    std::vector<vszip_plane> p1, p2;
    std::vector<const vszip_bilateral_cfg *> c1, c2;
    for (int i = 0; i < nplanes; ++i) {
        const bool a1 = cfgs[i] && cfgs[i]->process && cfgs[i]->algorithm == 1;
        (a1 ? p1 : p2).push_back(planes[i]);
        (a1 ? c1 : c2).push_back(cfgs[i]);
    }
    int rc1;
    switch (dtype) {
        case VSZIP_U8: rc1 = run_pbfic<uint8_t>(ctx, p1.data(), c1.data(), n1, peak); break;
        case VSZIP_U16: rc1 = run_pbfic<uint16_t>(ctx, p1.data(), c1.data(), n1, peak); break;
        case VSZIP_F16: rc1 = run_pbfic<_Float16>(ctx, p1.data(), c1.data(), n1, peak); break;
        case VSZIP_F32: rc1 = run_pbfic<float>(ctx, p1.data(), c1.data(), n1, peak); break;
        default: return vszip_set_error(ctx, VSZIP_ERR_ARG, "Bilateral: not supported Int format.");
    }
    if (rc1 != VSZIP_OK || p2.empty()) return rc1;
    return bilateral_alg2(ctx, dtype, p2.data(), c2.data(), (int)p2.size(), peak);
}

static int bilateral_alg2(vszip_ctx *ctx, int dtype, const vszip_plane *planes, const vszip_bilateral_cfg *const *cfgs, int nplanes, float peak) {
    int done = 0;
    while (done < nplanes) {
        BLParams prm;
        prm.peak = peak;
        prm.lut_len = (int)peak + 1;  // hist_len of the clip (bilateral.zig(vs):101-102); the planes of a call share it
        prm.lut_offset = 0;
        int n = 0, blocks = 0;
        // one launch group: the tiled kernel when every radius fits its LDS tile; `joint` if any
        // plane of the group brings a separate ref clip
        bool tiled = !ctx->opt.bilateral_untiled, joint = false;
        int max_radius = 0;
        for (int i = done; i < nplanes && i < done + kMaxPlanesBL; ++i) {
            if (!cfgs[i] || (cfgs[i]->process && cfgs[i]->algorithm == 1)) break;
            tiled = tiled && cfgs[i]->radius <= kTileMaxR;
            joint = joint || (planes[i].ref && planes[i].ref != planes[i].src);
            max_radius = std::max(max_radius, cfgs[i]->radius);
        }
        const int rows_per_block = tiled ? kTileH : kBY;
        for (; done + n < nplanes && n < kMaxPlanesBL; ++n) {
            const vszip_plane &s = planes[done + n];
            const vszip_bilateral_cfg *c = cfgs[done + n];
            if (c && c->process && c->algorithm == 1) break;  // next group
            if (!c || !s.src || !s.dst || s.w <= 0 || s.h <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "Bilateral: bad plane %d", done + n);
            if (!c->process) return vszip_set_error(ctx, VSZIP_ERR_ARG, "Bilateral: plane %d is not processed (sigma 0); copy it instead", done + n);
            if (c->algorithm != 2) return vszip_set_error(ctx, VSZIP_ERR_ARG, "Bilateral: invalid algorithm %d", c->algorithm);
            if (!c->gs_lut || !c->gr_lut) return vszip_set_error(ctx, VSZIP_ERR_ARG, "Bilateral: LUTs missing (vszip_bilateral_luts)");
            // src/vapoursynth/bilateral.zig:206-209
            if (s.w <= 2 * c->radius || s.h <= 2 * c->radius)
                return vszip_set_error(ctx, VSZIP_ERR_ARG, "Bilateral: plane too small for the spatial radius derived from sigmaS; lower sigmaS or use a larger clip.");
            BLPlane &d = prm.p[n];
            d.src = s.src;
            d.ref = s.ref ? s.ref : s.src;
            d.dst = s.dst;
            d.sstride = (int)s.src_stride;
            d.rstride = s.ref ? (int)s.ref_stride : (int)s.src_stride;
            d.dstride = (int)s.dst_stride;
            d.w = s.w;
            d.h = s.h;
            d.block0 = blocks;
            d.nbx = (s.w + kBX - 1) / kBX;
            d.radius = c->radius;
            d.step = c->step;
            d.gs = c->gs_lut;
            d.gr = c->gr_lut;
            blocks += d.nbx * ((s.h + rows_per_block - 1) / rows_per_block);
        }
        prm.nplanes = n;
        const int taken = n;  // planes this turn of the loop consumes (n shrinks when the walk kernel takes some of them)
        int rc;
        // 16-bit / float clips whose range LUTs are all registered in packed form: one persistent launch per
        // distinct table, the whole table in LDS
        bool lds16 = tiled && dtype != VSZIP_U8 && prm.lut_len == 65536;
        {
            const size_t bps = dtype == VSZIP_F32 ? 4 : 2;
            const int tile_h = (dtype == VSZIP_F32 || joint) ? 32 : 64;  // kL16TileH
            const int stages = (dtype == VSZIP_F32 || joint) ? 4 : 6;     // kL16MaxStage
            const size_t tile_elems = (size_t)(kBX + 2 * max_radius) * (tile_h + 2 * max_radius);
            lds16 = lds16 && tile_elems <= (size_t)kBX * kL16Rows * stages && kL16LutBytes + 2 * tile_elems * bps * (joint ? 2 : 1) <= (size_t)kL16MaxLds;
            for (int i = 0; i < n && lds16; ++i) lds16 = lut_is_packed(prm.p[i].gr);
        }
        // Steep tables (the filter's usual sigmaR) with the BASELINE's tap shapes: the walk kernel with the table's computed part in LDS
        // (PLATEAU form), one launch per distinct (table, taps)
        if (!lds16 && tiled && !joint && (dtype == VSZIP_U8 || dtype == VSZIP_U16 || dtype == VSZIP_F32) && !ctx->opt.bilateral_no_walk) {
            struct PKey {
                uint64_t table;
                uint32_t upper;
                int radius, step, form;
                bool operator==(const PKey &o) const { return table == o.table && upper == o.upper && radius == o.radius && step == o.step && form == o.form; }
            };
            std::vector<PKey> keys(n), groups;
            std::vector<char> walk(n, 0);
            for (int i = 0; i < n; ++i) {
                keys[i].radius = prm.p[i].radius;
                keys[i].step = prm.p[i].step;
                keys[i].form = lut_plateau(prm.p[i].gr, &keys[i].table, &keys[i].upper);
                walk[i] = keys[i].form != 0 &&
                          ((keys[i].radius == 3 && keys[i].step == 2) || (keys[i].radius == 2 && keys[i].step == 1) || (keys[i].radius == 5 && keys[i].step == 2 && !ctx->opt.bilateral_no_walk36));
                if (walk[i] && std::find(groups.begin(), groups.end(), keys[i]) == groups.end()) groups.push_back(keys[i]);
            }
            for (const PKey &g : groups) {
                BLParams q;
                q.peak = prm.peak;
                q.lut_len = prm.lut_len;
                q.lut_offset = 0;
                q.lut_upper = (int)g.upper;
                q.nplanes = 0;
                for (int i = 0; i < n; ++i)
                    if (walk[i] && keys[i] == g) q.p[q.nplanes++] = prm.p[i];
                const int f = g.form;
                if (g.radius == 5)
                    rc = dtype == VSZIP_U8 ? launch_walk36<uint8_t>(ctx, q, f) : (dtype == VSZIP_U16 ? launch_walk36<uint16_t>(ctx, q, f) : launch_walk36<float>(ctx, q, f));
                else if (g.radius == 3)
                    rc = dtype == VSZIP_U8 ? launch_walk16<3, 2, uint8_t>(ctx, q, f) : (dtype == VSZIP_U16 ? launch_walk16<3, 2, uint16_t>(ctx, q, f) : launch_walk16<3, 2, float>(ctx, q, f));
                else
                    rc = dtype == VSZIP_U8 ? launch_walk16<2, 1, uint8_t>(ctx, q, f) : (dtype == VSZIP_U16 ? launch_walk16<2, 1, uint16_t>(ctx, q, f) : launch_walk16<2, 1, float>(ctx, q, f));
                if (rc != VSZIP_OK) return rc;
            }
            // planes with other tap shapes (sigmaS = 1's chroma: radius 1; sigmaS = 3's luma: radius 5) stay with the tile kernel below
            int m = 0, nb = 0;
            for (int i = 0; i < n; ++i) {
                if (walk[i]) continue;
                BLPlane d = prm.p[i];
                d.block0 = nb;
                nb += d.nbx * ((d.h + rows_per_block - 1) / rows_per_block);
                prm.p[m++] = d;
            }
            if (m == 0) {
                done += taken;
                continue;
            }
            n = m;
            prm.nplanes = m;
            blocks = nb;
        }
        // ... and steep tables with any other taps (or a `ref` clip): the persistent tile kernel with the same PLATEAU table in LDS
        if (!lds16 && tiled && dtype != VSZIP_U8 && prm.lut_len == 65536 && !ctx->opt.bilateral_no_lds16) {
            const size_t bps = dtype == VSZIP_F32 ? 4 : 2;
            const int tile_h = (dtype == VSZIP_F32 || joint) ? 32 : 64, stages = (dtype == VSZIP_F32 || joint) ? 4 : 6;
            const size_t tile_elems = (size_t)(kBX + 2 * max_radius) * (tile_h + 2 * max_radius);
            bool ok = tile_elems <= (size_t)kBX * kL16Rows * stages && kL16LutBytes + 2 * tile_elems * bps * (joint ? 2 : 1) <= (size_t)kL16MaxLds;
            std::vector<std::pair<uint64_t, uint32_t>> keys(n), groups;
            for (int i = 0; i < n && ok; ++i) {
                ok = lut_plateau(prm.p[i].gr, &keys[i].first, &keys[i].second) == 2;  // (the tile kernel has the plain form only)
                if (ok && std::find(groups.begin(), groups.end(), keys[i]) == groups.end()) groups.push_back(keys[i]);
            }
            if (ok) {
                for (const auto &g : groups) {
                    BLParams q;
                    q.peak = prm.peak;
                    q.lut_len = prm.lut_len;
                    q.lut_offset = 0;
                    q.lut_upper = (int)g.second;
                    q.nplanes = 0;
                    int qb = 0;
                    for (int i = 0; i < n; ++i) {
                        if (!(keys[i] == g)) continue;
                        BLPlane &dp = q.p[q.nplanes++];
                        dp = prm.p[i];
                        dp.block0 = qb;
                        qb += dp.nbx * ((dp.h + tile_h - 1) / tile_h);
                    }
                    switch (dtype) {
                        case VSZIP_U16: rc = launch_lds16<uint16_t>(ctx, q, qb, joint, max_radius, true); break;
                        case VSZIP_F16: rc = launch_lds16<_Float16>(ctx, q, qb, joint, max_radius, true); break;
                        default: rc = launch_lds16<float>(ctx, q, qb, joint, max_radius, true); break;
                    }
                    if (rc != VSZIP_OK) return rc;
                }
                done += taken;
                continue;
            }
        }
        if (lds16) {
            // planes whose tables have the same content and the same radius / step (the planes of an RGB clip, the
            // chroma planes of a YUV one) share a launch, whichever allocation each of them points at
            // (compared field by field — table content = the bits of sigmaR, radius, step —, not through a hash: ADVICE r2)
            struct GroupKey {
                uint64_t table;
                int radius, step;
                bool operator==(const GroupKey &o) const { return table == o.table && radius == o.radius && step == o.step; }
            };
            std::vector<GroupKey> keys(n), tables;
            for (int i = 0; i < n; ++i) {
                lut_is_packed(prm.p[i].gr, &keys[i].table);
                // ... and the same radius / step: the launch then takes the kernel with compile-time taps
                keys[i].radius = prm.p[i].radius;
                keys[i].step = prm.p[i].step;
                if (std::find(tables.begin(), tables.end(), keys[i]) == tables.end()) tables.push_back(keys[i]);
            }
            rc = VSZIP_OK;
            for (const GroupKey &tbl : tables) {
                BLParams q;
                q.peak = prm.peak;
                q.lut_len = prm.lut_len;
                q.lut_offset = 0;
                q.lut_upper = 0;
                q.nplanes = 0;
                int qb = 0;
                for (int i = 0; i < n; ++i) {
                    if (!(keys[i] == tbl)) continue;
                    BLPlane &dp = q.p[q.nplanes++];
                    dp = prm.p[i];
                    dp.block0 = qb;
                    const int tile_h = (dtype == VSZIP_F32 || joint) ? 32 : 64;
                    qb += dp.nbx * ((dp.h + tile_h - 1) / tile_h);
                }
                const bool no_walk = ctx->opt.bilateral_no_walk != 0;
                if ((dtype == VSZIP_U16 || dtype == VSZIP_F32) && !joint && !no_walk && q.nplanes > 0) {
                    int r = q.p[0].radius, st = q.p[0].step;
                    for (int i = 1; i < q.nplanes; ++i)
                        if (q.p[i].radius != r || q.p[i].step != st) r = st = 0;
                    const bool no_fine = ctx->opt.bilateral_no_fine != 0;
                    bool fine = !no_fine;
                    for (int i = 0; i < q.nplanes && fine; ++i) lut_is_packed(q.p[i].gr, nullptr, &fine);
                    if (r == 3 && st == 2) {
                        rc = dtype == VSZIP_U16 ? launch_walk16<3, 2, uint16_t>(ctx, q, fine ? 1 : 0) : launch_walk16<3, 2, float>(ctx, q, fine ? 1 : 0);
                        if (rc != VSZIP_OK) return rc;
                        continue;
                    }
                    if (r == 2 && st == 1) {
                        rc = dtype == VSZIP_U16 ? launch_walk16<2, 1, uint16_t>(ctx, q, fine ? 1 : 0) : launch_walk16<2, 1, float>(ctx, q, fine ? 1 : 0);
                        if (rc != VSZIP_OK) return rc;
                        continue;
                    }
                    if (r == 5 && st == 2 && !ctx->opt.bilateral_no_walk36) {  // sigmaS = 3, the filter's default
                        rc = dtype == VSZIP_U16 ? launch_walk36<uint16_t>(ctx, q, fine ? 1 : 0) : launch_walk36<float>(ctx, q, fine ? 1 : 0);
                        if (rc != VSZIP_OK) return rc;
                        continue;
                    }
                }
                switch (dtype) {
                    case VSZIP_U16: rc = launch_lds16<uint16_t>(ctx, q, qb, joint, max_radius); break;
                    case VSZIP_F16: rc = launch_lds16<_Float16>(ctx, q, qb, joint, max_radius); break;
                    default: rc = launch_lds16<float>(ctx, q, qb, joint, max_radius); break;
                }
                if (rc != VSZIP_OK) return rc;
            }
            done += taken;
            continue;
        }
        switch (dtype) {
            case VSZIP_U8: rc = launch_truncated<uint8_t>(ctx, prm, blocks, tiled, joint, max_radius); break;
            case VSZIP_U16: rc = launch_truncated<uint16_t>(ctx, prm, blocks, tiled, joint, max_radius); break;
            case VSZIP_F16: rc = launch_truncated<_Float16>(ctx, prm, blocks, tiled, joint, max_radius); break;
            case VSZIP_F32: rc = launch_truncated<float>(ctx, prm, blocks, tiled, joint, max_radius); break;
            default: return vszip_set_error(ctx, VSZIP_ERR_ARG, "Bilateral: not supported Int format.");
        }
        if (rc != VSZIP_OK) return rc;
        done += taken;
    }
    return VSZIP_OK;
}
