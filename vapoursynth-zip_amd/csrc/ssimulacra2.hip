// vszip.SSIMULACRA2 on gfx950: the per-frame kernel of src/filters/ssimulacra2.zig
// (`process`, :46-136) — two linear-light RGBS frames in, one f64 score out.
//
// Scales 0 and 1 (94 % of the samples) come out of ONE pass over the source:
//   ssim_pyr_kernel       a thread owns a 4x4 block of source samples of both frames: it converts them
//                         to linear RGB (the colour pre-stage of hz.toRGBS + sRGBtoLinearRGB fused in:
//                         integer / gamma sources go through a lookup table, see vszip_ssimulacra2_src),
//                         writes the XYB planes scale 0 needs, box-filters to its 2x2 block of scale 1
//                         (downscale :138-209) and writes that scale's XYB planes, box-filters once more
//                         and writes its one sample of scale-2 linear RGB. The source is read once and
//                         neither scale 0's nor scale 1's linear RGB pyramid ever exists in HBM.
// Per scale s = 2..4 (scale 5 carries only pruned weights, :22-37, and is never built):
//   ssim_xyb_down_kernel  reads the linear RGB of both frames at scale s once and writes
//                         (a) the XYB planes scale s actually needs (toXYB :392-472 with the
//                         VCL cbrt of src/vcl.zig:40-81) and (b) the 2x2-box-filtered linear
//                         RGB of scale s+1 (downscale :138-209) — one read feeds both.
//   ssim_maps_kernel      one 32x32 tile per block and one launch per scale for all active
//                         planes: stages the XYB tile (+4 halo) of both frames in LDS, forms
//                         im1*im2, (im1+im2)^2, im1, im2 on the fly, runs the separable 9-tap
//                         FIR (blur :247-372: vertical then horizontal, asymmetric mirror),
//                         evaluates ssimMap / edgeMap (:480-628) per pixel in f64 and reduces
//                         d, d^4, artifact, artifact^4, detail, detail^4 to one partial per
//                         block. No blurred map ever touches HBM.
//   ssim_final_kernel     folds the partials in a fixed order (reproducible) into the
//                         6x6 / 6x12 average tables; the host applies `score` (:630-663).
// All pairs of a call share every launch (grid z = pair): each pair has its own XYB planes
// and RGB pyramid in the context scratch (261 MB per 4K pair — HBM is 288 GB), and a
// device table of plane pointers per (scale, pair) tells a block where its pair lives.
// Tiles that touch no plane border take a fast path with compile-time tap offsets.
// f32 arithmetic keeps the reference's operation order; the vertical FIR is fused
// (fmaf) for columns below w - w % 8 and unfused beyond, as the reference's AVX2 build
// does (:318 vs :326). Only the f64 pooling order differs (~1e-15 relative).
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kScales = 5;  // scales that carry a non-pruned weight
constexpr int TW = 56, TH = 32, HALO = 4, IW = TW + 2 * HALO, IH = TH + 2 * HALO;  // maps tiles: 56 x 32 outputs from 64 x 40 staged samples
constexpr int kVecW = 8;    // reference SIMD width baked into the FMA rule
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
#define VSZIP_GLOBAL __attribute__((address_space(1)))  // a pointer known to be global memory: global_load / global_store, not flat_*

__constant__ float c_kernel[9] = {
    0.0076144188642501831054687500f, 0.0360749699175357818603515625f, 0.1095860823988914489746093750f,
    0.2134445458650588989257812500f, 0.2665599882602691650390625000f, 0.2134445458650588989257812500f,
    0.1095860823988914489746093750f, 0.0360749699175357818603515625f, 0.0076144188642501831054687500f,
};

// src/vcl.zig:40-81
__device__ __forceinline__ float vcl_cbrt(float x) {
    const float one_third = 1.0f / 3.0f, four_third = 4.0f / 3.0f;
    const float xa = fabsf(x);
    const float xa3 = one_third * xa;
    const uint32_t m1 = __float_as_uint(xa);
    float a = __uint_as_float(0x54800000u - ((m1 >> 23) * 0x002AAAAAu));
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const float a2 = a * a;
        a = (four_third * a) - (xa3 * (a2 * a2));
    }
    const float a2 = a * a;
    a = a + (one_third * (a - (xa * (a2 * a2))));
    a = (a * a) * x;
    return m1 <= 0x00800000u ? 0.0f : a;
}

struct XybK {
    float m[9], bias, kd1;
};

// plane pointers of one (scale, pair): the kernels index this table by blockIdx.z
struct PairPtrs {
    const float *rgb1[3], *rgb2[3];  // scale s linear RGB
    float *xyb1[3], *xyb2[3];        // scale s XYB (NULL = plane not needed)
    float *next1[3], *next2[3];      // scale s+1 linear RGB (NULL at the last scale)
};

struct XybArgs {
    const PairPtrs *tab;             // [npairs] for this scale
    int stride, w, h;                // scale s geometry (elements)
    int nstride, nw, nh;             // scale s+1 geometry
    int xstride;
    XybK k;
};

__device__ __forceinline__ void to_xyb_px(const XybK &k, float r, float g, float b, bool need_b, float &X, float &Y, float &B) {
    const float ox = fmaf(k.m[0], r, fmaf(k.m[1], g, fmaf(k.m[2], b, k.bias)));
    const float oy = fmaf(k.m[3], r, fmaf(k.m[4], g, fmaf(k.m[5], b, k.bias)));
    const float cx = vcl_cbrt(fmaxf(ox, 0.0f)) - k.kd1;
    const float cy = vcl_cbrt(fmaxf(oy, 0.0f)) - k.kd1;
    const float xv = 0.5f * (cx - cy);
    const float yv = 0.5f * (cx + cy);
    X = xv * 14.0f + 0.42f;
    Y = yv + 0.01f;
    if (need_b) {
        const float oz = fmaf(k.m[6], r, fmaf(k.m[7], g, fmaf(k.m[8], b, k.bias)));
        const float cz = vcl_cbrt(fmaxf(oz, 0.0f)) - k.kd1;
        B = (cz - yv) + 0.55f;
    }
}

// Two samples per instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32): element-wise IEEE operations in
// the scalar forms' order, so every lane value equals vcl_cbrt / to_xyb_px of that sample.
__device__ __forceinline__ v2f vcl_cbrt2(v2f x) {
    const v2f one_third = {1.0f / 3.0f, 1.0f / 3.0f}, four_third = {4.0f / 3.0f, 4.0f / 3.0f};
    const v2f xa = {fabsf(x.x), fabsf(x.y)};
    const v2f xa3 = one_third * xa;
    const uint32_t m0 = __float_as_uint(xa.x), m1 = __float_as_uint(xa.y);
    v2f a = {__uint_as_float(0x54800000u - ((m0 >> 23) * 0x002AAAAAu)), __uint_as_float(0x54800000u - ((m1 >> 23) * 0x002AAAAAu))};
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const v2f a2 = a * a;
        a = (four_third * a) - (xa3 * (a2 * a2));
    }
    const v2f a2 = a * a;
    a = a + (one_third * (a - (xa * (a2 * a2))));
    a = (a * a) * x;
    return v2f{m0 <= 0x00800000u ? 0.0f : a.x, m1 <= 0x00800000u ? 0.0f : a.y};
}

__device__ __forceinline__ v2f splat2(float v) { return v2f{v, v}; }
__device__ __forceinline__ v2f max0_2(v2f v) { return v2f{fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f)}; }

__device__ __forceinline__ void to_xyb_px2(const XybK &k, v2f r, v2f g, v2f b, bool need_b, v2f &X, v2f &Y, v2f &B) {
    const v2f bias = splat2(k.bias), kd1 = splat2(k.kd1);
    const v2f ox = __builtin_elementwise_fma(splat2(k.m[0]), r, __builtin_elementwise_fma(splat2(k.m[1]), g, __builtin_elementwise_fma(splat2(k.m[2]), b, bias)));
    const v2f oy = __builtin_elementwise_fma(splat2(k.m[3]), r, __builtin_elementwise_fma(splat2(k.m[4]), g, __builtin_elementwise_fma(splat2(k.m[5]), b, bias)));
    const v2f cx = vcl_cbrt2(max0_2(ox)) - kd1;
    const v2f cy = vcl_cbrt2(max0_2(oy)) - kd1;
    const v2f xv = splat2(0.5f) * (cx - cy);
    const v2f yv = splat2(0.5f) * (cx + cy);
    X = xv * splat2(14.0f) + splat2(0.42f);
    Y = yv + splat2(0.01f);
    if (need_b) {
        const v2f oz = __builtin_elementwise_fma(splat2(k.m[6]), r, __builtin_elementwise_fma(splat2(k.m[7]), g, __builtin_elementwise_fma(splat2(k.m[8]), b, bias)));
        const v2f cz = vcl_cbrt2(max0_2(oz)) - kd1;
        B = (cz - yv) + splat2(0.55f);
    }
}

// One thread per scale-(s+1) pixel = one 2x2 block of scale-s pixels.
__global__ __launch_bounds__(256) void ssim_xyb_down_kernel(const XybArgs a) {
    const int ox = blockIdx.x * 32 + (threadIdx.x & 31);
    const int oy = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (ox >= a.nw || oy >= a.nh) return;
    const PairPtrs pp = a.tab[blockIdx.z];
    const bool need_b = pp.xyb1[2] != nullptr;
#pragma unroll
    for (int img = 0; img < 2; ++img) {
        const float *const *rgb = img ? pp.rgb2 : pp.rgb1;
        float *const *xyb = img ? pp.xyb2 : pp.xyb1;
        float *const *nxt = img ? pp.next2 : pp.next1;
        float v[3][4];
        // downscale :186-200: samples clamp to the last row/column, summed ((a+b)+c)+d
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                const int x = min(ox * 2 + ix, a.w - 1), y = min(oy * 2 + iy, a.h - 1);
                const size_t o = (size_t)y * a.stride + x;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c][iy * 2 + ix] = ((const float __attribute__((address_space(1))) *)rgb[c])[o];  // (the planes come out of a table: typed as global memory, or the accesses are flat ones)
            }
        if (nxt[0]) {
            const size_t o = (size_t)oy * a.nstride + ox;
#pragma unroll
            for (int c = 0; c < 3; ++c) ((float __attribute__((address_space(1))) *)nxt[c])[o] = (((v[c][0] + v[c][1]) + v[c][2]) + v[c][3]) * 0.25f;
        }
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                const int x = ox * 2 + ix, y = oy * 2 + iy;
                if (x >= a.w || y >= a.h) continue;
                float X, Y, B = 0.0f;
                to_xyb_px(a.k, v[0][iy * 2 + ix], v[1][iy * 2 + ix], v[2][iy * 2 + ix], need_b, X, Y, B);
                const size_t o = (size_t)y * a.xstride + x;
                typedef float __attribute__((address_space(1))) *GOut;
                if (xyb[0]) ((GOut)xyb[0])[o] = X;
                if (xyb[1]) ((GOut)xyb[1])[o] = Y;
                if (need_b) ((GOut)xyb[2])[o] = B;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Scales 0 + 1 + the scale-2 RGB in one pass, colour pre-stage included.
// ---------------------------------------------------------------------------------------------
enum { PYR_F32_LINEAR = 0, PYR_F32_GAMMA = 1, PYR_INT = 2, PYR_YUV = 3 };

struct PyrPair {
    const void *src1[3], *src2[3];  // source planes of the two frames (Gray: [0] only)
    float *x0a[3], *x0b[3];         // scale-0 XYB planes of frame 1 / 2 (NULL = pruned)
    float *x1a[3], *x1b[3];         // scale-1 XYB planes
    float *r2a[3], *r2b[3];         // scale-2 linear RGB
    float *rgb1[3], *rgb2[3];       // 4:2:0 integer clips (round 5): the frames' linear RGB, written by ssim_yuv420_rgb_kernel, read by the f32 pass (PyrArgs::from_rgb)
};

struct PyrArgs {
    const PyrPair *tab;
    const float *lut;   // PYR_INT: 2^bits entries indexed by the sample; PYR_F32_GAMMA: the 65537-entry transfer table
    int lut_lds;        // PYR_INT: number of entries staged in LDS (0: gathered from global memory)
    int vec_ok;         // every source plane base and the row pitch are aligned to 4 samples, and w % 4 == 0
    int sstride;        // source row pitch, elements
    int from_rgb;       // PYR_F32_LINEAR: the sources are the pair's rgb1 / rgb2 planes (dense rows of w floats), not src1 / src2
    int w, h, w1, h1, w2, h2;
    XybK k;
};

constexpr int kPyrLdsLut = 4096;  // integer clips up to 12 bit keep their whole table in LDS

template <typename T, int MODE>
__device__ __forceinline__ float pyr_linear(T v, const float *lut, const float *lds_lut, bool use_lds) {
    if constexpr (MODE == PYR_F32_LINEAR) {
        return (float)v;
    } else if constexpr (MODE == PYR_F32_GAMMA) {
        // zimg's approximate-gamma table: index rint(x * 32768 + 16384) over [-0.5, 1.5], clamped
        float t = rintf(fmaf((float)v, 32768.0f, 16384.0f));
        t = fminf(fmaxf(t, 0.0f), 65536.0f);
        return lut[(int)t];
    } else {
        // (typed address spaces and a workgroup-uniform branch: a select between the two generic pointers made every lookup a flat load - round 5)
        if (use_lds) return ((const float __attribute__((address_space(3))) *)lds_lut)[(uint32_t)v];
        return ((const float __attribute__((address_space(1))) *)lut)[(uint32_t)v];
    }
}

// four adjacent samples of one plane row: one 16-byte store, or the `left` (< 4) samples inside the plane
__device__ __forceinline__ void pyr_put4(float __attribute__((address_space(1))) *pl, size_t o, bool vec, int left, float v0, float v1, float v2, float v3) {
    if (!pl) return;
    if (vec) {
        *reinterpret_cast<v4f VSZIP_GLOBAL *>(pl + o) = v4f{v0, v1, v2, v3};
    } else {
        pl[o] = v0;
        if (left > 1) pl[o + 1] = v1;
        if (left > 2) pl[o + 2] = v2;
        if (left > 3) pl[o + 3] = v3;
    }
}

// The second half of a thread's work: its 4x4 block of linear RGB -> the XYB planes of scale 0, the 2x2 box and
// XYB of scale 1, and its one sample of scale-2 linear RGB.
typedef float __attribute__((address_space(1))) *PyrGOut;
template <bool FAST>
__device__ __forceinline__ void pyr_emit(const PyrArgs &a, const PyrGOut (&o0)[3], const PyrGOut (&o1)[3], const PyrGOut (&o2)[3], const float (&lin)[3][4][4], int bx, int by) {
    const int x0 = bx * 4, y0 = by * 4;
    // scale 0: XYB of the 16 samples, two per instruction; every value is computed, only the stores are guarded
    {
        const bool nb = o0[2] != nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v2f X01, Y01, B01 = {0.0f, 0.0f}, X23, Y23, B23 = {0.0f, 0.0f};
            to_xyb_px2(a.k, v2f{lin[0][r][0], lin[0][r][1]}, v2f{lin[1][r][0], lin[1][r][1]}, v2f{lin[2][r][0], lin[2][r][1]}, nb, X01, Y01, B01);
            to_xyb_px2(a.k, v2f{lin[0][r][2], lin[0][r][3]}, v2f{lin[1][r][2], lin[1][r][3]}, v2f{lin[2][r][2], lin[2][r][3]}, nb, X23, Y23, B23);
            if (FAST || y0 + r < a.h) {
                const size_t o = (size_t)(y0 + r) * a.w + x0;
                pyr_put4(o0[0], o, FAST, a.w - x0, X01.x, X01.y, X23.x, X23.y);
                pyr_put4(o0[1], o, FAST, a.w - x0, Y01.x, Y01.y, Y23.x, Y23.y);
                pyr_put4(o0[2], o, FAST, a.w - x0, B01.x, B01.y, B23.x, B23.y);
            }
        }
    }
    // scale 1: 2x2 box of linear RGB, summed ((a+b)+c)+d (:186-200), then XYB
    float l1[3][2][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                l1[c][j][i] = (((lin[c][2 * j][2 * i] + lin[c][2 * j][2 * i + 1]) + lin[c][2 * j + 1][2 * i]) + lin[c][2 * j + 1][2 * i + 1]) * 0.25f;
    {
        const bool nb = o1[2] != nullptr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            v2f X, Y, B = {0.0f, 0.0f};
            to_xyb_px2(a.k, v2f{l1[0][j][0], l1[0][j][1]}, v2f{l1[1][j][0], l1[1][j][1]}, v2f{l1[2][j][0], l1[2][j][1]}, nb, X, Y, B);
            const int x = 2 * bx, y = 2 * by + j;
            if (FAST || y < a.h1) {
                const size_t o = (size_t)y * a.w1 + x;
                const bool two = FAST || x + 1 < a.w1;
                if (FAST && (a.w1 & 1) == 0) {  // 8-byte stores: every row of scale 1 starts 8-byte aligned
                    typedef v2f VSZIP_GLOBAL *G2;
                    if (o1[0]) *reinterpret_cast<G2>(o1[0] + o) = X;
                    if (o1[1]) *reinterpret_cast<G2>(o1[1] + o) = Y;
                    if (nb) *reinterpret_cast<G2>(o1[2] + o) = B;
                } else {
                    if (o1[0]) { o1[0][o] = X.x; if (two) o1[0][o + 1] = X.y; }
                    if (o1[1]) { o1[1][o] = Y.x; if (two) o1[1][o + 1] = Y.y; }
                    if (nb) { o1[2][o] = B.x; if (two) o1[2][o + 1] = B.y; }
                }
            }
        }
    }
    // scale 2: one sample; a missing scale-1 column / row takes its neighbour (the same clamp one level up)
    if (FAST || (bx < a.w2 && by < a.h2)) {
        const bool i1 = FAST || 2 * bx + 1 < a.w1, j1 = FAST || 2 * by + 1 < a.h1;
        const size_t o = (size_t)by * a.w2 + bx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // selects, not l1[c][j1][i1]: a runtime index would move the array out of the registers
            const float p00 = l1[c][0][0], p01 = i1 ? l1[c][0][1] : l1[c][0][0];
            const float p10 = j1 ? l1[c][1][0] : l1[c][0][0];
            const float p11 = j1 ? (i1 ? l1[c][1][1] : l1[c][1][0]) : p01;
            o2[c][o] = (((p00 + p01) + p10) + p11) * 0.25f;
        }
    }
}

// The work of one thread. FAST is a property of the whole workgroup (every 4x4 block of it lies inside the
// plane and every row start is aligned): loads and stores are then unconditional vector accesses — no divergent
// branch around a memory instruction, so the compiler issues an image's 12 row loads back to back behind ONE
// s_waitcnt (with a per-thread `if (full)` around each load it waited for every load separately: 12 dependent
// round trips per image, 81 us per 4K pair instead of 50).
struct PyrPtrs {  // one frame's plane pointers, held in (scalar) registers
    const void *src[3];
    float *o0[3], *o1[3], *o2[3];
};

template <typename T, int MODE, bool GRAY, bool FAST>
__device__ __forceinline__ void pyr_image(const PyrArgs &a, const PyrPtrs &pp, const float *lds_lut, bool use_lds, int bx, int by) {
    const int x0 = bx * 4, y0 = by * 4;
    constexpr int NP = GRAY ? 1 : 3;
    {
        // plane pointers read from a table are generic to the compiler (flat_load / flat_store): they are global memory
        typedef const T __attribute__((address_space(1))) *GSrc;
        typedef float __attribute__((address_space(1))) *GOut;
        const GSrc src[3] = {(GSrc)pp.src[0], (GSrc)pp.src[1], (GSrc)pp.src[2]};
        const GOut o0[3] = {(GOut)pp.o0[0], (GOut)pp.o0[1], (GOut)pp.o0[2]};
        const GOut o1[3] = {(GOut)pp.o1[0], (GOut)pp.o1[1], (GOut)pp.o1[2]};
        const GOut o2[3] = {(GOut)pp.o2[0], (GOut)pp.o2[1], (GOut)pp.o2[2]};
        // 4x4 block -> linear RGB. Samples past the right / bottom edge take the edge sample, which is
        // exactly the clamp of downscale (:186-200): min(2*ox + ix, w - 1).
        T raw[NP][4][4];
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            const GSrc pl = src[c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (FAST) {
                    const GSrc row = pl + (size_t)(y0 + r) * a.sstride + x0;
                    // one 4 / 8 / 16-byte load, unpacked with shifts (an array of T in a struct goes to scratch)
                    if constexpr (sizeof(T) == 4) {
                        const v4f q = *reinterpret_cast<const v4f VSZIP_GLOBAL *>(row);
                        raw[c][r][0] = (T)q.x; raw[c][r][1] = (T)q.y; raw[c][r][2] = (T)q.z; raw[c][r][3] = (T)q.w;
                    } else if constexpr (sizeof(T) == 2) {
                        const v2u q = *reinterpret_cast<const v2u VSZIP_GLOBAL *>(row);
                        raw[c][r][0] = (T)(q.x & 0xffffu); raw[c][r][1] = (T)(q.x >> 16); raw[c][r][2] = (T)(q.y & 0xffffu); raw[c][r][3] = (T)(q.y >> 16);
                    } else {
                        const uint32_t q = *reinterpret_cast<const uint32_t __attribute__((address_space(1))) *>(row);
                        raw[c][r][0] = (T)(q & 0xffu); raw[c][r][1] = (T)((q >> 8) & 0xffu); raw[c][r][2] = (T)((q >> 16) & 0xffu); raw[c][r][3] = (T)(q >> 24);
                    }
                } else {
                    const GSrc row = pl + (size_t)min(y0 + r, a.h - 1) * a.sstride;
#pragma unroll
                    for (int i = 0; i < 4; ++i) raw[c][r][i] = row[min(x0 + i, a.w - 1)];
                }
            }
        }
        float lin[3][4][4];
        if (use_lds) {  // (one uniform branch around all 48 lookups, not one each)
#pragma unroll
            for (int c = 0; c < NP; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) lin[c][r][i] = pyr_linear<T, MODE>(raw[c][r][i], a.lut, lds_lut, true);
        } else {
#pragma unroll
            for (int c = 0; c < NP; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) lin[c][r][i] = pyr_linear<T, MODE>(raw[c][r][i], a.lut, lds_lut, false);
        }
        if constexpr (GRAY) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) lin[1][r][i] = lin[2][r][i] = lin[0][r][i];
        }
        pyr_emit<FAST>(a, o0, o1, o2, lin, bx, by);
    }
}

template <typename T, int MODE, bool GRAY>
__global__ __launch_bounds__(256) void ssim_pyr_kernel(const PyrArgs a) {
    __shared__ float lds_lut[(MODE == PYR_INT) ? kPyrLdsLut : 1];
    const bool use_lds = MODE == PYR_INT && a.lut_lds > 0;
    if (use_lds) {
        for (int i = threadIdx.x; i < a.lut_lds; i += 256) lds_lut[i] = a.lut[i];
        __syncthreads();
    }
    const int bx = blockIdx.x * 64 + (threadIdx.x & 63), by = blockIdx.y * 4 + (threadIdx.x >> 6);
    // Both frames' plane pointers, read ONCE before any store (through the table in memory the compiler must
    // assume that a store to an output plane changed them, and reloads — and waits for — a pointer before
    // every access). Statically indexed copies: they live in SGPRs.
    const PyrPair *__restrict__ tp = a.tab + blockIdx.z;
    PyrPtrs f1, f2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f1.src[c] = tp->src1[c]; f1.o0[c] = tp->x0a[c]; f1.o1[c] = tp->x1a[c]; f1.o2[c] = tp->r2a[c];
        f2.src[c] = tp->src2[c]; f2.o0[c] = tp->x0b[c]; f2.o1[c] = tp->x1b[c]; f2.o2[c] = tp->r2b[c];
        if constexpr (MODE == PYR_F32_LINEAR && !GRAY) {
            if (a.from_rgb) f1.src[c] = tp->rgb1[c], f2.src[c] = tp->rgb2[c];
        }
    }
    // workgroup-uniform: the whole 256 x 16 sample block is inside the plane and the planes allow vector accesses
    const bool fast = a.vec_ok && (int)(blockIdx.x + 1) * 256 <= a.w && (int)(blockIdx.y + 1) * 16 <= a.h;
    if (fast) {
        pyr_image<T, MODE, GRAY, true>(a, f1, lds_lut, use_lds, bx, by);
        pyr_image<T, MODE, GRAY, true>(a, f2, lds_lut, use_lds, bx, by);
    } else {
        if (bx * 4 >= a.w || by * 4 >= a.h) return;
        pyr_image<T, MODE, GRAY, false>(a, f1, lds_lut, use_lds, bx, by);
        pyr_image<T, MODE, GRAY, false>(a, f2, lds_lut, use_lds, bx, by);
    }
}

// ---------------------------------------------------------------------------------------------
// YUV sources (round 3): hz.toRGBS on a subsampled clip, fused into the same pass. Per frame and workgroup
// (256 x 16 luma samples): phase A brings the chroma rows the block needs to full WIDTH (zimg's horizontal pass
// comes first when both axes double) and parks them in LDS — thread t owns luma column X0 + t, its 4 taps and
// coefficients come from the per-axis tables of vszip_resample_table; phase B is pyr_image's job with a different
// front end: a thread reads its 4x4 luma block, finishes the chroma with the vertical 4-tap pass from LDS,
// applies zimg's integer -> float conversion, the YUV -> RGB matrix (FMA chain) and the transfer table.
// Every f32 operation is the one oracle/vs_host.py::yuv_to_rgbs performs, in its order: bit-exact.
// ---------------------------------------------------------------------------------------------
struct YuvArgs {
    const int *hleft, *vleft;      // [w], [h]: first tap of a luma column / row in the chroma plane (NULL: axis not subsampled)
    const float *hcoef, *vcoef;    // [4 w], [4 h]
    int cstride, cw, ch;           // chroma row pitch (elements) and size
    float ys, yo, cs, co;          // integer -> float: fma(v, ys, yo) for luma, fma(v, cs, co) for chroma
    float m[9];                    // YUV -> RGB, row major
    int linearize;                 // 0: _Transfer == LINEAR
};

constexpr int kYuvRows = 16;  // chroma rows a 16-row luma block can need: 16 (not subsampled vertically) or <= 8 + 3 + 1

__device__ __forceinline__ float yuv_two_acc(float c0, float c1, float c2, float c3, float x0, float x1, float x2, float x3) {
    float a0 = c0 * x0, a1 = c1 * x1;
    a0 = fmaf(c2, x2, a0);
    a1 = fmaf(c3, x3, a1);
    return a0 + a1;
}

template <typename T>
__device__ __forceinline__ float yuv_cvt(T v, float s, float o) {
    if constexpr (sizeof(T) == 4)
        return (float)v;
    else
        return fmaf((float)v, s, o);
}

__device__ __forceinline__ float yuv_transfer(float x, const float *lut, int linearize) {
    if (!linearize) return x;
    float t = rintf(fmaf(x, 32768.0f, 16384.0f));
    t = fminf(fmaxf(t, 0.0f), 65536.0f);
    return lut[(int)t];
}

template <typename T, bool FAST>
__device__ __forceinline__ void pyr_image_yuv(const PyrArgs &a, const YuvArgs &ya, const PyrPtrs &pp, float (*hbuf)[kYuvRows][256], int bx, int by) {
    typedef const T __attribute__((address_space(1))) *GSrc;
    const GSrc src[3] = {(GSrc)pp.src[0], (GSrc)pp.src[1], (GSrc)pp.src[2]};
    const PyrGOut o0[3] = {(PyrGOut)pp.o0[0], (PyrGOut)pp.o0[1], (PyrGOut)pp.o0[2]};
    const PyrGOut o1[3] = {(PyrGOut)pp.o1[0], (PyrGOut)pp.o1[1], (PyrGOut)pp.o1[2]};
    const PyrGOut o2[3] = {(PyrGOut)pp.o2[0], (PyrGOut)pp.o2[1], (PyrGOut)pp.o2[2]};
    const int X0 = blockIdx.x * 256, Y0 = blockIdx.y * 16;
    const int ylast = min(Y0 + 15, a.h - 1);
    // the chroma rows the tile taps: from the SMALLEST first tap of its rows (co-sited chroma: a row that sits on a chroma row has one
    // non-zero tap, the row below it four that start one row EARLIER - the table is not monotonic; round 5)
    int r0 = Y0, r1 = ylast;
    if (ya.vleft) {
        r0 = ya.ch;
        r1 = 0;
        for (int y = Y0; y <= ylast; ++y) {
            const int v = ya.vleft[y];
            r0 = min(r0, v);
            r1 = max(r1, min(v + 3, ya.ch - 1));
        }
    }
    // ---- phase A: chroma rows r0 .. r1 at full width into hbuf[plane][row - r0][column - X0]
    {
        const int t = threadIdx.x;
        const int X = min(X0 + t, a.w - 1);
        int hl = X;
        float c0 = 1.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
        if (ya.hleft) {
            hl = ya.hleft[X];
            const v4f c = *reinterpret_cast<const v4f *>(ya.hcoef + 4 * (size_t)X);
            c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
        }
        const int i0 = hl, i1 = min(hl + 1, ya.cw - 1), i2 = min(hl + 2, ya.cw - 1), i3 = min(hl + 3, ya.cw - 1);
        for (int r = r0; r <= r1; ++r) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const GSrc row = src[1 + pl] + (size_t)r * ya.cstride;
                float v;
                if (ya.hleft) {
                    const T q0 = row[i0], q1 = row[i1], q2 = row[i2], q3 = row[i3];
                    v = yuv_two_acc(c0, c1, c2, c3, yuv_cvt<T>(q0, ya.cs, ya.co), yuv_cvt<T>(q1, ya.cs, ya.co), yuv_cvt<T>(q2, ya.cs, ya.co), yuv_cvt<T>(q3, ya.cs, ya.co));
                } else {
                    v = yuv_cvt<T>(row[i0], ya.cs, ya.co);
                }
                hbuf[pl][r - r0][t] = v;
            }
        }
    }
    __syncthreads();
    // ---- phase B: the thread's 4x4 block
    if (FAST || (bx * 4 < a.w && by * 4 < a.h)) {
        const int x0 = bx * 4, y0 = by * 4;
        const int lx = x0 - X0;  // local column of the block in hbuf
        float lin[3][4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = FAST ? y0 + r : min(y0 + r, a.h - 1);
            // luma row
            float yy[4];
            {
                const GSrc row = src[0] + (size_t)y * a.sstride;
                if constexpr (FAST) {
                    if constexpr (sizeof(T) == 4) {
                        const v4f q = *reinterpret_cast<const v4f VSZIP_GLOBAL *>(row + x0);
                        yy[0] = q.x; yy[1] = q.y; yy[2] = q.z; yy[3] = q.w;
                    } else if constexpr (sizeof(T) == 2) {
                        const v2u q = *reinterpret_cast<const v2u VSZIP_GLOBAL *>(row + x0);
                        yy[0] = yuv_cvt<uint16_t>((uint16_t)(q.x & 0xffffu), ya.ys, ya.yo); yy[1] = yuv_cvt<uint16_t>((uint16_t)(q.x >> 16), ya.ys, ya.yo);
                        yy[2] = yuv_cvt<uint16_t>((uint16_t)(q.y & 0xffffu), ya.ys, ya.yo); yy[3] = yuv_cvt<uint16_t>((uint16_t)(q.y >> 16), ya.ys, ya.yo);
                    } else {
                        const uint32_t q = *reinterpret_cast<const uint32_t __attribute__((address_space(1))) *>(row + x0);
                        yy[0] = yuv_cvt<uint8_t>((uint8_t)(q & 0xffu), ya.ys, ya.yo); yy[1] = yuv_cvt<uint8_t>((uint8_t)((q >> 8) & 0xffu), ya.ys, ya.yo);
                        yy[2] = yuv_cvt<uint8_t>((uint8_t)((q >> 16) & 0xffu), ya.ys, ya.yo); yy[3] = yuv_cvt<uint8_t>((uint8_t)(q >> 24), ya.ys, ya.yo);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) yy[i] = yuv_cvt<T>(row[min(x0 + i, a.w - 1)], ya.ys, ya.yo);
                }
            }
            // chroma: vertical pass over the LDS rows (wave-uniform row and coefficients)
            float uv[2][4];
            {
                int vl = y - r0;
                float c0 = 1.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
                if (ya.vleft) {
                    vl = ya.vleft[y] - r0;
                    const v4f c = *reinterpret_cast<const v4f *>(ya.vcoef + 4 * (size_t)y);
                    c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
                }
                const int nr = r1 - r0;
                const int j0 = vl, j1 = min(vl + 1, nr), j2 = min(vl + 2, nr), j3 = min(vl + 3, nr);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    if (ya.vleft) {
                        float t0[4], t1[4], t2[4], t3[4];
                        if constexpr (FAST) {
                            const v4f q0 = *reinterpret_cast<const v4f *>(&hbuf[pl][j0][lx]), q1 = *reinterpret_cast<const v4f *>(&hbuf[pl][j1][lx]);
                            const v4f q2 = *reinterpret_cast<const v4f *>(&hbuf[pl][j2][lx]), q3 = *reinterpret_cast<const v4f *>(&hbuf[pl][j3][lx]);
                            t0[0] = q0.x; t0[1] = q0.y; t0[2] = q0.z; t0[3] = q0.w;
                            t1[0] = q1.x; t1[1] = q1.y; t1[2] = q1.z; t1[3] = q1.w;
                            t2[0] = q2.x; t2[1] = q2.y; t2[2] = q2.z; t2[3] = q2.w;
                            t3[0] = q3.x; t3[1] = q3.y; t3[2] = q3.z; t3[3] = q3.w;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int c = min(lx + i, min(a.w - 1 - X0, 255));
                                t0[i] = hbuf[pl][j0][c]; t1[i] = hbuf[pl][j1][c]; t2[i] = hbuf[pl][j2][c]; t3[i] = hbuf[pl][j3][c];
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) uv[pl][i] = yuv_two_acc(c0, c1, c2, c3, t0[i], t1[i], t2[i], t3[i]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) uv[pl][i] = hbuf[pl][j0][FAST ? lx + i : min(lx + i, min(a.w - 1 - X0, 255))];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float g = fmaf(ya.m[3 * c + 2], uv[1][i], fmaf(ya.m[3 * c + 1], uv[0][i], ya.m[3 * c] * yy[i]));
                    lin[c][r][i] = yuv_transfer(g, a.lut, ya.linearize);
                }
            }
        }
        pyr_emit<FAST>(a, o0, o1, o2, lin, bx, by);
    }
    __syncthreads();  // hbuf is reused by the other frame
}

template <typename T>
__global__ __launch_bounds__(256) void ssim_pyr_yuv_kernel(const PyrArgs a, const YuvArgs ya) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][kYuvRows][256];
    const int bx = blockIdx.x * 64 + (threadIdx.x & 63), by = blockIdx.y * 4 + (threadIdx.x >> 6);
    const PyrPair *__restrict__ tp = a.tab + blockIdx.z;
    PyrPtrs f1, f2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f1.src[c] = tp->src1[c]; f1.o0[c] = tp->x0a[c]; f1.o1[c] = tp->x1a[c]; f1.o2[c] = tp->r2a[c];
        f2.src[c] = tp->src2[c]; f2.o0[c] = tp->x0b[c]; f2.o1[c] = tp->x1b[c]; f2.o2[c] = tp->r2b[c];
    }
    const bool fast = a.vec_ok && (int)(blockIdx.x + 1) * 256 <= a.w && (int)(blockIdx.y + 1) * 16 <= a.h;
    if (fast) {
        pyr_image_yuv<T, true>(a, ya, f1, hbuf, bx, by);
        pyr_image_yuv<T, true>(a, ya, f2, hbuf, bx, by);
    } else {
        pyr_image_yuv<T, false>(a, ya, f1, hbuf, bx, by);
        pyr_image_yuv<T, false>(a, ya, f2, hbuf, bx, by);
    }
}

// ---------------------------------------------------------------------------------------------
// 4:2:0 integer YUV sources, round 5: the colour pre-stage as a pass of its own, with persistent workgroups.
// ssim_pyr_yuv_kernel spends 43 % of its time in the transfer table (48 gathers a thread from a 256 KB table: 40 GB of L2
// requests a launch, the L2's request rate) and 25 % in phase A's single-sample loads. A table in LDS needs a workgroup that
// lives long enough to pay for loading it, and one that owns a CU's LDS cannot also hold the XYB conversion's 120+ registers
// at a useful occupancy (a fused persistent kernel was built first: no faster, profiles/r05_notes.md 9, 12). So the work is split:
// ssim_yuv420_rgb_kernel (here: one workgroup a CU, the table from 0 to 1.14 in LDS, what lies beyond in global memory)
// writes each frame's LINEAR RGB planes, and ssim_pyr_kernel<float, PYR_F32_LINEAR> (PyrArgs::from_rgb) reads them back -
// 200 MB a 4K pair through HBM, both passes at rates the fused kernel is far from.
// A workgroup is TEAMS teams of 256 threads; a team takes a tile of 256 x 16 luma samples: its raw chroma samples (<= 12 rows
// of <= 132 a plane) and luma samples are fetched one tile AHEAD into registers with whole-dword loads and parked in the team's
// LDS slice; every thread filters what its 4 x 4 block needs straight from there - horizontal 4-tap on the 6 chroma
// rows the block touches, then the vertical 4-tap, both in zimg's order (yuv_two_acc): no plane of horizontally filtered rows.
// Whole tiles take taps and coefficients from the kernel argument - the resampling tables' period, which every column and row
// follows except the frame's first and last few: those samples come out wrong and ssim_yuv420_fixup_kernel redoes them, one thread
// a sample (72 k of a 4K frame's 8.3 M). Ragged tiles at the right and the bottom go through a second launch of the same kernel
// (EDGE) that reads the tables and clamps like ssim_pyr_yuv_kernel - a path of its own, because its table values cost the interior's
// a wave a SIMD, and as a called function its spills landed on the common path. Same f32 operations in the same order.
// ---------------------------------------------------------------------------------------------
constexpr int kYlRows = 12;  // chroma rows a 16-row luma tile of a 4:2:0 clip taps at most
constexpr int kYlNr = 6;     // ... and a 4-row luma block

struct YuvLds {
    const int *hspan, *vspan;  // [2 * tile column] / [2 * tile row]: first and last chroma column / row the tile taps
    int dh[4], dv[4];          // regular columns / rows: first tap of sample x0 + i = (x0 >> 1) + dh[i], x0 a multiple of 4
    float ch[4][4], cv[4][4];  // ... and their coefficients
    int xr0, xr1, yr0, yr1;    // the regular ranges [xr0, xr1) x [yr0, yr1), multiples of 4
    int tx0, tx1, ty0, ty1;    // the interior launch's tiles [tx0, tx1) x [ty0, ty1): whole and vector-aligned (the EDGE launch takes the others)
    int lut_lo, lut_n;         // LDS holds lut[lut_lo, lut_lo + lut_n)
    int low_zero;              // lut[0 .. lut_lo] are all 0.0f: an index below the range reads entry lut_lo
    int nbx, nby, ntiles;      // tiles across and down a frame; tiles of the launch (pairs x 2 frames x the interior's, or the rest's, tiles)
};

template <typename T>
struct YlGeo {
    static constexpr int B = (int)sizeof(T);
    static constexpr int PD = B == 1 ? 36 : 68;            // dwords a staged chroma row: 132 samples + alignment slack
    static constexpr int RAW_DW = 2 * kYlRows * PD;        // one tile's raw chroma, both planes
    static constexpr int NLD = (RAW_DW + 255) / 256;       // staging loads a thread
    static constexpr int TEAMS = B == 1 ? 4 : 3;           // what leaves (most of) the table's [0, 1] range room in 160 KB
};

constexpr int kYlEdgeTeams = 2;  // the EDGE launch: its table values and selects want 256 registers a thread
template <typename T, bool EDGE>  // EDGE: the launch over the frame's outermost tiles (everything outside the interior)
__global__ __launch_bounds__(256 * (EDGE ? kYlEdgeTeams : YlGeo<T>::TEAMS)) void ssim_yuv420_rgb_kernel(const PyrArgs a, const YuvArgs ya, const YuvLds yl) {
    using G = YlGeo<T>;
    constexpr int B = G::B, PD = G::PD, RAW_DW = G::RAW_DW, NLD = G::NLD, TEAMS = EDGE ? kYlEdgeTeams : G::TEAMS;
    typedef const T __attribute__((address_space(1))) *GSrc;
    typedef const uint32_t __attribute__((address_space(1))) *GDw;
    typedef const float __attribute__((address_space(3))) *LdsF;
    typedef const float __attribute__((address_space(1))) *GblF;
    extern __shared__ __attribute__((aligned(16))) uint32_t yl_lds[];
    float *lut = reinterpret_cast<float *>(yl_lds);
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), t = threadIdx.x & 255;
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    uint32_t *cur_raw = yl_lds + ((yl.lut_n + 3) & ~3) + team * RAW_DW;  // (one slice a team: a second one would cost the table 3 400 entries of its upper end)
    for (int i = threadIdx.x; i < yl.lut_n; i += 256 * TEAMS) lut[i] = a.lut[yl.lut_lo + i];

    const int tw = yl.tx1 - yl.tx0, th = yl.ty1 - yl.ty0;
    const int n_top = yl.nbx * yl.ty0, n_bottom = yl.nbx * (yl.nby - yl.ty1), side = yl.nbx - tw;  // EDGE: full rows above and below, the columns either side
    const int tiles_a_frame = EDGE ? n_top + n_bottom + th * side : tw * th, stride = gridDim.x * TEAMS;
    // one tile's geometry (team-uniform) and this thread's fetched samples
    struct Tile {
        int bxt, byt, z, f, cba, r0, r1;
        bool on, fast;
        uint32_t c[NLD], y[4][B];
    };
    auto fetch = [&](int id, Tile &q) {
        q.on = id < yl.ntiles;
        if (!q.on) return;
        const int fr = id / tiles_a_frame, rem = id - fr * tiles_a_frame;
        q.z = fr >> 1;
        q.f = fr & 1;
        if constexpr (!EDGE) {
            const int ty = rem / tw;
            q.byt = yl.ty0 + ty;
            q.bxt = yl.tx0 + rem - ty * tw;
        } else if (rem < n_top + n_bottom) {
            const int e = rem < n_top ? rem : rem - n_top, ty = e / yl.nbx;
            q.byt = rem < n_top ? ty : yl.ty1 + ty;
            q.bxt = e - ty * yl.nbx;
        } else {
            const int e = rem - n_top - n_bottom, ty = e / side, c = e - ty * side;
            q.byt = yl.ty0 + ty;
            q.bxt = c < yl.tx0 ? c : yl.tx1 + (c - yl.tx0);
        }
        q.r0 = yl.vspan[2 * q.byt];
        q.r1 = yl.vspan[2 * q.byt + 1];
        q.cba = (yl.hspan[2 * q.bxt] * B) & ~3;  // bytes
        q.fast = !EDGE || (a.vec_ok && (q.bxt + 1) * 256 <= a.w && (q.byt + 1) * 16 <= a.h);
        const PyrPair *__restrict__ tp = a.tab + q.z;
        const int crow_bytes = ya.cw * B;  // a multiple of 4 (host-checked): whole dwords never leave a row
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = t + 256 * k, prow = idx / PD, dcol = idx - prow * PD;
            const int pl = prow >= kYlRows ? 1 : 0, row = prow - pl * kYlRows;
            q.c[k] = 0;
            if (idx < RAW_DW && q.cba + 4 * dcol < crow_bytes) {
                const char *base = static_cast<const char *>(q.f ? tp->src2[1 + pl] : tp->src1[1 + pl]);
                q.c[k] = *(GDw)(base + (size_t)min(q.r0 + row, q.r1) * ya.cstride * B + q.cba + 4 * dcol);
            }
        }
        if (q.fast) {
            const char *base = static_cast<const char *>(q.f ? tp->src2[0] : tp->src1[0]);
            const int x0 = (q.bxt * 64 + lane) * 4, y0 = (q.byt * 4 + wv) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < B; ++j) q.y[r][j] = *(GDw)(base + ((size_t)(y0 + r) * a.sstride + x0) * B + 4 * j);
        }
    };

    Tile nx;
    int id = blockIdx.x * TEAMS + team;
    fetch(id, nx);
    const int rounds = (yl.ntiles + stride - 1) / stride;  // the same for every team: the barrier below is the whole workgroup's
    int dvmin = yl.dv[0];
#pragma unroll
    for (int r = 1; r < 4; ++r) dvmin = min(dvmin, yl.dv[r]);
    for (int round = 0; round < rounds; ++round, id += stride) {
        if (round) __syncthreads();  // every wave is done with the previous tile's samples
        if (nx.on) {
#pragma unroll
            for (int k = 0; k < NLD; ++k)
                if (t + 256 * k < RAW_DW) cur_raw[t + 256 * k] = nx.c[k];
        }
        __syncthreads();  // (also: the table is in place before the first tile)
        const Tile q = nx;
        fetch(id + stride, nx);
        if (!q.on) continue;

        const PyrPair *__restrict__ tp = a.tab + q.z;
        typedef float __attribute__((address_space(1))) *GOut;
        GOut out[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = (GOut)(q.f ? tp->rgb2[c] : tp->rgb1[c]);
        const int bx = q.bxt * 64 + lane, by = q.byt * 4 + wv, x0 = bx * 4, y0 = by * 4;
        if (!q.fast && !(x0 < a.w && y0 < a.h)) continue;
        float uv[2][4][4];
        if constexpr (!EDGE) {
            // Byte offset, in a staged row, of the first tap of each of the block's 4 columns (no clamp: the taps are inside the plane). The 16
            // taps of a row lie within NW aligned dwords from the first one's: those are read whole and each column's 4 samples cut out with
            // v_alignbyte (sample-wise LDS reads at odd byte addresses kept the LDS pipe busy 20 cycles an instruction).
            constexpr int NW = B == 1 ? 3 : 4;  // (host-checked: dh spans at most 2 samples)
            int tb[4], tbmin;
#pragma unroll
            for (int i = 0; i < 4; ++i) tb[i] = ((x0 >> 1) + yl.dh[i]) * B - q.cba;
            tbmin = min(min(tb[0], tb[1]), min(tb[2], tb[3])) & ~3;
#pragma unroll
            for (int i = 0; i < 4; ++i) tb[i] -= tbmin;  // 0 .. 4 NW - 4 B
            const int jrow = (y0 >> 1) + dvmin - q.r0;  // the block's first chroma row in the staged tile
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const char *row0 = reinterpret_cast<const char *>(cur_raw + pl * kYlRows * PD) + jrow * (PD * 4) + tbmin;
                float hv[kYlNr][4];
#pragma unroll
                for (int jr = 0; jr < kYlNr; ++jr) {
                    uint32_t d[NW + 1];
#pragma unroll
                    for (int m = 0; m < NW; ++m) d[m] = *reinterpret_cast<const uint32_t *>(row0 + jr * (PD * 4) + 4 * m);
                    d[NW] = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // the 4 B bytes from offset tb[i]: dwords tb[i] / 4 .. of d, shifted by tb[i] % 4 bytes
                        uint32_t wlo = d[0], wmd = d[1], whi = d[2];
#pragma unroll
                        for (int m = 1; m + 1 < NW; ++m)
                            if ((tb[i] >> 2) == m) wlo = d[m], wmd = d[m + 1], whi = d[m + 2];
                        const uint32_t s0 = __builtin_amdgcn_alignbyte(wmd, wlo, (uint32_t)tb[i] & 3u);
                        float x[4];
                        if constexpr (B == 1) {
                            x[0] = yuv_cvt<uint8_t>((uint8_t)(s0 & 0xffu), ya.cs, ya.co); x[1] = yuv_cvt<uint8_t>((uint8_t)((s0 >> 8) & 0xffu), ya.cs, ya.co);
                            x[2] = yuv_cvt<uint8_t>((uint8_t)((s0 >> 16) & 0xffu), ya.cs, ya.co); x[3] = yuv_cvt<uint8_t>((uint8_t)(s0 >> 24), ya.cs, ya.co);
                        } else {
                            const uint32_t s1 = __builtin_amdgcn_alignbyte(whi, wmd, (uint32_t)tb[i] & 3u);
                            x[0] = yuv_cvt<uint16_t>((uint16_t)(s0 & 0xffffu), ya.cs, ya.co); x[1] = yuv_cvt<uint16_t>((uint16_t)(s0 >> 16), ya.cs, ya.co);
                            x[2] = yuv_cvt<uint16_t>((uint16_t)(s1 & 0xffffu), ya.cs, ya.co); x[3] = yuv_cvt<uint16_t>((uint16_t)(s1 >> 16), ya.cs, ya.co);
                        }
                        hv[jr][i] = yuv_two_acc(yl.ch[i][0], yl.ch[i][1], yl.ch[i][2], yl.ch[i][3], x[0], x[1], x[2], x[3]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = yl.dv[r] - dvmin;  // uniform: 0 .. kYlNr - 4
#pragma unroll
                    for (int bb = 0; bb <= kYlNr - 4; ++bb)
                        if (b == bb) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                uv[pl][r][i] = yuv_two_acc(yl.cv[r][0], yl.cv[r][1], yl.cv[r][2], yl.cv[r][3], hv[bb][i], hv[bb + 1][i], hv[bb + 2][i], hv[bb + 3][i]);
                        }
                }
            }
        } else {
            // taps and coefficients out of the tables, samples clamped like ssim_pyr_yuv_kernel's; the vertical taps picked out of the kYlNr
            // filtered rows with (wave-uniform) selects
            int jmin = ya.ch;
#pragma unroll
            for (int r = 0; r < 4; ++r) jmin = min(jmin, ya.vleft[min(y0 + r, a.h - 1)]);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const char *plane = reinterpret_cast<const char *>(cur_raw + pl * kYlRows * PD);
                float hv[kYlNr][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int X = min(x0 + i, a.w - 1), hl = ya.hleft[X];
                    const v4f c = *reinterpret_cast<const v4f *>(ya.hcoef + 4 * (size_t)X);
                    int to[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) to[k] = min(hl + k, ya.cw - 1) * B - q.cba;
#pragma unroll
                    for (int jr = 0; jr < kYlNr; ++jr) {
                        const char *row = plane + (min(jmin + jr, q.r1) - q.r0) * (PD * 4);
                        float x[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {  // (the sample out of its aligned dword: sample-wise reads at odd addresses are slow)
                            const uint32_t dw = *reinterpret_cast<const uint32_t *>(row + (to[k] & ~3));
                            const uint32_t sv = B == 1 ? (dw >> (8 * (to[k] & 3))) & 0xffu : (dw >> (8 * (to[k] & 2))) & 0xffffu;
                            x[k] = yuv_cvt<T>((T)sv, ya.cs, ya.co);
                        }
                        hv[jr][i] = yuv_two_acc(c.x, c.y, c.z, c.w, x[0], x[1], x[2], x[3]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = min(y0 + r, a.h - 1), vl = ya.vleft[y];
                    const v4f c = *reinterpret_cast<const v4f *>(ya.vcoef + 4 * (size_t)y);
                    float x[4][4];  // [tap][column]: chroma row min(vl + k, ch - 1), one of the kYlNr rows from jmin
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int jj = min(vl + k, ya.ch - 1) - jmin;
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) {
                            float v = hv[0][ii];
#pragma unroll
                            for (int bb = 1; bb < kYlNr; ++bb) v = jj == bb ? hv[bb][ii] : v;
                            x[k][ii] = v;
                        }
                    }
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) uv[pl][r][ii] = yuv_two_acc(c.x, c.y, c.z, c.w, x[0][ii], x[1][ii], x[2][ii], x[3][ii]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float yy[4];
            if (q.fast) {
                if constexpr (B == 1) {
                    const uint32_t v = q.y[r][0];
                    yy[0] = yuv_cvt<uint8_t>((uint8_t)(v & 0xffu), ya.ys, ya.yo); yy[1] = yuv_cvt<uint8_t>((uint8_t)((v >> 8) & 0xffu), ya.ys, ya.yo);
                    yy[2] = yuv_cvt<uint8_t>((uint8_t)((v >> 16) & 0xffu), ya.ys, ya.yo); yy[3] = yuv_cvt<uint8_t>((uint8_t)(v >> 24), ya.ys, ya.yo);
                } else {
                    const uint32_t v0 = q.y[r][0], v1 = q.y[r][B - 1];
                    yy[0] = yuv_cvt<uint16_t>((uint16_t)(v0 & 0xffffu), ya.ys, ya.yo); yy[1] = yuv_cvt<uint16_t>((uint16_t)(v0 >> 16), ya.ys, ya.yo);
                    yy[2] = yuv_cvt<uint16_t>((uint16_t)(v1 & 0xffffu), ya.ys, ya.yo); yy[3] = yuv_cvt<uint16_t>((uint16_t)(v1 >> 16), ya.ys, ya.yo);
                }
            } else {
                const GSrc row = (GSrc)(q.f ? tp->src2[0] : tp->src1[0]) + (size_t)min(y0 + r, a.h - 1) * a.sstride;
#pragma unroll
                for (int i = 0; i < 4; ++i) yy[i] = yuv_cvt<T>(row[min(x0 + i, a.w - 1)], ya.ys, ya.yo);
            }
            // The row's 12 lookups: out of LDS (typed address spaces: a select between two generic pointers would turn every lookup into a flat
            // load); entries outside the staged range are fetched afterwards, all twelve loads of a row behind ONE branch (a branch and a
            // wait per lookup made the kernel four times slower: out-of-gamut samples are common in converted video). low_zero: the table is
            // 0 up to lut_lo (zimg clamps negative input first), so only values beyond the range's upper end leave LDS.
            float lin[3][4];
            uint32_t li[3][4], li_max = 0;  // index into the staged range; beyond it (or, without low_zero, below it: a wrapped negative): >= lut_n
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float g = fmaf(ya.m[3 * c + 2], uv[1][r][i], fmaf(ya.m[3 * c + 1], uv[0][r][i], ya.m[3 * c] * yy[i]));
                    // (no clamp to [0, 65536] here: the conversion saturates, the LDS index is clamped below and the global one in the branch)
                    int k = (int)rintf(fmaf(g, 32768.0f, 16384.0f)) - yl.lut_lo;
                    if (yl.low_zero) k = max(k, 0);
                    li[c][i] = (uint32_t)k;
                    li_max = max(li_max, (uint32_t)k);
                    lin[c][i] = ((LdsF)lut)[min((uint32_t)k, (uint32_t)yl.lut_n - 1u)];
                }
            }
            if (__builtin_amdgcn_ballot_w64(li_max >= (uint32_t)yl.lut_n) != 0) {  // (a wave-level branch: a lane-level one was if-converted into twelve loads a row, always)
                asm volatile("" ::: "memory");
                float gl[3][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int t = min(max((int)li[c][i] + yl.lut_lo, 0), 65536);
                        gl[c][i] = ((GblF)a.lut)[li[c][i] >= (uint32_t)yl.lut_n ? t : 0];
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 3; ++c) lin[c][i] = li[c][i] >= (uint32_t)yl.lut_n ? gl[c][i] : lin[c][i];
            }
            if (q.fast || y0 + r < a.h) {
                const size_t o = (size_t)(y0 + r) * a.w + x0;
#pragma unroll
                for (int c = 0; c < 3; ++c) pyr_put4(out[c], o, q.fast, a.w - x0, lin[c][0], lin[c][1], lin[c][2], lin[c][3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 4:4:4 and 4:2:2 integer YUV sources, round 6: the same split as 4:2:0's — the colour pre-stage as a pass of its own with the transfer
// table in LDS, the f32 pyramid pass behind it — for the formats whose chroma rows ARE the luma rows (no vertical resampling), where
// the pass needs no tiles at all: a persistent workgroup a CU (1024 threads), a work item = 4 neighbouring luma samples of a row,
// every sample read with whole-dword loads straight from global memory (4:2:2: the 6 chroma samples its four 4-tap windows span come as
// 3 / 4 aligned dwords, cut with v_alignbyte; taps and coefficients of the horizontal table's period in the kernel argument; the groups
// of the frame's first and last columns, where the table leaves its period, read the table and clamp). The fused kernel these formats
// took until now is bound by the L2's request rate (48 gathers a thread from the 256 KB table). Same f32 operations in the same order.
// ---------------------------------------------------------------------------------------------
struct YuvRow {
    int dh[4];            // regular columns: first tap of sample x0 + i = (x0 >> 1) + dh[i], x0 a multiple of 4 (4:2:2)
    float ch[4][4];       // ... and their coefficients
    int xr0, xr1;         // the regular range [xr0, xr1), multiples of 4
    int lut_lo, lut_n, low_zero;  // as YuvLds
    int gpr;              // groups a row (w / 4)
    long ngroups;         // of the launch: pairs x 2 frames x h x gpr
};

template <typename T, bool HSUB>
__global__ __launch_bounds__(1024) void ssim_yuvrow_rgb_kernel(const PyrArgs a, const YuvArgs ya, const YuvRow yr) {
    constexpr int B = (int)sizeof(T);
    typedef const uint32_t __attribute__((address_space(1))) *GDw;
    typedef const T __attribute__((address_space(1))) *GSrc;
    typedef const float __attribute__((address_space(3))) *LdsF;
    typedef const float __attribute__((address_space(1))) *GblF;
    typedef float __attribute__((address_space(1))) *GOut;
    extern __shared__ __attribute__((aligned(16))) uint32_t yr_lds[];
    float *lut = reinterpret_cast<float *>(yr_lds);
    for (int i = threadIdx.x; i < yr.lut_n; i += 1024) lut[i] = a.lut[yr.lut_lo + i];
    __syncthreads();
    const long per_frame = (long)a.h * yr.gpr, stride = (long)gridDim.x * 1024;
    auto unpack4 = [](const uint32_t *d, float s, float o, float *out) {  // four samples from their dword(s) -> zimg's float
        if constexpr (B == 1) {
            out[0] = yuv_cvt<uint8_t>((uint8_t)(d[0] & 0xffu), s, o); out[1] = yuv_cvt<uint8_t>((uint8_t)((d[0] >> 8) & 0xffu), s, o);
            out[2] = yuv_cvt<uint8_t>((uint8_t)((d[0] >> 16) & 0xffu), s, o); out[3] = yuv_cvt<uint8_t>((uint8_t)(d[0] >> 24), s, o);
        } else {
            out[0] = yuv_cvt<uint16_t>((uint16_t)(d[0] & 0xffffu), s, o); out[1] = yuv_cvt<uint16_t>((uint16_t)(d[0] >> 16), s, o);
            out[2] = yuv_cvt<uint16_t>((uint16_t)(d[1] & 0xffffu), s, o); out[3] = yuv_cvt<uint16_t>((uint16_t)(d[1] >> 16), s, o);
        }
    };
    // A work item's samples are requested one item AHEAD (the loop was bound by the latency of load -> convert -> store with one item a
    // thread in flight: 850 us a launch of 16 4K frames against 450 with the next item's loads under the current one's arithmetic).
    constexpr int NW = B == 1 ? 3 : 4;  // 4:2:2: aligned dwords that hold a group's 16 taps (host-checked: dh spans at most 2 samples)
    struct Item {
        long g;
        int y, x0, fr, tbmin;
        bool regular;
        uint32_t dy[B], dc[2][HSUB ? NW : B];
    };
    auto fetch = [&](long g, Item &q) {
        q.g = g;
        if (g >= yr.ngroups) return;
        q.fr = (int)(g / per_frame);
        const int rem = (int)(g - (long)q.fr * per_frame);
        q.y = rem / yr.gpr;
        q.x0 = (rem - q.y * yr.gpr) * 4;
        const PyrPair *__restrict__ tp = a.tab + (q.fr >> 1);
        const bool second = q.fr & 1;
        const char *sy = static_cast<const char *>(second ? tp->src2[0] : tp->src1[0]) + ((size_t)q.y * a.sstride + q.x0) * B;
#pragma unroll
        for (int j = 0; j < B; ++j) q.dy[j] = *(GDw)(sy + 4 * j);
        q.regular = !HSUB || (q.x0 >= yr.xr0 && q.x0 + 4 <= yr.xr1);
        q.tbmin = 0;
        if constexpr (HSUB) {
            int tmin = ((q.x0 >> 1) + yr.dh[0]) * B;
#pragma unroll
            for (int i = 1; i < 4; ++i) tmin = min(tmin, ((q.x0 >> 1) + yr.dh[i]) * B);
            q.tbmin = tmin & ~3;
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const char *crow = static_cast<const char *>(second ? tp->src2[1 + pl] : tp->src1[1 + pl]) + (size_t)q.y * ya.cstride * B;
            if constexpr (!HSUB) {
#pragma unroll
                for (int j = 0; j < B; ++j) q.dc[pl][j] = *(GDw)(crow + (size_t)q.x0 * B + 4 * j);
            } else {
                const int last_dw = ya.cw * B - 4;  // byte offset of the row's last whole dword (cw * B is a multiple of 4)
#pragma unroll
                for (int m = 0; m < NW; ++m) q.dc[pl][m] = *(GDw)(crow + min(max(q.tbmin, 0) + 4 * m, last_dw));
            }
        }
    };
    Item nx;
    fetch((long)blockIdx.x * 1024 + threadIdx.x, nx);
    while (nx.g < yr.ngroups) {
        const Item q = nx;
        fetch(q.g + stride, nx);
        const int y = q.y, x0 = q.x0;
        const PyrPair *__restrict__ tp = a.tab + (q.fr >> 1);
        const bool second = q.fr & 1;
        float yy[4], uv[2][4];
        unpack4(q.dy, ya.ys, ya.yo, yy);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            if constexpr (!HSUB) {
                unpack4(q.dc[pl], ya.cs, ya.co, uv[pl]);
            } else if (q.regular) {
                uint32_t d[NW + 1];
#pragma unroll
                for (int m = 0; m < NW; ++m) d[m] = q.dc[pl][m];
                d[NW] = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int off = ((x0 >> 1) + yr.dh[i]) * B - q.tbmin;  // 0 .. 4 NW - 4 B
                    uint32_t wlo = d[0], wmd = d[1], whi = d[2];
#pragma unroll
                    for (int m = 1; m + 1 < NW; ++m)
                        if ((off >> 2) == m) wlo = d[m], wmd = d[m + 1], whi = d[m + 2];
                    uint32_t sd[2];
                    sd[0] = __builtin_amdgcn_alignbyte(wmd, wlo, (uint32_t)off & 3u);
                    sd[B - 1] = B == 1 ? sd[0] : __builtin_amdgcn_alignbyte(whi, wmd, (uint32_t)off & 3u);
                    float x[4];
                    unpack4(sd, ya.cs, ya.co, x);
                    uv[pl][i] = yuv_two_acc(yr.ch[i][0], yr.ch[i][1], yr.ch[i][2], yr.ch[i][3], x[0], x[1], x[2], x[3]);
                }
            } else {  // the frame's first / last columns: taps and coefficients from the table, samples clamped into the row
                const GSrc row = (GSrc)(static_cast<const char *>(second ? tp->src2[1 + pl] : tp->src1[1 + pl]) + (size_t)y * ya.cstride * B);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int hl = ya.hleft[x0 + i];
                    const v4f c = *reinterpret_cast<const v4f *>(ya.hcoef + 4 * (size_t)(x0 + i));
                    float x[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[k] = yuv_cvt<T>(row[min(hl + k, ya.cw - 1)], ya.cs, ya.co);
                    uv[pl][i] = yuv_two_acc(c.x, c.y, c.z, c.w, x[0], x[1], x[2], x[3]);
                }
            }
        }
        // the 12 lookups: LDS first, what lies beyond the staged range fetched behind ONE wave-level branch (see ssim_yuv420_rgb_kernel)
        float lin[3][4];
        uint32_t li[3][4], li_max = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float gg = fmaf(ya.m[3 * c + 2], uv[1][i], fmaf(ya.m[3 * c + 1], uv[0][i], ya.m[3 * c] * yy[i]));
                int k = (int)rintf(fmaf(gg, 32768.0f, 16384.0f)) - yr.lut_lo;
                if (yr.low_zero) k = max(k, 0);
                li[c][i] = (uint32_t)k;
                li_max = max(li_max, (uint32_t)k);
                lin[c][i] = ((LdsF)lut)[min((uint32_t)k, (uint32_t)yr.lut_n - 1u)];
            }
        }
        if (__builtin_amdgcn_ballot_w64(li_max >= (uint32_t)yr.lut_n) != 0) {
            asm volatile("" ::: "memory");
            float gl[3][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int t = min(max((int)li[c][i] + yr.lut_lo, 0), 65536);
                    gl[c][i] = ((GblF)a.lut)[li[c][i] >= (uint32_t)yr.lut_n ? t : 0];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) lin[c][i] = li[c][i] >= (uint32_t)yr.lut_n ? gl[c][i] : lin[c][i];
        }
        const size_t o = (size_t)y * a.w + x0;
#pragma unroll
        for (int c = 0; c < 3; ++c) pyr_put4((GOut)(second ? tp->rgb2[c] : tp->rgb1[c]), o, true, 4, lin[c][0], lin[c][1], lin[c][2], lin[c][3]);
    }
}

// The YUV pre-stage alone (vszip_to_rgbs_linear): one thread per output sample, everything from global memory
// (16 chroma taps per plane through the caches) — the mixed-format fallback and the tests' view of the conversion.
template <typename T>
__global__ __launch_bounds__(256) void yuv_to_rgbs_kernel(const void *s0, const void *s1, const void *s2, float *d0, float *d1, float *d2, int sstride, int dstride, int w, int h,
                                                          const float *lut, const YuvArgs ya) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const T *src[3] = {static_cast<const T *>(s0), static_cast<const T *>(s1), static_cast<const T *>(s2)};
    int hl = x, vl = y;
    float hc[4] = {1.0f, 0.0f, 0.0f, 0.0f}, vc[4] = {1.0f, 0.0f, 0.0f, 0.0f};
    if (ya.hleft) {
        hl = ya.hleft[x];
        for (int k = 0; k < 4; ++k) hc[k] = ya.hcoef[4 * (size_t)x + k];
    }
    if (ya.vleft) {
        vl = ya.vleft[y];
        for (int k = 0; k < 4; ++k) vc[k] = ya.vcoef[4 * (size_t)y + k];
    }
    float uv[2];
    for (int pl = 0; pl < 2; ++pl) {
        float rows[4];
        const int nrows = ya.vleft ? 4 : 1;
        for (int j = 0; j < nrows; ++j) {
            const T *row = src[1 + pl] + (size_t)min(vl + j, ya.ch - 1) * ya.cstride;
            if (ya.hleft) {
                float t[4];
                for (int k = 0; k < 4; ++k) t[k] = yuv_cvt<T>(row[min(hl + k, ya.cw - 1)], ya.cs, ya.co);
                rows[j] = yuv_two_acc(hc[0], hc[1], hc[2], hc[3], t[0], t[1], t[2], t[3]);
            } else {
                rows[j] = yuv_cvt<T>(row[hl], ya.cs, ya.co);
            }
        }
        uv[pl] = ya.vleft ? yuv_two_acc(vc[0], vc[1], vc[2], vc[3], rows[0], rows[1], rows[2], rows[3]) : rows[0];
    }
    const float yy = yuv_cvt<T>(src[0][(size_t)y * sstride + x], ya.ys, ya.yo);
    float *dst[3] = {d0, d1, d2};
    for (int c = 0; c < 3; ++c) {
        const float g = fmaf(ya.m[3 * c + 2], uv[1], fmaf(ya.m[3 * c + 1], uv[0], ya.m[3 * c] * yy));
        dst[c][(size_t)y * dstride + x] = yuv_transfer(g, lut, ya.linearize);
    }
}

// ssim_yuv420_rgb_kernel's interior launch treats EVERY column and row of the whole tiles as if it followed the tables' period; the few that
// do not (the frame's first and last columns / rows: [0, xr0), [xr1, wf), [0, yr0), [yr1, hf)) get wrong values there and the right ones
// here, one thread a sample like yuv_to_rgbs_kernel, behind it in stream order. blockIdx.y = frame of the launch (pair * 2 + which).
template <typename T>
__global__ __launch_bounds__(256) void ssim_yuv420_fixup_kernel(const PyrArgs a, const YuvArgs ya, const YuvLds yl, int wf, int hf) {
    const int ncol = yl.xr0 + (wf - yl.xr1), nrow = yl.yr0 + (hf - yl.yr1);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    int x, y;
    if (idx < ncol * hf) {  // the column strips, column-major inside a row
        y = idx / ncol;
        const int c = idx - y * ncol;
        x = c < yl.xr0 ? c : yl.xr1 + (c - yl.xr0);
    } else if (idx < ncol * hf + nrow * wf) {
        const int e = idx - ncol * hf, r = e / wf;
        x = e - r * wf;
        y = r < yl.yr0 ? r : yl.yr1 + (r - yl.yr0);
    } else {
        return;
    }
    const PyrPair *__restrict__ tp = a.tab + (blockIdx.y >> 1);
    const bool second = blockIdx.y & 1;
    const int hl = ya.hleft[x], vl = ya.vleft[y];
    float hc[4], vc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hc[k] = ya.hcoef[4 * (size_t)x + k], vc[k] = ya.vcoef[4 * (size_t)y + k];
    float uv[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const T *src = static_cast<const T *>(second ? tp->src2[1 + pl] : tp->src1[1 + pl]);
        float rows[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T *row = src + (size_t)min(vl + j, ya.ch - 1) * ya.cstride;
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = yuv_cvt<T>(row[min(hl + k, ya.cw - 1)], ya.cs, ya.co);
            rows[j] = yuv_two_acc(hc[0], hc[1], hc[2], hc[3], t[0], t[1], t[2], t[3]);
        }
        uv[pl] = yuv_two_acc(vc[0], vc[1], vc[2], vc[3], rows[0], rows[1], rows[2], rows[3]);
    }
    const T *ysrc = static_cast<const T *>(second ? tp->src2[0] : tp->src1[0]);
    const float yy = yuv_cvt<T>(ysrc[(size_t)y * a.sstride + x], ya.ys, ya.yo);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = fmaf(ya.m[3 * c + 2], uv[1], fmaf(ya.m[3 * c + 1], uv[0], ya.m[3 * c] * yy));
        float *dst = second ? tp->rgb2[c] : tp->rgb1[c];
        dst[(size_t)y * a.w + x] = yuv_transfer(g, a.lut, 1);
    }
}

struct MapsArgs {
    const PairPtrs *tab;  // [npairs] for this scale
    int plane[3];         // XYB plane index of the active plane slots
    int flags[3];         // bit0: ssim map, bit1: edge map
    int slot[3];          // partial-table slot (scale * 3 + plane)
    int nactive;
    int stride, w, h;
    int tiles_x, tiles_y;
    int tpb;          // tiles (along x) per block
    double *partial;  // [pair][slot][tile][6]
    int max_tiles;
    int gx, gy, nblk;  // ssim_maps_ts_kernel: the logical (x, y, plane-pair) grid, launched as one dimension of ceil(nblk / 8) * 8 blocks
};

// the asymmetric mirror of blur (:254,260,357,364): reflect-101 at the start,
// mirror about the CURRENT index at the end
__device__ __forceinline__ int tap_index(int k, int i, int n) {
    const int dist_from_end = n - 1 - i;
    if (k < HALO) return (i < HALO - k) ? min(HALO - k - i, n - 1) : (i - HALO + k);
    return (dist_from_end < k - HALO) ? (i - min(k - HALO - dist_from_end, i)) : (i - HALO + k);
}

// a / b in f64 for the two quotients of maps_pixel, whose denominators are finite and far from the exponent limits (>= 1, or a
// variance term + 0.0009): v_rcp_f64 refined by ONE Newton step and one residual correction of the quotient, 6 instructions against
// the 12 of the v_div_scale / v_div_fmas / v_div_fixup sequence, which carries the range handling these operands never need. The maps
// kernel is issue bound and its f64 instructions (half rate) are more than half of its VALU cycles. tools/div_probe.hip compares the
// forms with the IEEE quotient on 16.7 M operand pairs of the kernel's ranges: one Newton step and two both give the IEEE quotient on
// EVERY pair (round 3 first used two), none leaves up to 18 units in the last place. -DVSZIP_SSIM_IEEE_DIV restores the IEEE sequence.
__device__ __forceinline__ double maps_div(double a, double b) {
#ifdef VSZIP_SSIM_IEEE_DIV
    return a / b;
#else
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
#endif
}

// ssimMap (:511-523) and edgeMap (:585-603) of one pixel, accumulated in f64
__device__ __forceinline__ void maps_pixel(float mu1, float mu2, float b12, float bsq, float v1, float v2, bool do_ssim, bool do_edge, double acc[6]) {
#ifdef VSZIP_SSIM_TIMING_NOF64  // timing only
    acc[0] += (double)(mu1 + mu2 + b12 + bsq + v1 + v2);
    return;
#endif
    if (do_ssim) {
        const float m11 = mu1 * mu1, m22 = mu2 * mu2, m12 = mu1 * mu2, md = mu1 - mu2;
        const double num_m = (double)fmaf(md, -md, 1.0f);
        const double num_s = (double)fmaf(b12 - m12, 2.0f, 0.0009f);
        const double denom_s = (double)(bsq - 2.0f * b12 - m11 - m22 + 0.0009f);
        const double d1 = fmax(1.0 - maps_div(num_m * num_s, denom_s), 0.0);
        double t = d1 * d1;
        acc[0] += d1;
        acc[1] += t * t;
    }
    if (do_edge) {
        const double n2 = (double)fabsf(v2 - mu2), n1 = (double)fabsf(v1 - mu1);
        const double d1 = maps_div(1.0 + n2, 1.0 + n1) - 1.0;
        const double art = fmax(d1, 0.0), det = fmax(-d1, 0.0);
        double t = art * art;
        acc[2] += art;
        acc[3] += t * t;
        t = det * det;
        acc[4] += det;
        acc[5] += t * t;
    }
}

// ---- the maps kernels ------------------------------------------------------------------------------------------------------------------
// skip_table (ssimulacra2.zig:22-37) fixes per (scale, plane) whether ssimMap runs and whether edgeMap runs (edgeMap computes artifact AND detail
// loss whenever either counts, :124): three term sets - S+E (scale 0's Y plane: more than half of the maps time at 4K), S, E - and one kernel
// instance per set, ssim_maps_ts_kernel<SSIM, EDGE>.
//
// Tile geometry (round 5): 56 x 32 outputs from a 64 x 40 staged tile of both frames. The kernels are issue bound, so what counts is how many
// of a block's 4 x 64 lanes every instruction occupies:
//   * vertical pass (blurV :308-330): thread = one staged column (64) x one group of 8 output rows (4) = all 256 threads; the 16 samples of a
//     column window sit in registers, every output accumulates its nine taps in tap order with fmaf. Round 4's 32 x 32 tile staged 40 columns:
//     160 threads, the third wave half empty and the fourth idle, i.e. one of a CU's four SIMDs did no vertical work at all.
//   * horizontal pass (blurH :247-306, unfused acc + k * s): thread = one row (32) x one strip of 7 adjacent outputs (8) = 256 threads, 15 LDS
//     reads for 7 outputs (12 for 4 before); the per-pixel f64 terms (maps_pixel) follow in the same thread.
// The means (mu1, mu2) are blurred first and stored as the .zw halves of the vertical results, then p, q are overwritten by p q and (p + q)^2
// and (b12, bsq) follow into the .xy halves: half the live registers of doing all four at once. The E set blurs the two means only; the S set
// skips edgeMap's division and its four f64 sums.
// LDS: s1 / s2 in rows of 64 floats with the column XOR-swizzled by the row (c ^ 8 (r & 7): the centre samples edgeMap reads are 8 rows x 8 strips a
// wave, which a plain pitch of 64 puts in 8 banks; in the vertical pass the swizzle is one of 8 per-thread constants, the row a compile-time offset),
// the vertical results in rows of 65 x 16 bytes: 53 760 bytes a block = 42 allocation granules of 1 280 bytes = three blocks a CU, to the byte.
constexpr int kVP = IW + 1;                   // row pitch of the vertical results, v4f slots
constexpr int kVtSlots = TH * kVP;
constexpr int kMapsLds = 2 * IH * IW * 4 + kVtSlots * 16;
static_assert(IW == 64 && TH == 32 && TW % 8 == 0 && IW * (TH / 8) == 256 && TH * (TW / 7) == 256, "the maps kernels' thread maps are written for a 64 x 40 staged tile, 56 x 32 outputs");
static_assert(kMapsLds <= 42 * 1280, "three blocks a CU: 160 KiB of LDS in granules of 1 280 bytes");
__device__ __forceinline__ int s_at(int r, int c) { return r * IW + (c ^ ((r & 7) << 3)); }
__device__ __forceinline__ int vt_at(int r, int c) { return r * kVP + c; }

constexpr int kMapsNS = IH * IW / 256;  // staged samples per thread and frame (10)

// the 64 x 40 input tile of both frames: all 20 loads of a thread in flight before anything waits on them (global address space: plane
// pointers read from the pair table are generic to the compiler - flat loads, each followed by a full s_waitcnt in the rolled loop).
// Top / left edge: the blur's padding there is reflect-101 (:254,357) - row / column -j is row / column j whatever the output index - so the
// padded tile is a fixed function of the plane. CLAMP (border tiles): coordinates past the plane's end are clamped; such samples reach only
// outputs that the border path recomputes or masks (the END of a line mirrors about the CURRENT output index, :260,364, which no padding expresses).
template <bool CLAMP>
__device__ __forceinline__ void ssim_maps_fetch(const MapsArgs &a, const float *im1, const float *im2, int x0, int y0, float v1[kMapsNS], float v2[kMapsNS]) {
    const float VSZIP_GLOBAL *g1 = (const float VSZIP_GLOBAL *)im1, *g2 = (const float VSZIP_GLOBAL *)im2;
    const int tid = threadIdx.x;
    const int c = tid & (IW - 1), rb = tid >> 6;
    int gx = abs(x0 - HALO + c);
    if (CLAMP) gx = min(gx, a.w - 1);
#pragma unroll
    for (int k = 0; k < kMapsNS; ++k) {
        int gy = abs(y0 - HALO + rb + 4 * k);
        if (CLAMP) gy = min(gy, a.h - 1);
        // a 32-bit BYTE offset (planes are below 4 GiB): "scalar base + 32-bit lane offset" loads instead of a 64-bit address pair per load
        const uint32_t o = (uint32_t)(gy * a.stride + gx) * 4u;
#ifdef VSZIP_SSIM_TIMING_NOFETCH  // timing only
        v1[k] = (float)(o & 255) * 0.001f;
        v2[k] = (float)(o & 127) * 0.002f;
        (void)g1;
        (void)g2;
#else
        v1[k] = *reinterpret_cast<const float VSZIP_GLOBAL *>(reinterpret_cast<const char VSZIP_GLOBAL *>(g1) + o);
        v2[k] = *reinterpret_cast<const float VSZIP_GLOBAL *>(reinterpret_cast<const char VSZIP_GLOBAL *>(g2) + o);
#endif
    }
}
__device__ __forceinline__ void ssim_maps_park(const float v1[kMapsNS], const float v2[kMapsNS], float *s1, float *s2) {
    const int tid = threadIdx.x;
    const int wv = tid >> 6, c = tid & (IW - 1);
    const int base[2] = {wv * IW + (c ^ (wv << 3)), wv * IW + (c ^ ((wv + 4) << 3))};  // rows wv + 4 k: (row & 7) = wv + 4 (k & 1)
#pragma unroll
    for (int k = 0; k < kMapsNS; ++k) {
        s1[base[k & 1] + 4 * k * IW] = v1[k];
        s2[base[k & 1] + 4 * k * IW] = v2[k];
    }
}

// vertical pass of a staged tile: thread (column c, rows 8 g .. 8 g + 7). FUSED: the reference's vector body (fmaf, :318); else its scalar
// tail (acc + k * s, :326) - columns at and beyond w - w % 8.
template <bool SSIM, bool FUSED>
__device__ __forceinline__ void ssim_maps_vertical(const float *s1, const float *s2, v4f *vt) {
    const int tid = threadIdx.x;
    constexpr int VR = 8;
    const int c = tid & (IW - 1), r0 = (tid >> 6) * VR;
    int sb[8];  // rows r0 + j, r0 a multiple of 8: the swizzle of row j is that of j & 7
#pragma unroll
    for (int m = 0; m < 8; ++m) sb[m] = r0 * IW + (c ^ (m << 3));
    float *vo = reinterpret_cast<float *>(vt + vt_at(r0, c));
    float p[VR + 8], q[VR + 8];
#pragma unroll
    for (int j = 0; j < VR + 8; ++j) {
        p[j] = s1[sb[j & 7] + j * IW];
        q[j] = s2[sb[j & 7] + j * IW];
    }
#pragma unroll
    for (int o = 0; o < VR; ++o) {
        v2f m = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const v2f kk = {c_kernel[k], c_kernel[k]}, pq = {p[o + k], q[o + k]};
            m = FUSED ? __builtin_elementwise_fma(kk, pq, m) : m + kk * pq;
        }
        *reinterpret_cast<v2f *>(vo + o * kVP * 4 + 2) = m;
    }
    if constexpr (SSIM) {
#pragma unroll
        for (int j = 0; j < VR + 8; ++j) {
            const float pp = p[j], qq = q[j], sum = pp + qq;
            p[j] = pp * qq;
            q[j] = sum * sum;
        }
#pragma unroll
        for (int o = 0; o < VR; ++o) {
            v2f ms = {0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const v2f kk = {c_kernel[k], c_kernel[k]}, pq = {p[o + k], q[o + k]};
                ms = FUSED ? __builtin_elementwise_fma(kk, pq, ms) : ms + kk * pq;
            }
            *reinterpret_cast<v2f *>(vo + o * kVP * 4) = ms;
        }
    }
}

constexpr int HC = 7;  // adjacent outputs per thread in the horizontal pass: 32 rows x 8 strips

// horizontal pass of the thread's 7 outputs from the vertical results
template <bool SSIM>
__device__ __forceinline__ void ssim_maps_horizontal(const v4f *vt, v2f mu[HC], v2f bs[HC]) {
    const int tid = threadIdx.x;
    const int r = tid >> 3, xs = (tid & 7) * HC;
    const float *vi = reinterpret_cast<const float *>(vt + vt_at(r, xs));
    {
        v2f t[HC + 8];
#pragma unroll
        for (int j = 0; j < HC + 8; ++j) t[j] = *reinterpret_cast<const v2f *>(vi + 4 * j + 2);
#pragma unroll
        for (int o = 0; o < HC; ++o) {
            v2f acc = {0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = acc + v2f{c_kernel[k], c_kernel[k]} * t[o + k];
            mu[o] = acc;
        }
    }
    if constexpr (SSIM) {
        // (a scheduling fence: without it the compiler fuses the two halves' reads into 16-byte reads and holds all 60 values at once - 30 registers
        // more than the kernel has at three blocks a CU, and a spilled value reloads through vmcnt, in line behind the next tile's 20 global loads)
        asm volatile("" ::: "memory");
        v2f t[HC + 8];
#pragma unroll
        for (int j = 0; j < HC + 8; ++j) t[j] = *reinterpret_cast<const v2f *>(vi + 4 * j);
#pragma unroll
        for (int o = 0; o < HC; ++o) {
            v2f acc = {0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = acc + v2f{c_kernel[k], c_kernel[k]} * t[o + k];
            bs[o] = acc;
        }
    } else {
#pragma unroll
        for (int o = 0; o < HC; ++o) bs[o] = v2f{0.0f, 0.0f};
    }
}

// the second half of a tile clear of the plane's bottom / right edge (its vertical pass is done, no barrier yet)
template <bool SSIM, bool EDGE>
__device__ __forceinline__ void ssim_maps_tile_ts_h(const float *s1, const float *s2, const v4f *vt, double acc[6]) {
    const int tid = threadIdx.x;
    // the centre samples edgeMap needs, read BEFORE the barrier: after it nobody reads s1 / s2 any more, so the next tile may be parked into them
    // without a barrier at the end of this one (vt is protected by the barrier that follows the park)
    float e1[HC], e2[HC];
    if constexpr (EDGE) {
        const int rr = (tid >> 3) + HALO, cb = (tid & 7) * HC + HALO;
#pragma unroll
        for (int o = 0; o < HC; ++o) {
            e1[o] = s1[s_at(rr, cb + o)];
            e2[o] = s2[s_at(rr, cb + o)];
        }
    }
    __syncthreads();
    v2f mu[HC], bs[HC];
    ssim_maps_horizontal<SSIM>(vt, mu, bs);
#pragma unroll
    for (int o = 0; o < HC; ++o) maps_pixel(mu[o].x, mu[o].y, bs[o].x, bs[o].y, EDGE ? e1[o] : 0.0f, EDGE ? e2[o] : 0.0f, SSIM, EDGE, acc);
}

// A tile on the plane's bottom / right edge (planes of at least 16 x 16). Round 4 sent these to the per-pixel generic path (4.5 % of a 4K
// plane's tiles at several times an interior tile's cost). The end-of-line rule (tap_index: taps past the end mirror about the CURRENT output
// index) changes the taps of the LAST FOUR outputs of a line only - an output i with n - 1 - i >= 4 has all its taps inside the line - so the
// tile runs the blocked passes on a staged tile whose coordinates are clamped at the plane's end, and then
//   * vertical: a thread recomputes those of its outputs that lie in the plane's last four rows with the mirrored taps (gathered from LDS),
//     and columns at and beyond w - w % 8 take the unfused form (the reference's scalar tail, :326) - per thread, a column is one thread's;
//   * horizontal: outputs in the plane's last four columns are recomputed with the mirrored taps (gathered from the vertical results),
//     outputs beyond the plane are dropped.
// (Not inlined, like the small-plane path below: inlined, their live ranges cost the interior path 29 scratch reloads a tile. They run on a few per
// cent of the tiles; their sums come back by value.)
struct MapsSums {
    double v[6];
};
struct MapsGeom {
    int stride, w, h;
};
template <bool SSIM, bool EDGE>
__device__ __attribute__((noinline)) MapsSums ssim_maps_tile_border(MapsGeom g, const float *im1, const float *im2, int x0, int y0, float *s1, float *s2, v4f *vt) {
    const int tid = threadIdx.x;
    const int w = g.w, h = g.h;
    MapsArgs a;  // (the fetch reads stride, w and h only)
    a.stride = g.stride;
    a.w = w;
    a.h = h;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    {
        float v1[kMapsNS], v2[kMapsNS];
        ssim_maps_fetch<true>(a, im1, im2, x0, y0, v1, v2);
        ssim_maps_park(v1, v2, s1, s2);
    }
    __syncthreads();
    {
        const int c = tid & (IW - 1), r0 = (tid >> 6) * 8;
        const bool fused = x0 - HALO + c < w - (w % kVecW);
        if (fused)
            ssim_maps_vertical<SSIM, true>(s1, s2, vt);
        else
            ssim_maps_vertical<SSIM, false>(s1, s2, vt);
        if (y0 + r0 + 7 >= h - HALO) {  // some of this thread's rows are among the plane's last four
#pragma unroll 1
            for (int o = 0; o < 8; ++o) {
                const int y = y0 + r0 + o;
                if (y < h - HALO || y >= h) continue;
                v2f m = {0.0f, 0.0f}, ms = {0.0f, 0.0f};
#pragma unroll 1
                for (int k = 0; k < 9; ++k) {
                    const int rr = tap_index(k, y, h) - (y0 - HALO);  // (a reflected tap of the first rows: the same sample sits at the positive coordinate)
                    const float p = s1[s_at(rr, c)], q = s2[s_at(rr, c)], sum = p + q;
                    const v2f kk = {c_kernel[k], c_kernel[k]}, pq = {p, q}, prod = {p * q, sum * sum};
                    if (fused) {
                        m = __builtin_elementwise_fma(kk, pq, m);
                        if (SSIM) ms = __builtin_elementwise_fma(kk, prod, ms);
                    } else {
                        m = m + kk * pq;
                        if (SSIM) ms = ms + kk * prod;
                    }
                }
                float *fix = reinterpret_cast<float *>(vt + vt_at(r0 + o, c));  // (through the same float-based v2f stores as ssim_maps_vertical: one access type for these bytes)
                *reinterpret_cast<v2f *>(fix + 2) = m;
                if (SSIM) *reinterpret_cast<v2f *>(fix) = ms;
            }
        }
    }
    __syncthreads();
    {
        const int r = tid >> 3, xs = (tid & 7) * HC;
        const int y = y0 + r;
        if (y < h) {
            {
                v2f mu[HC], bs[HC];
                ssim_maps_horizontal<SSIM>(vt, mu, bs);
#pragma unroll
                for (int o = 0; o < HC; ++o) {
                    if (x0 + xs + o >= w - HALO) continue;  // the plane's last four columns: below; beyond the plane: nothing
                    const int cc = s_at(r + HALO, xs + o + HALO);
                    maps_pixel(mu[o].x, mu[o].y, bs[o].x, bs[o].y, EDGE ? s1[cc] : 0.0f, EDGE ? s2[cc] : 0.0f, SSIM, EDGE, acc);
                }
            }
#pragma unroll 1
            for (int x = max(x0 + xs, w - HALO); x < min(x0 + xs + HC, w); ++x) {
                v2f m2 = {0.0f, 0.0f}, b2 = {0.0f, 0.0f};
#pragma unroll 1
                for (int k = 0; k < 9; ++k) {
                    const float *t = reinterpret_cast<const float *>(vt + vt_at(r, tap_index(k, x, w) - (x0 - HALO)));
                    const v2f kk = {c_kernel[k], c_kernel[k]};
                    m2 = m2 + kk * *reinterpret_cast<const v2f *>(t + 2);
                    if (SSIM) b2 = b2 + kk * *reinterpret_cast<const v2f *>(t);
                }
                const int cc = s_at(r + HALO, x - x0 + HALO);
                maps_pixel(m2.x, m2.y, b2.x, b2.y, EDGE ? s1[cc] : 0.0f, EDGE ? s2[cc] : 0.0f, SSIM, EDGE, acc);
            }
        }
    }
    MapsSums out;
#pragma unroll
    for (int q = 0; q < 6; ++q) out.v[q] = acc[q];
    return out;
}

// Planes smaller than 16 x 16 (the last scales of small clips): per-pixel, every tap through tap_index. (blurV :308-330, blurH :247-306)
__device__ __attribute__((noinline)) MapsSums ssim_maps_tile_small(MapsGeom g, const float *im1, const float *im2, bool do_ssim, bool do_edge, int x0, int y0, float *s1, float *s2, v4f *vt) {
    const int w = g.w, h = g.h;
    struct { int stride; } a = {g.stride};
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const int cx0 = max(x0 - HALO, 0), cy0 = max(y0 - HALO, 0);  // real coords of LDS (0,0)
    const int cw = min(x0 + TW + HALO, w) - cx0, ch = min(y0 + TH + HALO, h) - cy0;
    const int tid = threadIdx.x;
    for (int i = tid; i < ch * cw; i += 256) {
        const int r = i / cw, c = i - r * cw;
        const size_t o = (size_t)(cy0 + r) * a.stride + (cx0 + c);
        s1[s_at(r, c)] = im1[o];
        s2[s_at(r, c)] = im2[o];
    }
    __syncthreads();
    const int th = min(TH, h - y0);
    const int wv = w - (w % kVecW);
    for (int i = tid; i < th * cw; i += 256) {
        const int r = i / cw, c = i - r * cw;
        const int y = y0 + r;
        const bool fused = (cx0 + c) < wv;
        v2f m = {0.0f, 0.0f}, ms = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int rr = tap_index(k, y, h) - cy0;
            const float p = s1[s_at(rr, c)], q = s2[s_at(rr, c)];
            const v2f kk = {c_kernel[k], c_kernel[k]};
            const v2f pq = {p, q};
            const float sum = p + q;
            const v2f a2 = {p, sum}, b2 = {q, sum};
            if (fused) {
                m = __builtin_elementwise_fma(kk, pq, m);
                if (do_ssim) ms = __builtin_elementwise_fma(kk, a2 * b2, ms);
            } else {
                m = m + kk * pq;
                if (do_ssim) ms = ms + kk * (a2 * b2);
            }
        }
        vt[vt_at(r, c)] = v4f{ms.x, ms.y, m.x, m.y};
    }
    __syncthreads();
    const int tw = min(TW, w - x0);
    for (int i = tid; i < th * tw; i += 256) {
        const int r = i / tw, c = i - r * tw;
        const int x = x0 + c;
        v2f mu = {0.0f, 0.0f}, bs = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int cc = tap_index(k, x, w) - cx0;
            const v2f kk = {c_kernel[k], c_kernel[k]};
            const v4f t = vt[vt_at(r, cc)];
            mu = mu + kk * v2f{t.z, t.w};
            if (do_ssim) bs = bs + kk * v2f{t.x, t.y};
        }
        maps_pixel(mu.x, mu.y, bs.x, bs.y, s1[s_at(y0 + r - cy0, x - cx0)], s2[s_at(y0 + r - cy0, x - cx0)], do_ssim, do_edge, acc);
    }
    MapsSums out;
#pragma unroll
    for (int q = 0; q < 6; ++q) out.v[q] = acc[q];
    return out;
}

// Three blocks a CU (12 waves, 3 a SIMD, at most 168 VGPRs): bounded by the LDS above.
#ifndef VSZIP_SSIM_TS_BPC
#define VSZIP_SSIM_TS_BPC 3
#endif
template <bool SSIM, bool EDGE>
__global__ __launch_bounds__(256, VSZIP_SSIM_TS_BPC) void ssim_maps_ts_kernel(const MapsArgs a) {
    __shared__ float s1[IH * IW], s2[IH * IW];
    __shared__ v4f vt[kVtSlots];
    // XCD-aware order: workgroup i runs on XCD i % 8, so the logical blocks are dealt out in eight contiguous chunks - an XCD walks a band of tile
    // rows of one plane, and the 8-row / 8-column halo a tile shares with its neighbours is in that XCD's L2 (with the plain 3-D grid every
    // neighbour sat on another XCD and the halo came from memory again)
    const int chunk = (a.nblk + 7) >> 3;
    const int lb = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (lb >= a.nblk) return;
    // (the divisions run on the vector unit; readfirstlane tells the compiler what it cannot see - these are the same in every lane - so that the
    // plane pointers are scalar loads and the tile fetch is "scalar base + 32-bit lane offset")
    const int bx = __builtin_amdgcn_readfirstlane(lb % a.gx), by = __builtin_amdgcn_readfirstlane((lb / a.gx) % a.gy), bz = __builtin_amdgcn_readfirstlane(lb / (a.gx * a.gy));
    const int ps = __builtin_amdgcn_readfirstlane(bz % a.nactive), pair = __builtin_amdgcn_readfirstlane(bz / a.nactive);
    const PairPtrs &pp = a.tab[pair];
    const int pl = __builtin_amdgcn_readfirstlane(a.plane[ps]);
    auto uniform_ptr = [](const float *p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const float *>(((uint64_t)hi << 32) | lo);
    };
    const float *im1 = uniform_ptr(pp.xyb1[pl]), *im2 = uniform_ptr(pp.xyb2[pl]);
    const int w = a.w, h = a.h;
    const int y0 = by * TH;
    const int tid = threadIdx.x;
    const bool small = w < 16 || h < 16;
    // "interior": clear of the bottom / right edge and of the unfused column tail; top / left edge tiles are staged with their reflect-101 padding
    auto is_interior = [&](int x0) { return !small && x0 + TW + HALO <= w - (w % kVecW) && y0 + TH + HALO <= h; };
    const int tx0 = bx * a.tpb, tx1 = min(tx0 + a.tpb, a.tiles_x);
    float v1[kMapsNS], v2[kMapsNS];
    bool fetched = false;
    // The block's tiles accumulate into ONE set of per-thread sums, reduced once (fixed order: lanes, then the four waves). The block's sum goes
    // into the slot of its first tile, the other tiles' slots hold exact zeros (the final kernel folds every slot).
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int tx = tx0; tx < tx1; ++tx) {
        const int x0 = tx * TW;
        if (is_interior(x0)) {
            if (!fetched) ssim_maps_fetch<false>(a, im1, im2, x0, y0, v1, v2);
            ssim_maps_park(v1, v2, s1, s2);
            __syncthreads();
#ifdef VSZIP_SSIM_FETCH_EARLY  // (A/B: the next tile requested before the vertical pass)
            fetched = tx + 1 < tx1 && is_interior(x0 + TW);
            if (fetched) ssim_maps_fetch<false>(a, im1, im2, x0 + TW, y0, v1, v2);
            ssim_maps_vertical<SSIM, true>(s1, s2, vt);
#else
            ssim_maps_vertical<SSIM, true>(s1, s2, vt);
            // the next tile's samples are requested here, between the passes: they arrive under the horizontal pass and the f64 terms, and their 20
            // registers are not live during the vertical pass (whose column windows need 40)
            fetched = tx + 1 < tx1 && is_interior(x0 + TW);
            if (fetched) ssim_maps_fetch<false>(a, im1, im2, x0 + TW, y0, v1, v2);
#endif
            ssim_maps_tile_ts_h<SSIM, EDGE>(s1, s2, vt, acc);
            // (no barrier here: see the centre samples in ssim_maps_tile_ts_h; a tile of another kind that follows stages behind a barrier of its own)
            if (tx + 1 < tx1 && !is_interior(x0 + TW)) __syncthreads();
        } else {
            fetched = false;
            const MapsGeom geom = {a.stride, w, h};
            const MapsSums part = small ? ssim_maps_tile_small(geom, im1, im2, SSIM, EDGE, x0, y0, s1, s2, vt) : ssim_maps_tile_border<SSIM, EDGE>(geom, im1, im2, x0, y0, s1, s2, vt);
#pragma unroll
            for (int q = 0; q < 6; ++q) acc[q] += part.v[q];
            __syncthreads();  // these tiles read s1 / s2 / vt in their last phase: they are rewritten by the next tile
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q)
        if ((q < 2 && SSIM) || (q >= 2 && EDGE)) acc[q] = wave_reduce_sum(acc[q]);
    __syncthreads();
    double(*red)[6] = reinterpret_cast<double(*)[6]>(vt);  // (the tiles are done with it)
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) red[tid >> 6][q] = acc[q];
    }
    __syncthreads();
    if (tid < 6) {
        const double v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        for (int tx = tx0; tx < tx1; ++tx) {
            const int tile = by * a.tiles_x + tx;
            a.partial[(((size_t)pair * 18 + a.slot[ps]) * a.max_tiles + tile) * 6 + tid] = tx == tx0 ? v : 0.0;
        }
    }
}


struct FinalArgs {
    const double *partial;  // [pair][slot][tile][6]
    double *avg;  // [pair][18 slots][6]: ssim avg, ssim 4th-root, art avg, art 4th-root, det avg, det 4th-root
    int ntiles[18];
    double one_per_pixels[18];
    int max_tiles;
};

constexpr int kFinThreads = 256, kFinUnroll = 4;

// One block per (slot, pair). The tile partials are folded in a fixed order — thread t takes tiles
// t, t+256, ... in four interleaved chains (so four loads are in flight per thread instead of one
// dependent chain over n/64 round trips), then lanes, then the four waves — so the result is
// reproducible run to run.
__global__ __launch_bounds__(kFinThreads) void ssim_final_kernel(const FinalArgs a) {
    __shared__ double sh[kFinThreads / 64][6];
    const int slot = blockIdx.x, pair = blockIdx.y;
    const int n = a.ntiles[slot];
    const int tid = threadIdx.x;
    const double *base = a.partial + ((size_t)pair * 18 + slot) * a.max_tiles * 6;
    double acc[kFinUnroll][6];
#pragma unroll
    for (int u = 0; u < kFinUnroll; ++u)
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[u][q] = 0.0;
    for (int t0 = tid; t0 < n; t0 += kFinThreads * kFinUnroll) {
        double v[kFinUnroll][6];
#pragma unroll
        for (int u = 0; u < kFinUnroll; ++u) {
            const int t = t0 + u * kFinThreads;
            const double *p = base + (size_t)min(t, n - 1) * 6;
            const bool live = t < n;  // the clamped index keeps the load unconditional
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const double x = p[q];
                v[u][q] = live ? x : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < kFinUnroll; ++u)
#pragma unroll
            for (int q = 0; q < 6; ++q) acc[u][q] += v[u][q];
    }
    double s[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) s[q] = wave_reduce_sum((acc[0][q] + acc[1][q]) + (acc[2][q] + acc[3][q]));
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) sh[tid >> 6][q] = s[q];
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) s[q] = (sh[0][q] + sh[1][q]) + (sh[2][q] + sh[3][q]);
        const double opp = a.one_per_pixels[slot];
        double *o = a.avg + ((size_t)pair * 18 + slot) * 6;
        o[0] = opp * s[0];
        o[1] = sqrt(sqrt(opp * s[1]));
        o[2] = opp * s[2];
        o[3] = sqrt(sqrt(opp * s[3]));
        o[4] = opp * s[4];
        o[5] = sqrt(sqrt(opp * s[5]));
    }
}

const double kWeight[108] = {
    0.0, 0.0007376606707406586, 0.0, 0.0, 0.0007793481682867309, 0.0, 0.0, 0.0004371155730107379, 0.0,
    1.1041726426657346, 0.00066284834129271, 0.00015231632783718752, 0.0, 0.0016406437456599754, 0.0,
    1.8422455520539298, 11.441172603757666, 0.0, 0.0007989109436015163, 0.000176816438078653, 0.0,
    1.8787594979546387, 10.94906990605142, 0.0, 0.0007289346991508072, 0.9677937080626833, 0.0,
    0.00014003424285435884, 0.9981766977854967, 0.00031949755934435053, 0.0004550992113792063, 0.0, 0.0,
    0.0013648766163243398, 0.0, 0.0, 0.0, 0.0, 0.0, 7.466890328078848, 0.0, 17.445833984131262,
    0.0006235601634041466, 0.0, 0.0, 6.683678146179332, 0.00037724407979611296, 1.027889937768264,
    225.20515300849274, 0.0, 0.0, 19.213238186143016, 0.0011401524586618361, 0.001237755635509985,
    176.39317598450694, 0.0, 0.0, 24.43300999870476, 0.28520802612117757, 0.0004485436923833408, 0.0, 0.0,
    0.0, 34.77906344483772, 44.835625328877896, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0008680556573291698,
    0.0, 0.0, 0.0, 0.0, 0.0, 0.0005313191874358747, 0.0, 0.00016533814161379112, 0.0, 0.0, 0.0, 0.0, 0.0,
    0.0004179171803251336, 0.0017290828234722833, 0.0, 0.0020827005846636437, 0.0, 0.0, 8.826982764996862,
    23.19243343998926, 0.0, 95.1080498811086, 0.9863978034400682, 0.9834382792465353, 0.0012286405048278493,
    171.2667255897307, 0.9807858872435379, 0.0, 0.0, 0.0, 0.0005130064588990679, 0.0, 0.00010854057858411537,
};

struct Skip {
    bool ssim, artifact, detail;
    bool all() const { return ssim && artifact && detail; }
};
Skip skip_of(int plane, int scale) {  // ssimulacra2.zig:22-37
    const int base = plane * 36 + scale * 6;
    const double p = 0.01;
    return {kWeight[base + 0] <= p && kWeight[base + 3] <= p, kWeight[base + 1] <= p && kWeight[base + 4] <= p, kWeight[base + 2] <= p && kWeight[base + 5] <= p};
}

// musl cbrtf == Zig's std.math.cbrt(f32): K_D1 = cbrt(K_D0) at the reference's comptime
float cbrtf_musl(float x) {
    uint32_t ui;
    std::memcpy(&ui, &x, 4);
    uint32_t hx = (ui & 0x7fffffffu) / 3 + 709958130u;
    ui = (ui & 0x80000000u) | hx;
    float tf;
    std::memcpy(&tf, &ui, 4);
    double T = tf, r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    return (float)T;
}

XybK make_xyb_consts() {  // ssimulacra2.zig:374-390
    XybK k;
    const float K_D0 = 0.0037930734f, K_M02 = 0.078f, K_M00 = 0.30f, K_M12 = 0.078f, K_M10 = 0.23f, K_M20 = 0.24342269f, K_M21 = 0.20476745f;
    k.m[0] = K_M00;
    k.m[1] = 1.0f - K_M02 - K_M00;
    k.m[2] = K_M02;
    k.m[3] = K_M10;
    k.m[4] = 1.0f - K_M12 - K_M10;
    k.m[5] = K_M12;
    k.m[6] = K_M20;
    k.m[7] = K_M21;
    k.m[8] = 1.0f - K_M20 - K_M21;
    k.bias = K_D0;
    k.kd1 = cbrtf_musl(K_D0);
    return k;
}

double score_of(const double avg[18][6]) {  // ssimulacra2.zig:630-663; slot = scale * 3 + plane
    double s = 0.0;
    int i = 0;
    for (int plane = 0; plane < 3; ++plane)
        for (int sc = 0; sc < 6; ++sc)
            for (int n = 0; n < 2; ++n) {
                const double *a = sc < kScales ? avg[sc * 3 + plane] : nullptr;
                s = std::fma(kWeight[i++], a ? std::fabs(a[0 + n]) : 0.0, s);
                s = std::fma(kWeight[i++], a ? std::fabs(a[2 + n]) : 0.0, s);
                s = std::fma(kWeight[i++], a ? std::fabs(a[4 + n]) : 0.0, s);
            }
    s *= 0.9562382616834844;
    s = (6.248496625763138e-5 * s * s) * s + 2.326765642916932 * s - 0.020884521182843837 * s * s;
    if (s > 0.0)
        s = std::pow(s, 0.6276336467831387) * -10.0 + 100.0;
    else
        s = 100.0;
    return s;
}

}  // namespace

// zimg's sRGB EOTF (constants of its colorspace/gamma.cpp) evaluated in f64, rounded to f32 — the entries of
// its approximate-gamma table (VapourSynth's resize default approximate_gamma=1): 2^16 + 1 samples over
// [-0.5, 1.5]. Restated from the published algorithm; pinned by the reference's SSIMULACRA2 goldens
// (oracle/vs_host.py, tests/test_oracle_vs_host.py).
static float srgb_eotf_f32(float xf) {
    const double A = 1.055010718947587, B = 0.003041282560128;
    const double x = std::max((double)xf, 0.0);  // zimg's transfer functions clamp negative input first (round 3: the YUV goldens show it)
    return (float)(x < 12.92 * B ? x / 12.92 : std::pow((x + (A - 1.0)) / A, 2.4));
}
static const std::vector<float> &srgb_table() {
    static const std::vector<float> t = [] {
        std::vector<float> v(65537);
        for (int i = 0; i < 65537; ++i) v[i] = srgb_eotf_f32((float)i / 65536.0f * 2.0f - 0.5f);
        return v;
    }();
    return t;
}
// the table's leading zeros (negative input is clamped first): entries [0, n] are 0.0f - what ssim_yuv420_rgb_kernel need not stage
static int srgb_table_zero_upto() {
    static const int n = [] {
        const std::vector<float> &v = srgb_table();
        int k = 0;
        while (v[0] == 0.0f && k + 1 < 65537 && v[k + 1] == 0.0f) ++k;
        return k;
    }();
    return n;
}
static float srgb_lookup(float x) {
    float t = std::nearbyintf(x * 32768.0f + 16384.0f);
    t = std::min(std::max(t, 0.0f), 65536.0f);
    return srgb_table()[(int)t];
}

namespace {

struct SsimLutKey {
    int mode, bits, limited, linearize;
    bool operator==(const SsimLutKey &o) const { return mode == o.mode && bits == o.bits && limited == o.limited && linearize == o.linearize; }
};

// Device copy of the conversion table of one source format, cached in the context (a clip has one format).
struct SsimLutCache {
    SsimLutKey key{-1, 0, 0, 0};
    float *dev = nullptr;
    size_t entries = 0;
    // YUV sources: the resampling tables of the clip's geometry (hleft[w] | vleft[h] | hcoef[4 w] | vcoef[4 h])
    int yw = 0, yh = 0, yssw = -1, yssh = -1, yloc = -1;
    char *ydev = nullptr;
    size_t ybytes = 0;
    // ... and, for 4:2:0, what ssim_yuv420_rgb_kernel needs of them (ssim_yuv420_plan); yl_ok: the tables fit its assumptions
    bool yl_ok = false;
    YuvLds yl{};
    int yl_span = 0;       // widest tile span, chroma samples
    int yl_lds_limit = 0;  // bytes of LDS a workgroup may take (0: not asked yet, < 0: the attribute call failed)
    // ... for 4:4:4 / 4:2:2, what ssim_yuvrow_rgb_kernel needs (the horizontal table's period); yr_ok: the format is one of the two and the table fits
    bool yr_ok = false;
    YuvRow yr{};
    int yr_lds_limit = 0;
};

SsimLutCache *lut_cache_of(vszip_ctx *ctx) {
    if (!ctx->ssim_lut) ctx->ssim_lut = new SsimLutCache();
    return static_cast<SsimLutCache *>(ctx->ssim_lut);
}

template <typename T, int MODE>
void launch_pyr(bool gray, dim3 grid, hipStream_t st, const PyrArgs &pa) {
    if (gray)
        hipLaunchKernelGGL((ssim_pyr_kernel<T, MODE, true>), grid, dim3(256), 0, st, pa);
    else
        hipLaunchKernelGGL((ssim_pyr_kernel<T, MODE, false>), grid, dim3(256), 0, st, pa);
}

}  // namespace

void vszip_ssim_release(vszip_ctx *ctx) {
    if (!ctx->ssim_lut) return;
    SsimLutCache *c = static_cast<SsimLutCache *>(ctx->ssim_lut);
    if (c->dev) (void)hipFree(c->dev);
    if (c->ydev) (void)hipFree(c->ydev);
    delete c;
    ctx->ssim_lut = nullptr;
}

// Validates the source format and makes its conversion table resident (cached in the context).
static int ssim_prepare(vszip_ctx *ctx, const vszip_ssim_source *fmt, int *mode_out, bool *gray_out, const float **lut_out, int *lds_out) {
    const bool gray = fmt->family == VSZIP_CF_GRAY;
    *gray_out = gray;
    const bool yuv = fmt->family == VSZIP_CF_YUV;
    if (fmt->family != VSZIP_CF_RGB && !gray && !yuv) return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "SSIMULACRA2: colour family %d has no device pre-stage", fmt->family);
    int &mode = *mode_out;
    if (yuv) {
        mode = PYR_YUV;
        if (fmt->dtype != VSZIP_F32 && fmt->dtype != VSZIP_U8 && fmt->dtype != VSZIP_U16) return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "SSIMULACRA2: sample type %d", fmt->dtype);
        if (fmt->dtype != VSZIP_F32 && (fmt->bits < 8 || fmt->bits > (fmt->dtype == VSZIP_U8 ? 8 : 16))) return vszip_set_error(ctx, VSZIP_ERR_ARG, "SSIMULACRA2: %d-bit samples in this container", fmt->bits);
        if (fmt->ssw < 0 || fmt->ssw > 2 || fmt->ssh < 0 || fmt->ssh > 2 || fmt->chroma_loc < 0 || fmt->chroma_loc > 5) return vszip_set_error(ctx, VSZIP_ERR_ARG, "SSIMULACRA2: chroma layout %d/%d/%d", fmt->ssw, fmt->ssh, fmt->chroma_loc);
        if (fmt->matrix != 1 && fmt->matrix != 5 && fmt->matrix != 6 && fmt->matrix != 9) return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "SSIMULACRA2: _Matrix %d has no device pre-stage", fmt->matrix);
    } else if (fmt->dtype == VSZIP_F32) {
        mode = fmt->linearize ? PYR_F32_GAMMA : PYR_F32_LINEAR;
    } else if (fmt->dtype == VSZIP_U8 || fmt->dtype == VSZIP_U16) {
        mode = PYR_INT;
        if (fmt->bits < 8 || fmt->bits > (fmt->dtype == VSZIP_U8 ? 8 : 16)) return vszip_set_error(ctx, VSZIP_ERR_ARG, "SSIMULACRA2: %d-bit samples in this container", fmt->bits);
    } else {
        return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "SSIMULACRA2: sample type %d", fmt->dtype);  // f16 is rejected by the wrapper (:106-113)
    }
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));

    // conversion table of this source format
    const float *&lut_dev = *lut_out;
    int &lut_lds = *lds_out;
    lut_dev = nullptr;
    lut_lds = 0;
    if (mode != PYR_F32_LINEAR && !(mode == PYR_YUV && !fmt->linearize)) {
        SsimLutCache *lc = lut_cache_of(ctx);
        // (a YUV clip's samples go through arithmetic first: it shares the f32 transfer table)
        const SsimLutKey key = mode == PYR_YUV ? SsimLutKey{PYR_F32_GAMMA, 32, 0, 1} : SsimLutKey{mode, fmt->bits, fmt->limited, fmt->linearize};
        if (!(lc->key == key)) {
            std::vector<float> host;
            if (mode == PYR_F32_GAMMA || mode == PYR_YUV) {
                host = srgb_table();
            } else {
                // zimg integer -> float as its x86 kernels do it: fma(v, f32(1 / range), f32(-offset / range)); then the transfer table
                const int off = fmt->limited ? (16 << (fmt->bits - 8)) : 0;
                const int rng = fmt->limited ? (219 << (fmt->bits - 8)) : ((1 << fmt->bits) - 1);
                const float sc = (float)(1.0 / rng), so = (float)(-(double)off / rng);
                host.resize((size_t)1 << fmt->bits);
                for (size_t v = 0; v < host.size(); ++v) {
                    const float x = std::fmaf((float)v, sc, so);
                    host[v] = fmt->linearize ? srgb_lookup(x) : x;
                }
            }
            if (lc->entries < host.size()) {
                VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                if (lc->dev) (void)hipFree(lc->dev);
                lc->dev = nullptr;
                lc->entries = 0;
                VSZIP_HIP_CHECK(ctx, vszip_hip_malloc(ctx, reinterpret_cast<void **>(&lc->dev), host.size() * sizeof(float)));
                lc->entries = host.size();
            }
            VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // a launch still reading the old table
            VSZIP_HIP_CHECK(ctx, hipMemcpy(lc->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
            lc->key = key;
        }
        lut_dev = lc->dev;
        if (mode == PYR_INT && ((size_t)1 << fmt->bits) <= (size_t)kPyrLdsLut) lut_lds = 1 << fmt->bits;
    }

    return VSZIP_OK;
}

// A resampling table's period: taps and coefficients of the four samples of a group x0 .. x0 + 3 (x0 a multiple of 4) relative to x0 >> 1, taken
// from the middle of the axis, and the range [r0, r1) of samples (multiples of 4) that follow it.
static void ssim_axis_period(const int32_t *left, const float *coef, int n, int cn, int *d, float (*c)[4], int *r0, int *r1) {
    const int ref = ((n / 2) / 4) * 4;
    for (int i = 0; i < 4; ++i) {
        d[i] = left[ref + i] - (ref >> 1);  // first tap of sample x0 + i = (x0 >> 1) + d[i], x0 a multiple of 4
        std::memcpy(c[i], coef + 4 * (size_t)(ref + i), 16);
    }
    auto regular = [&](int x) {
        const int i = x & 3;
        return left[x] == ((x - i) >> 1) + d[i] && left[x] + 3 <= cn - 1 && std::memcmp(coef + 4 * (size_t)x, c[i], 16) == 0;
    };
    int lo = ref, hi = ref;
    while (hi < n && regular(hi)) ++hi;
    while (lo > 0 && regular(lo - 1)) --lo;
    *r0 = (lo + 3) & ~3;
    *r1 = hi == n ? n : (hi & ~3);
}

// 4:2:0: tile spans, the tables' period and the range of columns / rows that follow it, from the host copy of the
// resampling tables (left[w] | left[h] | coef[4 w] | coef[4 h]); spans[] receives hspan[2 nbx] | vspan[2 nby]. A table's first tap is
// NOT monotonic when chroma is co-sited (a sample on a chroma sample has one non-zero tap, its neighbour four that start one earlier):
// spans are min / max over the tile.
static bool ssim_yuv420_plan(const int32_t *hleft, const float *hcoef, const int32_t *vleft, const float *vcoef, int w, int h, int cw, int ch, YuvLds *yl, int *span,
                             std::vector<int32_t> *spans) {
    if (w < 8 || h < 8) return false;
    const int nbx = (w + 255) / 256, nby = (h + 15) / 16;
    spans->assign((size_t)2 * nbx + 2 * nby, 0);
    *span = 0;
    for (int b = 0; b < nbx; ++b) {
        int lo = INT32_MAX, hi = 0;
        for (int X = b * 256; X < std::min(b * 256 + 256, w); ++X) lo = std::min(lo, hleft[X]), hi = std::max(hi, std::min(hleft[X] + 3, cw - 1));
        if (lo < 0) return false;
        (*spans)[2 * b] = lo;
        (*spans)[2 * b + 1] = hi;
        *span = std::max(*span, hi - lo + 1);
    }
    for (int b = 0; b < nby; ++b) {
        int lo = INT32_MAX, hi = 0;
        for (int y = b * 16; y < std::min(b * 16 + 16, h); ++y) lo = std::min(lo, vleft[y]), hi = std::max(hi, std::min(vleft[y] + 3, ch - 1));
        if (lo < 0 || hi - lo + 1 > kYlRows) return false;
        (*spans)[2 * nbx + 2 * b] = lo;
        (*spans)[2 * nbx + 2 * b + 1] = hi;
    }
    for (int y0 = 0; y0 < h; y0 += 4) {  // a 4-row block's taps: kYlNr chroma rows from the smallest first tap
        int lo = INT32_MAX, hi = 0;
        for (int y = y0; y < y0 + 4; ++y) lo = std::min(lo, vleft[std::min(y, h - 1)]), hi = std::max(hi, std::min(vleft[std::min(y, h - 1)] + 3, ch - 1));
        if (hi - lo + 1 > kYlNr) return false;
    }
    ssim_axis_period(hleft, hcoef, w, cw, yl->dh, yl->ch, &yl->xr0, &yl->xr1);
    ssim_axis_period(vleft, vcoef, h, ch, yl->dv, yl->cv, &yl->yr0, &yl->yr1);
    int dvmin = yl->dv[0], dvmax = yl->dv[0];
    for (int r = 1; r < 4; ++r) dvmin = std::min(dvmin, yl->dv[r]), dvmax = std::max(dvmax, yl->dv[r]);
    if (dvmax - dvmin > kYlNr - 4) return false;
    int dhmin = yl->dh[0], dhmax = yl->dh[0];
    for (int i = 1; i < 4; ++i) dhmin = std::min(dhmin, yl->dh[i]), dhmax = std::max(dhmax, yl->dh[i]);
    if (dhmax - dhmin > 2) return false;  // (the interior path reads a row's 16 taps out of 3 / 4 aligned dwords)
    yl->nbx = nbx;
    yl->nby = nby;
    return true;
}

// YUV sources: the kernel arguments of one clip geometry — resampling tables resident (cached in the context),
// zimg's conversion constants and matrix.
static int ssim_yuv_args(vszip_ctx *ctx, const vszip_ssim_source *fmt, int w, int h, YuvArgs *ya) {
    const int cw = (w + (1 << fmt->ssw) - 1) >> fmt->ssw, ch = (h + (1 << fmt->ssh) - 1) >> fmt->ssh;
    SsimLutCache *lc = lut_cache_of(ctx);
    const size_t off_v = (size_t)w * 4, off_hc = off_v + (size_t)h * 4, off_vc = off_hc + (size_t)w * 16, off_sp = off_vc + (size_t)h * 16;
    const size_t bytes = off_sp + ((size_t)(w + 255) / 256 + (size_t)(h + 15) / 16) * 8;  // + hspan | vspan (ssim_yuv420_plan)
    if (!(lc->yw == w && lc->yh == h && lc->yssw == fmt->ssw && lc->yssh == fmt->ssh && lc->yloc == fmt->chroma_loc)) {
        std::vector<char> host(bytes);
        // Position of a sited chroma sample relative to the centre of its 2^ss luma samples, in luma samples, by zimg's
        // rule (graphbuilder: the raw siting shift is -/+0.5 whatever the subsampling, scaled by 1 / 2^ss into chroma
        // samples): half a luma sample for every ss > 0. At ss = 1 this is also the geometric co-sited position; at
        // ss = 2 (4:1:0 / 4:1:1) zimg's rule and the geometric one (1.5 luma samples) differ and the reference's
        // resize.Bicubic is zimg. No reference golden covers ss = 2: the plugin keeps those clips on the host resize.
        auto offset = [&](int ss, bool vertical) -> double {
            if (ss == 0) return 0.0;
            const double edge = -0.5;
            const int loc = fmt->chroma_loc;
            if (vertical) return (loc == 2 || loc == 3) ? edge : ((loc == 4 || loc == 5) ? -edge : 0.0);
            return (loc == 0 || loc == 2 || loc == 4) ? edge : 0.0;
        };
        if (fmt->ssw) {
            const int rc = vszip_resample_table(cw, w, -offset(fmt->ssw, false) / (1 << fmt->ssw), reinterpret_cast<int32_t *>(host.data()), reinterpret_cast<float *>(host.data() + off_hc));
            if (rc != VSZIP_OK) return vszip_set_error(ctx, rc, "SSIMULACRA2: no horizontal resampling table for %d -> %d", cw, w);
        }
        if (fmt->ssh) {
            const int rc = vszip_resample_table(ch, h, -offset(fmt->ssh, true) / (1 << fmt->ssh), reinterpret_cast<int32_t *>(host.data() + off_v), reinterpret_cast<float *>(host.data() + off_vc));
            if (rc != VSZIP_OK) return vszip_set_error(ctx, rc, "SSIMULACRA2: no vertical resampling table for %d -> %d", ch, h);
        }
        lc->yl_ok = false;
        if (fmt->ssw == 1 && fmt->ssh == 1) {
            std::vector<int32_t> spans;
            lc->yl_ok = ssim_yuv420_plan(reinterpret_cast<const int32_t *>(host.data()), reinterpret_cast<const float *>(host.data() + off_hc), reinterpret_cast<const int32_t *>(host.data() + off_v),
                                         reinterpret_cast<const float *>(host.data() + off_vc), w, h, cw, ch, &lc->yl, &lc->yl_span, &spans);
            if (lc->yl_ok) std::memcpy(host.data() + off_sp, spans.data(), spans.size() * 4);
        }
        lc->yr_ok = false;
        if (fmt->ssh == 0 && fmt->ssw == 0) {
            lc->yr_ok = true;  // 4:4:4: no table at all
        } else if (fmt->ssh == 0 && fmt->ssw == 1 && w >= 8) {
            YuvRow &yr = lc->yr;
            ssim_axis_period(reinterpret_cast<const int32_t *>(host.data()), reinterpret_cast<const float *>(host.data() + off_hc), w, cw, yr.dh, yr.ch, &yr.xr0, &yr.xr1);
            int dhmin = yr.dh[0], dhmax = yr.dh[0];
            for (int i = 1; i < 4; ++i) dhmin = std::min(dhmin, yr.dh[i]), dhmax = std::max(dhmax, yr.dh[i]);
            lc->yr_ok = dhmax - dhmin <= 2;  // (a group's 16 taps out of 3 / 4 aligned dwords)
        }
        VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // a launch still reading the old tables
        if (lc->ybytes < bytes) {
            if (lc->ydev) (void)hipFree(lc->ydev);
            lc->ydev = nullptr;
            lc->ybytes = 0;
            VSZIP_HIP_CHECK(ctx, vszip_hip_malloc(ctx, reinterpret_cast<void **>(&lc->ydev), bytes));
            lc->ybytes = bytes;
        }
        VSZIP_HIP_CHECK(ctx, hipMemcpy(lc->ydev, host.data(), bytes, hipMemcpyHostToDevice));
        lc->yw = w; lc->yh = h; lc->yssw = fmt->ssw; lc->yssh = fmt->ssh; lc->yloc = fmt->chroma_loc;
    }
    if (lc->yl_ok) {
        lc->yl.hspan = reinterpret_cast<const int *>(lc->ydev + off_sp);
        lc->yl.vspan = lc->yl.hspan + 2 * lc->yl.nbx;
    }
    ya->hleft = fmt->ssw ? reinterpret_cast<const int *>(lc->ydev) : nullptr;
    ya->vleft = fmt->ssh ? reinterpret_cast<const int *>(lc->ydev + off_v) : nullptr;
    ya->hcoef = reinterpret_cast<const float *>(lc->ydev + off_hc);
    ya->vcoef = reinterpret_cast<const float *>(lc->ydev + off_vc);
    ya->cstride = (int)fmt->chroma_stride;
    ya->cw = cw;
    ya->ch = ch;
    // zimg integer -> float: fma(v, f32(1 / range), f32(-offset / range))
    if (fmt->dtype == VSZIP_F32) {
        ya->ys = ya->cs = 1.0f;
        ya->yo = ya->co = 0.0f;
    } else {
        const int b = fmt->bits;
        const double yoff = fmt->limited ? (double)(16 << (b - 8)) : 0.0, yrng = fmt->limited ? (double)(219 << (b - 8)) : (double)((1 << b) - 1);
        const double coff = fmt->limited ? (double)(128 << (b - 8)) : (double)(1 << (b - 1)), crng = fmt->limited ? (double)(224 << (b - 8)) : (double)((1 << b) - 1);
        ya->ys = (float)(1.0 / yrng); ya->yo = (float)(-yoff / yrng);
        ya->cs = (float)(1.0 / crng); ya->co = (float)(-coff / crng);
    }
    // zimg's YUV -> RGB: the inverse (by cofactors, f64) of the non-constant-luminance RGB -> YUV matrix, rounded to f32
    {
        double kr, kb;
        switch (fmt->matrix) {
            case 1: kr = 0.2126; kb = 0.0722; break;
            case 9: kr = 0.2627; kb = 0.0593; break;
            default: kr = 0.299; kb = 0.114; break;  // 5, 6
        }
        const double kg = 1.0 - kr - kb, us = 1.0 / (2.0 - 2.0 * kb), vs = 1.0 / (2.0 - 2.0 * kr);
        const double m[3][3] = {{kr, kg, kb}, {-kr * us, -kg * us, (1.0 - kb) * us}, {(1.0 - kr) * vs, -kg * vs, -kb * vs}};
        const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const int r0 = j == 0 ? 1 : 0, r1 = j == 2 ? 1 : 2, c0 = i == 0 ? 1 : 0, c1 = i == 2 ? 1 : 2;
                const double minor = m[r0][c0] * m[r1][c1] - m[r0][c1] * m[r1][c0];
                ya->m[3 * i + j] = (float)((((i + j) & 1) ? -1.0 : 1.0) * minor / det);
            }
    }
    ya->linearize = fmt->linearize;
    return VSZIP_OK;
}

// Scratch per pair (floats): the XYB planes scales 0 and 1 need (2 n0 + 4 n1: Y of scale 0; X, Y of
// scale 1), the scale-2 linear RGB (6 n2), the XYB planes of one later scale at a time (6 n2) and the RGB
// of scales 3 / 4 (6 n3 + 6 n4) — 130 MB per 4K pair.
VSZIP_EXPORT int vszip_ssimulacra2_src(vszip_ctx *ctx, const vszip_ssim_source *fmt, const void *const *ref_planes, const void *const *dis_planes,
                                       ptrdiff_t stride, int w, int h, int npairs, double *scores) {
    if (!ctx || !fmt || !ref_planes || !dis_planes || !scores || npairs <= 0 || w <= 0 || h <= 0) return VSZIP_ERR_ARG;
    int mode, lut_lds;
    bool gray;
    const float *lut_dev;
    int rc = ssim_prepare(ctx, fmt, &mode, &gray, &lut_dev, &lut_lds);
    if (rc != VSZIP_OK) return rc;
    const int nsp = gray ? 1 : 3;
    static const XybK kx = make_xyb_consts();
    YuvArgs ya{};
    if (mode == PYR_YUV) {
        rc = ssim_yuv_args(ctx, fmt, w, h, &ya);
        if (rc != VSZIP_OK) return rc;
    }

    // 4:2:0 integer clips whose tables and planes fit ssim_yuv420_rgb_kernel: the colour pre-stage as a pass of its own (200 MB of scratch a 4K pair more)
    bool split = false;
    int yl_teams = 0, yl_pd = 0;
    if (mode == PYR_YUV) {
        SsimLutCache *lc = lut_cache_of(ctx);
        const int B = fmt->dtype == VSZIP_U8 ? 1 : 2;
        yl_teams = B == 1 ? YlGeo<uint8_t>::TEAMS : YlGeo<uint16_t>::TEAMS;
        yl_pd = B == 1 ? YlGeo<uint8_t>::PD : YlGeo<uint16_t>::PD;
        split = lc->yl_ok && !ctx->opt.ssim_no_yuv420_lds && fmt->linearize && (fmt->dtype == VSZIP_U8 || fmt->dtype == VSZIP_U16) && lc->yl_lds_limit >= 0 && (w & 3) == 0 &&
                (ya.cw * B) % 4 == 0 && ((size_t)ya.cstride * B) % 4 == 0 && lc->yl_span * B + 3 <= yl_pd * 4;
        for (int i = 0; split && i < npairs; ++i)
            for (int c = 1; c < 3; ++c)
                if ((reinterpret_cast<uintptr_t>(ref_planes[i * nsp + c]) | reinterpret_cast<uintptr_t>(dis_planes[i * nsp + c])) & 3) split = false;
        if (split && lc->yl_lds_limit == 0) {  // once a context: may a workgroup take the whole LDS?
            const int want = 160 * 1024;
            bool ok = true;
            for (const void *fn : {reinterpret_cast<const void *>(ssim_yuv420_rgb_kernel<uint8_t, false>), reinterpret_cast<const void *>(ssim_yuv420_rgb_kernel<uint8_t, true>),
                                   reinterpret_cast<const void *>(ssim_yuv420_rgb_kernel<uint16_t, false>), reinterpret_cast<const void *>(ssim_yuv420_rgb_kernel<uint16_t, true>)})
                ok = ok && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
            (void)hipGetLastError();
            lc->yl_lds_limit = ok ? want : -1;
            split = ok;
        }
    }

    // 4:4:4 / 4:2:2 integer clips (round 6): the same split with ssim_yuvrow_rgb_kernel
    bool split_row = false;
    if (mode == PYR_YUV && !split) {
        SsimLutCache *lc = lut_cache_of(ctx);
        const int B = fmt->dtype == VSZIP_U8 ? 1 : 2;
        split_row = lc->yr_ok && fmt->ssh == 0 && fmt->ssw <= 1 && !ctx->opt.ssim_no_yuv420_lds && fmt->linearize && (fmt->dtype == VSZIP_U8 || fmt->dtype == VSZIP_U16) &&
                    lc->yr_lds_limit >= 0 && (w & 3) == 0 && ((size_t)stride * B) % 4 == 0 && ((size_t)ya.cstride * B) % 4 == 0 && (ya.cw * B) % 4 == 0 && (size_t)w * h < ((size_t)1 << 30);
        for (int i = 0; split_row && i < npairs; ++i)
            for (int c = 0; c < 3; ++c)
                if ((reinterpret_cast<uintptr_t>(ref_planes[i * nsp + c]) | reinterpret_cast<uintptr_t>(dis_planes[i * nsp + c])) & 3) split_row = false;
        if (split_row && lc->yr_lds_limit == 0) {
            const int want = 160 * 1024;
            bool ok = true;
            for (const void *fn : {reinterpret_cast<const void *>(ssim_yuvrow_rgb_kernel<uint8_t, false>), reinterpret_cast<const void *>(ssim_yuvrow_rgb_kernel<uint8_t, true>),
                                   reinterpret_cast<const void *>(ssim_yuvrow_rgb_kernel<uint16_t, false>), reinterpret_cast<const void *>(ssim_yuvrow_rgb_kernel<uint16_t, true>)})
                ok = ok && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
            (void)hipGetLastError();
            lc->yr_lds_limit = ok ? want : -1;
            split_row = ok;
        }
    }

    int sw[kScales + 1], sh[kScales + 1];
    sw[0] = w;
    sh[0] = h;
    for (int s = 1; s <= kScales; ++s) {
        sw[s] = (sw[s - 1] + 1) / 2;
        sh[s] = (sh[s - 1] + 1) / 2;
    }
    size_t npx[kScales + 1];
    for (int s = 0; s <= kScales; ++s) npx[s] = (size_t)sw[s] * sh[s];
    int nneed[kScales];  // XYB planes a scale keeps
    for (int s = 0; s < kScales; ++s) {
        nneed[s] = 0;
        for (int c = 0; c < 3; ++c) nneed[s] += skip_of(c, s).all() ? 0 : 1;
    }
    const int tiles0 = ((w + TW - 1) / TW) * ((h + TH - 1) / TH);
    // (every region starts on a multiple of 4 floats: the linear-RGB planes behind them are written and read with 16-byte accesses - ADVICE r5: 6 npx[3] + 6 npx[4]
    // is 2 mod 4 for e.g. 1924 x 1080)
    auto al4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    const size_t f_x0 = al4(2 * nneed[0] * npx[0]), f_x1 = al4(2 * nneed[1] * npx[1]), f_r2 = al4(6 * npx[2]), f_xs = al4(6 * npx[2]), f_r3 = al4(6 * npx[3]), f_r4 = al4(6 * npx[4]);
    const size_t f_rgb = (split || split_row) ? 6 * npx[0] : 0;  // the two frames' linear RGB between the pre-stage pass and the pyramid pass
    const size_t f_pair = (f_x0 + f_x1 + f_r2 + f_xs + f_r3 + f_r4 + f_rgb + 63) & ~(size_t)63;
    const size_t bytes_part = ((size_t)npairs * 18 * tiles0 * 6 * sizeof(double) + 255) & ~(size_t)255;
    const size_t bytes_avg = ((size_t)npairs * 18 * 6 * sizeof(double) + 255) & ~(size_t)255;
    const size_t bytes_tab = ((size_t)kScales * npairs * sizeof(PairPtrs) + 255) & ~(size_t)255;
    const size_t bytes_pyr = ((size_t)npairs * sizeof(PyrPair) + 255) & ~(size_t)255;
    const size_t need = bytes_part + bytes_avg + bytes_tab + bytes_pyr + (size_t)npairs * f_pair * sizeof(float) + 1024;
    rc = vszip_ensure_scratch(ctx, need);
    if (rc != VSZIP_OK) return rc;
    rc = vszip_ensure_scalars(ctx, std::max(bytes_avg, bytes_tab + bytes_pyr));
    if (rc != VSZIP_OK) return rc;
    char *base = static_cast<char *>(ctx->scratch);
    double *partial = reinterpret_cast<double *>(base);
    PairPtrs *tab_dev = reinterpret_cast<PairPtrs *>(base + bytes_part + bytes_avg);
    PyrPair *pyr_dev = reinterpret_cast<PyrPair *>(base + bytes_part + bytes_avg + bytes_tab);
    float *fbase = reinterpret_cast<float *>(base + bytes_part + bytes_avg + bytes_tab + bytes_pyr);

    // plane pointers of every (scale, pair), staged in the pinned buffer (the final kernel overwrites it with the
    // averages, later in stream order)
    PairPtrs *tab = static_cast<PairPtrs *>(ctx->scalars_host);
    PyrPair *pyr = reinterpret_cast<PyrPair *>(static_cast<char *>(ctx->scalars_host) + bytes_tab);
    std::memset(tab, 0, bytes_tab + bytes_pyr);
    for (int pair = 0; pair < npairs; ++pair) {
        float *x0 = fbase + (size_t)pair * f_pair, *x1 = x0 + f_x0, *r2 = x1 + f_x1, *xs = r2 + f_r2, *r3 = xs + f_xs, *r4 = r3 + f_r3;
        PyrPair &py = pyr[pair];
        for (int c = 0; c < nsp; ++c) {
            py.src1[c] = ref_planes[pair * nsp + c];
            py.src2[c] = dis_planes[pair * nsp + c];
        }
        for (int scale = 0; scale < kScales; ++scale) {
            PairPtrs &pp = tab[(size_t)scale * npairs + pair];
            int k = 0;
            for (int c = 0; c < 3; ++c) {
                const bool need_plane = !skip_of(c, scale).all();
                float *xa = nullptr, *xb = nullptr;
                if (need_plane) {
                    float *xbase = scale == 0 ? x0 : (scale == 1 ? x1 : xs);
                    xa = xbase + (size_t)k * npx[scale];
                    xb = xbase + (size_t)(nneed[scale] + k) * npx[scale];
                    ++k;
                }
                pp.xyb1[c] = xa;
                pp.xyb2[c] = xb;
                if (scale == 0) { py.x0a[c] = xa; py.x0b[c] = xb; }
                if (scale == 1) { py.x1a[c] = xa; py.x1b[c] = xb; }
                if (scale >= 2) {
                    const float *cur = scale == 2 ? r2 : (scale == 3 ? r3 : r4);
                    pp.rgb1[c] = cur + (size_t)c * npx[scale];
                    pp.rgb2[c] = cur + (size_t)(3 + c) * npx[scale];
                    float *nxt = scale == 2 ? r3 : (scale == 3 ? r4 : nullptr);
                    pp.next1[c] = nxt ? nxt + (size_t)c * npx[scale + 1] : nullptr;
                    pp.next2[c] = nxt ? nxt + (size_t)(3 + c) * npx[scale + 1] : nullptr;
                }
            }
        }
        for (int c = 0; c < 3; ++c) {
            py.r2a[c] = r2 + (size_t)c * npx[2];
            py.r2b[c] = r2 + (size_t)(3 + c) * npx[2];
            if (split || split_row) {
                py.rgb1[c] = r4 + f_r4 + (size_t)c * npx[0];
                py.rgb2[c] = r4 + f_r4 + (size_t)(3 + c) * npx[0];
            }
        }
    }
    VSZIP_HIP_CHECK(ctx, hipMemcpyAsync(tab_dev, tab, bytes_tab + bytes_pyr, hipMemcpyHostToDevice, ctx->stream));

    // ---- the launches, by piece (pairs [p0, p0 + cnt) on stream st) ----
    // scales 0 + 1 + the scale-2 RGB: one pass over the source
    auto launch_pyramid = [&](int p0, int cnt, hipStream_t st) {
        PyrArgs pa;
        pa.tab = pyr_dev + p0;
        pa.lut = lut_dev;
        pa.lut_lds = lut_lds;
        {
            const size_t bps = fmt->dtype == VSZIP_U8 ? 1 : (fmt->dtype == VSZIP_U16 ? 2 : 4);
            uintptr_t bits = (uintptr_t)((size_t)stride * bps);
            for (int i = 0; i < npairs * nsp; ++i) bits |= reinterpret_cast<uintptr_t>(ref_planes[i]) | reinterpret_cast<uintptr_t>(dis_planes[i]);
            pa.vec_ok = (bits & (4 * bps - 1)) == 0 && (w & 3) == 0;  // (the XYB planes are dense: rows of w floats, 16-byte aligned when w % 4 == 0)
        }
        pa.sstride = (int)stride;
        pa.from_rgb = 0;
        pa.w = w;
        pa.h = h;
        pa.w1 = sw[1];
        pa.h1 = sh[1];
        pa.w2 = sw[2];
        pa.h2 = sh[2];
        pa.k = kx;
        const dim3 grid((w + 255) / 256, (h + 15) / 16, cnt);
        if (mode == PYR_YUV && split) {
            SsimLutCache *lc = lut_cache_of(ctx);
            const int raw_bytes = yl_teams * 2 * kYlRows * yl_pd * 4;
            YuvLds yl = lc->yl;
            yl.lut_n = std::min(65537, ((lc->yl_lds_limit - raw_bytes) / 4 - 4) & ~3);
            const int zero_upto = srgb_table_zero_upto();
            yl.low_zero = zero_upto > 0 && zero_upto + yl.lut_n <= 65537;
            yl.lut_lo = yl.low_zero ? zero_upto : std::min(std::max(32768 - yl.lut_n / 2, 0), 65537 - yl.lut_n);
            // the interior launch takes every WHOLE tile (its first / last columns and rows, where the tables leave their period, are redone by
            // ssim_yuv420_fixup_kernel); the EDGE launch the ragged tiles at the right and the bottom, or everything when the luma planes do not vector-load
            yl.tx0 = yl.ty0 = 0;
            yl.tx1 = pa.vec_ok ? w / 256 : 0;
            yl.ty1 = pa.vec_ok ? h / 16 : 0;
            if (yl.tx1 == 0 || yl.ty1 == 0) yl.tx1 = yl.ty1 = 0;
            int wf = yl.tx1 * 256, hf = yl.ty1 * 16;
            // (the period must hold somewhere inside the whole tiles, or the interior has nothing regular to offer)
            if (yl.xr0 >= std::min(yl.xr1, wf) || yl.yr0 >= std::min(yl.yr1, hf)) yl.tx1 = yl.ty1 = wf = hf = 0;
            const int lds = ((yl.lut_n + 3) & ~3) * 4 + raw_bytes;
            const int inner = (yl.tx1 - yl.tx0) * (yl.ty1 - yl.ty0), cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
            for (int edge = 0; edge < 2; ++edge) {
                yl.ntiles = cnt * 2 * (edge ? yl.nbx * yl.nby - inner : inner);
                if (yl.ntiles == 0) continue;
                const int teams = edge ? kYlEdgeTeams : yl_teams;  // (the LDS carve-up is the interior's either way: the EDGE launch uses its first slices)
                const dim3 groups(std::min(cus, (yl.ntiles + teams - 1) / teams)), threads(256 * teams);
                if (fmt->dtype == VSZIP_U8 && !edge)
                    hipLaunchKernelGGL((ssim_yuv420_rgb_kernel<uint8_t, false>), groups, threads, lds, st, pa, ya, yl);
                else if (fmt->dtype == VSZIP_U8)
                    hipLaunchKernelGGL((ssim_yuv420_rgb_kernel<uint8_t, true>), groups, threads, lds, st, pa, ya, yl);
                else if (!edge)
                    hipLaunchKernelGGL((ssim_yuv420_rgb_kernel<uint16_t, false>), groups, threads, lds, st, pa, ya, yl);
                else
                    hipLaunchKernelGGL((ssim_yuv420_rgb_kernel<uint16_t, true>), groups, threads, lds, st, pa, ya, yl);
            }
            if (inner) {
                YuvLds yf = yl;
                yf.xr1 = std::min(yl.xr1, wf);  // (columns / rows past the whole tiles are the EDGE launch's)
                yf.yr1 = std::min(yl.yr1, hf);
                const long npx_fix = (long)(yf.xr0 + wf - yf.xr1) * hf + (long)(yf.yr0 + hf - yf.yr1) * wf;
                if (npx_fix > 0) {
                    const dim3 fgrid((unsigned)((npx_fix + 255) / 256), (unsigned)(cnt * 2));
                    if (fmt->dtype == VSZIP_U8)
                        hipLaunchKernelGGL(ssim_yuv420_fixup_kernel<uint8_t>, fgrid, dim3(256), 0, st, pa, ya, yf, wf, hf);
                    else
                        hipLaunchKernelGGL(ssim_yuv420_fixup_kernel<uint16_t>, fgrid, dim3(256), 0, st, pa, ya, yf, wf, hf);
                }
            }
            PyrArgs pb = pa;  // the f32 pass over the frames' linear RGB planes (dense rows of w floats, 16-byte aligned: w % 4 == 0)
            pb.from_rgb = 1;
            pb.sstride = w;
            pb.vec_ok = 1;
            launch_pyr<float, PYR_F32_LINEAR>(false, grid, st, pb);
        } else if (mode == PYR_YUV && split_row) {
            SsimLutCache *lc = lut_cache_of(ctx);
            YuvRow yr = lc->yr;
            yr.lut_n = std::min(65537, (lc->yr_lds_limit / 4 - 8) & ~3);
            const int zero_upto = srgb_table_zero_upto();
            yr.low_zero = zero_upto > 0 && zero_upto + yr.lut_n <= 65537;
            yr.lut_lo = yr.low_zero ? zero_upto : std::min(std::max(32768 - yr.lut_n / 2, 0), 65537 - yr.lut_n);
            yr.gpr = w / 4;
            yr.ngroups = (long)cnt * 2 * h * yr.gpr;
            const int lds = ((yr.lut_n + 3) & ~3) * 4, cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
            const dim3 groups((unsigned)std::min<long>(cus, (yr.ngroups + 1023) / 1024)), threads(1024);
            const bool hsub = fmt->ssw == 1;
            if (fmt->dtype == VSZIP_U8 && !hsub)
                hipLaunchKernelGGL((ssim_yuvrow_rgb_kernel<uint8_t, false>), groups, threads, lds, st, pa, ya, yr);
            else if (fmt->dtype == VSZIP_U8)
                hipLaunchKernelGGL((ssim_yuvrow_rgb_kernel<uint8_t, true>), groups, threads, lds, st, pa, ya, yr);
            else if (!hsub)
                hipLaunchKernelGGL((ssim_yuvrow_rgb_kernel<uint16_t, false>), groups, threads, lds, st, pa, ya, yr);
            else
                hipLaunchKernelGGL((ssim_yuvrow_rgb_kernel<uint16_t, true>), groups, threads, lds, st, pa, ya, yr);
            PyrArgs pb = pa;  // the f32 pass over the frames' linear RGB planes
            pb.from_rgb = 1;
            pb.sstride = w;
            pb.vec_ok = 1;
            launch_pyr<float, PYR_F32_LINEAR>(false, grid, st, pb);
        } else if (mode == PYR_YUV) {
            if (fmt->dtype == VSZIP_F32)
                hipLaunchKernelGGL(ssim_pyr_yuv_kernel<float>, grid, dim3(256), 0, st, pa, ya);
            else if (fmt->dtype == VSZIP_U8)
                hipLaunchKernelGGL(ssim_pyr_yuv_kernel<uint8_t>, grid, dim3(256), 0, st, pa, ya);
            else
                hipLaunchKernelGGL(ssim_pyr_yuv_kernel<uint16_t>, grid, dim3(256), 0, st, pa, ya);
        } else if (mode == PYR_F32_LINEAR)
            launch_pyr<float, PYR_F32_LINEAR>(gray, grid, st, pa);
        else if (mode == PYR_F32_GAMMA)
            launch_pyr<float, PYR_F32_GAMMA>(gray, grid, st, pa);
        else if (fmt->dtype == VSZIP_U8)
            launch_pyr<uint8_t, PYR_INT>(gray, grid, st, pa);
        else
            launch_pyr<uint16_t, PYR_INT>(gray, grid, st, pa);
    };
    // scales 2 ..: XYB of this scale + the next scale's RGB
    auto launch_xyb_down = [&](int scale, int p0, int cnt, hipStream_t st) {
        const int cw = sw[scale], ch = sh[scale], nw = sw[scale + 1], nh = sh[scale + 1];
        XybArgs xa;
        xa.tab = tab_dev + (size_t)scale * npairs + p0;
        xa.stride = cw;
        xa.w = cw;
        xa.h = ch;
        xa.nstride = nw;
        xa.nw = nw;
        xa.nh = nh;
        xa.xstride = cw;
        xa.k = kx;
        hipLaunchKernelGGL(ssim_xyb_down_kernel, dim3((nw + 31) / 32, (nh + 7) / 8, cnt), dim3(256), 0, st, xa);
    };
    // the maps of one scale: one launch per term set of the scale's planes (ssim_maps_ts_kernel<SSIM, EDGE>)
    auto launch_maps = [&](int scale, int p0, int cnt, hipStream_t st, bool probe_it) {
        const int cw = sw[scale], ch = sh[scale];
        MapsArgs ma;
        ma.tab = tab_dev + (size_t)scale * npairs + p0;
        ma.nactive = 0;
        for (int c = 0; c < 3; ++c) {
            const Skip sk = skip_of(c, scale);
            if (sk.all()) continue;
            const int k = ma.nactive++;
            ma.plane[k] = c;
            ma.flags[k] = (sk.ssim ? 0 : 1) | ((!sk.artifact || !sk.detail) ? 2 : 0);
            ma.slot[k] = scale * 3 + c;
        }
        if (ma.nactive == 0) return;
        ma.stride = cw;
        ma.w = cw;
        ma.h = ch;
        ma.tiles_x = (cw + TW - 1) / TW;
        ma.tiles_y = (ch + TH - 1) / TH;
        ma.partial = partial + (size_t)p0 * 18 * tiles0 * 6;
        ma.max_tiles = tiles0;
        // tiles (along x) per block: small scales keep one (enough blocks to fill the chip); the large ones walk 4 with the next tile's inputs in flight,
        // and 6 or 12 while that still leaves 4 096 blocks (one reduction and one set of partials per block)
#ifdef VSZIP_SSIM_TPB
        ma.tpb = ma.tiles_x >= 32 ? VSZIP_SSIM_TPB : 1;
#else
        ma.tpb = 1;
        if (ma.tiles_x >= 32) {
            ma.tpb = 4;
            for (int t : {6, 12}) {
                const long nb = (long)((ma.tiles_x + t - 1) / t) * ma.tiles_y * ma.nactive * cnt;
                if (nb >= 4096) ma.tpb = t;
            }
        }
#endif
        const int gx = (ma.tiles_x + ma.tpb - 1) / ma.tpb, gy = ma.tiles_y;
        if (probe_it && ctx->probe_on) vszip_probe_mark(ctx);  // (HIP events on the context's stream: other streams' launches are not probed)
        for (int fl = 3; fl >= 1; --fl) {
            MapsArgs mt = ma;
            mt.nactive = 0;
            for (int k = 0; k < ma.nactive; ++k)
                if (ma.flags[k] == fl) {
                    mt.plane[mt.nactive] = ma.plane[k];
                    mt.flags[mt.nactive] = fl;
                    mt.slot[mt.nactive] = ma.slot[k];
                    ++mt.nactive;
                }
            if (mt.nactive == 0) continue;
            mt.gx = gx;
            mt.gy = gy;
            mt.nblk = mt.gx * mt.gy * mt.nactive * cnt;
            const dim3 g((unsigned)(((mt.nblk + 7) / 8) * 8));
            if (fl == 3)
                hipLaunchKernelGGL((ssim_maps_ts_kernel<true, true>), g, dim3(256), 0, st, mt);
            else if (fl == 1)
                hipLaunchKernelGGL((ssim_maps_ts_kernel<true, false>), g, dim3(256), 0, st, mt);
            else
                hipLaunchKernelGGL((ssim_maps_ts_kernel<false, true>), g, dim3(256), 0, st, mt);
        }
        if (probe_it && ctx->probe_on) vszip_probe_mark(ctx);
    };
    auto launch_final = [&](hipStream_t st) -> int {
        FinalArgs fin;
        fin.partial = partial;
        // the averages go straight into the pinned host buffer (device-visible): no copy command at the end
        VSZIP_HIP_CHECK(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&fin.avg), ctx->scalars_host, 0));
        fin.max_tiles = tiles0;
        for (int i = 0; i < 18; ++i) {
            fin.ntiles[i] = 0;
            fin.one_per_pixels[i] = 0;
        }
        for (int scale = 0; scale < kScales; ++scale)
            for (int c = 0; c < 3; ++c) {
                if (skip_of(c, scale).all()) continue;
                fin.ntiles[scale * 3 + c] = ((sw[scale] + TW - 1) / TW) * ((sh[scale] + TH - 1) / TH);
                fin.one_per_pixels[scale * 3 + c] = 1.0 / (double)((uint32_t)sw[scale] * (uint32_t)sh[scale]);
            }
        hipLaunchKernelGGL(ssim_final_kernel, dim3(18, npairs), dim3(kFinThreads), 0, st, fin);
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        return VSZIP_OK;
    };

    // ---- the schedule ----
    // Two streams: the pyramid pass of all pairs, then the large scales' maps kernels (issue bound) on the context's stream and BESIDE them, on a
    // second stream, the small scales (2 ..: a sixteenth of the samples and less, launch bound); join, final reduction.
    // Round 6 (tools/ssim_small_calls.py, interleaved library A/B, pairs/s, linear RGBS): round 5 cut a call of 4 pairs and more into two halves -
    // the second half's pyramid pass beside the first half's maps - and ran the small scales after both; the small scales beside the maps is
    // faster at every size: 1080p x 1 / 4 / 16 pairs a call 5.7 / 15.5 / 25.0 k against 4.5 / 11.7 / 22.5 k, 4K 3.95 / 5.8 / 7.7 k against 2.9 / 5.25 / 7.36 k
    // (one pair - a VapourSynth getFrame - ran on ONE stream before: ADVICE r5). Round 5's other measured alternatives: profiles/r05_notes.md 1.
    bool two = kScales > 2 && !ctx->opt.ssim_one_stream;
    if (two && !ctx->side_stream) {
        if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            ctx->side_stream = nullptr;
            two = false;
        }
    }
    if (!two) {
        launch_pyramid(0, npairs, ctx->stream);
        for (int scale = 0; scale < kScales; ++scale) {
            if (scale >= 2) launch_xyb_down(scale, 0, npairs, ctx->stream);
            launch_maps(scale, 0, npairs, ctx->stream, true);
        }
    } else {
        hipStream_t st[2] = {ctx->stream, ctx->side_stream};
        launch_pyramid(0, npairs, st[0]);
        VSZIP_HIP_CHECK(ctx, hipEventRecord(ctx->side_fork, st[0]));
        VSZIP_HIP_CHECK(ctx, hipStreamWaitEvent(st[1], ctx->side_fork, 0));  // (also orders the pointer-table upload before the second stream)
        for (int scale = 2; scale < kScales; ++scale) {
            launch_xyb_down(scale, 0, npairs, st[1]);
            launch_maps(scale, 0, npairs, st[1], false);
        }
        launch_maps(0, 0, npairs, st[0], true);
        launch_maps(1, 0, npairs, st[0], true);
        VSZIP_HIP_CHECK(ctx, hipEventRecord(ctx->side_join, st[1]));
        VSZIP_HIP_CHECK(ctx, hipStreamWaitEvent(st[0], ctx->side_join, 0));
    }
    rc = launch_final(ctx->stream);
    if (rc != VSZIP_OK) return rc;
    VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const double(*avg)[18][6] = reinterpret_cast<const double(*)[18][6]>(ctx->scalars_host);
    for (int pair = 0; pair < npairs; ++pair) scores[pair] = score_of(avg[pair]);
    return VSZIP_OK;
}

// Linear-light RGBS in (the kernel contract of the reference, :46): the pre-stage is the identity.
VSZIP_EXPORT int vszip_ssimulacra2(vszip_ctx *ctx, const float *const *ref3, const float *const *dis3, ptrdiff_t stride, int w, int h, int npairs, double *scores) {
    const vszip_ssim_source fmt = {VSZIP_CF_RGB, VSZIP_F32, 32, 0, 0};
    return vszip_ssimulacra2_src(ctx, &fmt, reinterpret_cast<const void *const *>(ref3), reinterpret_cast<const void *const *>(dis3), stride, w, h, npairs, scores);
}

// The pre-stage alone (hz.toRGBS + sRGBtoLinearRGB of one clip's frame): what the score is computed from, as
// planes — for hosts that want the converted frame (the reference's output clip is the converted reference
// clip, src/vapoursynth/ssimulacra2.zig:53) and for the tests.
namespace {
template <typename T, int MODE, bool GRAY>
__global__ __launch_bounds__(256) void to_rgbs_kernel(const void *s0, const void *s1, const void *s2, float *d0, float *d1, float *d2, int sstride, int dstride, int w, int h,
                                                      const float *lut) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const void *src[3] = {s0, s1, s2};
    float *dst[3] = {d0, d1, d2};
    float v[3];
#pragma unroll
    for (int c = 0; c < (GRAY ? 1 : 3); ++c) v[c] = pyr_linear<T, MODE>(static_cast<const T *>(src[c])[(size_t)y * sstride + x], lut, nullptr, false);
    if (GRAY) v[1] = v[2] = v[0];
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c][(size_t)y * dstride + x] = v[c];
}
}  // namespace

VSZIP_EXPORT int vszip_to_rgbs_linear(vszip_ctx *ctx, const vszip_ssim_source *fmt, const void *const *src_planes, ptrdiff_t src_stride, float *const *dst3, ptrdiff_t dst_stride,
                                      int w, int h) {
    if (!ctx || !fmt || !src_planes || !dst3 || w <= 0 || h <= 0) return VSZIP_ERR_ARG;
    int mode, lut_lds;
    bool gray;
    const float *lut;
    const int rc = ssim_prepare(ctx, fmt, &mode, &gray, &lut, &lut_lds);
    if (rc != VSZIP_OK) return rc;
    const dim3 grid((w + 255) / 256, h);
    const void *s0 = src_planes[0], *s1 = gray ? nullptr : src_planes[1], *s2 = gray ? nullptr : src_planes[2];
    if (mode == PYR_YUV) {
        YuvArgs ya{};
        const int yrc = ssim_yuv_args(ctx, fmt, w, h, &ya);
        if (yrc != VSZIP_OK) return yrc;
        if (fmt->dtype == VSZIP_F32)
            hipLaunchKernelGGL(yuv_to_rgbs_kernel<float>, grid, dim3(256), 0, ctx->stream, s0, s1, s2, dst3[0], dst3[1], dst3[2], (int)src_stride, (int)dst_stride, w, h, lut, ya);
        else if (fmt->dtype == VSZIP_U8)
            hipLaunchKernelGGL(yuv_to_rgbs_kernel<uint8_t>, grid, dim3(256), 0, ctx->stream, s0, s1, s2, dst3[0], dst3[1], dst3[2], (int)src_stride, (int)dst_stride, w, h, lut, ya);
        else
            hipLaunchKernelGGL(yuv_to_rgbs_kernel<uint16_t>, grid, dim3(256), 0, ctx->stream, s0, s1, s2, dst3[0], dst3[1], dst3[2], (int)src_stride, (int)dst_stride, w, h, lut, ya);
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        return VSZIP_OK;
    }
#define VSZIP_TO_RGBS(T, MODE)                                                                                                                              \
    do {                                                                                                                                                    \
        if (gray)                                                                                                                                           \
            hipLaunchKernelGGL((to_rgbs_kernel<T, MODE, true>), grid, dim3(256), 0, ctx->stream, s0, s1, s2, dst3[0], dst3[1], dst3[2], (int)src_stride, (int)dst_stride, w, h, lut); \
        else                                                                                                                                                \
            hipLaunchKernelGGL((to_rgbs_kernel<T, MODE, false>), grid, dim3(256), 0, ctx->stream, s0, s1, s2, dst3[0], dst3[1], dst3[2], (int)src_stride, (int)dst_stride, w, h, lut); \
    } while (0)
    if (mode == PYR_F32_LINEAR)
        VSZIP_TO_RGBS(float, PYR_F32_LINEAR);
    else if (mode == PYR_F32_GAMMA)
        VSZIP_TO_RGBS(float, PYR_F32_GAMMA);
    else if (fmt->dtype == VSZIP_U8)
        VSZIP_TO_RGBS(uint8_t, PYR_INT);
    else
        VSZIP_TO_RGBS(uint16_t, PYR_INT);
#undef VSZIP_TO_RGBS
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    return VSZIP_OK;
}
