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
#include "common.hpp"

namespace {

constexpr int kMaxPlanesF = 48;

struct FPlane {
    const void *src;
    void *dst;
    int sstride, dstride, w, h;
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
    int pi = 0;
    const int b = blockIdx.x;
    for (int i = 1; i < prm.nplanes; ++i)
        if (b >= prm.p[i].block0) pi = i;
    const FPlane pl = prm.p[pi];
    const int lb = b - pl.block0;
    const int w = pl.w, h = pl.h;
    const int x0 = (lb % pl.nbx) * FTW, y0 = (lb / pl.nbx) * FTH;
    const T *src = static_cast<const T *>(pl.src);
    T *dst = static_cast<T *>(pl.dst);
    const float div = 1.0f / (float)K;  // :39
    const int tid = threadIdx.x;
    const bool interior = x0 >= R && y0 >= R && x0 + FTW + R <= w && y0 + FTH + R <= h;
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
    const int cx0 = max(x0 - R, 0), cy0 = max(y0 - R, 0);
    const int cw = min(x0 + FTW + R, w) - cx0, ch = min(y0 + FTH + R, h) - cy0;
    for (int i = tid; i < ch * cw; i += 256) {
        const int r = i / cw, c = i - r * cw;
        tile[r][c] = (float)src[(size_t)(cy0 + r) * pl.sstride + cx0 + c];
    }
    __syncthreads();
    const int th = min(FTH, h - y0), tw = min(FTW, w - x0);
    for (int i = tid; i < th * cw; i += 256) {
        const int r = i / cw, c = i - r * cw;
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = acc + div * tile[ct_tap(k, y0 + r, R, h) - cy0][c];
        vt[r][c] = (float)(T)acc;
    }
    __syncthreads();
    for (int i = tid; i < th * tw; i += 256) {
        const int r = i / tw, c = i - r * tw;
        float sum = 0.0f;
        for (int k = 0; k < K; ++k) sum += div * vt[r][ct_tap(k, x0 + c, R, w) - cx0];
        dst[(size_t)(y0 + r) * pl.dstride + x0 + c] = (T)sum;
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

template <typename T>
int run_ct_float(vszip_ctx *ctx, const vszip_plane *planes, int nplanes, int radius) {
    if (radius < 1 || radius > 22) return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: CT float radius out of range");
    int done = 0;
    while (done < nplanes) {
        FParams prm;
        const int n = std::min(kMaxPlanesF, nplanes - done);
        prm.nplanes = n;
        int blocks = 0;
        for (int i = 0; i < n; ++i) {
            const vszip_plane &s = planes[done + i];
            FPlane &d = prm.p[i];
            d.src = s.src;
            d.dst = s.dst;
            d.sstride = (int)s.src_stride;
            d.dstride = (int)s.dst_stride;
            d.w = s.w;
            d.h = s.h;
            d.block0 = blocks;
            d.nbx = (s.w + FTW - 1) / FTW;
            blocks += d.nbx * ((s.h + FTH - 1) / FTH);
        }
        FloatDispatch<T, 22>::launch(ctx, radius, blocks, prm);
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += n;
    }
    return VSZIP_OK;
}

}  // namespace

int vszip_bb_ct_float(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, int radius) {
    if (dtype == VSZIP_F16) return run_ct_float<_Float16>(ctx, planes, nplanes, radius);
    return run_ct_float<float>(ctx, planes, nplanes, radius);
}
