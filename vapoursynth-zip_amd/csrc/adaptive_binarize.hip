// vszip.AdaptiveBinarize on gfx950 (src/vapoursynth/adaptive_binarize.zig:26-73): 8-bit planes,
// dst = 255 where clip2 - clip >= c (compared in i16), else 0. Two streams in, one out, 16 bytes
// per lane; one launch per table of planes.
#include <algorithm>

#include "common.hpp"

namespace {

constexpr int kMaxPlanesAB = 192;  // planes per launch (64 YUV frames are ONE launch since round 4: four 48-plane launches paid four ramps and tails)
constexpr int kRowsAB = 4;

struct ABPlane {
    const uint8_t *a, *b;
    uint8_t *dst;
    int astride, bstride, dstride, w, h, block0;
};
struct ABParams {
    ABPlane p[kMaxPlanesAB];
    int nplanes, c;
};

__device__ __forceinline__ uint8_t ab_px(uint8_t a, uint8_t b, int c) { return ((int)b - (int)a >= c) ? 255 : 0; }

__global__ __launch_bounds__(256) void adaptive_binarize_kernel(const ABParams prm) {
    typedef uint8_t V16 __attribute__((ext_vector_type(16)));
    int pi = 0;
    const int blk = blockIdx.x;

    {  // block0 ascends: eight scalar steps for 192 planes (the linear scan was part of every workgroup's fixed cost)
        int hi = prm.nplanes - 1;
        while (pi < hi) {
            const int mid = (pi + hi + 1) >> 1;
            if (blk >= prm.p[mid].block0)
                pi = mid;
            else
                hi = mid - 1;
        }
    }
    const ABPlane pl = prm.p[pi];
    const int y0 = (blk - pl.block0) * kRowsAB;
    const bool vec = ((reinterpret_cast<uintptr_t>(pl.a) | reinterpret_cast<uintptr_t>(pl.b) | reinterpret_cast<uintptr_t>(pl.dst) | (uintptr_t)pl.astride |
                       (uintptr_t)pl.bstride | (uintptr_t)pl.dstride) & 15) == 0;
    for (int r = 0; r < kRowsAB; ++r) {
        const int y = y0 + r;
        if (y >= pl.h) break;
        const uint8_t *a = pl.a + (size_t)y * pl.astride, *b = pl.b + (size_t)y * pl.bstride;
        uint8_t *d = pl.dst + (size_t)y * pl.dstride;
        int x = 0;
        if (vec) {
            const int nv = pl.w / 16;
            for (int i = threadIdx.x; i < nv; i += 256) {
                const V16 va = reinterpret_cast<const V16 *>(a)[i], vb = reinterpret_cast<const V16 *>(b)[i];
                V16 o;
#pragma unroll
                for (int k = 0; k < 16; ++k) o[k] = ab_px(va[k], vb[k], prm.c);
                __builtin_nontemporal_store(o, reinterpret_cast<V16 *>(d) + i);
            }
            x = nv * 16;
        }
        for (int i = x + threadIdx.x; i < pl.w; i += 256) d[i] = ab_px(a[i], b[i], prm.c);
    }
}

}  // namespace

VSZIP_EXPORT int vszip_adaptive_binarize(vszip_ctx *ctx, const vszip_plane *planes, int nplanes, int c) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (int done = 0; done < nplanes;) {
        ABParams prm;
        prm.c = std::min(std::max(c, -256), 256);  // :96-99
        int n = 0, blocks = 0;
        for (; done + n < nplanes && n < kMaxPlanesAB; ++n) {
            const vszip_plane &s = planes[done + n];
            if (!s.src || !s.ref || !s.dst || s.w <= 0 || s.h <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "AdaptiveBinarize: bad plane %d", done + n);
            ABPlane &d = prm.p[n];
            d.a = static_cast<const uint8_t *>(s.src);
            d.b = static_cast<const uint8_t *>(s.ref);
            d.dst = static_cast<uint8_t *>(s.dst);
            d.astride = (int)s.src_stride;
            d.bstride = (int)s.ref_stride;
            d.dstride = (int)s.dst_stride;
            d.w = s.w;
            d.h = s.h;
            d.block0 = blocks;
            blocks += (s.h + kRowsAB - 1) / kRowsAB;
        }
        prm.nplanes = n;
        {
            vszip_probe_scope probe(ctx);
            hipLaunchKernelGGL(adaptive_binarize_kernel, dim3(blocks), dim3(256), 0, ctx->stream, prm);
        }
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
        done += n;
    }
    return VSZIP_OK;
}
