// vszip.XPSNR on gfx950: the per-frame kernel getWSSE (src/filters/xpsnr.zig:376-524).
//
// The pixel work is exact integer arithmetic per XPSNR block: squared error
// (calcSquaredError :214-251), spatial activity (spatialAct :174-212, or highds :28-64 on
// the 2x-decimated grid for frames larger than 2048x1152) and temporal activity
// (tempDiff1/2 :111-170, diff1st/2nd :66-109). One 256-thread workgroup per luma block
// (and per chroma block) reduces those to u64 sums. The few hundred per-block sums are
// copied back and the f64 weighting (calcSquaredErrorAndWeight :315-357, the <=640x480
// minimum smoothing and the weighted sums, getWSSE :437-521) runs on the host in the
// reference's sequential block order, so wsse64 is identical to the reference's.
#include <algorithm>
#include <cmath>
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

}  // namespace

VSZIP_EXPORT int vszip_xpsnr_wsse(vszip_ctx *ctx, int bytes_per_sample, const void *const *org3, const void *const *rec3, const void *prev1, const void *prev2,
                                  const int *width3, const int *height3, const ptrdiff_t *stride3, int depth, int num_comps, unsigned frame_rate, int temporal,
                                  uint64_t *wsse3) {
    if (!ctx || !org3 || !rec3 || !width3 || !height3 || !stride3 || !wsse3 || num_comps < 1 || num_comps > 3) return VSZIP_ERR_ARG;
    if (bytes_per_sample != 1 && bytes_per_sample != 2) return vszip_set_error(ctx, VSZIP_ERR_ARG, "XPSNR : only supports 8 or 10 bit clips");
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t w = (uint32_t)width3[0], h = (uint32_t)height3[0];
    const uint32_t wh = w * h;
    // getWSSE :389-398
    const double r = (double)wh / (3840.0 * 2160.0);
    const double bq = 32.0 * std::sqrt(r) + 0.5;
    const uint32_t b = (uint32_t)(bq < 0 ? 0 : bq) * 4;
    const uint32_t w_blk = b >= 4 ? (w + b - 1) / b : 0, h_blk = b >= 4 ? (h + b - 1) / b : 0;
    const uint32_t sft = 1u << (2 * depth - 9);
    const double avg_act = std::sqrt(16.0 * (double)sft / std::sqrt(std::max(0.00001, r)));
    wsse3[0] = wsse3[1] = wsse3[2] = 0;

    // device result layout: luma [w_blk*h_blk][3], then per chroma plane its block SSEs
    size_t n_luma = (size_t)w_blk * h_blk, total = n_luma * 3;
    size_t coff[3] = {0, 0, 0}, cn[3] = {0, 0, 0};
    uint32_t cbx[3] = {0, 0, 0}, cby[3] = {0, 0, 0}, cnbx[3] = {0, 0, 0}, cnby[3] = {0, 0, 0};
    for (int c = 0; c < num_comps; ++c) {
        const uint32_t wp = (uint32_t)width3[c], hp = (uint32_t)height3[c];
        if (b < 4) {
            cbx[c] = wp;
            cby[c] = hp;
        } else if (c > 0) {
            cbx[c] = (b * wp) / w;
            cby[c] = (b * hp) / h;
        } else {
            continue;
        }
        if (cbx[c] == 0 || cby[c] == 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "XPSNR : plane %d too small", c);
        cnbx[c] = (wp + cbx[c] - 1) / cbx[c];
        cnby[c] = (hp + cby[c] - 1) / cby[c];
        coff[c] = total;
        cn[c] = (size_t)cnbx[c] * cnby[c];
        total += cn[c];
    }
    int rc = vszip_ensure_scratch(ctx, total * sizeof(uint64_t));
    if (rc != VSZIP_OK) return rc;
    rc = vszip_ensure_scalars(ctx, total * sizeof(uint64_t));
    if (rc != VSZIP_OK) return rc;
    uint64_t *dev = static_cast<uint64_t *>(ctx->scratch);

    const int b_val = ((uint64_t)w * h > 2048ull * 1152ull) ? 2 : 1;  // calcSquaredErrorAndWeight :279
    int tmode = 0;
    if (temporal) tmode = frame_rate < 32 ? 1 : 2;
    if (b >= 4) {
        XArgs a;
        a.org = org3[0];
        a.rec = rec3[0];
        a.p1 = temporal ? prev1 : nullptr;
        a.p2 = (temporal && tmode == 2) ? prev2 : nullptr;
        a.stride = (int)stride3[0];
        a.w = (int)w;
        a.h = (int)h;
        a.b = (int)b;
        a.w_blk = (int)w_blk;
        a.h_blk = (int)h_blk;
        a.b_val = b_val;
        a.temporal = tmode;
        a.out = dev;
        if (bytes_per_sample == 1)
            hipLaunchKernelGGL((xpsnr_luma_kernel<uint8_t>), dim3(w_blk, h_blk), dim3(256), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL((xpsnr_luma_kernel<uint16_t>), dim3(w_blk, h_blk), dim3(256), 0, ctx->stream, a);
    }
    for (int c = 0; c < num_comps; ++c) {
        if (cn[c] == 0) continue;
        CArgs a;
        a.org = org3[c];
        a.rec = rec3[c];
        a.stride = (int)stride3[c];
        a.w = width3[c];
        a.h = height3[c];
        a.bx = (int)cbx[c];
        a.by = (int)cby[c];
        a.nbx = (int)cnbx[c];
        a.out = dev + coff[c];
        if (bytes_per_sample == 1)
            hipLaunchKernelGGL((xpsnr_sse_kernel<uint8_t>), dim3(cnbx[c], cnby[c]), dim3(256), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL((xpsnr_sse_kernel<uint16_t>), dim3(cnbx[c], cnby[c]), dim3(256), 0, ctx->stream, a);
    }
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    VSZIP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->scalars_host, dev, total * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t *res = static_cast<const uint64_t *>(ctx->scalars_host);

    std::vector<double> weights(n_luma);
    if (b >= 4) {
        double wsse_luma = 0.0;
        std::vector<double> sse_luma(n_luma);
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
                if (wh <= 640u * 480u) {  // :450-467
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
        wsse3[0] = wsse_luma <= 0.0 ? 0 : (uint64_t)std::trunc(wsse_luma * avg_act + 0.5);
    }
    for (int c = 0; c < num_comps; ++c) {
        if (b < 4) {
            wsse3[c] = res[coff[c]];
        } else if (c > 0) {
            double wsse_chroma = 0.0;
            for (size_t i = 0; i < cn[c]; ++i) wsse_chroma += (double)res[coff[c] + i] * weights[i];
            const double v = wsse_chroma * avg_act + 0.5;
            wsse3[c] = wsse_chroma <= 0.0 ? 0 : (uint64_t)(v < 0 ? 0 : v);
        }
    }
    return VSZIP_OK;
}

// getFrameXPSNR :370-374 on sqrt(f64(wsse)) (src/vapoursynth/xpsnr.zig:84-86). Host only.
VSZIP_EXPORT double vszip_xpsnr_value(uint64_t wsse, uint64_t width, uint64_t height, int depth) {
    const double sq = std::sqrt((double)wsse);
    if (sq < 1) return INFINITY;
    uint64_t maxerr = ((uint64_t)1 << depth) - 1;
    maxerr *= maxerr;
    return 10.0 * std::log10((double)(width * height * maxerr) / (sq * sq));
}

// getAvgXPSNR :359-368: the per-clip average printed by xpsnrFree. Host only.
VSZIP_EXPORT double vszip_xpsnr_average(double sum_wdist, double sum_xpsnr, uint64_t width, uint64_t height, int depth, uint64_t num_frames) {
    const double nf = (double)num_frames;
    uint64_t maxerr = ((uint64_t)1 << depth) - 1;
    maxerr *= maxerr;
    if (sum_wdist >= nf) {
        const double avg = sum_wdist / nf;
        return 10.0 * std::log10((double)(width * height * maxerr) / (avg * avg));
    }
    return sum_xpsnr / nf;
}
