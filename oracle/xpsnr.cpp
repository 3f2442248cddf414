// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of vszip.XPSNR's
// per-frame kernel: weighted SSE per plane (u64) from a reference/distorted frame pair
// plus up to two previous reference luma planes.
//
// Follows (vszip v19.0.0):
//   src/filters/xpsnr.zig:28-64      highds (6x6 high-pass on the 2x-decimated grid)
//   src/filters/xpsnr.zig:66-109     diff1st / diff2nd (temporal activity, large frames)
//   src/filters/xpsnr.zig:111-170    tempDiff1 / tempDiff2 (temporal activity, small frames)
//   src/filters/xpsnr.zig:174-212    spatialAct (3x3 Laplacian)
//   src/filters/xpsnr.zig:214-251    calcSquaredError
//   src/filters/xpsnr.zig:253-357    calcSquaredErrorAndWeight
//   src/filters/xpsnr.zig:359-374    getAvgXPSNR / getFrameXPSNR
//   src/filters/xpsnr.zig:376-524    getWSSE
//   src/vapoursynth/xpsnr.zig:72-101 per-frame combination (sqrt, props)
#include <algorithm>

#include "oracle_common.h"

namespace {

template <typename T>
static uint64_t highds(int x_act, int y_act, int w_act, int h_act, const T* o_m0, ptrdiff_t o) {
    uint64_t sa = 0;
    for (int y = y_act; y < h_act; y += 2)
        for (int x = x_act; x < w_act; x += 2) {
            const T* p = o_m0 + (ptrdiff_t)y * o + x;
            auto g = [&](ptrdiff_t d) -> int32_t { return (int32_t)p[d]; };
            const int32_t f = 12 * (g(0) + g(1) + g(o) + g(o + 1))
                - 3 * (g(-o) + g(-o + 1) + g(2 * o) + g(2 * o + 1))
                - 3 * (g(-1) + g(2) + g(o - 1) + g(o + 2))
                - 2 * (g(-o - 1) + g(-o + 2) + g(2 * o - 1) + g(2 * o + 2))
                - (g(-2 * o - 1) + g(-2 * o) + g(-2 * o + 1) + g(-2 * o + 2)
                   + g(3 * o - 1) + g(3 * o) + g(3 * o + 1) + g(3 * o + 2)
                   + g(-o - 2) + g(-2) + g(o - 2) + g(2 * o - 2)
                   + g(-o + 3) + g(3) + g(o + 3) + g(2 * o + 3));
            sa += (uint64_t)std::abs(f);
        }
    return sa;
}

// diff1st / diff2nd: 2x2 block sums; missing previous frames count as zero.
template <typename T>
static uint64_t diff_blocks(int w_act, int h_act, const T* m0, const T* p1, const T* p2, ptrdiff_t o, bool second) {
    uint64_t ta = 0;
    for (int y = 0; y < h_act; y += 2)
        for (int x = 0; x < w_act; x += 2) {
            auto s4 = [&](const T* b) -> int32_t { return (int32_t)b[(ptrdiff_t)y * o + x] + (int32_t)b[(ptrdiff_t)y * o + x + 1] + (int32_t)b[(ptrdiff_t)(y + 1) * o + x] + (int32_t)b[(ptrdiff_t)(y + 1) * o + x + 1]; };
            int32_t t = s4(m0);
            if (!second) {
                if (p1) t -= s4(p1);
            } else {
                if (p1) t -= 2 * s4(p1);
                if (p2) t += s4(p2);
            }
            ta += (uint64_t)std::abs(t);
        }
    return ta * 2;  // XPSNR_GAMMA
}

// tempDiff1 / tempDiff2: per pixel
template <typename T>
static uint64_t temp_diff(int bw, int bh, const T* m0, const T* p1, const T* p2, ptrdiff_t o, bool second) {
    uint64_t ta = 0;
    for (int y = 0; y < bh; ++y)
        for (int x = 0; x < bw; ++x) {
            int32_t t = (int32_t)m0[(ptrdiff_t)y * o + x];
            if (!second) {
                if (p1) t -= (int32_t)p1[(ptrdiff_t)y * o + x];
            } else {
                if (p1) t -= 2 * (int32_t)p1[(ptrdiff_t)y * o + x];
                if (p2) t += (int32_t)p2[(ptrdiff_t)y * o + x];
            }
            ta += 2ull * (uint64_t)std::abs(t);
        }
    return ta;
}

template <typename T>
static uint64_t spatial_act(const T* pic, ptrdiff_t o, int x0, int x1, int y0, int y1) {
    uint64_t sa = 0;
    for (int y = y0; y < y1; ++y) {
        const T* rm = pic + (ptrdiff_t)(y - 1) * o;
        const T* rc = pic + (ptrdiff_t)y * o;
        const T* rp = pic + (ptrdiff_t)(y + 1) * o;
        for (int x = x0; x < x1; ++x) {
            const int32_t f = 12 * (int32_t)rc[x] - 2 * ((int32_t)rc[x - 1] + (int32_t)rc[x + 1] + (int32_t)rm[x] + (int32_t)rp[x]) -
                              ((int32_t)rm[x - 1] + (int32_t)rm[x + 1] + (int32_t)rp[x - 1] + (int32_t)rp[x + 1]);
            sa += (uint64_t)std::abs(f);
        }
    }
    return sa;
}

template <typename T>
static uint64_t sse_block(const T* org, const T* rec, ptrdiff_t stride, int bw, int bh) {
    uint64_t sse = 0;
    for (int y = 0; y < bh; ++y)
        for (int x = 0; x < bw; ++x) {
            const int64_t e = (int64_t)org[(ptrdiff_t)y * stride + x] - (int64_t)rec[(ptrdiff_t)y * stride + x];
            sse += (uint64_t)(e * e);
        }
    return sse;
}

// xpsnr.zig:253-357
template <typename T>
static double sse_and_weight(const T* pic_org, ptrdiff_t stride, const T* pic_rec, const T* prv1, const T* prv2, int ox, int oy, int bw, int bh,
                             int depth, uint32_t frame_rate, double* ms_act, int w0, int h0, bool temporal) {
    const ptrdiff_t off = (ptrdiff_t)oy * stride + ox;
    const T* o_m0 = pic_org + off;
    const T* p_m1 = prv1 ? prv1 + off : nullptr;
    const T* p_m2 = prv2 ? prv2 + off : nullptr;
    const T* r_m0 = pic_rec + off;
    const int b_val = ((uint64_t)w0 * (uint64_t)h0 > 2048ull * 1152ull) ? 2 : 1;
    const int x_act = ox > 0 ? 0 : b_val;
    const int y_act = oy > 0 ? 0 : b_val;
    const int w_act = (ox + bw < w0) ? bw : bw - b_val;
    const int h_act = (oy + bh < h0) ? bh : bh - b_val;
    const double sse = (double)sse_block<T>(o_m0, r_m0, stride, bw, bh);
    uint64_t sa = 0, ta = 0;
    if (w_act <= x_act || h_act <= y_act) return sse;
    if (b_val > 1) {
        if (w_act > 12) sa = highds<T>(x_act, y_act, w_act, h_act, o_m0, stride);
    } else {
        sa = spatial_act<T>(pic_org, stride, ox + x_act, ox + w_act, oy + y_act, oy + h_act);
    }
    *ms_act = (double)sa / ((double)(w_act - x_act) * (double)(h_act - y_act));
    if (temporal) {
        const bool second = !(frame_rate < 32);
        // second-order with p1 absent ignores p2 (diff2nd(false,false) :327)
        const T* q1 = p_m1;
        const T* q2 = (second && p_m1) ? p_m2 : nullptr;
        if (b_val > 1)
            ta = diff_blocks<T>(bw, bh, o_m0, q1, q2, stride, second);
        else
            ta = temp_diff<T>(bw, bh, o_m0, q1, q2, stride, second);
        *ms_act += (double)ta / ((double)bw * (double)bh);
    }
    const double sft = (double)((size_t)1 << (depth - 6));
    if (*ms_act < sft) *ms_act = sft;
    *ms_act *= *ms_act;
    return sse;
}

// xpsnr.zig:376-524
template <typename T>
static void get_wsse(const T* const org[3], const T* const rec[3], const T* prv1, const T* prv2, uint64_t wsse64[3], const int width[3], const int height[3],
                     const ptrdiff_t strides[3], int depth, int num_comps, uint32_t frame_rate, bool temporal) {
    const uint32_t w = (uint32_t)width[0], h = (uint32_t)height[0];
    const uint32_t wh = w * h;
    const double r = (double)wh / (3840.0 * 2160.0);
    const double bq = 32.0 * std::sqrt(r) + 0.5;
    const uint32_t b = (uint32_t)(bq < 0 ? 0 : bq) * 4;  // lossyCast(u32, ...) * 4
    const uint32_t w_blk = b >= 4 ? (w + b - 1) / b : 0, h_blk = b >= 4 ? (h + b - 1) / b : 0;
    const uint32_t sft = 1u << (2 * depth - 9);
    const double avg_act = std::sqrt(16.0 * (double)sft / std::sqrt(std::max(0.00001, r)));
    std::vector<double> sse_luma((size_t)w_blk * h_blk), weights((size_t)w_blk * h_blk);
    if (b >= 4) {
        const ptrdiff_t stride = strides[0];
        double wsse_luma = 0.0;
        size_t idx = 0;
        for (uint32_t y = 0; y < h; y += b) {
            const uint32_t bh = (y + b > h) ? (h - y) : b;
            for (uint32_t x = 0; x < w; x += b, ++idx) {
                const uint32_t bw = (x + b > w) ? (w - x) : b;
                double ms_act = 1.0, ms_act_prev = 0.0;
                sse_luma[idx] = sse_and_weight<T>(org[0], stride, rec[0], prv1, prv2, (int)x, (int)y, (int)bw, (int)bh, depth, frame_rate, &ms_act, (int)w, (int)h, temporal);
                weights[idx] = 1.0 / std::sqrt(ms_act);
                if (wh <= 640u * 480u) {
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
        wsse64[0] = wsse_luma <= 0.0 ? 0 : (uint64_t)std::trunc(wsse_luma * avg_act + 0.5);
    }
    for (int c = 0; c < num_comps; ++c) {
        const ptrdiff_t stride = strides[c];
        const uint32_t w_pln = (uint32_t)width[c], h_pln = (uint32_t)height[c];
        if (b < 4) {
            wsse64[c] = sse_block<T>(org[c], rec[c], stride, (int)w_pln, (int)h_pln);
        } else if (c > 0) {
            const uint32_t bx = (b * w_pln) / w, by = (b * h_pln) / h;
            double wsse_chroma = 0.0;
            size_t idx = 0;
            for (uint32_t y = 0; y < h_pln; y += by) {
                const uint32_t bh = (y + by > h_pln) ? (h_pln - y) : by;
                for (uint32_t x = 0; x < w_pln; x += bx, ++idx) {
                    const uint32_t bw = (x + bx > w_pln) ? (w_pln - x) : bx;
                    const uint64_t e = sse_block<T>(org[c] + (ptrdiff_t)y * stride + x, rec[c] + (ptrdiff_t)y * stride + x, stride, (int)bw, (int)bh);
                    wsse_chroma += (double)e * weights[idx];
                }
            }
            const double v = wsse_chroma * avg_act + 0.5;
            wsse64[c] = wsse_chroma <= 0.0 ? 0 : (uint64_t)(v < 0 ? 0 : v);  // lossyCast
        }
    }
}

}  // namespace

// bytes_per_sample 1 or 2; prv1/prv2 may be NULL; strides in elements.
VSZO_API int vszo_xpsnr_wsse(int bytes_per_sample, const void* const org[3], const void* const rec[3], const void* prv1, const void* prv2, uint64_t wsse64[3],
                             const int width[3], const int height[3], const ptrdiff_t strides[3], int depth, int num_comps, uint32_t frame_rate, int temporal) {
    wsse64[0] = wsse64[1] = wsse64[2] = 0;
    if (bytes_per_sample == 1) {
        const uint8_t* o[3] = {(const uint8_t*)org[0], (const uint8_t*)org[1], (const uint8_t*)org[2]};
        const uint8_t* r[3] = {(const uint8_t*)rec[0], (const uint8_t*)rec[1], (const uint8_t*)rec[2]};
        get_wsse<uint8_t>(o, r, (const uint8_t*)prv1, (const uint8_t*)prv2, wsse64, width, height, strides, depth, num_comps, frame_rate, temporal != 0);
        return 0;
    }
    if (bytes_per_sample == 2) {
        const uint16_t* o[3] = {(const uint16_t*)org[0], (const uint16_t*)org[1], (const uint16_t*)org[2]};
        const uint16_t* r[3] = {(const uint16_t*)rec[0], (const uint16_t*)rec[1], (const uint16_t*)rec[2]};
        get_wsse<uint16_t>(o, r, (const uint16_t*)prv1, (const uint16_t*)prv2, wsse64, width, height, strides, depth, num_comps, frame_rate, temporal != 0);
        return 0;
    }
    return -1;
}

// xpsnr.zig:370-374 with sqrt_wsse = sqrt(f64(wsse64)) (src/vapoursynth/xpsnr.zig:84-86)
VSZO_API double vszo_xpsnr_frame(uint64_t wsse, uint64_t width, uint64_t height, int depth) {
    const double sq = std::sqrt((double)wsse);
    if (sq < 1) return INFINITY;
    uint64_t maxerr = ((uint64_t)1 << depth) - 1;
    maxerr *= maxerr;
    const double num = (double)(width * height * maxerr);
    return 10.0 * std::log10(num / (sq * sq));
}

// xpsnr.zig:359-368
VSZO_API double vszo_xpsnr_avg(double sum_wdist, double sum_xpsnr, uint64_t width, uint64_t height, int depth, uint64_t num_frames) {
    const double nf = (double)num_frames;
    uint64_t maxerr = ((uint64_t)1 << depth) - 1;
    maxerr *= maxerr;
    if (sum_wdist >= nf) {
        const double avg = sum_wdist / nf;
        return 10.0 * std::log10((double)(width * height * maxerr) / (avg * avg));
    }
    return sum_xpsnr / nf;
}
