// Device-free entry points of include/vszip_hip.h — plain C++ (no HIP header), compiled into libvszip_hip.so by
// hipcc and, unchanged, into the sanitizer stub of tests/sanitize (g++ -fsanitize=address,undefined), so that the
// plugin's create-time paths run the real parameter derivation under ASan/UBSan without a GPU.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/vszip_hip.h"

#define VSZIP_EXPORT extern "C" __attribute__((visibility("default")))

// bilateralCreate's per-plane derivation, src/vapoursynth/bilateral.zig:104-199 (host only).
VSZIP_EXPORT int vszip_bilateral_derive(const double *sigmaS_in, int n_sigmaS, const double *sigmaR, const int *algorithm_in, const int *pbficnum_in,
                                        int is_yuv, int subsampling_w, int subsampling_h, const int *planes_in, vszip_bilateral_cfg *out) {
    if (!sigmaR || !algorithm_in || !pbficnum_in || !planes_in || !out || n_sigmaS < 0 || n_sigmaS > 3) return VSZIP_ERR_ARG;
    double sS[3];
    for (int i = 0; i < 3; ++i) {
        if (i < n_sigmaS)
            sS[i] = sigmaS_in[i];
        else if (i == 0)
            sS[0] = 3;
        else if (i == 1 && is_yuv && subsampling_h != 0 && subsampling_w != 0)
            sS[1] = sS[0] / std::sqrt((double)((1u << subsampling_h) * (1u << subsampling_w)));
        else
            sS[i] = sS[i - 1];
        if (sS[i] < 0) return VSZIP_ERR_ARG;  // "Invalid \"sigmaS\" assigned, must be non-negative float number"
    }
    for (int i = 0; i < 3; ++i) {
        vszip_bilateral_cfg &c = out[i];
        c.sigmaS = sS[i];
        c.sigmaR = sigmaR[i];
        c.process = planes_in[i] && !(sS[i] == 0 || sigmaR[i] == 0);
        c.algorithm = algorithm_in[i];
        c.pbficnum = pbficnum_in[i];
        c.radius = c.step = c.samples = 0;
        c.gs_lut = c.gr_lut = nullptr;
    }
    for (int i = 0; i < 3; ++i)
        if (out[i].pbficnum == 1) return VSZIP_ERR_ARG;  // "must be integer ranges in [0,256] except 1"
    for (int i = 0; i < 3; ++i) {
        vszip_bilateral_cfg &c = out[i];
        if (c.process && c.pbficnum == 0) {
            if (c.sigmaR >= 0.08)
                c.pbficnum = 4;
            else if (c.sigmaR >= 0.015)
                c.pbficnum = std::min(16, (int)std::trunc(4 * 0.08 / c.sigmaR + 0.5));
            else
                c.pbficnum = std::min(32, (int)std::trunc(16 * 0.015 / c.sigmaR + 0.5));
            if (i > 0 && is_yuv && (c.pbficnum % 2 == 0) && c.pbficnum < 256) c.pbficnum += 1;
        }
    }
    for (int i = 0; i < 3; ++i) {
        vszip_bilateral_cfg &c = out[i];
        if (!c.process) continue;
        const int orad = std::max((int)std::trunc(c.sigmaS * 2 + 0.5), 1);
        c.step = orad < 4 ? 1 : (orad < 8 ? 2 : 3);
        c.samples = 1;
        c.radius = 1 + (c.samples - 1) * c.step;
        while (orad * 2 > c.radius * 3) {
            c.samples += 1;
            c.radius = 1 + (c.samples - 1) * c.step;
            if (c.radius >= orad && c.samples > 2) {
                c.samples -= 1;
                c.radius = 1 + (c.samples - 1) * c.step;
                break;
            }
        }
        if (c.algorithm <= 0)
            c.algorithm = (c.step == 1) ? 2 : ((c.sigmaR < 0.08 && c.samples < 5) ? 2 : ((4 * c.samples * c.samples <= 15 * c.pbficnum) ? 2 : 1));
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

// ---------------------------------------------------------------------------------------------------------
// zimg's resampling table for one axis (round 3: the YUV colour pre-stage of SSIMULACRA2) — what
// `resize.Bicubic(format=RGBS)` in hz.toRGBS (src/helper.zig:225-243) uses to bring a subsampled chroma plane to
// 4:4:4. zimg is third-party and not in the reference tree; this restates its published filter construction
// (oracle/vs_host.py::zimg_filter is the twin the tests compare with, itself pinned by the reference's goldens):
// per output sample the window `pos = (i + 0.5) / scale + shift`, `filter_size = ceil(2 * support)` taps starting at
// `floor(pos - filter_size / 2 + 0.5)`, weights normalised, taps outside the line reflected about the edge (edge
// sample repeated) and ADDED to the sample they land on, rows trimmed to their non-zero span and aligned to the common
// width. Upscaling only (scale >= 1, support 2): at most 4 taps. coef4[4 * i + k] multiplies sample left[i] + k; the
// caller clamps left[i] + k to src_dim - 1 for lines shorter than 4 (those coefficients are 0).
// ---------------------------------------------------------------------------------------------------------
namespace {
double bicubic_weight(double x, double b, double c) {
    x = std::fabs(x);
    const double p0 = (6.0 - 2.0 * b) / 6.0, p2 = (-18.0 + 12.0 * b + 6.0 * c) / 6.0, p3 = (12.0 - 9.0 * b - 6.0 * c) / 6.0;
    const double q0 = (8.0 * b + 24.0 * c) / 6.0, q1 = (-12.0 * b - 48.0 * c) / 6.0, q2 = (6.0 * b + 30.0 * c) / 6.0, q3 = (-b - 6.0 * c) / 6.0;
    if (x < 1.0) return p0 + p2 * x * x + p3 * x * x * x;
    if (x < 2.0) return q0 + q1 * x + q2 * x * x + q3 * x * x * x;
    return 0.0;
}
}  // namespace

VSZIP_EXPORT int vszip_resample_table(int src_dim, int dst_dim, double shift, int32_t *left, float *coef4) {
    if (src_dim <= 0 || dst_dim < src_dim || !left || !coef4) return VSZIP_ERR_ARG;
    const double scale = (double)dst_dim / (double)src_dim;
    constexpr int kTaps = 4;  // ceil(2 * support), support 2, no widening when upscaling
    std::vector<double> row((size_t)src_dim);
    for (int i = 0; i < dst_dim; ++i) {
        std::fill(row.begin(), row.end(), 0.0);
        const double pos = (i + 0.5) / scale + shift;
        const double begin = std::floor(pos - kTaps / 2.0 + 0.5) + 0.5;
        double w[kTaps], total = 0.0;
        for (int k = 0; k < kTaps; ++k) {
            w[k] = bicubic_weight(begin + k - pos, 0.0, 0.5);  // VapourSynth's resize.Bicubic defaults: b = 0, c = 0.5
            total += w[k];
        }
        for (int k = 0; k < kTaps; ++k) {
            const double xpos = begin + k;
            double real = xpos < 0.0 ? -xpos : (xpos >= src_dim ? 2.0 * src_dim - xpos : xpos);
            real = std::min(std::max(real, 0.0), std::nextafter((double)src_dim, -INFINITY));
            row[(size_t)std::floor(real)] += w[k] / total;
        }
        int first = 0, last = 0;
        bool any = false;
        for (int j = 0; j < src_dim; ++j)
            if (row[j] != 0.0) {
                if (!any) first = j;
                last = j;
                any = true;
            }
        if (!any || last - first + 1 > kTaps) return VSZIP_ERR_UNSUPPORTED;
        // zimg aligns every row to the table's common width (4 wherever the line has 4 samples)
        const int width = std::min(kTaps, src_dim);
        const int l = std::min(first, src_dim - width);
        left[i] = l;
        for (int k = 0; k < kTaps; ++k) coef4[4 * i + k] = (l + k < src_dim) ? (float)row[(size_t)(l + k)] : 0.0f;
    }
    return VSZIP_OK;
}
