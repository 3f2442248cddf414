// Device-free entry points of include/vszip_hip.h — plain C++ (no HIP header), compiled into libvszip_hip.so by
// hipcc and, unchanged, into the sanitizer stub of tests/sanitize (g++ -fsanitize=address,undefined), so that the
// plugin's create-time paths run the real parameter derivation under ASan/UBSan without a GPU.
#include <algorithm>
#include <cmath>
#include <cstdint>

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
