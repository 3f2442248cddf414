// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of vszip.SSIMULACRA2's
// per-frame kernel (input: two linear-light RGBS frames; output: one f64 score).
//
// Follows (vszip v19.0.0):
//   src/filters/ssimulacra2.zig:22-37     skip_table (weights <= 0.01 pruned)
//   src/filters/ssimulacra2.zig:46-136    process (6 scales; downscale of LINEAR RGB; toXYB; maps)
//   src/filters/ssimulacra2.zig:138-209   downscale (2x2 box, edge clamp, sum order ((a+b)+c)+d)
//   src/filters/ssimulacra2.zig:211-245   multiply / addSquare
//   src/filters/ssimulacra2.zig:247-372   blur: 9-tap FIR, vertical then horizontal per row
//   src/filters/ssimulacra2.zig:374-472   toXYB (+ src/vcl.zig:40-81 cbrt)
//   src/filters/ssimulacra2.zig:480-628   ssimMap / edgeMap (f64 pooling)
//   src/filters/ssimulacra2.zig:630-774   score + the 108 weights
//
// The reference's SIMD width leaks into the arithmetic in two places, both restated
// with vec_size = 8 (x86_64_v3 / AVX2, the build the reference's CI tests):
//   * blurV uses a fused multiply-add in its vector body (columns < w - w % 8) and
//     an unfused multiply + add in the scalar tail (:318 vs :326);
//   * the f64 pooling sums keep 8 lane accumulators reduced in lane order (:551-552).
#include <algorithm>

#include "oracle_common.h"

namespace {

// vec_size of the reference build: 8 for the x86_64_v3 / haswell wheels and the CI build, 16 for the
// znver4 wheel (hatch_build.py:13-17). A test knob (vszo_ssim_set_vec) measures how far the other
// build's score moves: tests/test_oracle_goldens.py::test_ssimulacra2_vec_size_sensitivity.
constexpr int kVecMax = 16;
static int kVec = 8;

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

// ssimulacra2.zig:22-37
static Skip skip_of(int plane, int scale) {
    const int base = plane * 36 + scale * 6;
    const double p = 0.01;
    return {kWeight[base + 0] <= p && kWeight[base + 3] <= p, kWeight[base + 1] <= p && kWeight[base + 4] <= p, kWeight[base + 2] <= p && kWeight[base + 5] <= p};
}

// musl cbrtf (what Zig's std.math.cbrt(f32) ports) — used once, for K_D1.
static float cbrtf_musl(float x) {
    const unsigned B1 = 709958130;
    uint32_t ui;
    std::memcpy(&ui, &x, 4);
    uint32_t hx = ui & 0x7fffffffu;
    hx = hx / 3 + B1;
    ui = (ui & 0x80000000u) | hx;
    float tf;
    std::memcpy(&tf, &ui, 4);
    double T = tf, r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    r = T * T * T;
    T = T * ((double)x + x + r) / (x + r + r);
    return (float)T;
}

// src/vcl.zig:40-81 — VCL2 cbrt_f, one lane.
static inline float vcl_cbrt(float x) {
    const float one_third = 1.0f / 3.0f, four_third = 4.0f / 3.0f;
    const float xa = std::fabs(x);
    const float xa3 = one_third * xa;
    uint32_t m1;
    std::memcpy(&m1, &xa, 4);
    const uint32_t m2 = 0x54800000u - ((m1 >> 23) * 0x002AAAAAu);
    float a;
    std::memcpy(&a, &m2, 4);
    const bool underflow = m1 <= 0x00800000u;
    for (int it = 0; it < 3; ++it) {
        const float a2 = a * a;
        a = (four_third * a) - (xa3 * (a2 * a2));
    }
    const float a2 = a * a;
    a = a + (one_third * (a - (xa * (a2 * a2))));
    a = (a * a) * x;
    return underflow ? 0.0f : a;
}

struct XybConst {
    float m[9], bias, kd1;
    XybConst() {
        const float K_D0 = 0.0037930734f;
        const float K_M02 = 0.078f, K_M00 = 0.30f, K_M12 = 0.078f, K_M10 = 0.23f, K_M20 = 0.24342269f, K_M21 = 0.20476745f;
        m[0] = K_M00;
        m[1] = 1.0f - K_M02 - K_M00;
        m[2] = K_M02;
        m[3] = K_M10;
        m[4] = 1.0f - K_M12 - K_M10;
        m[5] = K_M12;
        m[6] = K_M20;
        m[7] = K_M21;
        m[8] = 1.0f - K_M20 - K_M21;
        bias = K_D0;
        kd1 = cbrtf_musl(K_D0);
    }
};
const XybConst kXyb;

// ssimulacra2.zig:392-472 (every pixel takes the same lane arithmetic)
static void to_xyb(const float* const src[3], float* const dst[3], int stride_s, int stride_d, int w, int h) {
    const float* m = kXyb.m;
    const float bias = kXyb.bias, kd1 = kXyb.kd1;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float r = src[0][(size_t)y * stride_s + x], g = src[1][(size_t)y * stride_s + x], b = src[2][(size_t)y * stride_s + x];
            const float ox = fmaf(m[0], r, fmaf(m[1], g, fmaf(m[2], b, bias)));
            const float oy = fmaf(m[3], r, fmaf(m[4], g, fmaf(m[5], b, bias)));
            const float oz = fmaf(m[6], r, fmaf(m[7], g, fmaf(m[8], b, bias)));
            const float cx = vcl_cbrt(std::fmax(ox, 0.0f)) - kd1;
            const float cy = vcl_cbrt(std::fmax(oy, 0.0f)) - kd1;
            const float cz = vcl_cbrt(std::fmax(oz, 0.0f)) - kd1;
            const float xv = 0.5f * (cx - cy);
            const float yv = 0.5f * (cx + cy);
            dst[0][(size_t)y * stride_d + x] = xv * 14.0f + 0.42f;
            dst[1][(size_t)y * stride_d + x] = yv + 0.01f;
            dst[2][(size_t)y * stride_d + x] = (cz - yv) + 0.55f;
        }
}

// ssimulacra2.zig:138-209
static void downscale(const float* src, float* dst, int src_stride, int in_w, int in_h) {
    const int out_w = (in_w + 1) / 2, out_h = (in_h + 1) / 2;
    for (int oy = 0; oy < out_h; ++oy)
        for (int ox = 0; ox < out_w; ++ox) {
            float sum = 0.0f;
            for (int iy = 0; iy < 2; ++iy)
                for (int ix = 0; ix < 2; ++ix) {
                    const int x = std::min(ox * 2 + ix, in_w - 1), y = std::min(oy * 2 + iy, in_h - 1);
                    sum += src[(size_t)y * src_stride + x];
                }
            dst[(size_t)oy * out_w + ox] = sum * 0.25f;
        }
}

const float kKernel[9] = {
    0.0076144188642501831054687500f, 0.0360749699175357818603515625f, 0.1095860823988914489746093750f,
    0.2134445458650588989257812500f, 0.2665599882602691650390625000f, 0.2134445458650588989257812500f,
    0.1095860823988914489746093750f, 0.0360749699175357818603515625f, 0.0076144188642501831054687500f,
};

// the asymmetric mirror shared with BoxBlur's CT path (ssimulacra2.zig:254,260,357,364)
static inline int tap_index(int k, int i, int n) {
    const int radius = 4;
    const int dist_from_end = n - 1 - i;
    if (k < radius) return (i < radius - k) ? std::min(radius - k - i, n - 1) : (i - radius + k);
    return (dist_from_end < k - radius) ? (i - std::min(k - radius - dist_from_end, i)) : (i - radius + k);
}

// ssimulacra2.zig:247-372
static void blur(const float* src, float* dst, int w, int h, std::vector<float>& tmp) {
    tmp.resize(w);
    const int wv = w - (w % kVec);
    for (int i = 0; i < h; ++i) {
        const float* rows[9];
        for (int k = 0; k < 9; ++k) rows[k] = src + (size_t)tap_index(k, i, h) * w;
        for (int j = 0; j < w; ++j) {
            float acc = 0.0f;
            if (j < wv) {
                for (int k = 0; k < 9; ++k) acc = fmaf(kKernel[k], rows[k][j], acc);  // :318 @mulAdd
            } else {
                for (int k = 0; k < 9; ++k) acc += kKernel[k] * rows[k][j];  // :326
            }
            tmp[j] = acc;
        }
        float* d = dst + (size_t)i * w;
        for (int j = 0; j < w; ++j) {
            float sum = 0.0f;
            for (int k = 0; k < 9; ++k) sum += kKernel[k] * tmp[tap_index(k, j, w)];  // :276 acc + k*s, unfused
            d[j] = sum;
        }
    }
}

struct LaneSum {
    double lane[kVecMax];
    double tail;
    LaneSum() : tail(0.0) {
        for (double& l : lane) l = 0.0;
    }
    inline void add(int x, int wv, double v) {
        if (x < wv)
            lane[x % kVec] += v;
        else
            tail += v;
    }
    double total() const {  // tail += @reduce(.Add, lanes), lanes summed in order
        double r = lane[0];
        for (int i = 1; i < kVec; ++i) r += lane[i];
        return tail + r;
    }
};

static inline double pow4(double y) {
    double x = y * y;
    return x * x;
}

// ssimulacra2.zig:480-553
static void ssim_map(const float* sq, const float* s12, const float* mu1, const float* mu2, int w, int h, double one_per_pixels, double out[2]) {
    LaneSum s0, s1;
    const int wv = w - (w % kVec);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const float m1 = mu1[i], m2 = mu2[i];
            const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2, md = m1 - m2;
            const double num_m = (double)fmaf(md, -md, 1.0f);
            const double num_s = (double)fmaf(s12[i] - m12, 2.0f, 0.0009f);
            const double denom_s = (double)(sq[i] - 2.0f * s12[i] - m11 - m22 + 0.0009f);
            const double d1 = std::max(1.0 - ((num_m * num_s) / denom_s), 0.0);
            s0.add(x, wv, d1);
            s1.add(x, wv, pow4(d1));
        }
    out[0] = one_per_pixels * s0.total();
    out[1] = std::sqrt(std::sqrt(one_per_pixels * s1.total()));
}

// ssimulacra2.zig:555-628
static void edge_map(const float* im1, const float* im2, const float* mu1, const float* mu2, int w, int h, double one_per_pixels, double out[4]) {
    LaneSum a0, a1, d0, dd1;
    const int wv = w - (w % kVec);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const double n2 = (double)std::fabs(im2[i] - mu2[i]);
            const double n1 = (double)std::fabs(im1[i] - mu1[i]);
            const double d1 = (1.0 + n2) / (1.0 + n1) - 1.0;
            const double art = std::max(d1, 0.0), det = std::max(-d1, 0.0);
            a0.add(x, wv, art);
            a1.add(x, wv, pow4(art));
            d0.add(x, wv, det);
            dd1.add(x, wv, pow4(det));
        }
    out[0] = one_per_pixels * a0.total();
    out[1] = std::sqrt(std::sqrt(one_per_pixels * a1.total()));
    out[2] = one_per_pixels * d0.total();
    out[3] = std::sqrt(std::sqrt(one_per_pixels * dd1.total()));
}

// ssimulacra2.zig:630-663
static double score(const double ssim[6][6], const double edge[6][12]) {
    double s = 0.0;
    int i = 0;
    for (int plane = 0; plane < 3; ++plane)
        for (int sc = 0; sc < 6; ++sc)
            for (int n = 0; n < 2; ++n) {
                s = std::fma(kWeight[i++], std::fabs(ssim[sc][plane * 2 + n]), s);
                s = std::fma(kWeight[i++], std::fabs(edge[sc][plane * 4 + n]), s);
                s = std::fma(kWeight[i++], std::fabs(edge[sc][plane * 4 + n + 2]), s);
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

// Two frames of three f32 planes each (linear-light RGB), strides in elements.
// Optionally returns the 6x6 / 6x12 per-scale averages (debugging aid for the GPU tests).
VSZO_API double vszo_ssimulacra2(const float* const ref[3], const float* const dis[3], ptrdiff_t stride, int w, int h, double* avg_ssim_out, double* avg_edge_out) {
    std::vector<float> r1[3], r2[3], x1[3], x2[3], t3, sq, s12, mu1, tmp;
    double avg_ssim[6][6], avg_edge[6][12];
    int w2 = w, h2 = h;
    const float* cur1[3] = {ref[0], ref[1], ref[2]};
    const float* cur2[3] = {dis[0], dis[1], dis[2]};
    int cstride = (int)stride;
    for (int scale = 0; scale < 6; ++scale) {
        if (scale > 0) {
            const int nw = (w2 + 1) / 2, nh = (h2 + 1) / 2;
            for (int p = 0; p < 3; ++p) {
                std::vector<float> n1((size_t)nw * nh), n2((size_t)nw * nh);
                downscale(cur1[p], n1.data(), cstride, w2, h2);
                downscale(cur2[p], n2.data(), cstride, w2, h2);
                r1[p].swap(n1);
                r2[p].swap(n2);
            }
            for (int p = 0; p < 3; ++p) {
                cur1[p] = r1[p].data();
                cur2[p] = r2[p].data();
            }
            w2 = nw;
            h2 = nh;
            cstride = w2;
        }
        const size_t n = (size_t)w2 * h2;
        const double one_per_pixels = 1.0 / (double)((uint32_t)w2 * (uint32_t)h2);
        float* d1[3];
        float* d2[3];
        for (int p = 0; p < 3; ++p) {
            x1[p].resize(n);
            x2[p].resize(n);
            d1[p] = x1[p].data();
            d2[p] = x2[p].data();
        }
        to_xyb(cur1, d1, cstride, w2, w2, h2);
        to_xyb(cur2, d2, cstride, w2, w2, h2);
        t3.resize(n);
        sq.resize(n);
        s12.resize(n);
        mu1.resize(n);
        for (int plane = 0; plane < 3; ++plane) {
            const Skip sk = skip_of(plane, scale);
            double* as = &avg_ssim[scale][plane * 2];
            double* ae = &avg_edge[scale][plane * 4];
            as[0] = as[1] = 0.0;
            ae[0] = ae[1] = ae[2] = ae[3] = 0.0;
            if (sk.all()) continue;
            const float* a = x1[plane].data();
            const float* b = x2[plane].data();
            if (!sk.ssim) {
                for (size_t i = 0; i < n; ++i) t3[i] = a[i] * b[i];  // multiply :211
                blur(t3.data(), s12.data(), w2, h2, tmp);
                for (size_t i = 0; i < n; ++i) {
                    const float v = a[i] + b[i];  // addSquare :228
                    t3[i] = v * v;
                }
                blur(t3.data(), sq.data(), w2, h2, tmp);
            }
            blur(a, mu1.data(), w2, h2, tmp);
            blur(b, t3.data(), w2, h2, tmp);  // mu2
            if (!sk.ssim) ssim_map(sq.data(), s12.data(), mu1.data(), t3.data(), w2, h2, one_per_pixels, as);
            if (!sk.artifact || !sk.detail) edge_map(a, b, mu1.data(), t3.data(), w2, h2, one_per_pixels, ae);
        }
    }
    if (avg_ssim_out) std::memcpy(avg_ssim_out, avg_ssim, sizeof avg_ssim);
    if (avg_edge_out) std::memcpy(avg_edge_out, avg_edge, sizeof avg_edge);
    return score(avg_ssim, avg_edge);
}

// Exposed pieces so the GPU tests can localise a mismatch.
VSZO_API int vszo_ssim_set_vec(int v) {  // test knob: 8 (default) or 16; returns the previous value
    const int old = kVec;
    if (v == 8 || v == 16) kVec = v;
    return old;
}
VSZO_API void vszo_ssim_to_xyb(const float* const src[3], float* const dst[3], int w, int h) { to_xyb(src, dst, w, w, w, h); }
VSZO_API void vszo_ssim_blur(const float* src, float* dst, int w, int h) {
    std::vector<float> tmp;
    blur(src, dst, w, h, tmp);
}
VSZO_API void vszo_ssim_downscale(const float* src, float* dst, int w, int h) { downscale(src, dst, w, w, h); }
