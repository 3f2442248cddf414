// TEST INFRASTRUCTURE ONLY (see oracle_common.h). CPU restatement of vszip.EEDI3 / EEDI3H
// (32-bit float planes only, like the reference).
//
// Follows (vszip v19.0.0):
//   src/vapoursynth/eedi3.zig:26-140     processPlane (field copy, 4 rotating padded rows, per-line interp, vcheck)
//   src/vapoursynth/eedi3.zig:220-246    EEDI3H = transpose -> vertical pipeline -> transpose
//   src/vapoursynth/eedi3.zig:321-333,465-473  parameter defaults and scaling
//   src/filters/eedi3.zig:88-116         reflectRow / srcCol / mirrorPad / fillPaddedRow
//   src/filters/eedi3.zig:285-304        buildBmask
//   src/filters/eedi3.zig:311-592        costBlockDirect + interpLine (cost volume, Viterbi DP, backtrack, cubic)
//   src/filters/eedi3.zig:602-904        computeHpRow + interpLineHP
//   src/filters/eedi3.zig:915-1046       vcheckLine
//
// The line buffers persist across lines exactly as the reference's 4 rotating
// scratch rows do (mirrorPad reads previously written padding when w < pad_h); they
// start zeroed here, where the reference starts from uninitialised memory.
#include <algorithm>
#include <cfloat>
#include <cstdlib>

#include "oracle_common.h"

namespace {

constexpr int kPadH = 96;  // alignForward(2*40 + 3 + 8, 8), eedi3.zig:80
const float kFltMax09 = FLT_MAX * 0.9f;

struct Params {
    int mdis, nrad, vcheck;
    float alpha, beta, gamma, one_minus_ab;
    float vthresh2, rcp0, rcp1, rcp2;
    bool hp, dh;
};

static inline int reflect_row(int y, int h) {  // eedi3.zig:88-96
    if (h == 1) return 0;
    int r = y;
    while (r < 0 || r >= h) {
        if (r < 0) r = -r;
        if (r >= h) r = 2 * (h - 1) - r;
    }
    return r;
}
static inline int src_col(bool dh, int off, int n_src) {  // :102-104
    return dh ? reflect_row(off, 2 * n_src) / 2 : reflect_row(off, n_src);
}
static void fill_padded_row(float* buf, const float* src, int w) {  // :107-116
    std::memcpy(buf + kPadH, src, sizeof(float) * w);
    for (int i = 0; i < kPadH; ++i) buf[kPadH + w + i] = buf[kPadH + w - 2 - i];
    for (int i = 0; i < kPadH; ++i) buf[i] = buf[2 * kPadH - i];
}

static void build_bmask(uint8_t* bmask, const uint8_t* maskp, int w, int mdis) {  // :285-304
    const int minmdis = std::min(w, mdis);
    long last = -666999;
    for (int x = 0; x < minmdis; ++x)
        if (maskp[x] != 0) last = (long)x + mdis;
    for (int x = 0; x < w - minmdis; ++x) {
        if (maskp[x + mdis] != 0) last = (long)x + mdis * 2;
        bmask[x] = (long)x <= last;
    }
    for (int x = w - minmdis; x < w; ++x) bmask[x] = (long)x <= last;
}

#define P(a) ((a) + kPadH)

// eedi3.zig:349-592
static void interp_line(const float* r3p, const float* r1p, const float* r1n, const float* r3n, float* dst, int8_t* pbackt, int* fpath, float* t_base,
                        float* t_costs, int* dmap, int stride, int w, const Params& d, const uint8_t* bmask) {
    const int mdis = d.mdis, nrad = d.nrad, tpitch = 2 * mdis + 1;
    if (w == 0) return;
    if (bmask) {
        bool any = false;
        for (int x = 0; x < w; ++x) any = any || bmask[x];
        if (!any) {
            for (int x = 0; x < w; ++x) {
                dmap[x] = 0;
                dst[x] = 0.5625f * (r1p[P(x)] + r1n[P(x)]) - 0.0625f * (r3p[P(x)] + r3n[P(x)]);
            }
            return;
        }
    }
    for (int u = -mdis; u <= mdis; ++u) {
        const int two_u = 2 * u;
        const int u_lo = std::min(u, std::min(0, two_u)), u_hi = std::max(u, std::max(0, two_u));
        for (int j = u_lo - nrad; j < w + u_hi + nrad + 1; ++j)  // :405-425 (vector-rounded range there)
            t_base[P(j)] = std::fabs(r3p[P(j)] - r1p[P(j - two_u)]) + std::fabs(r1p[P(j)] - r1n[P(j - two_u)]) + std::fabs(r1n[P(j)] - r3n[P(j - two_u)]);
        float* tc = t_costs + (size_t)(mdis + u) * stride;
        const float beta_abs_u = d.beta * (float)std::abs(u);
        for (int x = 0; x < w; ++x) {  // costBlockDirect :311-347 / scalar tail :443-456
            // (masked-off vector blocks are skipped there; their costs are never read)
            float sw0 = 0, sw1 = 0, sw2 = 0;
            for (int k = -nrad; k <= nrad; ++k) {
                sw1 += t_base[P(x + k)];
                sw0 += t_base[P(x + u + k)];
                sw2 += t_base[P(x + two_u + k)];
            }
            const float ip = (r1p[P(x + u)] + r1n[P(x - u)]) * 0.5f;
            const float v = std::fabs(r1p[P(x)] - ip) + std::fabs(r1n[P(x)] - ip);
            tc[x] = d.alpha * (sw0 + sw1 + sw2) + beta_abs_u + d.one_minus_ab * v;
        }
    }
    std::vector<float> pbuf[2];
    pbuf[0].assign(tpitch + 2, kFltMax09);
    pbuf[1].assign(tpitch + 2, kFltMax09);
    int ping = 0;
    for (int ui = 0; ui < tpitch; ++ui) pbuf[ping][ui + 1] = t_costs[(size_t)ui * stride + 0];
    for (int x = 1; x < w; ++x) {  // :468-555
        int8_t* piT = pbackt + (size_t)(x - 1) * tpitch;
        const int pong = ping ^ 1;
        if (bmask && !bmask[x]) {
            if (x == 1) {
                for (int ui = 0; ui < tpitch; ++ui) pbuf[pong][ui + 1] = t_costs[(size_t)ui * stride + x];
                std::memset(piT, 0, tpitch);
            } else {
                pbuf[pong] = pbuf[ping];
                std::memcpy(piT, pbackt + (size_t)(x - 2) * tpitch, tpitch);
            }
            ping = pong;
            continue;
        }
        const float* p = pbuf[ping].data();
        float* po = pbuf[pong].data();
        for (int ui = 0; ui < tpitch; ++ui) {
            const float left = p[ui] + d.gamma, cent = p[ui + 1], right = p[ui + 2] + d.gamma;
            float bval = cent;
            int8_t bd = 0;
            if (left < bval) {
                bval = left;
                bd = -1;
            }
            if (right < bval) {
                bval = right;
                bd = 1;
            }
            po[ui + 1] = std::fmin(bval + t_costs[(size_t)ui * stride + x], kFltMax09);
            piT[ui] = bd;
        }
        ping = pong;
    }
    fpath[w - 1] = 0;
    for (int bx = w - 2; bx >= 0; --bx) fpath[bx] = fpath[bx + 1] + pbackt[(size_t)bx * tpitch + (mdis + fpath[bx + 1])];
    if (bmask)
        for (int x = 0; x < w; ++x)
            if (!bmask[x]) fpath[x] = 0;
    for (int x = 0; x < w; ++x) {  // :577-591
        const int dir = fpath[x], ad = std::abs(dir);
        dmap[x] = dir;
        dst[x] = (x >= ad * 3 && x + ad * 3 <= w - 1)
                     ? 0.5625f * (r1p[P(x + dir)] + r1n[P(x - dir)]) - 0.0625f * (r3p[P(x + dir * 3)] + r3n[P(x - dir * 3)])
                     : (r1p[P(x + dir)] + r1n[P(x - dir)]) * 0.5f;
    }
}

static void compute_hp_row(float* dst, const float* a, int n) {  // :602-617
    for (int j = 1; j < n - 2; ++j) dst[j] = 0.5625f * (a[j] + a[j + 1]) - 0.0625f * (a[j - 1] + a[j + 2]);
}

// eedi3.zig:619-904
static void interp_line_hp(const float* r3p, const float* r1p, const float* r1n, const float* r3n, float* hp3p, float* hp1p, float* hp1n, float* hp3n, int buflen,
                           float* dst, int8_t* pbackt, int* fpath, float* t_costs, int* dmap, int stride, int w, const Params& d, const uint8_t* bmask) {
    if (w == 0) return;
    const int nrad = d.nrad, mdis = d.mdis, cen = 2 * mdis, tpitch = 4 * mdis + 1;
    compute_hp_row(hp3p, r3p, buflen);
    compute_hp_row(hp1p, r1p, buflen);
    compute_hp_row(hp1n, r1n, buflen);
    compute_hp_row(hp3n, r3n, buflen);
    if (bmask) {
        bool any = false;
        for (int x = 0; x < w; ++x) any = any || bmask[x];
        if (!any) {
            for (int x = 0; x < w; ++x) {
                dmap[x] = 0;
                dst[x] = 0.5625f * (r1p[P(x)] + r1n[P(x)]) - 0.0625f * (r3p[P(x)] + r3n[P(x)]);
            }
            return;
        }
    }
    std::vector<float> baseM(buflen, 0.0f), baseHp(buflen, 0.0f);
    for (int u = -cen; u <= cen; ++u) {
        const int uh = u >> 1;
        const bool odd = (u & 1) != 0;
        const int lo0 = odd ? -uh - 1 : -uh;
        const float* A0 = odd ? hp3p : r3p;
        const float* B0 = odd ? hp1p : r1p;
        const float* C0 = odd ? hp1n : r1n;
        const float* D0 = odd ? hp3n : r3n;
        float* tc = t_costs + (size_t)(cen + u) * stride;
        const float beta_term = d.beta * (float)std::abs(u) * 0.5f;
        for (int j = std::min(0, u) - nrad; j < w + std::max(0, u) + nrad; ++j)
            baseM[P(j)] = std::fabs(r3p[P(j)] - r1p[P(j - u)]) + std::fabs(r1p[P(j)] - r1n[P(j - u)]) + std::fabs(r1n[P(j)] - r3n[P(j - u)]);
        if (odd)
            for (int j = uh - nrad; j < w + uh + nrad; ++j)
                baseHp[P(j)] = std::fabs(A0[P(j)] - B0[P(j - u)]) + std::fabs(B0[P(j)] - C0[P(j - u)]) + std::fabs(C0[P(j)] - D0[P(j - u)]);
        const float* s0base = odd ? baseHp.data() : baseM.data();
        for (int x = 0; x < w; ++x) {
            float s0 = 0, s1 = 0, s2 = 0;
            for (int k = -nrad; k <= nrad; ++k) {
                s1 += baseM[P(x + k)];
                s2 += baseM[P(x + u + k)];
                s0 += s0base[P(x + uh + k)];
            }
            const float ip = (B0[P(x + uh)] + C0[P(x + lo0)]) * 0.5f;
            const float v = std::fabs(r1p[P(x)] - ip) + std::fabs(r1n[P(x)] - ip);
            tc[x] = d.alpha * (s0 + s1 + s2) + beta_term + d.one_minus_ab * v;
        }
    }
    std::vector<float> pc[2];
    pc[0].assign(tpitch + 4, kFltMax09);
    pc[1].assign(tpitch + 4, kFltMax09);
    int ping = 0;
    for (int ui = 0; ui < tpitch; ++ui) pc[ping][ui + 2] = t_costs[(size_t)ui * stride + 0];
    for (int xc = 1; xc < w; ++xc) {
        const int pong = ping ^ 1;
        int8_t* piT = pbackt + (size_t)(xc - 1) * tpitch;
        if (bmask && !bmask[xc]) {
            if (xc == 1) {
                for (int ui = 0; ui < tpitch; ++ui) pc[pong][ui + 2] = t_costs[(size_t)ui * stride + xc];
                std::memset(piT, 0, tpitch);
            } else {
                pc[pong] = pc[ping];
                std::memcpy(piT, pbackt + (size_t)(xc - 2) * tpitch, tpitch);
            }
            ping = pong;
            continue;
        }
        const float g1 = d.gamma * 0.5f, g2 = d.gamma;
        for (int ui = 0; ui < tpitch; ++ui) {
            // Vector body (:806-832): candidates in the order -2,-1,0,+1,+2 with strict <,
            // starting from the -2 candidate. The scalar tail (:834-849) starts from
            // flt_max_09 / delta 0 instead; the two differ only when every candidate is
            // >= flt_max_09, which needs unreachable nodes on both sides.
            const float* p = pc[ping].data();
            const int nvec = (tpitch / 8) * 8;
            float bval;
            int8_t bd;
            if (ui < nvec) {
                bval = p[ui] + g2;
                bd = -2;
                const float c_m1 = p[ui + 1] + g1, c_0 = p[ui + 2], c_p1 = p[ui + 3] + g1, c_p2 = p[ui + 4] + g2;
                if (c_m1 < bval) { bval = c_m1; bd = -1; }
                if (c_0 < bval) { bval = c_0; bd = 0; }
                if (c_p1 < bval) { bval = c_p1; bd = 1; }
                if (c_p2 < bval) { bval = c_p2; bd = 2; }
            } else {
                bval = kFltMax09;
                bd = 0;
                for (int dv = -2; dv <= 2; ++dv) {
                    const float gv = d.gamma * (float)std::abs(dv) * 0.5f;
                    const float cc = p[ui + dv + 2] + gv;
                    if (cc < bval) { bval = cc; bd = (int8_t)dv; }
                }
            }
            pc[pong][ui + 2] = std::fmin(bval + t_costs[(size_t)ui * stride + xc], kFltMax09);
            piT[ui] = bd;
        }
        ping = pong;
    }
    fpath[w - 1] = 0;
    for (int bx = w - 2; bx >= 0; --bx) fpath[bx] = fpath[bx + 1] + pbackt[(size_t)bx * tpitch + (cen + fpath[bx + 1])];
    for (int x = 0; x < w; ++x) {
        if (bmask && !bmask[x]) {
            dmap[x] = 0;
            dst[x] = 0.5625f * (r1p[P(x)] + r1n[P(x)]) - 0.0625f * (r3p[P(x)] + r3n[P(x)]);
            continue;
        }
        const int dir = fpath[x];
        dmap[x] = dir;
        if ((dir & 1) == 0) {
            const int d2 = dir >> 1, ad = std::abs(d2);
            if (x >= ad * 3 && x + ad * 3 <= w - 1)
                dst[x] = 0.5625f * (r1p[P(x + d2)] + r1n[P(x - d2)]) - 0.0625f * (r3p[P(x + d2 * 3)] + r3n[P(x - d2 * 3)]);
            else
                dst[x] = (r1p[P(x + d2)] + r1n[P(x - d2)]) * 0.5f;
        } else {
            const int d20 = dir >> 1, d21 = (dir + 1) >> 1, d30 = (dir * 3) >> 1, d31 = (dir * 3 + 1) >> 1;
            const int ad = std::max(std::abs(d30), std::abs(d31));
            if (x >= ad && x + ad <= w - 1) {
                const float c0 = r3p[P(x + d30)] + r3p[P(x + d31)];
                const float c1 = r1p[P(x + d20)] + r1p[P(x + d21)];
                const float c2 = r1n[P(x - d20)] + r1n[P(x - d21)];
                const float c3 = r3n[P(x - d30)] + r3n[P(x - d31)];
                dst[x] = 0.28125f * (c1 + c2) - 0.03125f * (c0 + c3);
            } else {
                dst[x] = (r1p[P(x + d20)] + r1p[P(x + d21)] + r1n[P(x - d20)] + r1n[P(x - d21)]) * 0.25f;
            }
        }
    }
}

// eedi3.zig:915-1046
static void vcheck_lines(const float* src, float* dst, const float* scp, const int* dmap, float* tline, int field, int L, int n_dst, int n_src, ptrdiff_t lstride,
                         ptrdiff_t dmap_stride, int n_interp, const Params& d) {
    for (int off = 1; off + 1 < n_interp; ++off) {
        const int pd = field + 2 * off;
        if (pd < 2 || pd + 2 >= n_dst) continue;
        float* dl = dst + (ptrdiff_t)pd * lstride;
        const float* d1p = dst + (ptrdiff_t)(pd - 1) * lstride;
        const float* d2p = dst + (ptrdiff_t)(pd - 2) * lstride;
        const float* d1n = dst + (ptrdiff_t)(pd + 1) * lstride;
        const float* d2n = dst + (ptrdiff_t)(pd + 2) * lstride;
        const float* d3p = src + (ptrdiff_t)src_col(d.dh, pd - 3, n_src) * lstride;
        const float* d3n = src + (ptrdiff_t)src_col(d.dh, pd + 3, n_src) * lstride;
        const int* dc = dmap + (ptrdiff_t)off * dmap_stride;
        const int* dp = dmap + (ptrdiff_t)(off - 1) * dmap_stride;
        const int* dn = dmap + (ptrdiff_t)(off + 1) * dmap_stride;
        const float* sl = scp ? scp + (ptrdiff_t)pd * lstride : nullptr;
        for (int i = 0; i < L; ++i) {
            const int dirc = dc[i];
            const float cint = sl ? sl[i] : 0.5625f * (d1p[i] + d1n[i]) - 0.0625f * (d3p[i] + d3n[i]);
            if (dirc == 0) {
                tline[i] = cint;
                continue;
            }
            const int dirt = dp[i], dirb = dn[i];
            if (std::max(dirc * dirt, dirc * dirb) < 0 || (dirt == dirb && dirt == 0)) {
                tline[i] = cint;
                continue;
            }
            int maxoff;
            if (d.hp)
                maxoff = ((dirc & 1) == 0) ? std::abs(dirc >> 1) : std::max(std::abs(dirc >> 1), std::abs((dirc + 1) >> 1));
            else
                maxoff = std::abs(dirc);
            if (i + maxoff >= L || i - maxoff < 0) {
                tline[i] = cint;
                continue;
            }
            float it, ib, vt, vb;
            int dabs;
            if (d.hp && (dirc & 1) != 0) {
                const int d20 = dirc >> 1, d21 = (dirc + 1) >> 1;
                const int ip0 = i + d20, ip1 = i + d21, im0 = i - d20, im1 = i - d21;
                const float s2p = d2p[ip0] + d2p[ip1], s1p = d1p[ip0] + d1p[ip1], pa0 = dl[ip0] + dl[ip1], ps0 = dl[im0] + dl[im1];
                const float s1n = d1n[im0] + d1n[im1], s2n = d2n[im0] + d2n[im1];
                it = (s2p + ps0) * 0.25f;
                vt = (std::fabs(s2p - s1p) + std::fabs(pa0 - s1p)) * 0.5f;
                ib = (pa0 + s2n) * 0.25f;
                vb = (std::fabs(s2n - s1n) + std::fabs(ps0 - s1n)) * 0.5f;
                dabs = std::abs(dirc) >> 1;
            } else {
                const int offh = d.hp ? dirc >> 1 : dirc;
                const int ipd = i + offh, imd = i - offh;
                it = (d2p[ipd] + dl[imd]) * 0.5f;
                ib = (dl[ipd] + d2n[imd]) * 0.5f;
                vt = std::fabs(d2p[ipd] - d1p[ipd]) + std::fabs(dl[ipd] - d1p[ipd]);
                vb = std::fabs(d2n[imd] - d1n[imd]) + std::fabs(dl[imd] - d1n[imd]);
                dabs = d.hp ? std::abs(dirc) >> 1 : std::abs(dirc);
            }
            const float vc = std::fabs(dl[i] - d1p[i]) + std::fabs(dl[i] - d1n[i]);
            const float e0 = std::fabs(it - d1p[i]), e1 = std::fabs(ib - d1n[i]), e2 = std::fabs(vt - vc), e3 = std::fabs(vb - vc);
            float m0, m1;
            if (d.vcheck == 1) {
                m0 = std::fmin(e0, e1);
                m1 = std::fmin(e2, e3);
            } else if (d.vcheck == 2) {
                m0 = (e0 + e1) * 0.5f;
                m1 = (e2 + e3) * 0.5f;
            } else {
                m0 = std::fmax(e0, e1);
                m1 = std::fmax(e2, e3);
            }
            const float a0 = m0 * d.rcp0, a1 = m1 * d.rcp1;
            const float a2 = std::fmax((d.vthresh2 - (float)dabs) * d.rcp2, 0.0f);
            const float a = std::fmin(std::fmax(a0, std::fmax(a1, a2)), 1.0f);
            tline[i] = (1.0f - a) * dl[i] + a * cint;
        }
        std::memcpy(dl, tline, sizeof(float) * L);
    }
}

// src/vapoursynth/eedi3.zig:26-140
static void process_plane(const Params& d, const float* srcl, float* dstl, const float* scpl, const uint8_t* maskl, ptrdiff_t mask_stride, int field, int L,
                          ptrdiff_t lstride, ptrdiff_t dstride, int n_src, int n_dst) {
    const int n_interp = d.dh ? n_src : n_src / 2;
    if (d.dh) {
        for (int k = 0; k < n_src; ++k) std::memcpy(dstl + (ptrdiff_t)(2 * k + (1 - field)) * dstride, srcl + (ptrdiff_t)k * lstride, sizeof(float) * L);
    } else {
        for (int k = 1 - field; k < n_src; k += 2) std::memcpy(dstl + (ptrdiff_t)k * dstride, srcl + (ptrdiff_t)k * lstride, sizeof(float) * L);
    }
    const int buflen = L + 2 * kPadH + 32;
    std::vector<float> bufs[4], hpb[4], t_base(buflen, 0.0f);
    for (auto& b : bufs) b.assign(buflen, 0.0f);
    if (d.hp)
        for (auto& b : hpb) b.assign(buflen, 0.0f);
    const int tpitch = d.hp ? 4 * d.mdis + 1 : 2 * d.mdis + 1;
    const int cstride = L;
    std::vector<float> t_costs((size_t)tpitch * cstride);
    std::vector<int8_t> pbackt((size_t)tpitch * cstride);
    std::vector<int> fpath(L), dmap((size_t)n_interp * L);
    std::vector<float> tline(L);
    std::vector<uint8_t> bmask(L);
    float *p3p = bufs[0].data(), *p1p = bufs[1].data(), *p1n = bufs[2].data(), *p3n = bufs[3].data();
    int interp_off = 0;
    for (int line = field; line < n_dst; line += 2) {
        if (interp_off == 0) {
            fill_padded_row(p3p, srcl + (ptrdiff_t)src_col(d.dh, line - 3, n_src) * lstride, L);
            fill_padded_row(p1p, srcl + (ptrdiff_t)src_col(d.dh, line - 1, n_src) * lstride, L);
            fill_padded_row(p1n, srcl + (ptrdiff_t)src_col(d.dh, line + 1, n_src) * lstride, L);
            fill_padded_row(p3n, srcl + (ptrdiff_t)src_col(d.dh, line + 3, n_src) * lstride, L);
        } else {
            std::swap(p3p, p1p);
            std::swap(p1p, p1n);
            std::swap(p1n, p3n);
            fill_padded_row(p3n, srcl + (ptrdiff_t)src_col(d.dh, line + 3, n_src) * lstride, L);
        }
        const uint8_t* bm = nullptr;
        if (maskl) {
            const int mrow = d.dh ? interp_off : line;
            build_bmask(bmask.data(), maskl + (ptrdiff_t)mrow * mask_stride, L, d.mdis);
            bm = bmask.data();
        }
        float* out_line = dstl + (ptrdiff_t)line * dstride;
        if (d.hp)
            interp_line_hp(p3p, p1p, p1n, p3n, hpb[0].data(), hpb[1].data(), hpb[2].data(), hpb[3].data(), L + 2 * kPadH, out_line, pbackt.data(), fpath.data(),
                           t_costs.data(), dmap.data() + (size_t)interp_off * L, cstride, L, d, bm);
        else
            interp_line(p3p, p1p, p1n, p3n, out_line, pbackt.data(), fpath.data(), t_base.data(), t_costs.data(), dmap.data() + (size_t)interp_off * L, cstride, L, d, bm);
        ++interp_off;
    }
    if (d.vcheck > 0) {
        // vcheckLine indexes src and dst with one line stride; both are dense here when they differ
        if (lstride == dstride) {
            vcheck_lines(srcl, dstl, scpl, dmap.data(), tline.data(), field, L, n_dst, n_src, dstride, L, n_interp, d);
        } else {
            std::vector<float> s2((size_t)n_src * dstride);
            for (int k = 0; k < n_src; ++k) std::memcpy(s2.data() + (size_t)k * dstride, srcl + (ptrdiff_t)k * lstride, sizeof(float) * L);
            vcheck_lines(s2.data(), dstl, scpl, dmap.data(), tline.data(), field, L, n_dst, n_src, dstride, L, n_interp, d);
        }
    }
}

}  // namespace

// One plane. User-level parameters (createImpl defaults: alpha .2 beta .25 gamma 20 nrad 2
// mdis 20 hp 0 vcheck 2 vthresh 32/64/4); scaling as src/vapoursynth/eedi3.zig:465-473.
// `field` is the resolved 0/1 parity of the frame (getFrame :166-172). horizontal != 0 is
// EEDI3H: the plane is transposed, run through the vertical pipeline, and transposed back
// (src is src_w x src_h; dst is 2*src_w (dh) or src_w wide). sclip/mclip planes may be NULL;
// sclip has the geometry of dst, mclip that of src (u8).
VSZO_API int vszo_eedi3_plane(const float* src, float* dst, const float* sclip, const uint8_t* mclip, ptrdiff_t sstride, ptrdiff_t dstride, ptrdiff_t scstride,
                              ptrdiff_t mstride, int src_w, int src_h, int field, int dh, float alpha, float beta, float gamma, int nrad, int mdis, int hp, int vcheck,
                              float vthresh0, float vthresh1, float vthresh2, int horizontal) {
    Params d;
    d.mdis = mdis;
    d.nrad = nrad;
    d.vcheck = vcheck;
    d.hp = hp != 0;
    d.dh = dh != 0;
    d.one_minus_ab = 1.0f - alpha - beta;
    d.alpha = alpha / 3.0f;
    d.beta = beta / 255.0f;
    d.gamma = gamma / 255.0f;
    vthresh0 /= 255.0f;
    vthresh1 /= 255.0f;
    d.vthresh2 = vthresh2;
    d.rcp0 = 1.0f / vthresh0;
    d.rcp1 = 1.0f / vthresh1;
    d.rcp2 = 1.0f / vthresh2;
    if (!horizontal) {
        const int dst_h = d.dh ? src_h * 2 : src_h;
        if (vcheck > 0 && sclip && scstride != dstride) return -2;  // the oracle wants sclip at the dst stride
        process_plane(d, src, dst, vcheck > 0 ? sclip : nullptr, mclip, mstride, field, src_w, sstride, dstride, src_h, dst_h);
        return 0;
    }
    const int dst_w = d.dh ? src_w * 2 : src_w;
    const int Ls = src_h;  // dense transposed line stride
    std::vector<float> srcT((size_t)src_w * Ls), dstT((size_t)dst_w * Ls), scT;
    std::vector<uint8_t> mT;
    for (int r = 0; r < src_h; ++r)
        for (int c = 0; c < src_w; ++c) srcT[(size_t)c * Ls + r] = src[(ptrdiff_t)r * sstride + c];
    if (mclip) {
        mT.resize((size_t)src_w * Ls);
        for (int r = 0; r < src_h; ++r)
            for (int c = 0; c < src_w; ++c) mT[(size_t)c * Ls + r] = mclip[(ptrdiff_t)r * mstride + c];
    }
    if (vcheck > 0 && sclip) {
        scT.resize((size_t)dst_w * Ls);
        for (int r = 0; r < src_h; ++r)
            for (int c = 0; c < dst_w; ++c) scT[(size_t)c * Ls + r] = sclip[(ptrdiff_t)r * scstride + c];
    }
    process_plane(d, srcT.data(), dstT.data(), scT.empty() ? nullptr : scT.data(), mT.empty() ? nullptr : mT.data(), Ls, field, src_h, Ls, Ls, src_w, dst_w);
    for (int c = 0; c < dst_w; ++c)
        for (int r = 0; r < src_h; ++r) dst[(ptrdiff_t)r * dstride + c] = dstT[(size_t)c * Ls + r];
    return 0;
}
