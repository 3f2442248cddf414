// vszip.EEDI3 / EEDI3H on gfx950 (32-bit float planes, like the reference).
//
// Replaces processPlane (src/vapoursynth/eedi3.zig:26-140) and its kernels
// interpLine (src/filters/eedi3.zig:349-592, with costBlockDirect :311-347) and
// vcheckLine (:915-1046); EEDI3H is transpose -> vertical pipeline -> transpose
// (src/vapoursynth/eedi3.zig:220-246). The tuned line kernel covers hp=False, mdis <= 31 without
// a mask; hp=True (interpLineHP), mdis up to 40 and mclip go through the general line kernel.
//
// eedi3_line_kernel — one wave per interpolated line, x processed in blocks of 64:
//   cost phase (lanes = x): for every direction u the per-column base term
//     t_base[j] = |r3p[j]-r1p[j-2u]| + |r1p[j]-r1n[j-2u]| + |r1n[j]-r3n[j-2u]|
//   goes to LDS, the three (2*nrad+1)-tap window sums are re-formed in the reference's
//   k order (fresh sums, not a running one), and cost[u][x] lands in a 64 x tpitch LDS tile;
//   DP phase (lanes = directions): the Viterbi recurrence over the 64 columns with the
//   reference's strict-< tie-breaking (centre, then left, then right), neighbours through
//   wave shuffles, back-pointers into an LDS tile that is flushed to a global scratch;
//   after the last block the path is walked back block by block and every lane writes its
//   column's cubic / linear interpolation and direction.
// eedi3_vcheck_kernel — the vertical-consistency blend; it reads the already-updated line
//   pd-2, so lines are processed in order by one workgroup per plane, 1024 columns wide.
// All f32 arithmetic is unfused and in the reference's operation order: results are
// bit-identical to the CPU oracle (the discrete path makes anything less visible).
#include <cfloat>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kMaxMdis = 31;
constexpr int kXB = 64;         // columns per block
constexpr int kMaxPlanesE = 48;  // planes per call (16 YUV frames): the per-plane vcheck chains run side by side

struct EPlane {
    const float *src;   // field-source plane (vertical layout: rows = lines)
    float *dst;
    int *dmap;          // [n_interp][w]
    int8_t *pback;      // [lines of this plane][w][tpitch]
    int sstride, dstride;
    int w, n_src, n_dst, n_interp;
    int line0;          // first global line id of this plane
};

struct EParams {
    EPlane p[kMaxPlanesE];
    int nplanes;
    int field, dh;
    int mdis, nrad;
    float alpha, beta, gamma, one_minus_ab;
    int line_base;  // first global line id of this launch (a call may launch its tall and its short planes separately)
    int copy_kept;  // the tuned line kernel also writes the KEPT field line beside its interpolated one (round 6: no copy kernel in front of it)
    // eedi3_line_kernel<..., MCLIP> only: the planes' mclip (geometry of the - transposed - source plane) or NULL
    const uint8_t *mask[kMaxPlanesE];
    int mstride[kMaxPlanesE];
};

__device__ __forceinline__ int reflect_row(int y, int h) {  // eedi3.zig:88-96
    if (h == 1) return 0;
    int r = y;
    while (r < 0 || r >= h) {
        if (r < 0) r = -r;
        if (r >= h) r = 2 * (h - 1) - r;
    }
    return r;
}
__device__ __forceinline__ int src_col(bool dh, int off, int n_src) {  // :102-104
    return dh ? reflect_row(off, 2 * n_src) / 2 : reflect_row(off, n_src);
}

// Column c of a mirror-padded row (mirrorPad :107-110): reflect-101 on both sides. Only columns
// within nrad of the line take part in an unmasked cost (a direction u is evaluated at x only for
// u <= min(x, w-1-x)), so on lines shorter than the staging reach the columns past a single
// reflection are staged as clamped filler that nothing reads back.
__device__ __forceinline__ float rowv(const float *row, int c, int w) {
    c = c < 0 ? -c : c;
    c = c >= w ? 2 * (w - 1) - c : c;
    return row[min(max(c, 0), w - 1)];
}

// Single-wave workgroups: LDS traffic is ordered by a wave-level fence, no s_barrier.
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// value of the lane below / above through a DPP wave shift (no LDS round trip); the first /
// last lane receives `edge`
__device__ __forceinline__ float lane_below(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));  // wave_shr:1
}
__device__ __forceinline__ float lane_above(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));  // wave_shl:1
}

// wave shifts whose first / last lane reads 0 (bound_ctrl): foldable into the consuming VALU instruction
__device__ __forceinline__ float lane_below0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));  // wave_shr:1
}
__device__ __forceinline__ float lane_above0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));  // wave_shl:1
}
template <int L>
__device__ __forceinline__ int write_lane(int old, int s) {  // lane L of `old` := the (wave-uniform) s
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(s), "n"(L));
    return old;
}
template <int B, int E, int S, class F>
__device__ __forceinline__ void static_for(F &&f) {  // f(integral_constant<B>), f(<B + S>) ... up to and including E
    if constexpr (B <= E) {
        f(std::integral_constant<int, B>{});
        static_for<B + S, E, S>(f);
    }
}

constexpr int kU = 4;  // directions per cost pass
// NRAD is a template parameter so that the window-sum loop unrolls (with a runtime trip count
// its LDS reads are issued one per iteration and each waits out the full LDS latency); MD is the
// largest mdis the instantiation serves (20: the default and below, 31: one DP state per lane) and
// sizes the LDS arrays — the default geometry takes 20 KiB per wave instead of 29, i.e. 7 instead of
// 5 waves per CU, and the kernel is latency bound.
// FIXED: mdis == MD (the default 20): the direction loop unrolls and every LDS offset of a pass is an immediate
// (with a runtime mdis a quarter of the cost phase's instructions are address arithmetic).
// MASK (round 4, with FIXED): the layout and the unrolled passes of mdis == MD, but only the directions |u| <= prm.mdis take part — the others keep an infinite
// cost, which pins their Viterbi state at the sentinel exactly like the lanes past the last direction. A run-time mdis below the default used to take the
// FIXED = false instance, whose cost pass is the round-3 form: slower at 21 directions than the fixed instance at 41 (tools/eedi3_param_sweep.py).
// (the default geometry's LDS, round 6, lets 14 waves share a CU; registers for three a SIMD: a fourth measured no faster, see profiles/r06_notes.md)
// MCLIP (round 6, late): the mclip forms of the reference's line loop (eedi3.zig:368-399, :431-440, :492-505, :567-577) on this kernel - before, any mclip sent
// the call to the general kernel (16 x 1080p a call: 710-810 fps against 4 300 without a mask; tools/eedi3_variants_timing.py). bmask[x] = "a mask sample
// within mdis of x" (buildBmask :285-304) is a 64-bit word a block, built with three ballots and shifts on the scalar unit:
//   * a line without any mask sample is the plain vertical cubic (:392-399);
//   * a column outside bmask keeps the costs of the column before it and REPEATS that column's back-pointers (:492-505: pbackt[x-1] = pbackt[x-2]; column 1:
//     the costs of column 1 themselves, pointers 0) - two instructions on the code word instead of a Viterbi step, behind a scalar branch on the word's bit;
//   * a block without any column in bmask (other than the line's first, whose costs the reference always computes) skips its staging and its whole cost phase:
//     nothing reads the costs of such columns, and its 64 codes are one code repeated;
//   * the path is walked as always and set to 0 outside bmask afterwards (:567-577).
template <int NRAD, int MD, bool FIXED, bool MASK = false, bool MCLIP = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FIXED && !MASK && MD % kU == 0 && MD <= 20 ? 3 : 1))) void eedi3_line_kernel(const EParams prm) {
    static_assert(!MASK || FIXED, "MASK rides on the fixed layout");
    static_assert(!MCLIP || FIXED, "MCLIP rides on the fixed layout");
    // The t_base and window-sum steps run a fixed number of 128-entry iterations (NIT2: every lane owns two
    // neighbouring entries, and the kU directions of a pass share ONE first column, so the span is the longest
    // of the pass) with no guard — entries past a direction's span are computed from padding and never read —
    // so that the compiler emits straight-line code (with `t < span` loops it vectorised them into
    // prologue/body/remainder blocks that cost more instructions than the arithmetic).
    constexpr int NIT2 = (kXB + 2 * MD + 2 * NRAD + 2 * (kU - 1) + 127) / 128;
    constexpr int kRowW = 128 * NIT2 + 2 * MD + 8;  // staged columns per source row: block + reach 2*mdis + nrad each side, padded for the overshoot
    constexpr int kTbW = 128 * NIT2 + 8;            // t_base / window-sum entries per direction
    const float kFltMax09 = FLT_MAX * 0.9f;
    // kRegWin (round 4, the default geometry): t_base never goes to LDS — see cost_pass_fixed below
    constexpr bool kRegWin = FIXED && NIT2 == 1 && MD % kU == 0;
    constexpr int kCtP = kXB + 4;  // 16-byte aligned rows: a direction's 64 costs come back as 16 ds_read_b128, free of bank conflicts at this pitch
    // LDS of a wave: rows[4][kRowW] (r3p, r1p, r1n, r3n around the block, mirror padding applied), tbws (the (2*nrad+1)-tap window sums of the kU
    // directions of the current pass and, without kRegWin, their t_base in front) and the cost tile ctile[2 MD + 1][kCtP].
    // Round 6 (kOverlay, the default geometry): the staged rows and the window sums live only during the cost phase, the tile is read only after it,
    // so they SHARE memory: rows and tbws lie over the tile's last rows, whose costs (the directions u >= kDefer0 - MD, produced by the last passes)
    // wait in registers until the last pass has read its taps. 11 152 B a wave instead of 16 144: 14 waves a CU instead of 9 or 10.
    constexpr bool kOverlay = kRegWin && !MASK;
    constexpr int kTileF = (2 * MD + 1) * kCtP, kRowsF = 4 * kRowW, kTbF = (kRegWin ? 1 : 2) * kU * kTbW, kTailF = kRowsF + kTbF;
    constexpr int kLdsF = kOverlay ? (kTileF > kTailF ? kTileF : kTailF) : kTileF + kTailF;
    constexpr int kTailAt = kLdsF - kTailF;  // (floats) where rows start
    static_assert(kTailAt % 4 == 0 && kRowsF % 4 == 0, "16-byte aligned rows");
    constexpr int kDefer0 = kOverlay ? kTailAt / kCtP : 2 * MD + 1;  // first tile row the staged rows overlap
    constexpr int kNDefer = 2 * MD + 1 - kDefer0;
    __shared__ __attribute__((aligned(16))) float lds[kLdsF];
    float (*ctile)[kCtP] = reinterpret_cast<float (*)[kCtP]>(lds);
    float (*rows)[kRowW] = reinterpret_cast<float (*)[kRowW]>(lds + kTailAt);
    float (*tbws)[kU][kTbW] = reinterpret_cast<float (*)[kU][kTbW]>(lds + kTailAt + kRowsF);
    float (*tb)[kTbW] = tbws[0], (*ws)[kTbW] = tbws[kRegWin ? 0 : 1];
#ifdef VSZIP_E3_PAD_LDS  // (occupancy sweeps only: that many bytes of LDS a wave more)
    __shared__ float padlds[VSZIP_E3_PAD_LDS / 4];
    if (prm.nplanes < 0) padlds[threadIdx.x] = prm.gamma;
    asm volatile("" ::"v"(padlds[threadIdx.x ^ 1]));
#endif

    int pi = 0;
    const int gl = prm.line_base + (int)blockIdx.x;
#pragma unroll 1
    for (int i = 1; i < prm.nplanes; ++i)
        if (gl >= prm.p[i].line0) pi = i;
    const EPlane pl = prm.p[pi];
    const int off = gl - pl.line0;              // interpolated-line index within the plane
    const int line = prm.field + 2 * off;       // destination line
    constexpr int nrad = NRAD;
    const int w = pl.w, mdis = FIXED ? MD : prm.mdis, tpitch = 2 * mdis + 1;
    const int lane = threadIdx.x;
    const bool dh = prm.dh != 0;
    const float *r3p = pl.src + (size_t)src_col(dh, line - 3, pl.n_src) * pl.sstride;
    const float *r1p = pl.src + (size_t)src_col(dh, line - 1, pl.n_src) * pl.sstride;
    const float *r1n = pl.src + (size_t)src_col(dh, line + 1, pl.n_src) * pl.sstride;
    const float *r3n = pl.src + (size_t)src_col(dh, line + 3, pl.n_src) * pl.sstride;
    // back-pointer codes of this line: [block][direction][64 columns], TWO BITS each (16 bytes a direction and block), the path's step as a
    // signed two-bit number (0: stay, 3 = -1: from the direction below, 1: from the direction above), column c in bits 31 - 2 (c & 15) .. 30 - 2 (c & 15)
    // of word c >> 4 (the forward pass shifts them in from the right); column 63 of a block is produced by the first step of the next block.
    // (The scratch area is sized for a byte a code, round 4's form; a line uses the first quarter of its share.)
    uint8_t *pback = reinterpret_cast<uint8_t *>(pl.pback) + (size_t)off * ((w + kXB - 1) / kXB * kXB) * tpitch;
    float *out = pl.dst + (size_t)line * pl.dstride;
    int *dmap = pl.dmap + (size_t)off * w;
    const int reach = 2 * mdis + nrad, roww = kXB + 2 * reach;
    // the four source rows start on 8-byte boundaries (every VapourSynth plane: 32-byte aligned rows): blocks inside the line are staged in pairs
    const bool pair_ok = __builtin_amdgcn_readfirstlane((int)(((reinterpret_cast<uintptr_t>(r3p) | reinterpret_cast<uintptr_t>(r1p) | reinterpret_cast<uintptr_t>(r1n) | reinterpret_cast<uintptr_t>(r3n)) & 7) == 0)) != 0;

    [[maybe_unused]] float *kept = prm.field ? out - pl.dstride : out + pl.dstride;  // (copy_kept) the kept field line beside this one
    [[maybe_unused]] const uint8_t *maskp = nullptr;
    if constexpr (MCLIP) {
        maskp = prm.mask[pi] ? prm.mask[pi] + (size_t)(dh ? off : line) * prm.mstride[pi] : nullptr;  // (a plane without an mclip: every column takes part)
        if (maskp) {
            bool m = false;
            for (int q = lane; q < w; q += 64) m = m || maskp[q] != 0;
            if (__builtin_amdgcn_ballot_w64(m) == 0) {  // :392-399: nothing to connect on this line
                for (int q = lane; q < w; q += 64) {
                    dmap[q] = 0;
                    out[q] = 0.5625f * (r1p[q] + r1n[q]) - 0.0625f * (r3p[q] + r3n[q]);
                    if (prm.copy_kept) kept[q] = (prm.field ? r1p : r1n)[q];
                }
                return;
            }
        }
    }
    // (MCLIP) bit l of a block's word: a mask sample within mdis of column 64 blk + l. Three ballots (the block and its neighbours), then the 128-bit pairs
    // (block : left neighbour) and (right neighbour : block) are OR-ed with their shifts by 0 ... mdis in doubling steps - all of it wave-uniform
    [[maybe_unused]] auto block_bmask = [&](int b) -> uint64_t {
        if (!maskp) return ~0ull;
        const int q0 = b * kXB + lane;
        auto word = [&](int q) -> uint64_t { return __builtin_amdgcn_ballot_w64(q >= 0 && q < w && maskp[min(max(q, 0), w - 1)] != 0); };
        const uint64_t a0 = word(q0 - kXB), a1 = word(q0), a2 = word(q0 + kXB);
        const int r = FIXED && !MASK ? MD : prm.mdis;  // (the reference dilates by the call's mdis)
        uint64_t hi = a1, lo = a0;  // towards higher columns: samples to the LEFT reach up to mdis columns right
        uint64_t uh = a2, ul = a1;  // towards lower columns
        auto up = [&](int sft) {
            hi |= (hi << sft) | (lo >> (64 - sft));
            lo |= lo << sft;
        };
        auto down = [&](int sft) {
            ul |= (ul >> sft) | (uh << (64 - sft));
            uh |= uh >> sft;
        };
        int cover = 1;
        while (2 * cover <= r + 1) {
            up(cover);
            down(cover);
            cover *= 2;
        }
        if (r + 1 - cover > 0) {
            up(r + 1 - cover);
            down(r + 1 - cover);
        }
        return hi | ul;
    };
    const int mlim = MASK ? prm.mdis : MD;  // (MASK) directions beyond it never get a cost
    if constexpr (MASK) {
        for (int r = 0; r < 2 * MD + 1; ++r) ctile[r][lane] = INFINITY;
    }
    float pcost = kFltMax09;  // DP state of direction `lane` (inactive lanes stay at the sentinel)
    [[maybe_unused]] float dp_l = INFINITY, dp_r = INFINITY, dp_gamma = prm.gamma;  // see dp_step
    uint32_t held[kXB / 16];  // back-pointer codes of the previous block, waiting for their last column
#pragma unroll
    for (int i = 0; i < kXB / 16; ++i) held[i] = 0;
    const int nblk = (w + kXB - 1) / kXB;
    for (int blk = 0; blk < nblk; ++blk) {
        const int xb = blk * kXB;
        const int x = xb + lane;
        const int c0 = xb - reach;  // plane column of rows[.][0]
        [[maybe_unused]] uint64_t bm = ~0ull;
        if constexpr (MCLIP) bm = block_bmask(blk);
        const bool skip = MCLIP && blk != 0 && bm == 0;  // (wave-uniform) no column of the block takes part: no staging, no costs
        if (skip) {
            if (prm.copy_kept && x < w) kept[x] = (prm.field ? r1p : r1n)[x];
        } else {
        // ---- stage the four source rows once per block ------------------------------------
        wave_fence();
        // Round 6: a block whose staged columns all lie inside the line (every block but the first and the last one or two) needs no mirror
        // arithmetic, and with an even first column its samples come as 8-byte pairs: 2 loads and 2 LDS writes a row instead of 3 + 3 and the
        // ~8 instructions of rowv() a sample - 5 % of the kernel's instructions went into this staging loop.
        if (c0 >= 0 && c0 + roww <= w && pair_ok && (c0 & 1) == 0) {  // (wave-uniform)
            for (int t = 2 * lane; t < roww; t += 128) {
                const float2 v0 = *reinterpret_cast<const float2 *>(r3p + c0 + t), v1 = *reinterpret_cast<const float2 *>(r1p + c0 + t);
                const float2 v2 = *reinterpret_cast<const float2 *>(r1n + c0 + t), v3 = *reinterpret_cast<const float2 *>(r3n + c0 + t);
                *reinterpret_cast<float2 *>(&rows[0][t]) = v0;
                *reinterpret_cast<float2 *>(&rows[1][t]) = v1;
                *reinterpret_cast<float2 *>(&rows[2][t]) = v2;
                *reinterpret_cast<float2 *>(&rows[3][t]) = v3;
            }
        } else {
            for (int t = lane; t < roww; t += 64) {
                const int c = min(c0 + t, w - 1 + reach);  // columns past the padding are never read back
                rows[0][t] = rowv(r3p, c, w);
                rows[1][t] = rowv(r1p, c, w);
                rows[2][t] = rowv(r1n, c, w);
                rows[3][t] = rowv(r3n, c, w);
            }
        }
        wave_fence();
        const int lx = lane + reach;  // rows[] index of column x
        // The kept field line next to this one (processPlane copies them first, eedi3.zig(vs):41-56): destination line - 1 (field 1) / + 1 (field 0) is the staged
        // row r1p / r1n as it is - one LDS read and one 256-byte store a block instead of a 400 MB copy kernel in front of every call (round 6).
        if (prm.copy_kept && x < w) kept[x] = rows[prm.field ? 1 : 2][lx];
        // ---- cost phase, lanes = x; kU directions per pass -------------------------------
        // The three steps of a direction (t_base -> window sums -> cost) are a chain of LDS
        // round trips; one direction at a time leaves the wave waiting on LDS latency most of
        // the time, so kU directions go through each step together (independent work in flight).
        auto cost_pass = [&](const int ug) __attribute__((always_inline)) {
            // Everything below is branch-free across the kU directions, so that the compiler can
            // keep the LDS reads of all kU directions in flight together: a direction whose span is
            // shorter than the pass's longest simply computes a few entries nobody reads (rows[],
            // tb[] and ws[] are padded for the overshoot).
            // The LDS pipeline is this kernel's limit (4 SIMDs share it: 1.45 x the VALU time before this
            // layout), so the pass moves as few LDS bytes as the data flow allows: the kU directions start
            // their t_base at ONE column (the unshifted taps r3p[j], r1p[j], r1n[j] are then read once for all
            // of them), every lane owns two neighbouring entries (8-byte reads), and a window sum reads its
            // 2*nrad+2 inputs once for both of its outputs.
            int uu[kU];
#pragma unroll
            for (int i = 0; i < kU; ++i) uu[i] = min(ug + i, mdis);  // past +mdis: a duplicate of the last direction (same values, same slots)
            // t_base columns xb+jlo .. are the ones read back; direction u needs them from min(u, 0, 2u) - nrad on,
            // the pass's smallest u sets the common start (even distance to the staged rows: 8-byte aligned pairs)
            const int jlo = min(0, 2 * ug) - nrad;
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                const int t = 2 * lane + 128 * it;
                const int j = jlo + t + reach;  // rows[] index of column xb + jlo + t (even)
                const float2 a = *reinterpret_cast<const float2 *>(&rows[0][j]);
                const float2 c = *reinterpret_cast<const float2 *>(&rows[1][j]);
                const float2 e = *reinterpret_cast<const float2 *>(&rows[2][j]);
                float2 val[kU];
#pragma unroll
                for (int i = 0; i < kU; ++i) {
                    const int two_u = 2 * uu[i];
                    const float2 b = *reinterpret_cast<const float2 *>(&rows[1][j - two_u]);
                    const float2 d = *reinterpret_cast<const float2 *>(&rows[2][j - two_u]);
                    const float2 f = *reinterpret_cast<const float2 *>(&rows[3][j - two_u]);
                    val[i].x = fabsf(a.x - b.x) + fabsf(c.x - d.x) + fabsf(e.x - f.x);  // :415-425
                    val[i].y = fabsf(a.y - b.y) + fabsf(c.y - d.y) + fabsf(e.y - f.y);
                }
#pragma unroll
                for (int i = 0; i < kU; ++i) *reinterpret_cast<float2 *>(&tb[i][t]) = val[i];
            }
            wave_fence();
            // window sums, accumulated from 0 in k order exactly like sw0/sw1/sw2 (:443-450): the
            // three sums of a pixel are the same function of t_base at x+u, x, x+2u
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                const int t = 2 * lane + 128 * it;  // inputs tb[t .. t + 2*nrad + 1], outputs ws[t + nrad], ws[t + nrad + 1]
                float2 val[kU];
#pragma unroll
                for (int i = 0; i < kU; ++i) {
                    float v[2 * NRAD + 2];
#pragma unroll
                    for (int m = 0; m <= NRAD; ++m) {
                        const float2 p = *reinterpret_cast<const float2 *>(&tb[i][t + 2 * m]);
                        v[2 * m] = p.x;
                        v[2 * m + 1] = p.y;
                    }
                    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                    for (int k = 0; k <= 2 * NRAD; ++k) {
                        s0 += v[k];
                        s1 += v[k + 1];
                    }
                    val[i].x = s0;
                    val[i].y = s1;
                }
#pragma unroll
                for (int i = 0; i < kU; ++i) {
                    ws[i][t + nrad] = val[i].x;
                    ws[i][t + nrad + 1] = val[i].y;
                }
            }
            wave_fence();
            {
                const int lxc = lx;  // lanes past the line end compute on staged (clamped) columns and store nothing
                const int base = lane - jlo;  // tb/ws index of column x
                float val[kU];
#pragma unroll
                for (int i = 0; i < kU; ++i) {
                    const int u = uu[i], two_u = 2 * u;
                    const float sw1 = ws[i][base], sw0 = ws[i][base + u], sw2 = ws[i][base + two_u];
                    const float ip = (rows[1][lxc + u] + rows[2][lxc - u]) * 0.5f;
                    const float v = fabsf(rows[1][lxc] - ip) + fabsf(rows[2][lxc] - ip);
                    val[i] = prm.alpha * (sw0 + sw1 + sw2) + prm.beta * (float)abs(u) + prm.one_minus_ab * v;
                }
                if (x < w) {
#pragma unroll
                    for (int i = 0; i < kU; ++i) ctile[mdis + uu[i]][lane] = val[i];
                }
            }
            wave_fence();  // tb / ws are rewritten by the next pass
        };
        // Round 4, the default geometry (kRegWin): the cost phase is bound by the LDS pipeline (per pass ~240 LDS cycles a wave against ~300 VALU
        // cycles, and four SIMDs share one LDS), so whatever a neighbouring LANE already holds comes through a DPP wave shift instead of LDS:
        //  - a lane owns the t_base pair (2L, 2L+1); the 2*nrad+2 inputs of its two window sums are its own pair and the pairs of the nrad lanes
        //    above, so t_base stays in registers (no store, no fence, no re-read; the sums keep the reference's k order);
        //  - the directions of a pass differ by one lane in the shifted taps r1p/r1n/r3n[j - 2u]: one direction reads them (the smallest |u| end
        //    that keeps the lanes that lose their neighbour outside the entries read back), the others shift them along.
        // A pass is one sign of u (MD % kU == 0) and the last one holds the single direction +MD. profiles/r04_notes.md section 10.
        auto stage_a = [&](auto ugc) __attribute__((always_inline)) {  // t_base and window sums of a pass -> ws
            constexpr int ug = decltype(ugc)::value;
            constexpr int nd = MD - ug + 1 < kU ? MD - ug + 1 : kU;  // directions of this pass
            constexpr int jlo = (ug < 0 ? 2 * ug : 0) - NRAD;
            constexpr int reachc = 2 * MD + NRAD;
            const int t = 2 * lane;
            const int j = jlo + t + reachc;  // rows[] index of column xb + jlo + t (even)
            float2 b[nd], d[nd], f[nd];
            // u < 0: direction i + 1 reads at lane L what direction i reads at lane L - 1, and needs its entries from 2 (i + 1) on only;
            // u >= 0: direction i reads what direction i + 1 reads at lane L + 1, and nothing past entry 64 + 2 MD + 2 nrad (< 122) is read back
            constexpr int ia = ug < 0 ? 0 : nd - 1;
            const float2 a = *reinterpret_cast<const float2 *>(&rows[0][j]);
            const float2 c = *reinterpret_cast<const float2 *>(&rows[1][j]);
            const float2 e = *reinterpret_cast<const float2 *>(&rows[2][j]);
            b[ia] = *reinterpret_cast<const float2 *>(&rows[1][j - 2 * (ug + ia)]);
            d[ia] = *reinterpret_cast<const float2 *>(&rows[2][j - 2 * (ug + ia)]);
            f[ia] = *reinterpret_cast<const float2 *>(&rows[3][j - 2 * (ug + ia)]);
#ifdef VSZIP_E3_TAPS_LDS  // (sweeps: every direction's shifted taps from LDS)
#pragma unroll
            for (int i = 0; i < nd; ++i) {
                if (i == ia) continue;
                b[i] = *reinterpret_cast<const float2 *>(&rows[1][j - 2 * (ug + i)]);
                d[i] = *reinterpret_cast<const float2 *>(&rows[2][j - 2 * (ug + i)]);
                f[i] = *reinterpret_cast<const float2 *>(&rows[3][j - 2 * (ug + i)]);
            }
            if constexpr (false) {
#else
            if constexpr (ug < 0) {
#endif
#pragma unroll
                for (int i = 1; i < nd; ++i) {
                    b[i] = make_float2(lane_below0(b[i - 1].x), lane_below0(b[i - 1].y));
                    d[i] = make_float2(lane_below0(d[i - 1].x), lane_below0(d[i - 1].y));
                    f[i] = make_float2(lane_below0(f[i - 1].x), lane_below0(f[i - 1].y));
                }
            } else
#ifdef VSZIP_E3_TAPS_LDS
                if constexpr (false)
#endif
            {
#pragma unroll
                for (int i = nd - 2; i >= 0; --i) {
                    b[i] = make_float2(lane_above0(b[i + 1].x), lane_above0(b[i + 1].y));
                    d[i] = make_float2(lane_above0(d[i + 1].x), lane_above0(d[i + 1].y));
                    f[i] = make_float2(lane_above0(f[i + 1].x), lane_above0(f[i + 1].y));
                }
            }
#pragma unroll
            for (int i = 0; i < nd; ++i) {
                float v[2 * NRAD + 2];
                v[0] = fabsf(a.x - b[i].x) + fabsf(c.x - d[i].x) + fabsf(e.x - f[i].x);  // :415-425
                v[1] = fabsf(a.y - b[i].y) + fabsf(c.y - d[i].y) + fabsf(e.y - f[i].y);
#pragma unroll
                for (int m = 1; m <= NRAD; ++m) {
                    v[2 * m] = lane_above0(v[2 * m - 2]);
                    v[2 * m + 1] = lane_above0(v[2 * m - 1]);
                }
                // sw0/sw1/sw2 (:443-450) start from 0 and add in k order; 0 + v[0] == v[0] for a sum of absolute values
                float s0 = v[0], s1 = v[1];
#pragma unroll
                for (int k = 1; k <= 2 * NRAD; ++k) {
                    s0 += v[k];
                    s1 += v[k + 1];
                }
                if constexpr (NRAD % 2 == 0) {
                    *reinterpret_cast<float2 *>(&ws[i][t + NRAD]) = make_float2(s0, s1);
                } else {
                    ws[i][t + NRAD] = s0;
                    ws[i][t + NRAD + 1] = s1;
                }
            }
        };
        // The cost step of a pass reads ws and two taps a direction and ends in the cost tile. The passes are software-pipelined: the step's
        // LDS reads are issued first, the NEXT pass's stage_a (whose ws stores stay behind those reads in program order — one wave's LDS
        // operations execute in order, so there is neither a second ws buffer nor a fence between them) runs while they are in flight, and the
        // step's arithmetic follows: one exposed LDS round trip per pass instead of two.
        struct BRegs {
            float sw0[kU], sw1[kU], sw2[kU], p1[kU], p2[kU];
        };
        const float r1c = rows[1][lx], r2c = rows[2][lx];
        [[maybe_unused]] float defer[kNDefer > 0 ? kNDefer : 1];
        auto read_b = [&](auto ugc, BRegs &r) __attribute__((always_inline)) {
            constexpr int ug = decltype(ugc)::value;
            constexpr int nd = MD - ug + 1 < kU ? MD - ug + 1 : kU;
            constexpr int jlo = (ug < 0 ? 2 * ug : 0) - NRAD;
            const int base = lane - jlo;  // ws index of column x
#pragma unroll
            for (int i = 0; i < nd; ++i) {
                const int u = ug + i;
                r.sw1[i] = ws[i][base], r.sw0[i] = ws[i][base + u], r.sw2[i] = ws[i][base + 2 * u];
                r.p1[i] = rows[1][lx + u], r.p2[i] = rows[2][lx - u];
            }
        };
        auto comp_b = [&](auto ugc, const BRegs &r) __attribute__((always_inline)) {
            constexpr int ug = decltype(ugc)::value;
            constexpr int nd = MD - ug + 1 < kU ? MD - ug + 1 : kU;
            float val[nd];
#pragma unroll
            for (int i = 0; i < nd; ++i) {
                const int u = ug + i;
                const float ip = (r.p1[i] + r.p2[i]) * 0.5f;
                const float vv = fabsf(r1c - ip) + fabsf(r2c - ip);
                val[i] = prm.alpha * (r.sw0[i] + r.sw1[i] + r.sw2[i]) + prm.beta * (float)abs(u) + prm.one_minus_ab * vv;
                if constexpr (MASK) val[i] = abs(u) <= mlim ? val[i] : INFINITY;
            }
            if constexpr (kOverlay) {  // tile rows under the staged rows wait in registers (see kOverlay)
#pragma unroll
                for (int i = 0; i < nd; ++i)
                    if (MD + ug + i >= kDefer0) defer[MD + ug + i - kDefer0] = val[i];
            }
            if (x < w) {
#pragma unroll
                for (int i = 0; i < nd; ++i)
                    if (MD + ug + i < kDefer0) ctile[MD + ug + i][lane] = val[i];
            }
        };
        // MASK: a pass none of whose directions takes part is skipped (its tile rows stay infinite from the kernel's start); no software pipeline here
        auto cost_pass_masked = [&](auto ugc) __attribute__((always_inline)) {
            constexpr int ug = decltype(ugc)::value;
            constexpr int nd = MD - ug + 1 < kU ? MD - ug + 1 : kU;
            constexpr int lo = ug < 0 ? -(ug + nd - 1) : ug;  // the pass's smallest |u|
            if (lo > mlim) return;
            BRegs r;
            stage_a(ugc);
            wave_fence();
            read_b(ugc, r);
            comp_b(ugc, r);
            wave_fence();
        };
        auto cost_pass_fixed = [&](auto ugc) __attribute__((always_inline)) {  // (pass ug's stage_a has run)
            constexpr int ug = decltype(ugc)::value;
            BRegs r;
#ifdef VSZIP_E3_NO_PIPE  // (sweeps: pass after pass)
            wave_fence();
            read_b(ugc, r);
            comp_b(ugc, r);
            wave_fence();
            if constexpr (ug + kU <= MD) stage_a(std::integral_constant<int, ug + kU>{});
#else
            read_b(ugc, r);
            __builtin_amdgcn_sched_barrier(0);  // the reads stay in front of the next pass's work
            if constexpr (ug + kU <= MD) stage_a(std::integral_constant<int, ug + kU>{});
            comp_b(ugc, r);
#endif
        };
#ifdef VSZIP_E3_DIAG_ONE_PASS  // (timing diagnostics only, tools/variant.sh: ONE of the eleven direction passes - wrong results)
        if constexpr (kRegWin) {
            stage_a(std::integral_constant<int, 0>{});
            BRegs r;
            read_b(std::integral_constant<int, 0>{}, r);
            comp_b(std::integral_constant<int, 0>{}, r);
        } else
            cost_pass(0);
#else
        if constexpr (kRegWin && MASK) {
            static_for<-MD, MD, kU>(cost_pass_masked);
        } else if constexpr (kRegWin) {
            stage_a(std::integral_constant<int, -MD>{});
            static_for<-MD, MD, kU>(cost_pass_fixed);
        } else if constexpr (FIXED) {
#pragma unroll
            for (int ug = -MD; ug <= MD; ug += kU) cost_pass(ug);
        } else {
            for (int ug = -mdis; ug <= mdis; ug += kU) cost_pass(ug);
        }
#endif
        wave_fence();
        if constexpr (kOverlay && kNDefer > 0) {  // every tap of the block has been read: the last directions' costs take their rows
            if (x < w) {
#pragma unroll
                for (int i = 0; i < kNDefer; ++i) ctile[kDefer0 + i][lane] = defer[i];
            }
            wave_fence();
        }
        }  // (!skip)
        // ---- DP phase, lanes = direction index ---------------------------------------
        const int xe = min(kXB, w - xb);
        // The Viterbi recurrence is a dependent chain along x, so nothing in it may wait on memory:
        // the costs of the block's 64 columns are read from LDS into registers up front, the 64 steps
        // are unrolled, the back-pointer of a step is a 2-bit code shifted into registers (4 of them
        // per block) and written out — straight to global memory, 16 bytes per direction — only after
        // the block.
        // Round 6: the phase runs with EXEC = the directions (lane < tpitch). A DPP read of a lane that EXEC disables is an invalid source like
        // a lane out of range: with bound_ctrl off the destination keeps its value, so the last direction's "above" register stays at the +inf it
        // was given once — the same mechanism that serves lanes 0 and 63 — and the lanes past the last direction need neither an infinite
        // cost (64 moves a block) nor a state.
        uint32_t pk[kXB / 16];
#pragma unroll
        for (int i = 0; i < kXB / 16; ++i) pk[i] = 0;
        if (lane < tpitch) {
            float tcv[kXB];
            if (!skip) {
#pragma unroll
                for (int q = 0; q < kXB / 4; ++q) {
                    const float4 v4 = *reinterpret_cast<const float4 *>(&ctile[lane][4 * q]);
                    tcv[4 * q] = v4.x;
                    tcv[4 * q + 1] = v4.y;
                    tcv[4 * q + 2] = v4.z;
                    tcv[4 * q + 3] = v4.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < kXB; ++q) tcv[q] = 0.f;  // (never used: every column of a skipped block repeats the codes before it)
            }
            // The step (round 4: five dependent instructions; round 6: nine VALU instructions instead of eleven). The new cost is
            // min3(own, below + gamma, above + gamma) + tc whatever the tie-breaking picks (strict < only decides WHICH of equal values is taken), so
            // the codes leave the chain. The reference's order — centre unless the one below is smaller, that unless the one above is smaller still
            // (:536-548) — is: moved = (own != the minimum), and if moved, from below = (below == the minimum). Both compares land in SGPR pairs and
            // enter the code word as carries: word = 2 word + bit, twice (hi = moved & below, lo = moved: 00 stay, 11 = -1 below, 01 = +1 above).
            // The neighbours come through v_add_f32_dpp into registers whose edge lane was set to +inf once and is never written again — a lane without
            // a neighbour keeps "+inf < own" false, as the reference's 0.9*FLT_MAX sentinel does (gamma >= 0 is validated).
            // Wait states inside the block: the two v_addc stand between the write of the state and the next step's DPP reads of it.
            auto dp_step = [&](float tc, uint32_t &word) __attribute__((always_inline)) {
                float bv;
                uint64_t cm, cl, co;
                asm volatile(
                    "v_add_f32_dpp %[vl], %[pc], %[g] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %[vr], %[pc], %[g] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_min3_f32 %[bv], %[pc], %[vl], %[vr]\n\t"
                    "v_cmp_neq_f32_e64 %[cm], %[pc], %[bv]\n\t"
                    "v_cmp_eq_f32_e64 %[cl], %[vl], %[bv]\n\t"
                    "v_add_f32_e32 %[bv], %[bv], %[tc]\n\t"
                    "s_and_b64 %[cl], %[cl], %[cm]\n\t"
                    "v_min_f32_e32 %[pc], %[mx], %[bv]\n\t"
                    "v_addc_co_u32_e64 %[wd], %[co], %[wd], %[wd], %[cl]\n\t"
                    "v_addc_co_u32_e64 %[wd], %[co], %[wd], %[wd], %[cm]"
                    : [vl] "+v"(dp_l), [vr] "+v"(dp_r), [pc] "+v"(pcost), [bv] "=&v"(bv), [cm] "=&s"(cm), [cl] "=&s"(cl), [co] "=&s"(co), [wd] "+v"(word)
                    : [g] "v"(dp_gamma), [tc] "v"(tc), [mx] "v"(kFltMax09)
                    : "scc");
            };
            auto store_block = [&](const uint32_t *q, int b) __attribute__((always_inline)) {  // 64 codes of direction `lane`, block b
                *reinterpret_cast<uint4 *>(pback + ((size_t)b * tpitch + lane) * (kXB / 4)) = make_uint4(q[0], q[1], q[2], q[3]);
            };
            // (MCLIP) column xb + xl of this block: a Viterbi step if bmask holds it; else the costs stay and the codes of the column before it are repeated
            // (:492-505) - the low two bits of the word that holds them: the step's own word, the word before it, or the previous block's last word
            auto column = [&](int xl, uint32_t &word, uint32_t before) __attribute__((always_inline)) {
                if (!MCLIP || ((bm >> xl) & 1ull)) {
                    dp_step(tcv[xl], word);
                } else if (blk == 0 && xl == 1) {  // :494-496: the line's second column starts from its own costs, pointers 0
                    pcost = tcv[1];
                    asm volatile("s_nop 1" : "+v"(pcost));
                    word <<= 2;
                } else {
                    word = (word << 2) | (before & 3u);
                }
            };
            auto column_at = [&](int xl) __attribute__((always_inline)) {  // xl >= 1: code position xl - 1 of this block
                const int pos = xl - 1, j = pos >> 4;
                column(xl, pk[j], (pos & 15) ? pk[j] : (j ? pk[j > 0 ? j - 1 : 0] : held[kXB / 16 - 1]));
            };
            // step 0 completes the previous block (its column 63)
            if (blk == 0) {
                pcost = tcv[0];  // :461-463
                asm volatile("s_nop 1" : "+v"(pcost));  // a DPP read two wait states after the register's last write; inside dp_step the step's own tail provides them
            } else {
                column(0, held[kXB / 16 - 1], held[kXB / 16 - 1]);
                store_block(held, blk - 1);
            }
#ifdef VSZIP_E3_DIAG_NO_DP  // (timing diagnostics only: four of the 64 Viterbi steps of a block - wrong results)
            if (xe == kXB) {
#pragma unroll
                for (int xl = 1; xl < 5; ++xl) dp_step(tcv[xl], pk[(xl - 1) >> 4]);
            } else
#else
            if (xe == kXB && (!MCLIP || bm == ~0ull)) {  // (MCLIP: a block wholly inside bmask - every block of a call whose mask is dense - takes the plain steps, no branch a column)
#pragma unroll
                for (int xl = 1; xl < kXB; ++xl) dp_step(tcv[xl], pk[(xl - 1) >> 4]);
            } else
#endif
            {
#pragma unroll
                for (int xl = 1; xl < kXB; ++xl)
                    if (xl < xe) column_at(xl);
            }
            if (blk == nblk - 1) {
                // the line's last block: its words are as full as they get (xe - 1 codes; even a whole block has no column 63) — move the codes to their places
#pragma unroll
                for (int j = 0; j < kXB / 16; ++j) {
                    const int n = min(max(xe - 1 - 16 * j, 0), 16);
                    pk[j] = n ? pk[j] << (2 * (16 - n)) : 0u;
                }
                store_block(pk, blk);
            }
        }
#pragma unroll
        for (int i = 0; i < kXB / 16; ++i) held[i] = pk[i];
    }
    __syncthreads();
    __threadfence_block();

    // ---- backtrack (:557-565) + output (:577-591), block by block from the right ------------
    // fpath[x] = fpath[x+1] + bp[x][fpath[x+1]] is a scalar chain: the block's codes come back into
    // registers (each lane its own direction's 64 bytes), a step is v_readlane of the word that holds
    // column xl at lane mdis+fpath, a signed bit-field extract and a scalar add; the path goes to the
    // lane of its column with v_writelane. No LDS and no barrier in the chain.
    // Round 4: software-pipelined. A block's step chain is serial (0.8 us of scalar work), but neither the NEXT block's codes nor this block's four
    // output taps depend on anything but memory: the codes of block b - 1 are requested before block b is walked, and a block's taps are requested
    // after its walk and consumed after the next block's — each round trip to memory (about 1 us, exposed twice per block before: a wave has 0.75
    // neighbours on its SIMD to hide it behind) passes under a walk. profiles/r04_notes.md section 7.
    int carry = mdis;  // mdis + fpath of the first column of the block to the right
    uint32_t qn[kXB / 16];
    auto load_codes = [&](int b, uint32_t *dstq) __attribute__((always_inline)) {
        const uint4 v = *reinterpret_cast<const uint4 *>(pback + ((size_t)b * tpitch + min(lane, tpitch - 1)) * (kXB / 4));
        dstq[0] = v.x;
        dstq[1] = v.y;
        dstq[2] = v.z;
        dstq[3] = v.w;
    };
#ifdef VSZIP_E3_DIAG_NO_BACKTRACK  // (timing diagnostics only: no backtrack, no output)
    if (nblk > 0) return;
#endif
    load_codes(nblk - 1, qn);
    // the taps of the block walked before this one, in flight: (a, b) at +-dir of r1p / r1n, (c, d) at +-3 dir of r3p / r3n
    float tpa = 0.f, tpb = 0.f, tpc = 0.f, tpd = 0.f;
    int pxx = -1;
    bool pcubic = false;
    auto emit_prev = [&]() __attribute__((always_inline)) {
        if (pxx >= 0) out[pxx] = pcubic ? 0.5625f * (tpa + tpb) - 0.0625f * (tpc + tpd) : (tpa + tpb) * 0.5f;
    };
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int xb = blk * kXB;
        const int xe = min(kXB, w - xb);
        uint32_t q[kXB / 16];
#pragma unroll
        for (int i = 0; i < kXB / 16; ++i) q[i] = qn[i];
        if (blk > 0) load_codes(blk - 1, qn);
        // the chain variable is mdis + fpath (the lane that holds the current direction's codes): per column one v_readlane, one signed two-bit
        // field extract and one add on the scalar unit; the column's value goes to its lane with v_writelane, off the chain
        int fpm = __builtin_amdgcn_readfirstlane(carry);
        int fpv = 0;  // lane xl: mdis + fpath of column xb + xl
        auto walk = [&](auto full) {
            static_for<0, kXB - 1, 1>([&](auto ic) __attribute__((always_inline)) {
                constexpr int xl = kXB - 1 - decltype(ic)::value;
                if (decltype(full)::value || xl < xe) {
                    if (!decltype(full)::value && xb + xl == w - 1) {
                        fpm = mdis;
                    } else {
                        const int word = __builtin_amdgcn_readlane((int)q[xl >> 4], fpm);
                        fpm += (int)((uint32_t)word << (2 * (xl & 15))) >> 30;  // s_bfe_i32: 0, +1, or 3 = -1
                    }
                    fpv = write_lane<xl>(fpv, fpm);
                }
            });
        };
        if (xe == kXB && blk != nblk - 1)
            walk(std::true_type{});
        else
            walk(std::false_type{});
        carry = fpm;
        [[maybe_unused]] uint64_t obm = ~0ull;
        if constexpr (MCLIP) obm = block_bmask(blk);
        emit_prev();  // (the block to the right: its taps were requested a walk ago)
        pxx = -1;
        if (lane < xe) {
            const int xx = xb + lane;
            const int dir = (!MCLIP || ((obm >> lane) & 1ull)) ? fpv - mdis : 0, ad = abs(dir);  // (:567-577: the path outside bmask is 0)
            dmap[xx] = dir;
            pxx = xx;
            pcubic = xx >= ad * 3 && xx + ad * 3 <= w - 1;
            tpa = rowv(r1p, xx + dir, w);
            tpb = rowv(r1n, xx - dir, w);
            if (pcubic) {
                tpc = rowv(r3p, xx + dir * 3, w);
                tpd = rowv(r3n, xx - dir * 3, w);
            }
        }
    }
    emit_prev();
}

// ---------------------------------------------------------------------------
// General line kernel: everything eedi3_line_kernel leaves out — mdis up to 40 (81 directions),
// hp=True (interpLineHP :619-904: half-pel directions, 4*mdis+1 of them, transitions up to +-2)
// and mclip (buildBmask :285-304 + the masked branches of :468-484 / :779-795). Same structure
// (one wave per line; cost phase lanes = x, DP phase lanes = directions), but 32-column blocks
// and up to three DP states per lane so that 161 directions fit one wave; neighbour states
// cross the lane boundary with v_readlane. Built for completeness of the EEDI3 signature, not
// tuned: one direction per cost pass.
// ---------------------------------------------------------------------------
constexpr int kGMaxMdis = 40;
constexpr int kGXB = 32;                                      // columns per block
constexpr int kGNit = 2;                                      // 64-entry iterations of the t_base / window steps: spans reach 32 + 80 + 6
constexpr int kGRowW = 64 * kGNit + 2 * kGMaxMdis + 8;        // block + reach each side, padded for the fixed-count overshoot
constexpr int kGTbMax = 64 * kGNit + 8;

struct GExtra {
    const uint8_t *mask[kMaxPlanesE];  // mclip rows (NULL: no mask), geometry of the (transposed) source plane
    int mstride[kMaxPlanesE];
    int hp;
};

// TPMAX: the most directions the instantiation serves (81: mdis <= 40, or hp with mdis <= 20; 161: hp
// with mdis up to 40) — it sizes the cost tile, the back-pointer tile and the DP states per lane, i.e.
// the LDS per wave: 20 KiB instead of 36 for the 81-direction geometries.
template <int NRAD, bool HP, int TPMAX>
__global__ __launch_bounds__(64) void eedi3_line_general_kernel(const EParams prm, const GExtra ex) {
    constexpr int kGNS = (TPMAX + 63) / 64;  // DP states per lane
    const float kFltMax09 = FLT_MAX * 0.9f;
    __shared__ float rows[4][kGRowW];  // r3p, r1p, r1n, r3n
    __shared__ float hrow[HP ? 4 : 1][HP ? kGRowW : 1];  // their half-pel rows (computeHpRow :602-617), HP only
    __shared__ float tbq[kU][kGTbMax], wsq[kU][kGTbMax];  // t_base and window sums of the kU directions of a pass
    __shared__ float tbhq[HP ? kU / 2 : 1][HP ? kGTbMax : 1], whq[HP ? kU / 2 : 1][HP ? kGTbMax : 1];  // hp: their half-pel twins (odd directions)
    __shared__ float ctile[TPMAX][kGXB + 1];
    __shared__ uint8_t bmt[kGXB];
    __shared__ int any_mask;

    int pi = 0;
    const int gl = blockIdx.x;
#pragma unroll 1
    for (int i = 1; i < prm.nplanes; ++i)
        if (gl >= prm.p[i].line0) pi = i;
    const EPlane pl = prm.p[pi];
    const int off = gl - pl.line0;
    const int line = prm.field + 2 * off;
    constexpr int nrad = NRAD;
    const int w = pl.w, mdis = prm.mdis;
    const int cen = HP ? 2 * mdis : mdis, tpitch = 2 * cen + 1;
    const int lane = threadIdx.x;
    const bool dh = prm.dh != 0;
    const float *r3p = pl.src + (size_t)src_col(dh, line - 3, pl.n_src) * pl.sstride;
    const float *r1p = pl.src + (size_t)src_col(dh, line - 1, pl.n_src) * pl.sstride;
    const float *r1n = pl.src + (size_t)src_col(dh, line + 1, pl.n_src) * pl.sstride;
    const float *r3n = pl.src + (size_t)src_col(dh, line + 3, pl.n_src) * pl.sstride;
    float *out = pl.dst + (size_t)line * pl.dstride;
    int *dmap = pl.dmap + (size_t)off * w;
    const uint8_t *maskp = ex.mask[pi] ? ex.mask[pi] + (size_t)(dh ? off : line) * ex.mstride[pi] : nullptr;
    const int reach = 2 * mdis + nrad + 2, roww = kGXB + 2 * reach;

    // bmask[x] = any mask sample within +-mdis of x (the running `last` of buildBmask in closed form)
    auto bmask_at = [&](int x) -> bool {
        const int lo = max(x - mdis, 0), hi = min(x + mdis, w - 1);
        bool m = false;
        for (int q = lo; q <= hi; ++q) m = m || (maskp[q] != 0);
        return m;
    };
    if (maskp) {
        if (lane == 0) any_mask = 0;
        __syncthreads();
        bool m = false;
        for (int q = lane; q < w; q += 64) m = m || (maskp[q] != 0);
        if (m) any_mask = 1;
        __syncthreads();
        if (!any_mask) {  // :361-373 / :637-649: nothing to connect, plain vertical cubic
            for (int x = lane; x < w; x += 64) {
                dmap[x] = 0;
                out[x] = 0.5625f * (r1p[x] + r1n[x]) - 0.0625f * (r3p[x] + r3n[x]);
            }
            return;
        }
    }

    float pc[kGNS];  // DP states ui = lane + 64 * s (out-of-range states stay at the sentinel)
    int8_t lastbd[kGNS];
    uint32_t held[kGNS][kGXB / 4];  // back-pointer bytes of the previous block, waiting for their last column
#pragma unroll
    for (int s = 0; s < kGNS; ++s) {
        pc[s] = kFltMax09;
        lastbd[s] = 0;
#pragma unroll
        for (int i = 0; i < kGXB / 4; ++i) held[s][i] = 0;
    }
    // back-pointers of this line: [block][direction][32 columns], one signed byte each
    uint8_t *pbk = reinterpret_cast<uint8_t *>(pl.pback) + (size_t)off * ((w + kXB - 1) / kXB * kXB) * tpitch;
    const float g1 = HP ? prm.gamma * 0.5f : prm.gamma, g2 = prm.gamma;
    const int nvec = (tpitch / 8) * 8;
    const int nblk = (w + kGXB - 1) / kGXB;
    for (int blk = 0; blk < nblk; ++blk) {
        const int xb = blk * kGXB;
        const int c0 = xb - reach;
        __syncthreads();
        for (int t = lane; t < roww; t += 64) {
            const int c = min(c0 + t, w - 1 + reach);
            rows[0][t] = rowv(r3p, c, w);
            rows[1][t] = rowv(r1p, c, w);
            rows[2][t] = rowv(r1n, c, w);
            rows[3][t] = rowv(r3n, c, w);
        }
        if (maskp && lane < kGXB && xb + lane < w) bmt[lane] = bmask_at(xb + lane) ? 1 : 0;
        __syncthreads();
        if (HP) {
            for (int t = lane + 1; t < roww - 2; t += 64) {
#pragma unroll
                for (int r = 0; r < 4; ++r) hrow[r][t] = 0.5625f * (rows[r][t] + rows[r][t + 1]) - 0.0625f * (rows[r][t - 1] + rows[r][t + 2]);
            }
            __syncthreads();
        }
        const bool act = lane < kGXB && xb + lane < w;
        const int lx = lane + reach;
        // ---- cost phase ---------------------------------------------------------------
        if constexpr (!HP) {
            // four directions per pass, like eedi3_line_kernel: independent LDS round trips in flight
            for (int ug = -cen; ug <= cen; ug += kU) {
                int uu[kU], jlo[kU];
#pragma unroll
                for (int i = 0; i < kU; ++i) {
                    const int u = min(ug + i, cen);  // past +cen: a duplicate of the last direction (same values, same slots)
                    uu[i] = u;
                    jlo[i] = min(u, min(0, 2 * u)) - nrad;  // t_base columns xb+jlo .. xb+31+max(u, 0, 2u)+nrad are read back
                }
#pragma unroll
                for (int it = 0; it < kGNit; ++it) {
                    const int t = lane + 64 * it;  // fixed trip count, entries past the span are padding nobody reads
                    float val[kU];
#pragma unroll
                    for (int i = 0; i < kU; ++i) {
                        const int two_u = 2 * uu[i];
                        const int j = jlo[i] + t + reach;
                        val[i] = fabsf(rows[0][j] - rows[1][j - two_u]) + fabsf(rows[1][j] - rows[2][j - two_u]) + fabsf(rows[2][j] - rows[3][j - two_u]);
                    }
#pragma unroll
                    for (int i = 0; i < kU; ++i) tbq[i][t] = val[i];
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < kGNit; ++it) {
                    const int t = lane + nrad + 64 * it;
                    float val[kU];
#pragma unroll
                    for (int i = 0; i < kU; ++i) {
                        float sw = 0.0f;
#pragma unroll
                        for (int k = -nrad; k <= nrad; ++k) sw += tbq[i][t + k];
                        val[i] = sw;
                    }
#pragma unroll
                    for (int i = 0; i < kU; ++i) wsq[i][t] = val[i];
                }
                __syncthreads();
                if (act) {
#pragma unroll
                    for (int i = 0; i < kU; ++i) {
                        const int u = uu[i], two_u = 2 * u;
                        const int base = lane - jlo[i];
                        const float sw1 = wsq[i][base], sw0 = wsq[i][base + u], sw2 = wsq[i][base + two_u];
                        const float ip = (rows[1][lx + u] + rows[2][lx - u]) * 0.5f;
                        const float v = fabsf(rows[1][lx] - ip) + fabsf(rows[2][lx] - ip);
                        ctile[cen + u][lane] = prm.alpha * (sw0 + sw1 + sw2) + prm.beta * (float)abs(u) + prm.one_minus_ab * v;
                    }
                }
                __syncthreads();
            }
        }
        if constexpr (HP) {
            // :659-702 — u in half pels; baseM pairs full-pel rows shifted by u, baseHp (odd u) pairs the
            // half-pel rows; s1/s2 window baseM at x and x+u, s0 the u/2 neighbour. Four directions per
            // pass (-cen is even, so slots 1 and 3 are the odd ones and own the two half-pel arrays).
            for (int ug = -cen; ug <= cen; ug += kU) {
                int uu[kU], jlo[kU];
#pragma unroll
                for (int i = 0; i < kU; ++i) {
                    const int u = min(ug + i, cen);  // past +cen: duplicates of the last (even) direction
                    uu[i] = u;
                    jlo[i] = min(0, min(u, u >> 1)) - nrad;  // columns xb+jlo .. xb+31+max(0, u, uh)+nrad are read back
                }
#pragma unroll
                for (int it = 0; it < kGNit; ++it) {
                    const int t = lane + 64 * it;  // fixed trip count, entries past the span are padding nobody reads
                    float vm[kU], vh[kU / 2];
#pragma unroll
                    for (int i = 0; i < kU; ++i) {
                        const int u = uu[i], j = jlo[i] + t + reach;
                        vm[i] = fabsf(rows[0][j] - rows[1][j - u]) + fabsf(rows[1][j] - rows[2][j - u]) + fabsf(rows[2][j] - rows[3][j - u]);
                        if (i & 1) vh[i >> 1] = fabsf(hrow[0][j] - hrow[1][j - u]) + fabsf(hrow[1][j] - hrow[2][j - u]) + fabsf(hrow[2][j] - hrow[3][j - u]);
                    }
#pragma unroll
                    for (int i = 0; i < kU; ++i) {
                        tbq[i][t] = vm[i];
                        if (i & 1) tbhq[i >> 1][t] = vh[i >> 1];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < kGNit; ++it) {
                    const int t = lane + nrad + 64 * it;
                    float sm[kU], sh[kU / 2];
#pragma unroll
                    for (int i = 0; i < kU; ++i) {
                        float am = 0.0f, ah = 0.0f;
#pragma unroll
                        for (int k = -nrad; k <= nrad; ++k) {
                            am += tbq[i][t + k];
                            if (i & 1) ah += tbhq[i >> 1][t + k];
                        }
                        sm[i] = am;
                        if (i & 1) sh[i >> 1] = ah;
                    }
#pragma unroll
                    for (int i = 0; i < kU; ++i) {
                        wsq[i][t] = sm[i];
                        if (i & 1) whq[i >> 1][t] = sh[i >> 1];
                    }
                }
                __syncthreads();
                if (act) {
#pragma unroll
                    for (int i = 0; i < kU; ++i) {
                        const int u = uu[i], uh = u >> 1;
                        const bool odd = (u & 1) != 0;  // false for the duplicates in slots 1 / 3 of the last pass
                        const int lo0 = odd ? -uh - 1 : -uh;
                        const int base = lane - jlo[i];
                        const float s1 = wsq[i][base], s2 = wsq[i][base + u];
                        float s0 = wsq[i][base + uh];
                        if ((i & 1) && odd) s0 = whq[i >> 1][base + uh];
                        const float bq = odd ? hrow[1][lx + uh] : rows[1][lx + uh];
                        const float cq = odd ? hrow[2][lx + lo0] : rows[2][lx + lo0];
                        const float ip = (bq + cq) * 0.5f;
                        const float v = fabsf(rows[1][lx] - ip) + fabsf(rows[2][lx] - ip);
                        ctile[cen + u][lane] = prm.alpha * (s0 + s1 + s2) + (prm.beta * (float)abs(u) * 0.5f) + prm.one_minus_ab * v;
                    }
                }
                __syncthreads();
            }
        }
        // ---- DP phase --------------------------------------------------------------------
        // As in eedi3_line_kernel the chain touches no memory: the block's costs (32 columns x kGNS
        // states per lane) come into registers first, the steps are unrolled, a step's back-pointer
        // (-2..2, one signed byte) is packed into registers and the block's 32 bytes per state go to
        // global memory afterwards ([block][direction][32 columns]); the mask bits of the block's
        // columns are one ballot.
        const int xe = min(kGXB, w - xb);
        float tcv[kGNS][kGXB];
#pragma unroll
        for (int s2 = 0; s2 < kGNS; ++s2)
#pragma unroll
            for (int xl = 0; xl < kGXB; ++xl) tcv[s2][xl] = (lane + 64 * s2 < tpitch) ? ctile[min(lane + 64 * s2, tpitch - 1)][xl] : 0.0f;
        const unsigned long long bmbits = maskp ? __ballot(lane < kGXB && bmt[lane & (kGXB - 1)] != 0) : ~0ull;
        uint32_t pk[kGNS][kGXB / 4];
#pragma unroll
        for (int s2 = 0; s2 < kGNS; ++s2)
#pragma unroll
            for (int i = 0; i < kGXB / 4; ++i) pk[s2][i] = 0;
        auto store_block = [&](uint32_t (*q)[kGXB / 4], int b2) {
#pragma unroll
            for (int s2 = 0; s2 < kGNS; ++s2) {
                const int ui = lane + 64 * s2;
                if (ui < tpitch) {
                    uint4 *dst = reinterpret_cast<uint4 *>(pbk + ((size_t)b2 * tpitch + ui) * kGXB);
                    dst[0] = make_uint4(q[s2][0], q[s2][1], q[s2][2], q[s2][3]);
                    dst[1] = make_uint4(q[s2][4], q[s2][5], q[s2][6], q[s2][7]);
                }
            }
        };
#pragma unroll
        for (int xl = 0; xl < kGXB; ++xl) {
            if (xl < xe) {
                const int xx = xb + xl;
                float tc[kGNS];
#pragma unroll
                for (int s2 = 0; s2 < kGNS; ++s2) tc[s2] = tcv[s2][xl];
                if (xx == 0) {
#pragma unroll
                    for (int s2 = 0; s2 < kGNS; ++s2) pc[s2] = (lane + 64 * s2 < tpitch) ? tc[s2] : kFltMax09;
                } else {
                    const bool masked = ((bmbits >> xl) & 1ull) == 0;
                    float pn[kGNS];
                    int8_t bdn[kGNS];
                    if (masked) {  // :474-484 / :785-795
#pragma unroll
                        for (int s = 0; s < kGNS; ++s) {
                            const bool valid = lane + 64 * s < tpitch;
                            pn[s] = xx == 1 ? (valid ? tc[s] : kFltMax09) : pc[s];
                            bdn[s] = xx == 1 ? (int8_t)0 : lastbd[s];
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < kGNS; ++s) {
                            const int ui = lane + 64 * s;
                            const bool valid = ui < tpitch;
                            // neighbours ui-1, ui+1 (and ui-2, ui+2): DPP shift inside the lane group, readlane across it
                            const float e_b1 = s > 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc[s > 0 ? s - 1 : 0]), 63)) : kFltMax09;
                            const float e_a1 = s < kGNS - 1 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc[s < kGNS - 1 ? s + 1 : s]), 0)) : kFltMax09;
                            const float b1 = lane_below(pc[s], e_b1), a1 = lane_above(pc[s], e_a1);
                            float bval;
                            int bd;
                            if (!HP) {
                                const float left = b1 + g2, right = a1 + g2;  // :536-548
                                bval = pc[s];
                                bd = 0;
                                if (left < bval) {
                                    bval = left;
                                    bd = -1;
                                }
                                if (right < bval) {
                                    bval = right;
                                    bd = 1;
                                }
                            } else {
                                const float e_b2 = s > 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc[s > 0 ? s - 1 : 0]), 62)) : kFltMax09;
                                const float e_a2 = s < kGNS - 1 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc[s < kGNS - 1 ? s + 1 : s]), 1)) : kFltMax09;
                                const float b2 = lane_below(b1, e_b2), a2 = lane_above(a1, e_a2);
                                const float c_m2 = b2 + g2, c_m1 = b1 + g1, c_0 = pc[s], c_p1 = a1 + g1, c_p2 = a2 + g2;
                                if (ui < nvec) {  // vector body :806-832 starts from the -2 candidate
                                    bval = c_m2;
                                    bd = -2;
                                } else {  // scalar tail :834-849 starts from the sentinel
                                    bval = kFltMax09;
                                    bd = 0;
                                    if (c_m2 < bval) {
                                        bval = c_m2;
                                        bd = -2;
                                    }
                                }
                                if (c_m1 < bval) {
                                    bval = c_m1;
                                    bd = -1;
                                }
                                if (c_0 < bval) {
                                    bval = c_0;
                                    bd = 0;
                                }
                                if (c_p1 < bval) {
                                    bval = c_p1;
                                    bd = 1;
                                }
                                if (c_p2 < bval) {
                                    bval = c_p2;
                                    bd = 2;
                                }
                            }
                            pn[s] = valid ? fminf(bval + tc[s], kFltMax09) : kFltMax09;
                            bdn[s] = (int8_t)bd;
                        }
                    }
#pragma unroll
                    for (int s = 0; s < kGNS; ++s) {
                        pc[s] = pn[s];
                        lastbd[s] = bdn[s];
                        // back-pointer of column xx-1: byte xl-1 of this block, or the last byte of the previous one
                        const uint32_t byte = (uint32_t)(uint8_t)bdn[s];
                        if (xl == 0)
                            held[s][kGXB / 4 - 1] |= byte << 24;
                        else
                            pk[s][(xl - 1) >> 2] |= byte << (8 * ((xl - 1) & 3));
                    }
                }
            }
        }
        if (blk > 0) store_block(held, blk - 1);
#pragma unroll
        for (int s2 = 0; s2 < kGNS; ++s2)
#pragma unroll
            for (int i = 0; i < kGXB / 4; ++i) held[s2][i] = pk[s2][i];
        if (blk == nblk - 1) store_block(held, blk);
    }
    __syncthreads();
    __threadfence_block();

    // ---- backtrack + output --------------------------------------------------------------
    // fpath as a scalar chain over register-held codes, like eedi3_line_kernel: state cen+fpath sits
    // in lane (cen+fpath) & 63 of state group (cen+fpath) >> 6
    int carry = 0;
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int xb = blk * kGXB;
        const int xe = min(kGXB, w - xb);
        __syncthreads();
        if (maskp && lane < xe) bmt[lane] = bmask_at(xb + lane) ? 1 : 0;
        uint32_t q[kGNS][kGXB / 4];
#pragma unroll
        for (int s2 = 0; s2 < kGNS; ++s2) {
            const uint4 *src = reinterpret_cast<const uint4 *>(pbk + ((size_t)blk * tpitch + min(lane + 64 * s2, tpitch - 1)) * kGXB);
            const uint4 v0 = src[0], v1 = src[1];
            q[s2][0] = v0.x, q[s2][1] = v0.y, q[s2][2] = v0.z, q[s2][3] = v0.w;
            q[s2][4] = v1.x, q[s2][5] = v1.y, q[s2][6] = v1.z, q[s2][7] = v1.w;
        }
        __syncthreads();
        int fp = __builtin_amdgcn_readfirstlane(carry);
        int fpv = 0;
#pragma unroll
        for (int xl = kGXB - 1; xl >= 0; --xl) {
            if (xl < xe) {
                if (xb + xl == w - 1) {
                    fp = 0;
                } else {
                    const int ui = cen + fp, l = ui & 63, sg = ui >> 6;
                    int word = __builtin_amdgcn_readlane((int)q[0][xl >> 2], l);
                    if (kGNS > 1 && sg == 1) word = __builtin_amdgcn_readlane((int)q[kGNS > 1 ? 1 : 0][xl >> 2], l);
                    if (kGNS > 2 && sg == 2) word = __builtin_amdgcn_readlane((int)q[kGNS > 2 ? 2 : 0][xl >> 2], l);
                    fp += (int)(int8_t)((word >> (8 * (xl & 3))) & 0xff);
                }
                fpv = lane == xl ? fp : fpv;
            }
        }
        carry = fp;
        if (lane < xe) {
            const int xx = xb + lane;
            const bool masked = maskp && !bmt[lane];
            int dir = fpv;
            float v;
            if (!HP) {
                if (masked) dir = 0;  // :566-569
                const int ad = abs(dir);
                if (xx >= ad * 3 && xx + ad * 3 <= w - 1)
                    v = 0.5625f * (rowv(r1p, xx + dir, w) + rowv(r1n, xx - dir, w)) - 0.0625f * (rowv(r3p, xx + dir * 3, w) + rowv(r3n, xx - dir * 3, w));
                else
                    v = (rowv(r1p, xx + dir, w) + rowv(r1n, xx - dir, w)) * 0.5f;
            } else if (masked) {  // :866-870
                dir = 0;
                v = 0.5625f * (r1p[xx] + r1n[xx]) - 0.0625f * (r3p[xx] + r3n[xx]);
            } else if ((dir & 1) == 0) {  // :874-881
                const int d2 = dir >> 1, ad = abs(d2);
                if (xx >= ad * 3 && xx + ad * 3 <= w - 1)
                    v = 0.5625f * (rowv(r1p, xx + d2, w) + rowv(r1n, xx - d2, w)) - 0.0625f * (rowv(r3p, xx + d2 * 3, w) + rowv(r3n, xx - d2 * 3, w));
                else
                    v = (rowv(r1p, xx + d2, w) + rowv(r1n, xx - d2, w)) * 0.5f;
            } else {  // :882-901
                const int d20 = dir >> 1, d21 = (dir + 1) >> 1, d30 = (dir * 3) >> 1, d31 = (dir * 3 + 1) >> 1;
                const int ad = max(abs(d30), abs(d31));
                if (xx >= ad && xx + ad <= w - 1) {
                    const float q0 = rowv(r3p, xx + d30, w) + rowv(r3p, xx + d31, w);
                    const float q1 = rowv(r1p, xx + d20, w) + rowv(r1p, xx + d21, w);
                    const float q2 = rowv(r1n, xx - d20, w) + rowv(r1n, xx - d21, w);
                    const float q3 = rowv(r3n, xx - d30, w) + rowv(r3n, xx - d31, w);
                    v = 0.28125f * (q1 + q2) - 0.03125f * (q0 + q3);
                } else {
                    v = (rowv(r1p, xx + d20, w) + rowv(r1p, xx + d21, w) + rowv(r1n, xx - d20, w) + rowv(r1n, xx - d21, w)) * 0.25f;
                }
            }
            dmap[xx] = dir;
            out[xx] = v;
        }
    }
}

// Copy the kept field (processPlane :41-53).
__global__ void eedi3_copy_kernel(const EParams prm) {
    const EPlane pl = prm.p[blockIdx.z];
    const int k = blockIdx.y;
    if (k >= pl.n_src) return;
    int dl;
    if (prm.dh) {
        dl = 2 * k + (1 - prm.field);
    } else {
        if ((k & 1) != ((1 - prm.field) & 1)) return;
        dl = k;
    }
    const float *s = pl.src + (size_t)k * pl.sstride;
    float *d = pl.dst + (size_t)dl * pl.dstride;
    // 16 bytes per lane where the rows allow it (every VapourSynth frame): a quarter of the instructions per byte
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
        const int n4 = pl.w >> 2;
        const float4 *s4 = reinterpret_cast<const float4 *>(s);
        float4 *d4 = reinterpret_cast<float4 *>(d);
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n4; x += gridDim.x * blockDim.x) d4[x] = s4[x];
        for (int x = 4 * n4 + blockIdx.x * blockDim.x + threadIdx.x; x < pl.w; x += gridDim.x * blockDim.x) d[x] = s[x];
    } else {
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < pl.w; x += gridDim.x * blockDim.x) d[x] = s[x];
    }
}

struct VParams {
    EPlane p[kMaxPlanesE];
    const float *scp[kMaxPlanesE];
    int scstride[kMaxPlanesE];
    int field, dh, vcheck, hp;
    float vthresh2, rcp0, rcp1, rcp2;
    float *gline;  // wide kernel, lines that do not fit LDS: 2 lines per plane in global memory (NULL: LDS)
    int gline_pitch;
    int plane_base;  // first plane slot of this launch (eedi3_vcheck_lds_kernel)
};

// vcheckLine (:915-1046), hp = false. Line pd blends against line pd-2 AS ALREADY BLENDED, so
// the lines of a plane form a sequential chain; one workgroup per plane walks it, 1024 columns
// at a time. Only two terms of a pixel depend on the chain (d2p[i+dir], twice): everything
// else — nine gathers and most of the arithmetic — is prepared one line ahead in registers
// while the current line resolves, and the blended line is handed to the next iteration
// through LDS, so the per-line critical path is one LDS read, ~30 ALU ops, one LDS write and
// one barrier instead of a round trip through global memory.
constexpr int kVcCols = 4;  // columns per thread: lines up to 4096 wide

struct VcPre {
    float cint, dl_i, d1p_i, d1n_i, dl_imd, d1p_ipd, t_ipd, ib, vb, vc, a2;
    int ipd;   // < 0: the pixel keeps cint (no direction / inconsistent neighbours / too close to the edge)
    int ipd2;  // hp, odd direction: the second of the two columns averaged (:969-985); -1 otherwise
};

__device__ __forceinline__ void vcheck_prepare(const EPlane &pl, const VParams &prm, const float *scp, int scstride, int off, int i, VcPre &q) {
    const int L = pl.w;
    const bool dh = prm.dh != 0;
    const int pd = prm.field + 2 * off;
    const float *dl = pl.dst + (size_t)pd * pl.dstride;
    const float *d1p = pl.dst + (size_t)(pd - 1) * pl.dstride, *d1n = pl.dst + (size_t)(pd + 1) * pl.dstride;
    const float *d2n = pl.dst + (size_t)(pd + 2) * pl.dstride;
    const float *d3p = pl.src + (size_t)src_col(dh, pd - 3, pl.n_src) * pl.sstride;
    const float *d3n = pl.src + (size_t)src_col(dh, pd + 3, pl.n_src) * pl.sstride;
    const int *dc = pl.dmap + (size_t)off * L, *dp = pl.dmap + (size_t)(off - 1) * L, *dn = pl.dmap + (size_t)(off + 1) * L;
    const int dirc = dc[i];
    q.d1p_i = d1p[i];
    q.d1n_i = d1n[i];
    q.cint = scp ? scp[(size_t)pd * scstride + i] : 0.5625f * (q.d1p_i + q.d1n_i) - 0.0625f * (d3p[i] + d3n[i]);
    q.dl_i = dl[i];
    q.ipd = -1;
    q.ipd2 = -1;
    const int dirt = dp[i], dirb = dn[i];
    const bool hp = prm.hp != 0;
    const int maxoff = !hp ? abs(dirc) : (((dirc & 1) == 0) ? abs(dirc >> 1) : max(abs(dirc >> 1), abs((dirc + 1) >> 1)));
    if (dirc != 0 && !(max(dirc * dirt, dirc * dirb) < 0 || (dirt == dirb && dirt == 0)) && !(i + maxoff >= L || i - maxoff < 0)) {
        int dabs;
        if (hp && (dirc & 1) != 0) {  // :969-985: both half-pel neighbours, summed
            const int d20 = dirc >> 1, d21 = (dirc + 1) >> 1;
            const int ip0 = i + d20, ip1 = i + d21, im0 = i - d20, im1 = i - d21;
            q.ipd = ip0;
            q.ipd2 = ip1;
            const float s1p = d1p[ip0] + d1p[ip1], pa0 = dl[ip0] + dl[ip1], ps0 = dl[im0] + dl[im1];
            const float s1n = d1n[im0] + d1n[im1], s2n = d2n[im0] + d2n[im1];
            q.dl_imd = ps0;
            q.d1p_ipd = s1p;
            q.t_ipd = fabsf(pa0 - s1p);
            q.ib = (pa0 + s2n) * 0.25f;
            q.vb = (fabsf(s2n - s1n) + fabsf(ps0 - s1n)) * 0.5f;
            dabs = abs(dirc) >> 1;
        } else {
            const int offh = hp ? dirc >> 1 : dirc;
            const int ipd = i + offh, imd = i - offh;
            q.ipd = ipd;
            q.dl_imd = dl[imd];
            q.d1p_ipd = d1p[ipd];
            const float dl_ipd = dl[ipd], d2n_imd = d2n[imd], d1n_imd = d1n[imd];
            q.t_ipd = fabsf(dl_ipd - q.d1p_ipd);
            q.ib = (dl_ipd + d2n_imd) * 0.5f;
            q.vb = fabsf(d2n_imd - d1n_imd) + fabsf(q.dl_imd - d1n_imd);
            dabs = hp ? abs(dirc) >> 1 : abs(dirc);
        }
        q.vc = fabsf(q.dl_i - q.d1p_i) + fabsf(q.dl_i - q.d1n_i);
        q.a2 = fmaxf((prm.vthresh2 - (float)dabs) * prm.rcp2, 0.0f);
    }
}

__device__ __forceinline__ float vcheck_resolve(const VParams &prm, const VcPre &q, const float *prev) {
    if (q.ipd < 0) return q.cint;
    float it, vt;
    if (q.ipd2 >= 0) {
        const float s2p = prev[q.ipd] + prev[q.ipd2];
        it = (s2p + q.dl_imd) * 0.25f;
        vt = (fabsf(s2p - q.d1p_ipd) + q.t_ipd) * 0.5f;
    } else {
        const float d2 = prev[q.ipd];
        it = (d2 + q.dl_imd) * 0.5f;
        vt = fabsf(d2 - q.d1p_ipd) + q.t_ipd;
    }
    const float e0 = fabsf(it - q.d1p_i), e1 = fabsf(q.ib - q.d1n_i), e2 = fabsf(vt - q.vc), e3 = fabsf(q.vb - q.vc);
    float m0, m1;
    if (prm.vcheck == 1) {
        m0 = fminf(e0, e1);
        m1 = fminf(e2, e3);
    } else if (prm.vcheck == 2) {
        m0 = (e0 + e1) * 0.5f;
        m1 = (e2 + e3) * 0.5f;
    } else {
        m0 = fmaxf(e0, e1);
        m1 = fmaxf(e2, e3);
    }
    const float a0 = m0 * prm.rcp0, a1 = m1 * prm.rcp1;
    const float a = fminf(fmaxf(a0, fmaxf(a1, q.a2)), 1.0f);
    return (1.0f - a) * q.dl_i + a * q.cint;
}

__global__ __launch_bounds__(1024) void eedi3_vcheck_kernel(const VParams prm) {
    __shared__ float prevl[2][1024 * kVcCols];
    const EPlane pl = prm.p[blockIdx.x];
    const float *scp = prm.scp[blockIdx.x];
    const int scstride = prm.scstride[blockIdx.x];
    const int L = pl.w;
    const int tid = threadIdx.x;
    auto processed = [&](int off) {
        const int pd = prm.field + 2 * off;
        return off >= 1 && off + 1 < pl.n_interp && pd >= 2 && pd + 2 < pl.n_dst;
    };
    int cur = 0;
    bool have_prev = false;  // prevl[cur] holds the blended line pd-2
    VcPre q[kVcCols], qn[kVcCols];
    int off = 1;
    while (off + 1 < pl.n_interp && !processed(off)) ++off;
    if (off + 1 < pl.n_interp) {
#pragma unroll
        for (int c = 0; c < kVcCols; ++c)
            if (tid + c * 1024 < L) vcheck_prepare(pl, prm, scp, scstride, off, tid + c * 1024, q[c]);
    }
    for (; off + 1 < pl.n_interp; ++off) {
        if (!processed(off)) {
            have_prev = false;
            continue;
        }
        const int pd = prm.field + 2 * off;
        float *dl = pl.dst + (size_t)pd * pl.dstride;
        if (!have_prev) {  // line pd-2 was not blended by this pass: take it from memory
            const float *d2p = pl.dst + (size_t)(pd - 2) * pl.dstride;
#pragma unroll
            for (int c = 0; c < kVcCols; ++c)
                if (tid + c * 1024 < L) prevl[cur][tid + c * 1024] = d2p[tid + c * 1024];
            __syncthreads();
        }
        // the next line's chain-independent part, issued before this line resolves
        const bool next = processed(off + 1);
        if (next) {
#pragma unroll
            for (int c = 0; c < kVcCols; ++c)
                if (tid + c * 1024 < L) vcheck_prepare(pl, prm, scp, scstride, off + 1, tid + c * 1024, qn[c]);
        }
#pragma unroll
        for (int c = 0; c < kVcCols; ++c) {
            const int i = tid + c * 1024;
            if (i < L) {
                const float r = vcheck_resolve(prm, q[c], prevl[cur]);
                prevl[cur ^ 1][i] = r;
                dl[i] = r;
            }
        }
        __syncthreads();
        cur ^= 1;
        have_prev = true;
        if (next) {
#pragma unroll
            for (int c = 0; c < kVcCols; ++c) q[c] = qn[c];
        } else if (off + 2 < pl.n_interp) {
            // a skipped line follows: prepare the one after it when its turn comes
            int o2 = off + 1;
            while (o2 + 1 < pl.n_interp && !processed(o2)) ++o2;
            if (o2 + 1 < pl.n_interp) {
#pragma unroll
                for (int c = 0; c < kVcCols; ++c)
                    if (tid + c * 1024 < L) vcheck_prepare(pl, prm, scp, scstride, o2, tid + c * 1024, q[c]);
            }
        }
    }
}

// vcheckLine for lines of at most 1920 columns, rows in LDS. The kernel above sends every load
// of a plane through one CU's memory pipeline (32 gather instructions per thread and line), and
// that pipeline, not the chain's latency, sets its pace. Consecutive lines share five of the seven
// dst rows they touch (pd-3 .. pd+3; the +-3 rows are the kept field lines the reference reads
// from `src`), so the rows live in a 7-slot LDS ring (row r -> slot r mod 7) next to a 4-slot int8
// ring of direction-map rows: a line loads two new rows and one map row with coalesced loads and
// gathers from LDS; the blended line is written back into its ring slot, where the next line
// finds it as d2p.
__device__ __forceinline__ float vcheck_pixel(const VParams &prm, int L, int i, int dirc, int dirt, int dirb, float cint, const float *d1p, const float *d1n,
                                              const float *dl, const float *d2p, const float *d2n) {
    if (dirc == 0) return cint;
    if (max(dirc * dirt, dirc * dirb) < 0 || (dirt == dirb && dirt == 0)) return cint;
    const bool hp = prm.hp != 0;
    const int maxoff = !hp ? abs(dirc) : (((dirc & 1) == 0) ? abs(dirc >> 1) : max(abs(dirc >> 1), abs((dirc + 1) >> 1)));
    if (i + maxoff >= L || i - maxoff < 0) return cint;
    float it, ib, vt, vb;
    int dabs;
    if (hp && (dirc & 1) != 0) {  // :969-985
        const int d20 = dirc >> 1, d21 = (dirc + 1) >> 1;
        const int ip0 = i + d20, ip1 = i + d21, im0 = i - d20, im1 = i - d21;
        const float s2p = d2p[ip0] + d2p[ip1], s1p = d1p[ip0] + d1p[ip1], pa0 = dl[ip0] + dl[ip1], ps0 = dl[im0] + dl[im1];
        const float s1n = d1n[im0] + d1n[im1], s2n = d2n[im0] + d2n[im1];
        it = (s2p + ps0) * 0.25f;
        vt = (fabsf(s2p - s1p) + fabsf(pa0 - s1p)) * 0.5f;
        ib = (pa0 + s2n) * 0.25f;
        vb = (fabsf(s2n - s1n) + fabsf(ps0 - s1n)) * 0.5f;
        dabs = abs(dirc) >> 1;
    } else {
        const int offh = hp ? dirc >> 1 : dirc;
        const int ipd = i + offh, imd = i - offh;
        it = (d2p[ipd] + dl[imd]) * 0.5f;
        ib = (dl[ipd] + d2n[imd]) * 0.5f;
        vt = fabsf(d2p[ipd] - d1p[ipd]) + fabsf(dl[ipd] - d1p[ipd]);
        vb = fabsf(d2n[imd] - d1n[imd]) + fabsf(dl[imd] - d1n[imd]);
        dabs = hp ? abs(dirc) >> 1 : abs(dirc);
    }
    const float vc = fabsf(dl[i] - d1p[i]) + fabsf(dl[i] - d1n[i]);
    const float e0 = fabsf(it - d1p[i]), e1 = fabsf(ib - d1n[i]), e2 = fabsf(vt - vc), e3 = fabsf(vb - vc);
    float m0, m1;
    if (prm.vcheck == 1) {
        m0 = fminf(e0, e1);
        m1 = fminf(e2, e3);
    } else if (prm.vcheck == 2) {
        m0 = (e0 + e1) * 0.5f;
        m1 = (e2 + e3) * 0.5f;
    } else {
        m0 = fmaxf(e0, e1);
        m1 = fmaxf(e2, e3);
    }
    const float a0 = m0 * prm.rcp0, a1 = m1 * prm.rcp1;
    const float a2 = fmaxf((prm.vthresh2 - (float)dabs) * prm.rcp2, 0.0f);
    const float a = fminf(fmaxf(a0, fmaxf(a1, a2)), 1.0f);
    return (1.0f - a) * dl[i] + a * cint;
}

// vcheckLine for very wide lines (L > 4096, e.g. 8K frames): the chain without any cross-line
// pipelining — every line prepares and resolves its pixels in column chunks of 1024, the blended
// line is handed on through (dynamic) LDS, or through two lines of global scratch per plane when a
// line pair does not fit LDS (L > 8192; a workgroup's own global writes are visible to it after
// __syncthreads). Correct first; not tuned.
__global__ __launch_bounds__(1024) void eedi3_vcheck_wide_kernel(const VParams prm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vsm[];
    const EPlane pl = prm.p[blockIdx.x];
    const float *scp = prm.scp[blockIdx.x];
    const int scstride = prm.scstride[blockIdx.x];
    const int L = pl.w;
    const int tid = threadIdx.x;
    float *prevl[2] = {reinterpret_cast<float *>(vsm), reinterpret_cast<float *>(vsm) + L};
    if (prm.gline) {
        prevl[0] = prm.gline + (size_t)blockIdx.x * 2 * prm.gline_pitch;
        prevl[1] = prevl[0] + prm.gline_pitch;
    }
    int cur = 0;
    bool have_prev = false;
    for (int off = 1; off + 1 < pl.n_interp; ++off) {
        const int pd = prm.field + 2 * off;
        if (pd < 2 || pd + 2 >= pl.n_dst) {
            have_prev = false;
            continue;
        }
        float *dl = pl.dst + (size_t)pd * pl.dstride;
        if (!have_prev) {
            const float *d2p = pl.dst + (size_t)(pd - 2) * pl.dstride;
            for (int i = tid; i < L; i += 1024) prevl[cur][i] = d2p[i];
            __syncthreads();
        }
        for (int i0 = 0; i0 < L; i0 += 4096) {  // results of a chunk go to the other buffer and to memory only after
            float res[4];                       // every pixel of the chunk has read the un-blended line around it
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = i0 + tid + c * 1024;
                if (i < L) {
                    VcPre q;
                    vcheck_prepare(pl, prm, scp, scstride, off, i, q);
                    res[c] = vcheck_resolve(prm, q, prevl[cur]);
                }
            }
            // a pixel gathers dl[] up to mdis columns away, possibly across the chunk boundary: the
            // writes of this chunk wait until the whole line has been computed (tline in the reference)
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = i0 + tid + c * 1024;
                if (i < L) prevl[cur ^ 1][i] = res[c];
            }
        }
        __syncthreads();
        for (int i = tid; i < L; i += 1024) dl[i] = prevl[cur ^ 1][i];  // memcpy(dl, tline) :1044
        __syncthreads();
        cur ^= 1;
        have_prev = true;
    }
}

constexpr int kVcLdsMaxL = 4096;  // widest line of the LDS chain kernel (four columns a thread)
// The per-line step is paced by instruction issue on ONE CU (PMC: 214 scalar + 126 vector
// instructions per wave and line before this form), so the kernel is specialised on hp and the
// vcheck mode, the pixel function is branch-free (a pixel that keeps cint gathers around itself and
// is selected out at the end — no exec-mask regions), and the ring slots of the seven rows are
// kept incrementally instead of as r % 7.
template <bool HP, int VC>
__device__ __forceinline__ float vcheck_pixel_bf(const VParams &prm, int L, int i, int dirc, int dirt, int dirb, float cint, const float *d1p, const float *d1n,
                                                 const float *dl, const float *d2p, const float *d2n) {
    const int maxoff = !HP ? abs(dirc) : (((dirc & 1) == 0) ? abs(dirc >> 1) : max(abs(dirc >> 1), abs((dirc + 1) >> 1)));
    // (bitwise: with && / || the compiler builds an exec-mask region around the later terms)
    const bool act = (dirc != 0) & !((max(dirc * dirt, dirc * dirb) < 0) | ((dirt | dirb) == 0)) & !((i + maxoff >= L) | (i - maxoff < 0));
    const int dsafe = act ? dirc : 0;
    float it, ib, vt, vb;
    int dabs;
    if (HP && (dsafe & 1) != 0) {  // :969-985
        const int d20 = dsafe >> 1, d21 = (dsafe + 1) >> 1;
        const int ip0 = i + d20, ip1 = i + d21, im0 = i - d20, im1 = i - d21;
        const float s2p = d2p[ip0] + d2p[ip1], s1p = d1p[ip0] + d1p[ip1], pa0 = dl[ip0] + dl[ip1], ps0 = dl[im0] + dl[im1];
        const float s1n = d1n[im0] + d1n[im1], s2n = d2n[im0] + d2n[im1];
        it = (s2p + ps0) * 0.25f;
        vt = (fabsf(s2p - s1p) + fabsf(pa0 - s1p)) * 0.5f;
        ib = (pa0 + s2n) * 0.25f;
        vb = (fabsf(s2n - s1n) + fabsf(ps0 - s1n)) * 0.5f;
        dabs = abs(dsafe) >> 1;
    } else {
        const int offh = HP ? dsafe >> 1 : dsafe;
        const int ipd = i + offh, imd = i - offh;
        const float d2p_i = d2p[ipd], dl_p = dl[ipd], d1p_p = d1p[ipd], dl_m = dl[imd], d2n_m = d2n[imd], d1n_m = d1n[imd];
        it = (d2p_i + dl_m) * 0.5f;
        ib = (dl_p + d2n_m) * 0.5f;
        vt = fabsf(d2p_i - d1p_p) + fabsf(dl_p - d1p_p);
        vb = fabsf(d2n_m - d1n_m) + fabsf(dl_m - d1n_m);
        dabs = HP ? abs(dsafe) >> 1 : abs(dsafe);
    }
    const float dl_i = dl[i], d1p_i = d1p[i], d1n_i = d1n[i];
    const float vc = fabsf(dl_i - d1p_i) + fabsf(dl_i - d1n_i);
    const float e0 = fabsf(it - d1p_i), e1 = fabsf(ib - d1n_i), e2 = fabsf(vt - vc), e3 = fabsf(vb - vc);
    float m0, m1;
    if (VC == 1) {
        m0 = fminf(e0, e1);
        m1 = fminf(e2, e3);
    } else if (VC == 2) {
        m0 = (e0 + e1) * 0.5f;
        m1 = (e2 + e3) * 0.5f;
    } else {
        m0 = fmaxf(e0, e1);
        m1 = fmaxf(e2, e3);
    }
    const float a0 = m0 * prm.rcp0, a1 = m1 * prm.rcp1;
    const float a2 = fmaxf((prm.vthresh2 - (float)dabs) * prm.rcp2, 0.0f);
    const float a = fminf(fmaxf(a0, fmaxf(a1, a2)), 1.0f);
    const float blended = (1.0f - a) * dl_i + a * cint;
    return act ? blended : cint;
}

constexpr int kVcNT = 1024;  // threads of a chain workgroup (512 / 256 with more columns a thread measured 18 % / 90 % slower: the step is one wave's instruction stream)
// Round 4. ONE barrier per line: nothing a line writes is anything it reads — the blended lines alternate between two rows of their own (line pd reads
// pd - 2 from one and writes pd to the other), new rows go to slots no read of this line touches. The rings hold only what a line GATHERS from: the rows
// pd -+ 3 are only ever read at the thread's own columns (the cubic of `cint`), so they stay in registers — pd + 3 is the odd row of the prefetch set of line
// off + 1, pd - 3 a copy kept from the set of line off - 2 — and LDS holds three odd rows (pd - 1, pd + 1; pd + 3 written), three even ones (pd, pd + 2;
// pd + 4 written), two blended lines and six map rows: 38 bytes a column, so that FOUR columns a thread (lines up to 4096 samples: the luma of a 4K frame)
// fit the 160 KB. All ring periods divide six lines and the rings are addressed relative to the chain's first line at a fixed row pitch: in the 6x unrolled
// loop every slot is a compile-time offset that folds into the LDS instructions' immediates (the per-line slot arithmetic was a quarter of the step's
// instructions, and the step is bound by one wave's instruction stream: profiles/r04_notes.md section 10).
// The rows a line adds (pd + 3, pd + 4, the map row off + 2, its sclip row) are loaded kAhead lines ahead into register sets addressed round robin (no
// moves: a move of a register that a load is still writing waits for the load) with unconditional, clamped loads (row pointer (scalar) + 32-bit lane offset),
// so that the compiler can count: the wait in front of a set's first use leaves the younger sets' loads in flight. Stores carry no lane masks or branches.
// C: columns per thread (2: lines up to 2048 samples, 3: up to 2560 — the second pass of a 2x upscale of 1080p is 2160 wide —, 4: up to 4096).
// SC: an sclip is present (its row rides in the prefetch sets; without one the sets are a quarter smaller).
template <bool HP, int VC, int C, bool SC>
__global__ __launch_bounds__(kVcNT) void eedi3_vcheck_lds_kernel(const VParams prm) {
    static_assert(kVcNT == 1024 && C >= 1 && C <= 4, "rows of C * 1024 samples");
    extern __shared__ __attribute__((aligned(16))) unsigned char vsm[];
    const int pslot = prm.plane_base + (int)blockIdx.x;
    const EPlane pl = prm.p[pslot];
    const float *scp = SC ? prm.scp[pslot] : nullptr;  // (a plane without one in a launch that has some: its rows are loaded from the plane and not used)
    const float *scsrc = scp ? scp : pl.dst;
    const int scstride = scp ? prm.scstride[pslot] : pl.dstride;
    const int L = pl.w, n_dst = pl.n_dst;
    const int tid = threadIdx.x;
    constexpr int kPitch = C == 1 ? 1024 : C == 2 ? 2048 : C == 3 ? 2560 : 4096;  // fixed: slot offsets are compile-time constants
    constexpr int kOdd = 3, kEven = 3, kBlend = 2, kMapRing = 6, kAhead = 3, kPeriod = 6;
    float *odd = reinterpret_cast<float *>(vsm);           // odd rows (kept field lines): row pd - 1 + 2 k of line q in slot (q + k) % 3
    float *even = odd + (size_t)kOdd * kPitch;             // interpolated, un-blended: row pd + 2 k in slot (q + k) % 3
    float *blend = even + (size_t)kEven * kPitch;          // blended: line q reads slot (q + 1) & 1 (row pd - 2), writes slot q & 1
    int8_t *dring = reinterpret_cast<int8_t *>(blend + (size_t)kBlend * kPitch);  // map row off + d in slot (q + d) % 6
    int first = 1, last = pl.n_interp - 2;  // (:921-925)
    while (first <= last && prm.field + 2 * first < 2) ++first;
    while (last >= first && prm.field + 2 * last + 2 >= n_dst) --last;
    if (first > last) return;
    const int pd0 = prm.field + 2 * first;
    uint32_t icl[C];  // the thread's columns, clamped into the line, as byte offsets
#pragma unroll
    for (int c = 0; c < C; ++c) icl[c] = 4u * (uint32_t)min(tid + c * kVcNT, L - 1);
    auto at = [](const void *rowp, uint32_t byte_off) { return *reinterpret_cast<const uint32_t *>(static_cast<const char *>(rowp) + byte_off); };
    auto rowp = [&](int r) { return pl.dst + (size_t)min(max(r, 0), n_dst - 1) * pl.dstride; };
    {   // rows pd0 - 1 .. pd0 + 2 and the (already final) line pd0 - 2; map rows first - 1 .. first + 1
        const float *g[5] = {rowp(pd0 - 1), rowp(pd0 + 1), rowp(pd0), rowp(pd0 + 2), rowp(pd0 - 2)};
        float *d[5] = {odd, odd + kPitch, even, even + kPitch, blend + kPitch};
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) d[k][icl[c] >> 2] = __uint_as_float(at(g[k], icl[c]));
#pragma unroll
        for (int k = -1; k <= 1; ++k) {
            const int *gm = pl.dmap + (size_t)(first + k) * L;
            int8_t *dm = dring + (size_t)((k + kMapRing) % kMapRing) * kPitch;
#pragma unroll
            for (int c = 0; c < C; ++c) dm[icl[c] >> 2] = (int8_t)at(gm, icl[c]);
        }
    }
    // own-column copies of the odd rows behind the window: hist[(q + 1) % 3] is row pd - 3 of line q (the reflected row 1 for row -1)
    float hist[3][C];
    {
        const float *h1 = rowp(pd0 - 3 < 0 ? pd0 - 1 : pd0 - 3), *h2 = rowp(pd0 - 1), *h0 = rowp(pd0 + 1);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            hist[1][c] = __uint_as_float(at(h1, icl[c]));
            hist[2][c] = __uint_as_float(at(h2, icl[c]));
            hist[0][c] = __uint_as_float(at(h0, icl[c]));
        }
    }
    __syncthreads();
    // prefetch set of line o (stored by line o - 1): rows pdo + 1 (odd) and pdo + 2 (even), map row o + 1, the sclip row pdo. A row past the plane is the row the
    // reference reflects onto when it is read as pd + 3 (n_dst -> n_dst - 2) and anything valid otherwise (never used).
    float so[kAhead][C], se[kAhead][C], ssc[kAhead][SC ? C : 1];
    int sd[kAhead][C];
    auto fetch = [&](int o, float *qo, float *qe, int *qd, float *qsc) __attribute__((always_inline)) {
        const int pdo = prm.field + 2 * o;
        const int ro = pdo + 1 < n_dst ? pdo + 1 : 2 * (n_dst - 1) - (pdo + 1);
        const float *po = rowp(ro), *pe = rowp(pdo + 2);
        const int *pm = pl.dmap + (size_t)min(o + 1, pl.n_interp - 1) * L;
        const float *ps = SC ? scsrc + (size_t)min(pdo, n_dst - 1) * scstride : nullptr;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            asm("" : "+v"(icl[c]));  // (keeps the offset's zero-extension in this block: saddr + 32-bit voffset loads)
            qo[c] = __uint_as_float(at(po, icl[c]));
            qe[c] = __uint_as_float(at(pe, icl[c]));
            qd[c] = (int)at(pm, icl[c]);
            if constexpr (SC) qsc[c] = __uint_as_float(at(ps, icl[c]));
        }
    };
    float sc[SC ? C : 1];
    if constexpr (SC) {
#pragma unroll
        for (int c = 0; c < C; ++c) sc[c] = __uint_as_float(at(scsrc + (size_t)pd0 * scstride, icl[c]));
    }
#pragma unroll
    for (int j = 1; j < kAhead; ++j) fetch(first + j, so[j], se[j], sd[j], ssc[j]);
    for (int off0 = first; off0 <= last; off0 += kPeriod) {
        static_for<0, kPeriod - 1, 1>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;  // q mod 6, q = off - first
            const int off = off0 + j;
            if (off > last) return;
            constexpr int js = j % kAhead, jn = (j + 1) % kAhead;  // the sets of lines off + kAhead (loaded now) and off + 1 (stored at the end; its odd row is pd + 3)
            fetch(off + kAhead, so[js], se[js], sd[js], ssc[js]);
            const float *d1p = odd + (size_t)(j % kOdd) * kPitch, *d1n = odd + (size_t)((j + 1) % kOdd) * kPitch;
            const float *dl = even + (size_t)(j % kEven) * kPitch, *d2n = even + (size_t)((j + 1) % kEven) * kPitch;
            const float *d2p = blend + (size_t)((j + 1) & 1) * kPitch;
            auto dmr = [&](int d) -> int8_t * { return dring + (size_t)((j + d + kMapRing) % kMapRing) * kPitch; };
            const int8_t *dc = dmr(0), *dp = dmr(-1), *dn = dmr(1);
            float res[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (int)(icl[c] >> 2);
                const float cubic = 0.5625f * (d1p[i] + d1n[i]) - 0.0625f * (hist[(j + 1) % 3][c] + so[jn][c]);
                const float cint = (SC && scp) ? sc[SC ? c : 0] : cubic;
                res[c] = vcheck_pixel_bf<HP, VC>(prm, L, i, dc[i], dp[i], dn[i], cint, d1p, d1n, dl, d2p, d2n);
            }
            // (a thread past the line repeats the last column's stores; the rows of a line past the chain go to their slots and are never read)
            float *gout = pl.dst + (size_t)(prm.field + 2 * off) * pl.dstride;
            float *blw = blend + (size_t)(j & 1) * kPitch, *ow = odd + (size_t)((j + 2) % kOdd) * kPitch, *ew = even + (size_t)((j + 2) % kEven) * kPitch;
            int8_t *dnew = dmr(2);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const uint32_t i = icl[c] >> 2;
                blw[i] = res[c];
                gout[i] = res[c];
                ow[i] = so[jn][c];
                ew[i] = se[jn][c];
                dnew[i] = (int8_t)sd[jn][c];
                hist[(j + 1) % 3][c] = so[jn][c];
                if constexpr (SC) sc[c] = ssc[jn][c];
            }
            __syncthreads();  // the line's writes against the next line's reads
        });
    }
}

constexpr int kTT = 32;
__global__ void transpose_kernel(const float *src, float *dst, int sstride, int dstride, int w, int h) {
    __shared__ float t[kTT][kTT + 1];
    const int x0 = blockIdx.x * kTT, y0 = blockIdx.y * kTT;
    for (int r = threadIdx.y; r < kTT; r += blockDim.y) {
        const int x = x0 + threadIdx.x, y = y0 + r;
        if (x < w && y < h) t[r][threadIdx.x] = src[(size_t)y * sstride + x];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < kTT; r += blockDim.y) {
        const int y = y0 + threadIdx.x, x = x0 + r;  // dst[x][y] = src[y][x]
        if (x < w && y < h) dst[(size_t)x * dstride + y] = t[threadIdx.x][r];
    }
}

__global__ void transpose_u8_kernel(const uint8_t *src, uint8_t *dst, int sstride, int dstride, int w, int h) {
    __shared__ uint8_t t[kTT][kTT + 4];
    const int x0 = blockIdx.x * kTT, y0 = blockIdx.y * kTT;
    for (int r = threadIdx.y; r < kTT; r += blockDim.y) {
        const int x = x0 + threadIdx.x, y = y0 + r;
        if (x < w && y < h) t[r][threadIdx.x] = src[(size_t)y * sstride + x];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < kTT; r += blockDim.y) {
        const int y = y0 + threadIdx.x, x = x0 + r;
        if (x < w && y < h) dst[(size_t)x * dstride + y] = t[threadIdx.x][r];
    }
}

template <int NRAD>
void launch_general(vszip_ctx *ctx, bool hp, unsigned lines, const EParams &ep, const GExtra &gx) {
    if (hp && 4 * ep.mdis + 1 > 81)
        hipLaunchKernelGGL((eedi3_line_general_kernel<NRAD, true, 4 * kGMaxMdis + 1>), dim3(lines), dim3(64), 0, ctx->stream, ep, gx);
    else if (hp)
        hipLaunchKernelGGL((eedi3_line_general_kernel<NRAD, true, 81>), dim3(lines), dim3(64), 0, ctx->stream, ep, gx);
    else
        hipLaunchKernelGGL((eedi3_line_general_kernel<NRAD, false, 81>), dim3(lines), dim3(64), 0, ctx->stream, ep, gx);
}

}  // namespace

VSZIP_EXPORT int vszip_eedi3(vszip_ctx *ctx, const vszip_plane *planes, const float *const *sclips, const ptrdiff_t *sclip_strides, int nplanes, int field,
                             int horizontal, const vszip_eedi3_params *up) {
    return vszip_eedi3_mclip(ctx, planes, sclips, sclip_strides, nullptr, nullptr, nplanes, field, horizontal, up);
}

static int eedi3_batch(vszip_ctx *ctx, const vszip_plane *planes, const float *const *sclips, const ptrdiff_t *sclip_strides, const uint8_t *const *mclips,
                       const ptrdiff_t *mclip_strides, int nplanes, int field, int horizontal, const vszip_eedi3_params *up);

// Any number of planes per call: batches of kMaxPlanesE (the per-plane tables travel in the kernel arguments).
VSZIP_EXPORT int vszip_eedi3_mclip(vszip_ctx *ctx, const vszip_plane *planes, const float *const *sclips, const ptrdiff_t *sclip_strides,
                                   const uint8_t *const *mclips, const ptrdiff_t *mclip_strides, int nplanes, int field, int horizontal,
                                   const vszip_eedi3_params *up) {
    if (!ctx || !planes || !up || nplanes <= 0) return VSZIP_ERR_ARG;
    if (!(up->gamma >= 0.0f)) return vszip_set_error(ctx, VSZIP_ERR_ARG, "EEDI3: gamma must be greater than or equal to 0.0.");  // eedi3.zig:368 (the DP kernels rely on it)
    for (int o = 0; o < nplanes; o += kMaxPlanesE) {
        const int rc = eedi3_batch(ctx, planes + o, sclips ? sclips + o : nullptr, sclip_strides ? sclip_strides + o : nullptr, mclips ? mclips + o : nullptr,
                                   mclip_strides ? mclip_strides + o : nullptr, std::min(kMaxPlanesE, nplanes - o), field, horizontal, up);
        if (rc != VSZIP_OK) return rc;
    }
    return VSZIP_OK;
}

static int eedi3_batch(vszip_ctx *ctx, const vszip_plane *planes, const float *const *sclips, const ptrdiff_t *sclip_strides, const uint8_t *const *mclips,
                       const ptrdiff_t *mclip_strides, int nplanes, int field, int horizontal, const vszip_eedi3_params *up) {
    if (!ctx || !planes || !up || nplanes <= 0 || nplanes > kMaxPlanesE) return VSZIP_ERR_ARG;
    const char *name = horizontal ? "EEDI3H" : "EEDI3";
    // createImpl :316-410
    if (field < 0 || field > 1) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: field must be 0 or 1 here (the wrapper resolves 2/3 per frame).", name);
    if (up->alpha < 0.0f || up->alpha > 1.0f) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: alpha must be between 0.0 and 1.0 (inclusive).", name);
    if (up->beta < 0.0f || up->beta > 1.0f) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: beta must be between 0.0 and 1.0 (inclusive).", name);
    if (up->alpha + up->beta > 1.0f) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: alpha + beta must be less than or equal to 1.0.", name);
    if (up->gamma < 0.0f) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: gamma must be greater than or equal to 0.0.", name);
    if (up->nrad < 0 || up->nrad > 3) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: nrad must be between 0 and 3 (inclusive).", name);
    if (up->mdis < 1 || up->mdis > 40) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: mdis must be between 1 and 40 (inclusive).", name);
    if (up->vcheck < 0 || up->vcheck > 3) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: vcheck must be 0, 1, 2, or 3.", name);
    if (up->vcheck > 0 && (up->vthresh0 <= 0.0f || up->vthresh1 <= 0.0f || up->vthresh2 <= 0.0f))
        return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: vthresh0, vthresh1 and vthresh2 must be greater than 0.0.", name);
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const bool dh = up->dh != 0;
    const bool hp = up->hp != 0;
    bool any_mask = false;
    for (int i = 0; i < nplanes && mclips; ++i) any_mask = any_mask || mclips[i] != nullptr;
    if (any_mask && !mclip_strides) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: mclip strides missing", name);
    // the tuned kernel covers the common case and, on its fixed layout (mdis <= 20), mclip; hp, mdis > 31 and mclip with 20 < mdis <= 31 take the general one
    const bool general = hp || up->mdis > kMaxMdis || (any_mask && (up->mdis > 20 || ctx->opt.eedi3_no_fixed));
    // (mdis <= 20 on the tuned kernel: the layout of mdis = 20, see MASK)
    const bool masked20 = !general && up->mdis < 20 && !ctx->opt.eedi3_no_fixed;
    const int tpitch = hp ? 4 * up->mdis + 1 : (masked20 ? 2 * 20 + 1 : 2 * up->mdis + 1);

    // geometry of the vertical pipeline per plane (EEDI3H runs it on the transposed plane)
    struct Geo {
        int L, n_src, n_dst, n_interp;
        size_t srcT, dstT, scT;  // float offsets into scratch (horizontal only)
        size_t mT;               // byte offset of the transposed mask (horizontal only)
    };
    std::vector<Geo> geo(nplanes);
    size_t fl = 0, lines = 0, pb = 0, dm = 0, mb = 0;
    for (int i = 0; i < nplanes; ++i) {
        const vszip_plane &s = planes[i];
        if (!s.src || !s.dst || s.w <= 0 || s.h <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: bad plane %d", name, i);
        const int axis = horizontal ? s.w : s.h;
        if (!dh && (axis & 1)) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s: %s must be mod 2 when dh=False.", name, horizontal ? "width" : "height");
        Geo &g = geo[i];
        g.L = horizontal ? s.h : s.w;
        g.n_src = horizontal ? s.w : s.h;
        g.n_dst = dh ? 2 * g.n_src : g.n_src;
        g.n_interp = dh ? g.n_src : g.n_src / 2;
        // The reference has no such check, and no defined result either: mirrorPad (src/filters/eedi3.zig:107-116) fills columns past one
        // reflection from what the rotating scratch row held before (uninitialised memory on a frame's first lines), and below nrad+2
        // samples the window sums that decide the path read them.
        if (g.L < up->nrad + 2)
            return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "%s: lines of %d samples are shorter than nrad+2 = %d; the reference's padded rows are undefined there", name, g.L,
                                   up->nrad + 2);
        if (horizontal) {
            g.srcT = fl;
            fl += (size_t)g.n_src * g.L;
            g.dstT = fl;
            fl += (size_t)g.n_dst * g.L;
            g.scT = fl;
            if (up->vcheck > 0 && sclips && sclips[i]) fl += (size_t)g.n_dst * g.L;
            g.mT = mb;
            if (mclips && mclips[i]) mb += (((size_t)g.n_src * g.L) + 255) & ~(size_t)255;
        }
        lines += g.n_interp;
        pb += (size_t)g.n_interp * ((g.L + kXB - 1) / kXB * kXB) * tpitch;  // the line kernel stores whole 64-column blocks
        dm += (size_t)g.n_interp * g.L;
    }
    size_t bytes = (fl * sizeof(float) + dm * sizeof(int) + pb + mb + 4096 + 255) & ~(size_t)255;
    const size_t gline_off = bytes;  // vcheck on lines wider than 8192: two lines per plane
    {
        int maxL = 0;
        for (int i = 0; i < nplanes; ++i) maxL = std::max(maxL, geo[i].L);
        if (up->vcheck > 0 && maxL > 8192) bytes += (size_t)nplanes * 2 * ((maxL + 63) & ~63) * sizeof(float);
    }
    int rc = vszip_ensure_scratch(ctx, bytes);
    if (rc != VSZIP_OK) return rc;
    char *base = static_cast<char *>(ctx->scratch);
    float *fbase = reinterpret_cast<float *>(base);
    int *dbase = reinterpret_cast<int *>(base + fl * sizeof(float));
    // the back-pointer area is written with 16-byte stores: keep it 256-byte aligned (the +4096 of `bytes` covers the padding)
    const size_t pb_off = (fl * sizeof(float) + dm * sizeof(int) + 255) & ~(size_t)255;
    int8_t *pbase = reinterpret_cast<int8_t *>(base + pb_off);
    uint8_t *mbase = reinterpret_cast<uint8_t *>(base + pb_off + pb);
    GExtra gx;
    gx.hp = hp;

    EParams ep;
    VParams vp;
    ep.nplanes = nplanes;
    ep.field = field;
    ep.dh = dh;
    ep.mdis = up->mdis;
    ep.nrad = up->nrad;
    // src/vapoursynth/eedi3.zig:465-473
    ep.one_minus_ab = 1.0f - up->alpha - up->beta;
    ep.alpha = up->alpha / 3.0f;
    ep.beta = up->beta / 255.0f;
    ep.gamma = up->gamma / 255.0f;
    vp.field = field;
    vp.dh = dh;
    vp.vcheck = up->vcheck;
    vp.hp = hp;
    const float vt0 = up->vthresh0 / 255.0f, vt1 = up->vthresh1 / 255.0f;
    vp.vthresh2 = up->vthresh2;
    vp.rcp0 = 1.0f / vt0;
    vp.rcp1 = 1.0f / vt1;
    vp.rcp2 = 1.0f / up->vthresh2;
    size_t dmo = 0, pbo = 0;
    int line0 = 0, maxw = 0, maxsrc = 0;
    const dim3 tb(kTT, 8);
    // Plane slots: the tall planes of the call first (the luma planes of a subsampled clip). The vertical-consistency
    // pass is a sequential chain per plane, as long as the plane is tall and busy on one CU per plane only; with the
    // tall planes' lines interpolated first their chains run on a second stream BESIDE the line kernel of the short
    // planes, and only the short planes' (half as long) chains remain after it.
    std::vector<int> order(nplanes);
    int ntall = 0, lines_tall = 0;
    {
        int max_interp = 0;
        for (int i = 0; i < nplanes; ++i) max_interp = std::max(max_interp, geo[i].n_interp);
        for (int i = 0; i < nplanes; ++i)
            if (4 * geo[i].n_interp >= 3 * max_interp) order[ntall++] = i;
        int k = ntall;
        for (int i = 0; i < nplanes; ++i)
            if (4 * geo[i].n_interp < 3 * max_interp) order[k++] = i;
    }
    for (int slot = 0; slot < nplanes; ++slot) {
        const int i = order[slot];
        const vszip_plane &s = planes[i];
        const Geo &g = geo[i];
        EPlane &d = ep.p[slot];
        if (slot == ntall) lines_tall = line0;
        if (horizontal) {
            float *srcT = fbase + g.srcT;
            hipLaunchKernelGGL(transpose_kernel, dim3((s.w + kTT - 1) / kTT, (s.h + kTT - 1) / kTT), tb, 0, ctx->stream, static_cast<const float *>(s.src), srcT,
                               (int)s.src_stride, g.L, s.w, s.h);
            d.src = srcT;
            d.dst = fbase + g.dstT;
            d.sstride = d.dstride = g.L;
            vp.scp[slot] = nullptr;
            vp.scstride[slot] = g.L;
            if (up->vcheck > 0 && sclips && sclips[i]) {
                float *scT = fbase + g.scT;
                hipLaunchKernelGGL(transpose_kernel, dim3((g.n_dst + kTT - 1) / kTT, (s.h + kTT - 1) / kTT), tb, 0, ctx->stream, sclips[i], scT,
                                   (int)sclip_strides[i], g.L, g.n_dst, s.h);
                vp.scp[slot] = scT;
            }
            gx.mask[slot] = nullptr;
            gx.mstride[slot] = g.L;
            if (mclips && mclips[i]) {  // mask has the source plane's geometry (w x h) -> transposed: w lines of h
                uint8_t *mT = mbase + g.mT;
                hipLaunchKernelGGL(transpose_u8_kernel, dim3((s.w + kTT - 1) / kTT, (s.h + kTT - 1) / kTT), tb, 0, ctx->stream, mclips[i], mT, (int)mclip_strides[i], g.L,
                                   s.w, s.h);
                gx.mask[slot] = mT;
            }
        } else {
            gx.mask[slot] = mclips ? mclips[i] : nullptr;
            gx.mstride[slot] = (mclips && mclips[i]) ? (int)mclip_strides[i] : 0;
            d.src = static_cast<const float *>(s.src);
            d.dst = static_cast<float *>(s.dst);
            d.sstride = (int)s.src_stride;
            d.dstride = (int)s.dst_stride;
            vp.scp[slot] = (up->vcheck > 0 && sclips) ? sclips[i] : nullptr;
            vp.scstride[slot] = (sclips && sclip_strides) ? (int)sclip_strides[i] : 0;
        }
        d.w = g.L;
        d.n_src = g.n_src;
        d.n_dst = g.n_dst;
        d.n_interp = g.n_interp;
        d.dmap = dbase + dmo;
        d.pback = pbase + pbo;
        d.line0 = line0;
        dmo += (size_t)g.n_interp * g.L;
        pbo += (size_t)g.n_interp * ((g.L + kXB - 1) / kXB * kXB) * tpitch;
        line0 += g.n_interp;
        maxw = std::max(maxw, g.L);
        maxsrc = std::max(maxsrc, g.n_src);
        vp.p[slot] = d;
        ep.mask[slot] = gx.mask[slot];
        ep.mstride[slot] = gx.mstride[slot];
    }
    if (ntall == nplanes) lines_tall = line0;
    int maxL = 0;
    for (int i = 0; i < nplanes; ++i) maxL = std::max(maxL, geo[i].L);
    vp.gline = nullptr;
    vp.gline_pitch = (maxL + 63) & ~63;
    if (maxL > 8192) vp.gline = reinterpret_cast<float *>(base + gline_off);
    const bool vc_lds = up->vcheck > 0 && maxL <= kVcLdsMaxL && !ctx->opt.vcheck_global;
    auto launch_vcheck_lds = [&](hipStream_t st, int first, int count) {
        int gl = 0;  // the widest line of THIS launch (the short planes of a 4K 4:2:0 clip are 1920 wide: two columns a thread, not the luma's four)
        for (int i = first; i < first + count; ++i) gl = std::max(gl, vp.p[i].w);
#ifdef VSZIP_VC_COLS_BY_CALL  // (sweeps)
        gl = maxL;
#endif
        const int cols = gl <= 1024 ? 1 : gl <= 2048 ? 2 : gl <= 2560 ? 3 : 4;  // (round 6: one column a thread for lines up to 1024 samples - the chroma of a 1080p 4:2:0 frame; with two, half the workgroup idled through every line)
        bool any_sc = false;
        for (int i = first; i < first + count; ++i) any_sc = any_sc || vp.scp[i] != nullptr;
        vp.plane_base = first;
        const size_t lds = (size_t)(cols == 1 ? 1024 : cols == 2 ? 2048 : cols == 3 ? 2560 : 4096) * (8 * sizeof(float) + 6);  // three odd, three even, two blended rows, six int8 map rows
#define VSZIP_VC_LAUNCH3(HPV, VCV, CV, SCV)                                                                                                          \
    do {                                                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(eedi3_vcheck_lds_kernel<HPV, VCV, CV, SCV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((eedi3_vcheck_lds_kernel<HPV, VCV, CV, SCV>), dim3(count), dim3(kVcNT), lds, st, vp);                                    \
    } while (0)
#define VSZIP_VC_LAUNCH2(HPV, VCV, CV)        \
    do {                                      \
        if (any_sc)                           \
            VSZIP_VC_LAUNCH3(HPV, VCV, CV, true);  \
        else                                  \
            VSZIP_VC_LAUNCH3(HPV, VCV, CV, false); \
    } while (0)
#define VSZIP_VC_LAUNCH1(HPV, VCV)            \
    do {                                      \
        if (cols == 1)                        \
            VSZIP_VC_LAUNCH2(HPV, VCV, 1);    \
        else if (cols == 2)                   \
            VSZIP_VC_LAUNCH2(HPV, VCV, 2);    \
        else if (cols == 3)                   \
            VSZIP_VC_LAUNCH2(HPV, VCV, 3);    \
        else                                  \
            VSZIP_VC_LAUNCH2(HPV, VCV, 4);    \
    } while (0)
#define VSZIP_VC_LAUNCH(HPV)                  \
    do {                                      \
        if (up->vcheck == 1)                  \
            VSZIP_VC_LAUNCH1(HPV, 1);         \
        else if (up->vcheck == 2)             \
            VSZIP_VC_LAUNCH1(HPV, 2);         \
        else                                  \
            VSZIP_VC_LAUNCH1(HPV, 3);         \
    } while (0)
        if (hp)
            VSZIP_VC_LAUNCH(true);
        else
            VSZIP_VC_LAUNCH(false);
#undef VSZIP_VC_LAUNCH3
#undef VSZIP_VC_LAUNCH2
#undef VSZIP_VC_LAUNCH1
#undef VSZIP_VC_LAUNCH
    };
    // two plane heights in the call, the tuned line kernel and the LDS chain kernel: overlap (see `order` above).
    // The chains need a whole CU's worth of LDS each; beside a full line-kernel launch they would never find one
    // (freed wave slots go to the next line wave), so the short planes' line kernel runs on a stream whose CU mask
    // leaves one CU per chain free.
    // Only for calls of 12 frames and more (tools/eedi3_overlap_ab.py, 1080p YUV420PS, frames per call: with / without — 1: 306 / 401 fps,
    // 2: 576 / 706, 4: 993 / 1119, 8: 1607 / 1630, 16: 2308 / 2170): below that the masked stream's launches and events cost more than the
    // overlap returns, and the plugin's one-frame calls from twelve contexts at once lost half their rate to it (757 against 1599 fps).
    // (VSZIP_EEDI3_FORCE_OVERLAP=1: from one tall plane on — the parity tests run small batches through this path)
    const int min_tall = ctx->opt.eedi3_force_overlap ? 1 : 12;
    bool split = !general && vc_lds && ntall >= min_tall && ntall < nplanes && ntall <= 64 && !ctx->opt.eedi3_no_overlap;
    if (split) {
        const int reserve = std::min(64, (ntall + 7) & ~7), cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
        if (ctx->aux_stream && ctx->aux_reserved != reserve) {
            (void)hipStreamSynchronize(ctx->aux_stream);
            (void)hipStreamDestroy(ctx->aux_stream);
            ctx->aux_stream = nullptr;
        }
        if (!ctx->aux_stream) {
            std::vector<uint32_t> mask((cus + 31) / 32, 0xffffffffu);
            for (int b = 0; b < reserve; ++b) mask[b / 32] &= ~(1u << (b % 32));  // the runtime spreads consecutive bits over the XCDs
            if (cus % 32) mask.back() &= (1u << (cus % 32)) - 1;
            bool ok = hipExtStreamCreateWithCUMask(&ctx->aux_stream, (uint32_t)mask.size(), mask.data()) == hipSuccess;
            if (ok && !ctx->aux_fork) ok = hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);  // the stream exists when only the events failed (it leaked, and the create was retried every call)
                ctx->aux_stream = nullptr;
                split = false;
            }
            ctx->aux_reserved = reserve;
            if (split) vszip_aux_register(ctx);
        }
    }
    if (!split) {
        ntall = nplanes;  // one group: the launches below cover every line / plane at once
        lines_tall = (int)lines;
    }
    ep.line_base = 0;
    vp.plane_base = 0;
    ep.copy_kept = general ? 0 : 1;  // (the tuned line kernel writes the kept lines itself)
    if (general) hipLaunchKernelGGL(eedi3_copy_kernel, dim3((maxw / 4 + 255) / 256, maxsrc, nplanes), dim3(256), 0, ctx->stream, ep);  // (grid-stride rows: any x grid serves)
    if (general) {
        switch (up->nrad) {
            case 0: launch_general<0>(ctx, hp, (unsigned)lines, ep, gx); break;
            case 1: launch_general<1>(ctx, hp, (unsigned)lines, ep, gx); break;
            case 2: launch_general<2>(ctx, hp, (unsigned)lines, ep, gx); break;
            default: launch_general<3>(ctx, hp, (unsigned)lines, ep, gx); break;
        }
    } else {
        const dim3 lblock(64);
#define VSZIP_E3_LAUNCH(N)                                                                        \
    do {                                                                                          \
        if (any_mask && up->mdis == 20)                                                            \
            hipLaunchKernelGGL((eedi3_line_kernel<N, 20, true, false, true>), lgrid, lblock, 0, lst, ep); \
        else if (any_mask)                                                                        \
            hipLaunchKernelGGL((eedi3_line_kernel<N, 20, true, true, true>), lgrid, lblock, 0, lst, ep); \
        else if (up->mdis == 20 && !ctx->opt.eedi3_no_fixed)                                    \
            hipLaunchKernelGGL((eedi3_line_kernel<N, 20, true>), lgrid, lblock, 0, lst, ep);      \
        else if (masked20)                                                                        \
            hipLaunchKernelGGL((eedi3_line_kernel<N, 20, true, true>), lgrid, lblock, 0, lst, ep); \
        else if (up->mdis <= 20)                                                                  \
            hipLaunchKernelGGL((eedi3_line_kernel<N, 20, false>), lgrid, lblock, 0, lst, ep);     \
        else                                                                                      \
            hipLaunchKernelGGL((eedi3_line_kernel<N, kMaxMdis, false>), lgrid, lblock, 0, lst, ep); \
    } while (0)
        auto launch_lines = [&](hipStream_t lst, int first, int count) {
            const dim3 lgrid((unsigned)count);
            ep.line_base = first;
            switch (up->nrad) {
                case 0: VSZIP_E3_LAUNCH(0); break;
                case 1: VSZIP_E3_LAUNCH(1); break;
                case 2: VSZIP_E3_LAUNCH(2); break;
                default: VSZIP_E3_LAUNCH(3); break;
            }
        };
        launch_lines(ctx->stream, 0, lines_tall);
        if (split) {
            // the tall planes' lines are done: the short planes' lines go to the masked stream, the tall planes' chains
            // start here at once on the CUs it leaves free
            // Any failure between the fork and the join must not leave the masked stream running kernels on the context's scratch
            // that the main stream no longer waits for (the next call may grow or free it): drain it before returning.
            auto forked = [&](hipError_t e) -> bool {
                if (e == hipSuccess) return true;
                (void)hipStreamSynchronize(ctx->aux_stream);
                return false;
            };
            VSZIP_HIP_CHECK(ctx, hipEventRecord(ctx->aux_fork, ctx->stream));
            VSZIP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork, 0));
            launch_lines(ctx->aux_stream, lines_tall, (int)lines - lines_tall);
            if (!forked(hipEventRecord(ctx->aux_join, ctx->aux_stream))) return vszip_set_error(ctx, VSZIP_ERR_HIP, "%s: joining the second stream failed", name);
            launch_vcheck_lds(ctx->stream, 0, ntall);
            if (!forked(hipStreamWaitEvent(ctx->stream, ctx->aux_join, 0))) return vszip_set_error(ctx, VSZIP_ERR_HIP, "%s: joining the second stream failed", name);
        } else if ((int)lines > lines_tall) {
            launch_lines(ctx->stream, lines_tall, (int)lines - lines_tall);
        }
#undef VSZIP_E3_LAUNCH
    }
    VSZIP_HIP_CHECK(ctx, hipGetLastError());
    if (up->vcheck > 0) {
        if (vc_lds) {
            if (split) {
                launch_vcheck_lds(ctx->stream, ntall, nplanes - ntall);  // (the tall planes' chains were launched beside the line kernel)
            } else {
                launch_vcheck_lds(ctx->stream, 0, nplanes);
            }
        } else if (maxL <= 4096) {
            hipLaunchKernelGGL(eedi3_vcheck_kernel, dim3(nplanes), dim3(1024), 0, ctx->stream, vp);
        } else {
            hipLaunchKernelGGL(eedi3_vcheck_wide_kernel, dim3(nplanes), dim3(1024), vp.gline ? 0 : (size_t)maxL * 2 * sizeof(float), ctx->stream, vp);
        }
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
    }
    if (horizontal) {
        for (int i = 0; i < nplanes; ++i) {
            const vszip_plane &s = planes[i];
            const Geo &g = geo[i];
            // dstT is n_dst lines of L -> dst is L (= src_h) rows of n_dst columns
            hipLaunchKernelGGL(transpose_kernel, dim3((g.L + kTT - 1) / kTT, (g.n_dst + kTT - 1) / kTT), tb, 0, ctx->stream, fbase + g.dstT, static_cast<float *>(s.dst), g.L,
                               (int)s.dst_stride, g.L, g.n_dst);
        }
        VSZIP_HIP_CHECK(ctx, hipGetLastError());
    }
    return VSZIP_OK;
}
