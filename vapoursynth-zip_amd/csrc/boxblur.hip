// vszip.BoxBlur entry point (C ABI): validation + path choice, mirroring
// src/vapoursynth/boxblur.zig:131-211. The CT integer kernels live in
// boxblur_ct.hpp and are instantiated by boxblur_ct_{u8,u16}_{a,b,c}.hip.
#include "common.hpp"

int vszip_bb_ct_u8_a(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u8_b(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u8_c(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u16_a(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u16_b(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u16_c(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u8_a_h(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u8_b_h(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u8_c_h(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u16_a_h(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u16_b_h(vszip_ctx *, int, const vszip_plane *, int);
int vszip_bb_ct_u16_c_h(vszip_ctx *, int, const vszip_plane *, int);
// boxblur_rt.hip
int vszip_bb_rt(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, int hradius, int hpasses, int vradius, int vpasses);
int vszip_bb_ct_float(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, int radius);

static int ct_int(vszip_ctx *ctx, int dtype, int r, const vszip_plane *planes, int nplanes) {
    if (dtype == VSZIP_U8) return r <= 8 ? vszip_bb_ct_u8_a(ctx, r, planes, nplanes) : r <= 15 ? vszip_bb_ct_u8_b(ctx, r, planes, nplanes) : vszip_bb_ct_u8_c(ctx, r, planes, nplanes);
    return r <= 8 ? vszip_bb_ct_u16_a(ctx, r, planes, nplanes) : r <= 15 ? vszip_bb_ct_u16_b(ctx, r, planes, nplanes) : vszip_bb_ct_u16_c(ctx, r, planes, nplanes);
}

// the horizontal pass alone through the compile-time-radius ring kernel (boxblur_ct.hpp, CtIntDispatch::run_h); VSZIP_ERR_UNSUPPORTED: not for these planes
static int ct_int_h(vszip_ctx *ctx, int dtype, int r, const vszip_plane *planes, int nplanes) {
    if (dtype == VSZIP_U8) return r <= 8 ? vszip_bb_ct_u8_a_h(ctx, r, planes, nplanes) : r <= 15 ? vszip_bb_ct_u8_b_h(ctx, r, planes, nplanes) : vszip_bb_ct_u8_c_h(ctx, r, planes, nplanes);
    return r <= 8 ? vszip_bb_ct_u16_a_h(ctx, r, planes, nplanes) : r <= 15 ? vszip_bb_ct_u16_b_h(ctx, r, planes, nplanes) : vszip_bb_ct_u16_c_h(ctx, r, planes, nplanes);
}

int vszip_bb_ct_row_pass(vszip_ctx *ctx, int dtype, int r, const vszip_plane *planes, int nplanes) { return ct_int_h(ctx, dtype, r, planes, nplanes); }

VSZIP_EXPORT int vszip_boxblur(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, int hradius, int hpasses, int vradius, int vpasses) {
    if (!ctx || !planes || nplanes <= 0) return VSZIP_ERR_ARG;
    if (hradius < 0 || vradius < 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: negative radius");
    const bool vblur = (vradius > 0) && (vpasses > 0);
    const bool hblur = (hradius > 0) && (hpasses > 0);
    if (!vblur && !hblur) return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: nothing to be performed");  // boxblur.zig:152
    const bool use_rt = (hradius != vradius) || (hradius > 22) || (hpasses > 1) || (vpasses > 1);         // boxblur.zig:188
    for (int i = 0; i < nplanes; ++i) {
        const vszip_plane &p = planes[i];
        if (!p.src || !p.dst || p.w <= 0 || p.h <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: bad plane %d", i);
        // boxblur.zig:158-179; the CT path additionally needs both (the reference reads
        // out of bounds there when a pass count is 0 and the plane is tiny)
        if ((hblur || !use_rt) && (2L * hradius >= p.w))
            return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: hradius too large; 2*hradius must be < the (smallest processed) plane width.");
        if ((vblur || !use_rt) && (2L * vradius >= p.h))
            return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: vradius too large; 2*vradius must be < the (smallest processed) plane height.");
    }
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!use_rt) {
        switch (dtype) {
            case VSZIP_U8:
            case VSZIP_U16: return ct_int(ctx, dtype, hradius, planes, nplanes);
            case VSZIP_F16:
            case VSZIP_F32: return vszip_bb_ct_float(ctx, dtype, planes, nplanes, hradius);
            default: return vszip_set_error(ctx, VSZIP_ERR_ARG, "BoxBlur: not supported Int format.");
        }
    }
    // one horizontal pass of r <= 22 on integer planes, nothing vertical: the ring kernel with a one-row window (same arithmetic as the RT row pass)
    if (hblur && !vblur && hpasses == 1 && hradius <= 22 && (dtype == VSZIP_U8 || dtype == VSZIP_U16) && !ctx->opt.boxblur_no_ct_h) {
        const int rc = ct_int_h(ctx, dtype, hradius, planes, nplanes);
        if (rc != VSZIP_ERR_UNSUPPORTED) return rc;
    }
    return vszip_bb_rt(ctx, dtype, planes, nplanes, hradius, hpasses, vradius, vpasses);
}
