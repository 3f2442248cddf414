// TEST INFRASTRUCTURE ONLY — a GPU-less stand-in for libvszip_hip.so, built with
// -fsanitize=address,undefined together with libvszip.so and libfakevs.so (tests/sanitize/Makefile), so the
// plugin's staging, error and unwind paths run under ASan/UBSan on the CPU (the reference's CI
// runs Zig Debug = bounds/overflow checked, .github/workflows/test.yml:27).
// "Device" memory is host memory: allocations, 2-D copies and memsets are real (so a wrong pitch, width or plane
// size in the plugin's staging code is an ASan report), contexts are real objects, and every filter entry point
// walks its plane table (touching first/last byte of every plane it would read or write) and then fails with
// VSZIP_ERR_HIP — the plugin must turn that into a filter error and release everything it holds.
// VSZIP_STUB_FAIL=alloc|copy makes allocations / copies fail instead, =none makes every filter succeed. The device-free entry points come from
// the product's own csrc/host_params.cpp, compiled alongside.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vszip_hip.h"

#define EXP extern "C" __attribute__((visibility("default")))

struct vszip_ctx {
    std::string err;
    int staging = 0;
    int device = 0;
    std::vector<void *> live;  // allocations not yet freed: destroy() asserts the plugin returned them... or reports
};

static std::atomic<long> g_calls[16];  // filter calls per fake device
static bool fail_mode(const char *what) {
    const char *e = getenv("VSZIP_STUB_FAIL");
    return e && strcmp(e, what) == 0;
}
static int kernel_failed(vszip_ctx *ctx, const char *name) {
    if (ctx) g_calls[ctx->device].fetch_add(1);
    // VSZIP_STUB_FAIL=none: the "kernel" succeeds (outputs keep whatever the staging left there), so every filter's
    // SUCCESS path - download, frame properties, release of every reference - runs under the sanitizer as well
    if (fail_mode("none")) return VSZIP_OK;
    if (ctx) ctx->err = std::string(name) + ": no device (sanitizer stub)";
    return VSZIP_ERR_HIP;
}
// read the first and last sample row of a plane: ASan checks the extents the plugin handed over
static void touch(const void *p, ptrdiff_t stride_elems, int w, int h, int bps, bool write) {
    if (!p || w <= 0 || h <= 0) return;
    volatile unsigned char *b = (volatile unsigned char *)p;
    const size_t last = ((size_t)(h - 1) * (size_t)stride_elems + (size_t)(w - 1)) * (size_t)bps + (size_t)bps - 1;
    unsigned char v0 = b[0], v1 = b[last];
    if (write) {
        b[0] = v0;
        b[last] = v1;
    }
}
static int bps_of(int dtype) { return dtype == VSZIP_U8 ? 1 : (dtype == VSZIP_U16 || dtype == VSZIP_F16 ? 2 : 4); }
static void touch_table(const vszip_plane *pl, int n, int bps) {
    for (int i = 0; i < n; ++i) {
        touch(pl[i].src, pl[i].src_stride, pl[i].w, pl[i].h, bps, false);
        touch(pl[i].ref, pl[i].ref_stride, pl[i].w, pl[i].h, bps, false);
        touch(pl[i].dst, pl[i].dst_stride, pl[i].w, pl[i].h, bps, true);
    }
}

// VSZIP_STUB_DEVICES=N: N fake devices (default one). With more than one, BoxBlur's "kernel" stamps the device index
// of the context it ran on into the first sample of every destination plane and the per-device call counters below
// count filter calls, so a test can see which GPU the plugin sent each frame to (frame-index round-robin,
// vszip_plugin.cpp device_of_frame) without a multi-GPU box.
static int stub_devices() {
    const char *e = getenv("VSZIP_STUB_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n < 1 ? 1 : (n > 16 ? 16 : n);
}
EXP long vszip_stub_device_calls(int device) { return device >= 0 && device < 16 ? g_calls[device].load() : -1; }

EXP int vszip_ctx_create(int device, vszip_ctx **out) {
    if (!out) return VSZIP_ERR_ARG;
    *out = nullptr;
    if (device < 0 || device >= stub_devices()) return VSZIP_ERR_HIP;
    *out = new vszip_ctx();
    (*out)->device = device;
    return VSZIP_OK;
}
EXP void vszip_ctx_destroy(vszip_ctx *ctx) { delete ctx; }
EXP int vszip_ctx_set_stream(vszip_ctx *ctx, void *) { return ctx ? VSZIP_OK : VSZIP_ERR_ARG; }
EXP void *vszip_ctx_stream(vszip_ctx *) { return nullptr; }
EXP int vszip_ctx_sync(vszip_ctx *ctx) { return ctx ? VSZIP_OK : VSZIP_ERR_ARG; }
EXP int vszip_ctx_set_staging(vszip_ctx *ctx, int mode) {
    if (!ctx) return VSZIP_ERR_ARG;
    ctx->staging = mode;
    return VSZIP_OK;
}
EXP int vszip_ctx_abort(vszip_ctx *ctx) { return ctx ? VSZIP_OK : VSZIP_ERR_ARG; }
EXP int vszip_ctx_set_option(vszip_ctx *ctx, const char *name, int) { return ctx && name ? VSZIP_OK : VSZIP_ERR_ARG; }
EXP const char *vszip_last_error(vszip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
EXP int vszip_abi_version(void) { return VSZIP_ABI_VERSION; }

EXP int vszip_dev_alloc(vszip_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return VSZIP_ERR_ARG;
    if (fail_mode("alloc")) return VSZIP_ERR_NOMEM;
    *dptr = malloc(bytes ? bytes : 1);
    return *dptr ? VSZIP_OK : VSZIP_ERR_NOMEM;
}
EXP int vszip_dev_free(vszip_ctx *ctx, void *dptr) {
    if (!ctx) return VSZIP_ERR_ARG;
    free(dptr);
    return VSZIP_OK;
}
EXP int vszip_dev_memset(vszip_ctx *ctx, void *dptr, int value, size_t bytes) {
    if (!ctx) return VSZIP_ERR_ARG;
    memset(dptr, value, bytes);
    return VSZIP_OK;
}
EXP int vszip_host_alloc_pinned(vszip_ctx *ctx, size_t bytes, void **hptr) { return vszip_dev_alloc(ctx, bytes, hptr); }
EXP int vszip_host_free_pinned(vszip_ctx *ctx, void *hptr) { return vszip_dev_free(ctx, hptr); }
static int copy2d(vszip_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t wb, size_t rows) {
    if (!ctx) return VSZIP_ERR_ARG;
    if (fail_mode("copy")) return kernel_failed(ctx, "copy");
    for (size_t y = 0; y < rows; ++y) memcpy((char *)dst + y * dpitch, (const char *)src + y * spitch, wb);
    return VSZIP_OK;
}
EXP int vszip_copy_h2d_2d(vszip_ctx *c, void *d, size_t dp, const void *s, size_t sp, size_t wb, size_t rows) { return copy2d(c, d, dp, s, sp, wb, rows); }
EXP int vszip_copy_d2h_2d(vszip_ctx *c, void *d, size_t dp, const void *s, size_t sp, size_t wb, size_t rows) { return copy2d(c, d, dp, s, sp, wb, rows); }
EXP int vszip_copy_d2d_2d(vszip_ctx *c, void *d, size_t dp, const void *s, size_t sp, size_t wb, size_t rows) { return copy2d(c, d, dp, s, sp, wb, rows); }
EXP int vszip_timer_start(vszip_ctx *ctx) { return ctx ? VSZIP_OK : VSZIP_ERR_ARG; }
EXP int vszip_timer_stop_ms(vszip_ctx *ctx, float *ms) {
    if (ms) *ms = 0;
    return ctx ? VSZIP_OK : VSZIP_ERR_ARG;
}
EXP int vszip_probe_enable(vszip_ctx *ctx, int) { return ctx ? VSZIP_OK : VSZIP_ERR_ARG; }
EXP int vszip_probe_read_each(vszip_ctx *ctx, double *t, int *n, float *, int) {
    if (t) *t = 0;
    if (n) *n = 0;
    return ctx ? VSZIP_OK : VSZIP_ERR_ARG;
}
EXP int vszip_probe_read(vszip_ctx *ctx, double *t, int *n) { return vszip_probe_read_each(ctx, t, n, nullptr, 0); }

EXP int vszip_boxblur(vszip_ctx *ctx, int dtype, const vszip_plane *pl, int n, int, int, int, int) {
    touch_table(pl, n, bps_of(dtype));
    if (ctx && stub_devices() > 1 && fail_mode("none"))
        for (int i = 0; i < n; ++i)
            if (pl[i].dst && pl[i].w > 0 && pl[i].h > 0) memset(pl[i].dst, 0, (size_t)bps_of(dtype)), *(unsigned char *)pl[i].dst = (unsigned char)ctx->device;
    return kernel_failed(ctx, "BoxBlur");
}
EXP int vszip_plane_average(vszip_ctx *ctx, int dtype, const vszip_plane *pl, int n, const int32_t *ex, int nex, int, double *avg, double *diff) {
    touch_table(pl, n, bps_of(dtype));
    for (int i = 0; i < nex; ++i) (void)*(volatile const int32_t *)&ex[i];
    for (int i = 0; i < n; ++i) avg[i] = diff[i] = 0;
    return kernel_failed(ctx, "PlaneAverage");
}
EXP int vszip_limiter(vszip_ctx *ctx, int dtype, const vszip_plane *pl, int n, const double *lo, const double *hi) {
    touch_table(pl, n, bps_of(dtype));
    for (int i = 0; i < n; ++i) (void)(lo[i] + hi[i]);
    return kernel_failed(ctx, "Limiter");
}
EXP int vszip_limit_filter(vszip_ctx *ctx, int dtype, const vszip_plane *pl, const void *const *refs, const ptrdiff_t *rs, int n, const float *d, const float *b, const float *e) {
    touch_table(pl, n, bps_of(dtype));
    for (int i = 0; i < n; ++i) {
        if (refs) touch(refs[i], rs[i], pl[i].w, pl[i].h, bps_of(dtype), false);
        (void)(d[i] + b[i] + e[i]);
    }
    return kernel_failed(ctx, "LimitFilter");
}
EXP int vszip_adaptive_binarize(vszip_ctx *ctx, const vszip_plane *pl, int n, int) {
    touch_table(pl, n, 1);
    return kernel_failed(ctx, "AdaptiveBinarize");
}
EXP int vszip_plane_minmax(vszip_ctx *ctx, int dtype, const vszip_plane *pl, int n, float, float, int, double *mn, double *mx, double *df) {
    touch_table(pl, n, bps_of(dtype));
    for (int i = 0; i < n; ++i) mn[i] = mx[i] = df[i] = 0;
    return kernel_failed(ctx, "PlaneMinMax");
}
EXP int vszip_bilateral_luts(vszip_ctx *ctx, vszip_bilateral_cfg *cfg, int hist_len) {
    if (!ctx || !cfg || hist_len <= 0) return VSZIP_ERR_ARG;
    cfg->gs_lut = cfg->gr_lut = nullptr;
    if (!cfg->process) return VSZIP_OK;
    if (fail_mode("alloc")) return VSZIP_ERR_NOMEM;
    cfg->gr_lut = (float *)malloc((size_t)hist_len * 4);
    cfg->gs_lut = (float *)malloc((size_t)(cfg->radius + 1) * (cfg->radius + 1) * 4);
    return VSZIP_OK;
}
EXP int vszip_bilateral(vszip_ctx *ctx, int dtype, const vszip_plane *pl, const vszip_bilateral_cfg *const *cfgs, int n, float) {
    touch_table(pl, n, bps_of(dtype));
    for (int i = 0; i < n; ++i) (void)*(volatile const int *)&cfgs[i]->radius;
    return kernel_failed(ctx, "Bilateral");
}
EXP int vszip_chain_run(vszip_ctx *ctx, int dtype, const vszip_chain_stage *st, int ns, const vszip_plane *pl, const int *slot, int n) {
    touch_table(pl, n, bps_of(dtype));
    for (int i = 0; i < n; ++i) (void)*(volatile const int *)&slot[i];
    for (int s = 0; s < ns; ++s) (void)*(volatile const int *)&st[s].kind;
    return kernel_failed(ctx, "chain");
}
EXP int vszip_ssimulacra2(vszip_ctx *ctx, const float *const *r, const float *const *d, ptrdiff_t stride, int w, int h, int np, double *scores) {
    for (int i = 0; i < 3 * np; ++i) {
        touch(r[i], stride, w, h, 4, false);
        touch(d[i], stride, w, h, 4, false);
    }
    for (int i = 0; i < np; ++i) scores[i] = 0;
    return kernel_failed(ctx, "SSIMULACRA2");
}
EXP int vszip_ssimulacra2_src(vszip_ctx *ctx, const vszip_ssim_source *f, const void *const *r, const void *const *d, ptrdiff_t stride, int w, int h, int np, double *scores) {
    const int per = f->family == VSZIP_CF_GRAY ? 1 : 3;
    for (int i = 0; i < per * np; ++i) {
        const bool chroma = f->family == VSZIP_CF_YUV && i % 3 != 0;  // a subsampled plane has its own pitch and size
        const int pw = chroma ? (w + (1 << f->ssw) - 1) >> f->ssw : w, ph = chroma ? (h + (1 << f->ssh) - 1) >> f->ssh : h;
        touch(r[i], chroma ? f->chroma_stride : stride, pw, ph, bps_of(f->dtype), false);
        touch(d[i], chroma ? f->chroma_stride : stride, pw, ph, bps_of(f->dtype), false);
    }
    for (int i = 0; i < np; ++i) scores[i] = 0;
    return kernel_failed(ctx, "SSIMULACRA2");
}
EXP int vszip_to_rgbs_linear(vszip_ctx *ctx, const vszip_ssim_source *f, const void *const *s, ptrdiff_t ss, float *const *d3, ptrdiff_t ds, int w, int h) {
    for (int i = 0; i < (f->family == VSZIP_CF_GRAY ? 1 : 3); ++i) {
        const bool chroma = f->family == VSZIP_CF_YUV && i != 0;
        touch(s[i], chroma ? f->chroma_stride : ss, chroma ? (w + (1 << f->ssw) - 1) >> f->ssw : w, chroma ? (h + (1 << f->ssh) - 1) >> f->ssh : h, bps_of(f->dtype), false);
    }
    for (int i = 0; i < 3; ++i) touch(d3[i], ds, w, h, 4, true);
    return kernel_failed(ctx, "to_rgbs_linear");
}
EXP int vszip_eedi3(vszip_ctx *ctx, const vszip_plane *pl, const float *const *sc, const ptrdiff_t *ss, int n, int, int horizontal, const vszip_eedi3_params *prm) {
    for (int i = 0; i < n; ++i) {
        touch(pl[i].src, pl[i].src_stride, pl[i].w, pl[i].h, 4, false);
        const int dw = horizontal && prm->dh ? pl[i].w * 2 : pl[i].w, dh = !horizontal && prm->dh ? pl[i].h * 2 : pl[i].h;
        touch(pl[i].dst, pl[i].dst_stride, dw, dh, 4, true);
        if (sc && sc[i]) touch(sc[i], ss[i], dw, dh, 4, false);
    }
    return kernel_failed(ctx, "EEDI3");
}
EXP int vszip_eedi3_mclip(vszip_ctx *ctx, const vszip_plane *pl, const float *const *sc, const ptrdiff_t *ss, const uint8_t *const *mc, const ptrdiff_t *ms, int n, int field, int horizontal,
                          const vszip_eedi3_params *prm) {
    for (int i = 0; i < n; ++i)
        if (mc && mc[i]) {
            const int dw = horizontal && prm->dh ? pl[i].w * 2 : pl[i].w, dh = !horizontal && prm->dh ? pl[i].h * 2 : pl[i].h;
            touch(mc[i], ms[i], dw, dh, 1, false);
        }
    return vszip_eedi3(ctx, pl, sc, ss, n, field, horizontal, prm);
}
EXP int vszip_xpsnr_wsse_batch(vszip_ctx *ctx, int bps, int nf, const void *const *org, const void *const *rec, const void *const *p1, const void *const *p2, const int *w, const int *h,
                               const ptrdiff_t *st, int, int nc, unsigned, int, uint64_t *out) {
    for (int f = 0; f < nf; ++f) {
        for (int c = 0; c < nc; ++c) {
            touch(org[f * nc + c], st[c], w[c], h[c], bps, false);
            touch(rec[f * nc + c], st[c], w[c], h[c], bps, false);
            out[3 * f + c] = 0;
        }
        if (p1 && p1[f]) touch(p1[f], st[0], w[0], h[0], bps, false);
        if (p2 && p2[f]) touch(p2[f], st[0], w[0], h[0], bps, false);
    }
    return kernel_failed(ctx, "XPSNR");
}
EXP int vszip_xpsnr_wsse(vszip_ctx *ctx, int bps, const void *const *org, const void *const *rec, const void *p1, const void *p2, const int *w, const int *h, const ptrdiff_t *st, int depth, int nc,
                         unsigned fr, int temporal, uint64_t *out) {
    return vszip_xpsnr_wsse_batch(ctx, bps, 1, org, rec, &p1, &p2, w, h, st, depth, nc, fr, temporal, out);
}
