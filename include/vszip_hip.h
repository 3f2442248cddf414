/*
 * vszip_hip.h — flat C ABI of the MI355X (gfx950) vszip pixel kernels.
 *
 * This is the drop-in boundary: the cut between the reference's filter wrappers
 * (src/vapoursynth/NAME.zig, layer L2) and its pixel kernels (src/filters/NAME.zig,
 * layer L3).  Every entry point replaces one L3 function the wrappers call, takes
 * plain pointers, strides IN ELEMENTS (as ZAPI getDimensions2 returns them,
 * src/vapoursynth/boxblur.zig:46) and scalars, and returns 0 or a negative
 * vszip_status.  No C++ / HIP / torch types appear in any signature; a Zig host
 * binds it with `extern "c"` declarations.
 *
 * All plane pointers are DEVICE pointers (hipMalloc, or any allocator that
 * yields device-accessible memory, e.g. a torch CUDA tensor's data_ptr).  Host
 * VSFrame planes are staged with vszip_copy_h2d_2d / vszip_copy_d2h_2d on the
 * context's stream.  Kernels are enqueued on the context's stream and return
 * without synchronising unless stated (the metric filters copy their scalars
 * back and synchronise, because their result is a host scalar).
 */
#ifndef VSZIP_HIP_H
#define VSZIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSZIP_ABI_VERSION 4 /* 4 (round 5): vszip_dev_alloc searches a bounded number of candidates and keeps nothing (below); vszip_dev_arena_info added; vszip_dev_trim, vszip_dev_placement_info and
                               vszip_dev_alloc_probed removed (nothing is searched for or parked any more); 3 (round 4): vszip_ctx_set_option / _get_option, vszip_dev_probe_region,
                               vszip_plane_average_async, vszip_plane_minmax_async added; 2 (round 3): vszip_ssim_source grew (YUV sources); entry points added since 1:
                               vszip_chain_run, vszip_ssimulacra2_src, vszip_to_rgbs_linear, vszip_probe_read_each, vszip_resample_table */

typedef struct vszip_ctx vszip_ctx;

/* sample types: helper.zig:59-108 DataType. U32 is accepted by vszip_plane_average and vszip_limiter only, like in
 * the reference (enable_u32, helper.zig:78), and without an exclude list (planeaverage.zig(vs):127). */
enum vszip_dtype { VSZIP_U8 = 0, VSZIP_U16 = 1, VSZIP_F16 = 2, VSZIP_F32 = 3, VSZIP_U32 = 4 };

enum vszip_status {
    VSZIP_OK = 0,
    VSZIP_ERR_ARG = -1,         /* invalid argument (the wrapper's create-time checks) */
    VSZIP_ERR_HIP = -2,         /* a HIP runtime call failed; see vszip_last_error */
    VSZIP_ERR_UNSUPPORTED = -3, /* valid in the reference, not built yet (DESIGN.md section 1 lists the cases) */
    VSZIP_ERR_NOMEM = -4
};

/* ---- context: one per (process, GPU); owns a stream and scratch ---------- */
int vszip_ctx_create(int device, vszip_ctx **out);
void vszip_ctx_destroy(vszip_ctx *ctx);
/* Use an externally owned hipStream_t (passed as void*) instead of the context's own. */
int vszip_ctx_set_stream(vszip_ctx *ctx, void *hip_stream);
void *vszip_ctx_stream(vszip_ctx *ctx);
int vszip_ctx_sync(vszip_ctx *ctx);
/* How vszip_copy_h2d_2d / _d2h_2d move host memory. 0 (default): straight from/to the caller's pointers
 * (right for pinned memory; pageable memory is pinned in place by the runtime, one copy at a time
 * per process). 1: through the context's own pinned arena with CPU copies — the DMA is asynchronous
 * and concurrent across contexts, D2H data reaches the caller's memory inside vszip_ctx_sync.
 * Env VSZIP_STAGING=pinned selects 1 for every new context. */
int vszip_ctx_set_staging(vszip_ctx *ctx, int mode);
/* Options. Every switch of the library (vapoursynth-zip_amd/csrc/options.inc lists them with their defaults) is read from the
 * environment ONCE, when the context is created, under the name given here (e.g. "VSZIP_PLACEMENT", "VSZIP_STAGING",
 * "VSZIP_RT_NO_ICHAIN"); these two change / read one on a live context. Flags are 0 / 1. VSZIP_ERR_ARG: no such option;
 * VSZIP_ERR_UNSUPPORTED: a development variant that this build does not contain (-DVSZIP_DEV_VARIANTS). */
int vszip_ctx_set_option(vszip_ctx *ctx, const char *name, int value);
int vszip_ctx_get_option(vszip_ctx *ctx, const char *name, int *value);
/* Error path of a caller that gives up on the current frame: drains the stream and forgets staged
 * D2H copies that have not reached their destination yet (the destinations may then be freed). */
int vszip_ctx_abort(vszip_ctx *ctx);
const char *vszip_last_error(vszip_ctx *ctx);
int vszip_abi_version(void);

/* ---- device memory + staging (replaces nothing: the reference is host-only) */
/* vszip_dev_alloc. Requests of VSZIP_PLACEMENT_MIN_MIB (256) or more are PLACED: kernels with thousands of concurrent row
 * streams (the BoxBlur ring kernels) run 15-20 % slower with their destination planes in most physical memory than in some,
 * a stable property of the allocation that nothing user space can see predicts (profiles/r05_placement.md). Up
 * to VSZIP_PLACEMENT_TRIES (24) candidate allocations of the requested size are made (all held meanwhile, so each lies
 * elsewhere; never more than a quarter of what hipMemGetInfo reports free), each classified with a 2 ms copy in the ring
 * kernels' access shape; the search ends with the first candidate of the best class, else the fastest is kept, and every other
 * candidate is freed before the call returns (a few milliseconds per candidate; more where the driver clears memory that was
 * used before). The search is bounded: no candidate is started after VSZIP_PLACEMENT_BUDGET_MS (300) of wall clock; it ends early
 * at a candidate within 1.5 % of the best probe rate this process has seen on the device (a device's first search: at the
 * calibrated best class); and a device whose searches found all candidates alike twice in a row (one class of memory, or too
 * busy to measure) is served with plain hipMalloc from then on. No memory is cached or parked; the only state is that per-device
 * record (best rate seen, verdict). vszip_dev_free is hipFree.
 * VSZIP_PLACEMENT=0 (or vszip_ctx_set_option): plain hipMalloc. */
int vszip_dev_alloc(vszip_ctx *ctx, size_t bytes, void **dptr);
/* What the search did for `dptr` (as returned by vszip_dev_alloc); any pointer may be NULL. *candidates = 0: a plain allocation;
 * *probe_bytes_per_second: the classification copy's rate on the one kept (0: one candidate, not probed). */
int vszip_dev_arena_info(vszip_ctx *ctx, const void *dptr, int *candidates, double *probe_bytes_per_second, double *search_ms);
/* Diagnostic: a copy in the ring kernels' access shape on a caller's region (overwrites its contents): bytes / s. `from` == NULL: tiles
 * are read and written inside the region; else they are read from `from` (another region of at least `bytes`, not modified). */
int vszip_dev_probe_region(vszip_ctx *ctx, void *dptr, size_t bytes, const void *from, double *bytes_per_second);
/* any context of the process may free any allocation of that device; NULL is accepted */
int vszip_dev_free(vszip_ctx *ctx, void *dptr);
int vszip_dev_memset(vszip_ctx *ctx, void *dptr, int value, size_t bytes);
int vszip_host_alloc_pinned(vszip_ctx *ctx, size_t bytes, void **hptr);
int vszip_host_free_pinned(vszip_ctx *ctx, void *hptr);
/* pitches in BYTES; async on the context stream: a copy is complete when vszip_ctx_sync returns */
int vszip_copy_h2d_2d(vszip_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows);
int vszip_copy_d2h_2d(vszip_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows);
int vszip_copy_d2d_2d(vszip_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows);

/* ---- timing helpers (HIP events on the context stream; used by bench.py) -- */
int vszip_timer_start(vszip_ctx *ctx);
int vszip_timer_stop_ms(vszip_ctx *ctx, float *ms); /* synchronises */
/* Dominant-kernel probe: while enabled, every filter call brackets the launch of its dominant
 * kernel (BoxBlur: the CT ring kernel; Bilateral: the truncated-window kernel; SSIMULACRA2: the
 * per-scale maps kernel) with HIP events on the context's stream. vszip_probe_read synchronises,
 * returns the summed kernel time and the number of launches, and resets the probe. Measurement
 * aid for bench.py's roofline (the figure rocprofv3 --kernel-trace reports per kernel). */
int vszip_probe_enable(vszip_ctx *ctx, int on);
int vszip_probe_read(vszip_ctx *ctx, double *total_ms, int *launches);
/* the same, also copying the first `cap` launch durations (ms, launch order) to each_ms */
int vszip_probe_read_each(vszip_ctx *ctx, double *total_ms, int *launches, float *each_ms, int cap);

/* One plane of one frame. Strides in elements of the sample type. */
typedef struct vszip_plane {
    const void *src; /* input plane */
    void *dst;       /* output plane (filters that write pixels) */
    const void *ref; /* second input: Bilateral `ref`, PlaneAverage/MinMax `clipb`; NULL if none */
    ptrdiff_t src_stride;
    ptrdiff_t dst_stride;
    ptrdiff_t ref_stride;
    int32_t w;
    int32_t h;
} vszip_plane;

/*
 * BoxBlur — replaces boxblur_ct.hvBlur (src/filters/boxblur_comptime.zig:10) and
 * the RT chain hblur / vblur / hvBlurFused (src/filters/boxblur_runtime.zig:121,
 * 153, 283) as dispatched by src/vapoursynth/boxblur.zig:85-113,188-209.
 * Processes `nplanes` independent planes (any mix of sizes, e.g. Y,U,V of many
 * frames) in one call; path choice (CT vs RT) follows boxblur.zig:188.
 * Errors mirror boxBlurCreate (boxblur.zig:150-179): nothing to be performed,
 * 2*radius >= plane size -> VSZIP_ERR_ARG.
 */
int vszip_boxblur(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes,
                  int hradius, int hpasses, int vradius, int vpasses);

/*
 * PlaneAverage — replaces filter.average / filter.averageRef
 * (src/filters/planeaverage.zig:26,47) called from src/vapoursynth/planeaverage.zig:55-61.
 * planes[i].ref != NULL on plane 0 selects the clipb variant for all planes.
 * `exclude` is the i32 list of the wrapper (:122-137; compared as @floatFromInt for
 * float clips), any length (up to 256 distinct values). bits_per_sample gives peak = 2^bits - 1 (:115).
 * Results are written to host arrays avg[nplanes] and (clipb) diff[nplanes]; the call
 * synchronises the stream. Integer planes are exact; float planes differ from the
 * reference's sequential f64 sum by rounding only.
 */
int vszip_plane_average(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes,
                        const int32_t *exclude, int nexclude, int bits_per_sample,
                        double *avg, double *diff);
/* The same without the synchronise (round 4): the per-plane results are written by the kernels straight into the caller's PINNED
 * host array `pinned_results` (vszip_host_alloc_pinned; nplanes x 4 doubles: [i][0] = avg, [i][1] = diff for the clipb variant)
 * and are valid after the caller's next vszip_ctx_sync — a host that reads several statistics of a frame queues them all and
 * waits once. */
int vszip_plane_average_async(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes,
                              const int32_t *exclude, int nexclude, int bits_per_sample, double *pinned_results);

/*
 * Limiter — replaces the getFrame bodies of LimiterRT / Limiter (src/vapoursynth/limiter.zig:28-96):
 * dst = min(max(lo, x), hi) in the sample type, per plane. lo[i] / hi[i] are the bounds the wrapper
 * resolved for planes[i] — the min/max arrays (u32 for integer clips, f32 for float clips), or the
 * comptime range tables of src/filters/limiter.zig:66-91 (full / tv_range yuv / rgb per depth,
 * yuvf / rgbf). dtype may be VSZIP_U32. Asynchronous on the context stream.
 */
int vszip_limiter(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes, const double *lo,
                  const double *hi);

/*
 * LimitFilter — replaces filter.process (src/filters/limit_filter.zig:3-34) called from
 * LimitFilter(T, refb).getFrame (src/vapoursynth/limit_filter.zig:27-80). planes[i].src is the
 * FILTERED clip's plane (`flt`), planes[i].ref the SOURCE clip's (`src`), planes[i].dst the output;
 * refs[i] (may be NULL, as may the array) is the optional third clip the difference is taken
 * against (default: the source), strides in elements. dark_thr / bright_thr are on the clip's own
 * scale (the wrapper applies hz.scaleValue, src/helper.zig:312-336, to the 8-bit-scale arguments),
 * elast as given; one value per plane. Asynchronous on the context stream.
 */
int vszip_limit_filter(vszip_ctx *ctx, int dtype, const vszip_plane *planes, const void *const *refs,
                       const ptrdiff_t *ref_strides, int nplanes, const float *dark_thr,
                       const float *bright_thr, const float *elast);

/*
 * AdaptiveBinarize — replaces the getFrame body of src/vapoursynth/adaptive_binarize.zig:26-73:
 * 8-bit planes, dst = 255 where clip2 - clip >= c (compared in i16, c clamped to [-256, 256]
 * :96-99), else 0. planes[i].src is `clip`'s plane, planes[i].ref `clip2`'s. Asynchronous.
 */
int vszip_adaptive_binarize(vszip_ctx *ctx, const vszip_plane *planes, int nplanes, int c);

/*
 * PlaneMinMax — replaces filter.minMax / minMaxRef / minMaxNoThr / minMaxNoThrRef
 * (src/filters/planeminmax.zig:72-133) called from src/vapoursynth/planeminmax.zig:60-77.
 * minthr == maxthr == 0 is the exact path; otherwise the histogram-percentile rule of
 * minMaxImpl (:43-57). vmin/vmax hold integers for integer clips and idx/65535 (as f32)
 * for float clips, exactly the values the wrapper stores in psmMin/psmMax (:59-68).
 */
int vszip_plane_minmax(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes,
                       float minthr, float maxthr, int bits_per_sample,
                       double *vmin, double *vmax, double *diff);
/* ... and without the synchronise: pinned_results[i][0] = min, [i][1] = max, [i][2] = diff (see vszip_plane_average_async); [i][3] is scratch
 * (a thresholded sweep may leave a marker there) */
int vszip_plane_minmax_async(vszip_ctx *ctx, int dtype, const vszip_plane *planes, int nplanes,
                             float minthr, float maxthr, int bits_per_sample, double *pinned_results);

/*
 * Bilateral — replaces filter.bilateral (src/filters/bilateral.zig:81) and the create-time
 * work of bilateralCreate (src/vapoursynth/bilateral.zig:104-231).
 *
 * One vszip_bilateral_cfg per PLANE INDEX of the clip (Y,U,V / R,G,B):
 *   vszip_bilateral_derive  fills sigmaS/sigmaR/process/algorithm/pbficnum/radius/step/samples
 *                           from the user arrays (already expanded to 3 entries with hz.getArray's
 *                           repeat-last rule; sigmaS is passed raw with its element count because
 *                           its chroma default depends on the subsampling, :104-124). Host only.
 *                           Returns VSZIP_ERR_ARG for the cases bilateralCreate rejects.
 *   vszip_bilateral_luts    builds gs_lut ((radius+1)^2) and gr_lut (hist_len) exactly as
 *                           bilateral.zig:306-339 and uploads them (device memory owned by the
 *                           caller: vszip_dev_free). hist_len = 1 << bits, 65536 for float clips.
 *   vszip_bilateral         filters `nplanes` planes; cfgs[i] is the config of planes[i]'s plane
 *                           index. planes[i].ref == NULL means ref == src (no joint clip).
 *                           peak = hist_len - 1 as float (:101-102).
 * Planes with process == 0 (sigmaS == 0 or sigmaR == 0) are pass-through in the reference
 * (newVideoFrame2 copies them); copy them with vszip_copy_d2d_2d.
 */
typedef struct vszip_bilateral_cfg {
    double sigmaS;
    double sigmaR;
    int32_t process;
    int32_t algorithm; /* 1 = PBFIC, 2 = truncated window */
    int32_t pbficnum;
    int32_t radius;
    int32_t step;
    int32_t samples;
    float *gs_lut; /* device */
    float *gr_lut; /* device */
} vszip_bilateral_cfg;

int vszip_bilateral_derive(const double *sigmaS, int n_sigmaS, const double *sigmaR3, const int *algorithm3,
                           const int *pbficnum3, int is_yuv, int subsampling_w, int subsampling_h,
                           const int *planes3, vszip_bilateral_cfg *out3);
int vszip_bilateral_luts(vszip_ctx *ctx, vszip_bilateral_cfg *cfg, int hist_len);
int vszip_bilateral(vszip_ctx *ctx, int dtype, const vszip_plane *planes,
                    const vszip_bilateral_cfg *const *cfgs, int nplanes, float peak);

/*
 * Chained pixel filters on resident planes (frames stay on the device between
 * chained vszip filters). Every filter entry point takes device pointers, so a host may simply call them one
 * after another on one context; vszip_chain_run does that for the frame-in / frame-out filters in ONE call and
 * owns the intermediate planes: upload once (vszip_copy_h2d_2d), vszip_chain_run, download once. What the
 * reference does as separate filter instances with a host frame in between — clip.vszip.Bilateral().vszip.BoxBlur()
 * = bilateralGetFrame (src/vapoursynth/bilateral.zig) feeding BoxBlur's getFrame (src/vapoursynth/boxblur.zig:29-50).
 *   stages[s].process[k]  whether stage s filters plane slot k (0..2: Y/U/V or R/G/B); other planes pass through,
 *                         as the reference's newVideoFrame2 plane copy does;
 *   planes[i]             src = resident input, dst = where the chain's result goes; plane_slot[i] in 0..2 picks
 *                         the per-plane parameters (a table may hold the planes of many frames).
 * libvszip.so applies the same scheme across filter INSTANCES: a vszip filter created on the output of another
 * vszip pixel filter runs that filter's kernels itself on its own upload ("Fused chains").
 */
enum { VSZIP_STAGE_BOXBLUR = 0, VSZIP_STAGE_BILATERAL = 1, VSZIP_STAGE_LIMITER = 2 };
typedef struct vszip_chain_stage {
    int32_t kind;
    int32_t process[3];
    int32_t hradius, hpasses, vradius, vpasses; /* BoxBlur */
    const struct vszip_bilateral_cfg *bilateral[3]; /* Bilateral: per plane slot, LUTs resident (vszip_bilateral_luts) */
    float peak;                                     /* Bilateral: as vszip_bilateral */
    double lo[3], hi[3];                            /* Limiter: resolved bounds per plane slot */
} vszip_chain_stage;
int vszip_chain_run(vszip_ctx *ctx, int dtype, const vszip_chain_stage *stages, int nstages,
                    const vszip_plane *planes, const int *plane_slot, int nplanes);

/*
 * SSIMULACRA2 — replaces filter_ssim.process (src/filters/ssimulacra2.zig:46) called from
 * ssimulacra2GetFrame (src/vapoursynth/ssimulacra2.zig:40-66). Inputs are linear-light
 * RGBS planes (what hz.toRGBS + sRGBtoLinearRGB hand to the kernel, :115-118): ref3 / dis3
 * are HOST arrays of 3 * npairs device pointers (R,G,B of pair 0, R,G,B of pair 1, ...),
 * all planes w x h with the same stride (elements). scores[npairs] receives the value the
 * wrapper stores in the "SSIMULACRA2" frame property. Synchronises the stream.
 */
int vszip_ssimulacra2(vszip_ctx *ctx, const float *const *ref3, const float *const *dis3,
                      ptrdiff_t stride, int w, int h, int npairs, double *scores);

/*
 * SSIMULACRA2 with its colour pre-stage on the device — additionally replaces, for the clips that
 * need no resampler, what ssimulacraCreate hangs in front of the kernel on the host:
 * hz.toRGBS (src/helper.zig:225-243: resize.Bicubic(format=RGBS, matrix_in=709|601)) and
 * sRGBtoLinearRGB (src/vapoursynth/ssimulacra2.zig:132-162: std.SetFrameProp(_Transfer=13) +
 * resize.Bicubic(transfer=LINEAR)), i.e. zimg's integer -> float conversion (full range for RGB,
 * limited for Gray), Gray -> R = G = B, and the sRGB EOTF through zimg's approximate-gamma table
 * (VapourSynth's resize default). The source planes are uploaded as they are (8/16-bit samples: a
 * quarter / half of the RGBS bytes), the conversion is fused into the first SSIMULACRA2 pass.
 *   family     VSZIP_CF_RGB (3 planes per frame), VSZIP_CF_GRAY (1 plane per frame) or — round 3 — VSZIP_CF_YUV
 *              (3 planes, chroma subsampled by 2^ssw x 2^ssh): zimg's integer -> float conversion per plane, the
 *              chroma planes brought to 4:4:4 with its Catmull-Rom resampler (resize.Bicubic: b = 0, c = 0.5;
 *              horizontal pass first, two interleaved FMA accumulators per sample — vszip_resample_table below),
 *              the YUV -> RGB matrix as an FMA chain, then the transfer table. Restated from zimg's published
 *              algorithm and pinned by the reference's seven YUV SSIMULACRA2 goldens (tests/test_oracle_zimg_goldens.py);
 *   dtype/bits VSZIP_U8 (8), VSZIP_U16 (9..16) or VSZIP_F32 (f16 is rejected by the wrapper, :106-113);
 *   limited    integer samples are limited range (zimg's default for Gray and YUV) or full (RGB);
 *   linearize  0 when frame 0 carries _Transfer == LINEAR (:139-141), else 1;
 *   YUV only:  ssw / ssh   log2 chroma subsampling (0..2). A sited chroma plane (chroma_loc != center) lies half a LUMA
 *                          sample off the centre of its 2^ss luma samples whatever ss — zimg's rule (0.5 / 2^ss chroma
 *                          samples); ss = 2 (4:1:0, 4:1:1) is not covered by any reference golden, and libvszip.so
 *                          leaves such clips to the host's resize;
 *              matrix      _Matrix of the clip's frames if set and specified, else what hz.toRGBS passes as matrix_in
 *                          (1 = BT.709 if height > 650 else 6 = BT.601; src/helper.zig:231) — VapourSynth's resize
 *                          lets a frame property win over the *_in argument; supported: 1, 5, 6, 9;
 *              chroma_loc  _ChromaLocation (0 left — the default —, 1 center, 2 top-left, 3 top, 4 bottom-left, 5 bottom);
 *              chroma_stride  row pitch of the U and V planes, elements.
 * ref_planes / dis_planes: HOST arrays of npairs * (3 | 1) device plane pointers, same stride (elements).
 */
enum { VSZIP_CF_RGB = 0, VSZIP_CF_GRAY = 1, VSZIP_CF_YUV = 2 };
typedef struct vszip_ssim_source {
    int family, dtype, bits, limited, linearize;
    int ssw, ssh, matrix, chroma_loc; /* VSZIP_CF_YUV */
    ptrdiff_t chroma_stride;          /* VSZIP_CF_YUV */
} vszip_ssim_source;
int vszip_ssimulacra2_src(vszip_ctx *ctx, const vszip_ssim_source *fmt, const void *const *ref_planes,
                          const void *const *dis_planes, ptrdiff_t stride, int w, int h, int npairs, double *scores);
/* The pre-stage alone: one frame's planes -> linear-light RGBS planes (dst3: R, G, B device pointers).
 * Asynchronous on the context stream. */
int vszip_to_rgbs_linear(vszip_ctx *ctx, const vszip_ssim_source *fmt, const void *const *src_planes, ptrdiff_t src_stride,
                         float *const *dst3, ptrdiff_t dst_stride, int w, int h);
/* zimg's Catmull-Rom table for one axis of an upscale (device-free): output sample i = sum_k coef4[4 i + k] *
 * src[min(left[i] + k, src_dim - 1)], accumulated as (c0 x0 + fma(c2, x2, .)) + (c1 x1 + fma(c3, x3, .)).
 * shift: position offset in source samples (4:2:0 left-sited chroma -> luma grid: 0.25 horizontally, 0 vertically). */
int vszip_resample_table(int src_dim, int dst_dim, double shift, int32_t *left, float *coef4);

/*
 * EEDI3 / EEDI3H — replaces processPlane (src/vapoursynth/eedi3.zig:26-140) with its kernels
 * interpLine / vcheckLine (src/filters/eedi3.zig:349-592, 915-1046) and, for horizontal != 0,
 * the transposes of :220-246. 32-bit float planes only (createImpl :316-319).
 * planes[i].src is src_w x src_h; planes[i].dst is src_w x (dh ? 2*src_h : src_h) for EEDI3 and
 * (dh ? 2*src_w : src_w) x src_h for EEDI3H. `field` is the frame's resolved parity 0/1
 * (getFrame :166-172 folds _FieldBased and field 2/3 into it). sclips[i] (may be NULL, as may
 * the array) has dst's geometry. Parameters are the user-level ones (defaults: alpha .2,
 * beta .25, gamma 20, nrad 2, mdis 20, hp 0, vcheck 2, vthresh 32/64/4); the scaling of
 * :465-473 happens inside. Errors mirror createImpl's messages.
 */
typedef struct vszip_eedi3_params {
    int32_t dh;
    float alpha, beta, gamma;
    int32_t nrad, mdis, hp, vcheck;
    float vthresh0, vthresh1, vthresh2;
} vszip_eedi3_params;

int vszip_eedi3(vszip_ctx *ctx, const vszip_plane *planes, const float *const *sclips,
                const ptrdiff_t *sclip_strides, int nplanes, int field, int horizontal,
                const vszip_eedi3_params *params);
/* The same with the `mclip` mask (createImpl :393-433, buildBmask src/filters/eedi3.zig:285-304):
 * mclips[i] (may be NULL, as may the array) is an 8-bit plane with the geometry of planes[i].src —
 * the wrapper passes the same Gray mask for every plane, like getFrame :215-218 — strides in
 * bytes == elements. Pixels whose +-mdis neighbourhood holds no mask sample keep the plain
 * vertical cubic. */
int vszip_eedi3_mclip(vszip_ctx *ctx, const vszip_plane *planes, const float *const *sclips,
                      const ptrdiff_t *sclip_strides, const uint8_t *const *mclips,
                      const ptrdiff_t *mclip_strides, int nplanes, int field, int horizontal,
                      const vszip_eedi3_params *params);

/*
 * XPSNR — replaces filter.getWSSE (src/filters/xpsnr.zig:376) called from XPSNR(T).getFrame
 * (src/vapoursynth/xpsnr.zig:72-82): org3/rec3 are host arrays of num_comps device plane
 * pointers (u8 or u16 samples: bytes_per_sample 1 or 2, depth 8 or 10), prev1/prev2 the luma of
 * reference frames n-1 / n-2 or NULL (n == 0/1, or temporal off). wsse3 receives wsse64[0..2].
 * Synchronises the stream. vszip_xpsnr_value / _average are getFrameXPSNR / getAvgXPSNR
 * (:359-374), host-only helpers for the wrapper's props and its per-clip summary.
 */
int vszip_xpsnr_wsse(vszip_ctx *ctx, int bytes_per_sample, const void *const *org3, const void *const *rec3,
                     const void *prev1, const void *prev2, const int *width3, const int *height3,
                     const ptrdiff_t *stride3, int depth, int num_comps, unsigned frame_rate, int temporal,
                     uint64_t *wsse3);
/* The same for a batch of frames in one launch (what a host that already holds several frames of
 * the clip calls; frame f of the batch is exactly vszip_xpsnr_wsse on its own pointers):
 * org3/rec3 hold nframes*num_comps plane pointers, frame-major; prev1/prev2 (arrays of nframes
 * luma pointers, entries or the arrays themselves may be NULL) are per frame; every frame shares
 * the geometry. wsse3 receives 3 values per frame. */
int vszip_xpsnr_wsse_batch(vszip_ctx *ctx, int bytes_per_sample, int nframes, const void *const *org3,
                           const void *const *rec3, const void *const *prev1, const void *const *prev2,
                           const int *width3, const int *height3, const ptrdiff_t *stride3, int depth,
                           int num_comps, unsigned frame_rate, int temporal, uint64_t *wsse3);
double vszip_xpsnr_value(uint64_t wsse, uint64_t width, uint64_t height, int depth);
double vszip_xpsnr_average(double sum_wdist, double sum_xpsnr, uint64_t width, uint64_t height, int depth,
                           uint64_t num_frames);

#ifdef __cplusplus
}
#endif
#endif /* VSZIP_HIP_H */
