// Shared host/device helpers for the gfx950 vszip kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vszip_hip.h"

#define VSZIP_EXPORT extern "C" __attribute__((visibility("default")))

// Timing-only macros compile kernels that knowingly produce WRONG results (a phase stubbed out to see what it costs; the measurements are
// on file under profiles/). A stray -D in a build script must not ship one: they exist only in development builds.
#if !defined(VSZIP_DEV_VARIANTS) &&                                                                                                              \
    (defined(VSZIP_XPSNR_TIMING_NOTEMP) || defined(VSZIP_XPSNR_TIMING_NOEDGE) || defined(VSZIP_SSIM_TIMING_NOF64) || defined(VSZIP_SSIM_TIMING_NOFETCH) || \
     defined(VSZIP_E3_DIAG_ONE_PASS) || defined(VSZIP_E3_DIAG_NO_DP) || defined(VSZIP_E3_DIAG_NO_BACKTRACK) || defined(VSZIP_DIAG_NO_KCOL) ||        \
     defined(VSZIP_DIAG_NO_FENCE) || defined(VSZIP_DIAG_NO_STORE) || defined(VSZIP_DIAG_NO_LOAD) || defined(VSZIP_DIAG_NO_HORIZONTAL))
#error "VSZIP_*_TIMING_* / VSZIP_*DIAG_* macros build kernels with wrong results: they need -DVSZIP_DEV_VARIANTS (tools/variant.sh)"
#endif

// csrc/options.inc: every switch, parsed once per context (ctx.hip)
struct vszip_options {
#define VSZIP_OPT(field, env, def) int field = def;
#ifdef VSZIP_DEV_VARIANTS
#define VSZIP_DEV_OPT(field, env, def) int field = def;
#else
#define VSZIP_DEV_OPT(field, env, def) static constexpr int field = def;
#endif
#include "options.inc"
#undef VSZIP_OPT
#undef VSZIP_DEV_OPT
};

struct vszip_ctx {
    vszip_options opt;
    int device = 0;
    int num_cus = 0;  // compute units of `device` (0: unknown)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // second stream for kernels that run BESIDE the main stream's (EEDI3: the line kernel of the short planes next to the
    // vertical-consistency chains of the tall ones); its CU mask leaves `aux_reserved` CUs to the main stream, created on
    // first use; aux_fork / aux_join order the two
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    int aux_reserved = 0;  // CUs the aux stream's mask leaves free for the main stream's kernels
    // plain second stream (SSIMULACRA2: the small scales' launch-bound kernels beside the large scales' maps kernels)
    hipStream_t side_stream = nullptr;
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    std::string err;
    // grow-only device scratch (filters that need intermediates) and a small
    // pinned + device pair for returning scalars
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    void *scalars_dev = nullptr;
    void *scalars_host = nullptr;
    size_t scalars_bytes = 0;
    // dominant-kernel probe (vszip_probe_*): HIP event pairs recorded on `stream` around the
    // kernel a filter names as its dominant one; resolved lazily in vszip_probe_read
    bool probe_on = false;
    std::vector<hipEvent_t> probe_events;  // begin/end pairs
    size_t probe_used = 0;
    // XPSNR block sums: a buffer of its own that the weighting kernel leaves zeroed for the next call
    void *xpsnr_sums = nullptr;
    size_t xpsnr_sums_bytes = 0;
    bool xpsnr_clean = false;
    // Host staging (vszip_ctx_set_staging): 0 = copy straight from/to the caller's pointers (pinned
    // callers, or pageable memory that the runtime pins in place), 1 = through this context's pinned
    // arena with CPU copies — H2D: copy in, DMA asynchronously; D2H: DMA now, copy out in vszip_ctx_sync.
    int &staging = opt.staging;
    char *stage = nullptr;
    size_t stage_bytes = 0, stage_used = 0;
    struct PendingOut {
        void *dst;
        size_t dpitch;
        const char *src;
        size_t wb, rows;
    };
    std::vector<PendingOut> pending_out;
    // thresholded PlaneMinMax on 16-bit / float planes: where the previous call's answers lay, per plane (device, [planes][2] words), and what that
    // call looked like — a call of the same shape (the next frames of a clip) looks there first and reads its planes once (planestats.hip)
    void *minmax_pred = nullptr;
    size_t minmax_pred_planes = 0;
    std::vector<uint64_t> minmax_sig;   // per plane group of a call, kPredSlots entries: signatures of the thresholded calls that left predictions (0: none)
    std::vector<uint64_t> minmax_used;  // ... and when each was last used (minmax_tick)
    uint64_t minmax_tick = 0;
    int minmax_predicted = 0;  // plane groups of thresholded calls that ran the single predicted sweep ("VSZIP_STAT_MINMAX_PREDICTED": read-only)
    int minmax_fallbacks = 0;  // synchronous predicted calls whose flags asked for the two sweeps (vszip_ctx_get_option "VSZIP_STAT_MINMAX_FALLBACKS": read-only)
    void *chain_buf = nullptr;  // vszip_chain_run: intermediate planes (grow-only)
    size_t chain_bytes = 0;
    void *ssim_lut = nullptr;  // SSIMULACRA2 colour pre-stage: cached conversion table (ssimulacra2.hip)
    int &scan_mode = opt.scan_mode;  // BoxBlur CT: 0 = ring kernel (DPP scan), 1 = generic kernel + shuffle scan, 2 = generic kernel + DPP scan
    hipEvent_t probe_ev0 = nullptr, probe_ev1 = nullptr;  // vszip_dev_probe_region's own timing events
};

int vszip_set_error(vszip_ctx *ctx, int code, const char *fmt, ...);
int vszip_ensure_scratch(vszip_ctx *ctx, size_t bytes);
hipError_t vszip_hip_malloc(vszip_ctx *ctx, void **p, size_t bytes);  // the library's internal allocations (plain hipMalloc)
int vszip_ensure_scalars(vszip_ctx *ctx, size_t bytes);
void vszip_ssim_release(vszip_ctx *ctx);  // frees ctx->ssim_lut
void vszip_chain_release(vszip_ctx *ctx);  // frees ctx->chain_buf
void vszip_planestats_release(vszip_ctx *ctx);  // frees ctx->minmax_pred
void vszip_bilateral_forget_lut(const void *dptr);  // vszip_dev_free: a packed range LUT goes with its allocation
// Bracket the launch of a filter's dominant kernel; no-ops unless the probe is enabled.
void vszip_probe_mark(vszip_ctx *ctx);
void vszip_aux_register(vszip_ctx *ctx);  // ctx.hip: the context owns a CU-masked stream (destroyed at exit if still alive)

struct vszip_probe_scope {
    vszip_ctx *c;
    explicit vszip_probe_scope(vszip_ctx *ctx) : c(ctx) { if (c->probe_on) vszip_probe_mark(c); }
    ~vszip_probe_scope() { if (c->probe_on) vszip_probe_mark(c); }
};

#define VSZIP_HIP_CHECK(ctx, call)                                                              \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return vszip_set_error((ctx), VSZIP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static inline int vszip_dtype_size(int dt) {
    switch (dt) {
        case VSZIP_U8: return 1;
        case VSZIP_U16: return 2;
        case VSZIP_F16: return 2;
        case VSZIP_F32: return 4;
        case VSZIP_U32: return 4;
    }
    return 0;
}

// ---- device-side helpers ---------------------------------------------------
#if defined(__HIPCC__)

// Inclusive prefix sum across the 64 lanes of a wave, DPP form (gfx9 row_shr /
// row_bcast). update_dpp(old=0, src, ...) yields 0 for lanes whose source lane
// is outside the row / masked off.
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_scan_shfl(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Orders LDS traffic between the lanes of ONE wave (single-wave workgroups): no s_barrier, no
// vmcnt drain.
__device__ __forceinline__ void vszip_wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    return v;  // valid in lane 0
}

#endif  // __HIPCC__
