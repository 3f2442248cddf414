// Context, device memory, staging copies and timers behind include/vszip_hip.h.
#include <vector>

#include "common.hpp"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>

int vszip_set_error(vszip_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}

int vszip_ensure_scratch(vszip_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return VSZIP_OK;
    if (ctx->scratch) {
        VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        VSZIP_HIP_CHECK(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    const size_t want = bytes + (bytes >> 3) + 4096;
    if (hipMalloc(&ctx->scratch, want) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "scratch allocation of %zu bytes failed", want);
    ctx->scratch_bytes = want;
    return VSZIP_OK;
}

int vszip_ensure_scalars(vszip_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->scalars_bytes) return VSZIP_OK;
    if (ctx->scalars_dev) {
        VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->scalars_dev);
        (void)hipHostFree(ctx->scalars_host);
        ctx->scalars_dev = ctx->scalars_host = nullptr;
        ctx->scalars_bytes = 0;
    }
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    if (hipMalloc(&ctx->scalars_dev, want) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "scalar buffer allocation failed");
    if (hipHostMalloc(&ctx->scalars_host, want, hipHostMallocDefault) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "pinned scalar buffer allocation failed");
    ctx->scalars_bytes = want;
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_abi_version(void) { return VSZIP_ABI_VERSION; }

// Streams created with a CU mask must be gone before the process tears the runtime down: with rocprofv3 attached
// a live one crashes the tool's finaliser (SIGSEGV inside __cxa_finalize, after the outputs are written). Contexts
// that own one are listed here and an atexit handler - registered after the runtime's and the profiler's own, so it
// runs before them - destroys what the host left alive.
// (heap objects that are never destroyed: a vszip_ctx_destroy from a late finaliser, after static destruction, must still
// find a live mutex and set - ADVICE r2; packed_luts() in bilateral.hip does the same)
static std::mutex &g_aux_mu_ref() {
    static std::mutex *m = new std::mutex();
    return *m;
}
static std::set<vszip_ctx *> &g_aux_ctxs_ref() {
    static std::set<vszip_ctx *> *s = new std::set<vszip_ctx *>();
    return *s;
}
#define g_aux_mu g_aux_mu_ref()
#define g_aux_ctxs g_aux_ctxs_ref()
static void vszip_aux_atexit() {
    std::lock_guard<std::mutex> lk(g_aux_mu);
    for (vszip_ctx *c : g_aux_ctxs)
        if (c->aux_stream) {
            (void)hipSetDevice(c->device);
            (void)hipStreamSynchronize(c->aux_stream);
            (void)hipStreamDestroy(c->aux_stream);
            c->aux_stream = nullptr;
        }
    g_aux_ctxs.clear();
}
void vszip_aux_register(vszip_ctx *ctx) {
    static std::once_flag once;
    std::call_once(once, [] { std::atexit(vszip_aux_atexit); });
    std::lock_guard<std::mutex> lk(g_aux_mu);
    g_aux_ctxs.insert(ctx);
}
static void vszip_aux_forget(vszip_ctx *ctx) {
    std::lock_guard<std::mutex> lk(g_aux_mu);
    g_aux_ctxs.erase(ctx);
}

VSZIP_EXPORT int vszip_ctx_create(int device, vszip_ctx **out) {
    if (!out) return VSZIP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return VSZIP_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return VSZIP_ERR_HIP;
    vszip_ctx *c = new vszip_ctx();
    c->device = device;
    (void)hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return VSZIP_ERR_HIP;
    }
    c->own_stream = true;
    (void)hipEventCreate(&c->ev0);
    (void)hipEventCreate(&c->ev1);
    const char *st = getenv("VSZIP_STAGING");
    if (st) c->staging = (strcmp(st, "pinned") == 0 || strcmp(st, "1") == 0) ? 1 : 0;
    const char *sm = getenv("VSZIP_SCAN_MODE");
    if (sm) c->scan_mode = atoi(sm);
    *out = c;
    return VSZIP_OK;
}

VSZIP_EXPORT void vszip_ctx_destroy(vszip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->xpsnr_sums) (void)hipFree(ctx->xpsnr_sums);
    vszip_ssim_release(ctx);
    vszip_chain_release(ctx);
    if (ctx->scalars_dev) (void)hipFree(ctx->scalars_dev);
    if (ctx->scalars_host) (void)hipHostFree(ctx->scalars_host);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    vszip_aux_forget(ctx);
    if (ctx->aux_stream) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
    }
    if (ctx->side_stream) {
        (void)hipStreamSynchronize(ctx->side_stream);
        (void)hipStreamDestroy(ctx->side_stream);
    }
    if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
    if (ctx->side_join) (void)hipEventDestroy(ctx->side_join);
    if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
    if (ctx->aux_join) (void)hipEventDestroy(ctx->aux_join);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->probe_events) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

VSZIP_EXPORT int vszip_ctx_set_stream(vszip_ctx *ctx, void *hip_stream) {
    if (!ctx) return VSZIP_ERR_ARG;
    if (ctx->own_stream && ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return VSZIP_OK;
}

VSZIP_EXPORT void *vszip_ctx_stream(vszip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

static void flush_pending_out(vszip_ctx *ctx);

VSZIP_EXPORT int vszip_ctx_sync(vszip_ctx *ctx) {
    if (!ctx) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->staging == 1) flush_pending_out(ctx);
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_ctx_abort(vszip_ctx *ctx) {
    if (!ctx) return VSZIP_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->pending_out.clear();  // the caller is about to release the destinations
    ctx->stage_used = 0;
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_ctx_set_staging(vszip_ctx *ctx, int mode) {
    if (!ctx || mode < 0 || mode > 1) return VSZIP_ERR_ARG;
    if (mode != ctx->staging) {
        VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        flush_pending_out(ctx);
        ctx->staging = mode;
    }
    return VSZIP_OK;
}

VSZIP_EXPORT const char *vszip_last_error(vszip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

VSZIP_EXPORT int vszip_dev_alloc(vszip_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (hipMalloc(dptr, bytes ? bytes : 1) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_dev_alloc_probed(vszip_ctx *ctx, size_t bytes, int tries, vszip_placement_probe probe, void *user, void **dptr, double *best_cost) {
    if (!ctx || !dptr) return VSZIP_ERR_ARG;
    if (tries <= 1 || !probe) {
        if (best_cost) *best_cost = 0.0;
        return vszip_dev_alloc(ctx, bytes, dptr);
    }
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::vector<void *> held;  // every candidate stays allocated until the walk is over: the next one lies elsewhere
    void *best = nullptr;
    double cost = 0.0;
    for (int k = 0; k < tries; ++k) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
            (void)hipGetLastError();  // out of memory: choose among what fitted
            break;
        }
        held.push_back(p);
        const double c = probe(user, p);
        if (!best || c < cost) {
            best = p;
            cost = c;
        }
    }
    for (void *p : held)
        if (p != best) (void)hipFree(p);
    if (!best) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    *dptr = best;
    if (best_cost) *best_cost = cost;
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_dev_free(vszip_ctx *ctx, void *dptr) {
    if (!ctx) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    vszip_bilateral_forget_lut(dptr);
    VSZIP_HIP_CHECK(ctx, hipFree(dptr));
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_dev_memset(vszip_ctx *ctx, void *dptr, int value, size_t bytes) {
    if (!ctx) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_host_alloc_pinned(vszip_ctx *ctx, size_t bytes, void **hptr) {
    if (!ctx || !hptr) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_host_free_pinned(vszip_ctx *ctx, void *hptr) {
    if (!ctx) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipHostFree(hptr));
    return VSZIP_OK;
}

static void copy_rows(void *dst, size_t dpitch, const void *src, size_t spitch, size_t wb, size_t rows) {
    if (dpitch == wb && spitch == wb) {
        memcpy(dst, src, wb * rows);
        return;
    }
    for (size_t y = 0; y < rows; ++y) memcpy(static_cast<char *>(dst) + y * dpitch, static_cast<const char *>(src) + y * spitch, wb);
}

// D2H copies staged through the arena land in the caller's memory here (after the stream drained).
static void flush_pending_out(vszip_ctx *ctx) {
    for (const auto &p : ctx->pending_out) copy_rows(p.dst, p.dpitch, p.src, p.wb, p.wb, p.rows);
    ctx->pending_out.clear();
    ctx->stage_used = 0;
}

// `bytes` of the pinned arena, valid until the next vszip_ctx_sync. A full arena drains the stream
// first (everything staged so far has then been consumed) and grows.
static int stage_take(vszip_ctx *ctx, size_t bytes, char **out) {
    const size_t need = (bytes + 255) & ~size_t(255);
    if (ctx->stage_used + need > ctx->stage_bytes) {
        VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        flush_pending_out(ctx);
        if (need > ctx->stage_bytes) {
            if (ctx->stage) (void)hipHostFree(ctx->stage);
            ctx->stage = nullptr;
            ctx->stage_bytes = 0;
            const size_t want = need * 2 > (size_t(64) << 20) ? need * 2 : (size_t(64) << 20);
            void *p = nullptr;
            if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "hipHostMalloc(%zu) failed", want);
            ctx->stage = static_cast<char *>(p);
            ctx->stage_bytes = want;
        }
    }
    *out = ctx->stage + ctx->stage_used;
    ctx->stage_used += need;
    return VSZIP_OK;
}

static int copy2d(vszip_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t wb, size_t rows, hipMemcpyKind kind) {
    if (!ctx) return VSZIP_ERR_ARG;
    if (wb == 0 || rows == 0) return VSZIP_OK;
    // The calling thread's current device may be another one: the plugin picks the GPU per frame index and a
    // worker thread serves frames of every GPU (pinned allocations and event/stream calls follow the current device).
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->staging == 1 && kind != hipMemcpyDeviceToDevice) {
        char *a = nullptr;
        const int rc = stage_take(ctx, wb * rows, &a);
        if (rc != VSZIP_OK) return rc;
        if (kind == hipMemcpyHostToDevice) {
            copy_rows(a, wb, src, spitch, wb, rows);
            VSZIP_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dpitch, a, wb, wb, rows, kind, ctx->stream));
        } else {
            VSZIP_HIP_CHECK(ctx, hipMemcpy2DAsync(a, wb, src, spitch, wb, rows, kind, ctx->stream));
            ctx->pending_out.push_back({dst, dpitch, a, wb, rows});
        }
        return VSZIP_OK;
    }
    VSZIP_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, wb, rows, kind, ctx->stream));
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_copy_h2d_2d(vszip_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t wb, size_t rows) {
    return copy2d(ctx, dst, dpitch, src, spitch, wb, rows, hipMemcpyHostToDevice);
}
VSZIP_EXPORT int vszip_copy_d2h_2d(vszip_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t wb, size_t rows) {
    return copy2d(ctx, dst, dpitch, src, spitch, wb, rows, hipMemcpyDeviceToHost);
}
VSZIP_EXPORT int vszip_copy_d2d_2d(vszip_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch, size_t wb, size_t rows) {
    return copy2d(ctx, dst, dpitch, src, spitch, wb, rows, hipMemcpyDeviceToDevice);
}

VSZIP_EXPORT int vszip_timer_start(vszip_ctx *ctx) {
    if (!ctx) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_timer_stop_ms(vszip_ctx *ctx, float *ms) {
    if (!ctx || !ms) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    VSZIP_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
    VSZIP_HIP_CHECK(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return VSZIP_OK;
}

// ---- dominant-kernel probe --------------------------------------------------
void vszip_probe_mark(vszip_ctx *ctx) {
    if (ctx->probe_used == ctx->probe_events.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return;
        ctx->probe_events.push_back(e);
    }
    (void)hipEventRecord(ctx->probe_events[ctx->probe_used++], ctx->stream);
}

VSZIP_EXPORT int vszip_probe_enable(vszip_ctx *ctx, int on) {
    if (!ctx) return VSZIP_ERR_ARG;
    ctx->probe_on = on != 0;
    ctx->probe_used = 0;
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_probe_read_each(vszip_ctx *ctx, double *total_ms, int *launches, float *each_ms, int cap) {
    if (!ctx || !total_ms || !launches) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0;
    const size_t pairs = ctx->probe_used / 2;
    for (size_t i = 0; i < pairs; ++i) {
        float ms = 0;
        VSZIP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->probe_events[2 * i], ctx->probe_events[2 * i + 1]));
        tot += ms;
        if (each_ms && (int)i < cap) each_ms[i] = ms;
    }
    *total_ms = tot;
    *launches = (int)pairs;
    ctx->probe_used = 0;
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_probe_read(vszip_ctx *ctx, double *total_ms, int *launches) {
    return vszip_probe_read_each(ctx, total_ms, launches, nullptr, 0);
}
