// Context, device memory, staging copies and timers behind include/vszip_hip.h.
#include <vector>

#include "common.hpp"

#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <set>

namespace {
size_t trim_parked(vszip_ctx *ctx);
}

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
    if (vszip_hip_malloc(ctx, &ctx->scratch, want) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "scratch allocation of %zu bytes failed", want);
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
    if (vszip_hip_malloc(ctx, &ctx->scalars_dev, want) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "scalar buffer allocation failed");
    if (hipHostMalloc(&ctx->scalars_host, want, hipHostMallocDefault) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "pinned scalar buffer allocation failed");
    ctx->scalars_bytes = want;
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_abi_version(void) { return VSZIP_ABI_VERSION; }

// ---- options (csrc/options.inc): the environment is read HERE, once per context, and nowhere else in csrc/ ----
namespace {
struct OptionDesc {
    const char *env;
    int vszip_options::*field;  // nullptr: a development variant this build does not contain
};
const OptionDesc kOptions[] = {
#define VSZIP_OPT(f, env, def) {env, &vszip_options::f},
#ifdef VSZIP_DEV_VARIANTS
#define VSZIP_DEV_OPT(f, env, def) {env, &vszip_options::f},
#else
#define VSZIP_DEV_OPT(f, env, def) {env, nullptr},
#endif
#include "options.inc"
#undef VSZIP_OPT
#undef VSZIP_DEV_OPT
};

// flags: "" / "0" off, any other text on; integers as they are ("pinned" is VSZIP_STAGING's historical spelling of 1)
int option_value(const char *text) {
    if (!text || !*text) return 0;
    char *end = nullptr;
    const long v = strtol(text, &end, 10);
    if (end != text && *end == 0) return (int)v;
    return 1;
}

void options_from_env(vszip_options *o) {
    for (const OptionDesc &d : kOptions) {
        if (!d.field) continue;
        if (const char *e = getenv(d.env)) o->*d.field = option_value(e);
    }
}
}  // namespace

VSZIP_EXPORT int vszip_ctx_set_option(vszip_ctx *ctx, const char *name, int value) {
    if (!ctx || !name) return VSZIP_ERR_ARG;
    for (const OptionDesc &d : kOptions)
        if (strcmp(d.env, name) == 0) {
            if (!d.field) return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "%s is a development variant: build with -DVSZIP_DEV_VARIANTS", name);
            if (d.field == &vszip_options::staging) return vszip_ctx_set_staging(ctx, value);  // (drains what is staged first)
            ctx->opt.*d.field = value;
            return VSZIP_OK;
        }
    return vszip_set_error(ctx, VSZIP_ERR_ARG, "unknown option %s", name);
}

VSZIP_EXPORT int vszip_ctx_get_option(vszip_ctx *ctx, const char *name, int *value) {
    if (!ctx || !name || !value) return VSZIP_ERR_ARG;
    for (const OptionDesc &d : kOptions)
        if (strcmp(d.env, name) == 0) {
            if (!d.field) return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "%s is a development variant: build with -DVSZIP_DEV_VARIANTS", name);
            *value = ctx->opt.*d.field;
            return VSZIP_OK;
        }
    return vszip_set_error(ctx, VSZIP_ERR_ARG, "unknown option %s", name);
}

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
    options_from_env(&c->opt);
    *out = c;
    return VSZIP_OK;
}

VSZIP_EXPORT void vszip_ctx_destroy(vszip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    trim_parked(ctx);
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
    if (ctx->probe_ev0) (void)hipEventDestroy(ctx->probe_ev0);
    if (ctx->probe_ev1) (void)hipEventDestroy(ctx->probe_ev1);
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

// ---- placed allocations -------------------------------------------------------------------------------------
// WHERE a resident batch lies in VRAM decides how fast kernels with many concurrent row streams run on it: the BoxBlur
// ring kernel's 64-frame 4K launch takes 545-565 us with its planes in some physical regions and 640-680 us in others
// (a property of the physical memory, in runs of ~10 GiB; no layout user space can choose changes it, streaming
// kernels do not see it: profiles/r03_placement.md). Round 3 searched for a fast region in bench.py; since round 4 the
// allocator does it for every caller: a request of VSZIP_PLACEMENT_MIN_MIB or more is served from a walk over candidate
// allocations of that size (all held meanwhile, so each lies elsewhere), each classified by placement_probe_kernel — the
// ring kernel's access shape without its arithmetic — in about 2 ms. The fastest is returned, the next best stay PARKED
// (allocated, classified, not in use) up to VSZIP_PLACEMENT_PARK_GIB for the requests that follow (a batch is a source
// and a destination arena), the rest are freed. vszip_dev_free parks a placed region again instead of freeing it.
// Parked memory goes back to the driver when an allocation fails, in vszip_dev_trim and in vszip_ctx_destroy.
// Bounded: the walk holds at most VSZIP_PLACEMENT_WALK_GIB and stops as soon as enough regions of the fast class have
// turned up; VSZIP_PLACEMENT=0 turns all of it off (plain hipMalloc).
namespace {

typedef unsigned int pv4u __attribute__((ext_vector_type(4)));

// One wave per (band, tile): moves a 960-byte column tile down `band_rows` rows of a 7680-byte-pitch view of the region,
// reading its own band and writing the band half a region away (every tile is read once and written once per launch).
// `from` != nullptr: the tiles are read from that region instead (a source arena allocated earlier) — the pair a batch will be.
__global__ __launch_bounds__(64) void placement_probe_kernel(char *base, const char *from, int bands, int band_rows) {
    constexpr long long kPitch = 7680;
    const int b = blockIdx.x >> 3, t = blockIdx.x & 7, lane = threadIdx.x;
    if (lane >= 60) return;
    const int wb = from ? b : (b + bands / 2 < bands ? b + bands / 2 : b + bands / 2 - bands);
    const char *sp = (from ? from : base) + (long long)b * band_rows * kPitch + t * 960 + lane * 16;
    char *dp = base + (long long)wb * band_rows * kPitch + t * 960 + lane * 16;
    pv4u a = *reinterpret_cast<const pv4u *>(sp);
    for (int r = 0; r < band_rows; ++r) {
        const pv4u v = a;
        if (r + 1 < band_rows) a = *reinterpret_cast<const pv4u *>(sp + (long long)(r + 1) * kPitch);
        __builtin_nontemporal_store(v, reinterpret_cast<pv4u *>(dp + (long long)r * kPitch));
    }
}

constexpr size_t kProbeSpan = (size_t)2 << 30;  // at most this much of a region is probed (its first 2 GiB)
// placement_probe_kernel's rate sorts regions into three classes on the devices seen (profiles/r04_placement_probe_calibration.txt, gpurun_out/r4_pool_probe*.txt):
// 5.55-5.9 TB/s <-> the real launch at 568-585 us with its destination there (0.68-0.70), 5.2-5.35 TB/s <-> 594-608 us (0.65-0.67), 4.6-5.1 TB/s <-> 630-695 us.
constexpr double kFastBytesPerSec = 5.2e12;      // good enough to keep
constexpr double kBestBytesPerSec = 5.55e12;     // ends a walk at once
constexpr size_t kPlacedGranule = (size_t)64 << 20;  // placed requests are rounded up to this: arenas of nearly equal size share parked regions

// seconds per byte moved by the probe on [ptr, ptr + bytes); < 0: could not measure
double probe_region(vszip_ctx *ctx, void *ptr, size_t bytes, const void *from = nullptr, size_t from_bytes = 0) {
    if (from && from_bytes < bytes) bytes = from_bytes;
    const size_t span = std::min(bytes, kProbeSpan);
    const long long rows = (long long)(span / 7680);
    // ~3072 streams like the ring kernel's launch: 384 bands x 8 tiles, bands of 64 ... 540 rows
    int band_rows = (int)std::min<long long>(540, std::max<long long>(64, rows / 384));
    const int bands = (int)(rows / band_rows);
    if (bands < 2) return -1.0;
    const dim3 grid(bands * 8);
    const char *fr = static_cast<const char *>(from);
    if (!ctx->probe_ev0 && (hipEventCreate(&ctx->probe_ev0) != hipSuccess || hipEventCreate(&ctx->probe_ev1) != hipSuccess)) {  // (events of their own: a caller's vszip_timer_* pair may be open)
        (void)hipGetLastError();
        return -1.0;
    }
    hipLaunchKernelGGL(placement_probe_kernel, grid, dim3(64), 0, ctx->stream, static_cast<char *>(ptr), fr, bands, band_rows);
    if (hipEventRecord(ctx->probe_ev0, ctx->stream) != hipSuccess) return -1.0;
    const int n = 2;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(placement_probe_kernel, grid, dim3(64), 0, ctx->stream, static_cast<char *>(ptr), fr, bands, band_rows);
    float ms = 0;
    if (hipEventRecord(ctx->probe_ev1, ctx->stream) != hipSuccess || hipEventSynchronize(ctx->probe_ev1) != hipSuccess || hipEventElapsedTime(&ms, ctx->probe_ev0, ctx->probe_ev1) != hipSuccess) {
        (void)hipGetLastError();
        return -1.0;
    }
    const double moved = 2.0 * n * (double)bands * band_rows * 8 * 960;
    return ms * 1e-3 / moved;
}

void park_region(vszip_ctx *ctx, const vszip_ctx::Region &r) {
    ctx->parked.push_back(r);
    // over the cap: the slowest go back to the driver
    const size_t cap = (size_t)std::max(0, ctx->opt.placement_park_gib) << 30;
    size_t total = 0;
    for (const auto &q : ctx->parked) total += q.bytes;
    while (total > cap && !ctx->parked.empty()) {
        size_t worst = 0;
        for (size_t i = 1; i < ctx->parked.size(); ++i)
            if (ctx->parked[i].cost > ctx->parked[worst].cost) worst = i;
        total -= ctx->parked[worst].bytes;
        (void)hipFree(ctx->parked[worst].ptr);
        ctx->parked.erase(ctx->parked.begin() + worst);
    }
}

size_t trim_parked(vszip_ctx *ctx) {
    size_t freed = 0;
    for (const auto &q : ctx->parked) {
        (void)hipFree(q.ptr);
        freed += q.bytes;
    }
    ctx->parked.clear();
    return freed;
}

}  // namespace

// hipMalloc that gives parked memory back to the driver before it reports failure (every allocation of the library)
hipError_t vszip_hip_malloc(vszip_ctx *ctx, void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && !ctx->parked.empty()) {
        (void)hipGetLastError();
        trim_parked(ctx);
        e = hipMalloc(p, bytes);
    }
    return e;
}

namespace {

int placed_alloc(vszip_ctx *ctx, size_t bytes, void **dptr) {
    bytes = (bytes + kPlacedGranule - 1) / kPlacedGranule * kPlacedGranule;
    // 1: a parked region of a fitting size (the fastest one)
    int pick = -1;
    for (size_t i = 0; i < ctx->parked.size(); ++i) {
        const auto &r = ctx->parked[i];
        if (r.bytes >= bytes && r.bytes / 2 <= bytes && (pick < 0 || r.cost < ctx->parked[pick].cost)) pick = (int)i;
    }
    const double fast_cost = 1.0 / kFastBytesPerSec, best_cost = 1.0 / kBestBytesPerSec;
    if (pick >= 0 && (ctx->parked[pick].cost <= best_cost || (ctx->placement_exhausted && ctx->parked[pick].cost <= fast_cost))) {
        ctx->placed.push_back(ctx->parked[pick]);
        *dptr = ctx->parked[pick].ptr;
        ctx->parked.erase(ctx->parked.begin() + pick);
        return VSZIP_OK;
    }
    if (ctx->placement_exhausted) {  // an earlier walk used its whole budget without meeting a fast region: this device's next tens of GiB are slow, no more searching
        if (pick >= 0) {
            ctx->placed.push_back(ctx->parked[pick]);
            *dptr = ctx->parked[pick].ptr;
            ctx->parked.erase(ctx->parked.begin() + pick);
            return VSZIP_OK;
        }
        if (vszip_hip_malloc(ctx, dptr, bytes) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
        return VSZIP_OK;
    }
    // 2: walk. Everything stays allocated until the walk ends; it ends with the first region of the best class (or eight candidates after the first
    // of the middle class), when the walk's budget (bytes held, wall time: memory the device has not handed out before is cleared on first use,
    // 50-170 ms per candidate of this size) is used, or when the device is full. Slow regions freed here come back first in the next walk and cost
    // 2.5 ms each then.
    VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t budget = (size_t)std::max(1, ctx->opt.placement_walk_gib) << 30;
    const int max_cand = (int)std::min<size_t>(64, std::max<size_t>(1, budget / bytes));
    const auto t_start = std::chrono::steady_clock::now();
    std::vector<vszip_ctx::Region> cand;
    int since_good = -1;  // candidates probed since the first one of the middle class
    if (pick >= 0) {  // a parked region of the middle or the slow class competes with what the walk finds
        cand.push_back(ctx->parked[pick]);
        if (ctx->parked[pick].cost <= fast_cost) since_good = 0;
        ctx->parked.erase(ctx->parked.begin() + pick);
    }
    bool cut_short = false;
    for (int k = 0; k < max_cand; ++k) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        double c = probe_region(ctx, p, bytes);
        if (c < 0) c = 1.0;  // unmeasurable: last choice
        cand.push_back({p, bytes, c});
        if (c <= best_cost) break;                           // the best class: done
        if (c <= fast_cost && since_good < 0) since_good = 0;  // the middle class: good enough, but look at eight more for the best
        if (since_good >= 0 && ++since_good > 8) break;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        if (ms > (double)std::max(1, ctx->opt.placement_walk_ms)) {
            cut_short = true;
            break;
        }
    }
    if (cand.empty()) {
        void *p = nullptr;
        if (vszip_hip_malloc(ctx, &p, bytes) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
        *dptr = p;
        return VSZIP_OK;
    }
    std::sort(cand.begin(), cand.end(), [](const vszip_ctx::Region &a, const vszip_ctx::Region &b) { return a.cost < b.cost; });
    *dptr = cand[0].ptr;
    ctx->placed.push_back(cand[0]);
    ctx->placement_walks += 1;
    ctx->placement_probed += (int)cand.size();
    ctx->placement_last_walk_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    // No region worth keeping: this context stops searching - unless the clock cut the walk short after a few candidates (memory that has to be
    // cleared costs up to 170 ms a candidate), then the next large request may look further, up to 32 slow candidates in all.
    if (cand[0].cost > fast_cost) {
        ctx->placement_slow_seen += (int)cand.size();
        if (!cut_short || ctx->placement_slow_seen >= 32) ctx->placement_exhausted = true;
    }
    // the slow ones are freed FIRST (in one go, after the walk), then the fast ones are parked
    for (size_t i = 1; i < cand.size(); ++i)
        if (cand[i].cost > fast_cost) (void)hipFree(cand[i].ptr);
    for (size_t i = 1; i < cand.size(); ++i)
        if (cand[i].cost <= fast_cost) park_region(ctx, cand[i]);
    return VSZIP_OK;
}

}  // namespace

VSZIP_EXPORT int vszip_dev_alloc(vszip_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->opt.placement && bytes >= ((size_t)std::max(1, ctx->opt.placement_min_mib) << 20)) return placed_alloc(ctx, bytes, dptr);
    if (vszip_hip_malloc(ctx, dptr, bytes ? bytes : 1) != hipSuccess) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_dev_trim(vszip_ctx *ctx, size_t *freed_bytes) {
    if (!ctx) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t f = trim_parked(ctx);
    if (freed_bytes) *freed_bytes = f;
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_dev_placement_info(vszip_ctx *ctx, const void *dptr, double *bytes_per_second, int *parked_regions, size_t *parked_bytes, int *walks, int *probed, double *last_walk_ms,
                                          int *exhausted) {
    if (!ctx) return VSZIP_ERR_ARG;
    if (bytes_per_second) {
        *bytes_per_second = 0.0;
        for (const auto &r : ctx->placed)
            if (r.ptr == dptr && r.cost > 0) *bytes_per_second = 1.0 / r.cost;
    }
    if (parked_regions) *parked_regions = (int)ctx->parked.size();
    if (parked_bytes) {
        *parked_bytes = 0;
        for (const auto &r : ctx->parked) *parked_bytes += r.bytes;
    }
    if (walks) *walks = ctx->placement_walks;
    if (probed) *probed = ctx->placement_probed;
    if (last_walk_ms) *last_walk_ms = ctx->placement_last_walk_ms;
    if (exhausted) *exhausted = ctx->placement_exhausted ? 1 : 0;
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_dev_probe_region(vszip_ctx *ctx, void *dptr, size_t bytes, const void *from, double *bytes_per_second) {
    if (!ctx || !dptr || !bytes_per_second) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double c = probe_region(ctx, dptr, bytes, from, bytes);
    if (c <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "region of %zu bytes is too small to probe", bytes);
    *bytes_per_second = 1.0 / c;
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
    for (size_t i = 0; i < ctx->placed.size(); ++i)
        if (ctx->placed[i].ptr == dptr) {  // a classified region: kept for the next request of its size
            const vszip_ctx::Region r = ctx->placed[i];
            ctx->placed.erase(ctx->placed.begin() + i);
            VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // (like hipFree: nothing in flight may still use it when it is handed out again)
            if (ctx->opt.placement) {
                park_region(ctx, r);
                return VSZIP_OK;
            }
            break;
        }
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
