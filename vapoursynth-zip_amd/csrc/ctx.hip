// Context, device memory, staging copies and timers behind include/vszip_hip.h.
#include <vector>

#include "common.hpp"

#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <atomic>
#include <map>
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

// Value ranges: the integer options by name, every other option is a flag (0 / 1).
struct OptionRange {
    const char *env;
    int lo, hi;
};
const OptionRange kRanges[] = {
    {"VSZIP_SCAN_MODE", 0, 2},         {"VSZIP_PLACEMENT_MIN_MIB", 16, 1 << 20}, {"VSZIP_PLACEMENT_TRIES", 1, 64},  {"VSZIP_RT_ICHAIN_BANDS", 0, 256},
    {"VSZIP_PLACEMENT_BUDGET_MS", 0, 60000},
    {"VSZIP_RT_VSMALL_MAX", 0, 8},     {"VSZIP_RT_VBAND", 0, 1 << 16},           {"VSZIP_RT_VRING_MAXR", 0, 127},   {"VSZIP_RT_GROUP_MB", 0, 1 << 20},
    {"VSZIP_RING_PERIODS", 0, 1 << 16},
};
bool option_in_range(const char *env, int v) {
    for (const OptionRange &r : kRanges)
        if (strcmp(r.env, env) == 0) return v >= r.lo && v <= r.hi;
    return v == 0 || v == 1;
}

// Environment text -> value. Integers as they are; for a flag "" / "0" is off and any other text on - except VSZIP_STAGING, whose historical
// spellings are "1" and "pinned" only (any other text leaves staging off, as before round 4). *ok = false: not a value of this option.
int option_value(const char *env, const char *text, bool *ok) {
    *ok = true;
    if (!text || !*text) return 0;
    char *end = nullptr;
    const long v = strtol(text, &end, 10);
    if (end != text && *end == 0) {
        *ok = v >= INT_MIN && v <= INT_MAX && option_in_range(env, (int)v);
        return (int)v;
    }
    if (strcmp(env, "VSZIP_STAGING") == 0) return strcmp(text, "pinned") == 0 ? 1 : 0;
    for (const OptionRange &r : kRanges)
        if (strcmp(r.env, env) == 0) {  // text where a number is due: ignored
            *ok = false;
            return 0;
        }
    return 1;
}

void options_from_env(vszip_options *o) {
    for (const OptionDesc &d : kOptions) {
        if (!d.field) continue;
        if (const char *e = getenv(d.env)) {
            bool ok;
            const int v = option_value(d.env, e, &ok);
            if (ok) o->*d.field = v;  // (an out-of-range value in the environment keeps the default)
        }
    }
}
}  // namespace

VSZIP_EXPORT int vszip_ctx_set_option(vszip_ctx *ctx, const char *name, int value) {
    if (!ctx || !name) return VSZIP_ERR_ARG;
    for (const OptionDesc &d : kOptions)
        if (strcmp(d.env, name) == 0) {
            if (!d.field) return vszip_set_error(ctx, VSZIP_ERR_UNSUPPORTED, "%s is a development variant: build with -DVSZIP_DEV_VARIANTS", name);
            if (!option_in_range(name, value)) return vszip_set_error(ctx, VSZIP_ERR_ARG, "%s = %d is out of range", name, value);
            if (d.field == &vszip_options::staging) return vszip_ctx_set_staging(ctx, value);  // (drains what is staged first)
            ctx->opt.*d.field = value;
            return VSZIP_OK;
        }
    return vszip_set_error(ctx, VSZIP_ERR_ARG, "unknown option %s", name);
}

VSZIP_EXPORT int vszip_ctx_get_option(vszip_ctx *ctx, const char *name, int *value) {
    if (!ctx || !name || !value) return VSZIP_ERR_ARG;
    if (strcmp(name, "VSZIP_STAT_MINMAX_PREDICTED") == 0) {
        *value = ctx->minmax_predicted;
        return VSZIP_OK;
    }
    if (strcmp(name, "VSZIP_STAT_MINMAX_FALLBACKS") == 0) {  // a counter, not a switch (tests: did a clip's steady state stay on the single sweep?)
        *value = ctx->minmax_fallbacks;
        return VSZIP_OK;
    }
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
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->xpsnr_sums) (void)hipFree(ctx->xpsnr_sums);
    vszip_ssim_release(ctx);
    vszip_chain_release(ctx);
    vszip_planestats_release(ctx);
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

// ---- placed allocations -------------------------------------------------------------------------------------------
// WHERE a resident batch lies in VRAM decides how fast kernels with thousands of concurrent row streams run on it: the
// BoxBlur ring kernel's 64-frame 4K launch takes 545-585 us with its DESTINATION planes in some memory and 600-690 us in
// most, in three classes - a property of the physical memory behind the destination, stable for the life of the
// allocation, nearly independent of the source arena (profiles/r03_placement.md, r05_placement.md). Nothing user space
// can see or request predicts it, but a 2 ms copy in the ring kernel's access shape (placement_probe_kernel) measures it.
// So a request of VSZIP_PLACEMENT_MIN_MIB or more is served from a short search: up to VSZIP_PLACEMENT_TRIES candidate
// allocations of the requested size (all held meanwhile, so each lies elsewhere; never more than a quarter of what
// hipMemGetInfo reports free), each classified; the search ends with the first candidate of the best class, else the
// fastest is kept, and every other candidate is freed before the call returns. Nothing is parked, there is no
// per-context state and vszip_dev_free is hipFree. Round 4's version of this walked up to 64 GiB for up to 3 s, parked
// 24 GiB of fast regions per context and gave up for good after one unlucky walk; round 5 also tried to BUILD
// fast arenas from hipMemCreate pieces instead of searching for them - more often fast than a plain block, never
// reliably, mispredicted by the probe, and at the price of two defects of this runtime's virtual-memory calls
// (profiles/r05_placement.md) - and kept plain allocations.
namespace {

struct PlacedInfo {
    int candidates;
    double probe_bytes_per_s, build_ms;
};
std::mutex &placed_mu() {
    static std::mutex *m = new std::mutex();
    return *m;
}
std::map<void *, PlacedInfo> &placed_map() {  // process-wide: what vszip_dev_arena_info reports (any context may free any allocation)
    static auto *m = new std::map<void *, PlacedInfo>();
    return *m;
}
std::atomic<int> g_placed_count{0};

double probe_region(vszip_ctx *ctx, void *ptr, size_t bytes, const void *from);

// placement_probe_kernel's rate on the devices seen: 5.55-5.9 TB/s <-> the real launch at 560-585 us with its destination there (0.68-0.71),
// 5.2-5.35 TB/s <-> 594-608 us, 4.6-5.1 TB/s <-> 630-695 us (profiles/r04_placement_probe_calibration.txt).
// Round 6: the search is bounded three ways, none of them a property of ONE device model:
//  - wall clock: VSZIP_PLACEMENT_BUDGET_MS (default 300) — a fresh allocation costs 3 ms to probe on an idle MI355X and 50-170 ms where
//    the driver clears memory first, so on a slow or shared device the search looks at the few candidates the budget pays for;
//  - early exit RELATIVE to what this process has seen on this device: a candidate within 1.5 % of the best probe rate on record
//    ends the search (the absolute 5.55 TB/s of the calibration above still ends a device's FIRST search early, and is never required);
//  - a verdict per device: a search that looked at four candidates or more and found them all within 3 % of each other (one class of
//    memory, or a device so busy that the probe measures its neighbours) marks the device "flat" for this process — later requests
//    take a plain hipMalloc. Two flat searches in a row are needed before the verdict sticks; one search that finds a spread clears it.
constexpr double kBestBytesPerSec = 5.55e12;
constexpr double kNearBest = 0.985, kFlatSpread = 0.03;
constexpr int kMaxDevices = 64;
struct DevicePlacement {
    double best_rate = 0.0;  // best probe rate any search of this process saw on the device
    int flat_searches = 0;   // consecutive searches whose candidates were all one class
};
DevicePlacement g_dev_placement[kMaxDevices];  // guarded by placed_mu()

int placed_alloc(vszip_ctx *ctx, size_t bytes, void **dptr) {
    const auto t_start = std::chrono::steady_clock::now();
    auto elapsed_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    const int dev = ctx->device >= 0 && ctx->device < kMaxDevices ? ctx->device : 0;
    double best_on_record = 0.0;
    bool flat = false;
    {
        std::lock_guard<std::mutex> lk(placed_mu());
        best_on_record = g_dev_placement[dev].best_rate;
        flat = g_dev_placement[dev].flat_searches >= 2;
    }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        free_b = bytes;
    }
    int tries = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, ctx->opt.placement_tries), free_b / 4 / std::max<size_t>(bytes, 1)));
    if (flat) tries = 1;  // (this process found nothing to choose between on this device)
    const double budget_ms = (double)std::max(0, ctx->opt.placement_budget_ms);
    if (tries > 1) VSZIP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<std::pair<void *, double>> cand;  // pointer, probe rate
    int best = -1;
    double worst_rate = 0.0;
    for (int k = 0; k < tries; ++k) {
        if (k > 0 && elapsed_ms() >= budget_ms) break;  // the wall-clock bound: keep the fastest so far
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;  // the device is fuller than it said: choose among what exists
        }
        double rate = 0.0;
        if (tries > 1) {
            double c = probe_region(ctx, p, bytes, nullptr);
            // the first candidate also brings the clocks up: an idle device runs its first milliseconds slower, and candidates must be compared
            // at one clock - repeat until two measurements in a row agree within 0.5 % (at most ~60 ms, and never past the budget)
            for (int i = 0; k == 0 && i < 20 && c > 0 && elapsed_ms() < budget_ms; ++i) {
                const double c2 = probe_region(ctx, p, bytes, nullptr);
                const bool steady = c2 > 0 && std::fabs(c2 - c) <= 0.005 * c;
                c = c2;
                if (steady) break;
            }
            rate = c > 0 ? 1.0 / c : 0.0;
        }
        cand.emplace_back(p, rate);
        if (best < 0 || rate > cand[best].second) best = (int)cand.size() - 1;
        if (rate > 0 && (worst_rate == 0.0 || rate < worst_rate)) worst_rate = rate;
        if (rate > 0 && best_on_record > 0 ? rate >= kNearBest * best_on_record : rate >= kBestBytesPerSec) break;
    }
    if (cand.empty()) return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    for (int k = 0; k < (int)cand.size(); ++k)
        if (k != best) (void)hipFree(cand[k].first);
    {
        std::lock_guard<std::mutex> lk(placed_mu());
        placed_map()[cand[best].first] = {(int)cand.size(), cand[best].second, elapsed_ms()};
        g_placed_count.fetch_add(1);
        DevicePlacement &dp = g_dev_placement[dev];
        if (cand[best].second > dp.best_rate) dp.best_rate = cand[best].second;
        if (cand.size() >= 4 && worst_rate > 0) {
            if (cand[best].second - worst_rate <= kFlatSpread * cand[best].second)
                ++dp.flat_searches;
            else
                dp.flat_searches = 0;
        }
    }
    *dptr = cand[best].first;
    return VSZIP_OK;
}

void placed_forget(void *dptr) {
    if (g_placed_count.load() == 0) return;
    std::lock_guard<std::mutex> lk(placed_mu());
    if (placed_map().erase(dptr)) g_placed_count.fetch_sub(1);
}

typedef unsigned int pv4u __attribute__((ext_vector_type(4)));

// One wave per (band, tile): moves a 960-byte column tile down `band_rows` rows of a 7680-byte-pitch view of the region,
// reading its own band and writing the band half a region away (every tile is read once and written once per launch) - the ring kernel's
// access shape without its arithmetic. `from` != nullptr: the tiles are read from that region instead - the pair a batch will be.
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

// seconds per byte moved by the probe on the first 2 GiB of [ptr, ptr + bytes); < 0: could not measure
double probe_region(vszip_ctx *ctx, void *ptr, size_t bytes, const void *from) {
    const size_t span = std::min(bytes, (size_t)2 << 30);
    const long long rows = (long long)(span / 7680);
    // ~3072 streams like the ring kernel's launch: 384 bands x 8 tiles, bands of 64 ... 540 rows
    const int band_rows = (int)std::min<long long>(540, std::max<long long>(64, rows / 384));
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

}  // namespace

hipError_t vszip_hip_malloc(vszip_ctx *, void **p, size_t bytes) { return hipMalloc(p, bytes); }

VSZIP_EXPORT int vszip_dev_alloc(vszip_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->opt.placement && bytes >= ((size_t)std::max(16, ctx->opt.placement_min_mib) << 20)) return placed_alloc(ctx, bytes, dptr);
    if (hipMalloc(dptr, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        return vszip_set_error(ctx, VSZIP_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    }
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_dev_arena_info(vszip_ctx *ctx, const void *dptr, int *candidates, double *probe_bytes_per_second, double *search_ms) {
    if (!ctx) return VSZIP_ERR_ARG;
    PlacedInfo pi = {0, 0.0, 0.0};
    if (g_placed_count.load() != 0) {
        std::lock_guard<std::mutex> lk(placed_mu());
        auto it = placed_map().find(const_cast<void *>(dptr));
        if (it != placed_map().end()) pi = it->second;
    }
    if (candidates) *candidates = pi.candidates;
    if (probe_bytes_per_second) *probe_bytes_per_second = pi.probe_bytes_per_s;
    if (search_ms) *search_ms = pi.build_ms;
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_dev_probe_region(vszip_ctx *ctx, void *dptr, size_t bytes, const void *from, double *bytes_per_second) {
    if (!ctx || !dptr || !bytes_per_second) return VSZIP_ERR_ARG;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const double c = probe_region(ctx, dptr, bytes, from);
    if (c <= 0) return vszip_set_error(ctx, VSZIP_ERR_ARG, "region of %zu bytes is too small to probe", bytes);
    *bytes_per_second = 1.0 / c;
    return VSZIP_OK;
}

VSZIP_EXPORT int vszip_dev_free(vszip_ctx *ctx, void *dptr) {
    if (!ctx) return VSZIP_ERR_ARG;
    if (!dptr) return VSZIP_OK;
    VSZIP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    vszip_bilateral_forget_lut(dptr);
    placed_forget(dptr);
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
