// Round 5, third step: WHICH way of assembling an arena from hipMemCreate pieces is fast on every
// device? Each policy builds `trials` destination arenas (all held, so every trial lies elsewhere) and times the ring
// kernel's access shape on them (tools/vmm/stripe_probe.hip); min / median / max per policy. Also: do the runtime's
// copies (hipMemcpy2DAsync H2D / D2H / D2D, hipMemsetAsync) accept ranges that span several mapped handles?
//   hipcc --offload-arch=gfx950 -O3 -o tools/vmm/stripe_policies.bin tools/vmm/stripe_policies.hip && ./tools/vmm/stripe_policies.bin [trials=10]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

struct Stream {
    long long src, dst;
    int stride, rows;
};
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void ring_copy(const char *sbase, char *dbase, const Stream *st) {
    const Stream s = st[blockIdx.x];
    const int lane = threadIdx.x;
    if (lane >= 60) return;
    const char *sp = sbase + s.src + lane * 16;
    char *dp = dbase + s.dst + lane * 16;
    v4u a = *reinterpret_cast<const v4u *>(sp);
    for (int r = 0; r < s.rows; ++r) {
        v4u b = a;
        if (r + 1 < s.rows) a = *reinterpret_cast<const v4u *>(sp + (size_t)(r + 1) * s.stride);
        __builtin_nontemporal_store(b, reinterpret_cast<v4u *>(dp + (size_t)r * s.stride));
    }
}

static const long long MiB = 1 << 20;
typedef hipMemGenericAllocationHandle_t Handle;

int main(int argc, char **argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 10;
    const int frames = 64;
    struct Pl { int stride, h, tiles, bands; };
    const Pl pls[3] = {{7680, 2160, 8, 4}, {3840, 1080, 4, 2}, {3840, 1080, 4, 2}};
    std::vector<Stream> hs;
    long long total = 0;
    for (int k = 0; k < frames * 3; ++k) {
        const Pl &p = pls[k % 3];
        total = (total + 2 * MiB - 1) / (2 * MiB) * (2 * MiB);
        const int band_rows = p.h / p.bands;
        for (int b = 0; b < p.bands; ++b)
            for (int t = 0; t < p.tiles; ++t) hs.push_back({total + (long long)b * band_rows * p.stride + t * 960, total + (long long)b * band_rows * p.stride + t * 960, p.stride, band_rows});
        total += (long long)p.stride * p.h;
    }
    const long long A = (total + 256 * MiB - 1) / (256 * MiB) * (256 * MiB);
    const int nstreams = (int)hs.size();
    Stream *dstreams;
    CK(hipMalloc(&dstreams, sizeof(Stream) * nstreams));
    CK(hipMemcpy(dstreams, hs.data(), sizeof(Stream) * nstreams, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    char *src;
    CK(hipMalloc(&src, A));
    CK(hipMemset(src, 1, A));
    auto time_us = [&](const char *s, char *dst) {
        hipLaunchKernelGGL(ring_copy, dim3(nstreams), dim3(64), 0, 0, s, dst, dstreams);
        CK(hipEventRecord(e0));
        const int n = 6;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(ring_copy, dim3(nstreams), dim3(64), 0, 0, s, dst, dstreams);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / n;
    };
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t fr, tot;
    CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f of %.1f GiB; arena %lld MiB; %d trials per policy\n", fr / 1073741824.0, tot / 1073741824.0, A / MiB, trials);

    auto report = [&](const char *name, std::vector<double> v) {
        std::vector<double> s = v;
        std::sort(s.begin(), s.end());
        printf("%-58s min %4.0f med %4.0f max %4.0f |", name, s.front(), s[s.size() / 2], s.back());
        for (double x : v) printf(" %3.0f", x);
        printf("\n");
        fflush(stdout);
    };
    auto map_arena = [&](const std::vector<Handle> &hv, long long C) {
        void *p = nullptr;
        CK(hipMemAddressReserve(&p, A, 2 * MiB, nullptr, 0));
        for (size_t i = 0; i < hv.size(); ++i) CK(hipMemMap((char *)p + i * C, C, 0, hv[i], 0));
        CK(hipMemSetAccess(p, A, &acc, 1));
        return (char *)p;
    };
    // P0: plain hipMalloc
    {
        std::vector<double> v;
        for (int t = 0; t < trials; ++t) {
            char *d;
            CK(hipMalloc(&d, A));
            v.push_back(time_us(src, d));
        }
        report("P0 plain hipMalloc", v);
    }
    // Q: a POOL of `pool_gib` in handles of C created back to back (same-size blocks come out of the driver's buddy allocator in address
    // order; blocks of mixed sizes do not: a spacer of another size lands elsewhere and the next piece fills the hole beside the last one),
    // the arena's pieces picked evenly spaced (or at random) from it, the rest released.
    struct QP { long long C; int pool_gib; int random; };
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    int pi = 0;
    for (QP q : {QP{64 * MiB, 16, 0}, QP{64 * MiB, 32, 0}, QP{256 * MiB, 32, 0}, QP{256 * MiB, 16, 0}, QP{128 * MiB, 16, 0}, QP{64 * MiB, 8, 0}, QP{64 * MiB, 12, 0}, QP{64 * MiB, 24, 0}, QP{64 * MiB, 32, 1}, QP{64 * MiB, 64, 0}, QP{16 * MiB, 32, 0}, QP{64 * MiB, 96, 0}}) {
        if (only >= 0 && pi++ != only) continue;
        std::vector<double> v;
        double ms_build = 0;
        unsigned rng = 12345;
        for (int t = 0; t < trials; ++t) {
            const auto t0 = std::chrono::steady_clock::now();
            const size_t n = A / q.C, np = (size_t)(q.pool_gib * 1024 * MiB / q.C);
            std::vector<Handle> pool(np), hv;
            size_t made = 0;
            for (; made < np; ++made)
                if (hipMemCreate(&pool[made], q.C, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            std::vector<char> used(made, 0);
            for (size_t i = 0; i < n; ++i) {
                rng = rng * 1664525u + 1013904223u;
                size_t k = q.random ? (rng >> 8) % made : (size_t)(((double)i + (double)(rng >> 8) / (1 << 24)) * made / n);
                while (used[k % made]) ++k;
                used[k % made] = 1;
                hv.push_back(pool[k % made]);
            }
            for (size_t i = 0; i < made; ++i)
                if (!used[i]) CK(hipMemRelease(pool[i]));
            char *d = map_arena(hv, q.C);
            ms_build += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            v.push_back(time_us(src, d));
        }
        char name[128];
        snprintf(name, sizeof name, "Q pieces of %3lld MiB, %s of a %2d GiB pool (build %.0f ms)", q.C / MiB, q.random ? "random" : "evenly spaced", q.pool_gib, ms_build / trials);
        report(name, v);
    }
    return 0;
    // copies across handle boundaries
    printf("copies on an arena of 2 / 16 / 64 MiB handles (a 16.6 MB plane that spans several):\n");
    for (long long C : {2 * MiB, 16 * MiB, 64 * MiB}) {
        std::vector<Handle> hv(A / C);
        for (auto &h : hv) CK(hipMemCreate(&h, C, &prop, 0));
        char *d = map_arena(hv, C);
        char *host;
        CK(hipHostMalloc(&host, 7680 * 2160, 0));
        char *off = d + C - 3 * MiB - 256;  // starts 3 MiB before a boundary
        printf("  C=%2lld MiB: memset %s", C / MiB, hipGetErrorString(hipMemsetAsync(off, 0, 7680 * 2160, 0)));
        printf(", 2D H2D %s", hipGetErrorString(hipMemcpy2DAsync(off, 7680, host, 7680, 7680, 2160, hipMemcpyHostToDevice, 0)));
        printf(", 2D D2H %s", hipGetErrorString(hipMemcpy2DAsync(host, 7680, off, 7680, 7680, 2160, hipMemcpyDeviceToHost, 0)));
        printf(", 2D D2D %s", hipGetErrorString(hipMemcpy2DAsync(off, 7680, src, 7680, 7680, 2160, hipMemcpyDeviceToDevice, 0)));
        printf(", 2D H2D pitch 8192 %s", hipGetErrorString(hipMemcpy2DAsync(off, 8192, host, 7680, 7680, 2000, hipMemcpyHostToDevice, 0)));
        printf(", 1D H2D %s", hipGetErrorString(hipMemcpyAsync(off, host, 7680 * 2160, hipMemcpyHostToDevice, 0)));
        printf(", 1D D2H %s", hipGetErrorString(hipMemcpyAsync(host, off, 7680 * 2160, hipMemcpyDeviceToHost, 0)));
        printf(", sync %s\n", hipGetErrorString(hipDeviceSynchronize()));
        (void)hipGetLastError();
    }
    return 0;
}
