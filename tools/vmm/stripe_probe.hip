// Round 5, bounded experiment: does an arena DELIBERATELY striped over far-apart physical regions
// run the BoxBlur ring kernel's access shape at the fast regions' rate everywhere?
//
// Physical memory comes from hipMemCreate in K groups with a large spacer handle between consecutive groups (spacers
// are released once every group exists), so the groups lie >= `spacer` GiB apart if the driver hands out physical
// memory in order. Each group holds: one handle of the whole arena size (the control: an arena inside ONE region),
// and A/K bytes' worth of handles of 2, 16 and 64 MiB. Destination arenas are then built with
// hipMemAddressReserve + hipMemMap:  (a) whole, from group g;  (b) striped: chunk i of the arena from group i mod K.
// The kernel is tools/placement_lattice.hip's ring_copy (3072 single-wave streams of 960-byte row segments over a
// 64-frame 4K YUV420P16 batch, 3.185 GB per launch) - it reproduces the real launch's classes (profiles/r03_placement.md).
//   hipcc --offload-arch=gfx950 -O3 -o tools/vmm/stripe_probe.bin tools/vmm/stripe_probe.hip
//   ./tools/vmm/stripe_probe.bin [K=4] [spacer_GiB=30] [plain_arenas=16]
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
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 4;
    const long long spacer = (argc > 2 ? atoll(argv[2]) : 30) << 30;
    const int nplain = argc > 3 ? atoi(argv[3]) : 16;
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
    const long long A = (total + 64 * MiB * K - 1) / (64 * MiB * K) * (64 * MiB * K);  // whole 64 MiB chunks per group
    const int nstreams = (int)hs.size();
    Stream *dstreams;
    CK(hipMalloc(&dstreams, sizeof(Stream) * nstreams));
    CK(hipMemcpy(dstreams, hs.data(), sizeof(Stream) * nstreams, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_us = [&](const char *src, char *dst) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(ring_copy, dim3(nstreams), dim3(64), 0, 0, src, dst, dstreams);
        CK(hipEventRecord(e0));
        const int n = 8;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(ring_copy, dim3(nstreams), dim3(64), 0, 0, src, dst, dstreams);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / n;
    };
    size_t fr, tot;
    CK(hipMemGetInfo(&fr, &tot));
    printf("device free %.1f GiB of %.1f; arena %.3f GiB (%lld MiB), %d streams, 3.185 GB per launch; K=%d spacer %lld GiB\n", fr / 1073741824.0, tot / 1073741824.0, A / 1073741824.0, A / MiB,
           nstreams, K, spacer >> 30);

    // 0: a plain source arena and the plain walk (what hipMalloc gives), all held, then freed
    char *src;
    CK(hipMalloc(&src, A));
    CK(hipMemset(src, 1, A));
    {
        std::vector<char *> held;
        printf("plain hipMalloc walk (us):");
        for (int a = 0; a < nplain; ++a) {
            char *d;
            if (hipMalloc(&d, A) != hipSuccess) break;
            held.push_back(d);
            printf(" %.0f", time_us(src, d));
            fflush(stdout);
        }
        printf("\n");
        for (char *d : held) CK(hipFree(d));
    }

    // 1: physical groups
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("VMM granularity %zu\n", gran);
    const long long chunkC[3] = {2 * MiB, 16 * MiB, 64 * MiB};
    hipMemAccessDesc acc0 = {};
    acc0.location.type = hipMemLocationTypeDevice;
    acc0.location.id = 0;
    acc0.flags = hipMemAccessFlagsProtReadWrite;
    // 0b: what an allocator could do with NO walk: the arena as handles of C MiB created back to back, mapped in order
    // (three rounds, everything held, so that each round lies elsewhere), beside ONE handle of the whole size
    for (int round = 0; round < 4; ++round) {
        printf("no spacers, round %d:", round);
        {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, A, &prop, 0));
            void *p = nullptr;
            CK(hipMemAddressReserve(&p, A, 2 * MiB, nullptr, 0));
            CK(hipMemMap(p, A, 0, h, 0));
            CK(hipMemSetAccess(p, A, &acc0, 1));
            printf("  one handle %.0f us;", time_us(src, (char *)p));
        }
        for (long long C : {64 * MiB, 256 * MiB}) {
            void *p = nullptr;
            CK(hipMemAddressReserve(&p, A, 2 * MiB, nullptr, 0));
            for (long long o = 0; o < A; o += C) {
                hipMemGenericAllocationHandle_t h;
                CK(hipMemCreate(&h, C, &prop, 0));
                CK(hipMemMap((char *)p + o, C, 0, h, 0));
            }
            CK(hipMemSetAccess(p, A, &acc0, 1));
            printf("  handles of %lld MiB in order %.0f us;", C / MiB, time_us(src, (char *)p));
        }
        {   // and a plain hipMalloc right after
            char *d;
            CK(hipMalloc(&d, A));
            printf("  hipMalloc %.0f us\n", time_us(src, d));
        }
        fflush(stdout);
    }
    struct Group {
        hipMemGenericAllocationHandle_t whole;
        std::vector<hipMemGenericAllocationHandle_t> ch[3];
    };
    std::vector<Group> groups(K);
    std::vector<hipMemGenericAllocationHandle_t> spacers;
    const double t0 = now_ms();
    for (int g = 0; g < K; ++g) {
        CK(hipMemCreate(&groups[g].whole, A, &prop, 0));
        for (int c = 0; c < 3; ++c) {
            const long long n = A / chunkC[c];  // group g backs chunks i = g, g + K, ... : ceil((n - g) / K)
            for (long long i = g; i < n; i += K) {
                hipMemGenericAllocationHandle_t h;
                CK(hipMemCreate(&h, chunkC[c], &prop, 0));
                groups[g].ch[c].push_back(h);
            }
        }
        if (g + 1 < K) {
            hipMemGenericAllocationHandle_t s;
            const hipError_t e = hipMemCreate(&s, spacer, &prop, 0);
            if (e != hipSuccess) {
                fprintf(stderr, "spacer %d: %s\n", g, hipGetErrorString(e));
                (void)hipGetLastError();
            } else
                spacers.push_back(s);
        }
    }
    for (auto s : spacers) CK(hipMemRelease(s));
    printf("groups + spacers created in %.0f ms\n", now_ms() - t0);

    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    auto reserve = [&]() {
        void *p = nullptr;
        CK(hipMemAddressReserve(&p, A, 2 * MiB, nullptr, 0));
        return (char *)p;
    };
    // 2: whole arenas, one per group
    std::vector<char *> whole(K);
    printf("one region each (us): ");
    for (int g = 0; g < K; ++g) {
        whole[g] = reserve();
        CK(hipMemMap(whole[g], A, 0, groups[g].whole, 0));
        CK(hipMemSetAccess(whole[g], A, &acc, 1));
        printf(" g%d %.0f", g, time_us(src, whole[g]));
        fflush(stdout);
    }
    printf("\n");
    // 3: striped
    char *striped[3];
    for (int c = 0; c < 3; ++c) {
        const double t1 = now_ms();
        char *va = reserve();
        const long long n = A / chunkC[c];
        for (long long i = 0; i < n; ++i) CK(hipMemMap(va + i * chunkC[c], chunkC[c], 0, groups[i % K].ch[c][i / K], 0));
        CK(hipMemSetAccess(va, A, &acc, 1));
        const double t2 = now_ms();
        striped[c] = va;
        printf("striped over %d groups, chunk %3lld MiB (mapped in %.1f ms): %.0f us", K, chunkC[c] / MiB, t2 - t1, time_us(src, va));
        printf("   again %.0f\n", time_us(src, va));
        fflush(stdout);
    }
    // 4: striped over PAIRS of groups (64 MiB chunks taken from two groups' handles only): which pairing matters?
    if (K >= 4) {
        for (int a = 0; a < K; ++a)
            for (int b = a + 1; b < K; ++b) {
                char *va = reserve();
                const long long n = A / chunkC[2];
                // group a's 64 MiB handles serve even chunks, group b's odd chunks; each group holds ceil(n / K) of them: cover what they can, the rest from the whole handle is not mappable piecewise -> use only the first 2 * per chunks
                const long long per = (long long)std::min(groups[a].ch[2].size(), groups[b].ch[2].size());
                const long long m = std::min(n, 2 * per);
                for (long long i = 0; i < m; ++i) CK(hipMemMap(va + i * chunkC[2], chunkC[2], 0, groups[i & 1 ? b : a].ch[2][i / 2], 0));
                CK(hipMemSetAccess(va, m * chunkC[2], &acc, 1));
                // the batch does not fit m chunks: time a batch cut to the streams that fit
                std::vector<Stream> cut;
                for (const Stream &s : hs)
                    if (s.dst + (long long)(s.rows - 1) * s.stride + 960 <= m * chunkC[2]) cut.push_back(s);
                Stream *dcut;
                CK(hipMalloc(&dcut, sizeof(Stream) * cut.size()));
                CK(hipMemcpy(dcut, cut.data(), sizeof(Stream) * cut.size(), hipMemcpyHostToDevice));
                auto t_cut = [&](char *dst) {
                    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(ring_copy, dim3(cut.size()), dim3(64), 0, 0, src, dst, dcut);
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(ring_copy, dim3(cut.size()), dim3(64), 0, 0, src, dst, dcut);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    return ms * 1e3 / 8;
                };
                printf("pair g%d+g%d (%zu of %d streams): striped %.0f us; the same streams on whole g%d %.0f, g%d %.0f\n", a, b, cut.size(), nstreams, t_cut(va), a, t_cut(whole[a]), b, t_cut(whole[b]));
                fflush(stdout);
                CK(hipMemUnmap(va, m * chunkC[2]));
                CK(hipMemAddressFree(va, A));
                CK(hipFree(dcut));
            }
    }
    // 5: source striped too
    printf("source AND destination striped (64 MiB): ");
    {
        // a second striped arena needs its own physical chunks: reuse the 16 MiB set as the source
        printf("src = striped 16 MiB, dst = striped 64 MiB: %.0f us; src striped 16, dst whole g0: %.0f us\n", time_us(striped[1], striped[2]), time_us(striped[1], whole[0]));
    }
    return 0;
}
