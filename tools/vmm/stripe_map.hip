// Round 5, second step: a MAP of the placement effect along physical memory, and which stripe
// distances cure it. N GiB of VRAM are taken as hipMemCreate handles of 64 MiB in allocation order (slab s = handles
// 16 s ... 16 s + 15 = 1 GiB); arenas of 1.5 GiB (24 chunks) are then assembled from any chunks with hipMemMap (a handle
// may be mapped many times) and timed with the ring kernel's access shape (tools/vmm/stripe_probe.hip).
//   1: contiguous arenas at every slab offset -> the class along physical order (if the driver allocates in order)
//   2: arenas alternating between two slab runs d GiB apart, d = 1 ... N/2, from several start slabs
//   3: K-way stripes with equal spacing, K = 2 ... 16, and random chunk picks
//   hipcc --offload-arch=gfx950 -O3 -o tools/vmm/stripe_map.bin tools/vmm/stripe_map.hip && ./tools/vmm/stripe_map.bin [N_GiB=128]
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
static const long long CH = 64 * MiB;

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 128;  // GiB mapped
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
    const int nch = (int)((total + CH - 1) / CH);  // 24
    const long long A = nch * CH;
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
        const int n = 5;
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
    const int nh = N * 16;
    std::vector<hipMemGenericAllocationHandle_t> h(nh);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nh; ++i) CK(hipMemCreate(&h[i], CH, &prop, 0));
    printf("%d handles of 64 MiB (%d GiB) created in %.0f ms; arena %d chunks\n", nh, N, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), nch);
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void *vap = nullptr;
    CK(hipMemAddressReserve(&vap, A, 2 * MiB, nullptr, 0));
    char *va = (char *)vap;
    // times an arena made of the given chunk indices (24 of them)
    auto run = [&](const std::vector<int> &idx, const char *s = nullptr) {
        for (int i = 0; i < nch; ++i) CK(hipMemMap(va + i * CH, CH, 0, h[idx[i]], 0));
        CK(hipMemSetAccess(va, A, &acc, 1));
        const double us = time_us(s ? s : src, va);
        CK(hipMemUnmap(va, A));
        return us;
    };
    // 1: contiguous, every 1 GiB
    printf("1: contiguous arena starting at slab s (GiB), us:\n");
    std::vector<double> cont;
    for (int s = 0; s * 16 + nch <= nh; ++s) {
        std::vector<int> idx(nch);
        for (int i = 0; i < nch; ++i) idx[i] = s * 16 + i;
        cont.push_back(run(idx));
        printf(" %3.0f", cont.back());
        if (s % 32 == 31) printf("\n");
    }
    printf("\n");
    // fine: every 64 MiB over the first 24 GiB
    printf("1b: contiguous arena starting at chunk c (every 4th = 256 MiB) over the first 24 GiB, us:\n");
    for (int c = 0; c + nch <= std::min(nh, 24 * 16); c += 4) {
        std::vector<int> idx(nch);
        for (int i = 0; i < nch; ++i) idx[i] = c + i;
        printf(" %3.0f", run(idx));
        if ((c / 4) % 32 == 31) printf("\n");
    }
    printf("\n");
    // 2: alternating between runs at slab a and slab a + d
    printf("2: chunks alternate between a run at slab a and a run at slab a + d (12 chunks each), us:\n      d:");
    std::vector<int> ds;
    for (int d = 1; d <= N / 2; d *= 2) ds.push_back(d);
    for (int d : {3, 5, 6, 10, 12, 20, 24, 40, 48}) if (d <= N / 2) ds.push_back(d);
    std::sort(ds.begin(), ds.end());
    for (int d : ds) printf(" %4d", d);
    printf("\n");
    for (int a : {0, 2, 5, 9, 14, 20, 27, 35, 44, 54}) {
        if (a + 1 >= N) break;
        printf(" a=%3d (%3.0f):", a, a < (int)cont.size() ? cont[a] : 0.0);
        for (int d : ds) {
            if ((a + d) * 16 + 12 > nh) { printf("    -"); continue; }
            std::vector<int> idx(nch);
            for (int i = 0; i < nch; ++i) idx[i] = (i & 1 ? a + d : a) * 16 + i / 2;
            printf(" %4.0f", run(idx));
        }
        printf("\n");
        fflush(stdout);
    }
    // 2b: split in HALVES (first 12 chunks at a, last 12 at a + d): straddling without interleaving
    printf("2b: first half at slab a, second half at slab a + d, us:\n");
    for (int a : {0, 5, 14, 27}) {
        printf(" a=%3d:", a);
        for (int d : ds) {
            if ((a + d) * 16 + 12 > nh) { printf("    -"); continue; }
            std::vector<int> idx(nch);
            for (int i = 0; i < nch; ++i) idx[i] = (i >= 12 ? a + d : a) * 16 + i % 12;
            printf(" %4.0f", run(idx));
        }
        printf("\n");
    }
    // 3: K-way equal spacing
    printf("3: K-way stripes, chunk i from slab (i mod K) * N / K + offset, us:\n");
    for (int K : {2, 3, 4, 6, 8, 12, 16, 24}) {
        printf(" K=%2d:", K);
        for (int off : {0, 1, 3}) {
            std::vector<int> idx(nch);
            bool ok = true;
            for (int i = 0; i < nch; ++i) {
                idx[i] = ((i % K) * (N / K) + off) * 16 + i / K;
                ok = ok && idx[i] < nh;
            }
            if (ok) printf(" %4.0f", run(idx));
        }
        printf("\n");
    }
    // random picks
    printf("3b: 24 random chunks out of all, us:");
    srand(12345);
    for (int rep = 0; rep < 12; ++rep) {
        std::vector<int> all(nh);
        for (int i = 0; i < nh; ++i) all[i] = i;
        for (int i = 0; i < nch; ++i) std::swap(all[i], all[i + rand() % (nh - i)]);
        all.resize(nch);
        printf(" %3.0f", run(all));
    }
    printf("\n");
    // random picks out of the FIRST 8 GiB only (a pool that small would do?)
    for (int pool : {2, 4, 8, 16, 32}) {
        if (pool > N) break;
        printf("3c: 24 random chunks out of the first %d GiB, us:", pool);
        for (int rep = 0; rep < 8; ++rep) {
            const int np = pool * 16;
            std::vector<int> all(np);
            for (int i = 0; i < np; ++i) all[i] = i;
            for (int i = 0; i < nch; ++i) std::swap(all[i], all[i + rand() % (np - i)]);
            all.resize(nch);
            printf(" %3.0f", run(all));
        }
        printf("\n");
    }
    // 4: the SOURCE arena striped as well (K = 8), against a striped destination
    {
        void *vsp = nullptr;
        CK(hipMemAddressReserve(&vsp, A, 2 * MiB, nullptr, 0));
        char *vs = (char *)vsp;
        const int K = 8;
        for (int i = 0; i < nch; ++i) CK(hipMemMap(vs + i * CH, CH, 0, h[((i % K) * (N / K) + 2) * 16 + 8 + i / K], 0));
        CK(hipMemSetAccess(vs, A, &acc, 1));
        CK(hipMemset(vs, 1, A));
        std::vector<int> idx(nch);
        for (int i = 0; i < nch; ++i) idx[i] = ((i % K) * (N / K)) * 16 + i / K;
        printf("4: source striped K=8 too: %.0f us (plain source: %.0f)\n", run(idx, vs), run(idx));
        for (int s : {0, 5, 14}) {
            for (int i = 0; i < nch; ++i) idx[i] = s * 16 + i;
            printf("   contiguous destination at slab %d, striped source: %.0f us (plain source %.0f)\n", s, run(idx, vs), run(idx));
        }
    }
    return 0;
}
