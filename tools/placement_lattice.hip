// Round 3, bounded experiment: WHICH address bits carry the BoxBlur ring kernel's placement effect?
//
// A copy in the ring kernel's own shape — 3072 single-wave streams, each moving a 960-byte column tile down 540 rows of a
// 4K YUV420P16 batch (64 frames: luma 7680-B rows in 8 tiles x 4 bands, chroma 3840-B rows in 4 tiles x 2 bands), 16-B
// loads, 16-B nt stores — timed with the DESTINATION planes laid out on a lattice: plane k starts at a 2 MiB boundary
// plus k x D, for D = 0, 256 B ... 64 MiB. Run on every one of N candidate destination arenas (all held, i.e. a walk
// through VRAM): if some D takes the slow arenas down to the fast arenas' time, the bits D toggles share a hash class
// with whatever the 32 GiB steps flip, and a layout rule can replace the placement lottery. If no D does, the effect
// is not reachable from the layout.
//   hipcc --offload-arch=gfx950 -O3 -o tools/placement_lattice.bin tools/placement_lattice.hip && ./tools/placement_lattice.bin [arenas]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

struct Stream {
    long long src, dst;  // byte offsets of the stream's first row segment
    int stride, rows;
    int dstride, unused;
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
        __builtin_nontemporal_store(b, reinterpret_cast<v4u *>(dp + (size_t)r * s.dstride));
    }
}

int main(int argc, char **argv) {
    const int narena = argc > 1 ? atoi(argv[1]) : 24;
    const int frames = 64;
    const long long MiB = 1 << 20;
    // plane geometry of a 4K YUV420P16 frame
    struct Pl { int stride, h, tiles, bands; };
    const Pl pls[3] = {{7680, 2160, 8, 4}, {3840, 1080, 4, 2}, {3840, 1080, 4, 2}};
    const int nplanes = frames * 3;
    std::vector<long long> psize(nplanes), pbase(nplanes);
    long long total = 0;
    for (int k = 0; k < nplanes; ++k) {
        psize[k] = (long long)pls[k % 3].stride * pls[k % 3].h;
        total = (total + 2 * MiB - 1) / (2 * MiB) * (2 * MiB);
        pbase[k] = total;
        total += psize[k];
    }
    const std::vector<long long> Ds = {0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 3 * 256, 5 * 4096, 7 * 65536,
                                       2 * MiB, 4 * MiB, 8 * MiB, 16 * MiB, 32 * MiB, 64 * MiB};
    const long long Dmax = 64 * MiB;
    const long long small_bytes = total + 192 * (1 * MiB + 7 * 65536) + 4 * MiB;  // room for every D <= 1 MiB (and the odd multiples)
    const long long big_bytes = total + (long long)nplanes * Dmax + 4 * MiB;      // one arena for the large strides
    long long pad = 0;  // extra bytes per DESTINATION row (a padded row pitch); planes then start (h * pad) further apart
    auto make_streams = [&](long long D, std::vector<Stream> &out) {
        out.clear();
        long long shift = 0;
        for (int k = 0; k < nplanes; ++k) {
            const Pl &p = pls[k % 3];
            const int band_rows = p.h / p.bands;
            const long long dstride = p.stride + (k % 3 == 0 ? pad : pad / 2);
            for (int b = 0; b < p.bands; ++b)
                for (int t = 0; t < p.tiles; ++t) {
                    Stream s;
                    s.src = pbase[k] + (long long)b * band_rows * p.stride + t * 960;
                    s.dst = pbase[k] + shift + (long long)k * D + (long long)b * band_rows * dstride + t * 960;
                    s.stride = p.stride;
                    s.dstride = (int)dstride;
                    s.unused = 0;
                    s.rows = band_rows;
                    out.push_back(s);
                }
            shift += (dstride - p.stride) * p.h;
            shift = (shift + 2 * MiB - 1) / (2 * MiB) * (2 * MiB);
        }
    };
    char *src;
    CK(hipMalloc(&src, total + 4 * MiB));
    CK(hipMemset(src, 1, total + 4 * MiB));
    Stream *dst_streams;
    std::vector<Stream> hs;
    make_streams(0, hs);
    const int nstreams = (int)hs.size();
    CK(hipMalloc(&dst_streams, sizeof(Stream) * nstreams));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_us = [&](char *dst, long long D) {
        make_streams(D, hs);
        CK(hipMemcpy(dst_streams, hs.data(), sizeof(Stream) * nstreams, hipMemcpyHostToDevice));
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(ring_copy, dim3(nstreams), dim3(64), 0, 0, src, dst, dst_streams);
        CK(hipEventRecord(e0));
        const int n = 8;
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(ring_copy, dim3(nstreams), dim3(64), 0, 0, src, dst, dst_streams);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / n;
    };
    printf("streams %d, bytes moved per launch %.3f GB (read + write), arena %.2f GiB\n", nstreams, 2.0 * frames * 24883200 / 1e9, small_bytes / 1073741824.0);
    if (argc > 2 && std::string(argv[2]) == "pitch") {
        // the same walk with a padded DESTINATION row pitch instead of a plane lattice (luma pad, chroma half of it)
        const std::vector<long long> pads = {0, 64, 128, 256, 512, 768, 1024, 1280, 2048, 4096, 512 + 7680};
        printf("arena");
        for (long long pd : pads) printf(" %8lld", pd);
        printf("   (destination pitch pad, bytes)\n");
        for (int a = 0; a < narena; ++a) {
            char *d;
            // room for the largest pad: every luma row + 8192 B, every chroma row + 4096 B, and a 2 MiB round-up per plane
            const long long pitch_bytes = total + 64LL * 2160 * 8192 + 128LL * 1080 * 4096 + 192LL * 2 * MiB + 8 * MiB;
            if (hipMalloc(&d, pitch_bytes) != hipSuccess) break;
            {
                pad = pads.back();
                make_streams(0, hs);
                long long hi = 0;
                for (const Stream &q : hs) hi = std::max(hi, q.dst + (long long)(q.rows - 1) * q.dstride + 960);
                pad = 0;
                if (hi > pitch_bytes) {
                    fprintf(stderr, "layout %lld exceeds the arena %lld\n", hi, pitch_bytes);
                    return 2;
                }
            }
            printf("%5d", a);
            for (long long pd : pads) {
                pad = pd;
                printf(" %8.1f", time_us(d, 0));
            }
            pad = 0;
            printf("\n");
            fflush(stdout);
        }
        return 0;
    }
    // 1: the walk — every candidate arena with D = 0 and the small lattices
    std::vector<char *> arenas;
    std::vector<double> base_us;
    printf("arena");
    for (long long D : Ds)
        if (D <= 1048576 || D == 3 * 256 || D == 5 * 4096 || D == 7 * 65536) printf(" %9lld", D);
    printf("\n");
    for (int a = 0; a < narena; ++a) {
        char *d;
        if (hipMalloc(&d, small_bytes) != hipSuccess) break;
        arenas.push_back(d);
        printf("%5d", a);
        for (long long D : Ds) {
            if (D > 7 * 65536 && D != 1048576 && D != 524288) continue;
            const double us = time_us(d, D);
            if (D == 0) base_us.push_back(us);
            printf(" %9.1f", us);
        }
        printf("\n");
        fflush(stdout);
    }
    // 2: large strides on two big arenas (the second after the first: elsewhere in VRAM)
    for (int rep = 0; rep < 3; ++rep) {
        char *big;
        if (hipMalloc(&big, big_bytes) != hipSuccess) break;
        printf("big arena %d (%.1f GiB):", rep, big_bytes / 1073741824.0);
        for (long long D : Ds)
            if (D == 0 || D >= 2 * MiB) printf("  D=%lldM %.1f", D / MiB, time_us(big, D));
        printf("\n");
        fflush(stdout);
        // keep it: the next one lies elsewhere
    }
    std::vector<double> s = base_us;
    std::sort(s.begin(), s.end());
    if (!s.empty()) printf("D=0 over %zu arenas: min %.1f median %.1f max %.1f us\n", s.size(), s.front(), s[s.size() / 2], s.back());
    return 0;
}
