// HBM access-shape probe (round 2): why does a copy in the BoxBlur ring kernel's shape (one wave
// per 960-byte column tile walking down a long band, 3072 bands at once) stop at ~5.0 TB/s when a
// sequential copy reaches 6.4?  Varies, at a fixed long band: tile alignment to 1 KiB DRAM pages,
// bytes per wave-row (1 / 2 KiB), waves per workgroup (lockstep tiles), XCD grouping, rows in flight.
// build: hipcc --offload-arch=gfx950 -O3 tools/membw3.hip -o tools/membw3.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

struct Shape {
    int pitch, rows, band, halo;
    int tile_pitch;   // bytes between tile starts (960: ring kernel, 1024: page aligned)
    int tile_skew;    // bytes subtracted from the tile start for the load (32: ring kernel halo lanes)
    int ntx;          // tiles per row
    int remap;        // 1: each XCD gets a contiguous eighth of the block list
    int nblocks;      // (tile, band) units
};

// NW waves per workgroup = NW adjacent tiles walking in lockstep; NL 16-byte loads per lane and row
// (NL = 2: a wave covers 2 KiB of a row, lane l reads [16l,16l+16) and [1024+16l, ...)).
template <int NW, int NL, int DEPTH, bool NTL, bool NTS>
__global__ __launch_bounds__(64 * NW) void copy_shape(const char *__restrict__ s, char *__restrict__ d, const Shape sh) {
    int wg = blockIdx.x;
    const int nwg = (sh.nblocks + NW - 1) / NW;
    if (sh.remap) {
        const int chunk = (nwg + 7) >> 3;
        wg = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    }
    if (wg >= nwg) return;
    const int b = wg * NW + (int)(threadIdx.x >> 6);
    if (b >= sh.nblocks) return;
    const int lane = threadIdx.x & 63;
    const int tx = b % sh.ntx, by = b / sh.ntx;
    int y0 = by * sh.band; if (y0 + sh.band > sh.rows) y0 = sh.rows - sh.band;
    int off = tx * sh.tile_pitch * NL - sh.tile_skew + lane * 16;
    off = off < 0 ? 0 : (off > sh.pitch - 16 * NL ? sh.pitch - 16 * NL : off);
    const bool out = sh.tile_skew == 0 || (lane >= 2 && lane < 62);
    u4 buf[DEPTH][NL];
    int ys = y0 - sh.halo; if (ys < 0) ys = 0;
    const int total = y0 + sh.band - ys;
    const char *sp = s + (size_t)ys * sh.pitch + off;
    char *dp = d + (size_t)y0 * sh.pitch + off;
    auto ld = [&](int r, int j) {
        const u4 *p = reinterpret_cast<const u4 *>(sp + (size_t)(r < total ? r : total - 1) * sh.pitch + j * 1024);
        return NTL ? __builtin_nontemporal_load(p) : *p;
    };
#pragma unroll
    for (int k = 0; k < DEPTH; ++k)
#pragma unroll
        for (int j = 0; j < NL; ++j) buf[k][j] = ld(k, j);
    const int skip = y0 - ys;
    for (int i = 0; i < total; i += DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            u4 v[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) { v[j] = buf[k][j]; buf[k][j] = ld(i + k + DEPTH, j); }
            const int r = i + k - skip;
            if (out && r >= 0 && i + k < total) {
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    u4 *q = reinterpret_cast<u4 *>(dp + (size_t)r * sh.pitch + j * 1024);
                    if (NTS) __builtin_nontemporal_store(v[j], q); else *q = v[j];
                }
            }
        }
    }
}

int main() {
    const int pitch = 8192;                       // 8 page-aligned 1 KiB tiles per row
    const int rows = 194400;                      // 1.593 GB, as the 64-frame BoxBlur launch
    const size_t bytes = (size_t)pitch * rows;
    char *s, *d;
    CK(hipMalloc(&s, bytes + (2u << 20))); CK(hipMalloc(&d, bytes + (2u << 20)));
    s = (char *)(((uintptr_t)s + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
    d = (char *)(((uintptr_t)d + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
    CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double moved, auto &&launch) {
        for (int i = 0; i < 2; ++i) launch();
        std::vector<float> t;
        for (int i = 0; i < 10; ++i) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("%-78s med %8.1f us  %7.1f GB/s  (best %7.1f)\n", name, t[5] * 1e3, moved / (t[5] * 1e-3) / 1e9, moved / (t[0] * 1e-3) / 1e9);
        fflush(stdout);
    };
    char nm[160];
    auto mk = [&](int band, int halo, int tile_pitch, int skew, int nl, int remap) {
        Shape sh; sh.pitch = pitch; sh.rows = rows; sh.band = band; sh.halo = halo; sh.tile_pitch = tile_pitch; sh.tile_skew = skew;
        sh.ntx = (pitch + tile_pitch * nl - 1) / (tile_pitch * nl); sh.remap = remap;
        sh.nblocks = ((rows + band - 1) / band) * sh.ntx; return sh;
    };
#define RUN(NW, NL, D, NTL, NTS, sh, label) do { const Shape q = (sh); const int nwg = (q.nblocks + NW - 1) / NW; const int grid = ((nwg + 7) / 8) * 8; \
        const double moved = 2.0 * bytes * (q.tile_pitch == 960 ? 960.0 * q.ntx / pitch : 1.0); \
        snprintf(nm, sizeof nm, "%s NW=%d NL=%d D=%d ntl=%d nts=%d band=%d halo=%d tp=%d skew=%d remap=%d waves=%d", label, NW, NL, D, NTL, NTS, q.band, q.halo, q.tile_pitch, q.tile_skew, q.remap, q.nblocks); \
        timeit(nm, moved, [&] { hipLaunchKernelGGL((copy_shape<NW, NL, D, NTL, NTS>), dim3(grid), dim3(64 * NW), 0, 0, s, d, q); }); } while (0)
    for (int band : {506, 253}) {   // 194400/506 = 384.2 bands -> ~3080 waves of 8 tiles; 253: two generations
        RUN(1, 1, 4, false, true, mk(band, 27, 960, 32, 1, 1), "ring-like ");
        RUN(1, 1, 4, false, true, mk(band, 27, 1024, 0, 1, 1), "aligned   ");
        RUN(1, 1, 4, false, true, mk(band, 0, 1024, 0, 1, 1), "aligned h0");
        RUN(1, 1, 4, true, true, mk(band, 27, 1024, 0, 1, 1), "aligned   ");
        RUN(1, 1, 8, false, true, mk(band, 27, 1024, 0, 1, 1), "aligned   ");
        RUN(1, 1, 2, false, true, mk(band, 27, 1024, 0, 1, 1), "aligned   ");
        RUN(1, 1, 4, false, false, mk(band, 27, 1024, 0, 1, 1), "aligned   ");
        RUN(1, 1, 4, false, true, mk(band, 27, 1024, 0, 1, 0), "aligned   ");
        RUN(2, 1, 4, false, true, mk(band, 27, 1024, 0, 1, 1), "aligned   ");
        RUN(4, 1, 4, false, true, mk(band, 27, 1024, 0, 1, 1), "aligned   ");
        RUN(8, 1, 4, false, true, mk(band, 27, 1024, 0, 1, 1), "aligned   ");
        RUN(8, 1, 4, false, true, mk(band, 27, 1024, 0, 1, 0), "aligned   ");
        RUN(1, 2, 4, false, true, mk(band, 27, 1024, 0, 2, 1), "aligned2K ");
        RUN(1, 2, 2, false, true, mk(band, 27, 1024, 0, 2, 1), "aligned2K ");
        RUN(4, 2, 2, false, true, mk(band, 27, 1024, 0, 2, 1), "aligned2K ");
    }
    // half / quarter the number of concurrent streams at the same total: longer bands, fewer waves
    RUN(1, 2, 4, false, true, mk(1012, 27, 1024, 0, 2, 1), "aligned2K ");
    RUN(1, 1, 8, false, true, mk(1012, 27, 1024, 0, 1, 1), "aligned   ");
    // short bands in dispatch order (compact window), no halo: what the window alone is worth
    for (int band : {8, 16, 32, 64, 128}) {
        RUN(1, 1, 4, false, true, mk(band, 0, 1024, 0, 1, 0), "raster h0 ");
        RUN(4, 1, 4, false, true, mk(band, 0, 1024, 0, 1, 0), "raster h0 ");
    }
    for (int band : {32, 64, 128}) RUN(1, 1, 4, false, true, mk(band, 27, 1024, 0, 1, 0), "raster    ");
    return 0;
}
