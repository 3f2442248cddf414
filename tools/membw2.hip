// HBM ceiling sweep (round 2): does a tuned copy on this pool plateau at the guide's 6.29 TB/s
// (MI355X_MICROARCH.md:36) or at the 5.4 TB/s round 1 measured?  Launches are >= 0.5 ms
// (1.59 GB in + 1.59 GB out = the BoxBlur headline's 64 4K YUV420P16 frames), 2 MiB-aligned
// bases, grids from 1 k to 128 k blocks and persistent 256*k grids, default / nt policies,
// grid-stride and block-contiguous traversal, plus the ring kernel's own access shape.
// build: hipcc --offload-arch=gfx950 -O3 tools/membw2.hip -o gpurun_out/membw2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS> __device__ __forceinline__ u4 ld(const u4 *p) {
    if (NTL) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NTS> __device__ __forceinline__ void st(u4 *p, u4 v) {
    if (NTS) __builtin_nontemporal_store(v, p); else *p = v;
}

// grid-stride, U loads in flight per lane
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_gs(const u4 *__restrict__ s, u4 *__restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stp = (size_t)gridDim.x * blockDim.x;
    for (; i + (U - 1) * stp < n; i += U * stp) {
        u4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = ld<NTL, NTS>(s + i + k * stp);
#pragma unroll
        for (int k = 0; k < U; ++k) st<NTS>(d + i + k * stp, v[k]);
    }
    for (; i < n; i += stp) st<NTS>(d + i, ld<NTL, NTS>(s + i));
}
// block-contiguous: block b copies [b*chunk, (b+1)*chunk), 256 lanes x 16 B = 4 KiB per step, U steps in flight
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_bc(const u4 *__restrict__ s, u4 *__restrict__ d, size_t n, size_t chunk) {
    size_t b0 = (size_t)blockIdx.x * chunk, b1 = b0 + chunk; if (b1 > n) b1 = n;
    size_t i = b0 + threadIdx.x;
    for (; i + (U - 1) * 256 < b1; i += U * 256) {
        u4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = ld<NTL, NTS>(s + i + k * 256);
#pragma unroll
        for (int k = 0; k < U; ++k) st<NTS>(d + i + k * 256, v[k]);
    }
    for (; i < b1; i += 256) st<NTS>(d + i, ld<NTL, NTS>(s + i));
}
template <int U, bool NTL>
__global__ __launch_bounds__(256) void read_gs(const u4 *__restrict__ s, u4 *__restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stp = (size_t)gridDim.x * blockDim.x;
    u4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stp < n; i += U * stp) {
#pragma unroll
        for (int k = 0; k < U; ++k) acc ^= ld<NTL, false>(s + i + k * stp);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) d[threadIdx.x] = acc;
}
template <bool NTS>
__global__ __launch_bounds__(256) void write_gs(u4 *__restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stp = (size_t)gridDim.x * blockDim.x;
    const u4 v = {1, 2, 3, (unsigned)i};
    for (; i < n; i += stp) st<NTS>(d + i, v);
}

// ring-kernel shape: one wave per (480-column tile, band); lanes read 1024 B of a row from 32 B left
// of the tile, lanes 2..61 write 960 B; DEPTH rows in flight; halo extra rows read above the band.
template <int DEPTH, bool NTS>
__global__ __launch_bounds__(64) void copy_tiles(const char *__restrict__ s, char *__restrict__ d, int pitch, int rows, int band, int ntx, int halo, int nblocks) {
    const int chunk = (nblocks + 7) >> 3;
    const int b = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (b >= nblocks) return;
    const int tx = b % ntx, by = b / ntx;
    const int lane = threadIdx.x;
    int y0 = by * band; if (y0 + band > rows) y0 = rows - band;
    int off = tx * 960 - 32 + lane * 16;
    off = off < 0 ? 0 : (off > pitch - 16 ? pitch - 16 : off);
    const bool out = lane >= 2 && lane < 62;
    u4 buf[DEPTH];
    int ys = y0 - halo; if (ys < 0) ys = 0;
    const int total = y0 + band - ys;
    const char *sp = s + (size_t)ys * pitch + off;
    char *dp = d + (size_t)y0 * pitch + off;
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) buf[k] = *reinterpret_cast<const u4 *>(sp + (size_t)(k < total ? k : total - 1) * pitch);
    int skip = y0 - ys;
    for (int i = 0; i < total; i += DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const u4 v = buf[k];
            const int nx = i + k + DEPTH;
            buf[k] = *reinterpret_cast<const u4 *>(sp + (size_t)(nx < total ? nx : total - 1) * pitch);
            const int r = i + k - skip;
            if (out && r >= 0 && i + k < total) st<NTS>(reinterpret_cast<u4 *>(dp + (size_t)r * pitch), v);
        }
    }
}

int main(int argc, char **argv) {
    const int pitch = 7680;
    const int rows = 2160 * 24 * 4;  // 64 4K YUV420P16 frames' worth of bytes in one 2-D array (1.59 GB)
    const size_t bytes = (size_t)pitch * rows;
    char *s, *d;
    CK(hipMalloc(&s, bytes + (2u << 20))); CK(hipMalloc(&d, bytes + (2u << 20)));
    s = (char *)(((uintptr_t)s + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
    d = (char *)(((uintptr_t)d + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
    CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("buffers: %.3f GB each, 2 MiB aligned\n", bytes / 1e9);
    auto timeit = [&](const char *name, double moved, auto &&launch) {
        for (int i = 0; i < 2; ++i) launch();
        std::vector<float> t;
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        const float med = t[t.size() / 2];
        printf("%-52s min %8.1f med %8.1f max %8.1f us  med %7.1f GB/s  best %7.1f GB/s\n", name, t.front() * 1e3, med * 1e3, t.back() * 1e3,
               moved / (med * 1e-3) / 1e9, moved / (t.front() * 1e-3) / 1e9);
        fflush(stdout);
    };
    const size_t n16 = bytes / 16;
    char nm[128];
#define GS(U, NTL, NTS, blocks) do { snprintf(nm, sizeof nm, "copy_gs U=%d ntl=%d nts=%d blocks=%d", U, NTL, NTS, blocks); \
        timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_gs<U, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, (const u4 *)s, (u4 *)d, n16); }); } while (0)
#define BC(U, NTL, NTS, blocks) do { snprintf(nm, sizeof nm, "copy_bc U=%d ntl=%d nts=%d blocks=%d", U, NTL, NTS, blocks); \
        const size_t chunk = ((n16 + (blocks) - 1) / (blocks) + 255) / 256 * 256; \
        timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_bc<U, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, (const u4 *)s, (u4 *)d, n16, chunk); }); } while (0)
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072}) { GS(4, false, false, blocks); }
    for (int blocks : {2048, 8192, 32768, 131072}) { GS(4, false, true, blocks); GS(4, true, true, blocks); GS(8, false, true, blocks); }
    for (int k : {2, 4, 8, 16, 32}) { GS(4, false, false, 256 * k); GS(4, false, true, 256 * k); GS(8, false, true, 256 * k); GS(2, false, true, 256 * k); }
    for (int blocks : {2048, 8192, 32768, 131072, 388800}) { BC(4, false, false, blocks); BC(4, false, true, blocks); BC(8, false, true, blocks); BC(4, true, true, blocks); }
    timeit("hipMemcpyDtoD", 2.0 * bytes, [&] { CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0)); });
    for (int blocks : {2048, 8192, 32768}) {
        snprintf(nm, sizeof nm, "read_gs U=4 blocks=%d", blocks);
        timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((read_gs<4, false>), dim3(blocks), dim3(256), 0, 0, (const u4 *)s, (u4 *)d, n16); });
        snprintf(nm, sizeof nm, "read_gs U=8 nt blocks=%d", blocks);
        timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((read_gs<8, true>), dim3(blocks), dim3(256), 0, 0, (const u4 *)s, (u4 *)d, n16); });
        snprintf(nm, sizeof nm, "write_gs blocks=%d", blocks);
        timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((write_gs<false>), dim3(blocks), dim3(256), 0, 0, (u4 *)d, n16); });
        snprintf(nm, sizeof nm, "write_gs nt blocks=%d", blocks);
        timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((write_gs<true>), dim3(blocks), dim3(256), 0, 0, (u4 *)d, n16); });
    }
    const int ntx = 8;
    for (int band : {128, 256, 540, 1080}) {
        for (int halo : {0, 27}) {
            const int nb = (rows + band - 1) / band;
            const int nblocks = nb * ntx;
            const int grid = ((nblocks + 7) / 8) * 8;
            snprintf(nm, sizeof nm, "copy_tiles D=2 nts band=%d halo=%d waves=%d", band, halo, nblocks);
            timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_tiles<2, true>), dim3(grid), dim3(64), 0, 0, s, d, pitch, rows, band, ntx, halo, nblocks); });
            snprintf(nm, sizeof nm, "copy_tiles D=4 nts band=%d halo=%d waves=%d", band, halo, nblocks);
            timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_tiles<4, true>), dim3(grid), dim3(64), 0, 0, s, d, pitch, rows, band, ntx, halo, nblocks); });
            snprintf(nm, sizeof nm, "copy_tiles D=8 nts band=%d halo=%d waves=%d", band, halo, nblocks);
            timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_tiles<8, true>), dim3(grid), dim3(64), 0, 0, s, d, pitch, rows, band, ntx, halo, nblocks); });
            snprintf(nm, sizeof nm, "copy_tiles D=8 band=%d halo=%d waves=%d", band, halo, nblocks);
            timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_tiles<8, false>), dim3(grid), dim3(64), 0, 0, s, d, pitch, rows, band, ntx, halo, nblocks); });
        }
    }
    return 0;
}
