// HBM calibration for the BoxBlur roofline: what does this MI355X deliver for (a) a linear
// copy, (b) a copy in the ring kernel's access shape (one wave per 960-byte column tile
// marching down a band of rows, 16 B per lane, D rows in flight), (c) read-only / write-only.
// build: hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o gpurun_out/membw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void copy_linear(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * st < n; i += 4 * st) {
        uint4 a = s[i], b = s[i + st], c = s[i + 2 * st], e = s[i + 3 * st];
        d[i] = a; d[i + st] = b; d[i + 2 * st] = c; d[i + 3 * st] = e;
    }
    for (; i < n; i += st) d[i] = s[i];
}
__global__ void read_linear(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i + 3 * st < n; i += 4 * st) {
        uint4 a = s[i], b = s[i + st], c = s[i + 2 * st], e = s[i + 3 * st];
        acc.x ^= a.x ^ b.x ^ c.x ^ e.x; acc.y ^= a.y ^ b.y ^ c.y ^ e.y; acc.z ^= a.z ^ b.z ^ c.z ^ e.z; acc.w ^= a.w ^ b.w ^ c.w ^ e.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) d[threadIdx.x] = acc;
}
__global__ void write_linear(uint4 *__restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    const uint4 v = make_uint4(1, 2, 3, (unsigned)i);
    for (; i < n; i += st) d[i] = v;
}

// ring-kernel shape: wave per (tile, band); lanes 0..63 read 1024 B of a row starting 32 B left
// of the tile (clamped), lanes 2..61 write 960 B. DEPTH rows in flight. halo = extra rows read
// above the band (read amplification like the blur's 2r+D).
template <int DEPTH>
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
    uint4 buf[DEPTH];
    int ys = y0 - halo; if (ys < 0) ys = 0;
    const int total = y0 + band - ys;
    const char *sp = s + (size_t)ys * pitch + off;
    char *dp = d + (size_t)y0 * pitch + off;
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) buf[k] = *reinterpret_cast<const uint4 *>(sp + (size_t)(k < total ? k : total - 1) * pitch);
    int skip = y0 - ys;
    for (int i = 0; i < total; i += DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const uint4 v = buf[k];
            const int nx = i + k + DEPTH;
            buf[k] = *reinterpret_cast<const uint4 *>(sp + (size_t)(nx < total ? nx : total - 1) * pitch);
            const int r = i + k - skip;
            if (out && r >= 0 && i + k < total) *reinterpret_cast<uint4 *>(dp + (size_t)r * pitch) = v;
        }
    }
}

int main() {
    const int pitch = 7680, rows = 2160 * 24;  // 16 4K YUV420P16 frames' worth of bytes in one 2-D array
    const size_t bytes = (size_t)pitch * rows;
    char *s, *d;
    CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes));
    CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double moved, auto &&launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        const int it = 20;
        for (int i = 0; i < it; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms * 1e3 / it, moved * it / (ms * 1e-3) / 1e9);
    };
    const size_t n16 = bytes / 16;
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy_linear blocks=%d x256", blocks);
        timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_linear, dim3(blocks), dim3(256), 0, 0, (const uint4 *)s, (uint4 *)d, n16); });
    }
    timeit("read_linear 8192x256", 1.0 * bytes, [&] { hipLaunchKernelGGL(read_linear, dim3(8192), dim3(256), 0, 0, (const uint4 *)s, (uint4 *)d, n16); });
    timeit("write_linear 8192x256", 1.0 * bytes, [&] { hipLaunchKernelGGL(write_linear, dim3(8192), dim3(256), 0, 0, (uint4 *)d, n16); });
    timeit("hipMemcpyDtoD", 2.0 * bytes, [&] { CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0)); });
    const int ntx = 8;
    for (int band : {30, 64, 128, 256, 512}) {
        for (int halo : {0, 31}) {
            const int nb = (rows + band - 1) / band;
            const int nblocks = nb * ntx;
            const int grid = ((nblocks + 7) / 8) * 8;
            char nm[96];
            snprintf(nm, sizeof nm, "copy_tiles D=4 band=%d halo=%d waves=%d", band, halo, nblocks);
            timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_tiles<4>, dim3(grid), dim3(64), 0, 0, s, d, pitch, rows, band, ntx, halo, nblocks); });
            snprintf(nm, sizeof nm, "copy_tiles D=8 band=%d halo=%d waves=%d", band, halo, nblocks);
            timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_tiles<8>, dim3(grid), dim3(64), 0, 0, s, d, pitch, rows, band, ntx, halo, nblocks); });
        }
    }
    return 0;
}
