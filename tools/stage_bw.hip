// Development: how fast can T host threads move 4K YUV420P16 frames (24.9 MB) from ordinary
// (pageable) memory to the GPU and back? Modes: 0 = hipMemcpyAsync on the pageable pointers (the
// runtime pins in place), 1 = CPU memcpy into a per-thread pinned buffer + DMA, 2 = hipHostRegister
// around every frame, 3 = mode 1 with the frame split in 4 chunks so that memcpy and DMA overlap.
// Build: hipcc -O2 --offload-arch=gfx950 tools/stage_bw.hip -o gpurun_out/stage_bw -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

static const size_t kFrame = (size_t)3840 * 2160 * 2 * 3 / 2;

static void worker(int mode, int iters, std::atomic<int> *go) {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    char *din, *dout, *pin_in = nullptr, *pin_out = nullptr;
    CK(hipMalloc(&din, kFrame));
    CK(hipMalloc(&dout, kFrame));
    char *src = (char *)aligned_alloc(64, kFrame), *dst = (char *)aligned_alloc(64, kFrame);
    memset(src, 1, kFrame);
    memset(dst, 2, kFrame);
    if (mode == 1 || mode == 3) {
        CK(hipHostMalloc(&pin_in, kFrame, hipHostMallocDefault));
        CK(hipHostMalloc(&pin_out, kFrame, hipHostMallocDefault));
    }
    while (go->load() == 0) std::this_thread::yield();
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {
            CK(hipMemcpyAsync(din, src, kFrame, hipMemcpyHostToDevice, st));
            CK(hipMemcpyAsync(dout, din, kFrame, hipMemcpyDeviceToDevice, st));
            CK(hipMemcpyAsync(dst, dout, kFrame, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
        } else if (mode == 1) {
            memcpy(pin_in, src, kFrame);
            CK(hipMemcpyAsync(din, pin_in, kFrame, hipMemcpyHostToDevice, st));
            CK(hipMemcpyAsync(dout, din, kFrame, hipMemcpyDeviceToDevice, st));
            CK(hipMemcpyAsync(pin_out, dout, kFrame, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            memcpy(dst, pin_out, kFrame);
        } else if (mode == 2) {
            CK(hipHostRegister(src, kFrame, hipHostRegisterDefault));
            CK(hipHostRegister(dst, kFrame, hipHostRegisterDefault));
            CK(hipMemcpyAsync(din, src, kFrame, hipMemcpyHostToDevice, st));
            CK(hipMemcpyAsync(dout, din, kFrame, hipMemcpyDeviceToDevice, st));
            CK(hipMemcpyAsync(dst, dout, kFrame, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            CK(hipHostUnregister(src));
            CK(hipHostUnregister(dst));
        } else {
            const int nc = 4;
            const size_t cs = kFrame / nc;
            hipEvent_t ev[nc];
            for (int c = 0; c < nc; ++c) {
                memcpy(pin_in + c * cs, src + c * cs, cs);
                CK(hipMemcpyAsync(din + c * cs, pin_in + c * cs, cs, hipMemcpyHostToDevice, st));
            }
            CK(hipMemcpyAsync(dout, din, kFrame, hipMemcpyDeviceToDevice, st));
            for (int c = 0; c < nc; ++c) {
                CK(hipMemcpyAsync(pin_out + c * cs, dout + c * cs, cs, hipMemcpyDeviceToHost, st));
                CK(hipEventCreateWithFlags(&ev[c], hipEventDisableTiming));
                CK(hipEventRecord(ev[c], st));
            }
            for (int c = 0; c < nc; ++c) {
                CK(hipEventSynchronize(ev[c]));
                memcpy(dst + c * cs, pin_out + c * cs, cs);
                CK(hipEventDestroy(ev[c]));
            }
        }
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 40;
    CK(hipSetDevice(0));
    for (int mode = 0; mode < 4; ++mode)
        for (int T : {1, 2, 4, 8, 16, 32}) {
            std::atomic<int> go{0};
            std::vector<std::thread> pool;
            for (int t = 0; t < T; ++t) pool.emplace_back(worker, mode, iters, &go);
            std::this_thread::sleep_for(std::chrono::milliseconds(300 + 40 * T));
            const auto t0 = std::chrono::steady_clock::now();
            go.store(1);
            for (auto &t : pool) t.join();
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("mode %d threads %2d: %7.1f frames/s  (%.1f GB/s each way)\n", mode, T, T * iters / sec, T * iters * kFrame / sec / 1e9);
            fflush(stdout);
        }
    return 0;
}
