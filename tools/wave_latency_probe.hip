// GPU box: what ONE wave alone on its SIMD pays per instruction — dependent f32 adds, independent f32 adds, an LDS write -> read round trip,
// independent f64 FMAs — in ns (HIP events around 1 workgroup x 64 threads, and around 1024 such workgroups: one per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/wave_latency_probe.bin tools/wave_latency_probe.hip && tools/wave_latency_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 200000
__global__ void dep_add(float *out, float a) {
    float x = threadIdx.x;
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    }
    out[threadIdx.x + blockIdx.x * 64] = x;
}
__global__ void indep_add(float *out, float a) {
    float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(a));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(x1) : "v"(a));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(x2) : "v"(a));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(x3) : "v"(a));
        }
    }
    out[threadIdx.x + blockIdx.x * 64] = x0 + x1 + x2 + x3;
}
__global__ void indep_add_e64(float *out, float a) {  // the same adds in the 8-byte VOP3 encoding
    float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(x0) : "v"(a));
            asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(x1) : "v"(a));
            asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(x2) : "v"(a));
            asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(x3) : "v"(a));
        }
    }
    out[threadIdx.x + blockIdx.x * 64] = x0 + x1 + x2 + x3;
}
__global__ void lds_indep(float *out, float a) {  // LDS reads nobody waits for until the end of the group: the issue cost of a DS instruction
    __shared__ float t[64 * 17];
    for (int k = threadIdx.x; k < 64 * 17; k += 64) t[k] = a;
    float acc = 0;
    const float *p = t + threadIdx.x;
    for (int i = 0; i < N; ++i) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[u]) : "v"((uint32_t)(threadIdx.x * 4)), "n"(u * 256));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    out[threadIdx.x + blockIdx.x * 64] = acc + p[0];
}
__global__ void lds_rt(float *out, float a) {
    __shared__ float t[128];
    float x = threadIdx.x;
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            t[threadIdx.x] = x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            x = t[threadIdx.x ^ 1] + a;
        }
    }
    out[threadIdx.x + blockIdx.x * 64] = x;
}
__global__ void indep_fma64(float *out, float a) {
    double x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, b = a;
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x0) : "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x1) : "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x2) : "v"(b));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x3) : "v"(b));
        }
    }
    out[threadIdx.x + blockIdx.x * 64] = (float)(x0 + x1 + x2 + x3);
}
template <typename F>
static void run(const char *name, F f, int blocks, double ops) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(blocks); hipDeviceSynchronize();
    hipEventRecord(e0); f(blocks); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %5d waves: %7.2f ns per op\n", name, blocks, ms * 1e6 / ops);
}
int main() {
    float *out; hipMalloc(&out, 8192 * 64 * 4);
    for (int blocks : {1, 1024, 2048, 4096, 8192}) {
        run("dependent v_add_f32", [&](int b) { dep_add<<<b, 64>>>(out, 1.0f); }, blocks, 16.0 * N);
        run("4 independent v_add_f32", [&](int b) { indep_add<<<b, 64>>>(out, 1.0f); }, blocks, 16.0 * N);
        run("4 independent v_add_f32_e64", [&](int b) { indep_add_e64<<<b, 64>>>(out, 1.0f); }, blocks, 16.0 * N);
        run("16 ds_read_b32 + 16 adds", [&](int b) { lds_indep<<<b, 64>>>(out, 1.0f); }, blocks, 16.0 * N);
        run("LDS write->read round trip", [&](int b) { lds_rt<<<b, 64>>>(out, 1.0f); }, blocks, 4.0 * N);
        run("4 independent v_fma_f64", [&](int b) { indep_fma64<<<b, 64>>>(out, 1.0f); }, blocks, 16.0 * N);
    }
    return 0;
}
