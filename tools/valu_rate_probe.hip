// Issue cost of the integer / conversion instructions the BoxBlur ring kernel leans on, per wave-instruction
// and SIMD (gfx950): v_mul_hi_u32 (the 8 divides-by-k of a row step) against the full-rate candidates that could
// replace it. One wave per SIMD slot, 8 independent chains per lane, inline asm so nothing is folded.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/valu_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ITER = 4096;

#define KERNEL(name, body)                                                                      \
    __global__ __launch_bounds__(64) void name(unsigned *out, unsigned m) {                     \
        unsigned a0 = threadIdx.x + 1, a1 = a0 + 7, a2 = a0 + 13, a3 = a0 + 29, a4 = a0 + 31, a5 = a0 + 37, a6 = a0 + 41, a7 = a0 + 43; \
        for (int i = 0; i < ITER; ++i) {                                                        \
            body(a0) body(a1) body(a2) body(a3) body(a4) body(a5) body(a6) body(a7)             \
        }                                                                                       \
        out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;             \
    }
#define OP_MULHI(a) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(m));
#define OP_MULLO(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(m));
#define OP_MUL24(a) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(m));
#define OP_MULHI24(a) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a) : "v"(m));
#define OP_MAD24(a) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a) : "v"(m));
#define OP_ADD(a) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(m));
#define OP_CVTF(a) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a));
#define OP_CVTU(a) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a));
#define OP_FMA(a) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(m));
#define OP_PERM(a) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(m));
#define OP_DIV3(a) asm volatile("v_cvt_f32_u32 %0, %0\n v_fma_f32 %0, %0, %1, %1\n v_cvt_u32_f32 %0, %0" : "+v"(a) : "v"(m));
KERNEL(k_mulhi, OP_MULHI)
KERNEL(k_mullo, OP_MULLO)
KERNEL(k_mul24, OP_MUL24)
KERNEL(k_mulhi24, OP_MULHI24)
KERNEL(k_mad24, OP_MAD24)
KERNEL(k_add, OP_ADD)
KERNEL(k_cvtf, OP_CVTF)
KERNEL(k_cvtu, OP_CVTU)
KERNEL(k_fma, OP_FMA)
KERNEL(k_perm, OP_PERM)
KERNEL(k_div3, OP_DIV3)

int main() {
    unsigned *out; CK(hipMalloc(&out, 1 << 24));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int waves_per_simd : {1, 3}) {
        const int blocks = 256 * 4 * waves_per_simd;  // one 64-thread block per SIMD slot
        auto run = [&](const char *name, auto kern, int ops_per_body) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 0x9E3779B9u);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, 0x9E3779B9u);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_simd = (double)ITER * 8 * ops_per_body * waves_per_simd;
            printf("%d wave(s)/SIMD  %-18s %8.1f us  -> %.2f cycles per wave-instruction at 2.4 GHz\n", waves_per_simd, name, ms * 1e3, ms * 1e-3 * 2.4e9 / instr_per_simd);
        };
        run("v_mul_hi_u32", k_mulhi, 1); run("v_mul_lo_u32", k_mullo, 1); run("v_mul_u32_u24", k_mul24, 1); run("v_mul_hi_u32_u24", k_mulhi24, 1);
        run("v_mad_u32_u24", k_mad24, 1); run("v_add_u32", k_add, 1); run("v_cvt_f32_u32", k_cvtf, 1); run("v_cvt_u32_f32", k_cvtu, 1);
        run("v_fma_f32", k_fma, 1); run("v_perm_b32", k_perm, 1); run("cvt+fma+cvt (3)", k_div3, 3);
    }
    return 0;
}
