// GPU box: how close to the IEEE quotient is v_rcp_f64 followed by 0 / 1 / 2 Newton steps and one residual correction of the quotient,
// on the operand ranges of ssim_maps (denominators >= 0.0009 up to a few, numerators of either sign)?  hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <vector>
template <int NEWTON>
__device__ __forceinline__ double div_v(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
#pragma unroll
    for (int k = 0; k < NEWTON; ++k) r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ long ulps(double x, double y) {
    long a = __double_as_longlong(x), b = __double_as_longlong(y);
    if (a < 0) a = (long)0x8000000000000000L - a;
    if (b < 0) b = (long)0x8000000000000000L - b;
    return a > b ? a - b : b - a;
}
__global__ void probe(const double *a, const double *b, int n, unsigned long long *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double q = a[i] / b[i];
    const long e0 = ulps(div_v<0>(a[i], b[i]), q), e1 = ulps(div_v<1>(a[i], b[i]), q), e2 = ulps(div_v<2>(a[i], b[i]), q);
    const long er = ulps(a[i] * __builtin_amdgcn_rcp(b[i]), q);
    if (e0) atomicAdd(&out[0], 1ull);
    if (e1) atomicAdd(&out[1], 1ull);
    if (e2) atomicAdd(&out[2], 1ull);
    atomicMax(&out[3], (unsigned long long)e0);
    atomicMax(&out[4], (unsigned long long)e1);
    atomicMax(&out[5], (unsigned long long)e2);
    atomicMax(&out[6], (unsigned long long)er);
}
int main() {
    const int n = 1 << 24;
    std::vector<double> a(n), b(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return (double)(s >> 11) / 9007199254740992.0;
    };
    for (int i = 0; i < n; ++i) {
        const int kind = i & 3;
        if (kind == 0) { a[i] = (double)(float)(rnd() * 2 - 1) * (double)(float)(rnd()); b[i] = (double)(float)(0.0009 + rnd() * rnd() * 0.5); }       // ssim: num_m * num_s / denom_s
        else if (kind == 1) { a[i] = 1.0 + (double)(float)(rnd() * rnd()); b[i] = 1.0 + (double)(float)(rnd() * rnd()); }                              // edge: (1 + n2) / (1 + n1)
        else if (kind == 2) { a[i] = (double)(float)(rnd() * 1e-3) * (double)(float)(rnd() * 1e-3 + 0.0009); b[i] = (double)(float)(0.0009 + rnd() * 1e-4); }
        else { a[i] = (rnd() * 2 - 1) * exp2(rnd() * 40 - 20); b[i] = exp2(rnd() * 40 - 20); }                                                           // wide range
    }
    double *da, *db; unsigned long long *dout;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, 64);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice); hipMemset(dout, 0, 64);
    probe<<<n / 256, 256>>>(da, db, n, dout);
    unsigned long long o[8];
    hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost);
    printf("n = %d operand pairs\n", n);
    printf("quotients that differ from IEEE: newton0 %llu  newton1 %llu  newton2 %llu\n", o[0], o[1], o[2]);
    printf("largest difference in units of the last place: newton0 %llu  newton1 %llu  newton2 %llu   (a * rcp(b) alone: %llu)\n", o[3], o[4], o[5], o[6]);
    return 0;
}
