// Does a large by-value kernel argument (> 4 KiB) work on this stack?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { unsigned v[N]; };
template <int N> __global__ void k(Big<N> b, unsigned *out) { if (threadIdx.x == 0 && blockIdx.x == 0) { unsigned s = 0; for (int i = 0; i < N; ++i) s += b.v[i]; *out = s; } }
template <int N> void run(unsigned *d) {
    Big<N> b; for (int i = 0; i < N; ++i) b.v[i] = i;
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
    hipError_t e = hipGetLastError(); unsigned h = 0; hipError_t e2 = hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("N=%d bytes=%d launch=%s copy=%s sum=%u expect=%u\n", N, N * 4, hipGetErrorString(e), hipGetErrorString(e2), h, (unsigned)(N * (N - 1) / 2));
}
int main() { unsigned *d; hipMalloc(&d, 4); run<512>(d); run<1023>(d); run<2048>(d); run<4096>(d); run<8192>(d); return 0; }
