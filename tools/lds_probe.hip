#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float sm[];
__global__ void k(float *out, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) sm[i] = (float)i;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sm[n - 1] + sm[n / 2];
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    float *d;
    hipMalloc(&d, 1024);
    for (size_t kb : {48, 64, 96, 128, 150, 160}) {
        size_t bytes = kb * 1024;
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL(k, dim3(4), dim3(1024), bytes, 0, d, (int)(bytes / 4));
        hipError_t e2 = hipDeviceSynchronize();
        hipError_t e3 = hipGetLastError();
        float h[4] = {0};
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%zu KB: setattr %s, sync %s, last %s, out %.0f\n", kb, hipGetErrorString(e), hipGetErrorString(e2), hipGetErrorString(e3), h[0]);
    }
    return 0;
}
