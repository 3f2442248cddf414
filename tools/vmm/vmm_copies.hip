// Which runtime copies accept a range that spans several hipMemMap'ed handles? One operation per process (some crash).
//   ./vmm_copies.bin <chunk MiB> <op 0..7>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv) {
    const long long MiB = 1 << 20, C = atoll(argv[1]) * MiB, A = 256 * MiB;
    const int op = atoi(argv[2]);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void *p = nullptr;
    CK(hipMemAddressReserve(&p, A, 2 * MiB, nullptr, 0));
    for (long long o = 0; o < A; o += C) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, C, &prop, 0));
        CK(hipMemMap((char *)p + o, C, 0, h, 0));
    }
    CK(hipMemSetAccess(p, A, &acc, 1));
    char *d = (char *)p, *host, *plain;
    const size_t pb = 7680 * 2160;
    CK(hipHostMalloc(&host, pb, 0));
    CK(hipMalloc(&plain, pb));
    memset(host, 0x5a, pb);
    char *off = d + 64 * MiB - 3 * MiB - 256;  // a 16.6 MB plane starting 3 MiB before a 64 MiB boundary (and before a boundary of every smaller chunk size)
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const char *names[] = {"memsetAsync", "2D H2D", "2D D2H", "2D D2D from plain", "2D H2D pitch 8192", "1D H2D", "1D D2H", "2D D2D arena to arena", "verify 2D H2D then D2H"};
    hipError_t e = hipSuccess;
    switch (op) {
    case 0: e = hipMemsetAsync(off, 0, pb, s); break;
    case 1: e = hipMemcpy2DAsync(off, 7680, host, 7680, 7680, 2160, hipMemcpyHostToDevice, s); break;
    case 2: e = hipMemcpy2DAsync(host, 7680, off, 7680, 7680, 2160, hipMemcpyDeviceToHost, s); break;
    case 3: e = hipMemcpy2DAsync(off, 7680, plain, 7680, 7680, 2160, hipMemcpyDeviceToDevice, s); break;
    case 4: e = hipMemcpy2DAsync(off, 8192, host, 7680, 7680, 2000, hipMemcpyHostToDevice, s); break;
    case 5: e = hipMemcpyAsync(off, host, pb, hipMemcpyHostToDevice, s); break;
    case 6: e = hipMemcpyAsync(host, off, pb, hipMemcpyDeviceToHost, s); break;
    case 7: e = hipMemcpy2DAsync(off + 100 * MiB, 7680, off, 7680, 7680, 2160, hipMemcpyDeviceToDevice, s); break;
    case 8: {
        for (size_t i = 0; i < pb; ++i) host[i] = (char)(i * 2654435761u >> 13);
        e = hipMemcpy2DAsync(off, 7680, host, 7680, 7680, 2160, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        std::vector<char> keep(host, host + pb);
        memset(host, 0, pb);
        if (e == hipSuccess) e = hipMemcpy2DAsync(host, 7680, off, 7680, 7680, 2160, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess && memcmp(keep.data(), host, pb) != 0) { printf("C=%lld %s: DATA MISMATCH\n", C / MiB, names[op]); return 1; }
        break;
    }
    }
    const hipError_t e2 = hipStreamSynchronize(s);
    printf("C=%3lld MiB %-24s: %s / sync %s\n", C / MiB, names[op], hipGetErrorString(e), hipGetErrorString(e2));
    return 0;
}
