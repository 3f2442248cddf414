// Experiment helper (tools/vmm_alias.py): one physical allocation mapped at several virtual addresses, to tell whether
// the BoxBlur ring kernel's sensitivity to "where the destination arena lies" follows the VIRTUAL address (TLB) or the
// PHYSICAL one (channel / bank hash). hipcc -shared -fPIC --offload-arch=gfx950 -o tools/vmm/libvmm_alias.so tools/vmm/vmm_alias.hip
#include <hip/hip_runtime.h>

#include <cstdio>

#define EXP extern "C" __attribute__((visibility("default")))

static hipMemAllocationProp prop_for(int device) {
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = device;
    return p;
}

EXP size_t vmm_granularity(int device) {
    const hipMemAllocationProp p = prop_for(device);
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &p, hipMemAllocationGranularityRecommended) != hipSuccess) return 0;
    return g;
}

// physical memory: bytes must be a multiple of the granularity
EXP int vmm_create(int device, size_t bytes, void **handle) {
    const hipMemAllocationProp p = prop_for(device);
    hipMemGenericAllocationHandle_t h;
    const hipError_t e = hipMemCreate(&h, bytes, &p, 0);
    if (e != hipSuccess) {
        fprintf(stderr, "hipMemCreate: %s\n", hipGetErrorString(e));
        return (int)e;
    }
    *handle = (void *)h;
    return 0;
}

// a new virtual range backed by that physical memory; `skew` bytes (a multiple of the granularity) shift the mapping
// inside a larger reservation, so that virtual and physical addresses disagree modulo 2 MiB (small PTE fragments only)
EXP int vmm_map(int device, void *handle, size_t bytes, size_t alignment, size_t skew, void **va) {
    void *p = nullptr;
    hipError_t e = hipMemAddressReserve(&p, bytes + skew, alignment, nullptr, 0);
    if (e != hipSuccess) {
        fprintf(stderr, "hipMemAddressReserve: %s\n", hipGetErrorString(e));
        return (int)e;
    }
    p = (char *)p + skew;
    e = hipMemMap(p, bytes, 0, (hipMemGenericAllocationHandle_t)handle, 0);
    if (e != hipSuccess) {
        fprintf(stderr, "hipMemMap: %s\n", hipGetErrorString(e));
        return (int)e;
    }
    hipMemAccessDesc d = {};
    d.location.type = hipMemLocationTypeDevice;
    d.location.id = device;
    d.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(p, bytes, &d, 1);
    if (e != hipSuccess) {
        fprintf(stderr, "hipMemSetAccess: %s\n", hipGetErrorString(e));
        return (int)e;
    }
    *va = p;
    return 0;
}
