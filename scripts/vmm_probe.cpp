// Measurement helper (not part of the product): device allocations through HIP's virtual-memory API, to see whether the physical
// backing of the read-word array explains why the walk's time differs between hipMalloc allocations.
// build: hipcc -O2 -fPIC -shared -o scripts/libvmm_probe.so scripts/vmm_probe.cpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

extern "C" {

// granularity of device allocations: min and recommended
int vmm_granularity(int device, size_t* gmin, size_t* grec) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    if (hipMemGetAllocationGranularity(gmin, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return 1;
    if (hipMemGetAllocationGranularity(grec, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return 2;
    return 0;
}

// `bytes` of device memory behind one virtual range aligned to `va_align`, backed by physical handles of `chunk` bytes each
// (0: one handle for everything).  -> 0 and the pointer, or the failing step
int vmm_alloc(int device, size_t bytes, size_t chunk, size_t va_align, void** out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return 1;
    if (chunk == 0) chunk = bytes;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t total = (bytes + chunk - 1) / chunk * chunk;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total, va_align, nullptr, 0) != hipSuccess) return 2;
    for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, chunk, &prop, 0);
        if (e != hipSuccess) { fprintf(stderr, "hipMemCreate(%zu): %s\n", chunk, hipGetErrorString(e)); return 3; }
        if (hipMemMap((char*)va + off, chunk, 0, h, 0) != hipSuccess) return 4;
        (void)hipMemRelease(h);                                  // (the mapping keeps it alive)
    }
    hipMemAccessDesc ad = {};
    ad.location.type = hipMemLocationTypeDevice; ad.location.id = device; ad.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, total, &ad, 1) != hipSuccess) return 5;
    *out = va;
    return 0;
}

int plain_alloc(size_t bytes, void** out) { return hipMalloc(out, bytes) == hipSuccess ? 0 : 1; }
int plain_free(void* p) { return hipFree(p) == hipSuccess ? 0 : 1; }
int mem_info(size_t* fr, size_t* tot) { return hipMemGetInfo(fr, tot) == hipSuccess ? 0 : 1; }

}  // extern "C"
