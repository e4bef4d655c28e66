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

// ---- does it matter WHICH XCD writes WHICH part of an allocation?  Workgroups are dealt to the eight XCDs round-robin by index:
// only those of one XCD stream-write one region; host times every (region, XCD) pair with events.
__global__ __launch_bounds__(256) void k_region_write(uint4* __restrict__ p, size_t n16, int xcd, uint32_t val) {
    if ((int)(blockIdx.x & 7u) != xcd && xcd >= 0) return;
    const size_t nb = xcd >= 0 ? gridDim.x / 8 : gridDim.x, b = xcd >= 0 ? blockIdx.x / 8 : blockIdx.x;
    for (size_t i = b * 256 + threadIdx.x; i < n16; i += nb * 256) p[i] = make_uint4(val, val, val, val);
}
__global__ __launch_bounds__(256) void k_region_read(const uint4* __restrict__ p, size_t n16, int xcd, uint32_t* sink) {
    if ((int)(blockIdx.x & 7u) != xcd && xcd >= 0) return;
    const size_t nb = xcd >= 0 ? gridDim.x / 8 : gridDim.x, b = xcd >= 0 ? blockIdx.x / 8 : blockIdx.x;
    uint32_t acc = 0;
    for (size_t i = b * 256 + threadIdx.x; i < n16; i += nb * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
extern "C" int region_matrix(void* base, size_t bytes, size_t region, int write, float* out_ms /* [n_regions][9]: XCD 0-7, then all */) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    uint32_t* sink = nullptr;
    if (hipMalloc(&sink, 4) != hipSuccess) return 2;
    const size_t nr = bytes / region;
    for (size_t r = 0; r < nr; ++r)
        for (int x = 0; x < 9; ++x) {
            uint4* p = (uint4*)((char*)base + r * region);
            const int xcd = x < 8 ? x : -1;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                if (write) hipLaunchKernelGGL(k_region_write, dim3(2048), dim3(256), 0, 0, p, region / 16, xcd, (uint32_t)rep);
                else hipLaunchKernelGGL(k_region_read, dim3(2048), dim3(256), 0, 0, (const uint4*)p, region / 16, xcd, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            out_ms[r * 9 + x] = best;
        }
    hipFree(sink); hipEventDestroy(e0); hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// the same range, its physical handles created in order and MAPPED IN A SHUFFLED ORDER: virtual neighbours are then physical strangers
extern "C" int vmm_alloc_shuffled(int device, size_t bytes, size_t chunk, uint64_t seed, void** out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return 1;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total, (size_t)2 << 20, nullptr, 0) != hipSuccess) return 2;
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (size_t i = 0; i < n; ++i) if (hipMemCreate(&h[i], chunk, &prop, 0) != hipSuccess) return 3;
    std::vector<size_t> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = i;
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    for (size_t i = n - 1; i > 0; --i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(perm[i], perm[(size_t)(x % (i + 1))]); }
    for (size_t i = 0; i < n; ++i) {
        if (hipMemMap((char*)va + i * chunk, chunk, 0, h[perm[i]], 0) != hipSuccess) return 4;
        (void)hipMemRelease(h[perm[i]]);
    }
    hipMemAccessDesc ad = {};
    ad.location.type = hipMemLocationTypeDevice; ad.location.id = device; ad.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, total, &ad, 1) != hipSuccess) return 5;
    *out = va;
    return 0;
}
