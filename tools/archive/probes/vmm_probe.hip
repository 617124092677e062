// How does the backing of the two crowd output arrays (614.4 MB each) affect the store-only replay of
// the deform kernel's output pattern?  hipMalloc vs the virtual-memory API with physical chunks of a
// chosen size mapped into one contiguous range.   hipcc --offload-arch=gfx950 -O3 tools/archive/probes/vmm_probe.hip -o tools/archive/probes/vmm_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } \
    } while (0)

constexpr uint32_t kThreads = 256, kTile = 512;

__global__ __launch_bounds__(kThreads) void pattern_fill(float4 *a, float4 *b, uint32_t nv, uint32_t ni, uint32_t ntiles,
                                                         uint32_t ngroups) {
    const uint32_t tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    const uint32_t v0 = tile * kTile, nvt = min(kTile, nv - v0);
    const uint32_t piece4 = nvt * 12 / 16;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (uint32_t k = 0; k < 16; ++k) {
        const uint32_t g = k * ngroups + grp;               // interleaved instance order, like the deform kernel
        if (g >= ni) break;
        const size_t base = (size_t(g) * nv + v0) * 12 / 16;
        for (uint32_t q = threadIdx.x; q < 2 * piece4; q += kThreads) {
            if (q < piece4) { if (a) a[base + q] = v; } else if (b) b[base + q - piece4] = v;
        }
    }
}

struct Mapping {
    void *ptr = nullptr;
    size_t size = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

Mapping vmm_alloc(size_t bytes, size_t chunk) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (chunk == 0) chunk = (bytes + gran - 1) / gran * gran;
    chunk = (chunk + gran - 1) / gran * gran;
    Mapping m;
    m.size = (bytes + chunk - 1) / chunk * chunk;
    CK(hipMemAddressReserve(&m.ptr, m.size, 0, nullptr, 0));
    for (size_t off = 0; off < m.size; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap(static_cast<char *>(m.ptr) + off, chunk, 0, h, 0));
        m.handles.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(m.ptr, m.size, &acc, 1));
    return m;
}

void vmm_free(Mapping &m) {
    CK(hipMemUnmap(m.ptr, m.size));
    for (auto h : m.handles) CK(hipMemRelease(h));
    CK(hipMemAddressFree(m.ptr, m.size));
}

float run(void *a, void *b) {
    const uint32_t nv = 50000, ni = 1024, ntiles = (nv + kTile - 1) / kTile, ngroups = ni / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) pattern_fill<<<ntiles * ngroups, kThreads>>>((float4 *)a, (float4 *)b, nv, ni, ntiles, ngroups);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) pattern_fill<<<ntiles * ngroups, kThreads>>>((float4 *)a, (float4 *)b, nv, ni, ntiles, ngroups);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ((a ? 1.0 : 0.0) + (b ? 1.0 : 0.0)) * 614.4e6 / (ms / 10 * 1e-3) / 1e9;
}

int main() {
    const size_t bytes = size_t(50000) * 1024 * 12 + (2 << 20);   // last tile of the last instance stays inside
    size_t gran = 0;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    std::printf("recommended granularity %zu\n", gran);
    // hipExtMallocWithFlags(hipDeviceMallocContiguous): physically contiguous backing.  Interleaved with plain
    // hipMalloc trials in the same (increasingly churned) process.
    for (int t = 0; t < 12; ++t) {
        void *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        const float plain = run(a, b);
        CK(hipFree(a)); CK(hipFree(b));
        hipError_t e1 = hipExtMallocWithFlags(&a, bytes, hipDeviceMallocContiguous);
        hipError_t e2 = hipExtMallocWithFlags(&b, bytes, hipDeviceMallocContiguous);
        if (e1 != hipSuccess || e2 != hipSuccess) { std::printf("contiguous alloc failed: %s %s\n", hipGetErrorString(e1), hipGetErrorString(e2)); return 1; }
        const float contig = run(a, b);
        CK(hipFree(a)); CK(hipFree(b));
        std::printf("trial %2d: hipMalloc %6.0f GB/s   contiguous %6.0f GB/s\n", t, plain, contig);
    }
    return 0;
}
