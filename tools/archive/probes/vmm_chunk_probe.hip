// Follow-up of vmm_shuffle_probe: every arrangement of 8 MiB hipMemCreate chunks stored at the FAST rate.  Is that the chunk size,
// the API, or the box?  Same process, interleaved: plain hipMalloc pairs and virtual-memory pairs of several chunk sizes, all kept
// allocated (every trial draws fresh physical memory), the store-only replay of the crowd pattern on each.
//   hipcc --offload-arch=gfx950 -O3 tools/archive/probes/vmm_chunk_probe.hip -o tools/archive/probes/vmm_chunk_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } \
    } while (0)

constexpr uint32_t kThreads = 256, kTile = 512;
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kThreads) void pattern_fill(v4f *a, v4f *b, uint32_t nv, uint32_t ni, uint32_t ntiles, uint32_t ngroups) {
    const uint32_t tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    const uint32_t v0 = tile * kTile, nvt = min(kTile, nv - v0);
    const uint32_t piece4 = nvt * 12 / 16;
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (uint32_t k = 0; k < 16; ++k) {
        const uint32_t g = k * ngroups + grp;
        if (g >= ni) break;
        const size_t base = (size_t(g) * nv + v0) * 12 / 16;
        for (uint32_t q = threadIdx.x; q < 2 * piece4; q += kThreads) {
            if (q < piece4) __builtin_nontemporal_store(v, a + base + q); else __builtin_nontemporal_store(v, b + base + q - piece4);
        }
    }
}

float run(void *a, void *b) {
    const uint32_t nv = 50000, ni = 1024, ntiles = (nv + kTile - 1) / kTile, ngroups = ni / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) pattern_fill<<<ntiles * ngroups, kThreads>>>((v4f *)a, (v4f *)b, nv, ni, ntiles, ngroups);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) pattern_fill<<<ntiles * ngroups, kThreads>>>((v4f *)a, (v4f *)b, nv, ni, ntiles, ngroups);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 2 * 614.4e6 / (ms / 10 * 1e-3) / 1e9;
}

void *vmm_array(size_t bytes, size_t chunk) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    if (chunk == 0) chunk = (bytes + (size_t(2) << 20) - 1) / (size_t(2) << 20) * (size_t(2) << 20);
    const size_t n = (bytes + chunk - 1) / chunk;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0));
    for (size_t i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap(static_cast<char *>(va) + i * chunk, chunk, 0, h, 0));
        CK(hipMemRelease(h));                                  // the mapping keeps the memory
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, n * chunk, &acc, 1));
    return va;
}

int main() {
    const size_t bytes = size_t(50000) * 1024 * 12;
    const size_t sizes[] = {size_t(2) << 20, size_t(8) << 20, size_t(32) << 20, size_t(128) << 20, 0};
    const char *names[] = {"vmm 2 MiB", "vmm 8 MiB", "vmm 32 MiB", "vmm 128 MiB", "vmm whole"};
    const int first_flavour = 3;                              // this run: 128 MiB chunks and whole-array chunks only
    for (int t = 0; t < 10; ++t) {
        void *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        std::printf("trial %d: hipMalloc %6.0f", t, run(a, b));
        for (int k = first_flavour; k < 5; ++k) {
            void *va = vmm_array(bytes, sizes[k]), *vb = vmm_array(bytes, sizes[k]);
            std::printf("   %s %6.0f", names[k], run(va, vb));
            std::fflush(stdout);
        }
        {   // physically contiguous backing asked for explicitly: two allocations, then one allocation split in two
            void *ca = nullptr, *cb = nullptr, *cc = nullptr;
            hipError_t e1 = hipExtMallocWithFlags(&ca, bytes, hipDeviceMallocContiguous);
            hipError_t e2 = hipExtMallocWithFlags(&cb, bytes, hipDeviceMallocContiguous);
            if (e1 == hipSuccess && e2 == hipSuccess) std::printf("   contiguous x2 %6.0f", run(ca, cb));
            else { std::printf("   contiguous x2 failed (%s)", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); (void)hipGetLastError(); }
            const size_t half = (bytes + 4095) / 4096 * 4096;
            hipError_t e3 = hipExtMallocWithFlags(&cc, 2 * half, hipDeviceMallocContiguous);
            if (e3 == hipSuccess) std::printf("   contiguous 1 block %6.0f", run(cc, static_cast<char *>(cc) + half));
            else { std::printf("   contiguous 1 block failed (%s)", hipGetErrorString(e3)); (void)hipGetLastError(); }
            std::fflush(stdout);
        }
        std::printf("  GB/s\n");
        std::fflush(stdout);
        // nothing is freed: the next trial draws other physical memory (8 x 6 pairs x 1.23 GB = 59 GB)
    }
    return 0;
}
