// All 16 sets of vmm_hybrid_probe -- allocated before the process launched its first kernel -- stored at the fast rate, while pairs
// allocated between kernel runs (vmm_chunk_probe) mostly did not.  Is it WHEN the memory is allocated?  hipMalloc pairs allocated
// before the first launch, then pairs allocated after kernels have run, all kept; the store-only replay of the crowd pattern on each.
//   hipcc --offload-arch=gfx950 -O3 tools/archive/probes/alloc_before_first_kernel_probe.hip -o tools/archive/probes/alloc_before_first_kernel_probe
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

int main() {
    const size_t bytes = size_t(50000) * 1024 * 12;
    struct Pair { void *a, *b; };
    std::vector<Pair> early(10), late(10);
    for (auto &p : early) { CK(hipMalloc(&p.a, bytes)); CK(hipMalloc(&p.b, bytes)); }
    std::printf("allocated before the first kernel launch:");
    for (auto &p : early) { std::printf(" %5.0f", run(p.a, p.b)); std::fflush(stdout); }
    std::printf("  GB/s\nallocated afterwards, one pair at a time:  ");
    for (auto &p : late) {
        CK(hipMalloc(&p.a, bytes)); CK(hipMalloc(&p.b, bytes));
        std::printf(" %5.0f", run(p.a, p.b));
        std::fflush(stdout);
    }
    std::printf("  GB/s\nthe early pairs once more:                ");
    for (auto &p : early) { std::printf(" %5.0f", run(p.a, p.b)); std::fflush(stdout); }
    std::printf("  GB/s\nfree three early pairs, allocate three new:");
    for (int i = 0; i < 3; ++i) { CK(hipFree(early[i].a)); CK(hipFree(early[i].b)); }
    for (int i = 0; i < 3; ++i) {
        Pair p;
        CK(hipMalloc(&p.a, bytes)); CK(hipMalloc(&p.b, bytes));
        std::printf(" %5.0f", run(p.a, p.b));
        std::fflush(stdout);
    }
    std::printf("  GB/s\n");
    return 0;
}
