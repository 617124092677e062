// vmm_shuffle_probe showed that a pair's store mode belongs to the SET of physical chunks behind it.  Is it additive -- does
// every chunk carry its own share -- or a property of the set as a whole?  A pool of 8 MiB chunks is cut into sets of 148 (one
// pair each), every set is rated; then hybrids of the fastest and the slowest set: the first k chunk positions take their chunk
// from the slow set, for k = 0 ... 148.   hipcc --offload-arch=gfx950 -O3 tools/archive/probes/vmm_hybrid_probe.hip -o tools/archive/probes/vmm_hybrid_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
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

int main(int argc, char **argv) {
    const int nsets = argc > 1 ? std::atoi(argv[1]) : 16;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t chunk = size_t(8) << 20, bytes = size_t(50000) * 1024 * 12, per = (bytes + chunk - 1) / chunk, set_n = 2 * per;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void *va = nullptr, *vb = nullptr;
    CK(hipMemAddressReserve(&va, per * chunk, 0, nullptr, 0));
    CK(hipMemAddressReserve(&vb, per * chunk, 0, nullptr, 0));
    // a spacer between the sets (kept): hipMalloc pairs, so that the pool is not one young, contiguous range
    std::vector<std::vector<hipMemGenericAllocationHandle_t>> sets(nsets);
    for (int s = 0; s < nsets; ++s) {
        sets[s].resize(set_n);
        for (auto &x : sets[s]) CK(hipMemCreate(&x, chunk, &prop, 0));
        void *spacer;
        CK(hipMalloc(&spacer, size_t(700) << 20));
    }
    auto rate = [&](const std::vector<hipMemGenericAllocationHandle_t> &h) {
        for (size_t i = 0; i < per; ++i) {
            CK(hipMemMap(static_cast<char *>(va) + i * chunk, chunk, 0, h[i], 0));
            CK(hipMemMap(static_cast<char *>(vb) + i * chunk, chunk, 0, h[per + i], 0));
        }
        CK(hipMemSetAccess(va, per * chunk, &acc, 1));
        CK(hipMemSetAccess(vb, per * chunk, &acc, 1));
        const float r = run(va, vb);
        CK(hipMemUnmap(va, per * chunk));
        CK(hipMemUnmap(vb, per * chunk));
        return r;
    };
    std::vector<float> r(nsets);
    for (int s = 0; s < nsets; ++s) {
        r[s] = rate(sets[s]);
        std::printf("set %2d: %6.0f GB/s\n", s, r[s]);
        std::fflush(stdout);
    }
    const int f = int(std::max_element(r.begin(), r.end()) - r.begin()), w = int(std::min_element(r.begin(), r.end()) - r.begin());
    std::printf("fastest set %d (%.0f), slowest set %d (%.0f)\n", f, r[f], w, r[w]);
    if (r[f] < 1.1f * r[w]) { std::printf("no spread between the sets of this process: nothing to mix\n"); return 0; }
    for (size_t k : {size_t(0), size_t(8), size_t(18), size_t(37), size_t(74), size_t(111), size_t(148)}) {
        std::vector<hipMemGenericAllocationHandle_t> h = sets[f];
        // positions 0..k-1 (array a first, then b) take the slow set's chunk of the same position
        for (size_t i = 0; i < k; ++i) h[i] = sets[w][i];
        std::printf("hybrid: first %3zu of 148 positions from the slow set: %6.0f GB/s\n", k, rate(h));
        std::fflush(stdout);
    }
    for (size_t k : {size_t(18), size_t(74)}) {               // the same counts, but every 148/k-th position instead of the first k
        std::vector<hipMemGenericAllocationHandle_t> h = sets[f];
        for (size_t j = 0; j < k; ++j) { const size_t i = j * set_n / k; h[i] = sets[w][i]; }
        std::printf("hybrid: %3zu positions spread evenly from the slow set:  %6.0f GB/s\n", k, rate(h));
        std::fflush(stdout);
    }
    return 0;
}
