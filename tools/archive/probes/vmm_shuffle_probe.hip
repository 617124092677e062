// Is the slow / fast store mode of a pair of crowd output arrays a property of WHICH physical memory backs them, or of HOW that
// memory is arranged under the virtual addresses?  One set of 2 MiB physical chunks (hipMemCreate), mapped under the two arrays
// in order, in random permutations, and dealt alternately to the two arrays; the store-only replay of the crowd pattern on each.
//   hipcc --offload-arch=gfx950 -O3 tools/archive/probes/vmm_shuffle_probe.hip -o tools/archive/probes/vmm_shuffle_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
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
        const uint32_t g = k * ngroups + grp;               // interleaved instance order, like the deform kernel
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
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t chunk = std::max<size_t>(gran, size_t(8) << 20), bytes = size_t(50000) * 1024 * 12;   // 8 MiB chunks: 74 per array
    const size_t per = (bytes + chunk - 1) / chunk;          // chunks per array
    std::printf("granularity %zu, chunk %zu, %zu chunks per array\n", gran, chunk, per); std::fflush(stdout);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::mt19937 rng(7);
    for (int set = 0; set < 3; ++set) {                      // four sets of physical chunks (each kept until its trials are done)
        std::vector<hipMemGenericAllocationHandle_t> h(2 * per);
        for (auto &x : h) CK(hipMemCreate(&x, chunk, &prop, 0));
        std::printf("set %d: %zu chunks created\n", set, h.size()); std::fflush(stdout);
        void *va = nullptr, *vb = nullptr;
        CK(hipMemAddressReserve(&va, per * chunk, 0, nullptr, 0));
        CK(hipMemAddressReserve(&vb, per * chunk, 0, nullptr, 0));
        auto trial = [&](const std::vector<size_t> &order, const char *what) {
            for (size_t i = 0; i < per; ++i) {
                CK(hipMemMap(static_cast<char *>(va) + i * chunk, chunk, 0, h[order[i]], 0));
                CK(hipMemMap(static_cast<char *>(vb) + i * chunk, chunk, 0, h[order[per + i]], 0));
            }
            CK(hipMemSetAccess(va, per * chunk, &acc, 1));
            CK(hipMemSetAccess(vb, per * chunk, &acc, 1));
            const float r = run(va, vb);
            std::printf("  set %d  %-44s %6.0f GB/s\n", set, what, r); std::fflush(stdout);
            CK(hipMemUnmap(va, per * chunk));
            CK(hipMemUnmap(vb, per * chunk));
        };
        std::vector<size_t> order(2 * per);
        std::iota(order.begin(), order.end(), size_t(0));
        trial(order, "in allocation order (a: first half)");
        trial(order, "the same again");
        std::vector<size_t> alt(2 * per);
        for (size_t i = 0; i < per; ++i) { alt[i] = 2 * i; alt[per + i] = 2 * i + 1; }
        trial(alt, "dealt alternately to a and b");
        std::vector<size_t> rev(order.rbegin(), order.rend());
        trial(rev, "reversed");
        for (int k = 0; k < 4; ++k) {
            std::shuffle(order.begin(), order.end(), rng);
            trial(order, "random permutation");
        }
        std::iota(order.begin(), order.end(), size_t(0));
        trial(order, "in allocation order once more");
        CK(hipMemAddressFree(va, per * chunk));
        CK(hipMemAddressFree(vb, per * chunk));
        // the sets are kept alive: the next set draws other physical memory
    }
    return 0;
}
