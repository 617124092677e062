// vmm_hybrid_probe: 16 of 16 sets of 8 MiB chunks created before the first kernel launch store at the fast rate; vmm_chunk_probe:
// 8 MiB-chunk pairs created between kernel runs (fresh address range each, handles released after mapping) mostly do not.  Which
// difference matters?  Sets created before / after the first launch, mapped under fixed or fresh address ranges, handles kept or
// released.   hipcc --offload-arch=gfx950 -O3 tools/archive/probes/vmm_when_probe.hip -o tools/archive/probes/vmm_when_probe
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

using Handles = std::vector<hipMemGenericAllocationHandle_t>;
hipMemAllocationProp g_prop;
hipMemAccessDesc g_acc;
constexpr size_t kChunk = size_t(8) << 20;
const size_t kBytes = size_t(50000) * 1024 * 12, kPer = (kBytes + kChunk - 1) / kChunk;

Handles create_set() {
    Handles h(2 * kPer);
    for (auto &x : h) CK(hipMemCreate(&x, kChunk, &g_prop, 0));
    return h;
}
void map_set(const Handles &h, void *va, void *vb) {
    for (size_t i = 0; i < kPer; ++i) {
        CK(hipMemMap(static_cast<char *>(va) + i * kChunk, kChunk, 0, h[i], 0));
        CK(hipMemMap(static_cast<char *>(vb) + i * kChunk, kChunk, 0, h[kPer + i], 0));
    }
    CK(hipMemSetAccess(va, kPer * kChunk, &g_acc, 1));
    CK(hipMemSetAccess(vb, kPer * kChunk, &g_acc, 1));
}
float rate_fixed(const Handles &h, void *va, void *vb) {
    map_set(h, va, vb);
    const float r = run(va, vb);
    CK(hipMemUnmap(va, kPer * kChunk));
    CK(hipMemUnmap(vb, kPer * kChunk));
    return r;
}
float rate_fresh(const Handles &h, bool release) {     // fresh address ranges, kept mapped (chunk-probe style)
    void *va = nullptr, *vb = nullptr;
    CK(hipMemAddressReserve(&va, kPer * kChunk, 0, nullptr, 0));
    CK(hipMemAddressReserve(&vb, kPer * kChunk, 0, nullptr, 0));
    map_set(h, va, vb);
    if (release) for (auto x : h) CK(hipMemRelease(x));
    return run(va, vb);
}

int main() {
    g_prop = {};
    g_prop.type = hipMemAllocationTypePinned;
    g_prop.location.type = hipMemLocationTypeDevice;
    g_prop.location.id = 0;
    g_acc = {};
    g_acc.location = g_prop.location;
    g_acc.flags = hipMemAccessFlagsProtReadWrite;
    void *va = nullptr, *vb = nullptr;
    CK(hipMemAddressReserve(&va, kPer * kChunk, 0, nullptr, 0));
    CK(hipMemAddressReserve(&vb, kPer * kChunk, 0, nullptr, 0));
    std::vector<Handles> early;
    for (int s = 0; s < 6; ++s) early.push_back(create_set());
    std::printf("created before the first launch, fixed ranges:         ");
    for (int s = 0; s < 3; ++s) { std::printf(" %5.0f", rate_fixed(early[s], va, vb)); std::fflush(stdout); }
    std::printf("\ncreated before the first launch, fresh ranges, kept:   ");
    for (int s = 3; s < 5; ++s) { std::printf(" %5.0f", rate_fresh(early[s], false)); std::fflush(stdout); }
    std::printf("\ncreated before the first launch, fresh ranges, released:");
    for (int s = 5; s < 6; ++s) { std::printf(" %5.0f", rate_fresh(early[s], true)); std::fflush(stdout); }
    std::printf("\ncreated now (kernels have run), fixed ranges:          ");
    std::vector<Handles> late;
    for (int s = 0; s < 4; ++s) { late.push_back(create_set()); std::printf(" %5.0f", rate_fixed(late.back(), va, vb)); std::fflush(stdout); }
    std::printf("\ncreated now, fresh ranges, kept:                       ");
    for (int s = 0; s < 3; ++s) { late.push_back(create_set()); std::printf(" %5.0f", rate_fresh(late.back(), false)); std::fflush(stdout); }
    std::printf("\ncreated now, fresh ranges, released:                   ");
    for (int s = 0; s < 3; ++s) { Handles h = create_set(); std::printf(" %5.0f", rate_fresh(h, true)); std::fflush(stdout); }
    std::printf("\nhipMalloc pairs now:                                   ");
    for (int s = 0; s < 4; ++s) { void *a, *b; CK(hipMalloc(&a, kBytes)); CK(hipMalloc(&b, kBytes)); std::printf(" %5.0f", run(a, b)); std::fflush(stdout); }
    std::printf("\ncreated now, chunk by chunk interleaved a/b, fixed:    ");
    for (int s = 0; s < 3; ++s) {
        Handles h(2 * kPer);
        for (size_t i = 0; i < kPer; ++i) { CK(hipMemCreate(&h[i], kChunk, &g_prop, 0)); CK(hipMemCreate(&h[kPer + i], kChunk, &g_prop, 0)); }
        std::printf(" %5.0f", rate_fixed(h, va, vb)); std::fflush(stdout);
        late.push_back(h);
    }
    std::printf("\nfixed ranges: a=%p b=%p\n", va, vb);
    // fresh ranges again, with an alignment asked for
    for (size_t align : {size_t(0), size_t(2) << 20, size_t(1) << 30, size_t(4) << 30}) {
        std::printf("created now, fresh ranges aligned to %5zu MiB:", align >> 20);
        for (int s = 0; s < 3; ++s) {
            Handles h = create_set();
            void *xa = nullptr, *xb = nullptr;
            CK(hipMemAddressReserve(&xa, kPer * kChunk, align, nullptr, 0));
            CK(hipMemAddressReserve(&xb, kPer * kChunk, align, nullptr, 0));
            map_set(h, xa, xb);
            std::printf("  %5.0f (a=%p b=%p)", run(xa, xb), xa, xb);
            std::fflush(stdout);
            late.push_back(h);
        }
        std::printf("\n");
    }
    // the fixed ranges once more with a set created now
    { Handles h = create_set(); std::printf("fixed ranges, set created last: %5.0f\n", rate_fixed(h, va, vb)); late.push_back(h); }
    return 0;
}
