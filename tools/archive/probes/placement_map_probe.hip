// tools/archive/probes/placement_map_probe.hip -- what distinguishes the placements of the crowd's two output arrays that store at the
// linear-fill rate from the ones that store ~25 % slower (tools/archive/probes/alloc_api_probe.hip, profiles/r02/placement_probes.txt)?
// Per pair of fresh allocations: each array alone, the pair, the pair with the tile -> XCD assignment rotated, the pair
// quantised to whole 4 KiB blocks, two lock-step linear fills, and the pair restricted to eighths of the instance range.
// Measurement tool only.   hipcc --offload-arch=gfx950 -O2 tools/archive/probes/placement_map_probe.hip -o tools/archive/probes/placement_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

struct P {
    float4 *a, *b;
    uint32_t nv, ni, ntiles, ngroups;
    uint32_t inst0;     // first instance of the range written
    uint32_t rot;       // rotation of the XCD id in the tile assignment
    uint32_t which;     // 1 = array a, 2 = array b, 3 = both
    uint32_t quant;     // 1 = write only whole 4 KiB blocks that START inside the piece
    uint32_t chunk;     // groups per interleave chunk: instance = (grp / chunk) * chunk * 16 + j * chunk + grp % chunk (ngroups = full interleave)
};

// the deform kernel's mapping: XCD x owns a contiguous range of tiles, instances interleaved over the groups
__global__ __launch_bounds__(256) void pattern(const P p) {
    const uint32_t xcd = ((blockIdx.x & 7u) + p.rot) & 7u, k = blockIdx.x >> 3, T = p.ntiles >> 3, main_count = T * p.ngroups;
    uint32_t tile, grp;
    if (k < main_count) { grp = k / T; tile = xcd * T + (k - grp * T); }
    else {
        const uint32_t rem = ((p.ntiles & 7u) * p.ngroups + 7u) / 8u, r = xcd * rem + (k - main_count);
        if (r >= (p.ntiles & 7u) * p.ngroups) return;
        const uint32_t rt = r / p.ngroups; tile = 8u * T + rt; grp = r - rt * p.ngroups;
    }
    const uint32_t v0 = tile * 512, nvt = min(512u, p.nv - v0);
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (uint32_t j = 0; j < 16; ++j) {
        const uint32_t g = (grp / p.chunk) * p.chunk * 16 + j * p.chunk + grp % p.chunk;
        if (g >= p.ni) continue;
        size_t lo = (size_t(p.inst0 + g) * p.nv + v0) * 12 / 16, hi = lo + nvt * 12 / 16;      // float4 units
        if (p.quant) { lo = (lo + 255) / 256 * 256; hi = (hi + 255) / 256 * 256; }             // 4 KiB = 256 float4
        const uint32_t n = uint32_t(hi - lo);
        if (p.which == 3) {
            for (uint32_t q = threadIdx.x; q < 2 * n; q += 256) { if (q < n) p.a[lo + q] = v; else p.b[lo + q - n] = v; }
        } else {
            float4 *d = p.which == 1 ? p.a : p.b;
            for (uint32_t q = threadIdx.x; q < n; q += 256) d[lo + q] = v;
        }
    }
}
__global__ __launch_bounds__(256) void fill(float4 *d, size_t n) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) d[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void fill2(float4 *a, float4 *b, size_t n) {     // chunk i of a, then chunk i of b
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) { a[i] = make_float4(1.f, 2.f, 3.f, 4.f); b[i] = make_float4(1.f, 2.f, 3.f, 4.f); }
}
template <typename F> float timeit(F f, int iters = 6) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters;
}
int main(int argc, char **argv) {
    const uint32_t nv = 50000, ni = 1024, ntiles = 98;
    const size_t arr = size_t(ni) * nv * 12;
    const int trials = argc > 1 ? atoi(argv[1]) : 8;
    const int keep = argc > 2 ? atoi(argv[2]) : 0;          // 1: never free (each trial gets new physical memory)
    std::vector<void *> kept;
    // settle clocks
    { float4 *w; CK(hipMalloc(&w, arr)); for (int i = 0; i < 40; ++i) fill<<<unsigned((arr / 16 + 255) / 256), 256>>>(w, arr / 16); CK(hipDeviceSynchronize()); CK(hipFree(w)); }
    for (int t = 0; t < trials; ++t) {
        float4 *a, *b; CK(hipMalloc(&a, arr + 8192)); CK(hipMalloc(&b, arr + 8192));
        auto rate = [&](uint32_t n_inst, uint32_t inst0, uint32_t rot, uint32_t which, uint32_t quant, uint32_t chunk = 0) {
            P p{a, b, nv, n_inst, ntiles, (n_inst + 15) / 16, inst0, rot, which, quant, chunk ? chunk : (n_inst + 15) / 16};
            const unsigned grid = 8u * ((ntiles >> 3) * p.ngroups + ((ntiles & 7u) * p.ngroups + 7u) / 8u);
            const float tp = timeit([&] { pattern<<<grid, 256>>>(p); }, n_inst < ni ? 24 : 6);
            return (which == 3 ? 2.0 : 1.0) * n_inst * nv * 12 / (tp * 1e-3) / 1e9;
        };
        const float tfa = timeit([&] { fill<<<unsigned((arr / 16 + 255) / 256), 256>>>(a, arr / 16); });
        const float tfb = timeit([&] { fill<<<unsigned((arr / 16 + 255) / 256), 256>>>(b, arr / 16); });
        const float tf2 = timeit([&] { fill2<<<unsigned((arr / 16 + 255) / 256), 256>>>(a, b, arr / 16); });
        printf("trial %d a=%p b=%p | fill a %5.0f b %5.0f a+b lock-step %5.0f | pattern pair %5.0f  a alone %5.0f  b alone %5.0f | 4 KiB-quantised pair %5.0f |",
               t, (void *)a, (void *)b, arr / (tfa * 1e-3) / 1e9, arr / (tfb * 1e-3) / 1e9, 2.0 * arr / (tf2 * 1e-3) / 1e9,
               rate(ni, 0, 0, 3, 0), rate(ni, 0, 0, 1, 0), rate(ni, 0, 0, 2, 0), rate(ni, 0, 0, 3, 1));
        printf(" rot");
        for (uint32_t r : {1u, 2u, 4u, 5u}) printf(" %u:%5.0f", r, rate(ni, 0, r, 3, 0));
        printf(" | chunked interleave");
        for (uint32_t c : {32u, 16u, 8u, 4u, 2u, 1u}) printf(" %u:%5.0f", c, rate(ni, 0, 0, 3, 0, c));
        printf(" | eighths (pair)");
        for (uint32_t e = 0; e < 8; ++e) printf(" %5.0f", rate(128, e * 128, 0, 3, 0));
        printf(" | eighths (a)");
        for (uint32_t e = 0; e < 8; ++e) printf(" %5.0f", rate(128, e * 128, 0, 1, 0));
        printf(" | pair again %5.0f GB/s\n", rate(ni, 0, 0, 3, 0));
        fflush(stdout);
        if (keep) { kept.push_back(a); kept.push_back(b); } else { CK(hipFree(a)); CK(hipFree(b)); }
    }
    for (void *q : kept) CK(hipFree(q));
    return 0;
}
