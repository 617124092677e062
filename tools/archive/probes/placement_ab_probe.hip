// tools/archive/probes/placement_ab_probe.hip -- is the slow store mode an interference between the TWO output arrays?  Per pair of fresh allocations:
//   P1  the crowd pattern: every workgroup writes its 6 KiB piece of a, then of b, per instance (the deform kernel's rhythm)
//   P2  the same bytes per workgroup and step, but a workgroup writes ONE array only: pieces of two instances per step; the first half
//       of the grid writes a, the second half b (a and b are then mostly written at different times)
//   P3  lock-step like P1, but the b piece belongs to the instance S rows further on (same bytes, the a / b pairing shifted by S x 600 KB)
//   P4  like P1 with 768-byte alternation inside the piece (the tile-order kernel's rhythm: wave w writes its 768 B of a, then of b)
// Measurement tool only.   hipcc --offload-arch=gfx950 -O2 tools/archive/probes/placement_ab_probe.hip -o tools/archive/probes/placement_ab_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

struct P { float4 *a, *b; uint32_t nv, ni, ntiles, ngroups, mode, shift; };

__global__ __launch_bounds__(256) void pattern(const P p) {
    uint32_t bid = blockIdx.x, half = 0;
    const uint32_t per = 8u * ((p.ntiles >> 3) * p.ngroups + ((p.ntiles & 7u) * p.ngroups + 7u) / 8u);
    if (p.mode == 2) { half = bid >= per / 2 ? 1u : 0u; }
    const uint32_t xcd = bid & 7u, k = (p.mode == 2 ? (bid % (per / 2)) : bid) >> 3, T = p.ntiles >> 3;
    const uint32_t ngr = p.mode == 2 ? p.ngroups / 2 : p.ngroups, main_count = T * ngr;
    uint32_t tile, grp;
    if (k < main_count) { grp = k / T; tile = xcd * T + (k - grp * T); }
    else {
        const uint32_t rem = ((p.ntiles & 7u) * ngr + 7u) / 8u, r = xcd * rem + (k - main_count);
        if (r >= (p.ntiles & 7u) * ngr) return;
        const uint32_t rt = r / ngr; tile = 8u * T + rt; grp = r - rt * ngr;
    }
    const uint32_t v0 = tile * 512, nvt = min(512u, p.nv - v0), n = nvt * 12 / 16;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    if (p.mode == 2) {                         // 32 instances per workgroup, two per step, one array
        float4 *d = half ? p.b : p.a;
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t g0 = (2 * j) * ngr + grp, g1 = (2 * j + 1) * ngr + grp;
            const size_t lo0 = (size_t(g0) * p.nv + v0) * 12 / 16, lo1 = (size_t(g1) * p.nv + v0) * 12 / 16;
            for (uint32_t q = threadIdx.x; q < 2 * n; q += 256) { if (q < n) d[lo0 + q] = v; else d[lo1 + q - n] = v; }
        }
        return;
    }
    for (uint32_t j = 0; j < 16; ++j) {
        const uint32_t g = j * p.ngroups + grp;
        if (g >= p.ni) continue;
        const uint32_t gb = p.mode == 3 ? (g + p.shift) % p.ni : g;
        const size_t lo = (size_t(g) * p.nv + v0) * 12 / 16, lob = (size_t(gb) * p.nv + v0) * 12 / 16;
        if (p.mode == 4) {                     // 768-byte alternation: 48 float4 of a, 48 of b, per wave
            const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63u;
            for (uint32_t c = w; c * 48 < n; c += 4) {
                if (l < 48 && c * 48 + l < n) p.a[lo + c * 48 + l] = v;
                if (l < 48 && c * 48 + l < n) p.b[lob + c * 48 + l] = v;
            }
        } else {
            for (uint32_t q = threadIdx.x; q < 2 * n; q += 256) { if (q < n) p.a[lo + q] = v; else p.b[lob + q - n] = v; }
        }
    }
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
    const uint32_t nv = 50000, ni = 1024, ntiles = 98, ngroups = 64;
    const size_t arr = size_t(ni) * nv * 12;
    const int trials = argc > 1 ? atoi(argv[1]) : 10;
    const unsigned grid = 8u * ((ntiles >> 3) * ngroups + ((ntiles & 7u) * ngroups + 7u) / 8u);
    for (int t = 0; t < trials; ++t) {
        float4 *a, *b; CK(hipMalloc(&a, arr)); CK(hipMalloc(&b, arr));
        auto rate = [&](uint32_t mode, uint32_t shift) {
            P p{a, b, nv, ni, ntiles, ngroups, mode, shift};
            return 2.0 * arr / (timeit([&] { pattern<<<grid, 256>>>(p); }) * 1e-3) / 1e9;
        };
        printf("trial %2d | P1 a+b per step %5.0f | P2 one array per workgroup %5.0f | P3 b shifted by 1 row %5.0f  by 4 %5.0f  by 64 %5.0f  by 512 %5.0f | P4 768-byte alternation %5.0f | P1 again %5.0f GB/s\n",
               t, rate(1, 0), rate(2, 0), rate(3, 1), rate(3, 4), rate(3, 64), rate(3, 512), rate(4, 0), rate(1, 0));
        fflush(stdout);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
