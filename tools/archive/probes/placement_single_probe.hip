// tools/archive/probes/placement_single_probe.hip -- is the fast / slow store mode a property of each ARRAY (then a pair is fast iff both are good,
// and arrays can be shopped for one by one) or of the pair?  N separately allocated arrays: the crowd pattern written into ONE array
// (every workgroup writes the pieces of two instances per step, so the bytes per workgroup and step are those of the pair pattern),
// then the pair pattern for all pairs.  Measurement tool only.
//   hipcc --offload-arch=gfx950 -O2 tools/archive/probes/placement_single_probe.hip -o tools/archive/probes/placement_single_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__device__ __forceinline__ bool map_wg(uint32_t ntiles, uint32_t ngroups, uint32_t &tile, uint32_t &grp) {
    const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3, T = ntiles >> 3, main_count = T * ngroups;
    if (k < main_count) { grp = k / T; tile = xcd * T + (k - grp * T); return true; }
    const uint32_t rem = ((ntiles & 7u) * ngroups + 7u) / 8u, r = xcd * rem + (k - main_count);
    if (r >= (ntiles & 7u) * ngroups) return false;
    const uint32_t rt = r / ngroups; tile = 8u * T + rt; grp = r - rt * ngroups;
    return true;
}
// pair: 16 instances per workgroup, a piece then b piece per step
__global__ __launch_bounds__(256) void pair_pattern(float4 *a, float4 *b, uint32_t nv, uint32_t ni, uint32_t ntiles, uint32_t ngroups) {
    uint32_t tile, grp;
    if (!map_wg(ntiles, ngroups, tile, grp)) return;
    const uint32_t v0 = tile * 512, nvt = min(512u, nv - v0), n = nvt * 12 / 16;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (uint32_t j = 0; j < 16; ++j) {
        const uint32_t g = j * ngroups + grp;
        if (g >= ni) continue;
        const size_t lo = (size_t(g) * nv + v0) * 12 / 16;
        for (uint32_t q = threadIdx.x; q < 2 * n; q += 256) { if (q < n) a[lo + q] = v; else b[lo + q - n] = v; }
    }
}
// single: 32 instances per workgroup, the pieces of two instances of ONE array per step (ngroups = ni / 32)
__global__ __launch_bounds__(256) void single_pattern(float4 *a, uint32_t nv, uint32_t ni, uint32_t ntiles, uint32_t ngroups) {
    uint32_t tile, grp;
    if (!map_wg(ntiles, ngroups, tile, grp)) return;
    const uint32_t v0 = tile * 512, nvt = min(512u, nv - v0), n = nvt * 12 / 16;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (uint32_t j = 0; j < 16; ++j) {
        const uint32_t g0 = (2 * j) * ngroups + grp, g1 = (2 * j + 1) * ngroups + grp;
        const size_t lo0 = (size_t(g0) * nv + v0) * 12 / 16, lo1 = (size_t(g1) * nv + v0) * 12 / 16;
        for (uint32_t q = threadIdx.x; q < 2 * n; q += 256) { if (q < n) a[lo0 + q] = v; else a[lo1 + q - n] = v; }
    }
}
__global__ __launch_bounds__(256) void fill(float4 *d, size_t n) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) d[i] = make_float4(1.f, 2.f, 3.f, 4.f);
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
    const int N = argc > 1 ? atoi(argv[1]) : 10;
    float4 *x[32];
    auto grid = [&](uint32_t ng) { return 8u * ((ntiles >> 3) * ng + ((ntiles & 7u) * ng + 7u) / 8u); };
    for (int i = 0; i < N; ++i) CK(hipMalloc(&x[i], arr));
    double single[32];
    // two arrays written by a pair pattern as a yardstick for the single pattern: the single pattern on a MALL-defeating size
    // (614 MB > 256 MB) needs no companion
    for (int i = 0; i < N; ++i) {
        single[i] = arr / (timeit([&] { single_pattern<<<grid(32), 256>>>(x[i], nv, ni, ntiles, 32); }) * 1e-3) / 1e9;
        printf("array %2d at %p: single-array pattern %5.0f GB/s\n", i, (void *)x[i], single[i]);
    }
    printf("pair pattern (GB/s), row / column = array; diagonal = single\n      ");
    for (int j = 0; j < N; ++j) printf("%6d", j);
    printf("\n");
    for (int i = 0; i < N; ++i) {
        printf("%6d", i);
        for (int j = 0; j < N; ++j) {
            if (j < i) printf("      ");
            else if (i == j) printf("%6.0f", single[i]);
            else printf("%6.0f", 2.0 * arr / (timeit([&] { pair_pattern<<<grid(64), 256>>>(x[i], x[j], nv, ni, ntiles, 64); }) * 1e-3) / 1e9);
        }
        printf("\n"); fflush(stdout);
    }
    for (int i = 0; i < N; ++i) CK(hipFree(x[i]));
    // fresh pairs, freed after each trial (the modes vary from trial to trial at the same virtual addresses): does the pair rate follow
    // from the two single-array rates?
    printf("fresh pairs: single a, single b, pair, linear fill of a (GB/s)\n");
    for (int t = 0; t < 24; ++t) {
        float4 *a, *b; CK(hipMalloc(&a, arr)); CK(hipMalloc(&b, arr));
        const double sa = arr / (timeit([&] { single_pattern<<<grid(32), 256>>>(a, nv, ni, ntiles, 32); }) * 1e-3) / 1e9;
        const double sb = arr / (timeit([&] { single_pattern<<<grid(32), 256>>>(b, nv, ni, ntiles, 32); }) * 1e-3) / 1e9;
        const double pr = 2.0 * arr / (timeit([&] { pair_pattern<<<grid(64), 256>>>(a, b, nv, ni, ntiles, 64); }) * 1e-3) / 1e9;
        const double fl = arr / (timeit([&] { fill<<<unsigned((arr / 16 + 255) / 256), 256>>>(a, arr / 16); }) * 1e-3) / 1e9;
        printf("trial %2d  %5.0f  %5.0f  %5.0f  %5.0f\n", t, sa, sb, pr, fl);
        fflush(stdout);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
