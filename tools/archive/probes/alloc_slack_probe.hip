// tools/archive/probes/alloc_slack_probe.hip -- why does tools/archive/probes/store_pattern_probe see the fast store mode on every allocation while
// tools/archive/probes/alloc_api_probe sees it on one in seven?  Candidates: the 4 MiB of slack the former allocates, its 97 (not 98)
// tiles per instance, its 2-D grid.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

// mode 0: block -> tile = b % ntiles, grp = b / ntiles (1-D);  mode 1: XCD-aware (the deform kernel's mapping, interleaved instances)
__global__ __launch_bounds__(256) void pattern(float4 *a, float4 *b, uint32_t nv, uint32_t ni, uint32_t ntiles, uint32_t ngroups, int mode) {
    uint32_t tile, grp;
    if (mode == 0) { tile = blockIdx.x % ntiles; grp = blockIdx.x / ntiles; }
    else {
        const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3, T = ntiles >> 3, main_count = T * ngroups;
        if (k < main_count) { grp = k / T; tile = xcd * T + (k - grp * T); }
        else {
            const uint32_t rem = ((ntiles & 7u) * ngroups + 7u) / 8u, r = xcd * rem + (k - main_count);
            if (r >= (ntiles & 7u) * ngroups) return;
            const uint32_t rt = r / ngroups; tile = 8u * T + rt; grp = r - rt * ngroups;
        }
    }
    const uint32_t v0 = tile * 512, nvt = min(512u, nv - v0);
    const uint32_t pa = nvt * 12 / 16;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (uint32_t j = 0; j < 16; ++j) {
        const uint32_t g = mode == 0 ? grp * 16 + j : j * ngroups + grp;
        if (g >= ni) continue;
        const size_t base = (size_t(g) * nv + v0) * 12 / 16;
        for (uint32_t q = threadIdx.x; q < 2 * pa; q += 256) { if (q < pa) a[base + q] = v; else b[base + q - pa] = v; }
    }
}
__global__ __launch_bounds__(256) void fill(float4 *d, size_t n) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) d[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <typename F> float timeit(F f, int iters = 5) {
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
    const uint32_t nv = 50000, ni = 1024, ngroups = ni / 16;
    const size_t arr = size_t(ni) * nv * 12;
    const int trials = argc > 1 ? atoi(argv[1]) : 6;
    for (size_t slack : {size_t(0), size_t(4) << 20, size_t(64) << 20}) {
        for (int t = 0; t < trials; ++t) {
            float4 *a, *b; CK(hipMalloc(&a, arr + slack)); CK(hipMalloc(&b, arr + slack));
            float tf = timeit([&] { fill<<<unsigned((arr / 16 + 255) / 256), 256>>>(a, arr / 16); });
            auto rate = [&](uint32_t ntiles, int mode) {
                const unsigned grid = mode == 0 ? ntiles * ngroups : 8u * ((ntiles >> 3) * ngroups + ((ntiles & 7u) * ngroups + 7u) / 8u);
                float tp = timeit([&] { pattern<<<grid, 256>>>(a, b, nv, ni, ntiles, ngroups, mode); });
                return 2.0 * ni * (ntiles == 98 ? nv : 97 * 512) * 12 / (tp * 1e-3) / 1e9;
            };
            printf("slack %3zu MiB trial %d a=%p b=%p fill %5.0f | 1-D 98 tiles %5.0f  1-D 97 tiles %5.0f | xcd-aware 98 %5.0f  97 %5.0f  96 %5.0f GB/s\n",
                   slack >> 20, t, (void *)a, (void *)b, arr / (tf * 1e-3) / 1e9, rate(98, 0), rate(97, 0), rate(98, 1), rate(97, 1), rate(96, 1));
            fflush(stdout);
            CK(hipFree(a)); CK(hipFree(b));
        }
    }
    return 0;
}
