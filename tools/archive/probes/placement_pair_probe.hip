// tools/archive/probes/placement_pair_probe.hip -- structure of the fast / slow placements of the crowd's two output arrays: is "fast" a property of
// each array or of the PAIR?  (1) 8 separately allocated arrays, the store pattern and a lock-step linear fill for all 28
// pairs;  (2) one large allocation, array b at a + D for a ladder of distances D (sub-page to hundreds of MiB).
// Measurement tool only.   hipcc --offload-arch=gfx950 -O2 tools/archive/probes/placement_pair_probe.hip -o tools/archive/probes/placement_pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void pattern(float4 *a, float4 *b, uint32_t nv, uint32_t ni, uint32_t ntiles, uint32_t ngroups) {
    const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3, T = ntiles >> 3, main_count = T * ngroups;
    uint32_t tile, grp;
    if (k < main_count) { grp = k / T; tile = xcd * T + (k - grp * T); }
    else {
        const uint32_t rem = ((ntiles & 7u) * ngroups + 7u) / 8u, r = xcd * rem + (k - main_count);
        if (r >= (ntiles & 7u) * ngroups) return;
        const uint32_t rt = r / ngroups; tile = 8u * T + rt; grp = r - rt * ngroups;
    }
    const uint32_t v0 = tile * 512, nvt = min(512u, nv - v0), n = nvt * 12 / 16;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (uint32_t j = 0; j < 16; ++j) {
        const uint32_t g = j * ngroups + grp;
        if (g >= ni) continue;
        const size_t lo = (size_t(g) * nv + v0) * 12 / 16;
        for (uint32_t q = threadIdx.x; q < 2 * n; q += 256) { if (q < n) a[lo + q] = v; else b[lo + q - n] = v; }
    }
}
__global__ __launch_bounds__(256) void fill2(float4 *a, float4 *b, size_t n) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) { a[i] = make_float4(1.f, 2.f, 3.f, 4.f); b[i] = make_float4(1.f, 2.f, 3.f, 4.f); }
}
template <typename F> float timeit(F f, int iters = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters;
}
int main() {
    const uint32_t nv = 50000, ni = 1024, ntiles = 98, ngroups = 64;
    const size_t arr = size_t(ni) * nv * 12;
    const unsigned grid = 8u * ((ntiles >> 3) * ngroups + ((ntiles & 7u) * ngroups + 7u) / 8u);
    auto pat = [&](float4 *a, float4 *b) { return 2.0 * arr / (timeit([&] { pattern<<<grid, 256>>>(a, b, nv, ni, ntiles, ngroups); }) * 1e-3) / 1e9; };
    auto f2 = [&](float4 *a, float4 *b) { return 2.0 * arr / (timeit([&] { fill2<<<unsigned((arr / 16 + 255) / 256), 256>>>(a, b, arr / 16); }) * 1e-3) / 1e9; };
    const int N = 8;
    float4 *x[N];
    for (int i = 0; i < N; ++i) { CK(hipMalloc(&x[i], arr)); printf("array %d at %p\n", i, (void *)x[i]); }
    for (int w = 0; w < 30; ++w) fill2<<<unsigned((arr / 16 + 255) / 256), 256>>>(x[0], x[1], arr / 16);
    CK(hipDeviceSynchronize());
    printf("== 1. all pairs of %d separately allocated arrays: pattern GB/s (upper triangle), lock-step fill GB/s (lower triangle)\n      ", N);
    for (int j = 0; j < N; ++j) printf("%6d", j);
    printf("\n");
    for (int i = 0; i < N; ++i) {
        printf("%6d", i);
        for (int j = 0; j < N; ++j) {
            if (i == j) printf("     -");
            else if (i < j) printf("%6.0f", pat(x[i], x[j]));
            else printf("%6.0f", f2(x[j], x[i]));
        }
        printf("\n"); fflush(stdout);
    }
    for (int i = 0; i < N; ++i) CK(hipFree(x[i]));
    printf("== 2. one allocation, b = a + D\n");
    const size_t big = 3 * arr + (size_t(64) << 20);
    for (int rep = 0; rep < 2; ++rep) {
        unsigned char *r; CK(hipMalloc(&r, big));
        float4 *a = reinterpret_cast<float4 *>(r);
        const size_t arr_up = (arr + (size_t(2) << 20) - 1) >> 21 << 21;
        printf("allocation %d at %p\n", rep, (void *)r);
        std::vector<size_t> extra = {0, 256, 1024, 4096, 8192, 16384, 32768, 65536, size_t(128) << 10, size_t(256) << 10, size_t(512) << 10,
                                     size_t(1) << 20, size_t(2) << 20, size_t(4) << 20, size_t(8) << 20, size_t(16) << 20, size_t(32) << 20,
                                     size_t(64) << 20, size_t(128) << 20, size_t(256) << 20, size_t(512) << 20, (size_t(512) << 20) + 4096,
                                     (size_t(512) << 20) + 65536, size_t(600) << 20};
        for (size_t e : extra) {
            float4 *b = reinterpret_cast<float4 *>(r + arr_up + e);
            printf("  D = %zu MiB + %9zu B: pattern %5.0f  lock-step fill %5.0f GB/s\n", arr_up >> 20, e, pat(a, b), f2(a, b));
            fflush(stdout);
        }
        CK(hipFree(r));
    }
    return 0;
}
