// tools/archive/probes/alloc_api_probe.hip -- does the allocation API decide the store-pattern mode?  For each way of obtaining the
// crowd's two output arrays (614 MB each): allocate, time a linear fill and the deform store pattern, free; repeated.
// Also: sub-ranges of ONE large allocation.  Measurement tool only (hipcc --offload-arch=gfx950 -O2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void pattern(float4 *a, float4 *b, uint32_t nv, uint32_t ni, uint32_t ntiles) {
    const uint32_t tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    const uint32_t v0 = tile * 512, nvt = min(512u, nv - v0);
    const uint32_t pa = nvt * 12 / 16;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (uint32_t g = grp * 16; g < min(ni, grp * 16 + 16); ++g) {
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
static const uint32_t nv = 50000, ni = 1024, ntiles = 98;
static const size_t arr = size_t(ni) * nv * 12;
static void measure(const char *what, int trial, void *a, void *b) {
    float tf = timeit([&] { fill<<<unsigned((arr / 16 + 255) / 256), 256>>>((float4 *)a, arr / 16); });
    float tp = timeit([&] { pattern<<<ntiles * (ni / 16), 256>>>((float4 *)a, (float4 *)b, nv, ni, ntiles); });
    printf("%-34s trial %2d  a=%p b=%p  fill %6.0f GB/s  pattern %6.0f GB/s\n", what, trial, a, b, arr / (tf * 1e-3) / 1e9,
           2.0 * arr / (tp * 1e-3) / 1e9);
    fflush(stdout);
}
int main(int argc, char **argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 8;
    for (int t = 0; t < trials; ++t) {
        void *a, *b; CK(hipMalloc(&a, arr)); CK(hipMalloc(&b, arr));
        measure("hipMalloc", t, a, b);
        CK(hipFree(a)); CK(hipFree(b));
    }
    for (int t = 0; t < trials; ++t) {
        void *a, *b; CK(hipMallocAsync(&a, arr, 0)); CK(hipMallocAsync(&b, arr, 0)); CK(hipDeviceSynchronize());
        measure("hipMallocAsync", t, a, b);
        CK(hipFreeAsync(a, 0)); CK(hipFreeAsync(b, 0)); CK(hipDeviceSynchronize());
    }
    for (unsigned flag : {hipDeviceMallocDefault, hipDeviceMallocFinegrained, hipDeviceMallocUncached}) {
        for (int t = 0; t < trials / 2; ++t) {
            void *a = nullptr, *b = nullptr;
            if (hipExtMallocWithFlags(&a, arr, flag) != hipSuccess || hipExtMallocWithFlags(&b, arr, flag) != hipSuccess) {
                printf("hipExtMallocWithFlags(%u) failed\n", flag); (void)hipGetLastError(); break;
            }
            char name[64]; snprintf(name, sizeof name, "hipExtMallocWithFlags(%u)", flag);
            measure(name, t, a, b);
            CK(hipFree(a)); CK(hipFree(b));
        }
    }
    {   // one large allocation, the two arrays at different offsets inside it
        const size_t big = size_t(12) << 30;
        char *pool = nullptr;
        if (hipMalloc(&pool, big) == hipSuccess) {
            const size_t step = (arr + (size_t(2) << 20) - 1) / (size_t(2) << 20) * (size_t(2) << 20);
            for (int k = 0; k + 2 <= int(big / step) && k < 16; k += 2) {
                char name[64]; snprintf(name, sizeof name, "12 GiB pool, arrays %d,%d", k, k + 1);
                measure(name, k / 2, pool + size_t(k) * step, pool + size_t(k + 1) * step);
            }
            measure("12 GiB pool, arrays 0 and 15", 0, pool, pool + 15 * step);
            CK(hipFree(pool));
        } else { (void)hipGetLastError(); printf("12 GiB pool: allocation failed\n"); }
    }
    {   // many allocations kept alive (no frees in between): is the mode tied to what was freed just before?
        std::vector<void *> keep;
        for (int t = 0; t < trials; ++t) {
            void *a, *b; CK(hipMalloc(&a, arr)); CK(hipMalloc(&b, arr));
            measure("hipMalloc, nothing freed", t, a, b);
            keep.push_back(a); keep.push_back(b);
        }
        for (void *p : keep) CK(hipFree(p));
    }
    return 0;
}
