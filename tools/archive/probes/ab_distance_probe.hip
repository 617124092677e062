// tools/archive/probes/ab_distance_probe.hip -- the deform store pattern's rate as a function of the DISTANCE between the two output
// arrays (positions a, normals b) inside one large allocation.  tools/archive/probes/alloc_api_probe showed: adjacent arrays are slow,
// arrays 9 GB apart are fast.  Which distances are fast?  Measurement tool only.
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
template <typename F> float timeit(F f, int iters = 4) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
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
    const size_t big = size_t(20) << 30;
    char *pool = nullptr;
    CK(hipMalloc(&pool, big));
    printf("pool %p, array %zu B (0x%zx)\n", (void *)pool, arr, arr);
    auto run = [&](size_t a_off, size_t dist) {
        float4 *a = (float4 *)(pool + a_off), *b = (float4 *)(pool + a_off + dist);
        float t = timeit([&] { pattern<<<ntiles * (ni / 16), 256>>>(a, b, nv, ni, ntiles); });
        printf("a_off %6zu MiB  b-a = %9.3f MiB (0x%010zx)  %6.0f GB/s\n", a_off >> 20, dist / 1048576.0, dist, 2.0 * arr / (t * 1e-3) / 1e9);
        fflush(stdout);
    };
    const size_t MB = size_t(1) << 20;
    const size_t base = (arr + 2 * MB - 1) / (2 * MB) * (2 * MB);          // first non-overlapping 2 MiB-aligned distance
    // 1. fine sweep: base + k * 2 MiB
    for (int k = 0; k < 40; ++k) run(0, base + size_t(k) * 2 * MB);
    // 2. coarse sweep: base + k * 64 MiB up to 4 GiB
    for (int k = 1; k <= 64; ++k) run(0, base + size_t(k) * 64 * MB);
    // 3. GiB steps
    for (int k = 1; k <= 16; ++k) run(0, size_t(k) << 30);
    // 4. the same distances from another starting point of a
    for (int k : {0, 1, 2, 3, 5, 8, 13}) run(size_t(1) << 30, base + size_t(k) * 64 * MB);
    // 5. sub-2MiB offsets on top of a slow and a fast distance
    for (size_t d : {size_t(0), size_t(4096), size_t(65536), size_t(1) << 18, size_t(1) << 20}) { run(0, base + d); run(0, (size_t(9) << 30) + d); }
    CK(hipFree(pool));
    return 0;
}
