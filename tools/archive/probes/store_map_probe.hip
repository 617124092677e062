// tools/archive/probes/store_map_probe.hip -- how should a 256-thread workgroup map its 768 16-byte stores (6 KB of
// array A + 6 KB of array B per instance) onto lanes?  M0: store i covers chunks [256i, 256i+256)
// (each wave-store 1 KB, a wave's three stores 4 KB apart).  M1: wave w owns 3 consecutive KB.
// M2: thread owns 3 consecutive chunks (48 B).  Crowd pattern (tile, 16 strided instances), for
// several A->B distances.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int MAP>
__global__ __launch_bounds__(256) void crowd(float4 *a, float4 *b, int ntiles, int group, size_t stride4) {
    const float4 v = make_float4(1, 2, 3, 4);
    const int tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles, t = threadIdx.x;
    for (int k = 0; k < group; ++k) {
        const size_t base = (size_t(grp) * group + k) * stride4 + size_t(tile) * 384;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int q;
            if (MAP == 0) q = t + 256 * i;
            else if (MAP == 1) q = ((t >> 6) * 3 + i) * 64 + (t & 63);
            else q = 3 * t + i;
            if (q < 384) a[base + q] = v; else b[base + q - 384] = v;
        }
    }
}
template <typename F> float timeit(F f, int iters = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}
int main() {
    const int ntiles = 98, ni = 1024, group = 16, ngroups = ni / group;
    const size_t stride = 600000, arr = size_t(ni) * stride;
    char *buf; CK(hipMalloc(&buf, size_t(3) << 30)); CK(hipMemset(buf, 0, size_t(3) << 30));
    const double by = 2.0 * ni * ntiles * 6144;
    const size_t MiB = 1 << 20;
    for (size_t dist : {586 * MiB, 588 * MiB, 589 * MiB, 590 * MiB, 592 * MiB, 640 * MiB, 1024 * MiB, 1200 * MiB, arr, arr + 2048, arr + 4096 + 256}) {
        float4 *a = (float4 *)buf, *b = (float4 *)(buf + dist);
        float t0 = timeit([&] { crowd<0><<<ntiles * ngroups, 256>>>(a, b, ntiles, group, stride / 16); });
        float t1 = timeit([&] { crowd<1><<<ntiles * ngroups, 256>>>(a, b, ntiles, group, stride / 16); });
        float t2 = timeit([&] { crowd<2><<<ntiles * ngroups, 256>>>(a, b, ntiles, group, stride / 16); });
        printf("B-A = %11zu (%7.2f MiB) | M0 store-major %6.1f us (%5.0f GB/s) | M1 wave-contiguous %6.1f (%5.0f) | M2 thread-contiguous %6.1f (%5.0f)\n",
               dist, double(dist) / MiB, t0 * 1e3, by / (t0 * 1e-3) / 1e9, t1 * 1e3, by / (t1 * 1e-3) / 1e9, t2 * 1e3, by / (t2 * 1e-3) / 1e9);
    }
    return 0;
}
