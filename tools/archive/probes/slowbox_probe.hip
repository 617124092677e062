// tools/archive/probes/slowbox_probe.hip -- which property of the crowd store pattern costs bandwidth on the "slow"
// boxes (two arrays? persistent blocks? distance between concurrently written regions?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// every block writes `pieces` consecutive 6 KB pieces per array; piece index p = blockIdx.x * pieces + k.
// MODE 0: A only (twice the pieces)   MODE 1: A and B, same relative offsets   MODE 2: A then B halves of grid
template <int MODE>
__global__ __launch_bounds__(256) void linear2(float4 *a, float4 *b, size_t npieces, int pieces) {
    const float4 v = make_float4(1, 2, 3, 4);
    for (int k = 0; k < pieces; ++k) {
        size_t p = size_t(blockIdx.x) * pieces + k;
        if (MODE == 2) { float4 *dst = p < npieces ? a : b; size_t q = p < npieces ? p : p - npieces;
            if (q < npieces) { for (int t = threadIdx.x; t < 384; t += 256) dst[q * 384 + t] = v; } continue; }
        if (p >= npieces) return;
        for (int t = threadIdx.x; t < 768; t += 256) {
            if (MODE == 0) a[p * 768 + t] = v;
            else if (t < 384) a[p * 384 + t] = v; else b[p * 384 + t - 384] = v;
        }
    }
}
// the crowd pattern: block (tile, grp) writes piece `tile` of instances grp*group .. +group
__global__ __launch_bounds__(256) void deformlike(float4 *a, float4 *b, int ntiles, int group, size_t stride4, int ilv, int ngroups) {
    const float4 v = make_float4(1, 2, 3, 4);
    int tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    for (int k = 0; k < group; ++k) {
        size_t inst = ilv ? size_t(k) * ngroups + grp : size_t(grp) * group + k;
        size_t base = inst * stride4 + size_t(tile) * 384;
        for (int t = threadIdx.x; t < 768; t += 256) { if (t < 384) a[base + t] = v; else b[base + t - 384] = v; }
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
    const size_t npieces = 100352;                 // 1024 x 98 pieces of 6 KB per array
    const size_t arr = npieces * 6144 + (1 << 20);  // 616.6 MB + slack for the 602112-byte stride
    float4 *a, *b, *big;
    CK(hipMalloc(&a, arr)); CK(hipMalloc(&b, arr)); CK(hipMalloc(&big, 2 * arr));
    const double bytes = 2.0 * arr;
    for (int pieces : {1, 16}) {
        int blocks = int((npieces + pieces - 1) / pieces);
        float t0 = timeit([&] { linear2<0><<<blocks, 256>>>(big, nullptr, npieces, pieces); });
        float t1 = timeit([&] { linear2<1><<<blocks, 256>>>(a, b, npieces, pieces); });
        float t2 = timeit([&] { linear2<2><<<2 * blocks, 256>>>(a, b, npieces, pieces); });
        float t3 = timeit([&] { linear2<1><<<blocks, 256>>>(big, big + arr / 16, npieces, pieces); });
        printf("pieces/block %2d | one array linear %6.1f us (%5.0f GB/s) | A+B lockstep %6.1f (%5.0f) | A then B %6.1f (%5.0f) | A+B in one allocation %6.1f (%5.0f)\n",
               pieces, t0 * 1e3, bytes / (t0 * 1e-3) / 1e9, t1 * 1e3, bytes / (t1 * 1e-3) / 1e9, t2 * 1e3, bytes / (t2 * 1e-3) / 1e9, t3 * 1e3, bytes / (t3 * 1e-3) / 1e9);
    }
    {
        const int ntiles = 98, ni = 1024, group = 16, ngroups = ni / group;
        for (size_t stride_bytes : {size_t(600000), size_t(602112)}) {
            float t0 = timeit([&] { deformlike<<<ntiles * ngroups, 256>>>(a, b, ntiles, group, stride_bytes / 16, 0, ngroups); });
            float t1 = timeit([&] { deformlike<<<ntiles * ngroups, 256>>>(a, b, ntiles, group, stride_bytes / 16, 1, ngroups); });
            float t2 = timeit([&] { deformlike<<<ntiles * ngroups, 256>>>(big, big + arr / 16, ntiles, group, stride_bytes / 16, 0, ngroups); });
            double by = 2.0 * ni * ntiles * 6144;
            printf("crowd pattern stride %zu | a,b blocked %6.1f us (%5.0f GB/s) | a,b interleaved %6.1f (%5.0f) | one allocation blocked %6.1f (%5.0f)\n",
                   stride_bytes, t0 * 1e3, by / (t0 * 1e-3) / 1e9, t1 * 1e3, by / (t1 * 1e-3) / 1e9, t2 * 1e3, by / (t2 * 1e-3) / 1e9);
        }
    }
    return 0;
}
