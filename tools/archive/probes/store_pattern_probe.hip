// tools/archive/probes/store_pattern_probe.hip -- which (tile size, grid order, stagger) makes the deform store
// pattern reach the linear-fill rate on every box?  Measurement tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// piece4 = float4 per tile per array; a block writes `group` instances of one tile.
// ORDER 0: blockIdx.x = tile (fast), blockIdx.y = group   ORDER 1: blockIdx.x = group (fast), y = tile
// STAG: instance order inside the group is rotated by the tile index
template <int ORDER, int STAG, int ILV = 0>
__global__ __launch_bounds__(256) void deformlike(float4 *a, float4 *b, int ntiles, int ni, int group, int piece4, size_t stride4) {
    int tile = ORDER ? blockIdx.y : blockIdx.x, grp = ORDER ? blockIdx.x : blockIdx.y;
    int g0 = grp * group;
    for (int k = 0; k < group; ++k) {
        int kk = STAG ? (k + tile) % group : k;
        int ngroups = (ni + group - 1) / group;
        int g = ILV ? kk * ngroups + grp : g0 + kk;   // ILV: instance = k*ngroups + grp (adjacent instances are written together)
        if (g >= ni) continue;
        size_t base = size_t(g) * stride4 + size_t(tile) * piece4;
        float4 v = make_float4(1, 2, 3, g);
        for (int q = threadIdx.x; q < 2 * piece4; q += 256) { if (q < piece4) a[base + q] = v; else b[base + q - piece4] = v; }
    }
}
// linear reference: block writes contiguous 6 KB chunk
__global__ __launch_bounds__(256) void fill_chunk(float4 *d, size_t n, size_t chunk) {
    const float4 v = make_float4(1, 2, 3, 4);
    size_t base = size_t(blockIdx.x) * chunk, end = base + chunk < n ? base + chunk : n;
    for (size_t i = base + threadIdx.x; i < end; i += 256) d[i] = v;
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
int main(int argc, char **argv) {
    const int ni = 1024; const size_t nv = argc > 1 ? atoi(argv[1]) : 50176;      // 98 * 512, so every tile size divides
    const size_t arr = size_t(ni) * nv * 12;
    float4 *a, *b; CK(hipMalloc(&a, arr + (4 << 20))); CK(hipMalloc(&b, arr + (4 << 20)));   // + slack
    {
        float ms = timeit([&] { fill_chunk<<<int((2 * arr / 16 + 383) / 384), 256>>>(a, arr / 16, 384); });
        printf("linear fill 6 KB chunks (one array, %zu MB)            %7.1f us %7.1f GB/s\n", arr >> 20, ms * 1e3, arr / (ms * 1e-3) / 1e9);
    }
    for (int tv : {512, 2048}) {
        int ntiles = int(nv / tv), piece4 = tv * 12 / 16;   // floor: never past the end of an instance
        for (int group : {4, 16, 64}) {
            double bytes = 2.0 * ni * ntiles * piece4 * 16;
            float t00 = timeit([&] { deformlike<0, 0><<<dim3(ntiles, ni / group), 256>>>(a, b, ntiles, ni, group, piece4, nv * 12 / 16); });
            float t01 = timeit([&] { deformlike<0, 1><<<dim3(ntiles, ni / group), 256>>>(a, b, ntiles, ni, group, piece4, nv * 12 / 16); });
            float t10 = timeit([&] { deformlike<1, 0><<<dim3(ni / group, ntiles), 256>>>(a, b, ntiles, ni, group, piece4, nv * 12 / 16); });
            float t11 = timeit([&] { deformlike<1, 1><<<dim3(ni / group, ntiles), 256>>>(a, b, ntiles, ni, group, piece4, nv * 12 / 16); });
            float i00 = timeit([&] { deformlike<0, 0, 1><<<dim3(ntiles, ni / group), 256>>>(a, b, ntiles, ni, group, piece4, nv * 12 / 16); });
            float i10 = timeit([&] { deformlike<1, 0, 1><<<dim3(ni / group, ntiles), 256>>>(a, b, ntiles, ni, group, piece4, nv * 12 / 16); });
            printf("tile %4d group %2d blocks %6d | tile-fast %6.1f us (%6.0f GB/s) +stagger %6.1f | group-fast %6.1f +stagger %6.1f | INTERLEAVED tile-fast %6.1f (%6.0f GB/s) group-fast %6.1f\n",
                   tv, group, ntiles * (ni / group), t00 * 1e3, bytes / (t00 * 1e-3) / 1e9, t01 * 1e3, t10 * 1e3, t11 * 1e3, i00 * 1e3, bytes / (i00 * 1e-3) / 1e9, i10 * 1e3);
        }
    }
    {   // single-array variants: only A (12 B/vertex pieces), and one interleaved 32 B/vertex array
        const int tv = 512, group = 16; int ntiles = int(nv / tv);
        int piece4 = tv * 12 / 16;
        float t = timeit([&] { deformlike<0, 0><<<dim3(ntiles, ni / group), 256>>>(a, a, ntiles, ni, group, piece4 / 2, nv * 12 / 16); });
        printf("single array, 3 KB+3 KB pieces both into A                 %7.1f us %7.0f GB/s\n", t * 1e3, 2.0 * ni * ntiles * (piece4 / 2) * 16 / (t * 1e-3) / 1e9);
        int p32 = tv * 32 / 16 / 2;   // deformlike writes 2*piece4 per instance: use halves of one 16 KB piece
        float4 *c; CK(hipMalloc(&c, size_t(ni) * nv * 32 + (4 << 20)));
        float t2 = timeit([&] { deformlike<0, 0><<<dim3(ntiles, ni / group), 256>>>(c, c + p32, ntiles, ni, group, p32, nv * 32 / 16); });
        printf("vertex32-like: one array, 16 KB pieces, stride nv*32        %7.1f us %7.0f GB/s\n", t2 * 1e3, 2.0 * ni * ntiles * p32 * 16 / (t2 * 1e-3) / 1e9);
    }
    return 0;
}
