// tools/archive/probes/store_align_probe.hip -- does the deform store pattern's bandwidth depend on the relative
// placement of the two output arrays / the instance stride?  Measurement tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void deformlike(float4 *a, float4 *b, int ni, int group, int piece4, size_t stride4) {
    int tile = blockIdx.x, g0 = blockIdx.y * group;
    for (int g = g0; g < g0 + group && g < ni; ++g) {
        size_t base = size_t(g) * stride4 + size_t(tile) * piece4;
        float4 v = make_float4(1, 2, 3, g);
        int t = threadIdx.x;
        a[base + t] = v;
        if (t < 128) a[base + 256 + t] = v; else b[base + t - 128] = v;
        b[base + 128 + t] = v;
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
    const size_t big = size_t(3) << 30;
    char *buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 0, big));
    printf("buf=%p\n", (void *)buf);
    const int ntiles = 98, ni = 1024, piece4 = 384, group = 16;
    const size_t arr = size_t(ni) * 600000;          // 614.4 MB
    auto run = [&](const char *nm, size_t offA, size_t offB, size_t stride_bytes) {
        float4 *a = (float4 *)(buf + offA), *b = (float4 *)(buf + offB);
        float ms = timeit([&] { deformlike<<<dim3(ntiles, ni / group), 256>>>(a, b, ni, group, piece4, stride_bytes / 16); });
        printf("%-52s offB-offA=%11zu stride=%7zu  %7.1f us %7.1f GB/s\n", nm, offB - offA, stride_bytes, ms * 1e3, 2.0 * ni * ntiles * piece4 * 16 / (ms * 1e-3) / 1e9);
    };
    size_t gib = size_t(1) << 30;
    run("B = A + 1 GiB", 0, gib, 600000);
    run("B = A + 1 GiB + 256", 0, gib + 256, 600000);
    run("B = A + 1 GiB + 1 KiB", 0, gib + 1024, 600000);
    run("B = A + 1 GiB + 4 KiB", 0, gib + 4096, 600000);
    run("B = A + 1 GiB + 16 KiB", 0, gib + 16384, 600000);
    run("B = A + 1 GiB + 64 KiB", 0, gib + 65536, 600000);
    run("B = A + 1 GiB + 1 MiB", 0, gib + (1 << 20), 600000);
    run("B = A + arr (packed)", 0, arr, 600000);
    run("B = A + arr rounded to 2 MiB", 0, (arr + (2 << 20) - 1) / (2 << 20) * (2 << 20), 600000);
    run("B = A + arr rounded to 2 MiB + 3 KiB", 0, (arr + (2 << 20) - 1) / (2 << 20) * (2 << 20) + 3072, 600000);
    run("stride 600064 (B = A + 1 GiB)", 0, gib, 600064);
    run("stride 602112 = 147*4096", 0, gib, 602112);
    run("stride 655360 = 640 KiB", 0, gib, 655360);
    run("stride 1 MiB", 0, gib + (256 << 20), 1 << 20);
    run("repeat: B = A + 1 GiB", 0, gib, 600000);
    run("INSTANCE-MAJOR: B_i right after A_i (stride 1.2 MB)", 0, 600000, 1200000);
    run("INSTANCE-MAJOR +4 KiB gap (stride 1.2 MB + 8 KiB)", 0, 600000 + 4096, 1200000 + 8192);
    run("repeat: B = A + 1 GiB", 0, gib, 600000);
    for (int i = 0; i < 4; ++i) {      // fresh allocations, as the bench does
        char *pa, *pb; CK(hipMalloc(&pa, arr)); CK(hipMalloc(&pb, arr));
        float ms = timeit([&] { deformlike<<<dim3(ntiles, ni / group), 256>>>((float4 *)pa, (float4 *)pb, ni, group, piece4, 600000 / 16); });
        printf("fresh hipMalloc pair a=%p b=%p (b-a=%td)  %7.1f us\n", (void *)pa, (void *)pb, pb - pa, ms * 1e3);
        if (i % 2) { CK(hipFree(pa)); CK(hipFree(pb)); }
    }
    return 0;
}
