// Store-only replay of the crowd output pattern with a configurable per-instance pitch: does some pitch make the
// slow placement mode go away?   hipcc --offload-arch=gfx950 -O3 tools/archive/probes/pitch_probe.hip -o tools/archive/probes/pitch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
constexpr uint32_t kThreads = 256, kTile = 512;
// pitch16 = float4 per instance row
__global__ __launch_bounds__(kThreads) void pattern_fill(float4 *a, float4 *b, uint32_t nv, uint32_t ni, uint32_t ntiles,
                                                         uint32_t ngroups, size_t pitch16) {
    const uint32_t tile = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    const uint32_t v0 = tile * kTile, nvt = min(kTile, nv - v0);
    const uint32_t piece4 = nvt * 12 / 16;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (uint32_t k = 0; k < 16; ++k) {
        const uint32_t g = k * ngroups + grp;
        if (g >= ni) break;
        const size_t base = size_t(g) * pitch16 + size_t(v0) * 12 / 16;
        for (uint32_t q = threadIdx.x; q < 2 * piece4; q += kThreads) {
            if (q < piece4) a[base + q] = v; else b[base + q - piece4] = v;
        }
    }
}
__global__ __launch_bounds__(kThreads) void fill(float4 *d, size_t n) {
    const size_t i = size_t(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n) d[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
float run(void *a, void *b, size_t pitch_bytes) {
    const uint32_t nv = 50000, ni = 1024, ntiles = (nv + kTile - 1) / kTile, ngroups = ni / 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) pattern_fill<<<ntiles * ngroups, kThreads>>>((float4 *)a, (float4 *)b, nv, ni, ntiles, ngroups, pitch_bytes / 16);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 10; ++w) pattern_fill<<<ntiles * ngroups, kThreads>>>((float4 *)a, (float4 *)b, nv, ni, ntiles, ngroups, pitch_bytes / 16);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 2.0 * 614.4e6 / (ms / 10 * 1e-3) / 1e9;
}
int main() {
    const size_t row = 600000;
    const std::vector<size_t> extra = {0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 55360 /* 640 KiB */, 65536, 131072, 448576 /* 1 MiB */};
    const size_t maxpitch = row + 448576;
    for (int t = 0; t < 6; ++t) {
        void *a, *b;
        CK(hipMalloc(&a, maxpitch * 1024 + (2 << 20))); CK(hipMalloc(&b, maxpitch * 1024 + (2 << 20)));
        if (t >= 2) {   // does touching the arrays with a linear fill first change anything?
            const size_t n = (maxpitch * 1024) / 16;
            fill<<<uint32_t((n + kThreads - 1) / kThreads), kThreads>>>((float4 *)a, n);
            fill<<<uint32_t((n + kThreads - 1) / kThreads), kThreads>>>((float4 *)b, n);
            CK(hipDeviceSynchronize());
        }
        std::printf("placement %d%s:", t, t >= 2 ? " (pre-filled)" : "");
        for (size_t x : extra) std::printf(" +%zu:%5.0f", x, run(a, b, row + x));
        std::printf("\n");
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
