// Which SIMD does wave w of a workgroup land on?  (LAB_NOTES R3.7: the class sort puts the heavy BDEF4 blocks in the last waves
// of a workgroup; whether that loads the SIMDs of a CU unequally depends on how the dispatcher deals a workgroup's waves.)
// Launches workgroups of 256 / 512 threads with the LDS footprint of the crowd / per-instance-morph kernels, every wave
// records HW_ID (wave, SIMD, CU, SE) and XCC_ID, spins ~20 us so that the CU fills, and the host prints the wave -> SIMD table.
//   hipcc --offload-arch=gfx950 -O3 tools/archive/probes/simd_map_probe.hip -o tools/archive/probes/simd_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void probe(uint32_t *out, int spin) {
    extern __shared__ unsigned char smem[];
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
    long long t0 = wall_clock64();
    float x = float(threadIdx.x);
    while (wall_clock64() - t0 < spin) x = x * 1.0001f + 1.f;
    if (x == 12345.f) smem[threadIdx.x] = 1;
    if ((threadIdx.x & 63) == 0) {
        uint32_t *o = out + (size_t(blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64) * 2;
        o[0] = hw; o[1] = xcc;
    }
}

int main() {
    for (int threads : {256, 512}) {
        const int lds = threads == 256 ? 42 * 1024 : 64 * 1024, nwg = 2048, wpw = threads / 64;
        uint32_t *d;
        hipMalloc(&d, size_t(nwg) * wpw * 8);
        hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(probe, dim3(nwg), dim3(threads), lds, 0, d, 2000);   // 100 MHz wall clock: 2000 ticks = 20 us
        std::vector<uint32_t> h(size_t(nwg) * wpw * 2);
        if (hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { std::puts("copy failed"); return 1; }
        std::printf("== %d threads, %d KB LDS: wave-in-workgroup -> SIMD histogram over %d workgroups\n", threads, lds / 1024, nwg);
        std::vector<std::vector<int>> hist(wpw, std::vector<int>(4, 0));
        std::map<std::vector<int>, int> patterns;
        for (int g = 0; g < nwg; ++g) {
            std::vector<int> pat;
            for (int w = 0; w < wpw; ++w) {
                const uint32_t hw = h[(size_t(g) * wpw + w) * 2];
                const int simd = (hw >> 4) & 3;
                ++hist[w][simd];
                pat.push_back(simd);
            }
            ++patterns[pat];
        }
        for (int w = 0; w < wpw; ++w)
            std::printf("  wave %d: SIMD0 %4d  SIMD1 %4d  SIMD2 %4d  SIMD3 %4d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        std::printf("  patterns (SIMD of wave 0..%d : workgroups):", wpw - 1);
        int shown = 0;
        for (auto &kv : patterns) {
            if (shown++ == 12) { std::printf(" ... (%zu patterns)", patterns.size()); break; }
            std::printf("  ");
            for (int s : kv.first) std::printf("%d", s);
            std::printf(":%d", kv.second);
        }
        std::printf("\n  first workgroups (hw_id: wave/simd/cu/se, xcc):");
        for (int g = 0; g < 4; ++g) {
            std::printf("\n    wg %d:", g);
            for (int w = 0; w < wpw; ++w) {
                const uint32_t hw = h[(size_t(g) * wpw + w) * 2], xc = h[(size_t(g) * wpw + w) * 2 + 1];
                std::printf(" [%u/%u/%u/%u x%u]", hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, xc & 15);
            }
        }
        std::printf("\n");
        hipFree(d);
    }
    return 0;
}
