// tools/archive/probes/bw_probe.hip -- store/copy bandwidth probe for gfx950: which write shape reaches the HBM
// ceiling?  Measurement tool only (not part of the product).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ inline void nt_store(float4 v, float4 *p) { vf4 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<vf4 *>(p)); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void fill_stride(float4 *d, size_t n) {
    const float4 v = make_float4(1, 2, 3, 4);
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) d[i] = v;
}
__global__ __launch_bounds__(256) void fill_stride_nt(float4 *d, size_t n) {
    const float4 v = make_float4(1, 2, 3, 4);
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256)
        nt_store(v, &d[i]);
}
// block b owns a contiguous chunk of `chunk` float4
template <bool NT>
__global__ __launch_bounds__(256) void fill_chunk(float4 *d, size_t n, size_t chunk) {
    const float4 v = make_float4(1, 2, 3, 4);
    size_t base = size_t(blockIdx.x) * chunk;
    size_t end = base + chunk < n ? base + chunk : n;
    for (size_t i = base + threadIdx.x; i < end; i += 256) {
        if (NT) nt_store(v, &d[i]); else d[i] = v;
    }
}
// one dword per lane
__global__ __launch_bounds__(256) void fill_dword(float *d, size_t n) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) d[i] = 1.0f;
}
// deform-like: block (tile, group) writes, per instance of its group, one 6 KB piece in each of two arrays
__global__ __launch_bounds__(256) void fill_deformlike(float4 *a, float4 *b, int ntiles, int ni, int group, int piece4, size_t inst_stride4) {
    const float4 v = make_float4(1, 2, 3, 4);
    int tile = blockIdx.x, g0 = blockIdx.y * group;
    for (int g = g0; g < g0 + group && g < ni; ++g) {
        size_t base = size_t(g) * inst_stride4 + size_t(tile) * piece4;
        for (int q = threadIdx.x; q < 2 * piece4; q += 256) {
            if (q < piece4) a[base + q] = v; else b[base + q - piece4] = v;
        }
    }
}
__global__ __launch_bounds__(256) void copy_stride(float4 *d, const float4 *s, size_t n) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) d[i] = s[i];
}
template <int U>
__global__ __launch_bounds__(256) void copy_unroll(float4 *d, const float4 *s, size_t n) {
    size_t stride = size_t(gridDim.x) * 256;
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) d[i + u * stride] = r[u];
    }
    for (; i < n; i += stride) d[i] = s[i];
}


// class-sorted scatter: lane s of a 512-vertex tile stores 12 B at perm[s] (perm ascending inside each
// of 3 class segments) into two arrays -- what a deform kernel WITHOUT LDS staging would issue.
struct f3 { float x, y, z; };
template <int MODE>   // 0: dwordx3 stores, 1: three dword stores, 2: interleaved 32B vertex (2 x float4)
__global__ __launch_bounds__(256) void fill_scatter(float *a, float *b, const unsigned short *perm, int ni, int group, size_t nv) {
    int tile = blockIdx.x, g0 = blockIdx.y * group;
    unsigned p0 = perm[tile * 512 + threadIdx.x], p1 = perm[tile * 512 + 256 + threadIdx.x];
    for (int g = g0; g < g0 + group && g < ni; ++g) {
        size_t vb = size_t(g) * nv + size_t(tile) * 512;
        for (int k = 0; k < 2; ++k) {
            size_t v = vb + (k ? p1 : p0);
            if (MODE == 0) {
                *reinterpret_cast<f3 *>(a + v * 3) = f3{1.f, 2.f, float(g)};
                *reinterpret_cast<f3 *>(b + v * 3) = f3{3.f, 4.f, float(g)};
            } else if (MODE == 1) {
                a[v * 3] = 1.f; a[v * 3 + 1] = 2.f; a[v * 3 + 2] = float(g);
                b[v * 3] = 3.f; b[v * 3 + 1] = 4.f; b[v * 3 + 2] = float(g);
            } else {
                float4 *o = reinterpret_cast<float4 *>(a) + v * 2;
                o[0] = make_float4(1, 2, 3, float(g)); o[1] = make_float4(4, 5, 6, 7);
            }
        }
    }
}


// deform-like with the real kernel's structure knobs: dynamic LDS (occupancy), per-instance barrier,
// store data read back from LDS.  MODE bit0: barrier, bit1: LDS reads feed the stores.
template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void fill_deformlike2(float4 *a, float4 *b, int ni, int group, int piece4, size_t inst_stride4) {
    extern __shared__ float4 lds4[];
    int tile = blockIdx.x, g0 = blockIdx.y * group;
    if (MODE & 2) for (int i = threadIdx.x; i < 2 * piece4; i += THREADS) lds4[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    for (int g = g0; g < g0 + group && g < ni; ++g) {
        if (MODE & 1) __syncthreads();
        size_t base = size_t(g) * inst_stride4 + size_t(tile) * piece4;
        constexpr int PER = (768 + THREADS - 1) / THREADS;
        float4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) { int q = threadIdx.x + i * THREADS; if (q < 2 * piece4) v[i] = (MODE & 2) ? lds4[q] : make_float4(1, 2, 3, g); }
#pragma unroll
        for (int i = 0; i < PER; ++i) { int q = threadIdx.x + i * THREADS; if (q < 2 * piece4) { if (q < piece4) a[base + q] = v[i]; else b[base + q - piece4] = v[i]; } }
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
    const size_t bytes = size_t(1200) << 20;   // 1.2 GiB, well past the 256 MiB Infinity Cache
    const size_t n4 = bytes / 16;
    float4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    auto rep = [&](const char *name, float ms, double moved) { printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, moved / (ms * 1e-3) / 1e9); };
    rep("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(a, 1, bytes)); }), bytes);
    for (int blocks : {1024, 2048, 4096, 8192, 32768, 262144}) {
        char nm[64]; snprintf(nm, 64, "fill_stride blocks=%d", blocks);
        rep(nm, timeit([&] { fill_stride<<<blocks, 256>>>(a, n4); }), bytes);
    }
    for (int blocks : {2048, 8192}) {
        char nm[64]; snprintf(nm, 64, "fill_stride_nt blocks=%d", blocks);
        rep(nm, timeit([&] { fill_stride_nt<<<blocks, 256>>>(a, n4); }), bytes);
    }
    for (size_t chunkKB : {4, 16, 64, 256, 1024}) {
        size_t chunk = chunkKB * 1024 / 16; int blocks = int((n4 + chunk - 1) / chunk);
        char nm[64]; snprintf(nm, 64, "fill_chunk %zuKB blocks=%d", chunkKB, blocks);
        rep(nm, timeit([&] { fill_chunk<false><<<blocks, 256>>>(a, n4, chunk); }), bytes);
        snprintf(nm, 64, "fill_chunk_nt %zuKB blocks=%d", chunkKB, blocks);
        rep(nm, timeit([&] { fill_chunk<true><<<blocks, 256>>>(a, n4, chunk); }), bytes);
    }
    rep("fill_dword blocks=8192", timeit([&] { fill_dword<<<8192, 256>>>((float *)a, n4 * 4); }), bytes);
    // deform-like: 98 tiles x 6 KB pieces, 1024 instances, two arrays of 600 000 B rows
    {
        const int ntiles = 98, ni = 1024, piece4 = 384; const size_t stride4 = 37500;  // 50 000*12/16
        for (int group : {1, 4, 8, 16, 32}) {
            char nm[64]; snprintf(nm, 64, "deformlike group=%d", group);
            rep(nm, timeit([&] { fill_deformlike<<<dim3(ntiles, (ni + group - 1) / group), 256>>>(a, b, ntiles, ni, group, piece4, stride4); }),
                2.0 * ni * ntiles * piece4 * 16);
        }
    }


    {
        const int ntiles = 98, ni = 1024, piece4 = 384, group = 16; const size_t stride4 = 37500;
        CK(hipFuncSetAttribute((const void*)fill_deformlike2<256,0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
        CK(hipFuncSetAttribute((const void*)fill_deformlike2<256,1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
        CK(hipFuncSetAttribute((const void*)fill_deformlike2<256,3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
        CK(hipFuncSetAttribute((const void*)fill_deformlike2<512,3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
        for (int ldsKB : {13, 20, 27, 41, 64}) {
            char nm[80]; dim3 grid(ntiles, ni / group);
            snprintf(nm, 80, "deformlike2 T256 lds=%dKB plain", ldsKB);
            rep(nm, timeit([&] { fill_deformlike2<256,0><<<grid, 256, ldsKB*1024>>>(a, b, ni, group, piece4, stride4); }), 2.0 * ni * ntiles * piece4 * 16);
            snprintf(nm, 80, "deformlike2 T256 lds=%dKB +barrier", ldsKB);
            rep(nm, timeit([&] { fill_deformlike2<256,1><<<grid, 256, ldsKB*1024>>>(a, b, ni, group, piece4, stride4); }), 2.0 * ni * ntiles * piece4 * 16);
            snprintf(nm, 80, "deformlike2 T256 lds=%dKB +barrier+ldsread", ldsKB);
            rep(nm, timeit([&] { fill_deformlike2<256,3><<<grid, 256, ldsKB*1024>>>(a, b, ni, group, piece4, stride4); }), 2.0 * ni * ntiles * piece4 * 16);
            snprintf(nm, 80, "deformlike2 T512 lds=%dKB +barrier+ldsread", ldsKB);
            rep(nm, timeit([&] { fill_deformlike2<512,3><<<grid, 512, ldsKB*1024>>>(a, b, ni, group, piece4, stride4); }), 2.0 * ni * ntiles * piece4 * 16);
        }
    }
    {   // scattered stores, class-sorted perm (20/55/25 mix), 97 full tiles x 1024 instances
        const int ntiles = 97, ni = 1024; const size_t nv = size_t(ntiles) * 512;
        std::vector<unsigned short> perm(ntiles * 512);
        srand(1);
        for (int t = 0; t < ntiles; ++t) {
            std::vector<int> cls(512); for (auto &c : cls) { int r = rand() % 100; c = r < 20 ? 0 : (r < 75 ? 1 : 2); }
            int s = 0; for (int c = 0; c < 3; ++c) for (int l = 0; l < 512; ++l) if (cls[l] == c) perm[t * 512 + s++] = (unsigned short)l;
        }
        unsigned short *dperm; CK(hipMalloc(&dperm, perm.size() * 2)); CK(hipMemcpy(dperm, perm.data(), perm.size() * 2, hipMemcpyHostToDevice));
        for (int group : {4, 16}) {
            char nm[64];
            snprintf(nm, 64, "scatter dwordx3 SoA group=%d", group);
            rep(nm, timeit([&] { fill_scatter<0><<<dim3(ntiles, ni / group), 256>>>((float *)a, (float *)b, dperm, ni, group, nv); }), 24.0 * ni * nv);
            snprintf(nm, 64, "scatter 3xdword SoA group=%d", group);
            rep(nm, timeit([&] { fill_scatter<1><<<dim3(ntiles, ni / group), 256>>>((float *)a, (float *)b, dperm, ni, group, nv); }), 24.0 * ni * nv);
            snprintf(nm, 64, "scatter vertex32 group=%d", group);
            rep(nm, timeit([&] { fill_scatter<2><<<dim3(ntiles, ni / group), 256>>>((float *)a, (float *)b, dperm, 512, group, nv); }), 32.0 * 512 * nv);
        }
    }
    for (int blocks : {2048, 8192, 32768}) {
        char nm[64]; snprintf(nm, 64, "copy_stride blocks=%d", blocks);
        rep(nm, timeit([&] { copy_stride<<<blocks, 256>>>(b, a, n4); }), 2.0 * bytes);
    }
    rep("copy_unroll<4> blocks=2048", timeit([&] { copy_unroll<4><<<2048, 256>>>(b, a, n4); }), 2.0 * bytes);
    rep("copy_unroll<8> blocks=2048", timeit([&] { copy_unroll<8><<<2048, 256>>>(b, a, n4); }), 2.0 * bytes);
    rep("copy_unroll<4> blocks=8192", timeit([&] { copy_unroll<4><<<8192, 256>>>(b, a, n4); }), 2.0 * bytes);
    rep("hipMemcpyDtoD", timeit([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice)); }), 2.0 * bytes);
    return 0;
}
