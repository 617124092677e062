// valu_probe.hip -- issue cost of packed vs scalar f32 VALU instructions on gfx950, one workgroup on one CU.
// Each wave runs a loop of 16 INDEPENDENT instructions of one kind (8 accumulator registers / register pairs, two
// rounds); reported: s_memtime ticks per instruction for 1, 2 and 4 waves per SIMD, relative to v_mul_f32.
//   hipcc --offload-arch=gfx950 -O2 tools/archive/probes/valu_probe.hip -o tools/archive/probes/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(1024) void probe(unsigned long long *ticks, float *sink, int iters) {
    v2f a[8], b = v2f{1.0000001f, 0.9999999f}, c = v2f{1e-9f, -1e-9f};
    float s[8], sb = 1.0000001f, sc = 1e-9f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = v2f{1.f + i, 2.f + i} * float(threadIdx.x + 1); s[i] = 1.f + i + threadIdx.x; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if constexpr (OP == 0) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(sb));
                REP8(X)
#undef X
            } else if constexpr (OP == 1) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == 2) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if constexpr (OP == 3) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (OP == 4) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(sb), "v"(sc));
                REP8(X)
#undef X
            } else if constexpr (OP == 5) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(sc));
                REP8(X)
#undef X
            } else if constexpr (OP == 6) {   // pk_mul with a scalar (SGPR-free) broadcast operand: op_sel picks the low half twice
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == 7) {   // dependent chain of scalar mul (latency)
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[0]) : "v"(sb));
                REP8(X)
#undef X
            } else if constexpr (OP == 8) {   // dependent chain of pk_mul
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == 9) {   // f64 fma
                double *d = reinterpret_cast<double *>(a);
                const double db = 1.0000001, dc = 1e-9;
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = t1 - t0;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i];
    sink[threadIdx.x] = acc;
}

template <int OP>
double run(int waves_per_simd, unsigned long long *dt, float *ds, int iters) {
    const int threads = 64 * 4 * waves_per_simd;
    hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(threads), 0, 0, dt, ds, iters);
    hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(threads), 0, 0, dt, ds, iters);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, dt, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
    return double(mx) / (double(iters) * 16.0 * waves_per_simd);   // ticks per instruction per SIMD
}

int main() {
    unsigned long long *dt;
    float *ds;
    hipMalloc(&dt, 16 * sizeof(unsigned long long));
    hipMalloc(&ds, 1024 * sizeof(float));
    const int iters = 20000;
    const char *names[] = {"v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_fma_f32", "v_add_f32",
                           "v_pk_mul_f32 op_sel_hi:[1,0]", "v_mul_f32 dependent", "v_pk_mul_f32 dependent", "v_fma_f64"};
    double base[3] = {0, 0, 0};
    for (int op = 0; op < 10; ++op) {
        double r[3];
        int k = 0;
        for (int w = 1; w <= 4; w *= 2, ++k) {
            switch (op) {
            case 0: r[k] = run<0>(w, dt, ds, iters); break;
            case 1: r[k] = run<1>(w, dt, ds, iters); break;
            case 2: r[k] = run<2>(w, dt, ds, iters); break;
            case 3: r[k] = run<3>(w, dt, ds, iters); break;
            case 4: r[k] = run<4>(w, dt, ds, iters); break;
            case 5: r[k] = run<5>(w, dt, ds, iters); break;
            case 6: r[k] = run<6>(w, dt, ds, iters); break;
            case 7: r[k] = run<7>(w, dt, ds, iters); break;
            case 8: r[k] = run<8>(w, dt, ds, iters); break;
            default: r[k] = run<9>(w, dt, ds, iters); break;
            }
            if (op == 0) base[k] = r[k];
        }
        std::printf("%-32s ticks/instr/SIMD  1 wave %.3f (x%.2f)  2 waves %.3f (x%.2f)  4 waves %.3f (x%.2f)\n", names[op],
                    r[0], r[0] / base[0], r[1], r[1] / base[1], r[2], r[2] / base[2]);
    }
    return 0;
}
