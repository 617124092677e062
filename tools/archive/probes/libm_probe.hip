// Exhaustive check, over all 2^32 float bit patterns, of two substitutions in the bone solver's arithmetic:
//   float(sqrt(double(x)))  ==  sqrtf(x)          (correctly rounded f32 sqrt; double rounding is innocuous
//                                                   for sqrt when the wide format has >= 2p+2 bits)
//   sin(double(x)), cos(double(x))  ==  the two results of sincos(double(x))
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/archive/probes/libm_probe.hip -o tools/archive/probes/libm_probe && tools/archive/probes/libm_probe
// With a file argument: replay mode, see replay() below.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

__global__ void probe(unsigned long long *bad) {
    const uint64_t tid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    unsigned long long b_sqrt = 0, b_sin = 0, b_cos = 0;
    for (uint64_t bits = tid; bits < (1ull << 32); bits += stride) {
        const float x = __uint_as_float(uint32_t(bits));
        const float a = float(sqrt(double(x))), b = sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) ++b_sqrt;
        const double s = sin(double(x)), c = cos(double(x));
        double s2, c2;
        sincos(double(x), &s2, &c2);
        if (__double_as_longlong(s) != __double_as_longlong(s2) && !(s != s && s2 != s2)) ++b_sin;
        if (__double_as_longlong(c) != __double_as_longlong(c2) && !(c != c && c2 != c2)) ++b_cos;
    }
    if (b_sqrt) atomicAdd(bad + 0, b_sqrt);
    if (b_sin) atomicAdd(bad + 1, b_sin);
    if (b_cos) atomicAdd(bad + 2, b_cos);
}

// Replay mode (tools/archive/probes/rig_mismatch_probe.py): records of the oracle's transcendental calls -- function id (0 sqrt,
// 1 sin, 2 cos, 3 asin, 4 acos, 5 atan2), argument bits, second argument bits, result bits -- evaluated the way
// the bone solver does on the device; prints every call whose float result differs from the host's.
__global__ void replay(const uint32_t *rec, size_t n, uint32_t *out_f, double *out_d) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = __uint_as_float(rec[4 * i + 1]), b = __uint_as_float(rec[4 * i + 2]);
    double d, other;
    switch (rec[4 * i]) {
        case 0: d = double(sqrtf(a)); break;
        case 1: sincos(double(a), &d, &other); break;
        case 2: sincos(double(a), &other, &d); break;
        case 3: d = asin(double(a)); break;
        case 4: d = acos(double(a)); break;
        default: d = atan2(double(a), double(b)); break;
    }
    out_d[i] = d;
    out_f[i] = __float_as_uint(float(d));
}

#include <cmath>
#include <vector>

static int replay_file(const char *path) {
    FILE *f = std::fopen(path, "rb");
    if (!f) { std::perror(path); return 2; }
    std::vector<uint32_t> rec;
    uint32_t buf[4096];
    size_t got;
    while ((got = std::fread(buf, 4, 4096, f)) > 0) rec.insert(rec.end(), buf, buf + got);
    std::fclose(f);
    const size_t n = rec.size() / 4;
    if (!n) { std::printf("no records\n"); return 0; }
    uint32_t *d_rec, *d_f;
    double *d_d;
    if (hipMalloc(&d_rec, n * 16) != hipSuccess || hipMalloc(&d_f, n * 4) != hipSuccess || hipMalloc(&d_d, n * 8) != hipSuccess) return 2;
    (void)hipMemcpy(d_rec, rec.data(), n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(replay, dim3(unsigned((n + 255) / 256)), dim3(256), 0, 0, d_rec, n, d_f, d_d);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    std::vector<uint32_t> gf(n);
    std::vector<double> gd(n);
    (void)hipMemcpy(gf.data(), d_f, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(gd.data(), d_d, n * 8, hipMemcpyDeviceToHost);
    static const char *names[] = {"sqrt", "sin", "cos", "asin", "acos", "atan2"};
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        float hf;
        std::memcpy(&hf, &rec[4 * i + 3], 4);
        float gff;
        std::memcpy(&gff, &gf[i], 4);
        if (gf[i] == rec[4 * i + 3] || (hf != hf && gff != gff)) continue;
        float a, b;
        std::memcpy(&a, &rec[4 * i + 1], 4);
        std::memcpy(&b, &rec[4 * i + 2], 4);
        double hd;
        switch (rec[4 * i]) {
            case 0: hd = std::sqrt(double(a)); break;
            case 1: hd = std::sin(double(a)); break;
            case 2: hd = std::cos(double(a)); break;
            case 3: hd = std::asin(double(a)); break;
            case 4: hd = std::acos(double(a)); break;
            default: hd = std::atan2(double(a), double(b)); break;
        }
        if (bad++ < 20)
            std::printf("call %zu: %s(%a%s%a): host double %a -> float %a; device double %a -> float %a\n", i,
                        names[rec[4 * i] < 6 ? rec[4 * i] : 5], a, rec[4 * i] == 5 ? ", " : " | ", b, hd, hf, gd[i], gff);
    }
    std::printf("%zu transcendental calls replayed on the device: %zu give a different float than the host's libm\n", n, bad);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1) return replay_file(argv[1]);
    unsigned long long *d, h[3] = {0, 0, 0};
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    (void)hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(256 * 16), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::printf("all 2^32 float bit patterns: sqrtf vs float(sqrt(double)) mismatches %llu; sincos vs sin %llu; sincos vs cos %llu\n",
                h[0], h[1], h[2]);
    return (h[0] || h[1] || h[2]) ? 1 : 0;
}
