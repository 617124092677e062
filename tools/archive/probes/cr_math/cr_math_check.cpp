// cr_math_check.cpp -- host-side proof run for simple_mmd_renderer_amd/csrc/cr_math.hpp (the same code the kernels compile):
//   g++ -O2 -std=c++17 -mfma -ffp-contract=off -fopenmp tools/archive/probes/cr_math/cr_math_check.cpp -o /tmp/cr_math_check && /tmp/cr_math_check [pairs [stride]]
// (40 s on 8 cores; `stride` > 1 takes every stride-th float only: the unit test's short form.)  For EVERY float argument of sin / cos (|x| <= 1.6), asin and acos (|x| <= 1, and a band outside), and `pairs` (default 4e9)
// argument pairs of atan2 (random bit patterns, random pairs of nearby magnitudes, zeros, tiny and huge values): wherever the
// fast routine settles, its float must equal float(libm(double(x))) -- what the reference computes (L/util/math.inl:28-45).
// Also reports how far the double value was from glibc's (in units of the double's last place; glibc itself is within 1)
// against the 1 024 the decision allows, and how often the routines decline.  Exit code 1 on any mismatch.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "cr_math.hpp"

using namespace mmdx::crm;

static inline float f_of(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t b_of(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
static inline bool same(float a, float b) { return b_of(a) == b_of(b) || (a != a && b != b); }
static inline double ulps(double got, double want) {
    if (want == 0.0 || got == want) return 0.0;
    int e; frexp(want, &e);
    return fabs(got - want) / ldexp(1.0, e - 53);
}

struct Tally { unsigned long long n = 0, settled = 0, bad = 0; double worst = 0; };
static void report(const char *name, const Tally &t) {
    printf("%-8s arguments %12llu  settled %12llu (declined %.3g per million)  mismatches %llu  worst distance from glibc %.1f ulp(double)\n",
           name, t.n, t.settled, t.n ? 1e6 * double(t.n - t.settled) / double(t.n) : 0.0, t.bad, t.worst);
}

int main(int argc, char **argv) {
    const unsigned long long pairs = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4000000000ull;
    const long long stride = argc > 2 ? atoll(argv[2]) : 1;
    unsigned long long total_bad = 0;
    // ---- one-argument functions: every float ---------------------------------------------------------------
    Tally ts, tc, tas, tac;
#pragma omp parallel
    {
        Tally ls, lc, las, lac;
#pragma omp for schedule(static)
        for (long long i0 = 0; i0 < (1ll << 32); i0 += stride) {
            const long long i = i0;
            const float x = f_of(uint32_t(i));
            const double xd = x;
            if (fabs(xd) <= 1.7) {        // a band beyond 1.6: must decline there
                float s, c; double raw[2];
                const bool ok = sincos_fast(x, s, c, raw);
                float s1; const bool ok1 = sin_fast(x, s1);
                const double ws = sin(xd), wc = cos(xd);
                ++ls.n; ++lc.n;
                if (fabs(xd) > 1.6 && (ok || ok1)) { ++ls.bad; }
                if (ok) {
                    ++ls.settled; ++lc.settled;
                    if (!same(s, float(ws))) ++ls.bad;
                    if (!same(c, float(wc))) ++lc.bad;
                    ls.worst = fmax(ls.worst, ulps(raw[0], ws)); lc.worst = fmax(lc.worst, ulps(raw[1], wc));
                }
                if (ok1 && !same(s1, float(ws))) ++ls.bad;
            }
            if (fabs(xd) <= 1.1 || x != x) {
                float a; double raw;
                bool ok = asin_fast(x, a, &raw);
                double w = asin(xd);
                ++las.n;
                if ((fabs(xd) > 1.0 || x != x) && ok) ++las.bad;
                if (ok) { ++las.settled; if (!same(a, float(w))) ++las.bad; las.worst = fmax(las.worst, ulps(raw, w)); }
                ok = acos_fast(x, a, &raw);
                w = acos(xd);
                ++lac.n;
                if ((fabs(xd) > 1.0 || x != x) && ok) ++lac.bad;
                if (ok) { ++lac.settled; if (!same(a, float(w))) ++lac.bad; lac.worst = fmax(lac.worst, ulps(raw, w)); }
            }
        }
#pragma omp critical
        {
            ts.n += ls.n; ts.settled += ls.settled; ts.bad += ls.bad; ts.worst = fmax(ts.worst, ls.worst);
            tc.n += lc.n; tc.settled += lc.settled; tc.bad += lc.bad; tc.worst = fmax(tc.worst, lc.worst);
            tas.n += las.n; tas.settled += las.settled; tas.bad += las.bad; tas.worst = fmax(tas.worst, las.worst);
            tac.n += lac.n; tac.settled += lac.settled; tac.bad += lac.bad; tac.worst = fmax(tac.worst, lac.worst);
        }
    }
    report("sin", ts); report("cos", tc); report("asin", tas); report("acos", tac);
    total_bad += ts.bad + tc.bad + tas.bad + tac.bad;
    // ---- atan2: argument pairs ---------------------------------------------------------------------------------
    Tally ta;
#pragma omp parallel
    {
        Tally l;
#pragma omp for schedule(static)
        for (long long i = 0; i < (long long)pairs; ++i) {
            // splitmix64 of the index: reproducible whatever the thread count
            uint64_t zz = uint64_t(i) * 0x9E3779B97F4A7C15ull + 0x1234567ull;
            auto next = [&]() { zz += 0x9E3779B97F4A7C15ull; uint64_t r = zz; r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull;
                                r = (r ^ (r >> 27)) * 0x94D049BB133111EBull; return r ^ (r >> 31); };
            const uint64_t r0 = next(), r1 = next();
            float y, x;
            switch (i & 7) {
            case 0: y = f_of(uint32_t(r0)); x = f_of(uint32_t(r0 >> 32)); break;                        // any two floats
            case 1: case 2: case 3: {                                                                   // the solver's: a rotation's products
                y = float(ldexp(double(int64_t(r0 >> 11)) , -52) - 1.0) * 2.0f;                         // (-2, 2)
                x = float(ldexp(double(int64_t(r1 >> 11)), -52) - 1.0) * 3.0f - 1.0f; break; }          // (-4, 2)
            case 4: { const float m = f_of(0x3f800000u | uint32_t(r0 & 0x7fffff));                      // nearly equal magnitudes
                      y = m * ((r0 >> 40) & 1 ? -1.f : 1.f); x = f_of(b_of(m) + int32_t((r1 & 15)) - 8) * ((r0 >> 41) & 1 ? -1.f : 1.f); break; }
            case 5: y = f_of(uint32_t(r0) & 0x807fffffu) ; x = f_of(uint32_t(r0 >> 32)); break;        // y subnormal or zero
            case 6: y = f_of(uint32_t(r0)); x = ((r1 & 3) == 0) ? 0.0f : ((r1 & 3) == 1 ? -0.0f : f_of(uint32_t(r1 >> 32) & 0x807fffffu)); break;
            default: { const int e = int(r1 % 60) - 30; y = float(ldexp(double(int64_t(r0 >> 11)), -52) - 1.0);    // ratios over 2^+-30
                       x = float(ldexp(double(int64_t(r1 >> 11)) * 0x1p-52 - 1.0, e)); break; }
            }
            float a; double raw;
            const bool ok = atan2_fast(y, x, a, &raw);
            const double w = atan2(double(y), double(x));
            ++l.n;
            if (ok) { ++l.settled; if (!same(a, float(w))) ++l.bad; l.worst = fmax(l.worst, ulps(raw, w)); }
        }
#pragma omp critical
        { ta.n += l.n; ta.settled += l.settled; ta.bad += l.bad; ta.worst = fmax(ta.worst, l.worst); }
    }
    report("atan2", ta);
    total_bad += ta.bad;
    printf("%s\n", total_bad ? "FAILED" : "all settled results equal the libm route's");
    return total_bad ? 1 : 0;
}
