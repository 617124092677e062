#!/usr/bin/env python3
"""Coefficients of cr_math.hpp (this directory) (the polynomial functions between its GENERATED markers).

    python3 tools/archive/probes/cr_math/gen_cr_math.py            # prints the tables and the approximation errors

Each table is the interpolant of its function at Chebyshev nodes of the interval (within a small factor of the minimax
polynomial), solved in 80-digit arithmetic (mpmath) and rounded to double; the error printed is that of the ROUNDED table,
in exact arithmetic, relative to the function the table is used for.  cr_math.hpp needs 2^-43 in total (tables + evaluation).
"""
import mpmath as mp

mp.mp.dps = 80


def fit(g, a, b, n):
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        y[i] = g(x)
    c = mp.lu_solve(A, y)
    return [float(c[i]) for i in range(n)]


def worst(cd, a, b, value, truth, npts=3000):
    w = mp.mpf(0)
    for k in range(1, npts + 1):
        z = a + (b - a) * mp.mpf(k) / npts
        p = sum(mp.mpf(cd[j]) * z ** j for j in range(len(cd)))
        w = max(w, abs(value(z, p) - truth(z)) / abs(truth(z)))
    return float(mp.log(w, 2))


def sinc(z):
    x = mp.sqrt(z)
    return mp.sin(x) / x


X2 = mp.mpf("1.6") ** 2
TABLES = [
    # name, g(z), interval, size, value(z, p), truth(z), what
    ("poly_sin", lambda z: (sinc(z) - 1) / z, (mp.mpf(0), X2), 8, lambda z, p: 1 + z * p, sinc,
     "sin(x) = x + x z S(z), z = x^2 <= 1.6^2; error relative to sin"),
    ("poly_cos", lambda z: (mp.cos(mp.sqrt(z)) - 1) / z, (mp.mpf(0), X2), 8, lambda z, p: 2 + z * p, lambda z: 1 + mp.cos(mp.sqrt(z)),
     "cos(x) = 1 + z C(z); ABSOLUTE error (shown relative to 1 + cos)"),
    ("poly_asin", lambda z: (mp.asin(mp.sqrt(z)) / mp.sqrt(z) - 1) / z, (mp.mpf(0), mp.mpf(1) / 4), 12, lambda z, p: 1 + z * p,
     lambda z: mp.asin(mp.sqrt(z)) / mp.sqrt(z), "asin(s) = s + s z R(z), z = s^2 <= 1/4; error relative to asin"),
    ("poly_atan", lambda z: (mp.atan(mp.sqrt(z)) / mp.sqrt(z) - 1) / z, (mp.mpf(0), mp.mpf(1)), 20, lambda z, p: 1 + z * p,
     lambda z: mp.atan(mp.sqrt(z)) / mp.sqrt(z), "atan(t) = t + t z A(z), z = t^2 <= 1; error relative to atan"),
]


def estrin(cd, lo, n):
    """c[lo] + c[lo+1] z + ... (n terms) as nested fused multiply-adds over z, z^2, z^4, ... (depth ~log2 n)."""
    if n == 1:
        return float.hex(cd[lo])
    h = 1
    while h * 2 < n:
        h *= 2
    return "fma(%s, z%d, %s)" % (estrin(cd, lo + h, n - h), h, estrin(cd, lo, h))


def main():
    for name, g, (a, b), n, value, truth, what in TABLES:
        gz = lambda z, g=g: g(z) if z != 0 else g(mp.mpf(10) ** -40)
        cd = fit(gz, a, b, n)
        print("// %s; table error 2^%.1f" % (what, worst(cd, a, b, value, truth)))
        print("MMDX_HD inline double %s(double z1) {" % name)
        k = 2
        while k < n:
            print("    const double z%d = z%d * z%d;" % (k, k // 2, k // 2))
            k *= 2
        expr = estrin(cd, 0, n)
        # one fma per line, innermost first, would be unreadable: keep the nested expression, wrapped
        out, line = [], "    return "
        for tok in expr.replace(", ", ",\x00").split("\x00"):
            if len(line) + len(tok) > 118:
                out.append(line.rstrip())
                line = "           "
            line += tok + " "
        out.append(line.rstrip() + ";")
        print("\n".join(out))
        print("}")
    for nm, v in (("kPio2", mp.pi / 2), ("kPi", mp.pi)):
        hi = float(v)
        lo = float(v - mp.mpf(hi))
        print("constexpr double %sHi = %s, %sLo = %s;" % (nm, float.hex(hi), nm, float.hex(lo)))


if __name__ == "__main__":
    main()
