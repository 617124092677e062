"""cr_math.hpp: the bone solve's sin / cos / asin / acos / atan2 -- short polynomials that settle on the float the reference's
`float(libm(double(x)))` (L/util/math.inl:28-45) gives wherever they can tell which float that is, the general routine otherwise.

CPU: the header compiled for the host against glibc (tools/cr_math_check.cpp; every 61st float here, every float when the tool is
run without a stride: 0 mismatches, worst distance 3 ulp(double) of the 1 024 the decision allows -- 131 for cos next to its
zero).  GPU: the functions as the kernels call them against the general device routine alone, for EVERY float argument and 2^31
argument pairs of atan2.
"""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_build_of_cr_math_against_glibc(tmp_path):
    exe = str(tmp_path / "cr_math_check")
    fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fopenmp", *fma, os.path.join(ROOT, "tools", "cr_math_check.cpp"),
                    "-o", exe], check=True)
    r = subprocess.run([exe, "3000000", "61"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = {ln.split()[0]: ln for ln in r.stdout.splitlines() if "mismatches" in ln}
    assert set(lines) == {"sin", "cos", "asin", "acos", "atan2"}
    for name, ln in lines.items():
        assert " mismatches 0 " in ln, ln
        worst = float(ln.split("glibc")[1].split()[0])
        assert worst <= 256.0, ln                                   # the decision allows 1 024


def test_generated_part_of_cr_math_is_the_generator_s_output():
    """The polynomial functions in the header are tools/gen_cr_math.py's output, not hand-edited."""
    r = subprocess.run(["python3", os.path.join(ROOT, "tools", "gen_cr_math.py")], capture_output=True, text=True, check=True)
    hdr = open(os.path.join(ROOT, "simple_mmd_renderer_amd", "csrc", "cr_math.hpp")).read()
    a = hdr.index("// ---- GENERATED"); b = hdr.index("// ---- end of generated part")
    body = hdr[hdr.index("\n", a) + 1:b].strip()
    assert body == r.stdout.strip()


@pytest.mark.gpu
def test_device_functions_equal_the_libm_route_for_every_float():
    from simple_mmd_renderer_amd import _capi
    lib = _capi.lib()
    bad, declined = C.c_uint64(0), C.c_uint64(0)
    names = ["sin", "cos", "asin", "acos"]
    for fn, name in enumerate(names):
        total_declined = 0
        for part in range(4):
            _capi.check(lib.mmdx_debug_cr_math_check(fn, part << 30, 1 << 30, C.byref(bad), C.byref(declined)))
            assert bad.value == 0, (name, part, bad.value)
            total_declined += declined.value
        # outside the range the polynomials cover (|x| > 1.6 resp. > 1, NaN, tiny values) everything is declined: most floats
        assert 0 < total_declined < (1 << 32), (name, total_declined)
    _capi.check(lib.mmdx_debug_cr_math_check(4, 0, 1 << 31, C.byref(bad), C.byref(declined)))
    assert bad.value == 0
    assert declined.value < (1 << 31) // 4


@pytest.mark.gpu
def test_cr_math_declines_one_call_in_many_thousands_inside_the_solver_s_ranges():
    """Floats in [0.5, 1) (a link step's clamped dot product; half angles): the polynomials answer all but a few per million."""
    from simple_mmd_renderer_amd import _capi
    lib = _capi.lib()
    bad, declined = C.c_uint64(0), C.c_uint64(0)
    first, count = 0x3F000000, 0x00800000                            # [0.5, 1.0): 2^23 floats
    for fn in range(4):
        _capi.check(lib.mmdx_debug_cr_math_check(fn, first, count, C.byref(bad), C.byref(declined)))
        assert bad.value == 0
        assert declined.value <= count // 20000, (fn, declined.value)
