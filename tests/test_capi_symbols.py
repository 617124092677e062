"""CPU: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports
every entry point include/mmdx.h declares.  No compute calls."""
import ctypes as C
import os
import re
import subprocess

import pytest

from simple_mmd_renderer_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mmdx.h")                 # the drop-in boundary
BENCH_HEADER = os.path.join(ROOT, "include", "mmdx_bench.h")     # timers / ceilings / A-B knobs: not the boundary


def declared_symbols(headers=(HEADER, BENCH_HEADER)):
    text = "".join(open(h).read() for h in headers)
    return sorted(set(re.findall(r"MMDX_API\s+[\w\s\*]+?\b(mmdx_\w+)\s*\(", text)))


def test_boundary_header_carries_no_bench_or_debug_entry_points():
    syms = declared_symbols((HEADER,))
    assert not [s for s in syms if s.startswith(("mmdx_bench_", "mmdx_debug_", "mmdx_timer_", "mmdx_profile_"))]
    assert set(declared_symbols((BENCH_HEADER,))) == {
        "mmdx_timer_start", "mmdx_timer_stop", "mmdx_profile_enable", "mmdx_profile_collect", "mmdx_debug_reload_env",
        "mmdx_debug_last_store_policy", "mmdx_debug_morph_pass_stats", "mmdx_build_source_sha",
        "mmdx_bench_copy", "mmdx_bench_fill", "mmdx_bench_store_pattern"}


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("mmdx_model_create", "mmdx_deform", "mmdx_deform_vertex32", "mmdx_deform_batched",
                 "mmdx_model_destroy", "mmdx_last_error_string"):
        assert must in syms
    assert len(syms) >= 25


def test_library_exports_every_declared_symbol(hip_lib):
    for name in declared_symbols():
        assert hasattr(hip_lib, name), f"{name} declared in include/*.h but not exported by libmmdx.so"
    # and the Python binding covers exactly the header
    assert sorted(_capi.SIGNATURES) == declared_symbols()


def test_library_contains_gfx950_code_object(hip_lib):
    out = subprocess.run(["strings", "-n", "6", _capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    assert "deform_kernel" in out


def test_struct_sizes_match_header():
    # 64-bit layout computed by hand from the header: 6 x u32 + 12 pointers; 4 x u32 + 4 ptr + f32 + u32
    assert C.sizeof(_capi.ModelDesc) == 6 * 4 + 12 * 8
    assert C.sizeof(_capi.DeformArgs) == 4 * 4 + 4 * 8 + 8
    assert C.sizeof(_capi.ModelInfo) == 80   # 13 u32, pad to 8, u64, 3 u32, tail pad


def test_library_is_stamped_with_the_source_revision_of_this_tree(hip_lib):
    """The binary proves which sources it was built from (mmdx_build_source_sha): the stamp read through the ABI, the stamp
    found in the file's bytes and the hash of the tree's sources are one and the same; build() rebuilds on content, not mtime."""
    from simple_mmd_renderer_amd import build
    got = _capi.check_library_matches_tree()
    assert got["library_source_sha"] == got["tree_source_sha"] == build.library_sha() == build.source_sha()
    assert len(got["library_source_sha"]) == 40
    assert build.up_to_date()


def test_abi_version_and_error_string(hip_lib):
    assert hip_lib.mmdx_abi_version() == 3
    st = hip_lib.mmdx_model_create(None, None)
    assert st == 1
    assert b"NULL" in hip_lib.mmdx_last_error_string()
