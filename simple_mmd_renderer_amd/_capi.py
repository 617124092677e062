"""ctypes declarations for the C ABI in include/mmdx.h (libmmdx.so, built in-tree by build.py).

This is the ONLY compute backend: if the HIP library is missing or no GPU is usable the package
raises -- there is no CPU or eager fallback to silently pass tests on.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMDX_LIB") or os.path.join(HERE, "libmmdx.so")   # MMDX_LIB: A/B of builds (tools/)

OK = 0
ABI_VERSION = 3          # include/mmdx.h MMDX_ABI_VERSION
ERR_NAMES = {1: "INVALID_ARGUMENT", 2: "BAD_INDEX", 3: "NO_DEVICE", 4: "HIP", 5: "OUT_OF_MEMORY",
             6: "UNSUPPORTED"}

CREATE_NORMALIZE, CREATE_HOST_ONLY, CREATE_F16_POSITIONS, CREATE_FAST_MATH, CREATE_TILE_ORDER = 1, 2, 4, 8, 16
OUT_SOA, OUT_VERTEX32, OUT_SOA_POS16 = 0, 1, 2
PALETTE_ON_DEVICE, WEIGHTS_ON_DEVICE, OUT_ON_DEVICE, WEIGHTS_SHARED, MORPH_UNCHANGED = 1, 2, 4, 8, 16
OUT_STORES_WRITE_THROUGH, OUT_STORES_CACHED = 32, 64

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


class ModelDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32),
                ("n_vertices", C.c_uint32), ("n_bones", C.c_uint32), ("n_morphs", C.c_uint32),
                ("reserved0", C.c_uint32),
                ("positions", _f32p), ("normals", _f32p), ("uvs", _f32p),
                ("skin_type", _i32p), ("bone_ids", _i32p), ("bone_weights", _f32p),
                ("sdef_params", _f32p), ("bone_parent", _i32p),
                ("morph_type", _i32p), ("morph_offset", _u32p), ("morph_index", _u32p),
                ("morph_value", _f32p)]


class DeformArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32),
                ("n_instances", C.c_uint32), ("out_layout", C.c_uint32),
                ("morph_weights", C.c_void_p), ("palettes", C.c_void_p),
                ("out_a", C.c_void_p), ("out_b", C.c_void_p),
                ("pos_scale", C.c_float), ("reserved0", C.c_uint32)]


class ModelInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32),
                ("n_vertices", C.c_uint32), ("n_bones", C.c_uint32), ("n_morphs", C.c_uint32),
                ("n_slots", C.c_uint32), ("n_entries", C.c_uint32), ("n_entries_padded", C.c_uint32),
                ("n_tiles", C.c_uint32), ("tile_vertices", C.c_uint32),
                ("n_bdef1", C.c_uint32), ("n_bdef2", C.c_uint32), ("n_bdef4", C.c_uint32),
                ("max_tile_bones", C.c_uint32),
                ("device_bytes", C.c_uint64),
                ("device_ordinal", C.c_uint32), ("flags", C.c_uint32), ("reserved0", C.c_uint32)]


class MmdxError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"mmdx error {status} ({ERR_NAMES.get(status, '?')}): {message}")
        self.status = status


# every entry point include/mmdx.h and include/mmdx_bench.h declare: name -> (restype, argtypes)
SIGNATURES = {
    "mmdx_abi_version": (C.c_uint32, []),
    "mmdx_last_error_string": (C.c_char_p, []),
    "mmdx_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "mmdx_device_select": (C.c_int32, [C.c_int32]),
    "mmdx_device_name": (C.c_int32, [C.c_int32, C.c_char_p, C.c_size_t]),
    "mmdx_model_create": (C.c_int32, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "mmdx_model_destroy": (C.c_int32, [C.c_void_p]),
    "mmdx_model_get_info": (C.c_int32, [C.c_void_p, C.POINTER(ModelInfo)]),
    "mmdx_model_get_skin": (C.c_int32, [C.c_void_p, _i32p, _i32p, _f32p]),
    "mmdx_model_get_vertex_order": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mmdx_model_slot_weights": (C.c_int32, [C.c_void_p, _f32p, _f32p]),
    "mmdx_model_set_stream": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mmdx_deform": (C.c_int32, [C.c_void_p, _f32p, _f32p, _f32p, _f32p]),
    "mmdx_deform_vertex32": (C.c_int32, [C.c_void_p, _f32p, _f32p, C.c_float, C.c_void_p]),
    "mmdx_deform_batched": (C.c_int32, [C.c_void_p, C.POINTER(DeformArgs)]),
    "mmdx_sync": (C.c_int32, [C.c_void_p]),
    "mmdx_timer_start": (C.c_int32, [C.c_void_p]),
    "mmdx_timer_stop": (C.c_int32, [C.c_void_p, _f32p]),
    "mmdx_profile_enable": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mmdx_profile_collect": (C.c_int32, [C.c_void_p, _u32p, _f32p, _f32p]),
    "mmdx_device_malloc": (C.c_int32, [C.POINTER(C.c_void_p), C.c_size_t]),
    "mmdx_device_free": (C.c_int32, [C.c_void_p]),
    "mmdx_host_malloc": (C.c_int32, [C.POINTER(C.c_void_p), C.c_size_t]),
    "mmdx_host_free": (C.c_int32, [C.c_void_p]),
    "mmdx_memcpy_h2d": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mmdx_memcpy_d2h": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mmdx_device_memset": (C.c_int32, [C.c_void_p, C.c_int, C.c_size_t]),
    "mmdx_device_synchronize": (C.c_int32, []),
    "mmdx_debug_reload_env": (None, []),
    "mmdx_debug_last_store_policy": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "mmdx_debug_morph_pass_stats": (C.c_int32, [C.c_void_p, _u32p, _u32p, _u32p]),
    "mmdx_build_source_sha": (C.c_char_p, []),
    "mmdx_bench_copy": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, _f32p]),
    "mmdx_bench_fill": (C.c_int32, [C.c_void_p, C.c_size_t, C.c_int32, _f32p]),
    "mmdx_bench_store_pattern": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, _f32p]),
    "mmdx_pmx_parse": (C.c_int32, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "mmdx_pmx_load_file": (C.c_int32, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "mmdx_pmd_parse": (C.c_int32, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "mmdx_pmd_load_file": (C.c_int32, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "mmdx_pmx_destroy": (None, [C.c_void_p]),
    "mmdx_pmx_get_info": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mmdx_pmx_get_model_desc": (C.c_int32, [C.c_void_p, C.POINTER(ModelDesc)]),
    "mmdx_pmx_get_arrays": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mmdx_pmx_get_name": (C.c_int32, [C.c_void_p, C.c_int32, C.c_uint32, C.c_char_p, C.c_size_t]),
    "mmdx_crowd_output_alloc": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p), C.c_void_p]),
    "mmdx_vmd_parse": (C.c_int32, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "mmdx_vmd_load_file": (C.c_int32, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "mmdx_vmd_destroy": (None, [C.c_void_p]),
    "mmdx_vmd_get_info": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mmdx_vmd_track_name": (C.c_int32, [C.c_void_p, C.c_int32, C.c_uint32, C.c_char_p, C.c_size_t]),
    "mmdx_vmd_bone_track": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, _u32p]),
    "mmdx_vmd_morph_track": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, _u32p]),
    "mmdx_vmd_bind_morphs": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]),
    "mmdx_morph_motion_get_info": (C.c_int32, [C.c_void_p, _u32p, _u32p, _u32p]),
    "mmdx_morph_motion_eval": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]),
    "mmdx_morph_motion_destroy": (None, [C.c_void_p]),
    "mmdx_vmd_bind_bones": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]),
    "mmdx_bone_motion_get_info": (C.c_int32, [C.c_void_p, _u32p, _u32p, _u32p, _u32p]),
    "mmdx_bone_motion_eval": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]),
    "mmdx_bone_motion_destroy": (None, [C.c_void_p]),
    "mmdx_skeleton_create": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mmdx_skeleton_get_info": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "mmdx_skeleton_solve": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]),
    "mmdx_skeleton_solve_motion": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]),
    "mmdx_skeleton_solve_morphed": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                                C.c_void_p]),
    "mmdx_skeleton_solve_pre": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                            C.c_void_p]),
    "mmdx_skeleton_solve_post": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]),
    "mmdx_skeleton_destroy": (None, [C.c_void_p]),
    "mmdx_graph_begin": (C.c_int32, [C.c_void_p]),
    "mmdx_graph_end": (C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mmdx_graph_launch": (C.c_int32, [C.c_void_p]),
    "mmdx_graph_destroy": (None, [C.c_void_p]),
    "mmdx_pmx_get_skeleton_desc": (C.c_int32, [C.c_void_p, C.c_void_p]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libmmdx.so (once).  Raises if it has not been built: no fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m simple_mmd_renderer_amd.build` "
                "(hipcc, gfx950).  simple_mmd_renderer_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.mmdx_abi_version() != ABI_VERSION:
            raise RuntimeError("libmmdx.so ABI version mismatch")
        _lib = l
    return _lib


def library_source_sha() -> str:
    """The source revision stamped into the loaded library (mmdx_bench.h, mmdx_build_source_sha)."""
    return (lib().mmdx_build_source_sha() or b"").decode()


def check_library_matches_tree() -> dict:
    """The loaded library must have been built from the sources in this tree: returns both hashes, raises on a mismatch
    (MMDX_LIB experiment builds are exempt: tools/ compare variants on purpose)."""
    from . import build
    lib_sha, tree_sha = library_source_sha(), build.source_sha()
    if lib_sha != tree_sha and not os.environ.get("MMDX_LIB"):
        raise RuntimeError(f"libmmdx.so was built from other sources than this tree's (library {lib_sha}, tree {tree_sha}): "
                           "rebuild with `python -m simple_mmd_renderer_amd.build`")
    return {"library_source_sha": lib_sha, "tree_source_sha": tree_sha}


def check(status: int) -> None:
    if status != OK:
        raise MmdxError(status, (lib().mmdx_last_error_string() or b"").decode("utf-8", "replace"))
