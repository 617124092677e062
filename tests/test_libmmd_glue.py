"""The reference-side binding (simple_mmd_renderer_amd/host/libmmd_glue.hpp: mmd::Model -> mmdx_model_desc, the palette tap
behind PhysicsReactor::GetPoserBoneImage, the morph-rate mirror of MotionPlayer::SeekFrame) as
COMPILED code (task 5).

Build container (needs /root/reference): tests/glue_driver.cpp is compiled with g++ against the real libmmd and libmmdx.so and
run without a GPU.  GPU box: the same header, compiled into oracle/_ref/libmmd_ref.so, turns libmmd's own PmxReader model into
the descriptor mmdx_model_create gets, and the palette read through the tap feeds mmdx_deform_vertex32 -- against libmmd's
golden vertices for tests/golden/pmx_small.pmx."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.pyoracle import Reference, reference_available
from simple_mmd_renderer_amd import _capi as api
from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/3rd_party/libmmd/include"
PMX = os.path.join(gu.GOLDEN_DIR, "pmx_small.pmx")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF_INC, "mmd")), reason="needs /root/reference (build container)")
def test_glue_header_compiles_against_libmmd_and_agrees_with_the_loader(hip_lib, tmp_path):
    from simple_mmd_renderer_amd import pmx, synth, vmd
    # the model of part 2: names written WITH the byte-order mark libmmd's VMD reader leaves in front of converted names on Linux
    m = synth.make_model(600, 24, 5, 80, 113)
    morph_names, bone_names = [f"モーフ{i}" for i in range(m.nm)], [f"ボーン{i}" for i in range(m.nb)]
    bom = "\ufeff"
    (tmp_path / "bom.pmx").write_bytes(pmx.write_pmx(m, pmx.PmxWriteOptions(morph_names=[bom + n for n in morph_names],
                                                                            bone_names=[bom + n for n in bone_names],
                                                                            bone_flag_variety=False)))
    rng = np.random.RandomState(9)
    keys = [(n, int(f), float(np.float32(rng.uniform(0.05, 1.0)))) for n in morph_names for f in sorted(rng.choice(120, 4, replace=False))]
    bone_keys = [(bone_names[1], 0, (0, 0, 0), (0, 0, 0, 1), None), (bone_names[1], 90, (0.5, 1, 0), (0, 0.3827, 0, 0.9239), None)]
    vpath = tmp_path / "glue.vmd"
    vpath.write_bytes(vmd.write_vmd(bone_keys, keys))
    exe = tmp_path / "glue_driver"
    lib_dir = os.path.join(ROOT, "simple_mmd_renderer_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-parameter", os.path.join(ROOT, "tests", "glue_driver.cpp"), "-I" + REF_INC,
           "-L" + lib_dir, "-lmmdx", "-Wl,-rpath," + lib_dir, "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    # libmmd's own headers warn (unused variables, sign compares); only OUR header must be clean
    ours = [ln for ln in r.stderr.splitlines() if "libmmd_glue.hpp" in ln and "warning" in ln]
    assert r.returncode == 0, r.stderr[-3000:]
    assert not ours, "\n".join(ours)
    r = subprocess.run([str(exe), PMX, str(tmp_path / "bom.pmx"), str(vpath)], capture_output=True, text=True)
    assert r.returncode == 0 and "GLUE OK" in r.stdout, r.stdout + r.stderr


def _glue_model(ref, hip_lib, flags):
    """mmd::Model of `ref` -> glue::Flatten (inside libmmd_ref.so) -> mmdx_model_create; returns (handle, desc)."""
    ref.lib.mmdref_glue_flatten.restype = C.c_void_p
    ref.lib.mmdref_glue_flatten.argtypes = [C.c_void_p, C.POINTER(api.ModelDesc), C.c_uint32]
    ref.lib.mmdref_glue_free.argtypes = [C.c_void_p]
    ref.lib.mmdref_glue_read_palette.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    desc = api.ModelDesc()
    flat = ref.lib.mmdref_glue_flatten(ref.h, C.byref(desc), flags)
    assert flat
    h = C.c_void_p()
    try:
        api.check(hip_lib.mmdx_model_create(C.byref(desc), C.byref(h)))
    finally:
        ref.lib.mmdref_glue_free(flat)
    return h, desc


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
def test_glue_through_the_reference_harness_host_only(hip_lib):
    """The call sequence of the GPU test below, without a GPU: the flattened model validates and compiles (host-only) to the same
    skin tags as the bundled loader's, and the tap reads back an injected palette."""
    from simple_mmd_renderer_amd import pmx
    from simple_mmd_renderer_amd.engine import DeformModel
    ref = Reference.from_pmx(PMX)
    h, desc = _glue_model(ref, hip_lib, api.CREATE_HOST_ONLY)         # PmxReader already ran Normalize: no MMDX_CREATE_NORMALIZE
    nv, nb = desc.n_vertices, desc.n_bones
    t, ids, w = np.empty(nv, np.int32), np.empty((nv, 4), np.int32), np.empty((nv, 4), np.float32)
    api.check(hip_lib.mmdx_model_get_skin(h, t.ctypes.data_as(C.POINTER(C.c_int32)), ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                          w.ctypes.data_as(C.POINTER(C.c_float))))
    hip_lib.mmdx_model_destroy(h)
    with DeformModel(pmx.load_pmx(PMX).flat, normalize=True, host_only=True) as dm:
        t2, ids2, w2 = dm.get_skin()
    assert np.array_equal(t, t2) and np.array_equal(ids, ids2) and np.array_equal(w.view(np.uint32), w2.view(np.uint32))
    z = np.load(os.path.join(gu.GOLDEN_DIR, "pmx_small_expect.npz"))
    ref.set_palette(z["palette"][1])
    pal = np.zeros((nb, 16), np.float32)
    ref.lib.mmdref_glue_read_palette(ref.h, pal.ctypes.data_as(C.POINTER(C.c_float)))
    gu.assert_bits_equal(pal, z["palette"][1].reshape(nb, 16), "the tap reads what the reactor-side door wrote")
    ref.close()


@pytest.mark.gpu
@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
def test_gpu_glue_to_deform_vertex32_vs_libmmd_golden(hip_lib):
    """libmmd's PmxReader model -> glue::Flatten (inside libmmd_ref.so) -> mmdx_model_create -> per frame: palette injected into
    libmmd's Poser, read back through glue::PaletteTap -> mmdx_deform_vertex32 -> libmmd's golden vertices."""
    from simple_mmd_renderer_amd.engine import device_count
    assert device_count() >= 1
    z = np.load(os.path.join(gu.GOLDEN_DIR, "pmx_small_expect.npz"))
    ref = Reference.from_pmx(PMX)
    h, desc = _glue_model(ref, hip_lib, 0)
    nv, nb = desc.n_vertices, desc.n_bones
    for f in range(z["rates"].shape[0]):
        ref.set_palette(z["palette"][f])
        pal = np.zeros((nb, 16), np.float32)
        ref.lib.mmdref_glue_read_palette(ref.h, pal.ctypes.data_as(C.POINTER(C.c_float)))
        gu.assert_bits_equal(pal, z["palette"][f].reshape(nb, 16), "the tap reads what the reactor-side door wrote")
        out = np.zeros((nv, 8), np.float32)
        rates = np.ascontiguousarray(z["rates"][f], np.float32)
        api.check(hip_lib.mmdx_deform_vertex32(h, rates.ctypes.data_as(C.POINTER(C.c_float)), pal.ctypes.data_as(C.POINTER(C.c_float)),
                                               C.c_float(0.1), out.ctypes.data))
        gu.assert_bits_equal(out, z["expect_v32"][f], f"frame {f}: glue -> mmdx_deform_vertex32 vs libmmd Deform + repack")
    hip_lib.mmdx_model_destroy(h)
    ref.close()
