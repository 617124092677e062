"""PMD 1.0 loader (csrc/pmd.cpp) against libmmd's PmdReader on the same bytes: the vertices deform
bit-identically and the converted rig (bone types -> flags / append / IK tables, extra IK bones, transform
levels) solves bit-identically.  Fixtures come from simple_mmd_renderer_amd.pmd.write_pmd (the reference
ships no asset); a committed golden file covers the GPU box."""
import os

import numpy as np
import pytest

from oracle.pyoracle import Reference, reference_available
from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import pmd, pmx, synth, vmd
from simple_mmd_renderer_amd.engine import DeformModel
from tests import golden_util as gu

rigged_pmd = pmd.make_rigged_pmd


@pytest.fixture(autouse=True)
def _lib(hip_lib):
    return hip_lib


def test_pmd_parse_structure():
    data, m, opt = rigged_pmd(1)
    pm = pmx.load_pmd(data)
    nb = m.nb + (len(opt.iks) - 1)                        # the second IK record of the bone appends a bone
    assert (pm.info["n_vertices"], pm.info["n_bones"], pm.info["n_morphs"]) == (m.nv, nb, m.nm + 1)
    assert np.all(pm.flat.skin_type == 1)
    assert np.array_equal(pm.flat.positions, m.positions) and np.array_equal(pm.flat.bone_ids[:, :2], m.bone_ids[:, :2])
    want_w = (np.clip(np.rint(m.bone_weights[:, 0].astype(np.float64) * 100), 0, 100).astype(np.int32).astype(np.float32)
              * np.float32(0.01))
    assert np.array_equal(pm.flat.bone_weights[:, 0], want_w)
    ikb = m.nb - 1
    assert pm.bone_flags[ikb] & 0x20 and pm.bone_transform_level[ikb] == 1
    assert pm.bone_flags[m.nb - 2] & 0x100 and pm.append_parent[m.nb - 2] == 3 and pm.append_ratio[m.nb - 2] == 1.0
    assert pm.bone_flags[m.nb - 3] & 0x100 and pm.append_parent[m.nb - 3] == 2 and pm.append_ratio[m.nb - 3] == np.float32(50 * 0.01)
    assert pm.bone_flags[m.nb - 5] & 0x20 and pm.ik["target"][m.nb - 5] == 0       # IK type, no record
    if len(opt.iks) == 2:
        assert pm.bone_names[-1] == "[IK]" + pm.bone_names[ikb] and pm.flat.bone_parent[-1] == ikb
        assert pm.ik["target"][-1] == opt.iks[1].target and pm.ik["angle"][-1] == np.float32(0.5) * 4
    assert pm.morph_names[1:] == ["あ", "い", "う", "まばたき"]
    # morph indices are resolved through the base morph
    for k in range(m.nm):
        lo, hi = int(m.morph_off[k]), int(m.morph_off[k + 1])
        plo = int(pm.flat.morph_off[k + 1])
        assert np.array_equal(pm.flat.morph_index[plo:plo + hi - lo], m.morph_index[lo:hi])
    assert pm.skeleton().info["solver"] == vmd.SOLVER_SERIAL


def test_pmd_knee_limit_by_decoded_name():
    data, m, _ = rigged_pmd(2, knee=True)
    pm = pmx.load_pmd(data)
    lim = pm.ik["link_limited"].astype(bool)
    assert lim.sum() == 1 and pm.bone_names[int(pm.ik["link_bone"][lim][0])] == "左ひざ"
    assert pm.ik["link_lo"][lim][0, 0] == -np.float32(np.pi) and pm.ik["link_hi"][lim][0, 0] == np.float32(-0.5) / np.float32(180.0) * np.float32(np.pi)


def test_malformed_pmd():
    data, _, _ = rigged_pmd(3)
    import ctypes as C
    lib = api.lib()
    for bad in (b"", b"Pmx" + data[3:], data[:100], data[:283 + 4 + 38 * 10], data[:-60]):
        h = C.c_void_p()
        buf = (C.c_char * max(len(bad), 1)).from_buffer_copy(bad or b"\0")
        assert lib.mmdx_pmd_parse(buf, len(bad), C.byref(h)) == 1 and not h.value
    # a morph entry that points behind the base morph's list
    raw = bytearray(data)
    pm = pmx.load_pmd(data)
    off = pm.info["bytes_consumed"] - 16                   # last morph entry
    raw[off:off + 4] = (10 ** 6).to_bytes(4, "little")
    with pytest.raises(api.MmdxError) as e:
        pmx.load_pmd(bytes(raw))
    assert "base morph" in str(e.value)


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
@pytest.mark.parametrize("seed,extended,base", [(0, False, True), (1, True, True), (2, False, False), (3, True, True)])
def test_pmd_through_both_readers(oracle, tmp_path, seed, extended, base):
    data, _, _ = rigged_pmd(seed, extended=extended, base_morph=base)
    path = tmp_path / "m.pmd"
    path.write_bytes(data)
    pm = pmx.load_pmd(str(path))
    ref = Reference.from_pmd(str(path))
    assert (ref.model.nv, ref.model.nb, ref.model.nm) == (pm.flat.nv, pm.flat.nb, pm.flat.nm)
    t, ids, w = ref.get_skin()                             # after libmmd's Normalize
    nt, nids, nw = oracle.normalize(pm.flat)
    assert np.array_equal(t, nt) and np.array_equal(w, nw)
    rng = np.random.RandomState(seed)
    for f in range(3):
        rates = rng.choice([0, 0.3, 1.0, -0.5], pm.flat.nm).astype(np.float32)
        pal = synth.make_palettes(pm.flat, [f * 7])[0]
        rp, rn, _ = ref.run(rates, pal)
        op, on = oracle.deform(pm.flat, rates, pal)
        gu.assert_bits_equal(op, rp, "positions")
        gu.assert_bits_equal(on, rn, "normals")
    nb = pm.flat.nb
    poses = np.zeros((nb, 8), np.float32)
    poses[:, 0:3] = rng.uniform(-1, 1, (nb, 3))
    q = rng.normal(size=(nb, 4))
    poses[:, 4:8] = q / np.linalg.norm(q, axis=1, keepdims=True)
    for b in range(nb):
        ref.set_bone_pose(b, poses[b, 0:3], poses[b, 4:8])
    ref.set_morphs(np.zeros(pm.flat.nm, np.float32))
    ref.pose()
    got = oracle.bone_solve_full(pm.flat.bone_pos, pm.flat.bone_parent, poses, pm.bone_transform_level, pm.bone_flags,
                                 pm.append_parent, pm.append_ratio, pm.ik)
    gu.assert_bits_equal(got, ref.get_palette(), "palette of the converted rig")
    ref.close()


def test_golden_pmd_oracle(oracle):
    """tests/golden/pmd_small.pmd + what libmmd's PmdReader + Poser made of it."""
    pm = pmx.load_pmd(os.path.join(gu.GOLDEN_DIR, "pmd_small.pmd"))
    z = np.load(os.path.join(gu.GOLDEN_DIR, "pmd_small_expect.npz"))
    for f in range(z["rates"].shape[0]):
        op, on = oracle.deform(pm.flat, z["rates"][f], z["palette"][f])
        gu.assert_bits_equal(op, z["expect_pos"][f], "positions")
        gu.assert_bits_equal(on, z["expect_nrm"][f], "normals")
        got = oracle.bone_solve_full(pm.flat.bone_pos, pm.flat.bone_parent, z["poses"][f], pm.bone_transform_level,
                                     pm.bone_flags, pm.append_parent, pm.append_ratio, pm.ik)
        gu.assert_bits_equal(got, z["expect_rig_palette"][f], "rig palette")


@pytest.mark.gpu
def test_gpu_pmd_file_end_to_end():
    """.pmd -> this loader -> GPU model + device bone solve, against libmmd's recorded answers."""
    pm = pmx.load_pmd(os.path.join(gu.GOLDEN_DIR, "pmd_small.pmd"))
    z = np.load(os.path.join(gu.GOLDEN_DIR, "pmd_small_expect.npz"))
    with DeformModel(pm.flat) as dm:
        pos, nrm = dm.deform_batched(z["rates"], z["palette"])
        gu.assert_bits_equal(pos, z["expect_pos"], "positions")
        gu.assert_bits_equal(nrm, z["expect_nrm"], "normals")
    gu.assert_bits_equal(pm.skeleton().solve(z["poses"]), z["expect_rig_palette"], "rig palettes")
