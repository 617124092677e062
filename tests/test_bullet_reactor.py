"""The physics seam pinned against the reference's REAL physics reactor (task 2).

tests/golden/rig_bullet_expect.npz holds 96 frames of the viewer's loop
    ResetPosing -> SetBonePose -> PrePhysicsPosing -> BulletPhysicsReactor::React(1/30) -> PostPhysicsPosing -> Deform
(main.cpp:1786-1821) run by libmmd's own mmd::BulletPhysicsReactor over the vendored Bullet -- compiled from the sources
where they lie under /root/reference by oracle/Makefile, driven by oracle/ref_bullet_harness.cpp, written by
oracle/gen_golden_bullet.py.  Per frame: local poses, the palette after PrePhysicsPosing, every body's transform as
PoserMotionState::Synchronize turns it into a skinning matrix (mmd-bullet_impl.inl:34-40), the final palette after Fix
(:42-56) + PostPhysicsPosing, and pose_image (checksums; every 8th frame in full).

CPU: the C restatement (oracle/mmdx_oracle.c: physics_fix, the two-list solve) reproduces every frame bit for bit from the
poses and the bodies' transforms; in the build container the fixture is regenerated from the real reactor and compared.
GPU: mmdx_skeleton_solve_pre / _post and the deform kernels reproduce palettes and vertices bit for bit."""
import os

import numpy as np
import pytest

from oracle.pyoracle import BulletReference, bullet_reference_available
from simple_mmd_renderer_amd import synth
from simple_mmd_renderer_amd.synth import FlatModel
from tests import golden_util as gu

GOLDEN = os.path.join(gu.GOLDEN_DIR, "rig_bullet_expect.npz")


def load():
    z = np.load(GOLDEN, allow_pickle=False)
    rig = {k[4:]: z[k] for k in z.files if k.startswith("rig_")}
    mesh = {k[5:]: z[k] for k in z.files if k.startswith("mesh_")}
    bodies = {k[5:]: z[k] for k in z.files if k.startswith("body_") and k != "body_xf"}
    joints = {k[6:]: z[k] for k in z.files if k.startswith("joint_")}
    # what React writes: Synchronize for every body that is neither kinematic nor a ghost, in body order; Fix for the strict ones
    moved = ~(z["passive"].astype(bool) | z["ghost"].astype(bool))
    over = bodies["bone"][moved].astype(np.int64)
    strict = z["strict"][moved].astype(np.uint8)
    nb = rig["rest"].shape[0]
    nv = mesh["positions"].shape[0]
    model = FlatModel(mesh["positions"], mesh["normals"], np.zeros((nv, 2), np.float32), mesh["skin_type"], mesh["bone_ids"],
                      mesh["bone_weights"], rig["rest"], rig["parent"].astype(np.int32), np.zeros(0, np.int32),
                      np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros((0, 3), np.float32))
    assert model.nb == nb
    return z, rig, mesh, bodies, joints, moved, over, strict, model


def test_fixture_shape_and_content():
    z, rig, mesh, bodies, joints, moved, over, strict, model = load()
    nf = z["poses"].shape[0]
    assert nf >= 60 and z["palette"].shape == (nf, model.nb, 16) and z["body_xf"].shape == (nf, bodies["bone"].shape[0], 16)
    # every kind of body is in it, strict ones with parents listed before AND after them, two bodies on one bone, one on a
    # post-physics bone; and physics really moved the bones (the final palette is not the pre-physics one)
    assert z["passive"].sum() >= 3 and z["ghost"].sum() >= 1 and strict.sum() >= 4 and (strict == 0).sum() >= 3
    assert len(set(over.tolist())) < over.size
    assert (rig["flags"][over] & 0x1000).any()
    assert np.abs(z["palette"] - z["palette_pre"])[:, over].max() > 1.0
    # Synchronize writes rigid transforms: getOpenGLMatrix's last column
    assert np.all(z["body_xf"][:, :, [3, 7, 11]] == 0) and np.all(z["body_xf"][:, :, 15] == 1)


@pytest.mark.skipif(not bullet_reference_available(), reason="oracle/_ref/libmmd_bullet_ref.so not built (needs /root/reference)")
def test_fixture_is_the_real_reactors_output():
    """Build container only: run mmd::BulletPhysicsReactor again and compare with the committed fixture, bit for bit."""
    z, rig, mesh, bodies, joints, *_ = load()
    ref = BulletReference(rig, mesh, bodies, joints)
    passive, strict, ghost = ref.body_info()
    assert np.array_equal(passive, z["passive"]) and np.array_equal(strict, z["strict"]) and np.array_equal(ghost, z["ghost"])
    full = {int(f): k for k, f in enumerate(z["full_frames"])}
    for f in range(z["poses"].shape[0]):
        o = ref.frame(z["poses"][f])
        for k in ("palette_pre", "body_xf", "palette"):
            gu.assert_bits_equal(o[k], z[k][f], f"frame {f} {k}")
        assert (synth.checksum64(o["pos"]), synth.checksum64(o["nrm"])) == tuple(int(x) for x in z["vertex_sums"][f])
        if f in full:
            gu.assert_bits_equal(o["pos"], z["full_pos"][full[f]], f"frame {f} pos")
    ref.close()


def test_oracle_restatement_reproduces_the_reactors_frames(oracle):
    z, rig, mesh, bodies, joints, moved, over, strict, model = load()
    pre_rows = np.flatnonzero((rig["flags"] & 0x1000) == 0)
    full = {int(f): k for k, f in enumerate(z["full_frames"])}
    for f in range(z["poses"].shape[0]):
        got, pre = oracle.bone_solve_physics(rig["rest"], rig["parent"], z["poses"][f], over, strict, z["body_xf"][f][moved],
                                             rig["level"], rig["flags"])
        gu.assert_bits_equal(pre[pre_rows], z["palette_pre"][f][pre_rows], f"frame {f}: palette after the pre-physics list")
        gu.assert_bits_equal(got, z["palette"][f], f"frame {f}: final palette")
        pos, nrm = oracle.skin(model, got)
        assert (synth.checksum64(pos), synth.checksum64(nrm)) == tuple(int(x) for x in z["vertex_sums"][f]), f"frame {f} vertices"
        if f in full:
            gu.assert_bits_equal(pos, z["full_pos"][full[f]], f"frame {f} pos")
            gu.assert_bits_equal(nrm, z["full_nrm"][full[f]], f"frame {f} nrm")


@pytest.mark.gpu
def test_gpu_physics_seam_reproduces_the_reactors_frames(hip_lib):
    """All frames as one batch (a frame depends on the others only through the bodies' transforms, which are inputs here):
    solve_pre -> palettes of the pre-physics bones; solve_post with the reactor's writes -> final palettes; deform -> vertices."""
    from simple_mmd_renderer_amd import vmd
    from simple_mmd_renderer_amd.engine import DeformModel
    z, rig, mesh, bodies, joints, moved, over, strict, model = load()
    nf = z["poses"].shape[0]
    sk = vmd.Skeleton(rig["rest"], rig["parent"].astype(np.int32), rig["level"], rig["flags"], physics_seam=True)
    pre = sk.solve_pre(z["poses"])
    pre_rows = np.flatnonzero((rig["flags"] & 0x1000) == 0)
    gu.assert_bits_equal(pre[:, pre_rows], z["palette_pre"][:, pre_rows], "palettes after the pre-physics list")
    pal = sk.solve_post(over, strict, np.ascontiguousarray(z["body_xf"][:, moved]))
    gu.assert_bits_equal(pal, z["palette"], "final palettes")
    sk.close()
    with DeformModel(model) as dm:
        pos, nrm = dm.deform_batched(np.zeros((nf, 0), np.float32), pal)
        v32 = dm.deform_batched(np.zeros((nf, 0), np.float32), pal, layout=1, pos_scale=0.1)
    for f in range(nf):
        assert (synth.checksum64(pos[f]), synth.checksum64(nrm[f])) == tuple(int(x) for x in z["vertex_sums"][f]), f"frame {f}"
    for k, f in enumerate(z["full_frames"]):
        gu.assert_bits_equal(pos[f], z["full_pos"][k], f"frame {f} pos")
        gu.assert_bits_equal(nrm[f], z["full_nrm"][k], f"frame {f} nrm")
        assert np.array_equal(v32[f][:, :3].view(np.uint32), (z["full_pos"][k] * np.float32(0.1)).view(np.uint32))
