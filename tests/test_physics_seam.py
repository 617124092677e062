"""The physics seam of the bone solve (the viewer's frame is PrePhysicsPosing ->
PhysicsReactor::React -> PostPhysicsPosing, main.cpp:1801-1810): between the two bone lists a reactor overwrites
the skinning matrices of the bones its bodies moved (PoserMotionState::Synchronize) and re-derives the local matrix
of the strict ones (Fix; mmd-bullet_impl.inl:34-56, :312-326), and the post-physics bones then hang off those.

CPU: the C restatement (oracle/mmdx_oracle.c: matrix inverse, Fix, the two-list solve with overrides) against the
real libmmd driven the same way (oracle/ref_harness.cpp: libmmd's own Poser, matrices and Inverse).
GPU: mmdx_skeleton_solve_pre / mmdx_skeleton_solve_post against the oracle and a committed golden from libmmd."""
import os

import numpy as np
import pytest

from oracle.pyoracle import Reference, reference_available
from simple_mmd_renderer_amd import synth, vmd
from tests import golden_util as gu
from tests.test_rig import random_poses

GOLDEN = os.path.join(gu.GOLDEN_DIR, "rig_physics_expect.npz")


def physics_case(nb, seed, n_ik=0, n_app=0, post=0.35, k=6):
    """A rig with post-physics bones and a set of 'physics' bones: random pre-physics bones (some with post-physics
    descendants), half of them strict, parents listed before and after their children; transforms = rigid motions
    with a little shear (Bullet only produces rigid ones; Fix must not depend on that)."""
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(nb, seed, n_ik=n_ik, n_append=n_app, post_physics=post)
    rng = np.random.RandomState(1000 + seed)
    pre = [b for b in range(nb) if not (flags[b] & 0x1000)]
    over = [int(b) for b in rng.choice(pre, min(k, len(pre)), replace=False)]
    # a strict child listed BEFORE its strict parent, and one after: Fix reads the parent's local matrix as it stands
    kids = [b for b in pre if parent[b] in over and b not in over]
    over = kids[:1] + over + kids[1:2]
    strict = np.asarray([k % 3 != 1 for k in range(len(over))], np.uint8)       # Synchronize-only bones in between
    return (rest, parent, level, flags, ap, ar, ik), np.asarray(over, np.int64), strict, rng


def random_transforms(rng, ni, k):
    out = np.zeros((ni, k, 16), np.float32)
    for i in range(ni):
        for j in range(k):
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            x, y, z, w = q
            r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)],
                          [2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)],
                          [2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)]])
            m = np.eye(4)
            m[:3, :3] = r + (rng.normal(size=(3, 3)) * 0.02 if (i + j) % 3 == 0 else 0)
            m[3, :3] = rng.uniform(-3, 3, 3)
            out[i, j] = m.astype(np.float32).reshape(16)
    return out


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
def test_matrix_inverse_vs_reference(oracle):
    rng = np.random.RandomState(5)
    for t in range(4000):
        m = rng.normal(size=16).astype(np.float32)
        if t % 7 == 0:
            m[rng.randint(16)] = 0
        if t % 11 == 0:
            m[4:8] = m[0:4] * 2                     # singular: the reference returns the zero matrix or garbage alike
        if t % 13 == 0:
            m[8:12] = 0                             # zero row
        if t % 5 == 0:
            m = random_transforms(rng, 1, 1)[0, 0]
        a, b = oracle.matrix_inverse(m), Reference.matrix_inverse(m)
        gu.assert_bits_equal_or_both_nan(a, b, f"inverse {t}")


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
@pytest.mark.parametrize("nb,seed,n_ik,n_app", [(12, 0, 0, 0), (40, 1, 0, 0), (60, 2, 3, 4), (150, 3, 6, 8)])
def test_oracle_physics_seam_vs_reference(oracle, nb, seed, n_ik, n_app):
    rig, over, strict, rng = physics_case(nb, seed, n_ik, n_app)
    rest, parent, level, flags, ap, ar, ik = rig
    ref = Reference.skeleton(rest, parent, level, flags, ap, ar, ik)
    poses = random_poses(4, nb, 300 + seed)
    xf = random_transforms(rng, 4, over.size)
    changed = 0
    for i in range(4):
        got, got_pre = oracle.bone_solve_physics(rest, parent, poses[i], over, strict, xf[i], level, flags, ap, ar, ik)
        want, want_pre = ref.solve_physics(poses[i], over, strict, xf[i])
        pre_rows = [b for b in range(nb) if not (flags[b] & 0x1000)]
        gu.assert_bits_equal(got_pre[pre_rows], want_pre[pre_rows], f"palette after the pre-physics list, {i}")
        gu.assert_bits_equal(got, want, f"palette {i}")
        plain = oracle.bone_solve_full(rest, parent, poses[i], level, flags, ap, ar, ik)
        changed += int(np.any(gu.bits(got) != gu.bits(plain)))
        # no overrides == the plain solve
        none, _ = oracle.bone_solve_physics(rest, parent, poses[i], [], [], [], level, flags, ap, ar, ik)
        gu.assert_bits_equal(none, plain, "empty override list")
    ref.close()
    assert changed == 4


def test_golden_physics_seam_oracle(oracle):
    z = np.load(GOLDEN, allow_pickle=False)
    rig, over, strict, _ = physics_case(int(z["nb"]), int(z["seed"]), int(z["n_ik"]), int(z["n_app"]))
    rest, parent, level, flags, ap, ar, ik = rig
    assert np.array_equal(over, z["over"]) and np.array_equal(strict, z["strict"])
    for i in range(z["poses"].shape[0]):
        got, pre = oracle.bone_solve_physics(rest, parent, z["poses"][i], over, strict, z["xf"][i], level, flags, ap, ar, ik)
        gu.assert_bits_equal(got, z["expect"][i], f"palette {i}")


# ---- GPU: mmdx_skeleton_solve_pre / _post ------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_golden_physics_seam():
    """libmmd's own answers (committed fixture) through the two-step device solve."""
    z = np.load(GOLDEN, allow_pickle=False)
    rig, over, strict, _ = physics_case(int(z["nb"]), int(z["seed"]), int(z["n_ik"]), int(z["n_app"]))
    sk = vmd.Skeleton(*rig, physics_seam=True)
    pre = sk.solve_pre(z["poses"])
    pre_rows = [b for b in range(int(z["nb"])) if not (rig[3][b] & 0x1000)]
    gu.assert_bits_equal(pre[:, pre_rows], z["expect_pre"][:, pre_rows], "palettes after the pre-physics list")
    got = sk.solve_post(over, strict, z["xf"])
    gu.assert_bits_equal(got, z["expect"], "palettes")
    sk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nb,seed,n_ik,n_app", [(12, 0, 0, 0), (40, 1, 0, 0), (60, 2, 3, 4), (150, 3, 6, 8), (300, 4, 8, 12)])
def test_gpu_physics_seam_vs_oracle(oracle, nb, seed, n_ik, n_app):
    """70 instances (the last workgroup partly empty), FK-only rigs included: pre + overrides + post == the oracle;
    no overrides == the one-call solve; the one-call solve of a seam skeleton == that of a plain one."""
    rig, over, strict, rng = physics_case(nb, seed, n_ik, n_app)
    rest, parent, level, flags, ap, ar, ik = rig
    ni = 70
    poses = random_poses(ni, nb, 500 + seed)
    xf = random_transforms(rng, ni, over.size)
    sk = vmd.Skeleton(*rig, physics_seam=True)
    assert sk.info["solver"] == vmd.SOLVER_SERIAL
    sk.solve_pre(poses)
    got = sk.solve_post(over, strict, xf)
    for i in range(ni):
        want, _ = oracle.bone_solve_physics(rest, parent, poses[i], over, strict, xf[i], level, flags, ap, ar, ik)
        gu.assert_bits_equal(got[i], want, f"palette of instance {i}")
    whole = sk.solve(poses)
    sk.solve_pre(poses)
    gu.assert_bits_equal(sk.solve_post([], [], np.zeros((ni, 0, 16), np.float32)), whole, "empty override list")
    plain = vmd.Skeleton(*rig)
    gu.assert_bits_equal(plain.solve(poses), whole, "seam skeleton vs plain skeleton, one-call solve")
    plain.close()
    sk.close()


@pytest.mark.gpu
def test_gpu_physics_seam_argument_checks():
    rig, over, strict, rng = physics_case(20, 7)
    poses = random_poses(3, 20, 1)
    plain = vmd.Skeleton(*rig)
    with pytest.raises(Exception, match="PHYSICS_SEAM"):
        plain.solve_pre(poses)
    plain.close()
    sk = vmd.Skeleton(*rig, physics_seam=True)
    with pytest.raises(Exception, match="without a matching"):
        sk.solve_post(over, strict, random_transforms(rng, 3, over.size))
    sk.solve_pre(poses)
    with pytest.raises(Exception, match="out of range"):
        sk.solve_post([99], [1], random_transforms(rng, 3, 1))
    sk.close()
