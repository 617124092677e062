#!/usr/bin/env python3
"""Generates tests/golden/rig_bullet_expect.npz by EXECUTING the reference's real physics reactor
(mmd::BulletPhysicsReactor over the vendored Bullet, oracle/ref_bullet_harness.cpp -> oracle/_ref/libmmd_bullet_ref.so).
Build container only (needs /root/reference); the fixture is data -- inputs and the reference's outputs -- and travels.

    python oracle/gen_golden_bullet.py

The model (all synthetic, seeded): a 3-bone kinematic spine, three 4-link chains of dynamic bodies hanging off it (plain /
strict / ghost bodies, capsules / spheres / boxes, one bone with TWO bodies), 6-DOF spring joints between consecutive bodies,
post-physics bones hanging off the chain bones, 640 vertices skinned BDEF1/2/4 to all of it.  FRAMES frames of the viewer's
loop ResetPosing -> SetBonePose -> PrePhysicsPosing -> React(1/30) -> PostPhysicsPosing -> Deform (main.cpp:1786-1821) with the
spine swinging.  Per frame the fixture keeps: the local poses, the palette after PrePhysicsPosing, every body's transform as
PoserMotionState::Synchronize turns it into a skinning matrix (mmd-bullet_impl.inl:34-40), the final palette and a checksum of
pose_image; full pose_image for every 8th frame."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import BulletReference  # noqa: E402
from simple_mmd_renderer_amd import synth  # noqa: E402

FRAMES = 96
OUT = os.path.join(ROOT, "tests", "golden", "rig_bullet_expect.npz")


def make_case(seed=77):
    rng = np.random.RandomState(seed)
    rest, parent, flags = [], [], []

    def bone(p, par, post=False):
        rest.append(p); parent.append(par); flags.append(0x1000 if post else 0)
        return len(rest) - 1
    b0 = bone((0.0, 10.0, 0.0), -1)
    b1 = bone((0.0, 14.0, 0.2), b0)
    b2 = bone((0.0, 17.0, 0.0), b1)
    chains = []
    for c, (anchor, x) in enumerate(((b2, -1.2), (b2, 1.3), (b1, 0.1))):
        prev, links = anchor, []
        for j in range(4):
            prev = bone((x + 0.15 * j, rest[anchor][1] - 1.0 - 1.4 * j, -0.8 - 0.1 * c), prev)
            links.append(prev)
        chains.append(links)
    posts = [bone((rest[chains[0][1]][0] - 0.5, rest[chains[0][1]][1] - 0.3, -1.0), chains[0][1], post=True),
             bone((rest[chains[1][3]][0] + 0.4, rest[chains[1][3]][1] - 0.6, -0.9), chains[1][3], post=True),
             bone((0.3, 15.0, 0.9), b1, post=True)]
    posts.append(bone((rest[posts[0]][0] - 0.4, rest[posts[0]][1] - 0.5, -1.1), posts[0], post=True))   # post-physics child of one
    extra = [bone((1.5, 12.0, 0.5), b0), bone((-1.5, 12.5, 0.4), b1)]
    nb = len(rest)
    rig = dict(rest=np.asarray(rest, np.float32), parent=np.asarray(parent, np.int64), level=np.zeros(nb, np.int32),
               flags=np.asarray(flags, np.uint16))
    rig["level"][posts[3]] = 1

    B = {k: [] for k in ("bone", "group", "mask", "shape", "dims", "pos", "rot", "mass", "tdamp", "rdamp", "restitution", "friction",
                         "type")}

    def body(bn, typ, shape, dims, group, off=(0, 0, 0), rot=(0, 0, 0), mass=1.0):
        B["bone"].append(bn); B["type"].append(typ); B["shape"].append(shape); B["dims"].append(dims)
        B["group"].append(group); B["mask"].append(0xFFFF & ~(1 << group))
        B["pos"].append(tuple(np.asarray(rest[bn]) + np.asarray(off))); B["rot"].append(rot)
        B["mass"].append(mass); B["tdamp"].append(0.5 + 0.1 * (len(B["bone"]) % 4)); B["rdamp"].append(0.6 + 0.1 * (len(B["bone"]) % 3))
        B["restitution"].append(0.1); B["friction"].append(0.5)
        return len(B["bone"]) - 1
    k0 = body(b0, 0, 1, (1.2, 1.5, 0.8), 0)                       # kinematic: box, sphere, capsule
    k1 = body(b1, 0, 0, (1.0, 0.0, 0.0), 0, off=(0, 0.5, 0))
    k2 = body(b2, 0, 2, (0.7, 1.0, 0.0), 0, rot=(0.2, 0.1, -0.3))
    J = {k: [] for k in ("body", "pos", "rot", "pos_lo", "pos_hi", "rot_lo", "rot_hi", "spring_t", "spring_r")}
    types = ((1, 2, 1, 2), (2, 2, 3, 1), (1, 3, 2, 2))                # plain / strict / ghost per chain link
    for c, links in enumerate(chains):
        prev_body = k2 if parent[links[0]] == b2 else k1
        for j, bn in enumerate(links):
            shape = (2, 0, 1)[(c + j) % 3]
            dims = ((0.25, 0.9, 0.0), (0.35, 0.0, 0.0), (0.3, 0.5, 0.25))[(c + j) % 3]
            rb = body(bn, types[c][j], shape, dims, c + 1, off=(0.02 * j, -0.6, 0.0), rot=(0.1 * c, -0.05 * j, 0.07),
                      mass=0.5 + 0.25 * j)
            J["body"].append((prev_body, rb)); J["pos"].append(rest[bn]); J["rot"].append((0.0, 0.1 * c, 0.05 * j))
            J["pos_lo"].append((0, 0, 0) if j % 2 == 0 else (-0.05, -0.05, -0.05)); J["pos_hi"].append((0, 0, 0) if j % 2 == 0 else (0.05, 0.05, 0.05))
            J["rot_lo"].append((-0.6, -0.3, -0.4)); J["rot_hi"].append((0.6, 0.3, 0.4))
            J["spring_t"].append((0, 0, 0) if c != 1 else (10.0, 10.0, 10.0)); J["spring_r"].append((20.0 * (j % 2), 5.0, 12.0))
            prev_body = rb
    # a second body on a chain bone that already has one (listed later: its Synchronize wins), and a body on a post-physics bone
    body(chains[2][2], 1, 0, (0.2, 0, 0), 4, off=(0.1, -0.2, 0.1), mass=0.3)
    body(posts[2], 1, 0, (0.25, 0, 0), 5, off=(0.0, -0.1, 0.0), mass=0.4)
    bodies = {k: np.asarray(v) for k, v in B.items()}
    joints = {k: np.asarray(v) for k, v in J.items()}

    nv = 640
    pos = np.empty((nv, 3), np.float32)
    ids = np.zeros((nv, 4), np.int64)
    wts = np.zeros((nv, 4), np.float32)
    st = np.empty(nv, np.int32)
    for v in range(nv):
        bn = int(rng.randint(nb))
        pos[v] = np.asarray(rest[bn]) + rng.uniform(-0.8, 0.8, 3)
        near = [bn, int(parent[bn]) if parent[bn] >= 0 else bn, int(rng.randint(nb)), int(rng.randint(nb))]
        t = int(rng.choice([0, 1, 1, 2, 3]))
        st[v] = t
        ids[v] = near
        if t == 2:
            w = rng.uniform(0.05, 1, 4); wts[v] = (w / w.sum()).astype(np.float32)
        else:
            wts[v, 0] = rng.uniform(0.05, 0.95)
    nrm = rng.uniform(-1, 1, (nv, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    mesh = dict(positions=pos, normals=nrm.astype(np.float32), skin_type=st, bone_ids=ids, bone_weights=wts)

    poses = np.zeros((FRAMES, nb, 8), np.float32)
    poses[:, :, 7] = 1.0

    def quat(axis, ang):
        axis = np.asarray(axis, np.float64); axis /= np.linalg.norm(axis)
        return np.concatenate([axis * np.sin(ang / 2), [np.cos(ang / 2)]]).astype(np.float32)
    for f in range(FRAMES):
        t = f / 30.0
        poses[f, b0, :3] = (0.8 * np.sin(2.1 * t), 0.3 * np.sin(3.3 * t), 0.5 * np.cos(1.7 * t) - 0.5)
        poses[f, b0, 4:] = quat((0, 1, 0), 0.6 * np.sin(1.9 * t))
        poses[f, b1, 4:] = quat((1, 0, 0.2), 0.4 * np.sin(2.7 * t + 0.3))
        poses[f, b2, 4:] = quat((0.3, 0.2, 1), 0.5 * np.sin(3.1 * t))
        # the motion also poses bones that carry dynamic bodies (MMD motions do): a strict one gets a translation, which
        # Fix() keeps (total_translation_ + local_offset_), a plain one a rotation
        poses[f, chains[0][1], :3] = (0.05 * np.sin(t), 0.02, 0.0)
        poses[f, chains[1][0], 4:] = quat((0, 0, 1), 0.2 * np.sin(2 * t))
        poses[f, posts[0], 4:] = quat((1, 1, 0), 0.3 * np.sin(2.3 * t))
        poses[f, posts[3], :3] = (0.0, 0.1 * np.sin(1.3 * t), 0.0)
        poses[f, extra[0], 4:] = quat((0, 1, 1), 0.4 * np.cos(1.1 * t))
    return rig, mesh, bodies, joints, poses


def run(rig, mesh, bodies, joints, poses):
    ref = BulletReference(rig, mesh, bodies, joints)
    passive, strict, ghost = ref.body_info()
    out = {k: [] for k in ("palette_pre", "body_xf", "palette")}
    sums, full = [], {}
    for f in range(poses.shape[0]):
        o = ref.frame(poses[f])
        for k in out:
            out[k].append(o[k])
        sums.append((synth.checksum64(o["pos"]), synth.checksum64(o["nrm"])))
        if f % 8 == 0 or f == poses.shape[0] - 1:
            full[f] = (o["pos"], o["nrm"])
    ref.close()
    res = {k: np.stack(v) for k, v in out.items()}
    res.update(passive=passive, strict=strict, ghost=ghost, vertex_sums=np.asarray(sums, np.uint64),
               full_frames=np.asarray(sorted(full), np.int32), full_pos=np.stack([full[f][0] for f in sorted(full)]),
               full_nrm=np.stack([full[f][1] for f in sorted(full)]))
    return res


def main():
    rig, mesh, bodies, joints, poses = make_case()
    res = run(rig, mesh, bodies, joints, poses)
    moved = ~(res["passive"].astype(bool) | res["ghost"].astype(bool))
    # sanity: physics actually did something -- the final palette differs from the pre-physics one on the bodies' bones
    diff = np.abs(res["palette"] - res["palette_pre"])[:, bodies["bone"][moved]].max()
    assert diff > 0.05, diff
    assert np.isfinite(res["palette"]).all() and np.isfinite(res["body_xf"]).all()
    np.savez_compressed(OUT, poses=poses, **{"rig_" + k: v for k, v in rig.items()}, **{"mesh_" + k: v for k, v in mesh.items()},
                        **{"body_" + k: v for k, v in bodies.items()}, **{"joint_" + k: v for k, v in joints.items()}, **res)
    print(f"wrote {OUT}: {os.path.getsize(OUT)} bytes; {poses.shape[0]} frames, {rig['rest'].shape[0]} bones, "
          f"{bodies['bone'].shape[0]} bodies ({int(moved.sum())} write skinning matrices, {int(res['strict'].sum())} strict), "
          f"{joints['body'].shape[0]} joints; max |palette - palette_pre| on physics bones {diff:.3f}")


if __name__ == "__main__":
    main()
