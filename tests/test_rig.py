"""Bone tracks -> local poses -> bone palette.

CPU tests pin the C restatement (oracle/mmdx_oracle.c: mmdx_oracle_bone_pose / _bone_solve) against the
real libmmd and against the committed golden fixture, and cover the host-side compilation behind
mmdx_vmd_bind_bones / mmdx_skeleton_create.  GPU tests compare the HIP kernels with the oracle and the
fixture through the C ABI: bit-exact, no tolerance, on every seed used here.  (Stated tolerance of the IK solve
beyond these seeds: its sin/cos/asin/acos/atan2 go through the device's double libm where the reference's go
through glibc's; soaks of 4.6 million random solves found one instance whose palette differs, by 3.8e-6 at most --
DESIGN.md section 7 row 3, tools/soak_rig.py, tools/archive/probes/rig_mismatch_probe.py.)
"""
import os

import numpy as np
import pytest

from oracle.pyoracle import Reference, ReferenceMotion, reference_available
from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import pmx, synth, vmd
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer
from tests import golden_util as gu

BONES = ["センター", "上半身", "首", "頭", "左腕", "左ひじ", "右腕", "右ひじ", "BoneEN"]
DEFAULT_IP = bytes([20, 20, 0, 0, 20, 20, 20, 20, 107, 107, 107, 107, 107, 107, 107, 107] * 4)


@pytest.fixture(autouse=True)
def _lib(hip_lib):
    return hip_lib


def tracks_of(v):
    """name -> (frames, tr, rot, interp) of a parsed motion (after 'last record wins')."""
    out = {}
    for i, n in enumerate(v.bone_track_names):
        ks = v.bone_track(i)
        out[n] = (np.array([k["frame"] for k in ks], np.uint32),
                  np.array([k["translation"] for k in ks], np.float32).reshape(-1, 3),
                  np.array([k["rotation"] for k in ks], np.float32).reshape(-1, 4),
                  np.frombuffer(b"".join(k["interpolation"] for k in ks), np.uint8).view(np.int8).reshape(-1, 64))
    return out


def oracle_poses(oracle, v, model_bones, frames):
    tr = tracks_of(v)
    out = np.zeros((len(frames), len(model_bones), 8), np.float32)
    for j, n in enumerate(model_bones):
        t = tr.get(n)
        for i, f in enumerate(frames):
            out[i, j] = oracle.bone_pose(*t, f) if t is not None else oracle.bone_pose([], [], [], [], f)
    return out


def random_poses(ni, nb, seed):
    rng = np.random.RandomState(seed)
    p = np.zeros((ni, nb, 8), np.float32)
    p[..., 0:3] = rng.uniform(-1.5, 1.5, (ni, nb, 3))
    q = rng.normal(size=(ni, nb, 4))
    p[..., 4:8] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    p[0, 0, 4:8] = (0, 0, 0, 1)
    p[0, 0, 0:3] = (-0.0, 0.0, 1e-39)          # -0 and a denormal through "0 + t"
    return p


# ---------------------------------------------------------------------------------------- CPU ----
def test_golden_rig_oracle(oracle):
    """tests/golden/rig_small.vmd: libmmd's Motion::GetBonePose and bone-solve answers, recorded."""
    z = np.load(os.path.join(gu.GOLDEN_DIR, "rig_small_expect.npz"))
    v = vmd.Vmd(os.path.join(gu.GOLDEN_DIR, "rig_small.vmd"))
    names = [str(n) for n in z["model_bone_names"]]
    poses = oracle_poses(oracle, v, names, z["frames"])
    gu.assert_bits_equal(poses, z["expect_poses"], "poses")
    for i in range(len(z["frames"])):
        pal = oracle.bone_solve(z["rest"], z["parent"], z["expect_poses"][i], z["level"], z["flags"])
        gu.assert_bits_equal(pal, z["expect_palettes"][i], f"palette of frame {z['frames'][i]}")
    assert len(set(bytes(k) for k in tracks_of(v)[names[0]][3])) > 1   # the fixture has curved keys


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_oracle_bone_pose_vs_reference(oracle, tmp_path, seed):
    keys = synth.make_bone_keys(BONES, seed, keys_per=5 + seed, span=150)
    keys.append((BONES[2], keys[3][1] if keys[3][0] == BONES[2] else 7, (1, 2, 3), (0, 0, 0, 1), None))
    keys.append((BONES[0], 2 ** 24 + 9, (0, 1, 0), (0, 0.6, 0, 0.8), bytes([64] * 64)))   # frames beyond float's exact range
    np.random.RandomState(seed).shuffle(keys)
    p = tmp_path / "b.vmd"
    p.write_bytes(vmd.write_vmd(keys, []))
    v = vmd.Vmd(str(p))
    rm = ReferenceMotion(str(p))
    at = np.r_[np.arange(0, 160), 2 ** 24 + np.arange(0, 12), 4_000_000_000].astype(np.uint32)
    got = oracle_poses(oracle, v, BONES, at)
    for j, n in enumerate(BONES):
        want = np.stack([rm.bone_pose(n.encode("shift_jis"), int(f)) for f in at])
        gu.assert_bits_equal(got[:, j], want, f"track {n}")
    assert rm.bone_pose("無い".encode("shift_jis"), 3) is None
    rm.close()


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
def test_reference_curves_are_not_degenerate(tmp_path):
    """The oracle harness mirrors the viewer's include order, so libmmd's Bezier bisection uses the
    floating-point abs (see oracle/ref_harness.cpp): an S-shaped curve must not evaluate to a constant."""
    ip = bytearray(DEFAULT_IP)
    ip[0], ip[4], ip[8], ip[12] = 20, 100, 90, 30            # x channel
    keys = [("センター", 0, (0, 0, 0), (0, 0, 0, 1), bytes(ip)), ("センター", 100, (10, 10, 10), (0, 0, 0, 1), None)]
    p = tmp_path / "c.vmd"
    p.write_bytes(vmd.write_vmd(keys, []))
    rm = ReferenceMotion(str(p))
    xs = [float(rm.bone_pose("センター".encode("shift_jis"), f)[0]) for f in (10, 50, 90)]
    ys = [float(rm.bone_pose("センター".encode("shift_jis"), f)[1]) for f in (10, 50, 90)]
    rm.close()
    assert ys == [1.0, 5.0, 9.0]                             # linear channel
    assert xs[0] < xs[1] < xs[2] and abs(xs[0] - 1.0) > 0.5  # curved channel: monotone, far from linear


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
@pytest.mark.parametrize("nb,fwd,post,levels,seed", [(1, 0, 0.0, 1, 0), (7, 0, 0.0, 1, 1), (40, 5, 0.25, 3, 2),
                                                     (120, 12, 0.5, 4, 3), (33, 30, 1.0, 2, 4)])
def test_oracle_bone_solve_vs_reference(oracle, nb, fwd, post, levels, seed):
    rest, parent, level, flags = synth.make_skeleton(nb, seed, fwd, post, levels)
    if seed == 3:
        level[5] = -1                                        # (size_t) of a negative level sorts last
    ref = Reference.skeleton(rest, parent, level, flags)
    for i, poses in enumerate(random_poses(3, nb, seed)):
        gu.assert_bits_equal(oracle.bone_solve(rest, parent, poses, level, flags), ref.solve(poses), f"palette {i}")
    ref.close()


def test_bind_bones_host_side():
    keys = synth.make_bone_keys(BONES[:4], 5, keys_per=4, curved=1.0)
    v = vmd.Vmd(vmd.write_vmd(keys, []))
    bm = v.bind_bones([BONES[1], "無い", BONES[0], BONES[3]])
    assert (bm.nb, bm.n_mapped, bm.n_keys) == (4, 3, 12)
    assert 0 < bm.n_curves <= 12 * 4
    lin = vmd.Vmd(vmd.write_vmd([(BONES[0], 0, (0, 0, 0), (0, 0, 0, 1), None),
                                 (BONES[0], 9, (1, 0, 0), (0, 0, 0, 1), None)], [])).bind_bones(BONES[:1])
    assert (lin.n_keys, lin.n_curves) == (2, 0)              # the default interpolation is linear: no table
    empty = v.bind_bones([])
    assert (empty.nb, empty.n_keys) == (0, 0)


def test_skeleton_create_host_side():
    rest, parent, level, flags = synth.make_skeleton(50, 1, 6, 0.3, 3)
    sk = vmd.Skeleton(rest, parent, level, flags)
    assert sk.info["n_bones"] == 50 and sk.info["n_pre_physics"] + sk.info["n_post_physics"] == 50
    assert 1 <= sk.info["max_chain"] <= 51
    for bad in (0x0020, 0x0100, 0x0200):                     # IK / append flags without their arrays
        f2 = flags.copy()
        f2[7] |= bad
        with pytest.raises(api.MmdxError) as e:
            vmd.Skeleton(rest, parent, level, f2)
        assert e.value.status == 1 and "NULL" in str(e.value)
    p2 = parent.copy()
    p2[9] = 9
    with pytest.raises(api.MmdxError) as e:
        vmd.Skeleton(rest, p2, level, flags)
    assert e.value.status == 1 and "own parent" in str(e.value)
    p3 = parent.copy()
    p3[3] = 1000                                             # out of range = no parent, as in the reference
    assert vmd.Skeleton(rest, p3).info["n_bones"] == 50


def test_ik_skeleton_create_host_side():
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(60, 3, n_ik=4, n_append=4)
    sk = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik)
    assert sk.info["solver"] == vmd.SOLVER_SERIAL and sk.info["n_ik_bones"] == 4
    assert sk.info["n_ik_links"] == ik["link_bone"].size and 0 < sk.info["n_append_bones"] <= 4
    ikb = int(np.flatnonzero(flags & 0x20)[0])
    bad = dict(ik, target=ik["target"].copy())
    bad["target"][ikb] = 60
    with pytest.raises(api.MmdxError) as e:
        vmd.Skeleton(rest, parent, level, flags, ap, ar, bad)
    assert e.value.status == 2                               # BAD_INDEX: validated at create, never read at solve
    bad = dict(ik, link_bone=ik["link_bone"].copy())
    bad["link_bone"][0] = -1
    with pytest.raises(api.MmdxError) as e:
        vmd.Skeleton(rest, parent, level, flags, ap, ar, bad)
    assert e.value.status == 2
    nested = dict(ik, target=ik["target"].copy())                # an IK bone as target: legal (nested solve) ...
    nested["target"][ikb] = int(np.flatnonzero(flags & 0x20)[1])
    vmd.Skeleton(rest, parent, level, flags, ap, ar, nested).close()
    nested["target"][ikb] = ikb                                   # ... its own target: endless recursion upstream
    with pytest.raises(api.MmdxError) as e:
        vmd.Skeleton(rest, parent, level, flags, ap, ar, nested)
    assert e.value.status == 6 and "own solve" in str(e.value)


def test_solve_round_schedule_host_side(monkeypatch):
    """The ordered solver's schedule (rig.cpp schedule_rounds): independent bones share a round, so a 300-bone
    rig takes far fewer rounds than bones; an FK-only rig has none; MMDX_SOLVE_SEQUENTIAL=1 switches it off.
    (That every event sees the serial sequence's state is checked on random rigs by the schedule replay in
    tests/plan_sanitizer_driver.cpp.)"""
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(300, 3003, n_ik=8, n_append=12, post_physics=0.3, levels=3)
    info = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik).info
    assert info["solver"] == vmd.SOLVER_SERIAL and 2 <= info["n_solve_rounds"] < 300 // 4
    assert vmd.Skeleton(rest, parent).info["n_solve_rounds"] == 0
    line = np.arange(-1, 39).astype(np.int32)                # a single 40-bone line: nothing to run side by side
    f = np.zeros(40, np.uint16)
    f[39] = 0x0100
    assert vmd.Skeleton(np.zeros((40, 3), np.float32), line, None, f, np.full(40, 3, np.int32),
                        np.ones(40, np.float32)).info["n_solve_rounds"] == 40
    monkeypatch.setenv("MMDX_SOLVE_SEQUENTIAL", "1")
    assert vmd.Skeleton(rest, parent, level, flags, ap, ar, ik).info["n_solve_rounds"] == 300


def test_ik_rounds_that_run_with_sixteen_lanes_per_solve_host_side():
    """Which rounds of the schedule go to the 16-lanes-per-solve kernel (rig.cpp round_coop, mmdx_skeleton_info): rounds made of
    CCD-IK solves on plain chains -- the bench rig's eight chains share ONE such round; a chain longer than the kernel's six links
    is solved by the ordered kernel; a rig with nested IK keeps one lane per solve throughout; FK / append-only rigs have none."""
    rig = synth.make_ik_rig(300, 3003, n_ik=8, n_append=12, post_physics=0.0, levels=1)         # tools/rig_bench.py's IK rig
    info = vmd.Skeleton(*rig).info
    assert info["n_ik_bones"] == 8 and 1 <= info["n_ik_rounds_16_lanes"] <= 8 and info["n_ik_rounds_16_lanes"] < info["n_solve_rounds"]
    assert vmd.Skeleton(rig[0], rig[1]).info["n_ik_rounds_16_lanes"] == 0
    no_ik = (np.asarray(rig[3]) & ~np.uint16(0x20)).astype(np.uint16)
    assert vmd.Skeleton(rig[0], rig[1], rig[2], no_ik, rig[4], rig[5]).info["n_ik_rounds_16_lanes"] == 0
    nested = synth.make_nested_ik_rig(60, 11)
    assert vmd.Skeleton(*nested).info["n_ik_rounds_16_lanes"] == 0
    nb = 40                                                   # one 7-link chain on a line: past kMaxFastLinks
    rest = np.stack([np.zeros(nb), np.arange(nb) * 0.1, np.zeros(nb)], 1).astype(np.float32)
    parent = np.arange(-1, nb - 1).astype(np.int32)
    parent[nb - 1] = 0
    flags = np.zeros(nb, np.uint16)
    flags[nb - 1] = 0x20
    for n_links, want in ((6, 1), (7, 0)):
        links = list(range(nb - 3, nb - 3 - n_links, -1))
        ik = dict(target=np.full(nb, -1, np.int32), loop=np.zeros(nb, np.int32), angle=np.zeros(nb, np.float32),
                  link_off=np.zeros(nb + 1, np.uint32), link_bone=np.asarray(links, np.int32),
                  link_limited=np.zeros(n_links, np.uint8), link_lo=np.zeros((n_links, 3), np.float32),
                  link_hi=np.zeros((n_links, 3), np.float32))
        ik["target"][nb - 1], ik["loop"][nb - 1], ik["angle"][nb - 1] = nb - 2, 12, 1.0
        ik["link_off"][nb:] = n_links
        assert vmd.Skeleton(rest, parent, None, flags, None, None, ik).info["n_ik_rounds_16_lanes"] == want, n_links


IK_CASES = [(30, 0, 2, 2), (44, 1, 3, 4), (80, 2, 5, 6), (150, 3, 6, 10), (61, 4, 4, 0), (52, 5, 0, 8)]


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
@pytest.mark.parametrize("nb,seed,n_ik,n_app", IK_CASES)
def test_oracle_full_bone_solve_vs_reference(oracle, nb, seed, n_ik, n_app):
    """Append bones + CCD-IK: the C restatement against libmmd's Poser, bit for bit."""
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(nb, seed, n_ik=n_ik, n_append=n_app)
    ref = Reference.skeleton(rest, parent, level, flags, ap, ar, ik)
    moved = 0
    for i, poses in enumerate(random_poses(4, nb, 100 + seed)):
        got = oracle.bone_solve_full(rest, parent, poses, level, flags, ap, ar, ik)
        gu.assert_bits_equal(got, ref.solve(poses), f"palette {i}")
        plain = oracle.bone_solve_full(rest, parent, poses, level, flags & 0x1000)
        moved += int(np.any(gu.bits(got) != gu.bits(plain)))
    ref.close()
    assert moved == 4                                        # the IK / append machinery really ran


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
@pytest.mark.parametrize("nb,seed,n_ik,n_app", [(30, 0, 0, 0), (44, 1, 3, 4), (80, 2, 5, 6)])
def test_oracle_bone_morphs_vs_reference(oracle, nb, seed, n_ik, n_app):
    """Bone morphs (translation + SLerp'ed rotation, through groups) feeding the bone solve."""
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(nb, seed, n_ik=n_ik, n_append=n_app)
    morphs = synth.make_bone_morphs(nb, 70 + seed)
    ref = Reference.skeleton(rest, parent, level, flags, ap, ar, ik, morphs)
    rng = np.random.RandomState(seed)
    for i, poses in enumerate(random_poses(4, nb, 400 + seed)):
        rates = rng.choice([0, 5e-8, 0.3, 1.0, 1.7, -0.5], morphs["type"].size).astype(np.float32)
        got = oracle.bone_solve_full(rest, parent, poses, level, flags, ap, ar, ik, morphs, rates)
        gu.assert_bits_equal(got, ref.solve(poses, rates), f"palette {i}")
    ref.close()


def test_bone_morph_table_host_side():
    rest, parent, level, flags = synth.make_skeleton(20, 3)
    morphs = synth.make_bone_morphs(20, 9)
    sk = vmd.Skeleton(rest, parent, level, flags, morphs=morphs)
    assert sk.info["n_bone_morph_entries"] > 0 and sk.info["solver"] == vmd.SOLVER_PARALLEL_FK
    bad = dict(morphs, index=morphs["index"].copy())
    e = int(morphs["offset"][np.flatnonzero(morphs["type"] == 2)[0]])
    bad["index"][e] = 20                                     # bone index out of range
    with pytest.raises(api.MmdxError) as ex:
        vmd.Skeleton(rest, parent, level, flags, morphs=bad)
    assert ex.value.status == 2
    g = int(np.flatnonzero(morphs["type"] == 0)[0])
    cyc = dict(morphs, index=morphs["index"].copy())
    cyc["index"][int(morphs["offset"][g])] = g               # a group that contains itself
    with pytest.raises(api.MmdxError) as ex:
        vmd.Skeleton(rest, parent, level, flags, morphs=cyc)
    assert ex.value.status == 6 and "contains itself" in str(ex.value)


def test_pmx_bone_morph_rotation_round_trip():
    m = synth.make_model(60, 12, 0, 1, seed=3)
    morphs = synth.make_bone_morphs(12, 4)
    m.morph_type, m.morph_off = morphs["type"], morphs["offset"]
    m.morph_index, m.morph_value = morphs["index"], morphs["value"]
    pm = pmx.load_pmx(pmx.write_pmx(m, pmx.PmxWriteOptions(morph_rotation=morphs["rotation"], bone_flag_variety=False)))
    is_bone = np.repeat(morphs["type"] == 2, np.diff(morphs["offset"].astype(np.int64)))
    assert np.array_equal(pm.morph_rotation[is_bone], morphs["rotation"][is_bone])
    assert np.all(pm.morph_rotation[~is_bone] == np.array([0, 0, 0, 1], np.float32))
    assert pm.skeleton().info["n_bone_morph_entries"] > 0


def test_golden_bone_morph_rig_oracle(oracle):
    z = np.load(os.path.join(gu.GOLDEN_DIR, "rig_morph_expect.npz"))
    for tag in ("fk", "ik"):
        ik = {k[len(tag) + 4:]: z[k] for k in z.files if k.startswith(tag + "_ik_")} or None
        morphs = {k[len(tag) + 7:]: z[k] for k in z.files if k.startswith(tag + "_morph_")}
        for i in range(z[tag + "_poses"].shape[0]):
            got = oracle.bone_solve_full(z[tag + "_rest"], z[tag + "_parent"], z[tag + "_poses"][i], z[tag + "_level"],
                                         z[tag + "_flags"], z[tag + "_append_parent"], z[tag + "_append_ratio"], ik,
                                         morphs, z[tag + "_rates"][i])
            gu.assert_bits_equal(got, z[tag + "_expect_palettes"][i], f"{tag} palette {i}")


def test_golden_ik_rig_oracle(oracle):
    """tests/golden/rig_ik_expect.npz: libmmd's palettes for rigs with IK chains and append bones."""
    z = np.load(os.path.join(gu.GOLDEN_DIR, "rig_ik_expect.npz"))
    ik = {k[3:]: z[k] for k in z.files if k.startswith("ik_")}
    for i in range(z["poses"].shape[0]):
        got = oracle.bone_solve_full(z["rest"], z["parent"], z["poses"][i], z["level"], z["flags"],
                                     z["append_parent"], z["append_ratio"], ik)
        gu.assert_bits_equal(got, z["expect_palettes"][i], f"palette {i}")


def test_rig_entry_points_fail_loudly_without_gpu_or_with_bad_arguments():
    """No CPU fallback: evaluation / solve need the HIP device; bad arguments are refused before that."""
    import ctypes as C
    lib = api.lib()
    keys = synth.make_bone_keys(BONES[:3], 1, keys_per=3)
    bm = vmd.Vmd(vmd.write_vmd(keys, [])).bind_bones(BONES[:3])
    rest, parent, level, flags = synth.make_skeleton(3, 1)
    sk = vmd.Skeleton(rest, parent, level, flags)
    frames = np.zeros(2, np.uint32)
    poses = np.zeros((2, 3, 8), np.float32)
    pal = np.zeros((2, 3, 16), np.float32)
    assert lib.mmdx_bone_motion_eval(bm.h, None, 0, frames.ctypes.data, 0, poses.ctypes.data) == 1   # n_instances == 0
    assert lib.mmdx_bone_motion_eval(None, None, 2, frames.ctypes.data, 0, poses.ctypes.data) == 1
    assert lib.mmdx_skeleton_solve(sk.h, None, 2, None, 0, pal.ctypes.data) == 1
    bad = vmd.SkeletonDesc()
    bad.struct_size = 8
    h = C.c_void_p()
    assert lib.mmdx_skeleton_create(C.byref(bad), C.byref(h)) == 1 and "struct_size" in lib.mmdx_last_error_string().decode()
    n = C.c_int32()
    if lib.mmdx_device_count(C.byref(n)) != 0 or n.value < 1:      # this machine has no GPU: must say so, not compute
        for call in (lambda: bm.eval(frames), lambda: sk.solve(poses)):
            with pytest.raises(api.MmdxError) as e:
                call()
            assert e.value.status == 3 and "no HIP device" in str(e.value)


def rigged_pmx(nb, seed):
    """A small PMX file whose bone block carries a synth.make_ik_rig rig."""
    rig = synth.make_ik_rig(nb, seed, n_ik=3, n_append=4)
    m = synth.make_model(120, nb, 2, 10, seed=seed)
    m.bone_pos, m.bone_parent = rig[0].copy(), rig[1].astype(m.bone_parent.dtype)
    return pmx.write_pmx(m, pmx.PmxWriteOptions(rig=rig, index_width=(0, 1, 1, 1 + seed % 2, 0, 1))), rig


@pytest.mark.parametrize("seed", [0, 1])
def test_pmx_rig_round_trip(seed):
    data, (rest, parent, level, flags, ap, ar, ik) = rigged_pmx(40, seed)
    pm = pmx.load_pmx(data)
    assert np.array_equal(pm.bone_transform_level, level)
    assert np.array_equal(pm.bone_flags & 0x1320, flags & 0x1320)
    has_ap = (flags & 0x0300) != 0
    want_ap = np.where(ap >= 40, ap.astype(np.int8 if pm.info["index_width"][3] == 1 else np.int16), ap)   # as stored
    assert np.array_equal(pm.append_parent[has_ap], want_ap[has_ap]) and np.array_equal(pm.append_ratio[has_ap], ar[has_ap])
    has_ik = (flags & 0x20) != 0
    for k in ("target", "loop", "angle"):
        assert np.array_equal(pm.ik[k][has_ik], ik[k][has_ik]), k
    for k in ("link_off", "link_bone", "link_limited"):
        assert np.array_equal(pm.ik[k], ik[k]), k
    lim = ik["link_limited"].astype(bool)
    assert np.array_equal(pm.ik["link_lo"][lim], ik["link_lo"][lim]) and np.array_equal(pm.ik["link_hi"][lim], ik["link_hi"][lim])
    sk = pm.skeleton()
    assert sk.info["solver"] == vmd.SOLVER_SERIAL and sk.info["n_ik_bones"] == int(has_ik.sum())


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_pmx_rig_through_both_readers(oracle, tmp_path, seed):
    """The same .pmx bytes: libmmd's PmxReader + Poser vs this loader's rig arrays through the oracle."""
    data, _ = rigged_pmx(36 + seed, seed)
    path = tmp_path / "rig.pmx"
    path.write_bytes(data)
    pm = pmx.load_pmx(str(path))
    ref = Reference.from_pmx(str(path))
    nb = pm.flat.nb
    for i, poses in enumerate(random_poses(3, nb, 300 + seed)):
        for b in range(nb):
            ref.set_bone_pose(b, poses[b, 0:3], poses[b, 4:8])
        ref.pose()
        got = oracle.bone_solve_full(pm.flat.bone_pos, pm.flat.bone_parent, poses, pm.bone_transform_level,
                                     pm.bone_flags, pm.append_parent, pm.append_ratio, pm.ik)
        gu.assert_bits_equal(got, ref.get_palette(), f"palette {i}")
    ref.close()


# ---------------------------------------------------------------------------------------- GPU ----
@pytest.mark.gpu
def test_gpu_golden_rig():
    z = np.load(os.path.join(gu.GOLDEN_DIR, "rig_small_expect.npz"))
    v = vmd.Vmd(os.path.join(gu.GOLDEN_DIR, "rig_small.vmd"))
    names = [str(n) for n in z["model_bone_names"]]
    bm = v.bind_bones(names)
    poses = bm.eval(z["frames"])
    gu.assert_bits_equal(poses, z["expect_poses"], "poses")
    sk = vmd.Skeleton(z["rest"], z["parent"], z["level"], z["flags"])
    gu.assert_bits_equal(sk.solve(poses), z["expect_palettes"], "palettes")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gpu_bone_motion_vs_oracle(oracle, seed):
    names = [f"bone{i}" for i in range(40)]
    keys = synth.make_bone_keys(names[:33], 10 + seed, keys_per=3 + 4 * seed, span=400)
    keys.append((names[0], 2 ** 24 + 9, (0, 1, 0), (0, 0.6, 0, 0.8), bytes([64] * 64)))
    v = vmd.Vmd(vmd.write_vmd(keys, []))
    at = np.r_[np.arange(0, 420, 1 + seed), 2 ** 24 + np.arange(0, 12), 4_000_000_000].astype(np.uint32)
    got = v.bind_bones(names).eval(at)
    gu.assert_bits_equal(got, oracle_poses(oracle, v, names, at), "poses")


@pytest.mark.gpu
@pytest.mark.parametrize("nb,fwd,post,levels,seed", [(1, 0, 0.0, 1, 0), (40, 5, 0.25, 3, 2), (300, 20, 0.3, 4, 3),
                                                     (33, 30, 1.0, 2, 4), (2048, 100, 0.1, 5, 5)])
def test_gpu_skeleton_vs_oracle(oracle, nb, fwd, post, levels, seed):
    rest, parent, level, flags = synth.make_skeleton(nb, seed, fwd, post, levels)
    poses = random_poses(5, nb, seed)
    got = vmd.Skeleton(rest, parent, level, flags).solve(poses)
    for i in range(poses.shape[0]):
        gu.assert_bits_equal(got[i], oracle.bone_solve(rest, parent, poses[i], level, flags), f"palette {i}")


@pytest.mark.gpu
def test_gpu_motion_to_vertices_all_on_device(oracle):
    """VMD -> local poses -> palettes -> skinned vertices without leaving HBM, against the oracle run
    step by step on the host: the palette producer feeds mmdx_deform_batched bit-exactly."""
    m = synth.make_model(3000, 48, 6, 200, seed=21)
    names = [f"b{i}" for i in range(m.nb)]
    mnames = [f"m{i}" for i in range(m.nm)]
    rng = np.random.RandomState(4)
    v = vmd.Vmd(vmd.write_vmd(synth.make_bone_keys(names[:40], 8, keys_per=5, span=120),
                              [(n, int(f), float(np.float32(rng.uniform(0, 1)))) for n in mnames for f in (0, 50, 110)]))
    parent = np.asarray(m.bone_parent, np.int32)
    sk = vmd.Skeleton(m.bone_pos, parent)
    bm, mm = v.bind_bones(names), v.bind_morphs(mnames)
    ni = 37
    frames = (np.arange(ni) * 3 + 1).astype(np.uint32)
    dm = DeformModel(m)
    d_fr = DeviceBuffer.from_numpy(frames)
    d_pose, d_pal = DeviceBuffer(ni * m.nb * 32), DeviceBuffer(ni * m.nb * 64)
    d_w = DeviceBuffer(ni * m.nm * 4)
    sa, sb = dm.out_sizes(api.OUT_SOA, ni)
    d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
    bm.eval_device(ni, d_fr.ptr, d_pose.ptr, dm)
    sk.solve_device(ni, d_pose.ptr, d_pal.ptr, dm)
    mm.eval_device(ni, d_fr.ptr, d_w.ptr, dm)
    dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA,
                          api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE)
    dm.sync()
    pos = d_a.download((ni, m.nv, 3), np.float32)
    nrm = d_b.download((ni, m.nv, 3), np.float32)
    poses = oracle_poses(oracle, v, names, frames)
    rates = mm.eval(frames)
    for i in range(ni):
        pal = oracle.bone_solve(m.bone_pos, parent.astype(np.int64), poses[i])
        want_p, want_n = oracle.deform(m, rates[i], pal)
        gu.assert_bits_equal(pos[i], want_p, f"positions of instance {i}")
        gu.assert_bits_equal(nrm[i], want_n, f"normals of instance {i}")


def _mismatch_report(got, want):
    g, w = gu.bits(got), gu.bits(want)
    bad = np.argwhere(g != w)
    return bad.shape[0], (float(np.nanmax(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))))
                          if bad.shape[0] else 0.0)


@pytest.mark.gpu
def test_gpu_golden_ik_rig():
    z = np.load(os.path.join(gu.GOLDEN_DIR, "rig_ik_expect.npz"))
    ik = {k[3:]: z[k] for k in z.files if k.startswith("ik_")}
    sk = vmd.Skeleton(z["rest"], z["parent"], z["level"], z["flags"], z["append_parent"], z["append_ratio"], ik)
    assert sk.info["solver"] == vmd.SOLVER_SERIAL
    gu.assert_bits_equal(sk.solve(z["poses"]), z["expect_palettes"], "palettes")


@pytest.mark.gpu
@pytest.mark.parametrize("nb,seed,n_ik,n_app", IK_CASES + [(300, 7, 8, 12)])
def test_gpu_ik_skeleton_vs_oracle(oracle, nb, seed, n_ik, n_app):
    """The ordered device solver against the oracle, 70 instances (the last workgroup partly empty)."""
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(nb, seed, n_ik=n_ik, n_append=n_app)
    poses = random_poses(70, nb, 200 + seed)
    got = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik).solve(poses)
    for i in range(poses.shape[0]):
        want = oracle.bone_solve_full(rest, parent, poses[i], level, flags, ap, ar, ik)
        gu.assert_bits_equal(got[i], want, f"palette of instance {i}")


@pytest.mark.gpu
def test_gpu_pmx_rig_to_palettes(oracle):
    """.pmx bytes -> this loader -> mmdx_pmx_get_skeleton_desc -> device solve, against the oracle."""
    data, _ = rigged_pmx(64, 5)
    pm = pmx.load_pmx(data)
    poses = random_poses(9, pm.flat.nb, 77)
    got = pm.skeleton().solve(poses)
    for i in range(poses.shape[0]):
        want = oracle.bone_solve_full(pm.flat.bone_pos, pm.flat.bone_parent, poses[i], pm.bone_transform_level,
                                      pm.bone_flags, pm.append_parent, pm.append_ratio, pm.ik)
        gu.assert_bits_equal(got[i], want, f"palette of instance {i}")


def _golden_morph_case(z, tag):
    ik = {k[len(tag) + 4:]: z[k] for k in z.files if k.startswith(tag + "_ik_")} or None
    morphs = {k[len(tag) + 7:]: z[k] for k in z.files if k.startswith(tag + "_morph_")}
    return ik, morphs


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["fk", "ik"])
def test_gpu_golden_bone_morph_rig(tag):
    z = np.load(os.path.join(gu.GOLDEN_DIR, "rig_morph_expect.npz"))
    ik, morphs = _golden_morph_case(z, tag)
    sk = vmd.Skeleton(z[tag + "_rest"], z[tag + "_parent"], z[tag + "_level"], z[tag + "_flags"],
                      z[tag + "_append_parent"], z[tag + "_append_ratio"], ik, morphs)
    assert sk.info["solver"] == (vmd.SOLVER_SERIAL if tag == "ik" else vmd.SOLVER_PARALLEL_FK)
    got = sk.solve(z[tag + "_poses"], morph_weights=z[tag + "_rates"])
    gu.assert_bits_equal(got, z[tag + "_expect_palettes"], "palettes")


@pytest.mark.gpu
@pytest.mark.parametrize("nb,seed,n_ik,n_app", [(30, 0, 0, 0), (44, 1, 3, 4), (150, 2, 6, 8)])
def test_gpu_bone_morphs_vs_oracle(oracle, nb, seed, n_ik, n_app):
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(nb, seed, n_ik=n_ik, n_append=n_app)
    morphs = synth.make_bone_morphs(nb, 90 + seed)
    nm = morphs["type"].size
    sk = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik, morphs)
    poses = random_poses(70, nb, 500 + seed)
    rng = np.random.RandomState(seed)
    rates = rng.choice([0, 5e-8, 0.3, 1.0, 1.7, -0.5], (70, nm)).astype(np.float32)
    got = sk.solve(poses, morph_weights=rates)
    shared = sk.solve(poses, morph_weights=rates[3])
    plain = sk.solve(poses)
    for i in range(70):
        gu.assert_bits_equal(got[i], oracle.bone_solve_full(rest, parent, poses[i], level, flags, ap, ar, ik, morphs, rates[i]),
                             f"palette of instance {i}")
        gu.assert_bits_equal(plain[i], oracle.bone_solve_full(rest, parent, poses[i], level, flags, ap, ar, ik), f"plain {i}")
    gu.assert_bits_equal(shared[9], oracle.bone_solve_full(rest, parent, poses[9], level, flags, ap, ar, ik, morphs, rates[3]),
                         "shared weights")
    # device-resident operands
    d_pose, d_w, d_out = DeviceBuffer.from_numpy(poses), DeviceBuffer.from_numpy(rates), DeviceBuffer(70 * nb * 64)
    sk.solve_device(70, d_pose.ptr, d_out.ptr, None, d_w.ptr)
    api.check(api.lib().mmdx_device_synchronize())
    gu.assert_bits_equal(d_out.download((70, nb, 16), np.float32), got, "device-resident call")


@pytest.mark.gpu
def test_gpu_ik_skeleton_crowd_beyond_one_workgroup_per_cu(oracle):
    """More instances than 16 x (number of CUs): the ordered solver switches to its two-workgroups-per-CU variant (256 VGPRs,
    spills; rig_kernels.hip launch_skeleton_ordered).  4 200 instances of a 44-bone rig with 3 IK chains and 4 append bones,
    every 7th instance plus both ends against the oracle; the same poses through the small-crowd variant (the first 70
    instances alone) must give the same bits."""
    nb, seed = 44, 1
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(nb, seed, n_ik=3, n_append=4)
    ni = 4200
    poses = np.tile(random_poses(300, nb, 977), (ni // 300, 1, 1))
    rng = np.random.RandomState(5)
    poses[..., 0:3] += rng.uniform(-0.2, 0.2, (ni, nb, 3)).astype(np.float32)     # every instance its own pose
    sk = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik)
    got = sk.solve(poses)
    for i in sorted(set(range(0, ni, 7)) | {1, ni - 2, ni - 1}):
        want = oracle.bone_solve_full(rest, parent, poses[i], level, flags, ap, ar, ik)
        gu.assert_bits_equal_or_both_nan(got[i], want, f"palette of instance {i}")
    small = sk.solve(poses[:70])
    assert np.array_equal(np.nan_to_num(small).view(np.uint32), np.nan_to_num(got[:70]).view(np.uint32))


@pytest.mark.gpu
def test_gpu_full_size_crowd_rig(oracle):
    """BASELINE config-3 dimensions for the palette producer: 1024 instances x 300 bones, a 6000-key motion with
    ~16k distinct curve tables, every instance at its own frame.  Poses of 48 sampled instances and ALL 1024
    palettes (FK rig and IK rig) against the oracle."""
    ni, nb = 1024, 300
    names = [f"b{i}" for i in range(nb)]
    v = vmd.Vmd(vmd.write_vmd(synth.make_bone_keys(names, 303, keys_per=20, span=600), []))
    frames = ((np.arange(ni) * 7) % 600).astype(np.uint32)
    poses = v.bind_bones(names).eval(frames)
    sample = np.r_[0:16, 500:516, ni - 16:ni]
    gu.assert_bits_equal(poses[sample], oracle_poses(oracle, v, names, frames[sample]), "sampled poses")
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(nb, 3003, n_ik=8, n_append=12, post_physics=0.0, levels=1)
    fk = vmd.Skeleton(rest, parent).solve(poses)
    iksk = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik)
    assert iksk.info["solver"] == vmd.SOLVER_SERIAL
    ikp = iksk.solve(poses)
    n_nan = 0
    for i in range(ni):
        gu.assert_bits_equal(fk[i], oracle.bone_solve(rest, parent, poses[i]), f"FK palette of instance {i}")
        want = oracle.bone_solve_full(rest, parent, poses[i], level, flags, ap, ar, ik)
        gu.assert_bits_equal_or_both_nan(ikp[i], want, f"IK palette of instance {i}")
        n_nan += int(np.isnan(want).any())
    assert n_nan < ni // 4                               # degenerate chains (NaN upstream too) stay the exception


@pytest.mark.gpu
@pytest.mark.parametrize("ni", [1, 37, 200])
def test_gpu_round_schedule_equals_sequential(monkeypatch, ni):
    """The round schedule against the same kernel fed one event per round (MMDX_SOLVE_SEQUENTIAL=1): identical
    bits on rigs dense with IK chains, append bones and post-physics bones; instance counts that leave the last
    workgroup partly empty."""
    for seed in range(12):
        nb = 24 + 7 * seed
        rig = synth.make_ik_rig(nb, 500 + seed, n_ik=1 + seed % 6, n_append=seed % 7, post_physics=0.3, levels=3)
        poses = random_poses(ni, nb, 900 + seed)
        monkeypatch.delenv("MMDX_SOLVE_SEQUENTIAL", raising=False)
        sched = vmd.Skeleton(*rig)
        monkeypatch.setenv("MMDX_SOLVE_SEQUENTIAL", "1")
        seq = vmd.Skeleton(*rig)
        assert seq.info["n_solve_rounds"] == nb and sched.info["n_solve_rounds"] < nb
        gu.assert_bits_equal_or_both_nan(sched.solve(poses), seq.solve(poses), f"rig {seed}")


@pytest.mark.gpu
@pytest.mark.parametrize("ni", [5, 16, 70])
def test_gpu_ik_sixteen_lanes_per_solve_equals_one_lane(monkeypatch, oracle, ni):
    """The two CCD-IK solvers -- ik_coop_kernel (16 lanes per solve, the schedule run in segments around the IK rounds: the
    default) and the ordered kernel's one lane per solve (MMDX_IK_COOP=0) -- on rigs dense with IK chains (every limit kind and
    Euler order of the reference), append bones, several transform levels and post-physics bones; instance counts that leave
    groups, waves and blocks partly empty.  Identical bits, and the oracle's on a sample."""
    for seed in range(10):
        nb = 30 + 9 * seed
        rig = synth.make_ik_rig(nb, 700 + seed, n_ik=1 + seed % 7, n_append=seed % 5, post_physics=0.25, levels=1 + seed % 3)
        poses = random_poses(ni, nb, 1700 + seed)
        sk = vmd.Skeleton(*rig)
        monkeypatch.setenv("MMDX_IK_COOP", "1")
        coop = sk.solve(poses)
        monkeypatch.setenv("MMDX_IK_COOP", "0")
        lane = sk.solve(poses)
        monkeypatch.delenv("MMDX_IK_COOP")
        gu.assert_bits_equal_or_both_nan(coop, lane, f"rig {seed}")
        want = oracle.bone_solve_full(rig[0], rig[1], poses[ni - 1], rig[2], rig[3], rig[4], rig[5], rig[6])
        gu.assert_bits_equal_or_both_nan(coop[ni - 1], want, f"rig {seed} vs oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("n_links", [1, 6, 7, 12])
def test_gpu_long_ik_chains(oracle, n_links):
    """A 200-bone line with one IK bone at its end: chains of up to 6 links run on the solver's LDS window,
    longer ones on the HBM scratch state -- both sides of that boundary against the oracle."""
    nb = 200
    rest = np.stack([np.zeros(nb), np.arange(nb) * 0.1, np.zeros(nb)], 1).astype(np.float32)
    parent = np.arange(-1, nb - 1).astype(np.int32)
    parent[nb - 1] = 0
    flags = np.zeros(nb, np.uint16)
    flags[nb - 1] = 0x20
    rest[nb - 1] = rest[nb - 2] + np.float32(0.05)
    tgt = nb - 2
    links = list(range(tgt - 1, tgt - 1 - n_links, -1))
    ik = dict(target=np.full(nb, -1, np.int32), loop=np.zeros(nb, np.int32), angle=np.zeros(nb, np.float32),
              link_off=np.zeros(nb + 1, np.uint32), link_bone=np.asarray(links, np.int32),
              link_limited=np.zeros(n_links, np.uint8), link_lo=np.zeros((n_links, 3), np.float32),
              link_hi=np.zeros((n_links, 3), np.float32))
    ik["target"][nb - 1], ik["loop"][nb - 1], ik["angle"][nb - 1] = tgt, 12, 1.0
    ik["link_off"][nb:] = n_links
    sk = vmd.Skeleton(rest, parent, None, flags, None, None, ik)
    rng = np.random.RandomState(n_links)
    poses = np.zeros((70, nb, 8), np.float32)
    poses[..., 7] = 1
    poses[:, nb - 1, 0:3] = rng.uniform(-1, 1, (70, 3))
    got = sk.solve(poses)
    moved = 0
    for i in range(70):
        want = oracle.bone_solve_full(rest, parent, poses[i], None, flags, None, None, ik)
        gu.assert_bits_equal_or_both_nan(got[i], want, f"instance {i}")
        moved += int(not np.array_equal(want, oracle.bone_solve(rest, parent, poses[i])))
    assert moved > 60                                    # the chains really bend


@pytest.mark.gpu
def test_gpu_many_long_chains_share_a_round(oracle):
    """Eight independent 6-link chains: all of them fit one round, each on its own LDS window -- 95 KB of dynamic
    LDS, past the 64 KB a kernel gets without asking -- and the result is the oracle's."""
    n_chains, seg, n_links = 8, 10, 6
    nb = n_chains * seg
    rest = np.zeros((nb, 3), np.float32)
    parent = np.full(nb, -1, np.int32)
    flags = np.zeros(nb, np.uint16)
    ik = dict(target=np.full(nb, -1, np.int32), loop=np.zeros(nb, np.int32), angle=np.zeros(nb, np.float32),
              link_off=np.zeros(nb + 1, np.uint32), link_bone=[], link_limited=[], link_lo=[], link_hi=[])
    for c in range(n_chains):
        base = c * seg
        for k in range(seg - 1):                               # a line base .. base+8
            rest[base + k] = (c * 2.0, 0.1 * k, 0.0)
            parent[base + k] = base + k - 1 if k else -1
        ikb, tgt = base + seg - 1, base + seg - 2
        parent[ikb] = base
        rest[ikb] = rest[tgt] + np.float32(0.05)
        flags[ikb] = 0x20
        ik["target"][ikb], ik["loop"][ikb], ik["angle"][ikb] = tgt, 10 + c, 1.0
        links = list(range(tgt - 1, tgt - 1 - n_links, -1))
        ik["link_bone"] += links
        ik["link_limited"] += [c % 2] * n_links
        ik["link_lo"] += [[-1.0, -0.5, 0.0]] * n_links
        ik["link_hi"] += [[1.0, 0.5, 0.0]] * n_links
        ik["link_off"][ikb + 1:] = len(ik["link_bone"])
    ik = dict(ik, link_bone=np.asarray(ik["link_bone"], np.int32), link_limited=np.asarray(ik["link_limited"], np.uint8),
              link_lo=np.asarray(ik["link_lo"], np.float32), link_hi=np.asarray(ik["link_hi"], np.float32))
    sk = vmd.Skeleton(rest, parent, None, flags, None, None, ik)
    assert sk.info["n_ik_bones"] == n_chains and sk.info["n_solve_rounds"] <= seg + 1   # the chains run side by side
    rng = np.random.RandomState(5)
    poses = np.zeros((40, nb, 8), np.float32)
    poses[..., 7] = 1
    poses[:, flags == 0x20, 0:3] = rng.uniform(-1, 1, (40, n_chains, 3))
    got = sk.solve(poses)
    for i in range(40):
        want = oracle.bone_solve_full(rest, parent, poses[i], None, flags, None, None, ik)
        gu.assert_bits_equal_or_both_nan(got[i], want, f"instance {i}")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_ik,n_app", [(11, 16, 20), (12, 4, 40), (13, 24, 0)])
def test_gpu_dense_rigs_with_levels_and_post_physics(oracle, seed, n_ik, n_app):
    """300-bone rigs dense with IK chains / append bones, three transform levels and 30 % post-physics bones (both
    evaluation lists populated, dependencies across them), 200 instances (the last workgroup partly empty)."""
    nb, ni = 300, 200
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(nb, seed, n_ik=n_ik, n_append=n_app, post_physics=0.3, levels=3)
    sk = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik)
    assert sk.info["n_post_physics"] > 0 and sk.info["n_solve_rounds"] < nb
    poses = random_poses(ni, nb, 700 + seed)
    got = sk.solve(poses)
    for i in range(ni):
        want = oracle.bone_solve_full(rest, parent, poses[i], level, flags, ap, ar, ik)
        gu.assert_bits_equal_or_both_nan(got[i], want, f"instance {i}")


# ---- nested IK: a link or target that is itself an IK bone (the reference's UpdateBoneTransform recurses) ---------
NESTED_CASES = [(40, 1), (60, 2), (90, 3), (120, 4)]


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
@pytest.mark.parametrize("nb,seed", NESTED_CASES)
def test_oracle_nested_ik_vs_reference(oracle, nb, seed):
    rest, parent, level, flags, ap, ar, ik = synth.make_nested_ik_rig(nb, seed)
    ref = Reference.skeleton(rest, parent, level, flags, ap, ar, ik)
    base = synth.make_ik_rig(nb, seed, n_ik=4, n_append=3, post_physics=0.0, levels=1)
    moved = 0
    for i, poses in enumerate(random_poses(4, nb, 900 + seed)):
        got = oracle.bone_solve_full(rest, parent, poses, level, flags, ap, ar, ik)
        gu.assert_bits_equal_or_both_nan(got, ref.solve(poses), f"palette {i}")
        moved += int(np.any(gu.bits(got) != gu.bits(oracle.bone_solve_full(base[0], base[1], poses, *base[2:]))))
    ref.close()
    assert moved == 4                                        # the inner solves really changed the result


def test_nested_ik_skeleton_create_host_side():
    rest, parent, level, flags, ap, ar, ik = synth.make_nested_ik_rig(60, 2)
    sk = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik)
    assert sk.info["solver"] == vmd.SOLVER_SERIAL and sk.info["n_ik_bones"] == 4
    sk.close()
    iks = [b for b in range(60) if flags[b] & 0x20]
    # an IK bone that is its own target / a two-bone cycle: the reference recurses without end -> rejected
    for a_, b_ in ((iks[0], iks[0]), (iks[3], iks[2])):
        bad = dict(ik, target=ik["target"].copy())
        bad["target"][a_] = b_
        if a_ != b_:
            bad["target"][b_] = a_
        with pytest.raises(Exception, match="own solve|recurses"):
            vmd.Skeleton(rest, parent, level, flags, ap, ar, bad)
    # nesting deeper than 3 solves: C -> B (target), and A already holds B as a link; chain D -> C -> B is depth 3 (ok),
    # E -> D -> C -> B is 4 (rejected).  Build by re-targeting.
    deep = dict(ik, target=ik["target"].copy())
    deep["target"][iks[3]] = iks[2]                         # D -> C -> B: 3 deep
    vmd.Skeleton(rest, parent, level, flags, ap, ar, deep).close()
    deep["target"][iks[0]] = iks[3]                         # A -> D -> C -> B: 4 deep
    with pytest.raises(Exception, match="nested more than"):
        vmd.Skeleton(rest, parent, level, flags, ap, ar, deep)


@pytest.mark.gpu
@pytest.mark.parametrize("nb,seed", NESTED_CASES)
def test_gpu_nested_ik_vs_oracle(oracle, nb, seed):
    rest, parent, level, flags, ap, ar, ik = synth.make_nested_ik_rig(nb, seed)
    poses = random_poses(40, nb, 950 + seed)
    got = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik).solve(poses)
    for i in range(poses.shape[0]):
        want = oracle.bone_solve_full(rest, parent, poses[i], level, flags, ap, ar, ik)
        gu.assert_bits_equal_or_both_nan(got[i], want, f"palette of instance {i}")


@pytest.mark.gpu
def test_gpu_nested_ik_three_deep(oracle):
    rest, parent, level, flags, ap, ar, ik = synth.make_nested_ik_rig(60, 2)
    iks = [b for b in range(60) if flags[b] & 0x20]
    deep = dict(ik, target=ik["target"].copy())
    deep["target"][iks[3]] = iks[2]                         # D -> C -> B
    poses = random_poses(20, 60, 77)
    got = vmd.Skeleton(rest, parent, level, flags, ap, ar, deep).solve(poses)
    for i in range(poses.shape[0]):
        want = oracle.bone_solve_full(rest, parent, poses[i], level, flags, ap, ar, deep)
        gu.assert_bits_equal_or_both_nan(got[i], want, f"palette of instance {i}")


# ---- bone tracks -> palettes in one call (mmdx_skeleton_solve_motion) ------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("nb,fwd,post,levels,seed", [(1, 0, 0.0, 1, 0), (40, 5, 0.25, 3, 2), (300, 20, 0.3, 4, 3), (2048, 100, 0.1, 5, 5),
                                                     (2500, 10, 0.0, 2, 6)])
def test_gpu_solve_motion_equals_eval_then_solve_and_oracle(oracle, nb, fwd, post, levels, seed):
    """mmdx_skeleton_solve_motion on parallel-FK skeletons (one launch, the poses in LDS; 2 500 bones: more than the LDS table holds,
    the two launches inside) is bit-identical to mmdx_bone_motion_eval + mmdx_skeleton_solve -- host and device operands -- and to
    the oracle run step by step."""
    rest, parent, level, flags = synth.make_skeleton(nb, seed, fwd, post, levels)
    names = [f"bone{i}" for i in range(nb)]
    keys = synth.make_bone_keys(names[:min(nb, 60)], 40 + seed, keys_per=5, span=300)
    v = vmd.Vmd(vmd.write_vmd(keys, []))
    bm = v.bind_bones(names)
    sk = vmd.Skeleton(rest, parent, level, flags)
    frames = np.r_[np.arange(0, 320, 37), 5, 5, 4_000_000_000].astype(np.uint32)
    ni = frames.size
    two = sk.solve(bm.eval(frames))
    one = sk.solve_motion(bm, frames)
    gu.assert_bits_equal(one, two, "host operands")
    d_fr, d_pal = DeviceBuffer.from_numpy(frames), DeviceBuffer(ni * nb * 64)
    d_pal.memset(0xFF)
    sk.solve_motion_device(bm, ni, d_fr.ptr, d_pal.ptr)
    api.check(api.lib().mmdx_device_synchronize())
    gu.assert_bits_equal(d_pal.download((ni, nb, 16), np.float32), two, "device operands")
    poses = oracle_poses(oracle, v, names, frames)
    for i in (0, ni - 1):
        gu.assert_bits_equal(one[i], oracle.bone_solve(rest, parent, poses[i], level, flags), f"oracle, instance {i}")
    d_fr.free(); d_pal.free()


@pytest.mark.gpu
def test_gpu_solve_motion_on_an_ik_rig_takes_the_ordered_solver():
    """A rig with append bones and CCD-IK: the call evaluates the tracks into the motion's scratch buffer and runs the ordered solver --
    the same bits as the two calls; a motion bound to another bone count is rejected."""
    nb = 48
    rig = synth.make_ik_rig(nb, 77, n_ik=3, n_append=4)
    names = [f"b{i}" for i in range(nb)]
    v = vmd.Vmd(vmd.write_vmd(synth.make_bone_keys(names, 9, keys_per=4, span=100), []))
    bm, sk = v.bind_bones(names), vmd.Skeleton(*rig)
    assert sk.info["solver"] == 1
    frames = np.arange(0, 110, 7, dtype=np.uint32)
    gu.assert_bits_equal(sk.solve_motion(bm, frames), sk.solve(bm.eval(frames)), "ordered solver")
    with pytest.raises(api.MmdxError, match="bones"):
        sk.solve_motion(v.bind_bones(names[:-1]), frames)
