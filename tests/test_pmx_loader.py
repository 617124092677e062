"""CPU: the from-scratch PMX 2.0 loader (csrc/pmx.cpp, through the C ABI) -- write -> read round trips
on synthetic models, a committed golden .pmx with what libmmd made of it, malformed files, and (where
oracle/_ref is built) the reference's own PmxReader on the same bytes."""
import os

import numpy as np
import pytest

from oracle.pyoracle import Reference, reference_available
from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import pmx, synth
from simple_mmd_renderer_amd.synth import BDEF2, BDEF4, MORPH_GROUP, MORPH_VERTEX
from tests import golden_util as gu


@pytest.fixture(autouse=True)
def _lib(hip_lib):
    return hip_lib


def rich_model(seed=3, nv=900, nb=40):
    """Vertex + group + bone + uv + material morphs, all four deform types."""
    m = synth.make_model(nv, nb, 4, 60, seed)
    t = list(m.morph_type) + [MORPH_GROUP, 2, 3, 8, 5]
    off = list(m.morph_off)
    idx = list(m.morph_index)
    val = [tuple(v) for v in m.morph_value]
    for entries in ([(0, (0.5, 0, 0)), (2, (2.0, 0, 0))], [(3, (0.1, 0.2, 0.3))], [(7, (0.5, 0.25, 0.0))],
                    [(1, (0, 0, 0))], [(9, (0.1, 0.1, 0.0))]):
        for i, v in entries:
            idx.append(i)
            val.append(v)
        off.append(len(idx))
    m.morph_type = np.asarray(t, np.int32)
    m.morph_off = np.asarray(off, np.uint32)
    m.morph_index = np.asarray(idx, np.uint32)
    m.morph_value = np.asarray(val, np.float32).reshape(-1, 3)
    return m


def assert_same_model(a, b):
    for k in ("positions", "normals", "uvs", "skin_type", "bone_pos", "bone_parent", "morph_type", "morph_off",
              "morph_index", "morph_value"):
        assert np.array_equal(np.asarray(getattr(a, k)), np.asarray(getattr(b, k))), k
    t = np.asarray(a.skin_type)
    n_ids = np.where(t == 0, 1, np.where(t == BDEF4, 4, 2))
    for k in range(4):
        use = n_ids > k
        assert np.array_equal(a.bone_ids[use, k], b.bone_ids[use, k])
    assert np.array_equal(a.bone_weights[t == BDEF4], b.bone_weights[t == BDEF4])
    assert np.array_equal(a.bone_weights[(t == BDEF2) | (t == 3), 0], b.bone_weights[(t == BDEF2) | (t == 3), 0])
    assert np.array_equal(a.sdef[t == 3], b.sdef[t == 3])


@pytest.mark.parametrize("opt", [
    pmx.PmxWriteOptions(),
    pmx.PmxWriteOptions(utf8=True, extra_uv=2),
    pmx.PmxWriteOptions(index_width=(4, 4, 4, 4, 4, 4), extra_uv=4, display_frames=0, rigid_bodies=0),
    pmx.PmxWriteOptions(index_width=(2, 2, 2, 2, 2, 2), n_textures=0, n_materials=1, bone_flag_variety=False),
])
def test_write_read_round_trip(opt):
    m = rich_model()
    data = pmx.write_pmx(m, opt)
    pm = pmx.load_pmx(data)
    assert_same_model(m, pm.flat)
    assert pm.info["n_vertices"] == m.nv and pm.info["n_bones"] == m.nb and pm.info["n_morphs"] == m.nm
    assert pm.info["utf8"] == int(opt.utf8) and pm.info["extra_uv"] == opt.extra_uv
    assert pm.name == opt.model_name and pm.bone_names[1] == "ボーン1" and pm.morph_names[0] == "モーフ0"
    assert pm.triangles.size == (m.nv // 3) * 3 and pm.material_index_count.sum() == pm.triangles.size
    assert pm.info["bytes_consumed"] <= len(data)


def test_load_from_file_and_index_extension(tmp_path):
    """1-byte bone indices are ZERO-extended like libmmd does (dwarf_impl.inl:90-95): PMX's -1 = 'no bone'
    arrives as 255; mmdx_model_create then accepts it only where its weight makes it irrelevant."""
    m = synth.make_model(300, 20, 2, 30, seed=8)
    m.skin_type[:] = BDEF4
    m.bone_weights[:] = np.array([0.75, 0.25, 0.0, 0.0], np.float32)
    m.bone_ids[:, 2:] = -1
    p = tmp_path / "m.pmx"
    p.write_bytes(pmx.write_pmx(m, pmx.PmxWriteOptions(index_width=(0, 1, 1, 1, 1, 1))))
    pm = pmx.load_pmx(str(p))
    assert (pm.flat.bone_ids[:, 2:] == 255).all()
    from simple_mmd_renderer_amd.engine import DeformModel
    dm = DeformModel(pm.flat, host_only=True)
    _, ids, _ = dm.get_skin()
    assert (ids < m.nb).all()
    dm.close()
    # 4-byte indices are sign-extended: -1 stays -1
    pm4 = pmx.load_pmx(pmx.write_pmx(m, pmx.PmxWriteOptions(index_width=(4, 4, 4, 4, 4, 4))))
    assert (pm4.flat.bone_ids[:, 2:] == -1).all()


def test_malformed_files_are_rejected_not_crashed():
    m = rich_model(nv=120, nb=9)
    good = pmx.write_pmx(m)

    def fails(data, needle=None):
        with pytest.raises(api.MmdxError) as e:
            pmx.load_pmx(bytes(data))
        assert e.value.status == 1
        if needle:
            assert needle in str(e.value)

    fails(b"PMD " + good[4:], "not a PMX 2.0")
    fails(good[:4] + np.float32(2.1).tobytes() + good[8:], "not a PMX 2.0")
    for cut in (0, 3, 9, 17, 40, 200, len(good) // 3, len(good) // 2):
        fails(good[:cut], "file ends")
    pm = pmx.load_pmx(good)
    fails(good[:pm.info["bytes_consumed"] - 1], "file ends")           # last morph byte missing
    pmx.load_pmx(good[:pm.info["bytes_consumed"]])                      # sections behind the morphs are optional for us
    # deform type 4 (QDEF is PMX 2.1) in the first vertex: walk the 4 header texts to find it
    at = 17
    for _ in range(4):
        at += 4 + int(np.frombuffer(good[at:at + 4], "<i4")[0])
    assert int(np.frombuffer(good[at:at + 4], "<i4")[0]) == m.nv
    bad = bytearray(good)
    bad[at + 4 + 32] = 4
    fails(bad, "deform type 4")
    rng = np.random.RandomState(0)
    for _ in range(200):                                                 # byte fuzz: error or success, never a crash
        b = bytearray(good)
        for k in rng.randint(0, len(b), rng.randint(1, 6)):
            b[k] = rng.randint(0, 256)
        try:
            pmx.load_pmx(bytes(b))
        except api.MmdxError:
            pass


def test_golden_pmx_file(oracle):
    """tests/golden/pmx_small.pmx + what libmmd (PmxReader -> Normalize -> Poser) produced from it."""
    z = np.load(os.path.join(gu.GOLDEN_DIR, "pmx_small_expect.npz"))
    pm = pmx.load_pmx(os.path.join(gu.GOLDEN_DIR, "pmx_small.pmx"))
    skin = oracle.normalize(pm.flat)
    assert np.array_equal(skin[0], z["norm_type"])
    for f in range(z["rates"].shape[0]):
        pos, nrm = oracle.skin(pm.flat, z["palette"][f], oracle.morph(pm.flat, z["rates"][f]), skin)
        gu.assert_bits_equal(pos, z["expect_pos"][f], "pos")
        gu.assert_bits_equal(nrm, z["expect_nrm"][f], "nrm")
        gu.assert_bits_equal(oracle.repack32(pm.flat, pos, nrm, 0.1), z["expect_v32"][f], "v32")


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
@pytest.mark.parametrize("seed,opt", [(1, pmx.PmxWriteOptions()),
                                      (2, pmx.PmxWriteOptions(utf8=True, extra_uv=4, index_width=(4, 2, 2, 2, 2, 2))),
                                      (3, pmx.PmxWriteOptions(index_width=(2, 1, 1, 4, 4, 1), display_frames=0))])
def test_reference_reader_agrees_on_the_same_bytes(oracle, tmp_path, seed, opt):
    # additional UV sets: 0 or 4 only -- libmmd's reader crashes on 1..3 (fall-through switch in
    # Vertex::SetExtraUVCoordinate, L/model/model_vertex_impl.inl:105-116, writes through null proxies);
    # this repo's loader accepts 0..4 (test_write_read_round_trip)
    m = rich_model(seed=seed, nv=1500, nb=60)
    m.bone_weights[::7, 0] = 0.0
    m.bone_weights[3::11, 0] = 1.0
    p = tmp_path / "m.pmx"
    p.write_bytes(pmx.write_pmx(m, opt))
    ref = Reference.from_pmx(str(p))
    pm = pmx.load_pmx(str(p))
    assert (ref.model.nv, ref.model.nb, ref.model.nm) == (pm.flat.nv, pm.flat.nb, pm.flat.nm)
    assert ref.model.ntri * 3 == pm.triangles.size
    rt, rids, rw = ref.get_skin()
    skin = oracle.normalize(pm.flat)
    assert np.array_equal(rt, skin[0])
    for frame in (0, 25):
        rates = synth.morph_weights(pm.flat.nm, frame)[0]
        pal = synth.make_palettes(pm.flat, [frame])[0]
        rp, rn, _ = ref.run(rates, pal)          # group/bone morphs move libmmd's own palette: inject ours after
        op, on = oracle.skin(pm.flat, pal, oracle.morph(pm.flat, rates), skin)
        gu.assert_bits_equal(op, rp, "pos")
        gu.assert_bits_equal(on, rn, "nrm")
        gu.assert_bits_equal(oracle.repack32(pm.flat, op, on, 0.1), ref.repack32(0.1), "v32")
    ref.close()


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
def test_reference_reader_rejects_what_we_reject(tmp_path):
    m = rich_model(nv=90, nb=5)
    good = pmx.write_pmx(m)
    for i, data in enumerate([b"XXXX" + good[4:], good[:len(good) // 2]]):
        p = tmp_path / f"bad{i}.pmx"
        p.write_bytes(data)
        with pytest.raises(RuntimeError):
            Reference.from_pmx(str(p))
        with pytest.raises(api.MmdxError):
            pmx.load_pmx(str(p))
