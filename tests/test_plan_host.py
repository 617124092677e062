"""CPU: host logic behind the C ABI (model validation, Model::Normalize retagging, group-morph
flattening) through MMDX_CREATE_HOST_ONLY handles -- no GPU, no compute."""
import numpy as np
import pytest

from simple_mmd_renderer_amd import _capi, synth
from simple_mmd_renderer_amd.engine import DeformModel
from simple_mmd_renderer_amd.synth import BDEF1, BDEF2, BDEF4, MORPH_GROUP, MORPH_VERTEX
from tests import golden_util as gu


@pytest.fixture(autouse=True)
def _lib(hip_lib):
    return hip_lib


def class_of(t):
    t = np.asarray(t)
    return np.where(t == BDEF1, 0, np.where(t == BDEF4, 2, 1))


@pytest.mark.parametrize("name", gu.fixture_names())
def test_normalize_matches_reference_tags(name):
    m, exp = gu.load(name)
    dm = DeformModel(m, normalize=exp["normalize"], host_only=True)
    t, ids, w = dm.get_skin()
    assert np.array_equal(class_of(t), class_of(exp["norm_type"]))
    c = class_of(t)
    nid = exp["norm_ids"]
    assert np.array_equal(ids[c == 0, 0], nid[c == 0, 0])
    assert np.array_equal(ids[c == 1, :2], nid[c == 1, :2])
    assert np.array_equal(ids[c == 2], nid[c == 2])
    assert np.array_equal(w[c == 1, 0].view(np.uint32), exp["norm_w"][c == 1, 0].view(np.uint32))
    info = dm.info
    assert info.n_bdef1 + info.n_bdef2 + info.n_bdef4 == m.nv
    assert info.n_tiles == (m.nv + info.tile_vertices - 1) // info.tile_vertices
    dm.close()


def py_slot_weights(m, rates):
    """UpdateMorphTransform's recursion (poser_impl.inl:328-339) restated in Python: the list of
    applied (vertex morph, rate) pairs in traversal order; skipped applications get rate 0."""
    out = []

    def visit(i, r, skipped):
        skip = skipped or (float(np.float32(r)) < 1e-7)
        if m.morph_type[i] == MORPH_GROUP:
            for j in range(m.morph_off[i], m.morph_off[i + 1]):
                visit(int(m.morph_index[j]), np.float32(m.morph_value[j, 0]) * np.float32(r), skip)
        elif m.morph_type[i] == MORPH_VERTEX:
            out.append(np.float32(0.0) if skip else np.float32(r))

    for i in range(m.nm):
        visit(i, rates[i], False)
    return np.asarray(out, np.float32)


def test_group_flattening_matches_reference_recursion():
    m, exp = gu.load("g08_group_morph")
    dm = DeformModel(m, host_only=True)
    assert dm.ns == py_slot_weights(m, np.ones(m.nm, np.float32)).size
    for r in exp["rates"]:
        got = dm.slot_weights(r)
        want = py_slot_weights(m, r)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    dm.close()


def test_slot_weights_reproduce_oracle_vertex_images(oracle):
    """slot weights x CSR entries == the oracle's morph pass, checked by applying the slots in order
    on the host (the same sum the device gather performs)."""
    m, exp = gu.load("g08_group_morph")
    dm = DeformModel(m, host_only=True)
    # slot -> vertex morph, in traversal order
    order = []

    def visit(i):
        if m.morph_type[i] == MORPH_GROUP:
            for j in range(m.morph_off[i], m.morph_off[i + 1]):
                visit(int(m.morph_index[j]))
        elif m.morph_type[i] == MORPH_VERTEX:
            order.append(i)

    for i in range(m.nm):
        visit(i)
    for r in exp["rates"]:
        w = dm.slot_weights(r)
        vimg = np.zeros((m.nv, 3), np.float32)
        for s, vm in enumerate(order):
            if not (w[s] < np.float32(1e-7)):
                for j in range(m.morph_off[vm], m.morph_off[vm + 1]):
                    v = m.morph_index[j]
                    vimg[v] = vimg[v] + m.morph_value[j] * w[s]
        gu.assert_bits_equal(vimg, oracle.morph(m, r), "vertex images")
    dm.close()


def test_validation_errors():
    base = synth.make_model(100, 8, 3, 10, 9)
    m = base.copy()
    m.bone_ids[5, 0] = 99
    m.skin_type[5] = BDEF1
    with pytest.raises(_capi.MmdxError) as e:
        DeformModel(m, host_only=True)
    assert e.value.status == 2 and "vertex 5" in str(e.value)

    m = base.copy()
    m.morph_index[3] = 100000
    with pytest.raises(_capi.MmdxError) as e:
        DeformModel(m, host_only=True)
    assert e.value.status == 2

    m = base.copy()   # group cycle 0 -> 1 -> 0
    m.morph_type = np.array([MORPH_GROUP, MORPH_GROUP], np.int32)
    m.morph_off = np.array([0, 1, 2], np.uint32)
    m.morph_index = np.array([1, 0], np.uint32)
    m.morph_value = np.ones((2, 3), np.float32)
    with pytest.raises(_capi.MmdxError) as e:
        DeformModel(m, host_only=True)
    assert e.value.status == 6


def test_pmx_no_bone_ids_with_zero_weight_are_accepted():
    m = synth.make_model(64, 8, 1, 4, 11)
    m.skin_type[:] = BDEF4
    m.bone_weights[:] = np.array([0.5, 0.5, 0.0, 0.0], np.float32)
    m.bone_ids[:, 2] = -1       # PMX "none"
    m.bone_ids[:, 3] = 65535    # the same after libmmd's zero extension of 2-byte indices
    dm = DeformModel(m, host_only=True)
    t, ids, w = dm.get_skin()
    assert (ids >= 0).all() and (ids < m.nb).all()
    dm.close()
    m.bone_weights[7, 3] = 0.25
    with pytest.raises(_capi.MmdxError):
        DeformModel(m, host_only=True)
    m2 = synth.make_model(64, 8, 1, 4, 11)
    m2.skin_type[:] = BDEF2
    m2.bone_weights[:, 0] = 1.0
    m2.bone_ids[:, 1] = 255
    DeformModel(m2, normalize=True, host_only=True).close()    # retagged BDEF1(id0)
    DeformModel(m2, normalize=False, host_only=True).close()   # Lerp short-circuits to S[b0]


def test_host_only_model_cannot_deform():
    m = synth.make_model(64, 4, 1, 4, 3)
    dm = DeformModel(m, host_only=True)
    with pytest.raises(_capi.MmdxError) as e:
        dm.deform(np.zeros(m.nm, np.float32), synth.make_palettes(m, [0])[0])
    assert e.value.status == 3 and "no CPU fallback" in str(e.value)
    dm.close()


def test_large_config_plans_build():
    m = synth.make_config("config2_50k")
    dm = DeformModel(m, host_only=True)
    i = dm.info
    assert (i.n_vertices, i.n_bones, i.n_morphs, i.n_slots) == (50000, 300, 200, 200)
    assert i.n_entries == 200 * 2048
    assert i.max_tile_bones <= 64     # 16-wide bone window + drift across a 512-vertex tile
    dm.close()


def test_presort_by_class_is_a_vertex_permutation(oracle):
    """synth.presort_by_class (the "bucketed" bench variant) only reorders vertices: the oracle's deformed vertices
    of the reordered model are the original's, permuted."""
    m = synth.make_model(2500, 40, 8, 200, seed=5)
    s = synth.presort_by_class(m)
    assert s.nv == m.nv and s.meta.get("presorted")
    rates = synth.morph_weights(m.nm, 11)[0]
    pal = synth.make_palettes(m, [3])[0]
    pa, na = oracle.skin(m, pal, oracle.morph(m, rates), oracle.normalize(m))
    pb, nb = oracle.skin(s, pal, oracle.morph(s, rates), oracle.normalize(s))
    key = lambda p, n: np.sort(np.concatenate([p, n], 1).view(np.uint32).view([("", np.uint32)] * 6).ravel())
    assert np.array_equal(key(pa, na), key(pb, nb))
    st = np.asarray(s.skin_type)[:512]
    cls = np.where(st == synth.BDEF1, 0, np.where(st == synth.BDEF4, 2, 1))
    assert (np.diff(cls) >= 0).all()                      # first tile: classes in order
