"""CPU: the VMD loader (csrc/vmd.cpp) and the restatement of Motion::GetMorphPose -- written fixtures,
a committed golden .vmd with libmmd's answers, malformed input, and (where oracle/_ref is built) the
reference's own VmdReader + Motion on the same bytes."""
import os

import numpy as np
import pytest

from oracle.pyoracle import Reference, ReferenceMotion, reference_available
from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import pmx, synth, vmd
from tests import golden_util as gu

MORPH_NAMES = ["あ", "にこり", "まばたき", "ウィンク右", "MorphEN", "使われない"]


@pytest.fixture(autouse=True)
def _lib(hip_lib):
    return hip_lib


def make_motion(seed=0, names=MORPH_NAMES[:5], keys_per=7, span=300):
    rng = np.random.RandomState(seed)
    mk = []
    for n in names:
        for f in sorted(rng.choice(span, keys_per, replace=False)):
            mk.append((n, int(f), float(np.float32(rng.uniform(-0.2, 1.2)))))
    mk.append((names[0], mk[0][1], 0.875))                     # same (name, frame) again: later wins
    mk.append((names[1], 2 ** 24 + 7, 0.5))                    # frame numbers beyond float's exact range
    rng.shuffle(mk)
    bk = [("センター", 0, (0, 0, 0), (0, 0, 0, 1), None), ("センター", 30, (1, 2, 3), (0, 0.7071, 0, 0.7071), None),
          ("左足ＩＫ", 12, (0.5, 0, 0), (0, 0, 0, 1), bytes(range(64)))]
    return bk, mk


def keys_for(v, model_names):
    """(key_off, frames, weights) in model-morph order, as mmdx_vmd_bind_morphs builds them."""
    tracks = {n: v.morph_track(i) for i, n in enumerate(v.morph_track_names)}
    off, fr, w = [0], [], []
    for n in model_names:
        if n in tracks:
            fr += list(tracks[n][0])
            w += list(tracks[n][1])
        off.append(len(fr))
    return np.asarray(off, np.uint32), np.asarray(fr, np.uint32), np.asarray(w, np.float32)


def test_write_parse_round_trip():
    bk, mk = make_motion()
    v = vmd.Vmd(vmd.write_vmd(bk, mk))
    assert v.info["n_morph_records"] == len(mk) and v.info["n_bone_records"] == 3
    assert sorted(v.morph_track_names) == sorted(MORPH_NAMES[:5]) and v.bone_track_names == ["センター", "左足ＩＫ"]
    want = {}
    for n, f, w in mk:
        want[(n, f)] = np.float32(w)                           # later record wins
    got = {}
    for i, n in enumerate(v.morph_track_names):
        fr, w = v.morph_track(i)
        assert (np.diff(fr.astype(np.int64)) > 0).all()
        for f, x in zip(fr, w):
            got[(n, int(f))] = x
    assert got == want and v.info["n_morph_keys"] == len(want)
    k = v.bone_track(1)[0]
    assert k["frame"] == 12 and k["translation"] == (0.5, 0.0, 0.0) and k["interpolation"] == bytes(range(64))
    assert v.info["max_frame"] == 2 ** 24 + 7


def test_binding_by_name_and_unmapped_morphs():
    bk, mk = make_motion()
    v = vmd.Vmd(vmd.write_vmd(bk, mk))
    mm = v.bind_morphs(MORPH_NAMES + ["", "あ"])               # unknown, empty and repeated names
    assert (mm.nm, mm.n_mapped) == (8, 6)                      # 5 tracks + the repeated "あ"
    mm.close()


def test_malformed_vmd():
    bk, mk = make_motion()
    good = vmd.write_vmd(bk, mk)
    for bad in (b"", good[:40], b"Vocaloid Motion Data file" + good[25:], good[:54 + 50], good[:-40]):
        with pytest.raises(api.MmdxError) as e:
            vmd.Vmd(bad)
        assert e.value.status == 1
    v = vmd.Vmd(good[:vmd.Vmd(good).info["bytes_consumed"]])    # camera/light/shadow sections are optional
    assert v.info["n_morph_records"] == len(mk)
    rng = np.random.RandomState(0)
    for _ in range(200):                                         # byte fuzz: error or success, never a crash
        b = bytearray(good)
        for k in rng.randint(0, len(b), rng.randint(1, 6)):
            b[k] = rng.randint(0, 256)
        try:
            vmd.Vmd(bytes(b)).close()
        except api.MmdxError:
            pass


def test_golden_vmd_rates(oracle):
    """tests/golden/vmd_small.vmd + Motion::GetMorphPose answers recorded from libmmd."""
    z = np.load(os.path.join(gu.GOLDEN_DIR, "vmd_small_expect.npz"))
    v = vmd.Vmd(os.path.join(gu.GOLDEN_DIR, "vmd_small.vmd"))
    names = [str(n) for n in z["model_morph_names"]]
    off, fr, w = keys_for(v, names)
    got = oracle.morph_tracks(off, fr, w, z["frames"])
    gu.assert_bits_equal(got, z["expect_rates"], "rates")


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_reference_motion_agrees_on_the_same_bytes(oracle, tmp_path, seed):
    bk, mk = make_motion(seed)
    p = tmp_path / "m.vmd"
    p.write_bytes(vmd.write_vmd(bk, mk))
    v = vmd.Vmd(str(p))
    rm = ReferenceMotion(str(p))
    at = np.r_[np.arange(0, 320), 2 ** 24 + np.arange(0, 12), 4_000_000_000].astype(np.uint32)
    names = v.morph_track_names
    off, fr, w = keys_for(v, names)
    got = oracle.morph_tracks(off, fr, w, at)
    for j, n in enumerate(names):
        want = np.array([rm.morph_weight(n.encode("shift_jis"), int(f)) for f in at], np.float32)
        gu.assert_bits_equal(got[:, j], want, f"track {n}")
    assert np.isnan(rm.morph_weight("無い".encode("shift_jis"), 3))
    rm.close()


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref/libmmd_ref.so not built")
def test_reference_name_mapping_defect_on_linux(tmp_path):
    """Documented reference defect: libmmd decodes VMD names with iconv 'UTF-16', which prepends a
    byte-order mark, so on Linux no VMD track name equals a PMX name and MotionPlayer maps nothing
    (L/util/dwarf_impl.inl:221-230).  This loader matches names by their decoded text."""
    m = synth.make_model(90, 5, 3, 10, seed=4)
    names = MORPH_NAMES[:3]
    (tmp_path / "m.pmx").write_bytes(pmx.write_pmx(m, pmx.PmxWriteOptions(morph_names=names, bone_flag_variety=False)))
    (tmp_path / "m.vmd").write_bytes(vmd.write_vmd([], [(n, 5, 0.5) for n in names]))
    ref = Reference.from_pmx(str(tmp_path / "m.pmx"))
    rm = ReferenceMotion(str(tmp_path / "m.vmd"))
    assert rm.names_match_model(ref) == 0                       # libmmd: nothing matches
    mm = vmd.Vmd(str(tmp_path / "m.vmd")).bind_morphs(pmx.load_pmx(str(tmp_path / "m.pmx")).morph_names)
    assert mm.n_mapped == 3                                     # intended behaviour
    ref.close(); rm.close(); mm.close()
