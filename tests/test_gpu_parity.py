"""GPU (MI355X): the HIP path, called through the C ABI, against (a) the golden vectors produced by
the real libmmd and (b) the C restatement on seeded inputs.  Bit-exact: integer/index work AND the
f32 arithmetic (kernels are built with -ffp-contract=off and keep the reference's operation order),
so no tolerance is needed; the fp16 bandwidth variant is compared against the f32 oracle on
f16-quantised inputs with its positions rounded to f16 once -- also bit-exact."""
import os

import numpy as np
import pytest

from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import synth
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer, Poser, device_count
from simple_mmd_renderer_amd.synth import BDEF1, BDEF2, BDEF4, MORPH_GROUP, MORPH_VERTEX
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(hip_lib):
    assert device_count() >= 1, "no HIP device visible: the GPU tests must run on the MI355X box"


def oracle_expect(oracle, m, rates, pal, normalize=True):
    skin = oracle.normalize(m) if normalize else None
    pos, nrm = oracle.skin(m, pal, oracle.morph(m, rates), skin)
    return pos, nrm


# ---- golden vectors (reference's own outputs) ------------------------------------------------------
@pytest.mark.parametrize("name", gu.fixture_names())
def test_golden_single_deform(name):
    m, exp = gu.load(name)
    with DeformModel(m, normalize=exp["normalize"]) as dm:
        for f in range(exp["rates"].shape[0]):
            pos, nrm = dm.deform(exp["rates"][f], exp["palette"][f])
            gu.assert_bits_equal(pos, exp["expect_pos"][f], f"{name}[{f}] pos")
            gu.assert_bits_equal(nrm, exp["expect_nrm"][f], f"{name}[{f}] nrm")
            v32 = dm.deform_vertex32(exp["rates"][f], exp["palette"][f], 0.1)
            gu.assert_bits_equal(v32, exp["expect_v32"][f], f"{name}[{f}] vertex32")


@pytest.mark.parametrize("name", ["g07_vertex_morph", "g08_group_morph", "g11_solved_palette",
                                  "g12_mini_model", "g14_denormals"])
@pytest.mark.parametrize("layout", [api.OUT_SOA, api.OUT_VERTEX32])
def test_golden_batched_frames_as_instances(name, layout):
    """Frames are independent given palettes + rates: run all frames of a fixture as one batched call
    with per-instance morph weights (fused morph gather, 4 instances per CSR pass)."""
    m, exp = gu.load(name)
    nf = exp["rates"].shape[0]
    reps = 3                                    # > 4 instances, ragged last quad
    rates = np.tile(exp["rates"], (reps, 1))
    pals = np.tile(exp["palette"], (reps, 1, 1))
    with DeformModel(m, normalize=exp["normalize"]) as dm:
        out = dm.deform_batched(rates, pals, layout=layout, pos_scale=0.1 if layout == api.OUT_VERTEX32 else 1.0)
        for i in range(nf * reps):
            f = i % nf
            if layout == api.OUT_SOA:
                gu.assert_bits_equal(out[0][i], exp["expect_pos"][f], f"{name} inst {i} pos")
                gu.assert_bits_equal(out[1][i], exp["expect_nrm"][f], f"{name} inst {i} nrm")
            else:
                gu.assert_bits_equal(out[i], exp["expect_v32"][f], f"{name} inst {i} v32")


def test_golden_shared_weights_crowd():
    """Crowd form: one morph state for all instances (separate morph pass), per-instance palettes."""
    m, exp = gu.load("g12_mini_model")
    nf = exp["rates"].shape[0]
    with DeformModel(m, normalize=True) as dm:
        for f in range(nf):
            # every frame's palette as an instance, all with frame f's morph weights
            pos, nrm = dm.deform_batched(exp["rates"][f], exp["palette"], shared_weights=True)
            gu.assert_bits_equal(pos[f], exp["expect_pos"][f], "pos")
            gu.assert_bits_equal(nrm[f], exp["expect_nrm"][f], "nrm")


def test_config1_600_frames_checksums():
    """configs[0] through the GPU path: per-frame checksums recorded from libmmd."""
    import os
    z = np.load(os.path.join(gu.GOLDEN_DIR, "g13_config1_checksums.npz"))
    m = synth.make_config("config1_20k")
    frames = z["frames"]
    rates = synth.morph_weights(m.nm, frames)
    pals = synth.make_palettes(m, frames)
    with DeformModel(m) as dm:
        for b in range(0, 600, 100):            # 100 frames per batched call
            sl = slice(b, b + 100)
            pos, nrm = dm.deform_batched(rates[sl], pals[sl])
            v32 = dm.deform_batched(rates[sl], pals[sl], layout=api.OUT_VERTEX32, pos_scale=0.1)
            for i in range(100):
                got = (synth.checksum64(pos[i]), synth.checksum64(nrm[i]), synth.checksum64(v32[i]))
                assert got == tuple(int(x) for x in z["checksums"][b + i]), f"frame {b + i}"


# ---- seeded inputs vs the C restatement ------------------------------------------------------------
@pytest.mark.parametrize("nv", [1, 2, 63, 64, 255, 256, 257, 511, 512, 513, 1000, 1025, 4099])
def test_ragged_sizes_and_unaligned_rows(oracle, nv):
    """Every tail shape of the 512-vertex tile, odd NV (output rows of instance i start at i*NV*12
    bytes: exercises the unaligned head/tail of the staged copy-out)."""
    nb = 1 if nv == 1 else 17
    m = synth.make_model(nv, nb, 3, min(nv, 40), seed=1000 + nv)
    ni = 5
    rates = synth.morph_weights(m.nm, np.arange(ni) * 7)
    pals = synth.make_palettes(m, np.arange(ni) * 3)
    with DeformModel(m) as dm:
        pos, nrm = dm.deform_batched(rates, pals)
        v32 = dm.deform_batched(rates, pals, layout=api.OUT_VERTEX32, pos_scale=0.1)
        spos, snrm = dm.deform_batched(rates[2], pals, shared_weights=True)
        for i in range(ni):
            ep, en = oracle_expect(oracle, m, rates[i], pals[i])
            gu.assert_bits_equal(pos[i], ep, f"nv={nv} inst {i} pos")
            gu.assert_bits_equal(nrm[i], en, f"nv={nv} inst {i} nrm")
            gu.assert_bits_equal(v32[i], oracle.repack32(m, ep, en, 0.1), f"nv={nv} inst {i} v32")
            sp, sn = oracle_expect(oracle, m, rates[2], pals[i])
            gu.assert_bits_equal(spos[i], sp, f"nv={nv} shared inst {i} pos")
            gu.assert_bits_equal(snrm[i], sn, f"nv={nv} shared inst {i} nrm")


@pytest.mark.parametrize("mix", [(1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1),
                                 (0.02, 0.02, 0.95, 0.01), (0.25, 0.25, 0.25, 0.25)])
def test_class_mixes(oracle, mix):
    m = synth.make_model(3000, 90, 6, 300, seed=55, mix=mix)
    rates = synth.morph_weights(m.nm, [4, 50])
    pals = synth.make_palettes(m, [4, 50])
    with DeformModel(m) as dm:
        pos, nrm = dm.deform_batched(rates, pals)
        for i in range(2):
            ep, en = oracle_expect(oracle, m, rates[i], pals[i])
            gu.assert_bits_equal(pos[i], ep, "pos")
            gu.assert_bits_equal(nrm[i], en, "nrm")


def test_many_bones_per_tile(oracle):
    """Bone ids spread over the whole skeleton: every tile needs (nearly) all 512 bones in LDS."""
    m = synth.make_model(2048, 512, 4, 100, seed=77, window=512)
    rates = synth.morph_weights(m.nm, [1, 2, 3])
    pals = synth.make_palettes(m, [1, 2, 3])
    with DeformModel(m) as dm:
        assert dm.info.max_tile_bones > 400
        pos, nrm = dm.deform_batched(rates, pals)
        for i in range(3):
            ep, en = oracle_expect(oracle, m, rates[i], pals[i])
            gu.assert_bits_equal(pos[i], ep, "pos")
            gu.assert_bits_equal(nrm[i], en, "nrm")


def test_no_morph_model_and_empty_rows(oracle):
    m = synth.make_model(700, 12, 1, 1, seed=5)
    m.morph_type = np.zeros(0, np.int32)
    m.morph_off = np.zeros(1, np.uint32)
    m.morph_index = np.zeros(0, np.uint32)
    m.morph_value = np.zeros((0, 3), np.float32)
    m.positions[::7] = np.float32(-0.0)
    pal = synth.make_palettes(m, [9])[0]
    with DeformModel(m) as dm:
        pos, nrm = dm.deform(np.zeros(0, np.float32), pal)
        ep, en = oracle_expect(oracle, m, np.zeros(0, np.float32), pal)
        gu.assert_bits_equal(pos, ep, "pos")
        gu.assert_bits_equal(nrm, en, "nrm")


def test_nonfinite_morph_offsets_keep_skip_semantics(oracle):
    """inf / NaN morph offsets switch the kernels to the predicated skip (a skipped morph must not
    turn inf*0 into NaN); finite-offset models take the branch-free path.  Both against the oracle."""
    m = synth.make_model(2000, 30, 6, 150, seed=404)
    m.morph_value[m.morph_off[2]:m.morph_off[3], 0] = np.inf        # morph 2: +inf x offsets
    m.morph_value[m.morph_off[4] + 3, 1] = np.nan                    # morph 4: one NaN
    rates = np.array([[0.3, 0.0, 0.0, 0.7, 5e-8, 1.0],               # 2 and 4 skipped -> finite result
                      [0.3, 0.2, 0.5, 0.0, 0.0, 0.4],                # morph 2 applied -> +-inf appears
                      [0.0, 0.0, 0.0, 0.0, 1.0, 0.0],                # morph 4 applied -> NaN appears
                      [1.0, 1.0, 0.0, 1.0, 0.0, 1.0],
                      [0.1, 0.0, 0.0, 0.0, 0.0, 0.0]], np.float32)
    pals = synth.make_palettes(m, np.arange(5))

    def same(a, b):
        a, b = np.asarray(a), np.asarray(b)
        return np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))

    with DeformModel(m) as dm:
        pos, nrm = dm.deform_batched(rates, pals)                    # fused, 4 instances per pass
        for i in range(5):
            ep, en = oracle_expect(oracle, m, rates[i], pals[i])
            assert same(pos[i], ep) and same(nrm[i], en), f"instance {i}"
            sp, sn = dm.deform(rates[i], pals[i])                    # fused, single instance
            assert same(sp, ep) and same(sn, en)
            cp, cn = dm.deform_batched(rates[i], pals, shared_weights=True)   # shared morph pass
            assert same(cp[i], ep) and same(cn[i], en)
        assert np.isfinite(pos[0]).all() and not np.isfinite(pos[1]).all() and np.isnan(pos[2]).any()


def _random_model(rng):
    nv = int(rng.choice([1, 3, 65, 300, 777, 1500, 2600, 5200]))
    nb = int(rng.choice([1, 2, 9, 40, 130, 600]))
    mix = rng.dirichlet([0.6, 1.0, 0.8, 0.3])
    m = synth.make_model(nv, nb, 1, 1, seed=int(rng.randint(1 << 30)), mix=tuple(mix),
                         window=int(rng.choice([1, 4, 16, 64, 1024])))
    # random morph table: vertex morphs (with duplicates), groups (depth <= 2), ignored types
    n_vm = int(rng.randint(0, 7))
    types, off, idx, val = [], [0], [], []
    for _ in range(n_vm):
        k = int(rng.randint(0, min(nv, 60) + 1))
        types.append(MORPH_VERTEX)
        idx += list(rng.randint(0, nv, k))
        val += list(rng.uniform(-0.5, 0.5, (k, 3)))
        off.append(len(idx))
    n_extra = int(rng.randint(0, 4)) if n_vm else 0
    for e in range(n_extra):
        kind = rng.choice(["group", "bone", "uv"])
        if kind == "group":
            members = rng.randint(0, len(types), int(rng.randint(1, 4)))   # may reference earlier groups
            types.append(MORPH_GROUP)
            for mm in members:
                idx.append(int(mm))
                val.append((float(rng.choice([0.5, 1.0, 2.0, 1e-4, -1.0])), 0.0, 0.0))
        else:
            types.append(2 if kind == "bone" else 3)
            idx.append(0)
            val.append((0.1, 0.2, 0.3))
        off.append(len(idx))
    m.morph_type = np.asarray(types, np.int32)
    m.morph_off = np.asarray(off, np.uint32)
    m.morph_index = np.asarray(idx, np.uint32).reshape(-1)
    m.morph_value = np.asarray(val, np.float32).reshape(-1, 3)
    m.bone_weights[rng.randint(0, nv, max(1, nv // 8)), 0] = rng.choice([0.0, 1.0, 5e-8, 0.9999999])
    return m


# MMDX_SOAK_SEEDS=N widens the sweep (tools/ and the round's soak runs); the default keeps the suite short
@pytest.mark.parametrize("seed", range(int(os.environ.get("MMDX_SOAK_SEEDS", "24"))))
def test_randomized_models_all_call_forms(oracle, seed):
    """Random sizes / class mixes / bone windows / morph tables (duplicates, groups of groups, ignored
    types) through every call form: single, batched per-instance, batched shared, vertex32."""
    rng = np.random.RandomState(9000 + seed)
    m = _random_model(rng)
    ni = int(rng.choice([1, 2, 5, 9, 33]))
    normalize = bool(rng.randint(2))
    rates = rng.choice([0.0, 1.0, 0.3, -0.2, 5e-8, 2.5], size=(ni, m.nm)).astype(np.float32)
    pals = synth.make_palettes(m, rng.randint(0, 500, ni))
    skin = oracle.normalize(m) if normalize else None
    with DeformModel(m, normalize=normalize) as dm:
        pos, nrm = dm.deform_batched(rates, pals)
        v32 = dm.deform_batched(rates, pals, layout=api.OUT_VERTEX32, pos_scale=0.1)
        spos, snrm = dm.deform_batched(rates[0], pals, shared_weights=True)
        for i in range(ni):
            ep, en = oracle.skin(m, pals[i], oracle.morph(m, rates[i]), skin)
            gu.assert_bits_equal(pos[i], ep, f"seed {seed} inst {i} pos")
            gu.assert_bits_equal(nrm[i], en, f"seed {seed} inst {i} nrm")
            gu.assert_bits_equal(v32[i], oracle.repack32(m, ep, en, 0.1), f"seed {seed} inst {i} v32")
            sp, sn = oracle.skin(m, pals[i], oracle.morph(m, rates[0]), skin)
            gu.assert_bits_equal(spos[i], sp, f"seed {seed} shared inst {i} pos")
            gu.assert_bits_equal(snrm[i], sn, f"seed {seed} shared inst {i} nrm")
        p1, n1 = dm.deform(rates[ni - 1], pals[ni - 1])
        ep, en = oracle.skin(m, pals[ni - 1], oracle.morph(m, rates[ni - 1]), skin)
        gu.assert_bits_equal(p1, ep, "single pos")
        gu.assert_bits_equal(n1, en, "single nrm")
        # the same frame with every operand in device memory (the frame kernel's route)
        a, b = _one_frame_device(dm, m, rates[ni - 1] if m.nm else np.zeros(1, np.float32), pals[ni - 1], api.OUT_SOA)
        assert np.array_equal(a.view(np.uint32), ep.view(np.uint32).ravel()), f"seed {seed} device-resident frame pos"
        assert np.array_equal(b.view(np.uint32), en.view(np.uint32).ravel()), f"seed {seed} device-resident frame nrm"
        a, _ = _one_frame_device(dm, m, rates[ni - 1] if m.nm else np.zeros(1, np.float32), pals[ni - 1], api.OUT_VERTEX32, 0.1)
        assert np.array_equal(a.view(np.uint32), oracle.repack32(m, ep, en, 0.1).view(np.uint32).ravel()), f"seed {seed} device-resident v32"


def test_device_resident_io_matches_host_io(oracle):
    """The bench path: palettes, weights and outputs all resident in HBM."""
    m = synth.make_model(5000, 120, 10, 400, seed=31)
    ni = 37
    rates = synth.morph_weights(m.nm, np.arange(ni))
    pals = synth.make_palettes(m, np.arange(ni))
    with DeformModel(m) as dm:
        d_pal = DeviceBuffer.from_numpy(pals)
        d_w = DeviceBuffer.from_numpy(rates)
        sa, sb = dm.out_sizes(api.OUT_SOA, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
        dm.sync()
        pos = d_a.download((ni, m.nv, 3), np.float32)
        nrm = d_b.download((ni, m.nv, 3), np.float32)
        for i in (0, 1, 17, 36):
            ep, en = oracle_expect(oracle, m, rates[i], pals[i])
            gu.assert_bits_equal(pos[i], ep, "pos")
            gu.assert_bits_equal(nrm[i], en, "nrm")
        # shared weights, device resident
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA,
                              flags | api.WEIGHTS_SHARED)
        dm.sync()
        pos = d_a.download((ni, m.nv, 3), np.float32)
        ep, en = oracle_expect(oracle, m, rates[0], pals[20])
        gu.assert_bits_equal(pos[20], ep, "shared pos")
        for b in (d_pal, d_w, d_a, d_b):
            b.free()


def test_device_pointers_without_their_flag_are_rejected():
    """Device memory passed as if it were host memory (a missing *_ON_DEVICE flag) fails with a message instead of
    reaching a CPU memcpy."""
    m = synth.make_model(700, 20, 4, 30, seed=3)
    rates = synth.morph_weights(m.nm, 1)[0]
    pal = synth.make_palettes(m, [0])[0]
    out_a, out_b = np.empty((m.nv, 3), np.float32), np.empty((m.nv, 3), np.float32)
    with DeformModel(m) as dm:
        d_pal, d_w, d_a = DeviceBuffer.from_numpy(pal), DeviceBuffer.from_numpy(rates), DeviceBuffer(m.nv * 12)
        for kw, frag in ((dict(pal=d_pal.ptr), "MMDX_PALETTE_ON_DEVICE"), (dict(w=d_w.ptr), "MMDX_WEIGHTS_ON_DEVICE"),
                         (dict(a=d_a.ptr), "MMDX_OUT_ON_DEVICE")):
            with pytest.raises(api.MmdxError) as e:
                dm.deform_batched_raw(1, kw.get("w", rates.ctypes.data), kw.get("pal", pal.ctypes.data),
                                      kw.get("a", out_a.ctypes.data), out_b.ctypes.data, api.OUT_SOA, 0)
            assert e.value.status == 1 and frag in str(e.value)
        for b in (d_pal, d_w, d_a):
            b.free()


def test_bucketed_model_matches_oracle(oracle):
    """SURVEY 8d's "bucketed" variant: vertices pre-sorted by deform type inside each tile (synth.presort_by_class)
    -- another model as far as parity goes, the identity lane -> output permutation as far as the kernel goes."""
    m = synth.presort_by_class(synth.make_model(3001, 70, 9, 300, seed=91))
    rates = synth.morph_weights(m.nm, np.arange(3) * 5)
    pals = synth.make_palettes(m, np.arange(3))
    with DeformModel(m) as dm:
        pos, nrm = dm.deform_batched(rates, pals)
        for i in range(3):
            ep, en = oracle_expect(oracle, m, rates[i], pals[i])
            gu.assert_bits_equal(pos[i], ep, "pos")
            gu.assert_bits_equal(nrm[i], en, "nrm")


@pytest.mark.parametrize("nv", [1, 513, 4099])
def test_page_locked_host_outputs_are_written_directly(oracle, nv):
    """Host outputs in page-locked memory (mmdx_host_malloc) take the direct path -- the kernel stores into them
    over PCIe, no staging copy -- pageable ones the staging path: same bits, for every layout, single and batched
    calls, and for an output that starts in the middle of the locked allocation at an address that is not 16-byte
    aligned."""
    from simple_mmd_renderer_amd.engine import PinnedArray
    m = synth.make_model(nv, 40, 6, 60, seed=77 + nv)
    ni = 5
    rates = synth.morph_weights(m.nm, np.arange(ni) * 3)
    pals = synth.make_palettes(m, np.arange(ni) * 2)
    with DeformModel(m) as dm:
        pos, nrm = dm.deform_batched(rates, pals)                       # pageable numpy outputs: staging
        v32 = dm.deform_batched(rates, pals, layout=api.OUT_VERTEX32, pos_scale=0.1)
        for i in range(ni):
            ep, en = oracle_expect(oracle, m, rates[i], pals[i])
            gu.assert_bits_equal(pos[i], ep, "staged pos")
        pa, pb = PinnedArray((ni * nv * 3 + 8,), np.float32), PinnedArray((ni * nv * 3 + 8,), np.float32)
        pv = PinnedArray((ni * nv * 8 + 8,), np.float32)
        for off in (0, 1, 3):                                           # floats: 0, 4 and 12 bytes into the allocation
            pa.array[:] = -7.0
            pb.array[:] = -7.0
            pv.array[:] = -7.0
            dm.deform_batched_raw(ni, rates.ctypes.data, pals.ctypes.data, pa.ptr + 4 * off, pb.ptr + 4 * off, api.OUT_SOA, 0)
            gu.assert_bits_equal(pa.array[off:off + ni * nv * 3].reshape(ni, nv, 3), pos, f"direct pos, offset {off}")
            gu.assert_bits_equal(pb.array[off:off + ni * nv * 3].reshape(ni, nv, 3), nrm, f"direct nrm, offset {off}")
            assert (pa.array[:off] == -7.0).all() and (pa.array[off + ni * nv * 3:] == -7.0).all()   # nothing outside
            dm.deform_batched_raw(ni, rates.ctypes.data, pals.ctypes.data, pv.ptr + 4 * off, None, api.OUT_VERTEX32, 0, 0.1)
            gu.assert_bits_equal(pv.array[off:off + ni * nv * 8].reshape(ni, nv, 8), v32, f"direct vertex32, offset {off}")
            assert (pv.array[:off] == -7.0).all() and (pv.array[off + ni * nv * 8:] == -7.0).all()
        one = PinnedArray((nv, 8), np.float32)
        import ctypes as C
        f32p = C.POINTER(C.c_float)
        api.check(api.lib().mmdx_deform_vertex32(dm.h, rates[1].ctypes.data_as(f32p), pals[1].ctypes.data_as(f32p),
                                                 C.c_float(0.1), one.ptr))
        gu.assert_bits_equal(one.array, v32[1], "single frame, direct")
        for x in (pa, pb, pv, one):
            x.free()


@pytest.mark.parametrize("nv,layout,tries", [(4096, api.OUT_SOA, 6), (4100, api.OUT_VERTEX32, 6), (1001, api.OUT_SOA, 6),
                                             (4096, api.OUT_SOA, 1), (2048, api.OUT_SOA_POS16, 4)])
def test_placement_aware_output_alloc(oracle, nv, layout, tries):
    """mmdx_crowd_output_alloc: whatever placement it settles on, the arrays are the right size, usable by
    mmdx_deform_batched, and hold bit-exact results afterwards (the probe's fill pattern is overwritten)."""
    m = synth.make_model(nv, 40, 4, 100, seed=61)
    ni = 21
    rates = synth.morph_weights(m.nm, np.arange(ni))
    pals = synth.make_palettes(m, np.arange(ni))
    f16 = layout == api.OUT_SOA_POS16
    with DeformModel(m, f16_positions=f16) as dm:
        d_a, d_b, info = dm.alloc_outputs(layout, ni, tries)
        assert 1 <= info["tries"] <= max(tries, 1)
        probeable = tries > 1 and all((nv * bpv) % 16 == 0 for bpv in {api.OUT_SOA: (12,), api.OUT_VERTEX32: (32,),
                                                                        api.OUT_SOA_POS16: (6, 12)}[layout])
        assert info["probed"] == probeable
        if probeable:
            assert info["store_GBs"] > 0 and info["fill_GBs"] > 0
        assert (d_b is None) == (layout == api.OUT_VERTEX32)
        d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
        flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr if d_b else None, layout, flags,
                              0.1 if layout == api.OUT_VERTEX32 else 1.0)
        dm.sync()
        if layout == api.OUT_VERTEX32:
            got = d_a.download((ni, nv, 8), np.float32)
            want = dm.deform_batched(rates, pals, layout=layout, pos_scale=0.1)
            gu.assert_bits_equal(got, want, "v32 in placed arrays vs host-io call")
            ep, en = oracle_expect(oracle, m, rates[5], pals[5])
            gu.assert_bits_equal(got[5, :, 3:6], en, "normals")
        elif layout == api.OUT_SOA:
            pos, nrm = d_a.download((ni, nv, 3), np.float32), d_b.download((ni, nv, 3), np.float32)
            for i in (0, 9, 20):
                ep, en = oracle_expect(oracle, m, rates[i], pals[i])
                gu.assert_bits_equal(pos[i], ep, "pos")
                gu.assert_bits_equal(nrm[i], en, "nrm")
        else:
            want_p, want_n = dm.deform_batched(rates, pals, layout=layout)
            assert np.array_equal(d_a.download((ni, nv, 3), np.uint16), np.asarray(want_p).view(np.uint16).reshape(ni, nv, 3))
            gu.assert_bits_equal(d_b.download((ni, nv, 3), np.float32), want_n, "nrm")
        for b in (d_pal, d_w, d_a, d_b):
            if b is not None:
                b.free()


def test_fp16_position_variant(oracle):
    """MMDX_CREATE_F16_POSITIONS / MMDX_OUT_SOA_POS16 (config 5): base positions and morph offsets
    stored as binary16, arithmetic f32, output positions rounded to binary16 once."""
    m = synth.make_model(3001, 64, 12, 500, seed=808)
    q = m.copy()
    q.positions = m.positions.astype(np.float16).astype(np.float32)
    q.morph_value = m.morph_value.astype(np.float16).astype(np.float32)
    ni = 6
    rates = synth.morph_weights(m.nm, np.arange(ni) * 11)
    pals = synth.make_palettes(m, np.arange(ni) * 5)
    with DeformModel(m, f16_positions=True) as dm:
        pos16, nrm = dm.deform_batched(rates, pals, layout=api.OUT_SOA_POS16)
        spos16, _ = dm.deform_batched(rates[1], pals, layout=api.OUT_SOA_POS16, shared_weights=True)
        for i in range(ni):
            ep, en = oracle_expect(oracle, q, rates[i], pals[i])
            assert np.array_equal(pos16[i].view(np.uint16), ep.astype(np.float16).view(np.uint16)), f"inst {i}"
            gu.assert_bits_equal(nrm[i], en, "nrm")
            sp, _ = oracle_expect(oracle, q, rates[1], pals[i])
            assert np.array_equal(spos16[i].view(np.uint16), sp.astype(np.float16).view(np.uint16))
        with pytest.raises(api.MmdxError):
            dm.deform_batched(rates, pals, layout=api.OUT_SOA)


def test_poser_mirror_of_reference_interface(oracle):
    """mmd::Poser-shaped host API: SetMorphPose / palette injection / Deform / pose_image /
    UpdateDeformedVertices."""
    m, exp = gu.load("g12_mini_model")
    poser = Poser(m)
    for f in range(exp["rates"].shape[0]):
        poser.ResetPosing()
        for i, w in enumerate(exp["rates"][f]):
            poser.SetMorphPose(i, w)
        poser.SetSkinningMatrices(exp["palette"][f])
        poser.Deform()
        gu.assert_bits_equal(poser.pose_image.coordinates, exp["expect_pos"][f], "coordinates")
        gu.assert_bits_equal(poser.pose_image.normals, exp["expect_nrm"][f], "normals")
        gu.assert_bits_equal(poser.UpdateDeformedVertices(0.1), exp["expect_v32"][f], "vertex32")
    poser.close()


# ---- full BASELINE sizes: size-independent properties + sampled oracle checks -----------------------
def test_config3_crowd_full_size_properties(oracle):
    """1024 x 50 000: (a) instances with identical palettes produce identical outputs, (b) a sample
    of instances equals the oracle bit for bit, (c) the result does not depend on how instances are
    grouped into workgroups (batched in one call vs. two half calls)."""
    m = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(m, np.arange(ni) % 257)       # instances i and i+257 share a palette
    rates = synth.morph_weights(m.nm, 12)[0]
    with DeformModel(m) as dm:
        d_pal = DeviceBuffer.from_numpy(pals)
        sa, sb = dm.out_sizes(api.OUT_SOA, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        d_w = DeviceBuffer.from_numpy(rates)
        flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
        dm.sync()
        row = m.nv * 12

        def inst(buf, i):
            return buf.download((m.nv, 3), np.float32, offset=i * row)

        for i in (0, 3, 100, 511, 766):
            a, b = inst(d_a, i), inst(d_a, i + 257)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        for i in (0, 513, 1023):
            ep, en = oracle_expect(oracle, m, rates, pals[i])
            gu.assert_bits_equal(inst(d_a, i), ep, f"inst {i} pos")
            gu.assert_bits_equal(inst(d_b, i), en, f"inst {i} nrm")
        full = synth.checksum64(d_a.download((ni * m.nv * 3,), np.float32))
        # two half-size calls into the same buffers
        d_a.memset(0)
        half = ni // 2
        dm.deform_batched_raw(half, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
        dm.deform_batched_raw(half, d_w.ptr, d_pal.ptr + half * m.nb * 64, d_a.ptr + half * row,
                              d_b.ptr + half * row, api.OUT_SOA, flags)
        dm.sync()
        assert synth.checksum64(d_a.download((ni * m.nv * 3,), np.float32)) == full
        for b in (d_pal, d_a, d_b, d_w):
            b.free()


def _device_io(dm, m, rates, pals, layout, shared=False):
    """Run one batched call the way bench.py does (everything resident in HBM) and hand back the output buffers."""
    ni = pals.shape[0]
    d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
    sa, sb = dm.out_sizes(layout, ni)
    d_a, d_b = DeviceBuffer(sa), DeviceBuffer(max(sb, 16))
    d_a.memset(0xFF); d_b.memset(0xFF)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | (api.WEIGHTS_SHARED if shared else 0)
    dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr if sb else None, layout, flags)
    dm.sync()
    d_pal.free(); d_w.free()
    return d_a, d_b


def test_config2_full_size_single_frames(oracle):
    """BASELINE config 2 at its size (50 000 verts / 300 bones / 200 active morphs, f32): ONE frame per call
    through mmdx_deform (morph gather + group flatten fused into the deform kernel, kMorphFused1), as the viewer's
    frame() would call it -- every vertex of several frames against the oracle, SoA and 32-byte layouts."""
    m = synth.make_config("config2_50k")
    frames = [0, 1, 17, 63, 599]
    rates = synth.morph_weights(m.nm, frames)
    pals = synth.make_palettes(m, frames)
    assert (rates > 1e-7).sum(axis=1).min() >= 150              # "200 active morphs": most of them really are
    with DeformModel(m) as dm:
        for k, f in enumerate(frames):
            pos, nrm = dm.deform(rates[k], pals[k])
            ep, en = oracle_expect(oracle, m, rates[k], pals[k])
            gu.assert_bits_equal(pos, ep, f"config2 frame {f} pos")
            gu.assert_bits_equal(nrm, en, f"config2 frame {f} nrm")
            v32 = dm.deform_vertex32(rates[k], pals[k], 0.1)
            gu.assert_bits_equal(v32, oracle.repack32(m, ep, en, 0.1), f"config2 frame {f} v32")


def test_config2_full_size_64_frames_per_launch(oracle, fused_shape):
    """BASELINE config 2, the bandwidth form bench.py times: 64 frames of the 50k model in ONE launch with
    per-frame morph weights (kMorphFused4: 512-thread workgroups, 8 instances per walk over a morph row), all
    buffers in HBM -- EVERY frame, every vertex against the oracle."""
    m = synth.make_config("config2_50k")
    nfr = 64
    frames = np.arange(nfr)
    rates = synth.morph_weights(m.nm, frames)
    pals = synth.make_palettes(m, frames)
    with DeformModel(m) as dm:
        d_a, d_b = _device_io(dm, m, rates, pals, api.OUT_SOA)
        pos = d_a.download((nfr, m.nv, 3), np.float32)
        nrm = d_b.download((nfr, m.nv, 3), np.float32)
        d_a.free(); d_b.free()
        skin = oracle.normalize(m)
        for i in range(nfr):
            ep, en = oracle.skin(m, pals[i], oracle.morph(m, rates[i]), skin)
            gu.assert_bits_equal(pos[i], ep, f"config2 x64 frame {i} pos")
            gu.assert_bits_equal(nrm[i], en, f"config2 x64 frame {i} nrm")
        # a ragged count (not a multiple of the 8-instance walk) through the host-pointer form of the same call
        p2, n2 = dm.deform_batched(rates[:13], pals[:13])
        assert np.array_equal(p2.view(np.uint32), pos[:13].view(np.uint32))
        assert np.array_equal(n2.view(np.uint32), nrm[:13].view(np.uint32))


def test_config3_crowd_every_instance_vs_oracle(oracle):
    """BASELINE config 3 exactly as bench.py runs it (1024 instances, shared morph state, crowd_frames palettes,
    outputs in HBM): ALL 1024 instances, every vertex, bit for bit against the oracle."""
    from simple_mmd_renderer_amd.crowd import crowd_frames
    m = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(m, crowd_frames(0, ni))
    rates = synth.morph_weights(m.nm, 30)[0]
    with DeformModel(m) as dm:
        d_a, d_b = _device_io(dm, m, rates, pals, api.OUT_SOA, shared=True)
        skin = oracle.normalize(m)
        vimg = oracle.morph(m, rates)
        row = m.nv * 12
        chunk = 128
        for c in range(0, ni, chunk):
            pos = d_a.download((chunk, m.nv, 3), np.float32, offset=c * row)
            nrm = d_b.download((chunk, m.nv, 3), np.float32, offset=c * row)
            for j in range(chunk):
                ep, en = oracle.skin(m, pals[c + j], vimg, skin)
                gu.assert_bits_equal(pos[j], ep, f"crowd inst {c + j} pos")
                gu.assert_bits_equal(nrm[j], en, f"crowd inst {c + j} nrm")
        d_a.free(); d_b.free()


def test_config4_every_ranks_shard_full_size_on_one_device(oracle):
    """BASELINE config 4 (8192 instances sharded 1024 per GPU over 8 GPUs, no collective) has no node here: run what EVERY one
    of its 8 ranks runs -- bench.py's set-up for world 8: shard_instances(8192, 8, r), crowd_frames of the GLOBAL instance ids,
    1024 x 50k into placement-probed device arrays, shared rates -- one rank's shard after the other on the one device, and
    compare a sample of every shard (its first, last and four inner instances, every vertex) bit for bit with the oracle posed
    for the same global instance.  All 8 shards' outputs stay allocated together (9.8 GB), as they would on 8 devices."""
    from simple_mmd_renderer_amd.crowd import crowd_frames, shard_instances
    m = synth.make_config("config3_crowd")
    world, per = 8, 1024
    rates = synth.morph_weights(m.nm, 30)[0]
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    skin = oracle.normalize(m)
    vimg = oracle.morph(m, rates)
    row = m.nv * 12
    keep = []
    covered = 0
    with DeformModel(m) as dm:
        d_w = DeviceBuffer.from_numpy(rates)
        for r in range(world):
            lo, hi = shard_instances(per * world, world, r)
            assert hi - lo == per and lo == covered
            covered = hi
            pals = synth.make_palettes(m, crowd_frames(lo, hi))
            d_pal = DeviceBuffer.from_numpy(pals)
            d_a, d_b, _pl = dm.alloc_outputs(api.OUT_SOA, per, 2)
            d_a.memset(0xFF); d_b.memset(0xFF)
            dm.deform_batched_raw(per, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags, 1.0)
            dm.sync()
            keep.append((d_a, d_b, d_pal))
            for j in (0, 1, 341, 682, per - 2, per - 1):
                want = synth.make_palettes(m, crowd_frames(lo + j, lo + j + 1))[0]       # posed from the GLOBAL id alone
                ep, en = oracle.skin(m, want, vimg, skin)
                gu.assert_bits_equal(d_a.download((m.nv, 3), np.float32, offset=j * row), ep, f"rank {r} inst {lo + j} pos")
                gu.assert_bits_equal(d_b.download((m.nv, 3), np.float32, offset=j * row), en, f"rank {r} inst {lo + j} nrm")
        assert covered == per * world
        # nothing a later shard did touched an earlier shard's arrays
        d_a0, d_b0, d_pal0 = keep[0]
        ep, en = oracle.skin(m, synth.make_palettes(m, crowd_frames(5, 6))[0], vimg, skin)
        gu.assert_bits_equal(d_a0.download((m.nv, 3), np.float32, offset=5 * row), ep, "rank 0 inst 5 pos after all shards")
        for d_a, d_b, d_pal in keep:
            d_a.free(); d_b.free(); d_pal.free()
        d_w.free()


def test_crowd_store_policy_hints_and_defaults(oracle):
    """The store flavour of the crowd kernel's copy-out (mmdx.h MMDX_OUT_STORES_*): write-through (`sc1 nt`) or cached
    non-temporal (`nt`) stores -- same results, bit for bit, whichever is chosen.  The decision is made from the call alone (no
    table of addresses behind the boundary): the caller's hint -- mmdx_crowd_output_alloc hands out the probe's verdict as one --
    else write-through for outputs of >= 512 MB, cached for smaller ones; and only kernels that have the flavour report it."""
    from simple_mmd_renderer_amd.crowd import crowd_frames
    m = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(m, crowd_frames(0, ni))
    rates = synth.morph_weights(m.nm, 30)[0]
    base = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    skin = oracle.normalize(m)
    vimg = oracle.morph(m, rates)
    row = m.nv * 12
    sample = sorted(set(range(0, ni, 128)) | {1, 17, ni - 1})
    want = {i: oracle.skin(m, pals[i], vimg, skin) for i in sample}
    with DeformModel(m) as dm:
        d_w, d_pal = DeviceBuffer.from_numpy(rates), DeviceBuffer.from_numpy(pals)
        d_a, d_b, pl = dm.alloc_outputs(api.OUT_SOA, ni, 4)                 # probed: the verdict comes back as a flag
        p_a, p_b = DeviceBuffer(ni * row), DeviceBuffer(ni * row)           # plain

        def run(a, b, flags, n=ni):
            a.memset(0xFF); b.memset(0xFF)
            dm.deform_batched_raw(n, d_w.ptr, d_pal.ptr, a.ptr, b.ptr, api.OUT_SOA, flags, 1.0)
            dm.sync()
            return dm.last_store_policy()

        def check(a, b, what):
            for i in sample:
                gu.assert_bits_equal(a.download((m.nv, 3), np.float32, offset=i * row), want[i][0], f"{what}: inst {i} pos")
                gu.assert_bits_equal(b.download((m.nv, 3), np.float32, offset=i * row), want[i][1], f"{what}: inst {i} nrm")

        assert run(p_a, p_b, base | api.OUT_STORES_WRITE_THROUGH) == "sc1 nt"
        check(p_a, p_b, "write-through hint")
        assert run(p_a, p_b, base | api.OUT_STORES_CACHED) == "nt"
        check(p_a, p_b, "cached hint")
        assert run(p_a, p_b, base) == "sc1 nt"                               # nothing known about these arrays
        check(p_a, p_b, "default, unknown arrays")
        fast = pl["store_GBs"] >= 0.92 * pl["fill_GBs"]
        assert pl["store_flags"] == (api.OUT_STORES_CACHED if fast else api.OUT_STORES_WRITE_THROUGH)
        assert run(d_a, d_b, base | pl["store_flags"]) == ("nt" if fast else "sc1 nt")   # the probe's verdict, carried by the caller
        check(d_a, d_b, "probed arrays, verdict passed on")
        assert run(d_a, d_b, base) == "sc1 nt"                               # ... and nothing is remembered without it
        # free and re-allocate: whatever lands on the old addresses inherits nothing (the library keeps no record of arrays)
        old = (d_a.ptr, d_b.ptr)
        d_a.free(); d_b.free()
        d_a, d_b = DeviceBuffer(ni * row), DeviceBuffer(ni * row)
        assert run(d_a, d_b, base) == "sc1 nt", f"re-allocated arrays (old {old}, new {(d_a.ptr, d_b.ptr)})"
        check(d_a, d_b, "default, re-allocated arrays")
        assert run(d_a, d_b, base | api.MORPH_UNCHANGED | api.OUT_STORES_WRITE_THROUGH) == "sc1 nt"   # the no-morph-pass kernel too
        check(d_a, d_b, "write-through, morph pass skipped")
        assert run(p_a, p_b, base, n=64) == "nt"                             # 77 MB of output: stays cached
        with pytest.raises(api.MmdxError):
            dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, p_a.ptr, p_b.ptr, api.OUT_SOA,
                                  base | api.OUT_STORES_WRITE_THROUGH | api.OUT_STORES_CACHED, 1.0)
        # layouts / kernels without a write-through flavour take the hint and run their usual stores
        v32 = DeviceBuffer(ni * m.nv * 32)
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, v32.ptr, None, api.OUT_VERTEX32, base | api.OUT_STORES_WRITE_THROUGH, 0.1)
        dm.sync()
        assert dm.last_store_policy() == "nt"                                # what RAN, not what was asked for
        for i in sample[:3]:
            gu.assert_bits_equal(v32.download((m.nv, 8), np.float32, offset=i * m.nv * 32),
                                 oracle.repack32(m, want[i][0], want[i][1], 0.1), f"v32 with the hint: inst {i}")
        for b in (d_a, d_b, p_a, p_b, v32, d_w, d_pal):
            b.free()


def test_config3_crowd_vertex32_full_size_bench_call_form(oracle):
    """BASELINE config 3 with the viewer's interleaved 32-byte vertex as output (Deform + UpdateDeformedVertices in one kernel,
    main.cpp:838-859), in EXACTLY the call form bench.py times as `config3_vertex32_output`: 1024 instances, shared rates of
    frame 30, palettes (i*3) % 1801, pos_scale 0.1, output array from mmdx_crowd_output_alloc, profiling events on.  A 72-instance
    sample (every 16th instance plus the first and last groups), every vertex, bit for bit against oracle.repack32."""
    m = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(m, (np.arange(ni) * 3) % 1801)
    rates = synth.morph_weights(m.nm, 30)[0]
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    with DeformModel(m) as dm:
        d_w, d_pal = DeviceBuffer.from_numpy(rates), DeviceBuffer.from_numpy(pals)
        d_v32, none_b, _pl = dm.alloc_outputs(api.OUT_VERTEX32, ni, 4)
        assert none_b is None
        d_v32.memset(0xFF)
        dm.profile_enable(True)
        for _ in range(2):
            dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_v32.ptr, None, api.OUT_VERTEX32, flags, 0.1)
        dm.profile_collect()
        dm.profile_enable(False)
        dm.sync()
        skin = oracle.normalize(m)
        vimg = oracle.morph(m, rates)
        row = m.nv * 32
        for i in sorted(set(range(0, ni, 16)) | set(range(4)) | set(range(ni - 4, ni))):
            ep, en = oracle.skin(m, pals[i], vimg, skin)
            gu.assert_bits_equal(d_v32.download((m.nv, 8), np.float32, offset=i * row), oracle.repack32(m, ep, en, 0.1),
                                 f"crowd v32 inst {i}")
        for b in (d_w, d_pal, d_v32):
            b.free()


def test_config3_bucketed_vertices_bench_call_form(oracle):
    """bench.py's `config3_bucketed_vertices`: the config-3 model with its vertices pre-sorted by deform type inside each tile
    (synth.presort_by_class), 1024-instance shared-morph crowd into arrays from mmdx_crowd_output_alloc -- a 40-instance sample,
    every vertex, against the oracle run on the SORTED model."""
    m = synth.presort_by_class(synth.make_config("config3_crowd"))
    ni = 1024
    pals = synth.make_palettes(m, (np.arange(ni) * 3) % 1801)
    rates = synth.morph_weights(m.nm, 30)[0]
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    with DeformModel(m) as dm:
        d_w, d_pal = DeviceBuffer.from_numpy(rates), DeviceBuffer.from_numpy(pals)
        d_a, d_b, _pl = dm.alloc_outputs(api.OUT_SOA, ni, 4)
        d_a.memset(0xFF); d_b.memset(0xFF)
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
        dm.sync()
        skin = oracle.normalize(m)
        vimg = oracle.morph(m, rates)
        row = m.nv * 12
        for i in sorted(set(range(0, ni, 32)) | set(range(4)) | set(range(ni - 4, ni))):
            ep, en = oracle.skin(m, pals[i], vimg, skin)
            gu.assert_bits_equal(d_a.download((m.nv, 3), np.float32, offset=i * row), ep, f"bucketed inst {i} pos")
            gu.assert_bits_equal(d_b.download((m.nv, 3), np.float32, offset=i * row), en, f"bucketed inst {i} nrm")
        for b in (d_w, d_pal, d_a, d_b):
            b.free()


def test_config3prime_per_instance_morph_crowd_sample(oracle, fused_shape):
    """Config 3' (every instance its own facial state, 1024 x 50k, fused gather): a strided sample of 64 instances
    plus the first and last packs of 8, every vertex against the oracle."""
    m = synth.make_config("config3_crowd")
    ni = 1024
    fr = (np.arange(ni) * 7) % 600
    rates = synth.morph_weights(m.nm, fr)
    pals = synth.make_palettes(m, fr)
    with DeformModel(m) as dm:
        d_a, d_b = _device_io(dm, m, rates, pals, api.OUT_SOA)
        skin = oracle.normalize(m)
        row = m.nv * 12
        sample = sorted(set(range(0, ni, 16)) | set(range(8)) | set(range(ni - 8, ni)))
        for i in sample:
            ep, en = oracle.skin(m, pals[i], oracle.morph(m, rates[i]), skin)
            gu.assert_bits_equal(d_a.download((m.nv, 3), np.float32, offset=i * row), ep, f"3' inst {i} pos")
            gu.assert_bits_equal(d_b.download((m.nv, 3), np.float32, offset=i * row), en, f"3' inst {i} nrm")
        d_a.free(); d_b.free()


def test_config5_fp16_64_frames_per_launch(oracle, fused_shape):
    """BASELINE config 5 in the form bench.py times (64 frames per launch, f16 positions, per-frame morph weights):
    every 4th frame plus the last pack, every vertex against the f32 oracle on f16-quantised inputs."""
    m = synth.make_config("config5_256k")
    q = m.copy()
    q.positions = m.positions.astype(np.float16).astype(np.float32)
    q.morph_value = m.morph_value.astype(np.float16).astype(np.float32)
    nfr = 64
    frames = np.arange(nfr)
    rates = synth.morph_weights(m.nm, frames)
    pals = synth.make_palettes(m, frames)
    with DeformModel(m, f16_positions=True) as dm:
        d_a, d_b = _device_io(dm, m, rates, pals, api.OUT_SOA_POS16)
        skin = oracle.normalize(q)
        for i in sorted(set(range(0, nfr, 4)) | set(range(nfr - 8, nfr))):
            ep, en = oracle.skin(q, pals[i], oracle.morph(q, rates[i]), skin)
            got16 = d_a.download((m.nv, 3), np.float16, offset=i * m.nv * 6)
            assert np.array_equal(got16.view(np.uint16), ep.astype(np.float16).view(np.uint16)), f"config5 frame {i} pos16"
            gu.assert_bits_equal(d_b.download((m.nv, 3), np.float32, offset=i * m.nv * 12), en, f"config5 frame {i} nrm")
        d_a.free(); d_b.free()


def test_config5_fp16_full_size_sample(oracle):
    """262 144 verts / 512 bones / 1024 morphs x 4096 entries, f16 positions: one frame against the
    oracle (f16-quantised inputs), plus idempotence of a repeated call."""
    m = synth.make_config("config5_256k")
    q = m.copy()
    q.positions = m.positions.astype(np.float16).astype(np.float32)
    q.morph_value = m.morph_value.astype(np.float16).astype(np.float32)
    rates = synth.morph_weights(m.nm, [3, 40])
    pals = synth.make_palettes(m, [3, 40])
    with DeformModel(m, f16_positions=True) as dm:
        pos16, nrm = dm.deform_batched(rates, pals, layout=api.OUT_SOA_POS16)
        again16, again_n = dm.deform_batched(rates, pals, layout=api.OUT_SOA_POS16)
        assert np.array_equal(pos16.view(np.uint16), again16.view(np.uint16))
        for i in range(2):
            ep, en = oracle_expect(oracle, q, rates[i], pals[i])
            assert np.array_equal(pos16[i].view(np.uint16), ep.astype(np.float16).view(np.uint16))
            gu.assert_bits_equal(nrm[i], en, "nrm")


def test_cpp_host_mirror_frame_loop():
    """The C++ host side (host/mmdx_poser.hpp: mmd::Poser-shaped class over the C ABI) runs the
    reference's per-frame sequence; the program checks pose_image*0.1f == vertex stream itself."""
    import subprocess
    from simple_mmd_renderer_amd import build
    exe = build.build_host_example()
    r = subprocess.run([exe, "20000", "30"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "frames=30 nv=20000" in r.stdout and "MISMATCH" not in r.stdout
    again = subprocess.run([exe, "20000", "30"], capture_output=True, text=True, timeout=120)
    assert again.stdout == r.stdout            # deterministic, run to run


def test_cpp_motion_example_matches_python_path(tmp_path):
    """host/motion_example.cpp (mmd::Poser / Motion / MotionPlayer mirror: .pmx + .vmd -> SeekFrame ->
    PrePhysicsPosing -> Deform -> UpdateDeformedVertices, all on the GPU) against the same steps driven
    from Python over the same C ABI: identical checksums."""
    import subprocess
    from simple_mmd_renderer_amd import build, pmx, vmd
    nb = 40
    rig = synth.make_ik_rig(nb, 11, n_ik=3, n_append=4)
    m = synth.make_model(600, nb, 5, 60, seed=12)
    m.bone_pos, m.bone_parent = rig[0].copy(), rig[1].astype(m.bone_parent.dtype)
    bnames = ["骨%d" % b for b in range(nb)]
    mnames = ["表情%d" % k for k in range(m.nm)]
    (tmp_path / "m.pmx").write_bytes(pmx.write_pmx(m, pmx.PmxWriteOptions(rig=rig, bone_names=bnames, morph_names=mnames)))
    rng = np.random.RandomState(5)
    mk = [(n, int(f), float(np.float32(rng.uniform(0, 1)))) for n in mnames[:4] for f in (0, 7, 19)]
    (tmp_path / "m.vmd").write_bytes(vmd.write_vmd(synth.make_bone_keys(bnames[:30], 13, keys_per=4, span=24), mk))
    frames = 25
    exe = build.build_host_example(name="motion_example")
    r = subprocess.run([exe, str(tmp_path / "m.pmx"), str(tmp_path / "m.vmd"), str(frames)], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0 and "checksum=" in r.stdout, r.stdout + r.stderr
    # the same through the Python binding
    pm = pmx.load_pmx(str(tmp_path / "m.pmx"))
    v = vmd.Vmd(str(tmp_path / "m.vmd"))
    bm, mm, sk = v.bind_bones(pm.bone_names), v.bind_morphs(pm.morph_names), pm.skeleton()

    def fnv(a):
        h = 1469598103934665603
        for byte in np.ascontiguousarray(a).view(np.uint8).ravel().tolist():
            h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    h = 0
    with DeformModel(pm.flat) as dm:
        for f in range(frames):
            rates = mm.eval([f], dm)[0]
            pal = sk.solve(bm.eval([f], dm), dm, morph_weights=rates)[0]
            _pos, nrm = dm.deform(rates, pal)
            v32 = dm.deform_vertex32(rates, pal, 0.1)
            h = (h * 31 + fnv(v32) + fnv(nrm)) & 0xFFFFFFFFFFFFFFFF
    want = "frames=%d nv=%d nb=%d mapped_bones=30 checksum=%016x" % (frames, pm.flat.nv, nb, h)
    assert r.stdout.strip() == want


def test_pmx_file_to_gpu_end_to_end():
    """tests/golden/pmx_small.pmx -> this repo's PMX loader -> mmdx_model_create -> deform, against what
    libmmd (PmxReader + Normalize + Poser::Deform + repack) produced from the same file."""
    import os
    from simple_mmd_renderer_amd import pmx
    z = np.load(os.path.join(gu.GOLDEN_DIR, "pmx_small_expect.npz"))
    pm = pmx.load_pmx(os.path.join(gu.GOLDEN_DIR, "pmx_small.pmx"))
    with DeformModel(pm.flat, normalize=True) as dm:
        t, _, _ = dm.get_skin()
        want = z["norm_type"]
        assert np.array_equal(np.where(t == 0, 0, np.where(t == 2, 2, 1)),
                              np.where(want == 0, 0, np.where(want == 2, 2, 1)))
        for f in range(z["rates"].shape[0]):
            pos, nrm = dm.deform(z["rates"][f], z["palette"][f])
            gu.assert_bits_equal(pos, z["expect_pos"][f], "pos")
            gu.assert_bits_equal(nrm, z["expect_nrm"][f], "nrm")
            gu.assert_bits_equal(dm.deform_vertex32(z["rates"][f], z["palette"][f], 0.1), z["expect_v32"][f], "v32")


def test_vmd_morph_rates_on_device_golden():
    """tests/golden/vmd_small.vmd -> device evaluation of every model morph at 261 frames, against
    libmmd's Motion::GetMorphPose answers."""
    import os
    from simple_mmd_renderer_amd import vmd
    z = np.load(os.path.join(gu.GOLDEN_DIR, "vmd_small_expect.npz"))
    v = vmd.Vmd(os.path.join(gu.GOLDEN_DIR, "vmd_small.vmd"))
    mm = v.bind_morphs([str(n) for n in z["model_morph_names"]])
    got = mm.eval(z["frames"])
    gu.assert_bits_equal(got, z["expect_rates"], "rates")
    mm.close()


def test_vmd_to_crowd_end_to_end(oracle):
    """PMX file + VMD file -> rates for a crowd evaluated in HBM (every instance at its own frame) ->
    mmdx_deform_batched with device-resident weights; against the oracle per instance."""
    import os
    from simple_mmd_renderer_amd import pmx, vmd
    pm = pmx.load_pmx(os.path.join(gu.GOLDEN_DIR, "pmx_small.pmx"))
    m = pm.flat
    rng = np.random.RandomState(5)
    keys = []
    for n in pm.morph_names:
        for f in sorted(rng.choice(120, 5, replace=False)):
            keys.append((n, int(f), float(np.float32(rng.uniform(0, 1)))))
    v = vmd.Vmd(vmd.write_vmd([], keys))
    mm = v.bind_morphs(pm.morph_names)
    assert mm.n_mapped == m.nm
    ni = 21
    frames = (np.arange(ni) * 7 % 130).astype(np.uint32)
    pals = synth.make_palettes(m, frames)
    with DeformModel(m) as dm:
        d_fr, d_pal = DeviceBuffer.from_numpy(frames), DeviceBuffer.from_numpy(pals)
        d_w = DeviceBuffer(ni * m.nm * 4)
        sa, sb = dm.out_sizes(api.OUT_SOA, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        mm.eval_device(ni, d_fr.ptr, d_w.ptr, model=dm)        # same stream as the deform that follows
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA,
                              api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE)
        dm.sync()
        rates = d_w.download((ni, m.nm), np.float32)
        pos = d_a.download((ni, m.nv, 3), np.float32)
        nrm = d_b.download((ni, m.nv, 3), np.float32)
        # rates: the oracle's restatement of GetMorphPose on the bound tracks
        tracks = {n: v.morph_track(i) for i, n in enumerate(v.morph_track_names)}
        off, fr, w = [0], [], []
        for n in pm.morph_names:
            fr += list(tracks[n][0]); w += list(tracks[n][1]); off.append(len(fr))
        want = oracle.morph_tracks(np.asarray(off, np.uint32), np.asarray(fr, np.uint32), np.asarray(w, np.float32), frames)
        gu.assert_bits_equal(rates, want, "rates")
        for i in range(ni):
            ep, en = oracle_expect(oracle, m, want[i], pals[i])
            gu.assert_bits_equal(pos[i], ep, f"inst {i} pos")
            gu.assert_bits_equal(nrm[i], en, f"inst {i} nrm")
        for b in (d_fr, d_pal, d_w, d_a, d_b):
            b.free()
    mm.close()



@pytest.mark.parametrize("ni,fused", [(21, "1"), (21, "2"), (5, "1"), (5, "0")])
def test_shared_crowd_paths_agree_and_unchanged_flag(oracle, monkeypatch, ni, fused):
    """A crowd with a shared facial state has three routes -- the morph gather inside the deform kernel (small crowds;
    MMDX_SHARED_FUSED=2 forces it), the separate morph pass, and the kept morphed positions with MMDX_MORPH_UNCHANGED --
    all bit-identical to the oracle; UNCHANGED before any shared call is an error; after new weights the kept
    positions are the new ones."""
    monkeypatch.setenv("MMDX_SHARED_FUSED", fused)
    api.lib().mmdx_debug_reload_env()
    m = synth.make_model(3000, 50, 7, 250, seed=555)
    pals = synth.make_palettes(m, np.arange(ni) * 3)
    r1, r2 = synth.morph_weights(m.nm, 5)[0], synth.morph_weights(m.nm, 40)[0]
    skin = oracle.normalize(m)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    with DeformModel(m) as dm:
        d_pal = DeviceBuffer.from_numpy(pals)
        d_w1, d_w2 = DeviceBuffer.from_numpy(r1), DeviceBuffer.from_numpy(r2)
        sa, sb = dm.out_sizes(api.OUT_SOA, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)

        def check(rates, what):
            dm.sync()
            pos, nrm = d_a.download((ni, m.nv, 3), np.float32), d_b.download((ni, m.nv, 3), np.float32)
            vimg = oracle.morph(m, rates)
            for i in range(ni):
                ep, en = oracle.skin(m, pals[i], vimg, skin)
                gu.assert_bits_equal(pos[i], ep, f"{what} inst {i} pos")
                gu.assert_bits_equal(nrm[i], en, f"{what} inst {i} nrm")
            d_a.memset(0); d_b.memset(0)

        with pytest.raises(api.MmdxError, match="UNCHANGED"):
            dm.deform_batched_raw(ni, None, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags | api.MORPH_UNCHANGED)
        dm.deform_batched_raw(ni, d_w1.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
        check(r1, "gather in the kernel")
        dm.deform_batched_raw(ni, None, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags | api.MORPH_UNCHANGED)
        check(r1, "unchanged (weights pointer NULL)")
        dm.deform_batched_raw(ni, d_w2.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
        check(r2, "new weights")
        dm.deform_batched_raw(ni, d_w1.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags | api.MORPH_UNCHANGED)
        check(r2, "unchanged after new weights (the pointer passed is ignored)")
        for b in (d_pal, d_w1, d_w2, d_a, d_b):
            b.free()
    monkeypatch.delenv("MMDX_SHARED_FUSED")
    api.lib().mmdx_debug_reload_env()


# ---- one frame of one model: the latency-ordered frame kernel vs the tile kernel -------------------------------------
def _one_frame_device(dm, m, rates, pal, layout, pos_scale=1.0, misalign=0):
    """ni = 1 with every operand in device memory (the frame kernel's case); `misalign` shifts the output pointers by that
    many bytes (4-byte multiples keep the element alignment; the 16-byte fast paths must then not be taken)."""
    d_pal, d_w = DeviceBuffer.from_numpy(pal[None]), DeviceBuffer.from_numpy(rates[None])
    sa, sb = dm.out_sizes(layout, 1)
    d_a, d_b = DeviceBuffer(sa + 64), DeviceBuffer(max(sb, 16) + 64)
    d_a.memset(0xFF); d_b.memset(0xFF)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
    dm.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr + misalign, (d_b.ptr + misalign) if sb else None, layout, flags, pos_scale)
    dm.sync()
    a = d_a.download((sa + 64,), np.uint8)[misalign:misalign + sa]
    b = d_b.download((max(sb, 16) + 64,), np.uint8)[misalign:misalign + sb]
    for x in (d_pal, d_w, d_a, d_b):
        x.free()
    return a, b


def _with_group_morphs(m):
    """Two group morphs on top of the model's vertex morphs: one over three of them (sub-rates incl. one that falls under the 1e-7
    skip once multiplied), one over a vertex morph and the first group (depth 2)."""
    nm = m.nm
    m.morph_type = np.concatenate([m.morph_type, [MORPH_GROUP, MORPH_GROUP]]).astype(np.int32)
    idx = list(m.morph_index) + [0, 1 % nm, 2 % nm] + [3 % nm, nm]
    val = list(map(tuple, m.morph_value)) + [(0.5, 0, 0), (1e-4, 0, 0), (2.0, 0, 0)] + [(1.0, 0, 0), (0.75, 0, 0)]
    m.morph_off = np.concatenate([m.morph_off, [len(idx) - 2, len(idx)]]).astype(np.uint32)
    m.morph_index = np.asarray(idx, np.uint32)
    m.morph_value = np.asarray(val, np.float32).reshape(-1, 3)
    return m


FRAME_VARIANTS = [("tile kernel", {"MMDX_FRAME_KERNEL": "0"}), ("tile kernel, 256 threads", {"MMDX_FRAME_KERNEL": "0", "MMDX_THREADS": "256"}),
                  ("frame kernel, 128", {"MMDX_FRAME_KERNEL": "2", "MMDX_FRAME_THREADS": "128"}),
                  ("frame kernel, 256", {"MMDX_FRAME_KERNEL": "2", "MMDX_FRAME_THREADS": "256"}), ("default", {})]


def _with_env(monkeypatch, env):
    for k in ("MMDX_FRAME_KERNEL", "MMDX_FRAME_THREADS", "MMDX_THREADS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    api.lib().mmdx_debug_reload_env()


@pytest.mark.parametrize("nv", [1, 63, 129, 512, 513, 1000, 4099])
def test_frame_kernel_every_layout_and_size(oracle, monkeypatch, nv):
    """One frame through every single-frame route (tile kernel with 256 / 512 threads, frame kernel with 128 / 256 lanes per
    workgroup, the default choice): SoA, the 32-byte vertex (16-byte aligned and not) and the f16-position layout, ragged
    sizes, group morphs with rates around the 1e-7 skip -- all bit-identical to the oracle."""
    m = _with_group_morphs(synth.make_model(nv, 40, 9, min(300, max(nv // 2, 1)), seed=9100 + nv))
    q = m.copy()
    q.positions = m.positions.astype(np.float16).astype(np.float32)
    q.morph_value = m.morph_value.astype(np.float16).astype(np.float32)
    rates = synth.morph_weights(m.nm, 23)[0]
    rates[0] = 0.0; rates[1 % m.nm] = 5e-8; rates[2 % m.nm] = 1.0       # skipped, skipped (< 1e-7), full
    pal = synth.make_palettes(m, 11)[0]
    ep, en = oracle_expect(oracle, m, rates, pal)
    ev32 = oracle.repack32(m, ep, en, 0.1)
    ep16, en16 = oracle_expect(oracle, q, rates, pal)
    with DeformModel(m) as dm, DeformModel(m, f16_positions=True) as dm16:
        for what, env in FRAME_VARIANTS:
            _with_env(monkeypatch, env)
            a, b = _one_frame_device(dm, m, rates, pal, api.OUT_SOA)
            assert np.array_equal(a.view(np.uint32), ep.view(np.uint32).ravel()), f"{what}: pos"
            assert np.array_equal(b.view(np.uint32), en.view(np.uint32).ravel()), f"{what}: nrm"
            for mis in (0, 4):
                a, _ = _one_frame_device(dm, m, rates, pal, api.OUT_VERTEX32, 0.1, mis)
                assert np.array_equal(a.view(np.uint32), ev32.view(np.uint32).ravel()), f"{what}: vertex32 (+{mis} B)"
            a, b = _one_frame_device(dm16, m, rates, pal, api.OUT_SOA_POS16)
            assert np.array_equal(a.view(np.uint16), ep16.astype(np.float16).view(np.uint16).ravel()), f"{what}: f16 pos"
            assert np.array_equal(b.view(np.uint32), en16.view(np.uint32).ravel()), f"{what}: f16 nrm"
    _with_env(monkeypatch, {})


def test_frame_kernel_no_morphs_many_bones_nonfinite(oracle, monkeypatch):
    """The frame kernel's other corners: a model without morph slots (kMorphNone), a tile that uses more than a hundred bones
    (palette rows beyond the first batch of a 128-lane workgroup), non-finite morph offsets (skipped slots must not touch
    them) and more slots than one batch of the weight staging (> 1024)."""
    rng = np.random.RandomState(77)
    cases = []
    m0 = synth.make_model(1500, 30, 1, 1, seed=31)                              # no morphs at all
    m0.morph_type, m0.morph_off = np.zeros(0, np.int32), np.zeros(1, np.uint32)
    m0.morph_index, m0.morph_value = np.zeros(0, np.uint32), np.zeros((0, 3), np.float32)
    cases.append(("no morphs", m0, np.zeros(0, np.float32)))
    m1 = synth.make_model(700, 300, 4, 100, seed=32)
    m1.bone_ids[:] = rng.randint(0, 300, m1.bone_ids.shape)                     # every tile touches ~all bones
    cases.append(("many bones", m1, synth.morph_weights(m1.nm, 3)[0]))
    m2 = synth.make_model(900, 20, 6, 200, seed=33)
    m2.morph_value[m2.morph_off[0]:m2.morph_off[1], 0] = np.inf; m2.morph_value[m2.morph_off[2] + 1, 2] = np.nan
    r2 = synth.morph_weights(m2.nm, 9)[0]; r2[::2] = 0.0                        # the non-finite morphs 0 and 2 are skipped
    cases.append(("non-finite offsets", m2, r2))
    m3 = synth.make_model(2000, 20, 1300, 3, seed=34)                           # 1300 slots
    cases.append(("1300 slots", m3, synth.morph_weights(m3.nm, 4)[0]))
    for name, m, rates in cases:
        pal = synth.make_palettes(m, 5)[0]
        ep, en = oracle_expect(oracle, m, rates, pal)
        with DeformModel(m) as dm:
            for what, env in FRAME_VARIANTS:
                _with_env(monkeypatch, env)
                a, b = _one_frame_device(dm, m, rates if rates.size else np.zeros(1, np.float32), pal, api.OUT_SOA)
                assert np.isfinite(ep).all()
                assert np.array_equal(a.view(np.uint32), ep.view(np.uint32).ravel()), f"{name}, {what}: pos"
                assert np.array_equal(b.view(np.uint32), en.view(np.uint32).ravel()), f"{name}, {what}: nrm"
    _with_env(monkeypatch, {})


def test_config2_single_frames_device_resident(oracle, monkeypatch):
    """BASELINE config 2, one frame per call with everything in HBM (what bench.py times as config2_single_frame): the frame
    kernel (default for a 98-tile model) and the tile kernel, every vertex of three frames against the oracle."""
    m = synth.make_config("config2_50k")
    frames = [0, 17, 599]
    rates, pals = synth.morph_weights(m.nm, frames), synth.make_palettes(m, frames)
    with DeformModel(m) as dm:
        for k, f in enumerate(frames):
            ep, en = oracle_expect(oracle, m, rates[k], pals[k])
            for what, env in (("default", {}), ("tile kernel", {"MMDX_FRAME_KERNEL": "0"})):
                _with_env(monkeypatch, env)
                a, b = _one_frame_device(dm, m, rates[k], pals[k], api.OUT_SOA)
                assert np.array_equal(a.view(np.uint32), ep.view(np.uint32).ravel()), f"frame {f}, {what}: pos"
                assert np.array_equal(b.view(np.uint32), en.view(np.uint32).ravel()), f"frame {f}, {what}: nrm"
    _with_env(monkeypatch, {})


# ---- per-instance morph weights with everything in HBM ------------------------------------------------------------------
def _batch_device(dm, rates, pals, layout, pos_scale=1.0, misalign=0):
    ni = pals.shape[0]
    d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
    sa, sb = dm.out_sizes(layout, ni)
    d_a, d_b = DeviceBuffer(sa + 64), DeviceBuffer(max(sb, 16) + 64)
    d_a.memset(0xFF); d_b.memset(0xFF)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
    dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr + misalign, (d_b.ptr + misalign) if sb else None, layout, flags, pos_scale)
    dm.sync()
    a = d_a.download((sa + 64,), np.uint8)[misalign:misalign + sa]
    b = d_b.download((max(sb, 16) + 64,), np.uint8)[misalign:misalign + sb]
    for x in (d_pal, d_w, d_a, d_b):
        x.free()
    return a, b


@pytest.fixture(params=[0, 1], ids=["fused4-kernel", "pack-kernel"])
def fused_shape(request, monkeypatch, hip_lib):
    """The two kernels per-instance morph weights can run: deform_kernel<512, ., kMorphFused4> (default) and the opt-in
    pack_kernel (MMDX_FUSED_PACK=1, round 4's 80-register / three-workgroups-per-CU shape): same results, bit for bit."""
    monkeypatch.setenv("MMDX_FUSED_PACK", str(request.param))
    hip_lib.mmdx_debug_reload_env()
    yield request.param
    monkeypatch.delenv("MMDX_FUSED_PACK")
    hip_lib.mmdx_debug_reload_env()


@pytest.mark.parametrize("nv,ni", [(1, 2), (63, 5), (300, 8), (513, 9), (1000, 13), (4099, 17), (2600, 33)])
def test_per_instance_morphs_device_resident_every_layout_and_size(oracle, nv, ni, fused_shape):
    """Per-instance morph weights, device-resident operands (the form bench.py times): ragged vertex counts and pack counts
    (partial last quad and pack of 8), group morphs with rates around the 1e-7 skip, SoA / 32-byte vertex (16-byte aligned and
    not) / f16 positions: bit-identical to the oracle."""
    m = _with_group_morphs(synth.make_model(nv, 40, 9, min(300, max(nv // 2, 1)), seed=9300 + nv))
    q = m.copy()
    q.positions = m.positions.astype(np.float16).astype(np.float32)
    q.morph_value = m.morph_value.astype(np.float16).astype(np.float32)
    rates = synth.morph_weights(m.nm, np.arange(ni) * 7 + 3)
    rates[:, 0] = np.where(np.arange(ni) % 3 == 0, 0.0, rates[:, 0])
    rates[:, 1 % m.nm] = np.where(np.arange(ni) % 2 == 0, 5e-8, 1.0)
    pals = synth.make_palettes(m, np.arange(ni) * 5)
    exp = [oracle_expect(oracle, m, rates[i], pals[i]) for i in range(ni)]
    exp16 = [oracle_expect(oracle, q, rates[i], pals[i]) for i in range(ni)]
    ep = np.stack([e[0] for e in exp]); en = np.stack([e[1] for e in exp])
    ev32 = np.stack([oracle.repack32(m, e[0], e[1], 0.1) for e in exp])
    ep16 = np.stack([e[0] for e in exp16]).astype(np.float16); en16 = np.stack([e[1] for e in exp16])
    with DeformModel(m) as dm, DeformModel(m, f16_positions=True) as dm16:
        for what in ("tile kernel",):
            a, b = _batch_device(dm, rates, pals, api.OUT_SOA)
            assert np.array_equal(a.view(np.uint32), ep.view(np.uint32).ravel()), f"{what}: pos"
            assert np.array_equal(b.view(np.uint32), en.view(np.uint32).ravel()), f"{what}: nrm"
            for mis in (0, 4):
                a, _ = _batch_device(dm, rates, pals, api.OUT_VERTEX32, 0.1, mis)
                assert np.array_equal(a.view(np.uint32), ev32.view(np.uint32).ravel()), f"{what}: vertex32 (+{mis} B)"
            a, b = _batch_device(dm16, rates, pals, api.OUT_SOA_POS16)
            assert np.array_equal(a.view(np.uint16), ep16.view(np.uint16).ravel()), f"{what}: f16 pos"
            assert np.array_equal(b.view(np.uint32), en16.view(np.uint32).ravel()), f"{what}: f16 nrm"


def test_per_instance_morphs_device_resident_corners(oracle, fused_shape):
    """Device-resident per-instance morphs: tiles that use hundreds of bones, non-finite morph offsets (the predicated skip),
    1 300 slots."""
    rng = np.random.RandomState(78)
    ni = 11
    cases = []
    m1 = synth.make_model(700, 300, 4, 100, seed=42)
    m1.bone_ids[:] = rng.randint(0, 300, m1.bone_ids.shape)
    cases.append(("many bones", m1, synth.morph_weights(m1.nm, np.arange(ni))))
    m2 = synth.make_model(900, 20, 6, 200, seed=43)
    m2.morph_value[m2.morph_off[0]:m2.morph_off[1], 0] = np.inf; m2.morph_value[m2.morph_off[2] + 1, 2] = np.nan
    r2 = synth.morph_weights(m2.nm, np.arange(ni) * 3); r2[:, ::2] = 0.0
    cases.append(("non-finite offsets", m2, r2))
    m3 = synth.make_model(2000, 20, 1300, 3, seed=44)
    cases.append(("1300 slots", m3, synth.morph_weights(m3.nm, np.arange(ni) * 2)))
    for name, m, rates in cases:
        pals = synth.make_palettes(m, np.arange(ni) * 4)
        with DeformModel(m) as dm:
            a, b = _batch_device(dm, rates, pals, api.OUT_SOA)
            pos, nrm = a.view(np.float32).reshape(ni, m.nv, 3), b.view(np.float32).reshape(ni, m.nv, 3)
            for i in range(ni):
                ep, en = oracle_expect(oracle, m, rates[i], pals[i])
                assert np.isfinite(ep).all()
                gu.assert_bits_equal(pos[i], ep, f"{name} inst {i} pos")
                gu.assert_bits_equal(nrm[i], en, f"{name} inst {i} nrm")
