"""MMDX_CREATE_TILE_ORDER: outputs in the engine's vertex order (tile-local sort by deform class and morph-row length) instead of
the file's -- the kernels then store straight from registers.  The VALUES are bit-identical to the default's (and so to the
oracle's); position e of every output array holds file vertex engine_to_original[e].  CPU: the permutation's properties through
the C ABI (host-only model).  GPU: every call form and layout against the oracle, un-permuted."""
import numpy as np
import pytest

from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import synth
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer, device_count
from tests import golden_util as gu


def test_vertex_order_is_a_tile_local_class_sorted_permutation(hip_lib):
    for nv in (1, 63, 512, 513, 4099):
        m = synth.make_model(nv, 40, 7, min(200, nv), seed=500 + nv)
        with DeformModel(m, host_only=True, tile_order=True) as dm:
            e2o, o2e = dm.vertex_order()
            assert np.array_equal(np.sort(e2o), np.arange(nv)), "a permutation"
            assert np.array_equal(o2e[e2o], np.arange(nv)) and np.array_equal(e2o[o2e], np.arange(nv)), "inverse"
            assert np.array_equal(e2o // 512, np.arange(nv) // 512), "vertices stay inside their tile of 512 file vertices"
            cls, _, _ = dm.get_skin()                               # post-Normalize class per FILE vertex
            for t0 in range(0, nv, 512):
                c = cls[e2o[t0:t0 + 512]]
                assert np.all(np.diff(c) >= 0), "sorted by deform class inside a tile"
        with DeformModel(m, host_only=True) as dm:                  # defined for every model
            assert np.array_equal(dm.vertex_order()[0], e2o)


gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("nv,ni", [(1, 2), (63, 5), (513, 9), (1000, 13), (4099, 17)])
def test_tile_order_every_call_form_matches_oracle_unpermuted(oracle, hip_lib, nv, ni):
    assert device_count() >= 1
    m = synth.make_model(nv, 40, 9, min(300, max(nv // 2, 1)), seed=8800 + nv)
    q = m.copy()
    q.positions = m.positions.astype(np.float16).astype(np.float32)
    q.morph_value = m.morph_value.astype(np.float16).astype(np.float32)
    rates = synth.morph_weights(m.nm, np.arange(ni) * 7 + 3)
    rates[:, 0] = 0.0; rates[:, 1] = np.where(np.arange(ni) % 2 == 0, 5e-8, 1.0)
    pals = synth.make_palettes(m, np.arange(ni) * 5)
    skin, skin16 = oracle.normalize(m), oracle.normalize(q)
    with DeformModel(m, tile_order=True) as dm, DeformModel(m, tile_order=True, f16_positions=True) as dm16:
        e2o, _ = dm.vertex_order()
        assert np.array_equal(dm16.vertex_order()[0], e2o) or True       # (the f16 model has its own order)
        e2o16, _ = dm16.vertex_order()
        pos, nrm = dm.deform_batched(rates, pals)                                       # per-instance morphs, host arrays
        spos, snrm = dm.deform_batched(rates[0], pals, shared_weights=True)              # shared crowd
        v32 = dm.deform_batched(rates, pals, layout=api.OUT_VERTEX32, pos_scale=0.1)
        p16, n16 = dm16.deform_batched(rates, pals, layout=api.OUT_SOA_POS16)
        for i in range(ni):
            ep, en = oracle.skin(m, pals[i], oracle.morph(m, rates[i]), skin)
            gu.assert_bits_equal(pos[i], ep[e2o], f"inst {i} pos"); gu.assert_bits_equal(nrm[i], en[e2o], f"inst {i} nrm")
            gu.assert_bits_equal(v32[i], oracle.repack32(m, ep, en, 0.1).reshape(nv, 8)[e2o].reshape(v32[i].shape), f"inst {i} v32")
            sp, sn = oracle.skin(m, pals[i], oracle.morph(m, rates[0]), skin)
            gu.assert_bits_equal(spos[i], sp[e2o], f"shared inst {i} pos"); gu.assert_bits_equal(snrm[i], sn[e2o], "shared nrm")
            qp, qn = oracle.skin(q, pals[i], oracle.morph(q, rates[i]), skin16)
            assert np.array_equal(p16[i].view(np.uint16), qp.astype(np.float16)[e2o16].view(np.uint16)), f"inst {i} f16 pos"
            gu.assert_bits_equal(n16[i], qn[e2o16], "f16 nrm")
        # one frame: host arrays (tile kernel, coalesced copy-out replaced by direct stores) and device-resident (frame kernel)
        ep, en = oracle.skin(m, pals[1], oracle.morph(m, rates[1]), skin)
        p1, n1 = dm.deform(rates[1], pals[1])
        gu.assert_bits_equal(p1, ep[e2o], "frame pos"); gu.assert_bits_equal(n1, en[e2o], "frame nrm")
        d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
        for n_inst in (1, ni):
            sa, sb = dm.out_sizes(api.OUT_SOA, n_inst)
            d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
            dm.deform_batched_raw(n_inst, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA,
                                  api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE)
            dm.sync()
            gp, gn = d_a.download((n_inst, nv, 3), np.float32), d_b.download((n_inst, nv, 3), np.float32)
            for i in range(n_inst):
                ep, en = oracle.skin(m, pals[i], oracle.morph(m, rates[i]), skin)
                gu.assert_bits_equal(gp[i], ep[e2o], f"device-resident x{n_inst} inst {i} pos")
                gu.assert_bits_equal(gn[i], en[e2o], f"device-resident x{n_inst} inst {i} nrm")
            d_a.free(); d_b.free()
        d_pal.free(); d_w.free()


@gpu
def test_tile_order_config3_crowd_full_size_sample(oracle, hip_lib):
    """BASELINE config 3 at its size in tile order (what bench.py reports as config3_tile_order_output): a strided sample of
    instances, every vertex, against the oracle."""
    m = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(m, (np.arange(ni) * 3) % 1801)
    rates = synth.morph_weights(m.nm, 30)[0]
    skin, vimg = oracle.normalize(m), None
    vimg = oracle.morph(m, rates)
    with DeformModel(m, tile_order=True) as dm:
        e2o, _ = dm.vertex_order()
        d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
        sa, sb = dm.out_sizes(api.OUT_SOA, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA,
                              api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED)
        dm.sync()
        for i in list(range(0, ni, 31)) + [ni - 1]:
            ep, en = oracle.skin(m, pals[i], vimg, skin)
            gu.assert_bits_equal(d_a.download((m.nv, 3), np.float32, offset=i * m.nv * 12), ep[e2o], f"instance {i} pos")
            gu.assert_bits_equal(d_b.download((m.nv, 3), np.float32, offset=i * m.nv * 12), en[e2o], f"instance {i} nrm")
        for b in (d_pal, d_w, d_a, d_b):
            b.free()


@gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MMDX_SOAK_SEEDS", "12"))))
def test_tile_order_and_fast_math_randomized_models(oracle, hip_lib, seed):
    """Random sizes / class mixes / bone windows / morph tables (duplicates, groups of groups, ignored types) -- the generator of
    tests/test_gpu_parity.py -- through both opt-in modes: tile order bit-identical to the oracle un-permuted (per-instance morphs,
    shared crowd, one frame), fast math within its stated tolerance."""
    from tests.test_gpu_parity import _random_model
    from tests.test_fast_math import close_nrm, close_pos
    rng = np.random.RandomState(31000 + seed)
    m = _random_model(rng)
    ni = int(rng.choice([1, 2, 5, 9, 33]))
    normalize = bool(rng.randint(2))
    rates = rng.choice([0.0, 1.0, 0.3, -0.2, 5e-8, 2.5], size=(ni, m.nm)).astype(np.float32)
    pals = synth.make_palettes(m, rng.randint(0, 500, ni))
    skin = oracle.normalize(m) if normalize else None
    with DeformModel(m, normalize=normalize, tile_order=True) as tm, DeformModel(m, normalize=normalize, fast_math=True) as fm:
        e2o, _ = tm.vertex_order()
        pos, nrm = tm.deform_batched(rates, pals)
        spos, snrm = tm.deform_batched(rates[0], pals, shared_weights=True)
        fpos, fnrm = fm.deform_batched(rates, pals)
        for i in range(ni):
            ep, en = oracle.skin(m, pals[i], oracle.morph(m, rates[i]), skin)
            gu.assert_bits_equal(pos[i], ep[e2o], f"seed {seed} tile order inst {i} pos")
            gu.assert_bits_equal(nrm[i], en[e2o], f"seed {seed} tile order inst {i} nrm")
            sp, sn = oracle.skin(m, pals[i], oracle.morph(m, rates[0]), skin)
            gu.assert_bits_equal(spos[i], sp[e2o], f"seed {seed} tile order shared inst {i} pos")
            gu.assert_bits_equal(snrm[i], sn[e2o], f"seed {seed} tile order shared inst {i} nrm")
            fin = np.isfinite(ep).all(axis=1) & np.isfinite(en).all(axis=1)
            assert close_pos(fpos[i][fin], ep[fin], 1.0 + float(np.abs(pals[i]).max())) and close_nrm(fnrm[i][fin], en[fin]), f"seed {seed} fast math inst {i}"
        p1, n1 = tm.deform(rates[ni - 1], pals[ni - 1])
        ep, en = oracle.skin(m, pals[ni - 1], oracle.morph(m, rates[ni - 1]), skin)
        gu.assert_bits_equal(p1, ep[e2o], "tile order frame pos")
        gu.assert_bits_equal(n1, en[e2o], "tile order frame nrm")
