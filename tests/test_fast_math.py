"""GPU (MI355X): MMDX_CREATE_FAST_MATH, the opt-in contracted-arithmetic build of the deform kernels (csrc/kernels_fast.hip).
Not bit-exact by design; the tolerance stated in include/mmdx.h is written HERE and checked against the oracle for every call
form: per position component |x - x_ref| <= 1e-5 * (1 + |x_ref|) (measured 2.4e-6 on the benchmark models: a handful of
binary32 ulps, from fusing each multiply with the add that consumes it), per normal component |n - n_ref| <= 2e-6 (measured
1.8e-7); positions rounded to binary16 (config 5's layout) within that plus one binary16 ulp.  Models created WITHOUT the flag stay
bit-identical to the oracle (tests/test_gpu_parity.py) -- also asserted here side by side."""
import numpy as np
import pytest

from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import synth
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer, device_count
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

POS_TOL, NRM_TOL = 1e-5, 2e-6


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(hip_lib):
    assert device_count() >= 1


def close_pos(a, ref, scale=1.0):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return bool(np.all(np.abs(a - ref) <= POS_TOL * (scale + np.abs(ref))))


def close_nrm(a, ref):
    return bool(np.all(np.abs(np.asarray(a, np.float64) - np.asarray(ref, np.float64)) <= NRM_TOL))


@pytest.mark.parametrize("name", ["g07_vertex_morph", "g08_group_morph", "g11_solved_palette", "g12_mini_model"])
def test_fast_math_golden_fixtures_within_tolerance(name):
    """libmmd's own outputs (golden vectors): single frames, SoA and the 32-byte vertex."""
    m, exp = gu.load(name)
    with DeformModel(m, normalize=exp["normalize"], fast_math=True) as dm:
        for f in range(exp["rates"].shape[0]):
            pos, nrm = dm.deform(exp["rates"][f], exp["palette"][f])
            assert close_pos(pos, exp["expect_pos"][f]) and close_nrm(nrm, exp["expect_nrm"][f]), f"{name}[{f}]"
            v32 = dm.deform_vertex32(exp["rates"][f], exp["palette"][f], 0.1).reshape(-1, 8)
            e32 = exp["expect_v32"][f].reshape(-1, 8)
            assert close_pos(v32[:, :3], e32[:, :3], 0.1) and close_nrm(v32[:, 3:6], e32[:, 3:6])
            assert np.array_equal(v32[:, 6:].view(np.uint32), e32[:, 6:].view(np.uint32))        # uv is a copy


@pytest.mark.parametrize("nv,ni", [(63, 3), (1000, 9), (4099, 17)])
def test_fast_math_every_call_form_within_tolerance(oracle, nv, ni):
    """Per-instance morphs (8 instances per walk), shared-morph crowd (separate morph pass for ni > 8, gathered in the kernel
    otherwise), single frames host-side and device-resident (frame kernel), the f16-position layout -- all within the stated
    tolerance of the oracle; the same calls on a model created without the flag are bit-identical to it."""
    m = synth.make_model(nv, 40, 9, min(300, nv // 2), seed=7700 + nv)
    q = m.copy()
    q.positions = m.positions.astype(np.float16).astype(np.float32)
    q.morph_value = m.morph_value.astype(np.float16).astype(np.float32)
    rates = synth.morph_weights(m.nm, np.arange(ni) * 7 + 3)
    rates[:, 0] = 0.0; rates[:, 1] = np.where(np.arange(ni) % 2 == 0, 5e-8, 1.0)
    pals = synth.make_palettes(m, np.arange(ni) * 5)
    skin = oracle.normalize(m)
    with DeformModel(m, fast_math=True) as fm, DeformModel(m) as em, DeformModel(m, f16_positions=True, fast_math=True) as fm16:
        pos, nrm = fm.deform_batched(rates, pals)
        xpos, xnrm = em.deform_batched(rates, pals)
        spos, snrm = fm.deform_batched(rates[0], pals, shared_weights=True)
        p16, n16 = fm16.deform_batched(rates, pals, layout=api.OUT_SOA_POS16)
        differs = 0
        for i in range(ni):
            ep, en = oracle.skin(m, pals[i], oracle.morph(m, rates[i]), skin)
            assert close_pos(pos[i], ep) and close_nrm(nrm[i], en), f"per-instance morphs, instance {i}"
            gu.assert_bits_equal(xpos[i], ep, "default model stays bit-exact"); gu.assert_bits_equal(xnrm[i], en, "default nrm")
            differs += int((pos[i].view(np.uint32) != ep.view(np.uint32)).sum())
            sp, sn = oracle.skin(m, pals[i], oracle.morph(m, rates[0]), skin)
            assert close_pos(spos[i], sp) and close_nrm(snrm[i], sn), f"shared crowd, instance {i}"
            qp, qn = oracle.skin(q, pals[i], oracle.morph(q, rates[i]), oracle.normalize(q))
            ref16 = qp.astype(np.float16).astype(np.float64)
            # the f32 tolerance, then one rounding to binary16 (an ulp is at most 2^-10 |x|, 2^-24 for subnormals)
            ulp16 = POS_TOL * (1 + np.abs(ref16)) + np.abs(ref16) * 2.0 ** -10 + 2.0 ** -24
            assert np.all(np.abs(p16[i].astype(np.float64) - ref16) <= ulp16), f"f16 positions, instance {i}"
            assert close_nrm(n16[i], qn)
        assert differs > 0 or nv < 100, "the contracted build is expected to differ from the oracle in the last place somewhere"
        # one frame: host arrays (tile kernel) and device-resident (frame kernel)
        ep, en = oracle.skin(m, pals[1], oracle.morph(m, rates[1]), skin)
        p1, n1 = fm.deform(rates[1], pals[1])
        assert close_pos(p1, ep) and close_nrm(n1, en)
        d_pal, d_w = DeviceBuffer.from_numpy(pals[1:2]), DeviceBuffer.from_numpy(rates[1:2])
        sa, sb = fm.out_sizes(api.OUT_SOA, 1)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        fm.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA,
                              api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE)
        fm.sync()
        assert close_pos(d_a.download((nv, 3), np.float32), ep) and close_nrm(d_b.download((nv, 3), np.float32), en)
        for b in (d_pal, d_w, d_a, d_b):
            b.free()


def test_fast_math_config3_crowd_sample_within_tolerance(oracle):
    """BASELINE config 3 at its size with the flag: a strided sample of 40 instances of the 1 024 against the oracle."""
    m = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(m, (np.arange(ni) * 3) % 1801)
    rates = synth.morph_weights(m.nm, 30)[0]
    skin = oracle.normalize(m)
    vimg = oracle.morph(m, rates)
    with DeformModel(m, fast_math=True) as dm:
        d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
        sa, sb = dm.out_sizes(api.OUT_SOA, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA,
                              api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED)
        dm.sync()
        for i in list(range(0, ni, 27)) + [ni - 1]:
            ep, en = oracle.skin(m, pals[i], vimg, skin)
            gp = d_a.download((m.nv, 3), np.float32, offset=i * m.nv * 12)
            gn = d_b.download((m.nv, 3), np.float32, offset=i * m.nv * 12)
            assert close_pos(gp, ep) and close_nrm(gn, en), f"instance {i}"
        for b in (d_pal, d_w, d_a, d_b):
            b.free()
