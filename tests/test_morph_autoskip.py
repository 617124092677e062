"""GPU: the shared morph pass of a crowd is skipped BY THE LIBRARY when the shared morph rates did not change
(include/mmdx.h, MMDX_MORPH_UNCHANGED: the flag is the caller's promise; without it the library compares).  The reference's
vertex_images_ depends on morph_rates_ only (L/motion/poser_impl.inl:362-386), so skipping is exact -- every case below is
checked bit for bit against the oracle, and the pass counters (mmdx_debug_morph_pass_stats) say what really happened:
  * rates in device memory: the morph pass compares on the device and skips its walk;
  * rates in host memory: the host compares and skips the launch;
  * a CHANGED rate (one bit) re-runs the walk;
  * every other writer of the kept positions (small-crowd kernel, graph replays, device calls between host calls) voids the record."""
import os

import numpy as np
import pytest

from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import synth
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer, device_count
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

DEV = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(hip_lib):
    assert device_count() >= 1


@pytest.fixture(scope="module")
def crowd():
    m = synth.make_model(6100, 60, 12, 400, seed=404)
    ni = 12                                              # > 8: the crowd takes the separate morph pass
    pals = synth.make_palettes(m, np.arange(ni) * 3)
    return m, ni, pals


def expect(oracle, m, rates, pals, skin, insts):
    vimg = oracle.morph(m, rates)
    return {i: oracle.skin(m, pals[i], vimg, skin) for i in insts}


def check_dev(d_a, d_b, m, want, what):
    row = m.nv * 12
    for i, (ep, en) in want.items():
        gu.assert_bits_equal(d_a.download((m.nv, 3), np.float32, offset=i * row), ep, f"{what}: inst {i} pos")
        gu.assert_bits_equal(d_b.download((m.nv, 3), np.float32, offset=i * row), en, f"{what}: inst {i} nrm")


def test_device_rates_unchanged_skip_the_walk_changed_rates_rerun_it(oracle, crowd):
    m, ni, pals = crowd
    skin = oracle.normalize(m)
    r0 = synth.morph_weights(m.nm, 30)[0]
    r1 = r0.copy()
    r1[3] = np.nextafter(r1[3] if r1[3] > 0.5 else np.float32(0.7), np.float32(2))      # one bit of one rate
    insts = (0, 5, ni - 1)
    with DeformModel(m) as dm:
        d_w, d_pal = DeviceBuffer.from_numpy(r0), DeviceBuffer.from_numpy(pals)
        sa, sb = dm.out_sizes(api.OUT_SOA, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)

        def run():
            d_a.memset(0xFF); d_b.memset(0xFF)
            dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, DEV)
            dm.sync()
        run(); check_dev(d_a, d_b, m, expect(oracle, m, r0, pals, skin, insts), "first call")
        assert dm.morph_pass_stats() == (1, 0, 0)
        run(); run()
        check_dev(d_a, d_b, m, expect(oracle, m, r0, pals, skin, insts), "third call, rates unchanged")
        assert dm.morph_pass_stats() == (1, 2, 0)                       # two launches found nothing to do
        d_w.upload(r1)                                                   # the caller changes a rate in place: same pointer
        run(); check_dev(d_a, d_b, m, expect(oracle, m, r1, pals, skin, insts), "changed rate")
        assert dm.morph_pass_stats() == (2, 2, 0)
        run()
        assert dm.morph_pass_stats() == (2, 3, 0)
        d_w.upload(r0)                                                   # ... and back
        run(); check_dev(d_a, d_b, m, expect(oracle, m, r0, pals, skin, insts), "changed back")
        assert dm.morph_pass_stats() == (3, 3, 0)
        # a small crowd (<= 8 instances) gathers inside its deform kernel and overwrites the kept positions with ITS rates:
        # the record is void afterwards, the next crowd pass walks even though its rates equal the recorded ones
        d_w1 = DeviceBuffer.from_numpy(r1)
        dm.deform_batched_raw(4, d_w1.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, DEV)
        dm.sync()
        check_dev(d_a, d_b, m, expect(oracle, m, r1, pals, skin, (0, 3)), "small crowd, other rates")
        run(); check_dev(d_a, d_b, m, expect(oracle, m, r0, pals, skin, insts), "after the small crowd")
        assert dm.morph_pass_stats() == (4, 3, 0)
        # the caller's promise still skips the launch altogether
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, DEV | api.MORPH_UNCHANGED)
        dm.sync()
        check_dev(d_a, d_b, m, expect(oracle, m, r0, pals, skin, insts), "explicit flag")
        assert dm.morph_pass_stats() == (4, 3, 0)
        for b in (d_w, d_w1, d_pal, d_a, d_b):
            b.free()


def test_host_rates_unchanged_skip_the_launch_and_every_other_writer_voids_the_record(oracle, crowd):
    m, ni, pals = crowd
    skin = oracle.normalize(m)
    r0, r1 = synth.morph_weights(m.nm, 30)[0], synth.morph_weights(m.nm, 47)[0]
    insts = range(ni)

    def check(out, rates, what):
        want = expect(oracle, m, rates, pals, skin, insts)
        for i in insts:
            gu.assert_bits_equal(out[0][i], want[i][0], f"{what}: inst {i} pos")
            gu.assert_bits_equal(out[1][i], want[i][1], f"{what}: inst {i} nrm")
    with DeformModel(m) as dm:
        check(dm.deform_batched(r0, pals, shared_weights=True), r0, "first")
        check(dm.deform_batched(r0.copy(), pals, shared_weights=True), r0, "same values, other buffer")
        assert dm.morph_pass_stats() == (1, 0, 1)
        check(dm.deform_batched(r1, pals, shared_weights=True), r1, "changed")
        assert dm.morph_pass_stats() == (2, 0, 1)
        check(dm.deform_batched(r1, pals, shared_weights=True), r1, "unchanged again")
        assert dm.morph_pass_stats() == (2, 0, 2)
        # a device-rates call in between recomputes the kept positions from rates the host never saw
        d_w, d_pal = DeviceBuffer.from_numpy(r0), DeviceBuffer.from_numpy(pals)
        sa, sb = dm.out_sizes(api.OUT_SOA, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, DEV)
        dm.sync()
        assert dm.morph_pass_stats() == (3, 0, 2)
        check(dm.deform_batched(r1, pals, shared_weights=True), r1, "host call after a device call")
        assert dm.morph_pass_stats() == (4, 0, 2)
        # ... and so does a graph replay (recorded with device rates r0)
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, DEV)
        dm.sync()
        dm.graph_begin()
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, DEV)
        g = dm.graph_end()
        check(dm.deform_batched(r1, pals, shared_weights=True), r1, "host call before the replay")
        walks = dm.morph_pass_stats()[0]
        g.launch(); dm.sync()
        check_dev(d_a, d_b, m, expect(oracle, m, r0, pals, skin, (0, ni - 1)), "replay")
        assert dm.morph_pass_stats()[0] == walks + 1                     # the replay walked (rates differ from the host call's)
        g.launch(); dm.sync()
        assert dm.morph_pass_stats()[0] == walks + 1                     # ... and its second replay did not
        check(dm.deform_batched(r1, pals, shared_weights=True), r1, "host call after the replays")
        g.close()
        for b in (d_w, d_pal, d_a, d_b):
            b.free()


def test_models_past_the_fused_flatten_limit_and_f16_models(oracle):
    """More than 8 192 slots: the morph pass runs behind a separate flatten launch and keeps no record of its rates (every call
    walks, results exact); an f16-position model takes the recorded path like an f32 one."""
    ni = 12
    big = synth.make_model(900, 12, 8300, 2, seed=77)                    # 8 300 vertex morphs of 2 entries each
    pals = synth.make_palettes(big, np.arange(ni))
    r = synth.morph_weights(big.nm, 9)[0]
    skin = oracle.normalize(big)
    with DeformModel(big) as dm:
        for k in range(3):
            pos, nrm = dm.deform_batched(r, pals, shared_weights=True)
            want = expect(oracle, big, r, pals, skin, (0, ni - 1))
            for i, (ep, en) in want.items():
                gu.assert_bits_equal(pos[i], ep, f"call {k} inst {i} pos")
                gu.assert_bits_equal(nrm[i], en, f"call {k} inst {i} nrm")
        walks, dskips, hskips = dm.morph_pass_stats()
        assert dskips == 0 and hskips == 2          # the HOST comparison still works (it needs no device record); no device skip
    m = synth.make_model(3000, 30, 10, 200, seed=78)
    q = m.copy()
    q.positions = m.positions.astype(np.float16).astype(np.float32)
    q.morph_value = m.morph_value.astype(np.float16).astype(np.float32)
    pals = synth.make_palettes(m, np.arange(ni))
    r = synth.morph_weights(m.nm, 21)[0]
    skin = oracle.normalize(q)
    with DeformModel(m, f16_positions=True) as dm:
        d_w, d_pal = DeviceBuffer.from_numpy(r), DeviceBuffer.from_numpy(pals)
        sa, sb = dm.out_sizes(api.OUT_SOA_POS16, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        for k in range(3):
            d_a.memset(0xFF); d_b.memset(0xFF)
            dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA_POS16, DEV)
            dm.sync()
        assert dm.morph_pass_stats() == (1, 2, 0)
        want = expect(oracle, q, r, pals, skin, (0, ni - 1))
        for i, (ep, en) in want.items():
            got16 = d_a.download((m.nv, 3), np.float16, offset=i * m.nv * 6)
            assert np.array_equal(got16.view(np.uint16), ep.astype(np.float16).view(np.uint16)), f"f16 inst {i} pos"
            gu.assert_bits_equal(d_b.download((m.nv, 3), np.float32, offset=i * m.nv * 12), en, f"f16 inst {i} nrm")
        for b in (d_w, d_pal, d_a, d_b):
            b.free()


def test_autoskip_can_be_switched_off_for_ab_runs(oracle, crowd):
    m, ni, pals = crowd
    r0 = synth.morph_weights(m.nm, 30)[0]
    os.environ["MMDX_MORPH_AUTOSKIP"] = "0"
    api.lib().mmdx_debug_reload_env()
    try:
        with DeformModel(m) as dm:
            for _ in range(3):
                dm.deform_batched(r0, pals, shared_weights=True)
            assert dm.morph_pass_stats() == (0, 0, 0)                   # no record is kept at all: every call walks
    finally:
        del os.environ["MMDX_MORPH_AUTOSKIP"]
        api.lib().mmdx_debug_reload_env()


def test_destroying_another_model_on_a_recording_thread_is_deferred(oracle):
    """model B destroyed while model A records on the same thread -- no stream / event / host-memory call may
    happen until A's recording has ended; the recording stays valid and replays."""
    m = synth.make_model(3000, 40, 4, 100, seed=5)
    frames = np.arange(2) * 7
    rates, pals = synth.morph_weights(m.nm, frames), synth.make_palettes(m, frames)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
    skin = oracle.normalize(m)
    a, b = DeformModel(m), DeformModel(m)
    b.deform(rates[1], pals[1])                                          # B owns bounce buffers, events, a stream
    d_pal, d_w = DeviceBuffer.from_numpy(pals[0]), DeviceBuffer.from_numpy(rates[0])
    sa, sb = a.out_sizes(api.OUT_SOA, 1)
    d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
    a.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
    a.sync()
    a.graph_begin()
    a.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
    b.close()                                                            # in the middle of A's recording
    g = a.graph_end()
    d_a.memset(0xFF); d_b.memset(0xFF)
    g.launch(); a.sync()
    ep, en = oracle.skin(m, pals[0], oracle.morph(m, rates[0]), skin)
    gu.assert_bits_equal(d_a.download((m.nv, 3), np.float32), ep, "replay pos")
    gu.assert_bits_equal(d_b.download((m.nv, 3), np.float32), en, "replay nrm")
    g.close()
    a.close()
    for buf in (d_pal, d_w, d_a, d_b):
        buf.free()
