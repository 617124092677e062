"""GPU: HIP-graph record / replay of the library's device work (mmdx_graph_*): a single-model frame, and the whole
motion -> poses -> palettes -> vertices frame, replayed after the inputs were updated in place; results bit-identical to
the directly executed calls.  Misuse (host operands, first-use allocation while recording) is rejected, never recorded."""
import numpy as np
import pytest

from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import synth, vmd
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer, device_count
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(hip_lib):
    assert device_count() >= 1


def test_graph_single_frames_replayed_with_new_inputs(oracle):
    m = synth.make_model(5000, 60, 8, 300, seed=31)
    frames = np.arange(5) * 11
    rates = synth.morph_weights(m.nm, frames)
    pals = synth.make_palettes(m, frames)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
    with DeformModel(m) as dm:
        d_pal, d_w = DeviceBuffer.from_numpy(pals[0]), DeviceBuffer.from_numpy(rates[0])
        sa, sb = dm.out_sizes(api.OUT_SOA, 1)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        dm.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)    # sizes the scratch buffers
        dm.sync()
        dm.graph_begin()
        dm.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
        g = dm.graph_end()
        skin = oracle.normalize(m)
        for k in range(5):
            d_pal.upload(pals[k]); d_w.upload(rates[k])
            d_a.memset(0xFF); d_b.memset(0xFF)
            g.launch()
            dm.sync()
            ep, en = oracle.skin(m, pals[k], oracle.morph(m, rates[k]), skin)
            gu.assert_bits_equal(d_a.download((m.nv, 3), np.float32), ep, f"replay {k} pos")
            gu.assert_bits_equal(d_b.download((m.nv, 3), np.float32), en, f"replay {k} nrm")
        g.close()
        # misuse: host operands while recording are rejected; the recording still ends cleanly
        dm.graph_begin()
        with pytest.raises(api.MmdxError, match="device memory"):
            dm.deform(rates[0], pals[0])
        dm.graph_end().close()
        for b in (d_pal, d_w, d_a, d_b):
            b.free()


def test_graph_whole_frame_motion_to_vertices(oracle):
    """bone tracks -> poses -> palettes -> crowd deform (morph pass + skinning): five launches recorded once, replayed
    for three sets of per-instance frame numbers written into the same device buffer."""
    m = synth.make_model(3000, 40, 6, 200, seed=77)
    ni = 24
    names = [f"b{i}" for i in range(m.nb)]
    v = vmd.Vmd(vmd.write_vmd(synth.make_bone_keys(names, 5, keys_per=8, span=120), []))
    bm = v.bind_bones(names)
    sk = vmd.Skeleton(m.bone_pos, np.asarray(m.bone_parent, np.int32))
    shared = synth.morph_weights(m.nm, 9)[0]
    with DeformModel(m) as dm:
        d_fr = DeviceBuffer.from_numpy(np.zeros(ni, np.uint32))
        d_pose, d_pal = DeviceBuffer(ni * m.nb * 32), DeviceBuffer(ni * m.nb * 64)
        d_w = DeviceBuffer.from_numpy(shared)
        sa, sb = dm.out_sizes(api.OUT_SOA, ni)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED

        def frame():
            bm.eval_device(ni, d_fr.ptr, d_pose.ptr, dm)
            sk.solve_device(ni, d_pose.ptr, d_pal.ptr, dm)
            dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
        frame()
        dm.sync()
        dm.graph_begin()
        frame()
        g = dm.graph_end()
        for rep in range(3):
            fr = ((np.arange(ni) * 7 + rep * 13) % 120).astype(np.uint32)
            d_fr.upload(fr)
            frame()
            dm.sync()
            want_a, want_b = d_a.download((ni, m.nv, 3), np.float32), d_b.download((ni, m.nv, 3), np.float32)
            d_a.memset(0); d_b.memset(0)
            g.launch()
            dm.sync()
            gu.assert_bits_equal(d_a.download((ni, m.nv, 3), np.float32), want_a, f"rep {rep} pos")
            gu.assert_bits_equal(d_b.download((ni, m.nv, 3), np.float32), want_b, f"rep {rep} nrm")
        g.close()
        for b in (d_fr, d_pose, d_pal, d_w, d_a, d_b):
            b.free()
    sk.close()


def test_graph_frame_with_the_one_call_palette_producer(oracle):
    """mmdx_skeleton_solve_motion (bone tracks -> palettes, one launch) + the frame kernel recorded as one graph: replays for new
    frame numbers and rates reproduce the eager calls bit for bit; before its first eager run a recording is rejected."""
    m = synth.make_model(3000, 40, 6, 200, seed=78)
    names = [f"b{i}" for i in range(m.nb)]
    v = vmd.Vmd(vmd.write_vmd(synth.make_bone_keys(names, 6, keys_per=8, span=120), []))
    bm = v.bind_bones(names)
    sk = vmd.Skeleton(m.bone_pos, np.asarray(m.bone_parent, np.int32))
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
    with DeformModel(m) as dm:
        d_fr = DeviceBuffer.from_numpy(np.zeros(1, np.uint32))
        d_pal, d_w = DeviceBuffer(m.nb * 64), DeviceBuffer.from_numpy(synth.morph_weights(m.nm, 0)[0])
        sa, sb = dm.out_sizes(api.OUT_SOA, 1)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)

        def frame():
            sk.solve_motion_device(bm, 1, d_fr.ptr, d_pal.ptr, dm)
            dm.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
        dm.graph_begin()
        with pytest.raises(api.MmdxError):               # the motion / skeleton tables are not on the device yet
            sk.solve_motion_device(bm, 1, d_fr.ptr, d_pal.ptr, dm)
        dm.graph_end().close()
        frame()
        dm.sync()
        dm.graph_begin()
        frame()
        g = dm.graph_end()
        for rep in range(3):
            d_fr.upload(np.asarray([17 * rep + 3], np.uint32)); d_w.upload(synth.morph_weights(m.nm, 40 * rep)[0])
            frame()
            dm.sync()
            want_a, want_b = d_a.download((m.nv, 3), np.float32), d_b.download((m.nv, 3), np.float32)
            d_a.memset(0); d_b.memset(0)
            g.launch()
            dm.sync()
            gu.assert_bits_equal(d_a.download((m.nv, 3), np.float32), want_a, f"rep {rep} pos")
            gu.assert_bits_equal(d_b.download((m.nv, 3), np.float32), want_b, f"rep {rep} nrm")
        g.close()
        for b in (d_fr, d_pal, d_w, d_a, d_b):
            b.free()
    sk.close()
