"""Regression tests for the round-2 advisor findings:
 * tile-order outputs + per-instance morph weights when one workgroup serves SEVERAL packs of instances (the packs' slot
   weights in LDS were replaced without a barrier: csrc/kernels.hip, kMorphFused4 loop);
 * MMDX_MORPH_UNCHANGED after a single-frame call that ran the tile kernel (the frame overwrote the crowd's kept positions);
 * recorded graphs vs. later changes to the handles they were recorded from (growing a scratch buffer, destroying a handle,
   ending a recording on another thread);
 * unknown flag bits are rejected, the ABI version moved.
CPU part: flag validation through host-only models.  Everything else needs the GPU."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd import synth, vmd
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer, device_count
from tests import golden_util as gu

gpu = pytest.mark.gpu


# ---- CPU -------------------------------------------------------------------------------------------------------------
def test_unknown_create_flags_are_rejected(hip_lib):
    m = synth.make_model(100, 8, 2, 10, seed=1)
    with DeformModel(m, host_only=True):
        pass
    d = api.ModelDesc()
    d.struct_size = C.sizeof(api.ModelDesc)
    d.flags = api.CREATE_HOST_ONLY | (1 << 9)
    h = C.c_void_p()
    st = hip_lib.mmdx_model_create(C.byref(d), C.byref(h))
    assert st == 1 and b"unknown bits" in hip_lib.mmdx_last_error_string()


def test_abi_version_is_3(hip_lib):
    assert hip_lib.mmdx_abi_version() == api.ABI_VERSION == 3


# ---- GPU -------------------------------------------------------------------------------------------------------------
def _ragged_model(nv, seed):
    """Morph rows of very different lengths inside one tile: a few vertices are hit by every morph, most by none or one, so
    the waves of a workgroup leave their walk at very different times."""
    m = synth.make_model(nv, 48, 24, 40, seed=seed)
    rng = np.random.RandomState(seed)
    hot = rng.choice(nv, max(nv // 40, 1), replace=False).astype(np.uint32)   # vertices every morph touches
    idx, val, off = [], [], [0]
    for k in range(m.nm):
        own = m.morph_index[m.morph_off[k]:m.morph_off[k + 1]]
        cur = np.concatenate([hot, own[~np.isin(own, hot)]]).astype(np.uint32)
        idx.append(cur)
        val.append(rng.uniform(-0.5, 0.5, (cur.size, 3)).astype(np.float32))
        off.append(off[-1] + cur.size)
    m.morph_index = np.concatenate(idx)
    m.morph_value = np.concatenate(val)
    m.morph_off = np.asarray(off, np.uint32)
    return m


@gpu
@pytest.mark.parametrize("threads", [512, 256])
def test_tile_order_per_instance_morphs_several_packs_per_workgroup(oracle, hip_lib, threads):
    """MMDX_GROUP=16 forces 2 packs (512 threads: 8 instances per pack) / 4 packs (256 threads: 4 per pack) per workgroup; the
    rates of consecutive packs differ in every slot, rows are ragged.  Bit-exact against the oracle, three repetitions."""
    assert device_count() >= 1
    nv, ni = 2048 + 77, 48
    m = _ragged_model(nv, 9100)
    rng = np.random.RandomState(5)
    rates = rng.uniform(0.05, 1.0, (ni, m.nm)).astype(np.float32)
    rates[8:16] *= -1.0                      # a whole pack of skipped slots next to packs that apply everything
    rates[20:24] = 0.0
    pals = synth.make_palettes(m, np.arange(ni) * 3)
    skin = oracle.normalize(m)
    want = [oracle.skin(m, pals[i], oracle.morph(m, rates[i]), skin) for i in range(ni)]
    old = {k: os.environ.get(k) for k in ("MMDX_GROUP", "MMDX_THREADS")}
    os.environ["MMDX_GROUP"] = "16"
    os.environ["MMDX_THREADS"] = str(threads)
    hip_lib.mmdx_debug_reload_env()
    try:
        for tile_order in (True, False):
            with DeformModel(m, tile_order=tile_order) as dm:
                e2o = dm.vertex_order()[0] if tile_order else np.arange(nv)
                for rep in range(3):
                    pos, nrm = dm.deform_batched(rates, pals)
                    for i in range(ni):
                        gu.assert_bits_equal(pos[i], want[i][0][e2o], f"tile_order={tile_order} rep {rep} inst {i} pos")
                        gu.assert_bits_equal(nrm[i], want[i][1][e2o], f"tile_order={tile_order} rep {rep} inst {i} nrm")
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        hip_lib.mmdx_debug_reload_env()


@gpu
@pytest.mark.parametrize("frame_kernel", ["1", "0"])
def test_morph_unchanged_survives_single_frame_calls(oracle, hip_lib, frame_kernel):
    """crowd(shared W1) -> single frames with W2 (host outputs: tile kernel; device outputs: frame kernel or, with
    MMDX_FRAME_KERNEL=0, the tile kernel) -> crowd(MMDX_MORPH_UNCHANGED) must still be W1's crowd."""
    m = synth.make_model(3000, 40, 9, 400, seed=77)
    ni = 12
    w1, w2 = synth.morph_weights(m.nm, 10)[0], synth.morph_weights(m.nm, 55)[0]
    pals = synth.make_palettes(m, np.arange(ni) * 5)
    skin = oracle.normalize(m)
    vimg1 = oracle.morph(m, w1)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
    old = os.environ.get("MMDX_FRAME_KERNEL")
    os.environ["MMDX_FRAME_KERNEL"] = frame_kernel
    hip_lib.mmdx_debug_reload_env()
    try:
        with DeformModel(m) as dm:
            d_pal, d_w1, d_w2 = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(w1), DeviceBuffer.from_numpy(w2)
            sa, sb = dm.out_sizes(api.OUT_SOA, ni)
            d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
            dm.deform_batched_raw(ni, d_w1.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags | api.WEIGHTS_SHARED)
            dm.sync()
            # single frames with other rates, every route
            p2, n2 = dm.deform(w2, pals[3])                                               # host outputs
            ep, en = oracle.skin(m, pals[3], oracle.morph(m, w2), skin)
            gu.assert_bits_equal(p2, ep, "frame pos"); gu.assert_bits_equal(n2, en, "frame nrm")
            s1a, s1b = DeviceBuffer(m.nv * 12), DeviceBuffer(m.nv * 12)
            dm.deform_batched_raw(1, d_w2.ptr, d_pal.ptr, s1a.ptr, s1b.ptr, api.OUT_SOA, flags)   # device outputs
            dm.sync()
            d_a.memset(0); d_b.memset(0)
            dm.deform_batched_raw(ni, None, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA,
                                  flags | api.WEIGHTS_SHARED | api.MORPH_UNCHANGED)
            dm.sync()
            gp, gn = d_a.download((ni, m.nv, 3), np.float32), d_b.download((ni, m.nv, 3), np.float32)
            for i in range(ni):
                ep, en = oracle.skin(m, pals[i], vimg1, skin)
                gu.assert_bits_equal(gp[i], ep, f"unchanged crowd inst {i} pos")
                gu.assert_bits_equal(gn[i], en, f"unchanged crowd inst {i} nrm")
            for b in (d_pal, d_w1, d_w2, d_a, d_b, s1a, s1b):
                b.free()
    finally:
        if old is None:
            os.environ.pop("MMDX_FRAME_KERNEL", None)
        else:
            os.environ["MMDX_FRAME_KERNEL"] = old
        hip_lib.mmdx_debug_reload_env()


@gpu
def test_unknown_deform_flags_are_rejected(hip_lib):
    m = synth.make_model(600, 12, 3, 50, seed=3)
    with DeformModel(m) as dm:
        d_pal, d_w = DeviceBuffer.from_numpy(synth.make_palettes(m, [0])), DeviceBuffer.from_numpy(synth.morph_weights(m.nm, 1)[0])
        d_a, d_b = DeviceBuffer(m.nv * 12), DeviceBuffer(m.nv * 12)
        with pytest.raises(api.MmdxError, match="unknown bits"):
            dm.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, 7 | (1 << 12))
        for b in (d_pal, d_w, d_a, d_b):
            b.free()


@gpu
def test_graph_pins_its_handles(oracle, hip_lib):
    """While a recorded graph is alive: a call that would have to GROW a scratch buffer the graph holds fails cleanly (and the
    graph still replays correctly); destroying a handle it was recorded from invalidates it (launch fails, no replay into freed
    memory); ending a recording on another thread is refused and the right thread can still end it."""
    m = synth.make_model(2500, 30, 6, 200, seed=91)
    ni = 8
    names = [f"b{i}" for i in range(m.nb)]
    v = vmd.Vmd(vmd.write_vmd(synth.make_bone_keys(names, 5, keys_per=6, span=100), []))
    bm = v.bind_bones(names)
    sk = vmd.Skeleton(m.bone_pos, np.asarray(m.bone_parent, np.int32))
    rates = synth.morph_weights(m.nm, np.arange(4 * ni))                    # per-instance rates: wslot is a scratch buffer
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
    dm = DeformModel(m)
    d_fr = DeviceBuffer.from_numpy(((np.arange(4 * ni) * 7) % 100).astype(np.uint32))
    d_pose, d_pal = DeviceBuffer(4 * ni * m.nb * 32), DeviceBuffer(4 * ni * m.nb * 64)
    d_w = DeviceBuffer.from_numpy(rates)
    sa, sb = dm.out_sizes(api.OUT_SOA, 4 * ni)
    d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)

    def frame(n):
        bm.eval_device(n, d_fr.ptr, d_pose.ptr, dm)
        sk.solve_device(n, d_pose.ptr, d_pal.ptr, dm)
        dm.deform_batched_raw(n, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
    frame(ni)
    dm.sync()
    want_a, want_b = d_a.download((ni, m.nv, 3), np.float32), d_b.download((ni, m.nv, 3), np.float32)
    # (1) ending on another thread is refused; this thread ends it
    dm.graph_begin()
    frame(ni)
    err = []
    t = threading.Thread(target=lambda: err.append(_try(lambda: dm.graph_end())))
    t.start(); t.join()
    assert isinstance(err[0], api.MmdxError) and "thread" in str(err[0])
    # ... and so is a RECORDED call from another thread (the per-thread "nothing may allocate" guards would not see it)
    for call in (lambda: dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags),
                 lambda: sk.solve_device(ni, d_pose.ptr, d_pal.ptr, dm)):
        err = []
        t = threading.Thread(target=lambda c=call: err.append(_try(c)))
        t.start(); t.join()
        assert isinstance(err[0], api.MmdxError) and "another thread" in str(err[0]), err
    g = dm.graph_end()
    # (2) a larger crowd would have to grow the model's slot-weight scratch (which the graph holds): refused, with a reason
    with pytest.raises(api.MmdxError, match="recorded"):
        dm.deform_batched_raw(4 * ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
    d_a.memset(0); d_b.memset(0)
    g.launch(); dm.sync()
    gu.assert_bits_equal(d_a.download((ni, m.nv, 3), np.float32), want_a, "replay after the refused call: pos")
    gu.assert_bits_equal(d_b.download((ni, m.nv, 3), np.float32), want_b, "replay after the refused call: nrm")
    # ... and once the graph is gone the same call goes through
    g.close()
    dm.deform_batched_raw(4 * ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
    dm.sync()
    # (3) destroying a handle the graph was recorded from invalidates the graph
    dm.graph_begin()
    frame(ni)
    g2 = dm.graph_end()
    g2.launch(); dm.sync()
    sk.close()                                       # the skeleton took part in the recording
    with pytest.raises(api.MmdxError, match="destroyed"):
        g2.launch()
    g2.close()
    # (4) a handle destroyed DURING a recording that used it: the recording cannot become a graph
    sk2 = vmd.Skeleton(m.bone_pos, np.asarray(m.bone_parent, np.int32))
    sk2.solve_device(ni, d_pose.ptr, d_pal.ptr, dm)
    dm.sync()
    dm.graph_begin()
    sk2.solve_device(ni, d_pose.ptr, d_pal.ptr, dm)
    sk2.close()
    with pytest.raises(api.MmdxError, match="destroyed before"):
        dm.graph_end()
    dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)    # the model is usable again
    dm.sync()
    dm.graph_begin()
    dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
    g3 = dm.graph_end()
    dm.close()                                       # the model itself
    with pytest.raises(api.MmdxError, match="destroyed"):
        g3.launch()
    g3.close()
    # (5) a model destroyed in the middle of its own recording leaves no recording state behind on this thread
    dm2 = DeformModel(m)
    dm2.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
    dm2.sync()
    dm2.graph_begin()
    dm2.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
    dm2.close()
    with DeformModel(m) as dm3:                      # allocating calls work again (nothing thinks this thread still records)
        pos, _ = dm3.deform_batched(rates[:ni], synth.make_palettes(m, np.arange(ni)))
        assert np.isfinite(pos).all()
    for b in (d_fr, d_pose, d_pal, d_w, d_a, d_b):
        b.free()


def _try(f):
    try:
        return f()
    except Exception as e:      # noqa: BLE001 -- handed to the asserting thread
        return e
