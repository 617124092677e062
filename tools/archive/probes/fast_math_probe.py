#!/usr/bin/env python3
"""MMDX_CREATE_FAST_MATH: how far the contracted kernels' results are from the oracle's (bit-exact default next to them), and what
the contraction buys: config-3 crowd step, per-instance-morph workloads, single frames.  python tools/archive/probes/fast_math_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402


def errors():
    orc = Oracle()
    for name in ("config1_20k", "config2_50k"):
        m = synth.make_config(name)
        skin = orc.normalize(m)
        ni = 12
        fr = np.arange(ni) * 37
        rates, pals = synth.morph_weights(m.nm, fr), synth.make_palettes(m, fr)
        with DeformModel(m, fast_math=True) as dm:
            pos, nrm = dm.deform_batched(rates, pals)
            spos, snrm = dm.deform_batched(rates[0], pals, shared_weights=True)
            worst = dict(pos_abs=0.0, pos_rel=0.0, nrm_abs=0.0, differ=0)
            for i in range(ni):
                for (gp, gn, r) in ((pos[i], nrm[i], rates[i]), (spos[i], snrm[i], rates[0])):
                    ep, en = orc.skin(m, pals[i], orc.morph(m, r), skin)
                    d = np.abs(gp.astype(np.float64) - ep)
                    worst["pos_abs"] = max(worst["pos_abs"], d.max())
                    worst["pos_rel"] = max(worst["pos_rel"], (d / (1 + np.abs(ep))).max())
                    worst["nrm_abs"] = max(worst["nrm_abs"], np.abs(gn.astype(np.float64) - en).max())
                    worst["differ"] += int((gp.view(np.uint32) != ep.view(np.uint32)).sum())
            p1, n1 = dm.deform(rates[3], pals[3])
            ep, en = orc.skin(m, pals[3], orc.morph(m, rates[3]), skin)
            worst["frame_pos_rel"] = (np.abs(p1.astype(np.float64) - ep) / (1 + np.abs(ep))).max()
            print(name, {k: (float(f"{v:.3g}") if isinstance(v, float) else v) for k, v in worst.items()},
                  "max |pos|", float(np.abs(ep).max()), flush=True)


def speed():
    m = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(m, (np.arange(ni) * 3) % 1801)
    fr = (np.arange(ni) * 7) % 600
    rates_i = synth.morph_weights(m.nm, fr)
    rates_s = synth.morph_weights(m.nm, 30)[0]
    d_pal, d_wi, d_ws = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates_i), DeviceBuffer.from_numpy(rates_s)
    for fast in (False, True, False, True):
        dm = DeformModel(m, fast_math=fast)
        d_a, d_b, pl = dm.alloc_outputs(api.OUT_SOA, ni, 64)
        flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
        t_crowd = bench.time_calls(dm, lambda: dm.deform_batched_raw(ni, d_ws.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA,
                                                                     flags | api.WEIGHTS_SHARED), 20)
        t_inst = bench.time_calls(dm, lambda: dm.deform_batched_raw(ni, d_wi.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags), 10)
        t_64 = bench.time_calls(dm, lambda: dm.deform_batched_raw(64, d_wi.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags), 50)
        t_1 = bench.time_calls(dm, lambda: dm.deform_batched_raw(1, d_wi.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags), 200)
        print(f"fast_math={fast!s:5}  placement {pl.get('store_GBs', 0):.0f} GB/s   crowd step {t_crowd * 1e3:7.1f} us   per-instance morphs "
              f"{t_inst * 1e3:7.1f} us   64 frames {t_64 * 1e3:6.1f} us   1 frame {t_1 * 1e3:5.2f} us", flush=True)
        d_a.free(); d_b.free() if d_b else None
        dm.close()
    m5 = synth.make_config("config5_256k")
    fr = np.arange(64)
    d_p5, d_w5 = DeviceBuffer.from_numpy(synth.make_palettes(m5, fr)), DeviceBuffer.from_numpy(synth.morph_weights(m5.nm, fr))
    for fast in (False, True, False, True):
        dm = DeformModel(m5, f16_positions=True, fast_math=fast)
        sa, sb = dm.out_sizes(api.OUT_SOA_POS16, 64)
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
        flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
        t64 = bench.time_calls(dm, lambda: dm.deform_batched_raw(64, d_w5.ptr, d_p5.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA_POS16, flags), 20)
        t1 = bench.time_calls(dm, lambda: dm.deform_batched_raw(1, d_w5.ptr, d_p5.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA_POS16, flags), 100)
        print(f"config 5 fast_math={fast!s:5}  64 frames {t64 * 1e3:6.1f} us   1 frame {t1 * 1e3:5.2f} us", flush=True)
        d_a.free(); d_b.free(); dm.close()


if __name__ == "__main__":
    errors()
    speed()
