#!/usr/bin/env python3
"""A/B: crowd with ONE shared facial state -- morph gather inside the deform kernel (MMDX_SHARED_FUSED=1) vs the separate
morph pass (=0) -- over crowd sizes.  ms per step, back-to-back steps between two HIP events."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402

m = synth.make_config("config3_crowd")
dm = DeformModel(m)
rates = synth.morph_weights(m.nm, 30)[0]
d_w = DeviceBuffer.from_numpy(rates)
flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
for ni in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
    d_pal = DeviceBuffer.from_numpy(synth.make_palettes(m, (np.arange(ni) * 3) % 1801))
    sa, sb = dm.out_sizes(api.OUT_SOA, ni)
    d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
    res = {}
    for rep in range(2):
        for mode in ("0", "1"):
            os.environ["MMDX_SHARED_FUSED"] = mode
            api.lib().mmdx_debug_reload_env()
            ms = bench.time_calls(dm, lambda: dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags),
                                  max(10, min(200, 4096 // ni)))
            res.setdefault(mode, []).append(ms)
    print(f"ni={ni:5d}  separate pass {min(res['0']) * 1e3:8.1f} us   gather in kernel {min(res['1']) * 1e3:8.1f} us", flush=True)
    for b in (d_pal, d_a, d_b):
        b.free()
