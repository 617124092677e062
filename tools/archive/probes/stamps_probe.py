#!/usr/bin/env python3
"""Diagnostic (needs a build with in-kernel s_memtime stamps, see profiles/r02/stamps_*.txt): prints the per-wave cycle
split of the deform kernel for the shared-morph crowd and the per-instance-morph crowd."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MMDX_DEBUG_STAMPS"] = "1"
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402

m = synth.make_config("config3_crowd")
ni = 1024
dm = DeformModel(m)
fr = (np.arange(ni) * 7) % 600
d_pal = DeviceBuffer.from_numpy(synth.make_palettes(m, fr))
d_wi = DeviceBuffer.from_numpy(synth.morph_weights(m.nm, fr))
d_ws = DeviceBuffer.from_numpy(synth.morph_weights(m.nm, 30)[0])
sa, sb = dm.out_sizes(api.OUT_SOA, ni)
d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
for rep in range(4):
    sys.stderr.write("shared-morph crowd:   ")
    dm.deform_batched_raw(ni, d_ws.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags | api.WEIGHTS_SHARED)
for rep in range(4):
    sys.stderr.write("per-instance morphs:  ")
    dm.deform_batched_raw(ni, d_wi.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
