#!/usr/bin/env python3
"""Ablation of the fused-morph batched path (config 5, 64 frames per launch) in an MMDX_BUILD_ABLATE library:
MMDX_LIB=<ablate .so> python tools/ablate_fused.py 0 128 256 1 4 16 ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402

which = os.environ.get("CONFIG", "config5_256k")
m = synth.make_config(which)
f16 = which.startswith("config5")
layout = api.OUT_SOA_POS16 if f16 else api.OUT_SOA
dm = DeformModel(m, f16_positions=f16)
nfr = 64
frames = np.arange(nfr)
d_pal = DeviceBuffer.from_numpy(synth.make_palettes(m, frames))
d_w = DeviceBuffer.from_numpy(synth.morph_weights(m.nm, frames))
d_a, d_b, pl = dm.alloc_outputs(layout, nfr, 16)
print("placement", pl)
flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
for a in sys.argv[1:] or ["0"]:
    os.environ["MMDX_ABLATE"] = a
    ms = bench.time_calls(dm, lambda: dm.deform_batched_raw(nfr, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, layout, flags), 20)
    print(f"ablate={a:>4s}: {ms * 1e3:8.1f} us", flush=True)
