#!/usr/bin/env python3
"""Timing of the secondary BASELINE configs (2: single 50k model; 5: 256k verts fp16) through
bench.py's `extras`, without the config-3 crowd.  For A/B of builds: MMDX_LIB=<other .so>."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402

model = synth.make_config("config2_50k")
dm = DeformModel(model)
out = bench.extras(api, synth, DeformModel, DeviceBuffer, dm, model)
for k, v in out.items():
    if "ms_per_call" in v and "algorithmic_GBs" in v:
        print(f"{k:36s} {v['ms_per_call'] * 1e3:9.1f} us  {v['vertices_per_s'] / 1e9:8.2f} Gverts/s  "
              f"{v['algorithmic_GBs']:8.0f} GB/s")
    else:
        print(f"{k:36s} " + "  ".join(f"{a}={b:.4g}" if isinstance(b, float) else f"{a}={b}" for a, b in v.items()))
