#!/usr/bin/env python3
"""Distribution of the placement probe's per-try rates (MMDX_PLACEMENT_LOG=1 prints every try)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MMDX_PLACEMENT_LOG", "1")
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel  # noqa: E402

dm = DeformModel(synth.make_config("config3_crowd"))
for rep in range(int(os.environ.get("REPS", "3"))):
    a, b, info = dm.alloc_outputs(api.OUT_SOA, 1024, int(os.environ.get("TRIES", "40")))
    print("rep", rep, info, flush=True)
    a.free(); b.free()
