#!/usr/bin/env python3
"""Per trial: re-allocate the crowd's two output arrays, then time (a) the store-only replay of the
output pattern and (b) the real deform kernel (sustained launches, HIP events) on THOSE arrays.
Shows whether the kernel's run-to-run spread is the placement of the output arrays."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402

model = synth.make_config("config3_crowd")
ni = 1024
pals = synth.make_palettes(model, (np.arange(ni) * 3) % 1801)
rates = synth.morph_weights(model.nm, 30)[0]
dm = DeformModel(model)
d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
sa, sb = dm.out_sizes(api.OUT_SOA, ni)
flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
dm.profile_enable(True)
ms = C.c_float()
parked = []
for trial in range(int(os.environ.get("TRIALS", "12"))):
    if os.environ.get("SHOP"):
        d_a, d_b, info = dm.alloc_outputs(api.OUT_SOA, ni, int(os.environ["SHOP"]))
        print("   placement:", info, flush=True)
    else:
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
    api.check(api.lib().mmdx_bench_store_pattern(d_a.ptr, d_b.ptr, model.nv, ni, 10, C.byref(ms)))
    pat = ni * model.nv * 24 / (ms.value * 1e-3) / 1e9
    ks = []
    per = int(os.environ.get("PER_ROUND", "20"))
    for r in range(int(os.environ.get("ROUNDS", "4"))):
        for _ in range(per):
            dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags, 1.0)
        n, skin, _ = dm.profile_collect()
        if r or os.environ.get("KEEP_FIRST"):
            ks.append(skin / n * 1e3)
    print(f"trial {trial:2d}: pattern {pat:6.0f} GB/s   kernel " + " ".join(f"{k:6.1f}" for k in ks) + " us", flush=True)
    if os.environ.get("PARK") and pat < 6000:
        parked += [d_a, d_b]          # keep the slow placement allocated so the next try lands elsewhere
    else:
        d_a.free(); d_b.free()
