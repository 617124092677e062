#!/usr/bin/env python3
"""What the per-kernel HIP events cost the crowd step (config 3): no events, events on every step (four records per
step), events on every 5th step -- interleaved in one process on the same output arrays."""
import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from simple_mmd_renderer_amd import _capi as api, synth
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer
model = synth.make_config("config3_crowd"); ni = 1024
pals = synth.make_palettes(model, (np.arange(ni) * 3) % 1801); rates = synth.morph_weights(model.nm, 30)[0]
dm = DeformModel(model)
d_a, d_b, placement = dm.alloc_outputs(api.OUT_SOA, ni, 24); print(placement)
d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
def step(): dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags, 1.0)
for _ in range(300): step()
dm.sync()
for rnd in range(4):
    for prof in (0, 1, 5):
        dm.profile_enable(bool(prof), every=max(prof, 1))
        for _ in range(20): step()
        dm.sync()
        if prof: dm.profile_collect()
        t0 = time.perf_counter()
        for _ in range(200): step()
        dm.sync()
        dt = (time.perf_counter() - t0) / 200 * 1e3
        extra = ""
        if prof:
            n, skin, morph = dm.profile_collect(); extra = " skin %.4f morph %.4f" % (skin / n, morph / n)
        print("events on every %d-th step (0 = none):  ms/step %.4f%s" % (prof, dt, extra))
