#!/usr/bin/env python3
"""How long may the GPU idle before the crowd step is "cold" again?  bench.py's cold_ms_per_step (W warm-up + K steps right after
set-up) is ~20 % above the settled figure although the placement probe keeps the GPU busy right before it -- so what the first
launches pay is not simply low clocks after idleness.  Settle (400 steps), then for every gap: sleep, time 10 back-to-back steps
(first batch), 10 more, 10 more; three rounds.   python tools/archive/probes/idle_gap_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402

m = synth.make_config("config3_crowd")
ni = 1024
dm = DeformModel(m)
d_pal = DeviceBuffer.from_numpy(synth.make_palettes(m, (np.arange(ni) * 3) % 1801))
d_w = DeviceBuffer.from_numpy(synth.morph_weights(m.nm, 30)[0])
d_a, d_b, pl = dm.alloc_outputs(api.OUT_SOA, ni, 64)
print("placement", pl, flush=True)
flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED


def batch(n):
    dm.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
    dm.sync()
    return (time.perf_counter() - t0) / n * 1e6


first = [batch(10) for _ in range(8)]
print("right after set-up, batches of 10 steps (us/step):", " ".join(f"{x:.1f}" for x in first), flush=True)
for _ in range(40):
    batch(10)
print(f"settled: {batch(20):.1f} us/step", flush=True)
for gap_ms in (0, 0.5, 2, 5, 10, 20, 50, 200, 1000):
    rows = []
    for rnd in range(3):
        for _ in range(20):
            batch(10)                               # settled again
        time.sleep(gap_ms * 1e-3)
        rows.append([batch(10) for _ in range(4)])
    r = np.median(np.asarray(rows), axis=0)
    print(f"idle {gap_ms:7.1f} ms -> next four batches of 10 steps: " + " ".join(f"{x:6.1f}" for x in r) + " us/step", flush=True)
