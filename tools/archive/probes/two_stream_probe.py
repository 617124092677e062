#!/usr/bin/env python3
"""Does splitting the 1 024-instance crowd over TWO handles (two HIP streams on the same device) hide the kernel boundaries?
Back-to-back deform kernels of one stream cost 202-206 us each against 193-197 us in isolation (the next kernel's set-up reads
wait behind the previous kernel's draining stores, and the stream serialises tail and head).  Two half-crowds on two streams may
overlap one kernel's tail with the other's head.   python tools/archive/probes/two_stream_probe.py"""
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
pals = synth.make_palettes(m, (np.arange(ni) * 3) % 1801)
d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(synth.morph_weights(m.nm, 30)[0])
dms = [DeformModel(m) for _ in range(4)]
d_a, d_b, pl = dms[0].alloc_outputs(api.OUT_SOA, ni, 64)
print("placement", pl, flush=True)
flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
row, prow = m.nv * 12, m.nb * 64


def step(parts, extra=0):
    per = ni // parts
    for k in range(parts):
        dms[k].deform_batched_raw(per, d_w.ptr, d_pal.ptr + k * per * prow, d_a.ptr + k * per * row, d_b.ptr + k * per * row,
                                  api.OUT_SOA, flags | extra)


def timed(parts, n, extra=0):
    for k in range(parts):
        dms[k].sync()
    t0 = time.perf_counter()
    for _ in range(n):
        step(parts, extra)
    for k in range(parts):
        dms[k].sync()
    return (time.perf_counter() - t0) / n * 1e6


for _ in range(10):
    timed(1, 30)
res = {}
for rnd in range(7):
    for parts in (1, 2, 4):
        res.setdefault((parts, 0), []).append(timed(parts, 40))
        res.setdefault((parts, 1), []).append(timed(parts, 40, api.MORPH_UNCHANGED))
for (parts, un), v in sorted(res.items()):
    print(f"{parts} handle(s) / stream(s), {'deform kernels only' if un else 'whole steps        '}: median {np.median(v):7.1f} us per 1 024-instance step "
          f"(min {min(v):.1f})", flush=True)
