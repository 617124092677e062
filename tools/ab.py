#!/usr/bin/env python3
"""Interleaved A/B timing of deform-kernel variants in ONE process (cdna guide, rule 24):
N configurations x R rounds, median and min of the HIP-event kernel time per configuration.

    python tools/ab.py "MMDX_GROUP=16" "MMDX_GROUP=8 MMDX_THREADS=256" ...

Each argument is a space-separated list of VAR=VALUE settings read by libmmdx at call time
(MMDX_GROUP, MMDX_THREADS, MMDX_LDS_TARGET, MMDX_INTERLEAVE; re-read through mmdx_debug_reload_env).
Workload: BASELINE config 3 (1024 x 50k crowd, shared morphs), override with AB_WORKLOAD=v32.
AB_DENSE=1: no events, wall clock per whole step (morph pass + deform kernel back to back) instead of the
event-bracketed deform kernel.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402

KNOBS = ("MMDX_GROUP", "MMDX_THREADS", "MMDX_LDS_TARGET", "MMDX_INTERLEAVE", "MMDX_SHARED_FUSED", "MMDX_EXPERIMENT_DIRECT", "MMDX_STORE_WT",
         "MMDX_MORPH_AUTOSKIP")


def main():
    cfgs = [dict(kv.split("=") for kv in a.split()) if a.strip() else {} for a in sys.argv[1:]] or [{}]
    rounds = int(os.environ.get("AB_ROUNDS", "7"))
    iters = int(os.environ.get("AB_ITERS", "10"))
    layout = api.OUT_VERTEX32 if os.environ.get("AB_WORKLOAD") == "v32" else api.OUT_SOA
    model = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(model, (np.arange(ni) * 3) % 1801)
    rates = synth.morph_weights(model.nm, 30)[0]
    dm = DeformModel(model, tile_order=os.environ.get("AB_TILE_ORDER") == "1", fast_math=os.environ.get("AB_FAST_MATH") == "1")   # opt-in modes
    d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
    d_a, d_b, placement = dm.alloc_outputs(layout, ni, int(os.environ.get("AB_TRIES", "24")))   # AB_TRIES=1: whatever hipMalloc hands out first
    print("output placement:", placement, flush=True)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    if "MMDX_STORE_WT" not in " ".join(sys.argv[1:]):
        flags |= placement.get("store_flags", 0)       # the probe's verdict travels with the call (unless the A/B is about it)
    dense = os.environ.get("AB_DENSE") == "1"
    dm.profile_enable(not dense)
    res = [[] for _ in cfgs]
    warm = int(os.environ.get("AB_WARM_ROUNDS", "3"))               # clock transient after idle: ~100 launches
    for r in range(rounds + warm):
        for ci, cfg in enumerate(cfgs):
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(cfg)
            api.lib().mmdx_debug_reload_env()
            def burst(n_):
                for _ in range(n_):                 # back-to-back, no host sync: sustained rate
                    dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr if d_b else None, layout,
                                          flags, 0.1 if layout == api.OUT_VERTEX32 else 1.0)
            if dense:       # no events at all: wall clock per step (kernels back to back)
                burst(5)
                dm.sync()
                t0 = time.perf_counter()
                burst(iters * 10)
                dm.sync()
                if r >= warm:
                    res[ci].append((time.perf_counter() - t0) / (iters * 10) * 1e3)
                continue
            burst(iters)
            n, skin, _ = dm.profile_collect()
            if r >= warm:
                res[ci].append(skin / n)
    for cfg, r in zip(cfgs, res):
        r = np.asarray(r) * 1e3
        name = " ".join(f"{k}={v}" for k, v in cfg.items()) or "(defaults)"
        print(f"{name:48s} median {np.median(r):7.1f} us   min {r.min():7.1f}   max {r.max():7.1f}")


if __name__ == "__main__":
    main()
