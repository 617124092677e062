#!/usr/bin/env python3
"""Interleaved A/B of the model creation modes (default / MMDX_CREATE_TILE_ORDER / MMDX_CREATE_FAST_MATH / both) on the SAME output
arrays in ONE process: config-3 crowd step (shared morphs) and the per-instance-morph crowd, median of R rounds.
AB_TRIES=1: plainly allocated outputs (whatever placement hipMalloc hands out)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402


def main():
    rounds, iters = int(os.environ.get("AB_ROUNDS", "7")), int(os.environ.get("AB_ITERS", "60"))
    m = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(m, (np.arange(ni) * 3) % 1801)
    d_pal = DeviceBuffer.from_numpy(pals)
    d_ws = DeviceBuffer.from_numpy(synth.morph_weights(m.nm, 30)[0])
    d_wi = DeviceBuffer.from_numpy(synth.morph_weights(m.nm, (np.arange(ni) * 7) % 600))
    modes = [("default", {}), ("tile order", dict(tile_order=True)), ("fast math", dict(fast_math=True)),
             ("tile order + fast math", dict(tile_order=True, fast_math=True))]
    dms = [DeformModel(m, **kw) for _, kw in modes]
    d_a, d_b, pl = dms[0].alloc_outputs(api.OUT_SOA, ni, int(os.environ.get("AB_TRIES", "64")))
    print("output placement:", pl, flush=True)
    base = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | pl.get("store_flags", 0)
    res = {(n, w): [] for n, _ in modes for w in ("crowd", "per-instance")}
    for r in range(rounds + 2):
        for (name, _), dm in zip(modes, dms):
            for what, w, fl, it in (("crowd", d_ws, base | api.WEIGHTS_SHARED, iters), ("per-instance", d_wi, base, iters // 3)):
                for _ in range(5):
                    dm.deform_batched_raw(ni, w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, fl)
                dm.sync()
                t0 = time.perf_counter()
                for _ in range(it):
                    dm.deform_batched_raw(ni, w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, fl)
                dm.sync()
                if r >= 2:
                    res[(name, what)].append((time.perf_counter() - t0) / it * 1e6)
    for (name, what), v in res.items():
        print(f"{name:26s} {what:14s} median {np.median(v):7.1f} us   min {min(v):7.1f}   max {max(v):7.1f}")


if __name__ == "__main__":
    main()
