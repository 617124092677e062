#!/usr/bin/env python3
"""Upper bound of what the LDS image scatter costs the crowd kernel: the config-3 model as is (vertices in file
order, so the class-sorted lanes scatter into the image through a random permutation: bank conflicts) against
the same model with its vertices pre-sorted the way the plan sorts them inside each 512-vertex tile (identity
permutation: every lane writes its own slot, no conflicts).  Interleaved rounds on the same output arrays."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402


def main():
    rounds, iters, warm = 7, 10, 3
    model = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(model, (np.arange(ni) * 3) % 1801)
    rates = synth.morph_weights(model.nm, 30)[0]
    dms = [("file order (random scatter)", DeformModel(model)), ("pre-sorted (identity scatter)", DeformModel(synth.presort_by_class(model)))]
    d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
    d_a, d_b, placement = dms[0][1].alloc_outputs(api.OUT_SOA, ni, 24)
    print("output placement:", placement, flush=True)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    res = [[] for _ in dms]
    for _, dm in dms:
        dm.profile_enable(True)
    for r in range(rounds + warm):
        for ci, (_, dm) in enumerate(dms):
            for _ in range(iters):
                dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags, 1.0)
            n, skin, _ = dm.profile_collect()
            if r >= warm:
                res[ci].append(skin / n)
    for (name, _), r in zip(dms, res):
        r = np.asarray(r) * 1e3
        print(f"{name:32s} median {np.median(r):7.1f} us   min {r.min():7.1f}   max {r.max():7.1f}")


if __name__ == "__main__":
    main()
