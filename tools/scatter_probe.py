#!/usr/bin/env python3
"""Upper bound of what the LDS image scatter costs the crowd kernel: the config-3 model as is (vertices in file
order, so the class-sorted lanes scatter into the image through a random permutation: bank conflicts) against
the same model with its vertices pre-sorted the way the plan sorts them inside each 512-vertex tile (identity
permutation: every lane writes its own slot, no conflicts).  Interleaved rounds on the same output arrays."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402


def presorted(m, tile=512):
    st = np.asarray(m.skin_type)
    cls = np.where(st == 0, 0, np.where(st == 2, 2, 1))
    cnt = np.bincount(np.asarray(m.morph_index), minlength=m.nv)
    order = []
    for t0 in range(0, m.nv, tile):
        vs = np.arange(t0, min(t0 + tile, m.nv))
        order.append(vs[np.lexsort((vs, -cnt[vs], cls[vs]))])
    order = np.concatenate(order)
    new_of = np.empty(m.nv, np.int64)
    new_of[order] = np.arange(m.nv)
    s = m.copy()
    for f in ("positions", "normals", "uvs", "skin_type", "bone_ids", "bone_weights"):
        setattr(s, f, np.ascontiguousarray(getattr(m, f)[order]))
    if m.sdef is not None:
        s.sdef = np.ascontiguousarray(m.sdef[order])
    s.morph_index = new_of[np.asarray(m.morph_index)].astype(np.uint32)
    return s


def main():
    rounds, iters, warm = 7, 10, 3
    model = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(model, (np.arange(ni) * 3) % 1801)
    rates = synth.morph_weights(model.nm, 30)[0]
    dms = [("file order (random scatter)", DeformModel(model)), ("pre-sorted (identity scatter)", DeformModel(presorted(model)))]
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
