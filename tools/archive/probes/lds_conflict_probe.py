#!/usr/bin/env python3
"""Upper bound of what ANY conflict-free LDS palette layout could buy the crowd kernel (task 6).

The staged palette is gathered with three ds_read_b128 per bone at a 48-byte entry stride: the 16 lanes of one LDS pass hit
distinct bank quads unless two of their bones are congruent mod 16 (12 banks * lb mod 64 walks the 16 quads once per 16
entries) -- i.e. the stride is already the best a 3 x float4 entry can have, and the conflicts come from tiles that use MORE
than 16 bones (the benchmark model: ~22 per tile).  So instead of a new layout this probe changes the MODEL: the same config-3
crowd with the bone window narrowed until no tile has more than 16 bones -- every gather is then conflict-free by construction.
If the kernel does not speed up with ZERO gather conflicts, no re-layout of the palette can pay.

    python tools/archive/probes/lds_conflict_probe.py [window ...]        default: 16 12      (16 = BASELINE config 3)
    LDS_ONLY=12 ... under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE for the counters of one window
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402


def main():
    windows = [int(a) for a in sys.argv[1:]] or [16, 12]
    c = synth.CONFIGS["config3_crowd"]
    ni = 1024
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    models, dms, bufs = [], [], []
    for w in windows:
        m = synth.make_model(c["nv"], c["nb"], c["nm"], c["k"], c["seed"], window=w)
        dm = DeformModel(m)
        models.append(m); dms.append(dm)
        bufs.append((DeviceBuffer.from_numpy(synth.make_palettes(m, (np.arange(ni) * 3) % 1801)),
                     DeviceBuffer.from_numpy(synth.morph_weights(m.nm, 30)[0])))
        print(f"window {w}: max bones per tile {dm.info.max_tile_bones}", flush=True)
    d_a, d_b, placement = dms[0].alloc_outputs(api.OUT_SOA, ni, int(os.environ.get("AB_TRIES", "48")))
    print("placement:", placement, flush=True)

    def timed(i, n, extra=0):
        dm, (d_pal, d_w) = dms[i], bufs[i]
        for k in range(3 + n):
            if k == 3:
                dm.sync(); t0 = time.perf_counter()
            dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags | extra)
        dm.sync()
        return (time.perf_counter() - t0) / n * 1e6
    res = [[] for _ in windows]
    for r in range(int(os.environ.get("AB_ROUNDS", "9")) + 2):
        for i in range(len(windows)):
            s, k = timed(i, 60), timed(i, 60, api.MORPH_UNCHANGED)
            if r >= 2:
                res[i].append((s, k))
    for w, dm, r in zip(windows, dms, res):
        r = np.asarray(r)
        print(f"window {w:3d} (<= {dm.info.max_tile_bones} bones per tile): step median {np.median(r[:, 0]):7.2f} us   kernel-only median "
              f"{np.median(r[:, 1]):7.2f} us (min {r[:, 1].min():7.2f})", flush=True)


if __name__ == "__main__":
    main()
