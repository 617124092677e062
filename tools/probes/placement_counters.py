#!/usr/bin/env python3
"""The slow / fast store mode of the crowd's output arrays (a property of the physical backing hipMalloc
hands out) under the hardware counters: which unit is waiting in the slow mode?

    python tools/probes/placement_counters.py fast|slow       (under rocprofv3 --pmc ...: tools/probes/r03_run3.sh)

Allocates up to PC_TRIES candidate pairs of output arrays WITHOUT freeing any (every try draws fresh physical memory), times
the store-only replay of the crowd pattern on each, keeps the fastest (`fast`) or the slowest (`slow`) pair, frees the rest,
then runs 12 pattern replays, 12 linear fills and 24 crowd kernels (config 3, morph pass skipped) on the chosen pair.
Prints the pair's pattern rate so the counters can be read against it."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
    m = synth.make_config("config3_crowd")
    ni = 1024
    dm = DeformModel(m)
    sa, sb = dm.out_sizes(api.OUT_SOA, ni)
    lib = api.lib()

    def rate(a, b, iters=5):
        ms = C.c_float(0)
        api.check(lib.mmdx_bench_store_pattern(a.ptr, b.ptr, m.nv, ni, iters, C.byref(ms)))
        return (sa + sb) / (ms.value * 1e-3) / 1e9
    cands = []
    for _ in range(int(os.environ.get("PC_TRIES", "24"))):
        a, b = DeviceBuffer(sa), DeviceBuffer(sb)
        cands.append((rate(a, b), a, b))
        if mode == "fast" and cands[-1][0] > 6400:
            break
        if mode == "slow" and cands[-1][0] < 5300:
            break
    cands.sort(key=lambda c: c[0])
    chosen = cands[-1] if mode == "fast" else cands[0]
    print("candidate pattern rates GB/s:", " ".join(f"{c[0]:.0f}" for c in cands), flush=True)
    for c in cands:
        if c is not chosen:
            c[1].free(); c[2].free()
    _, d_a, d_b = chosen
    for _ in range(30):                                   # the driver wipes the freed candidates in the background: wait it out
        rate(d_a, d_b, 5)
    print(f"mode {mode}: chosen pair stores the crowd pattern at {rate(d_a, d_b, 10):.0f} GB/s", flush=True)
    ms = C.c_float(0)
    api.check(lib.mmdx_bench_fill(d_a.ptr, sa, 12, C.byref(ms)))
    print(f"linear fill of array a: {sa / (ms.value * 1e-3) / 1e9:.0f} GB/s", flush=True)
    pals = synth.make_palettes(m, (np.arange(ni) * 3) % 1801)
    d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(synth.morph_weights(m.nm, 30)[0])
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
    dm.sync()
    dm.timer_start()
    for _ in range(24):
        dm.deform_batched_raw(ni, None, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags | api.MORPH_UNCHANGED)
    print(f"crowd kernel alone: {dm.timer_stop() / 24 * 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    main()
