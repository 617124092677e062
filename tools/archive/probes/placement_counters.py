#!/usr/bin/env python3
"""The slow / fast store mode of the crowd's output arrays (LAB_NOTES.md: a property of the physical backing hipMalloc hands
out) under the hardware counters: which unit is waiting in the slow mode?

    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d DIR -o p -- python3 tools/archive/probes/placement_counters.py
    python3 tools/archive/probes/placement_counters.py --analyze DIR [DIR ...]

ONE process allocates PC_PAIRS pairs of output arrays and keeps them all (every pair is fresh physical memory); on every
pair, in turn: 3 store-only replays of the crowd pattern, then 3 crowd kernels (config 3, morph pass skipped).  Nothing is
selected by a rate measured under the profiler: the analysis reads every dispatch's own duration from the kernel trace and puts
the counters of the same dispatch next to it, pair by pair -- fast and slow placements side by side from one run."""
import collections
import csv
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

REPS = 3


def analyze(dirs):
    for d in dirs:
        kt = sorted(csv.DictReader(open(os.path.join(d, "p_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
        cc = list(csv.DictReader(open(os.path.join(d, "p_counter_collection.csv"))))
        ctr = collections.defaultdict(dict)
        for r in cc:
            ctr[r["Dispatch_Id"]][r["Counter_Name"]] = ctr[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names = sorted({r["Counter_Name"] for r in cc})
        for kind in ("pattern_fill_kernel", "deform_kernel"):
            rows = [r for r in kt if kind in r["Kernel_Name"]]
            if kind == "deform_kernel":
                rows = rows[1:]                               # the first launch ran the morph pass's set-up call
            groups = [rows[i:i + REPS] for i in range(0, len(rows) - len(rows) % REPS, REPS)]
            print(f"== {d}: {kind}, per pair (mean of {REPS} launches): duration us | " + " | ".join(names))
            for g in sorted(groups, key=lambda g: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g)):
                us = np.mean([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in g])
                vals = [np.mean([ctr[r["Dispatch_Id"]].get(n, 0.0) for r in g]) for n in names]
                print(f"   {us:8.1f} | " + " | ".join(f"{v:14.0f}" for v in vals))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
        return analyze(sys.argv[2:])
    from simple_mmd_renderer_amd import _capi as api, synth
    from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer
    import ctypes as C
    m = synth.make_config("config3_crowd")
    ni = 1024
    dm = DeformModel(m)
    sa, sb = dm.out_sizes(api.OUT_SOA, ni)
    lib = api.lib()
    pals = synth.make_palettes(m, (np.arange(ni) * 3) % 1801)
    d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(synth.morph_weights(m.nm, 30)[0])
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
    pairs = [(DeviceBuffer(sa), DeviceBuffer(sb)) for _ in range(int(os.environ.get("PC_PAIRS", "10")))]
    dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, pairs[0][0].ptr, pairs[0][1].ptr, api.OUT_SOA, flags)     # the morph pass, once
    dm.sync()
    ms = C.c_float(0)
    for a, b in pairs:
        api.check(lib.mmdx_bench_store_pattern(a.ptr, b.ptr, m.nv, ni, REPS - 1, C.byref(ms)))     # 1 warm-up + REPS-1 timed = REPS launches
    for a, b in pairs:
        for _ in range(REPS):
            dm.deform_batched_raw(ni, None, d_pal.ptr, a.ptr, b.ptr, api.OUT_SOA, flags | api.MORPH_UNCHANGED)
        dm.sync()
    print("done", flush=True)


if __name__ == "__main__":
    main()
