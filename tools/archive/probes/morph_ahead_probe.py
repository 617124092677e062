#!/usr/bin/env python3
"""(Needs commit c300087, reverted: the flag no longer exists -- kept as the record of how profiles/r03/morph_ahead_side_stream_negative.txt was measured.)
MMDX_MORPH_AHEAD under the kernel trace: 40 plain crowd steps, then 40 with the morph pass on the side stream -- per phase the
wall clock per step, and (from rocprofv3's kernel trace of this process) the deform kernels' durations and the gaps between them.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o a -- python3 tools/archive/probes/morph_ahead_probe.py
    python3 tools/archive/probes/morph_ahead_probe.py --analyze DIR/a_kernel_trace.csv"""
import csv
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

N = 40


def analyze(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    d = [r for r in rows if "deform_kernel" in r["Kernel_Name"]]
    mo = [r for r in rows if "morph_apply" in r["Kernel_Name"]]
    print("deform launches", len(d), "morph passes", len(mo))
    for name, lo in (("plain", len(d) - 2 * N), ("ahead", len(d) - N)):
        seg = d[lo:lo + N]
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg]
        gap = [(int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3 for i in range(len(seg) - 1)]
        span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3 / len(seg)
        print(f"{name}: deform kernel {np.median(dur):.1f} us (min {min(dur):.1f} max {max(dur):.1f}); gap between deform kernels median "
              f"{np.median(gap):.1f} us (max {max(gap):.1f}); per step {span:.1f} us")
    last = mo[-N:]
    for r in last[-3:]:
        # where the side-stream pass ran relative to the deform kernel before it
        prev = [x for x in d if int(x["Start_Timestamp"]) <= int(r["Start_Timestamp"])][-1]
        print("  morph pass start - previous deform start: %.1f us; its duration %.1f us" % (
            (int(r["Start_Timestamp"]) - int(prev["Start_Timestamp"])) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
        return analyze(sys.argv[2])
    from simple_mmd_renderer_amd import _capi as api, synth
    from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer
    m = synth.make_config("config3_crowd")
    ni = 1024
    dm = DeformModel(m)
    d_pal = DeviceBuffer.from_numpy(synth.make_palettes(m, (np.arange(ni) * 3) % 1801))
    d_w = DeviceBuffer.from_numpy(synth.morph_weights(m.nm, 30)[0])
    d_a, d_b, pl = dm.alloc_outputs(api.OUT_SOA, ni, 64)
    print("placement", pl, flush=True)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED

    def run(extra, n):
        dm.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags | extra)
        dm.sync()
        return (time.perf_counter() - t0) / n * 1e6
    for _ in range(12):
        run(0, 30)                                  # settle
    print(f"plain {run(0, N):.1f} us/step", flush=True)
    print(f"ahead {run(api.MORPH_AHEAD, N):.1f} us/step", flush=True)


if __name__ == "__main__":
    main()
