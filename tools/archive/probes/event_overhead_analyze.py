#!/usr/bin/env python3
"""Per-phase kernel durations and launch gaps of tools/archive/probes/event_overhead_probe.py from its rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d OUT -o evt -- python3 tools/archive/probes/event_overhead_probe.py
    python tools/archive/probes/event_overhead_analyze.py OUT/evt_kernel_trace.csv"""
import csv
import sys

import numpy as np

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
d = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
defo = [(s, e) for n, s, e in d if "deform_kernel" in n][300:]          # drop the probe's 300 warm-up steps
morph = [(s, e) for n, s, e in d if "morph_apply" in n][300:]
i = 0
for rnd in range(4):
    for mode in (0, 1, 5):
        seg, mseg = defo[i + 20:i + 220], morph[i + 20:i + 220]
        i += 220
        dur = np.array([e - s for s, e in seg]) / 1e3
        mdur = np.array([e - s for s, e in mseg]) / 1e3
        g1 = np.array([seg[k][0] - mseg[k][1] for k in range(len(seg))]) / 1e3
        g2 = np.array([mseg[k + 1][0] - seg[k][1] for k in range(len(seg) - 1)]) / 1e3
        step = (seg[-1][1] - seg[0][1]) / (len(seg) - 1) / 1e3
        what = "no events" if mode == 0 else ("events on every step" if mode == 1 else "events on every 5th step")
        print(f"round {rnd}, {what:24s}: deform {dur.mean():7.2f} us  morph pass {mdur.mean():5.2f}  "
              f"gap morph->deform {g1.mean():5.2f}  gap deform->morph {g2.mean():5.2f}  step {step:7.2f}")
