#!/usr/bin/env python3
"""ONE diagnostic run for the round-2 record "rocprofv3 --kernel-trace segfaults while bench.py records a HIP graph"
(gpurun_out/profile_r02.log; the crashing run's stderr was overwritten by later runs, so no frame of it exists):

    rocprofv3 --kernel-trace --output-format csv -d DIR -o g -- python3 -X faulthandler tools/archive/probes/graph_under_profiler.py

Prints a marker before every library call of a minimal graph record / replay (one model, one frame).  If the process dies,
the last marker and faulthandler's Python stack say which mmdx_* call was active; if it survives, the kernel trace shows the
replayed kernels and the graph leg can go back into profiled runs.  Run once; never loop it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402


def mark(s):
    print("MARK", s, flush=True)
    sys.stderr.write("MARK " + s + "\n"); sys.stderr.flush()


m = synth.make_model(5000, 60, 8, 300, seed=31)
flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
mark("create model")
dm = DeformModel(m)
d_pal, d_w = DeviceBuffer.from_numpy(synth.make_palettes(m, [3])), DeviceBuffer.from_numpy(synth.morph_weights(m.nm, [3])[0])
d_a, d_b = DeviceBuffer(m.nv * 12), DeviceBuffer(m.nv * 12)
mark("eager call")
dm.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
dm.sync()
mark("mmdx_graph_begin (hipStreamBeginCapture)")
dm.graph_begin()
mark("recorded mmdx_deform_batched")
dm.deform_batched_raw(1, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
mark("mmdx_graph_end (hipStreamEndCapture + hipGraphInstantiate)")
g = dm.graph_end()
for k in range(3):
    mark(f"mmdx_graph_launch {k}")
    g.launch()
mark("sync")
dm.sync()
mark("destroy")
g.close()
dm.close()
mark("SURVIVED: graph record + 3 replays under the profiler")
