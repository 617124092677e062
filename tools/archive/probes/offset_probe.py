#!/usr/bin/env python3
"""One allocation holding both crowd output arrays: how does the store-pattern rate depend on the byte
offset between array a and array b?  (tools/archive/probes/alloc_probe.py showed the rate is bimodal across separate
allocations, i.e. it depends on the physical placement of the two arrays relative to each other.)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api  # noqa: E402
from simple_mmd_renderer_amd.engine import DeviceBuffer  # noqa: E402

nv, ni = 50000, 1024
nbytes = nv * ni * 12
lib = api.lib()
ms = C.c_float()
whole = DeviceBuffer(2 * nbytes + (160 << 20))
base = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
deltas = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 3 << 19,
          2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20, 64 << 20, 96 << 20, 128 << 20]
for rep in range(int(os.environ.get("REPS", "2"))):
    for d in deltas:
        api.check(lib.mmdx_bench_store_pattern(whole.ptr, whole.ptr + base + d, nv, ni, 10, C.byref(ms)))
        print(f"rep {rep} b-a = {base / 2**20:.0f} MiB + {d:>10d} B: {2 * nbytes / (ms.value * 1e-3) / 1e9:6.0f} GB/s", flush=True)
