#!/usr/bin/env python3
"""Is the bimodal store-pattern rate tied to the allocation or to the moment?  One pair of output arrays,
allocated once; measure repeatedly with different idle gaps in between."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api  # noqa: E402
from simple_mmd_renderer_amd.engine import DeviceBuffer  # noqa: E402

nv, ni = 50000, 1024
nbytes = nv * ni * 12
lib = api.lib()
ms = C.c_float()
a, b = DeviceBuffer(nbytes + (1 << 20)), DeviceBuffer(nbytes + (1 << 20))


def rate():
    api.check(lib.mmdx_bench_store_pattern(a.ptr, b.ptr, nv, ni, 10, C.byref(ms)))
    return 2 * nbytes / (ms.value * 1e-3) / 1e9


for gap in (0.0, 0.0, 0.001, 0.01, 0.1, 0.5, 0.0, 0.1, 0.0, 1.0, 0.0):
    time.sleep(gap)
    print(f"same allocation, after {gap * 1e3:6.0f} ms idle: " + " ".join(f"{rate():6.0f}" for _ in range(6)), flush=True)
# now free / allocate something unrelated in between (the output arrays stay)
for k in range(6):
    d = DeviceBuffer((64 << 20) * (k + 1))
    d.free()
    print(f"same allocation, after an unrelated malloc/free of {64 * (k + 1)} MiB: " + " ".join(f"{rate():6.0f}" for _ in range(4)), flush=True)
