#!/usr/bin/env python3
"""Does the crowd store pattern's rate depend on WHERE the two output arrays were allocated?
Repeats, in one process: allocate the two 614 MB arrays (optionally after some decoy allocations that
shift the addresses), time the store-only replay of the deform kernel's pattern and a linear fill of the
same bytes, free.  Prints the device addresses next to the rates."""
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
MODE = os.environ.get("MODE", "separate")


class View:
    def __init__(self, ptr):
        self.ptr = ptr

    def free(self):
        pass


for trial in range(int(os.environ.get("TRIALS", "10"))):
    decoys = [DeviceBuffer((trial * 37 % 11 + 1) * (1 << 20) + 4096 * trial) for _ in range(trial % 4)]
    if MODE == "single":                      # one allocation, array b 2 MiB-aligned behind array a
        whole = DeviceBuffer(2 * nbytes + (8 << 20))
        off = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        a, b = View(whole.ptr), View(whole.ptr + off)
        decoys.append(whole)
    elif MODE == "gib":                       # sizes rounded up to 1 GiB
        a, b = DeviceBuffer(1 << 30), DeviceBuffer(1 << 30)
    else:
        a, b = DeviceBuffer(nbytes + (1 << 20)), DeviceBuffer(nbytes + (1 << 20))
    rates = []
    for rep in range(3):
        api.check(lib.mmdx_bench_store_pattern(a.ptr, b.ptr, nv, ni, 10, C.byref(ms)))
        rates.append(2 * nbytes / (ms.value * 1e-3) / 1e9)
    api.check(lib.mmdx_bench_fill(a.ptr, nbytes, 10, C.byref(ms)))
    fill = nbytes / (ms.value * 1e-3) / 1e9
    print(f"trial {trial}: a={a.ptr:#x} b={b.ptr:#x} (b-a = {(b.ptr - a.ptr) / 2**20:9.2f} MiB)  pattern "
          + " ".join(f"{r:6.0f}" for r in rates) + f" GB/s   fill {fill:6.0f} GB/s", flush=True)
    for d in decoys + [a, b]:
        d.free()
