#!/usr/bin/env python3
"""Where one host-in / host-out frame (mmdx_deform_vertex32 as a patched viewer would call it) spends its time:
the whole call, the same work with device-resident buffers + an explicit wait, and the copies alone.  Boxes of the
pool differ by 3x on the whole call (45 us vs 145 us); this shows which piece carries the difference."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer, PinnedArray  # noqa: E402


def wall(fn, n=300):
    for _ in range(20):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


m = synth.make_config("config1_20k")
dm = DeformModel(m)
lib = api.lib()
f32p = C.POINTER(C.c_float)
rates = synth.morph_weights(m.nm, np.arange(1) * 9)[0].copy()
pal = synth.make_palettes(m, np.arange(1))[0].copy()
p_pal, p_rt, p_out = PinnedArray(pal.shape, np.float32), PinnedArray(rates.shape, np.float32), PinnedArray((m.nv, 8), np.float32)
p_pal.array[:] = pal
p_rt.array[:] = rates
out_pg = np.empty((m.nv, 8), np.float32)
d_pal, d_rt, d_out = DeviceBuffer.from_numpy(pal), DeviceBuffer.from_numpy(rates), DeviceBuffer(m.nv * 32)
dev_flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE


def host_call(pl, rt, out):
    api.check(lib.mmdx_deform_vertex32(dm.h, rt.ctypes.data_as(f32p), pl.ctypes.data_as(f32p), C.c_float(0.1), out.ctypes.data))


def device_call():
    dm.deform_batched_raw(1, d_rt.ptr, d_pal.ptr, d_out.ptr, None, api.OUT_VERTEX32, dev_flags, 0.1)


def device_call_wait():
    device_call()
    dm.sync()


print("MMDX_SPIN_WAIT_US =", os.environ.get("MMDX_SPIN_WAIT_US", "(default)"))
print("host in / host out, pageable   %7.1f us" % wall(lambda: host_call(pal, rates, out_pg)))
print("host in / host out, pinned     %7.1f us" % wall(lambda: host_call(p_pal.array, p_rt.array, p_out.array)))
print("device buffers, launch only    %7.1f us" % wall(device_call))
dm.sync()
print("device buffers, launch + wait  %7.1f us" % wall(device_call_wait))
print("hipMemcpy D2H 640 KB, pinned   %7.1f us" % wall(lambda: api.check(lib.mmdx_memcpy_d2h(C.c_void_p(p_out.ptr), C.c_void_p(d_out.ptr), C.c_size_t(m.nv * 32)))))
print("hipMemcpy H2D 9.6 KB, pinned   %7.1f us" % wall(lambda: api.check(lib.mmdx_memcpy_h2d(C.c_void_p(d_pal.ptr), C.c_void_p(p_pal.ptr), C.c_size_t(pal.nbytes)))))
print("mmdx_sync on an idle stream    %7.1f us" % wall(dm.sync))
