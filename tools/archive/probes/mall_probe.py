import ctypes as C, sys
sys.path.insert(0,'.')
from simple_mmd_renderer_amd import _capi as api
from simple_mmd_renderer_amd.engine import DeviceBuffer, device_select
device_select(0)
big = DeviceBuffer(1<<30); big2 = DeviceBuffer(1<<30)
ms = C.c_float(0)
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = mb << 20
    api.check(api.lib().mmdx_bench_fill(big.ptr, n, 50, C.byref(ms)))
    f = n / (ms.value*1e-3) / 1e9
    api.check(api.lib().mmdx_bench_copy(big2.ptr, big.ptr, n, 50, C.byref(ms)))
    c = 2*n / (ms.value*1e-3) / 1e9
    print(f"{mb:5d} MB  fill {f:8.0f} GB/s   copy (read+write) {c:8.0f} GB/s", flush=True)
