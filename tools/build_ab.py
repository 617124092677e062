#!/usr/bin/env python3
"""Interleaved A/B of BUILDS of libmmdx.so in ONE process on the SAME output arrays (written in round 3 for the cache policy of
the copy-out stores; the general tool for any build-time knob: MMDX_BUILD_DEFS=... MMDX_BUILD_OUT=build/variants/libmmdx_x.so
python -m simple_mmd_renderer_amd.build, then)

    python tools/build_ab.py name=path.so [name=path.so ...]

Every library is loaded side by side (ctypes, distinct file names => distinct HIP modules), each gets its own model handle of
BASELINE config 3; the output arrays are allocated ONCE per placement and shared, so that a placement's store mode (DESIGN.md
section 6: fast / slow, a property of the physical backing) is the same for every build.  Per placement and build, R rounds
interleaved: the whole step (morph pass + deform kernel, back to back, wall clock around sync) and the deform kernel alone
(MMDX_MORPH_UNCHANGED).  AB_WORKLOAD=v32 | c3p (per-instance morph weights) | c5x64 | c2x64.
"""
import contextlib
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402


def load(path):
    l = C.CDLL(path)
    for name, (res, args) in api.SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype, fn.argtypes = res, args
    assert l.mmdx_abi_version() == api.ABI_VERSION
    return l


@contextlib.contextmanager
def use(l):
    old = api._lib
    api._lib = l
    try:
        yield
    finally:
        api._lib = old


def main():
    # name=path.so[:VAR=VALUE[,VAR=VALUE...]]: launch-heuristic overrides (MMDX_THREADS, MMDX_GROUP, MMDX_LDS_TARGET ...) for
    # THAT library only -- every build keeps its own copy of them, re-read by its mmdx_debug_reload_env; path "shipped" = the
    # shipped build once more under other overrides
    specs = [a.split("=", 1) for a in sys.argv[1:]]
    base = api.lib()                                  # the shipped build: owns the shared buffers
    base.mmdx_debug_reload_env()
    libs = [("shipped", base)]
    lib_flags = {"shipped": 0}
    n_copies = 0
    for n, spec in specs:
        path, _, envs = spec.partition(":")
        if path == "shipped":                        # a second copy of the file = a second HIP module with its own overrides
            import shutil
            import tempfile
            n_copies += 1
            cp = os.path.join(tempfile.gettempdir(), "libmmdx_abcopy%d_%d.so" % (os.getpid(), n_copies))
            shutil.copy(os.path.join(ROOT, "simple_mmd_renderer_amd", "libmmdx.so"), cp)
            path = cp
        l = load(os.path.abspath(path))
        kv = [e.split("=", 1) for e in envs.split(",") if e]
        lib_flags[n] = sum(int(v, 0) for k, v in kv if k == "FLAGS")     # FLAGS=32: extra mmdx_deform_args.flags for this entry
        kv = [(k, v) for k, v in kv if k != "FLAGS"]
        for k, v in kv:
            os.environ[k] = v
        l.mmdx_debug_reload_env()
        for k, _ in kv:
            del os.environ[k]
        libs.append((n, l))
    rounds, iters = int(os.environ.get("AB_ROUNDS", "9")), int(os.environ.get("AB_ITERS", "60"))
    wl = os.environ.get("AB_WORKLOAD", "c3")
    f16 = False
    if wl in ("c3", "v32", "c3p"):
        model, ni = synth.make_config("config3_crowd"), int(os.environ.get("AB_NI", "1024"))
        pals = synth.make_palettes(model, (np.arange(ni) * 3) % 1801)
        rates = synth.morph_weights(model.nm, 30)[0] if wl != "c3p" else synth.morph_weights(model.nm, np.arange(ni) % 600)
        layout = api.OUT_VERTEX32 if wl == "v32" else api.OUT_SOA
        shared = wl != "c3p"
    elif wl == "c5s":                                  # the 256k-vertex f16 model as a crowd with shared morphs (two arrays: 6 + 12 B)
        model, ni = synth.make_config("config5_256k"), int(os.environ.get("AB_NI", "256"))
        pals = synth.make_palettes(model, np.arange(ni))
        rates = synth.morph_weights(model.nm, 30)[0]
        f16, layout, shared = True, api.OUT_SOA_POS16, True
    else:
        cfg, ni = ("config5_256k", 64) if wl == "c5x64" else ("config2_50k", 64)
        model = synth.make_config(cfg)
        pals = synth.make_palettes(model, np.arange(ni))
        rates = synth.morph_weights(model.nm, np.arange(ni))
        f16 = wl == "c5x64"
        layout = api.OUT_SOA_POS16 if f16 else api.OUT_SOA
        shared = False
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | (api.WEIGHTS_SHARED if shared else 0)
    scale = 0.1 if layout == api.OUT_VERTEX32 else 1.0
    dms = []
    for name, l in libs:
        with use(l):
            dms.append(DeformModel(model, f16_positions=f16))
    with use(base):
        d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
        placements = []
        tries = int(os.environ.get("AB_TRIES", "48"))
        a, b, info = dms[0].alloc_outputs(layout, ni, tries)
        placements.append(("shopped(%d tries, %.0f of %.0f GB/s)" % (info["tries"], info["store_GBs"], info["fill_GBs"]), a, b))
        for _plain in range(int(os.environ.get("AB_PLAIN_N", "1")) if os.environ.get("AB_PLAIN", "1") == "1" else 0):   # AB_PLAIN_N pairs, all kept: fresh memory each
            sa, sb = dms[0].out_sizes(layout, ni)
            pa, pb = DeviceBuffer(sa), (DeviceBuffer(sb) if sb else None)
            rate = ""
            if layout == api.OUT_SOA and model.nv % 4 == 0:
                ms = C.c_float(0)
                api.check(base.mmdx_bench_store_pattern(pa.ptr, pb.ptr, model.nv, ni, 5, C.byref(ms)))
                rate = " store pattern %.0f GB/s" % ((sa + sb) / (ms.value * 1e-3) / 1e9)
            placements.append(("plain hipMalloc" + rate, pa, pb))

    def burst(dm, l, n, a, b, extra=0):
        extra |= lib_flags[[nm for nm, ll in libs if ll is l][0]] if sum(1 for _, ll in libs if ll is l) == 1 else 0
        with use(l):
            for _ in range(n):
                dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, a.ptr, b.ptr if b else None, layout, flags | extra, scale)

    def timed(dm, l, n, a, b, extra=0):
        burst(dm, l, 3, a, b, extra)
        with use(l):
            dm.sync()
        t0 = time.perf_counter()
        burst(dm, l, n, a, b, extra)
        with use(l):
            dm.sync()
        return (time.perf_counter() - t0) / n * 1e6

    for pname, a, b in placements:
        print(f"== workload {wl}, placement: {pname}", flush=True)
        step = [[] for _ in libs]
        kern = [[] for _ in libs]
        for r in range(rounds + 2):                       # two settle rounds (clock transient after idle)
            for i, (name, l) in enumerate(libs):
                s = timed(dms[i], l, iters, a, b)
                k = timed(dms[i], l, iters, a, b, api.MORPH_UNCHANGED) if shared else float("nan")
                if r >= 2:
                    step[i].append(s); kern[i].append(k)
        for i, (name, _) in enumerate(libs):
            s, k = np.asarray(step[i]), np.asarray(kern[i])
            print(f"  {name:14s} step median {np.median(s):7.2f} us (min {s.min():7.2f} max {s.max():7.2f})"
                  + (f"   kernel-only median {np.median(k):7.2f} (min {k.min():7.2f})" if shared else ""), flush=True)
    # every build's result is the shipped build's, bit for bit
    with use(base):
        ref_a = placements[0][1].download((placements[0][1].nbytes,), np.uint8)
    for i, (name, l) in enumerate(libs[1:], 1):
        placements[0][1].memset(0)
        burst(dms[i], l, 1, placements[0][1], placements[0][2])
        with use(l):
            dms[i].sync()
        with use(base):
            got = placements[0][1].download((placements[0][1].nbytes,), np.uint8)
        print(f"  {name}: out_a identical to the shipped build's: {bool(np.array_equal(got, ref_a))}")
    for dm, (_, l) in zip(dms, libs):               # every handle goes back to the library that made it
        with use(l):
            dm.close()


if __name__ == "__main__":
    main()
