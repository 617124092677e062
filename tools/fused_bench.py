#!/usr/bin/env python3
"""The per-instance-morph ("fused gather") workloads alone: config 2 x 64 frames, config 5 x 64 frames (f16
positions), config 3' (1024-instance crowd, every instance its own morph weights).  Prints ms per call and the
algorithmic-bytes rate; small enough to run under rocprofv3 (--kernel-trace / --pmc passes).

    python tools/fused_bench.py [c2] [c5] [c3p] [c2x1] [c5x1] [--iters N]   (c2x1 / c5x1: one frame per launch)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402

HBM = 8000.0


def run(name, model, ni, frames, layout, iters, f16=False, shared=False):
    dm = DeformModel(model, f16_positions=f16, tile_order=os.environ.get("FB_TILE_ORDER") == "1",
                     fast_math=os.environ.get("FB_FAST_MATH") == "1")                     # opt-in modes
    pals = synth.make_palettes(model, frames)
    rates = synth.morph_weights(model.nm, frames)
    d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
    sa, sb = dm.out_sizes(layout, ni)
    d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
    flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | (api.WEIGHTS_SHARED if shared else 0)
    call = lambda: dm.deform_batched_raw(ni, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, layout, flags)   # noqa: E731
    # FB_SWEEP="MMDX_XCD_CHUNK=0,1,2,4": one launch-shape override swept in this process (interleaved, FB_ROUNDS rounds, medians)
    if os.environ.get("FB_SWEEP"):
        knob, vals = os.environ["FB_SWEEP"].split("=")
        res = {v: [] for v in vals.split(",")}
        for r in range(int(os.environ.get("FB_ROUNDS", "5")) + 1):
            for v in res:
                os.environ[knob] = v
                api.lib().mmdx_debug_reload_env()
                t = bench.time_calls(dm, call, iters)
                if r:
                    res[v].append(t * 1e3)
        os.environ.pop(knob, None)
        api.lib().mmdx_debug_reload_env()
        print(f"{name:8s} sweep {knob}: " + "  ".join(f"{v}: {np.median(t):.1f} us" for v, t in res.items()), flush=True)
    ms = bench.time_calls(dm, call, iters)
    i = dm.info
    if f16:
        static = model.nv * (6 + 12 + 1) + i.n_bdef1 * 2 + i.n_bdef2 * 8 + i.n_bdef4 * 24
        table, outb = i.n_entries * 10, 18
    else:
        static = model.nv * 25 + i.n_bdef1 * 2 + i.n_bdef2 * 8 + i.n_bdef4 * 24
        table, outb = i.n_entries * 16, 24
    b = static + table + ni * (model.nv * outb + model.nb * 48 + model.nm * 4)
    print(f"{name:8s} ni={ni:5d} {ms * 1e3:9.1f} us  {ni * model.nv / (ms * 1e-3) / 1e9:8.2f} Gverts/s  "
          f"{b / (ms * 1e-3) / 1e9:7.0f} GB/s = {b / (ms * 1e-3) / 1e9 / HBM:.3f} of 8 TB/s   "
          f"(entries {i.n_entries} padded {i.n_entries_padded}, slots {i.n_slots}, tile bones {i.max_tile_bones})", flush=True)
    for x in (d_pal, d_w, d_a, d_b):
        x.free()
    dm.close()


def main():
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c2", "c5", "c3p"]
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
    if "c2" in which or "c3p" in which or "c2x1" in which:
        m = synth.make_config("config2_50k")
        # FB_WINDOW=12: the same model with bone ids drawn from a window of 12 (<= 16 bones per tile: the palette gathers cannot
        # conflict); FB_PRESORT=1: vertices pre-sorted the way the plan sorts them (the scatter into the LDS image is the identity)
        if os.environ.get("FB_WINDOW"):
            c = synth.CONFIGS["config2_50k"]
            m = synth.make_model(c["nv"], c["nb"], c["nm"], c["k"], c["seed"], window=int(os.environ["FB_WINDOW"]))
        if os.environ.get("FB_PRESORT") == "1":
            m = synth.presort_by_class(m)
        if "c2" in which:
            run("c2x64", m, 64, np.arange(64), api.OUT_SOA, iters)
        if "c2x1" in which:
            run("c2x1", m, 1, np.arange(1) + 17, api.OUT_SOA, iters * 10)
        if "c3p" in which:
            run("c3prime", m, 1024, (np.arange(1024) * 7) % 600, api.OUT_SOA, max(iters // 2, 5))
        if "c3" in which:          # the shared-morph crowd kernel on the same box, for reference (morph pass skipped after the first call)
            run("c3crowd", m, 1024, (np.arange(1024) * 7) % 600, api.OUT_SOA, max(iters // 2, 5), shared=True)
    if "c5" in which:
        m5 = synth.make_config("config5_256k")
        run("c5x64", m5, 64, np.arange(64), api.OUT_SOA_POS16, iters, f16=True)
    if "c5x1" in which:
        m5 = synth.make_config("config5_256k")
        run("c5x1", m5, 1, np.arange(1) + 17, api.OUT_SOA_POS16, iters * 10, f16=True)


if __name__ == "__main__":
    main()
