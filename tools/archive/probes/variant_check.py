#!/usr/bin/env python3
"""Parity of BUILD VARIANTS of libmmdx.so (tools/archive/probes/store_policy_ab.py loads them side by side): every named library runs the
config-3 crowd in both output layouts (and the per-instance-morph form) and is compared with the oracle on a sample of
instances, twice (run-to-run determinism).      python tools/archive/probes/variant_check.py name=path.so ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from store_policy_ab import load, use  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402
from simple_mmd_renderer_amd import _capi as api, synth  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402


def main():
    libs = [("shipped", api.lib())] + [(n, load(os.path.abspath(p))) for n, p in (a.split("=", 1) for a in sys.argv[1:])]
    orc = Oracle()
    m = synth.make_config("config3_crowd")
    ni = 1024
    pals = synth.make_palettes(m, (np.arange(ni) * 3) % 1801)
    shared = synth.morph_weights(m.nm, 30)[0]
    per = synth.morph_weights(m.nm, (np.arange(ni) * 7) % 600)
    skin, vimg = orc.normalize(m), orc.morph(m, shared)
    sample = sorted(set(range(0, ni, 97)) | {1, ni - 1})
    with use(libs[0][1]):
        d_pal, d_ws, d_wp = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(shared), DeviceBuffer.from_numpy(per)
        d_a, d_b = DeviceBuffer(ni * m.nv * 32), DeviceBuffer(ni * m.nv * 12)
    base = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE
    bad = 0
    for name, l in libs:
        with use(l):
            dm = DeformModel(m)
            for form, layout, w, flags in (("v32 shared", api.OUT_VERTEX32, d_ws, base | api.WEIGHTS_SHARED),
                                           ("soa shared", api.OUT_SOA, d_ws, base | api.WEIGHTS_SHARED),
                                           ("soa per-instance", api.OUT_SOA, d_wp, base),
                                           ("v32 per-instance", api.OUT_VERTEX32, d_wp, base)):
                for rep in range(2):
                    d_a.memset(0xEE); d_b.memset(0xEE)
                    dm.deform_batched_raw(ni, w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr if layout == api.OUT_SOA else None, layout, flags,
                                          0.1 if layout == api.OUT_VERTEX32 else 1.0)
                    dm.sync()
                    wrong = []
                    for i in sample:
                        ep, en = orc.skin(m, pals[i], vimg if "shared" in form else orc.morph(m, per[i]), skin)
                        if layout == api.OUT_VERTEX32:
                            got = d_a.download((m.nv, 8), np.float32, offset=i * m.nv * 32)
                            ok = np.array_equal(got.view(np.uint32), orc.repack32(m, ep, en, 0.1).reshape(m.nv, 8).view(np.uint32))
                        else:
                            ok = (np.array_equal(d_a.download((m.nv, 3), np.float32, offset=i * m.nv * 12).view(np.uint32), ep.view(np.uint32)) and
                                  np.array_equal(d_b.download((m.nv, 3), np.float32, offset=i * m.nv * 12).view(np.uint32), en.view(np.uint32)))
                        if not ok:
                            wrong.append(i)
                    bad += len(wrong)
                    print(f"{name:12s} {form:18s} run {rep}: {'bit-exact' if not wrong else 'MISMATCH in instances ' + str(wrong)}", flush=True)
            dm.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
