#!/usr/bin/env python3
"""Soak test of the device bone solve against the C oracle: many random rigs with IK chains, append bones and
bone morphs, many instances each; counts bitwise mismatches.  The serial solver's transcendentals go through the
device's double libm where the reference's go through glibc's: this is the evidence behind "bit-exact in
practice" (DESIGN.md section 7, row 3).   python tools/soak_rig.py [n_rigs] [instances]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle  # noqa: E402
from simple_mmd_renderer_amd import synth, vmd  # noqa: E402

n_rigs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ni = int(sys.argv[2]) if len(sys.argv) > 2 else 64
o = Oracle()
t0 = time.time()
bad_inst = tot_inst = bad_vals = 0
worst = 0.0
for seed in range(n_rigs):
    nb = 24 + seed % 60
    rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(nb, 1000 + seed, n_ik=1 + seed % 6, n_append=seed % 7)
    morphs = synth.make_bone_morphs(nb, 2000 + seed) if seed % 2 else None
    sk = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik, morphs)
    rng = np.random.RandomState(seed)
    poses = np.zeros((ni, nb, 8), np.float32)
    poses[..., 0:3] = rng.uniform(-1.5, 1.5, (ni, nb, 3))
    q = rng.normal(size=(ni, nb, 4))
    poses[..., 4:8] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    rates = rng.choice([0, 5e-8, 0.3, 1.0, 1.7, -0.5], (ni, morphs["type"].size)).astype(np.float32) if morphs else None
    got = sk.solve(poses, morph_weights=rates)
    for i in range(ni):
        want = o.bone_solve_full(rest, parent, poses[i], level, flags, ap, ar, ik, morphs, rates[i] if morphs else None)
        g, w = got[i].view(np.uint32), want.view(np.uint32)
        nanboth = np.isnan(got[i]) & np.isnan(want)
        diff = (g != w) & ~nanboth
        tot_inst += 1
        if diff.any():
            bad_inst += 1
            bad_vals += int(diff.sum())
            worst = max(worst, float(np.nanmax(np.abs(got[i].astype(np.float64) - want.astype(np.float64))[diff])))
    if seed % 20 == 19:
        print(f"rigs {seed + 1:5d}  instances {tot_inst:7d}  mismatching instances {bad_inst}  values {bad_vals}  "
              f"worst abs diff {worst:.3g}  ({time.time() - t0:.0f} s)", flush=True)
print(f"TOTAL rigs {n_rigs} instances {tot_inst} mismatching instances {bad_inst} values {bad_vals} worst {worst:.3g}")
