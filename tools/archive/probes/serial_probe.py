#!/usr/bin/env python3
"""Where the ordered bone solver's time goes: the same 300-bone rig with (a) no IK / append (parallel FK),
(b) one append bone (ordered solver, no IK), (c) 8 IK chains with loop counts capped at 0 / 3 / 40 / 256.
(The labels say "serial": the solver's ABI name, MMDX_SOLVER_SERIAL.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from simple_mmd_renderer_amd import _capi as api, synth, vmd
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer
m = synth.make_config("config2_50k")
dm = DeformModel(m)
ni, nb = 1024, m.nb
rng = np.random.RandomState(0)
poses = np.zeros((ni, nb, 8), np.float32)
poses[..., 0:3] = rng.uniform(-0.5, 0.5, (ni, nb, 3))
q = rng.normal(size=(ni, nb, 4)); poses[..., 4:8] = q / np.linalg.norm(q, axis=-1, keepdims=True)
d_pose, d_pal = DeviceBuffer.from_numpy(poses), DeviceBuffer(ni * nb * 64)
def t(sk, iters=10):
    return bench.time_calls(dm, lambda: sk.solve_device(ni, d_pose.ptr, d_pal.ptr, dm), iters) * 1e3
rest, parent, level, flags, ap, ar, ik = synth.make_ik_rig(nb, 3003, n_ik=8, n_append=12, post_physics=0.0, levels=1)
print("parallel FK            %8.1f us" % t(vmd.Skeleton(rest, parent), 50))
f1 = np.zeros(nb, np.uint16); f1[5] = 0x0200
print("serial, 1 append bone  %8.1f us" % t(vmd.Skeleton(rest, parent, None, f1, ap * 0 + 2, ar)))
for cap in (0, 3, 40, 256):
    ik2 = dict(ik, loop=np.minimum(np.where(ik["loop"] < 0, 256, ik["loop"]), cap).astype(np.int32))
    sk = vmd.Skeleton(rest, parent, level, flags, ap, ar, ik2)
    print("serial, 8 IK chains, loop cap %3d: %8.1f us  (loops %s)" % (cap, t(sk), ik2["loop"][ik2["loop"] > 0].tolist()))
