#!/usr/bin/env python3
"""The palette producers alone (bone tracks -> poses -> palettes for 1024 instances x 300 bones): FK rig, append rig, IK rig.
Small enough to run under rocprofv3 (tools/profile_round.sh)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from simple_mmd_renderer_amd import synth, vmd as vmdmod  # noqa: E402
from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer  # noqa: E402

m = synth.make_config("config3_crowd")
dm = DeformModel(m)
ni = 1024
names = [f"b{i}" for i in range(m.nb)]
vm = vmdmod.Vmd(vmdmod.write_vmd(synth.make_bone_keys(names, 303, keys_per=20, span=600), []))
bm = vm.bind_bones(names)
d_fr = DeviceBuffer.from_numpy(((np.arange(ni) * 7) % 600).astype(np.uint32))
d_pose, d_pal = DeviceBuffer(ni * m.nb * 32), DeviceBuffer(ni * m.nb * 64)
rig = synth.make_ik_rig(m.nb, 3003, n_ik=8, n_append=12, post_physics=0.0, levels=1)
rigs = {"fk": vmdmod.Skeleton(m.bone_pos, np.asarray(m.bone_parent, np.int32)),
        "append": vmdmod.Skeleton(rig[0], rig[1], rig[2], (np.asarray(rig[3]) & ~np.uint16(0x20)).astype(np.uint16), rig[4], rig[5]),
        "ik": vmdmod.Skeleton(*rig)}
for name, sk in rigs.items():
    ms = bench.time_calls(dm, lambda: (bm.eval_device(ni, d_fr.ptr, d_pose.ptr, dm), sk.solve_device(ni, d_pose.ptr, d_pal.ptr, dm)),
                          10 if name == "ik" else 30)
    ms1 = bench.time_calls(dm, lambda: sk.solve_motion_device(bm, ni, d_fr.ptr, d_pal.ptr, dm), 10 if name == "ik" else 30)
    print(f"{name:7s} rig: poses + palettes of {ni} x {m.nb} bones  {ms * 1e3:9.1f} us   as one call {ms1 * 1e3:9.1f} us   "
          f"solver {sk.info['solver']} rounds {sk.info['n_solve_rounds']}", flush=True)
