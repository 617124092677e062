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
names = [f"b{i}" for i in range(m.nb)]
vm = vmdmod.Vmd(vmdmod.write_vmd(synth.make_bone_keys(names, 303, keys_per=20, span=600), []))
bm = vm.bind_bones(names)
rig = synth.make_ik_rig(m.nb, 3003, n_ik=8, n_append=12, post_physics=0.0, levels=1)
rigs = {"fk": vmdmod.Skeleton(m.bone_pos, np.asarray(m.bone_parent, np.int32)),
        "append": vmdmod.Skeleton(rig[0], rig[1], rig[2], (np.asarray(rig[3]) & ~np.uint16(0x20)).astype(np.uint16), rig[4], rig[5]),
        "ik": vmdmod.Skeleton(*rig)}
# the same IK rig with its iteration limits capped at 40, the value the leg-IK bones of MMD models usually carry (the rig above
# draws 300 -> 256 for some chains: one chain of 256 x 3 link steps is what the "ik" row waits for)
rig40 = list(rig)
_l = rig[6]["loop"].astype(np.int64)
rig40[6] = dict(rig[6], loop=np.where((_l < 0) | (_l > 40), 40, _l).astype(rig[6]["loop"].dtype))   # (-1 reads as "no limit")
rigs["ik40"] = vmdmod.Skeleton(*rig40)
# RIG_NI=1024,4096,16384: the same rigs at larger crowds (task 7: 1024 instances are 256 waves, one per CU on a
# quarter of the SIMDs -- what does the solver do when the chip is filled?); RIG_ONLY=ik restricts the rigs
for ni in [int(x) for x in os.environ.get("RIG_NI", "1024").split(",")]:
    d_fr = DeviceBuffer.from_numpy(((np.arange(ni) * 7) % 600).astype(np.uint32))
    d_pose, d_pal = DeviceBuffer(ni * m.nb * 32), DeviceBuffer(ni * m.nb * 64)
    for name, sk in rigs.items():
        if os.environ.get("RIG_ONLY") and name not in os.environ["RIG_ONLY"].split(","):
            continue
        iters = max(2, (10 if name.startswith("ik") else 30) * 1024 // ni)
        ms = bench.time_calls(dm, lambda: (bm.eval_device(ni, d_fr.ptr, d_pose.ptr, dm), sk.solve_device(ni, d_pose.ptr, d_pal.ptr, dm)),
                              iters)
        ms1 = bench.time_calls(dm, lambda: sk.solve_motion_device(bm, ni, d_fr.ptr, d_pal.ptr, dm), iters)
        print(f"{name:7s} rig: poses + palettes of {ni:6d} x {m.nb} bones  {ms * 1e3:9.1f} us   as one call {ms1 * 1e3:9.1f} us   "
              f"{ni * 1e-6 / (ms * 1e-3):8.2f} M palettes/s   solver {sk.info['solver']} rounds {sk.info['n_solve_rounds']}", flush=True)
    for b in (d_fr, d_pose, d_pal):
        b.free()
