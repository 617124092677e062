#!/usr/bin/env python3
"""When does the long CCD chain of the bench rig start to repeat itself?  Builds the C oracle with -DMMDX_IK_CYCLE_STATS (a diagnostic
that hashes the links' IK rotations after every sweep) into /tmp, solves the rig for N instances at the bench's poses on the CPU and
prints, per half of the iterations, the distribution of the first sweep that reproduced the state of 1, 2, 3 or 4 sweeps earlier.
The poses come from the device's keyframe evaluation (the bench's motion), the solve runs on the CPU.   python tools/archive/probes/ik_cycle_probe.py [instances]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
so = "/tmp/libmmdx_oracle_cycle.so"
subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-DMMDX_IK_CYCLE_STATS",
                       "-o", so, os.path.join(ROOT, "oracle", "mmdx_oracle.c"), "-lm"])
from oracle import pyoracle  # noqa: E402
pyoracle.ORACLE_SO = so
from simple_mmd_renderer_amd import synth, vmd as vmdmod  # noqa: E402

ni = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = synth.make_config("config3_crowd")
names = [f"b{i}" for i in range(m.nb)]
keys = synth.make_bone_keys(names, 303, keys_per=20, span=600)
vm = vmdmod.Vmd(vmdmod.write_vmd(keys, []))
rig = synth.make_ik_rig(m.nb, 3003, n_ik=8, n_append=12, post_physics=0.0, levels=1)
orc = pyoracle.Oracle()
orc.lib = C.CDLL(so)
frames = ((np.arange(ni) * 7) % 600).astype(np.uint32)
bm = vm.bind_bones(names)
all_poses = bm.eval(frames)
import io, contextlib, tempfile
errf = tempfile.TemporaryFile(mode="w+")
old = os.dup(2)
os.dup2(errf.fileno(), 2)
try:
    for i in range(ni):
        orc.bone_solve_full(rig[0], rig[1], all_poses[i], rig[2], rig[3], rig[4], rig[5], rig[6])
finally:
    os.dup2(old, 2)
errf.seek(0)
rows = [l.split() for l in errf if l.startswith("ikcycle")]
a = np.array([[int(r[3]), int(r[5]), int(r[7]), int(r[9]), int(r[13]), int(r[15]), int(r[17]), int(r[19])] for r in rows])
print(len(rows), "long-chain solves")
for k, name in enumerate(["half1 period 1", "half1 period 2", "half1 period 3", "half1 period 4", "half2 period 1", "half2 period 2",
                          "half2 period 3", "half2 period 4"]):
    v = a[:, k]
    ok = v[v >= 0]
    print(f"{name}: detected in {len(ok)}/{len(v)} solves; first sweep min {ok.min() if len(ok) else -1} median "
          f"{int(np.median(ok)) if len(ok) else -1} max {ok.max() if len(ok) else -1}")
