#!/usr/bin/env python3
"""Diagnose a soak mismatch of the device bone solve: for the given rig seeds (tools/soak_rig.py numbering) find
the instances that differ from the C oracle, re-solve them a few times (a race would not repeat), and solve the
same inputs with the schedule switched off (MMDX_SOLVE_SEQUENTIAL=1, one event per round) in a child process:
equal results there mean the difference is arithmetic, not ordering.  With tools/archive/probes/libm_probe built (see
tools/archive/probes/libm_probe.hip) every transcendental call the oracle made for the instance is replayed on the device to show
which call the two libms disagree on.
    python tools/archive/probes/rig_mismatch_probe.py first_seed last_seed [instances]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import synth, vmd  # noqa: E402


def case(seed, ni):
    nb = 24 + seed % 60
    rig = synth.make_ik_rig(nb, 1000 + seed, n_ik=1 + seed % 6, n_append=seed % 7)
    morphs = synth.make_bone_morphs(nb, 2000 + seed) if seed % 2 else None
    rng = np.random.RandomState(seed)
    poses = np.zeros((ni, nb, 8), np.float32)
    poses[..., 0:3] = rng.uniform(-1.5, 1.5, (ni, nb, 3))
    q = rng.normal(size=(ni, nb, 4))
    poses[..., 4:8] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    rates = rng.choice([0, 5e-8, 0.3, 1.0, 1.7, -0.5], (ni, morphs["type"].size)).astype(np.float32) if morphs else None
    return rig, morphs, poses, rates


def solve(seed, ni):
    rig, morphs, poses, rates = case(seed, ni)
    sk = vmd.Skeleton(*rig, morphs)
    return sk, sk.solve(poses, morph_weights=rates)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        seed, ni, path = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
        np.save(path, solve(seed, ni)[1])
        sys.exit(0)
    from oracle.pyoracle import Oracle
    o = Oracle()
    s0, s1 = int(sys.argv[1]), int(sys.argv[2])
    ni = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    for seed in range(s0, s1 + 1):
        rig, morphs, poses, rates = case(seed, ni)
        sk, got = solve(seed, ni)
        bad = []
        for i in range(ni):
            want = o.bone_solve_full(rig[0], rig[1], poses[i], rig[2], rig[3], rig[4], rig[5], rig[6], morphs,
                                     rates[i] if morphs else None)
            d = (got[i].view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got[i]) & np.isnan(want))
            if d.any():
                bad.append((i, want, d))
        if not bad:
            continue
        print(f"seed {seed}: {sk.info}")
        again = [solve(seed, ni)[1] for _ in range(4)]
        tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rig_seq_{seed}.npy")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(seed), str(ni), tmp], check=True,
                       env=dict(os.environ, MMDX_SOLVE_SEQUENTIAL="1"))
        seq = np.load(tmp)
        for i, want, d in bad:
            bones = np.unique(np.nonzero(d.reshape(d.shape[0], -1))[0])
            print(f"  instance {i}: {int(d.sum())} values differ from the oracle, bones {bones.tolist()}, "
                  f"max abs {np.abs(got[i].astype(np.float64) - want)[d].max():.3g}")
            print(f"    repeats identical to the first solve: {[bool(np.array_equal(a[i].view(np.uint32), got[i].view(np.uint32))) for a in again]}")
            print(f"    sequential schedule identical to the round schedule: {bool(np.array_equal(seq[i].view(np.uint32), got[i].view(np.uint32)))}; "
                  f"sequential vs oracle: {int((seq[i].view(np.uint32) != want.view(np.uint32)).sum())} values differ")
            # every transcendental call the oracle made for this instance, re-evaluated on the device
            probe = os.path.join(ROOT, "tools", "libm_probe")
            if os.path.exists(probe):
                rec = o.trace_libm(lambda: o.bone_solve_full(rig[0], rig[1], poses[i], rig[2], rig[3], rig[4], rig[5], rig[6],
                                                             morphs, rates[i] if morphs else None))
                path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rig_libm_{seed}_{i}.bin")
                rec.tofile(path)
                r = subprocess.run([probe, path], capture_output=True, text=True)
                print("    " + (r.stdout + r.stderr).strip().replace("\n", "\n    "))
