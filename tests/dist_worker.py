"""Worker for tests/test_distributed_gloo.py: one rank of an instance-sharded crowd on CPU.
The HIP path cannot run here (no GPU), so the ORACLE stands in as the deformer -- what is under
test is the N>1 plumbing: sharding, per-instance posing by global id, barriers, max-reduce, gather."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle  # noqa: E402
from simple_mmd_renderer_amd import synth  # noqa: E402
from simple_mmd_renderer_amd.crowd import Rendezvous, crowd_frames, shard_instances  # noqa: E402


def main():
    out_path, total = sys.argv[1], int(sys.argv[2])
    rv = Rendezvous()
    model = synth.make_model(1500, 40, 6, 100, seed=99)
    rates = synth.morph_weights(model.nm, 30)[0]
    lo, hi = shard_instances(total, rv.world, rv.rank)
    pals = synth.make_palettes(model, crowd_frames(lo, hi))
    orc = Oracle()
    skin = orc.normalize(model)
    vimg = orc.morph(model, rates)
    if rv.rank == 1:
        time.sleep(0.3)                                      # a rank that is late for the barrier
    spins = rv.barrier_while(lambda: time.sleep(0.01))       # the others stay busy meanwhile
    rv.barrier()
    t0 = time.perf_counter()
    sums = []
    for i in range(hi - lo):
        pos, nrm = orc.skin(model, pals[i], vimg, skin)
        sums.append(synth.checksum64(np.concatenate([pos.ravel(), nrm.ravel()])))
    elapsed = time.perf_counter() - t0 + 0.01 * rv.rank     # make the ranks' times differ
    rv.barrier()
    slowest = rv.max(elapsed)
    n_total = rv.sum(hi - lo)
    gathered = rv.gather_u64(sums)
    if rv.rank == 0:
        json.dump(dict(world=rv.world, ranges=[shard_instances(total, rv.world, r) for r in range(rv.world)],
                       checksums=[c for part in gathered for c in part], slowest=slowest,
                       rank0_elapsed=elapsed, n_total=n_total, rank0_busy_calls=spins), open(out_path, "w"))
    rv.close()


if __name__ == "__main__":
    main()
