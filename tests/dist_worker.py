"""Worker for tests/test_distributed_gloo.py (CPU) and tests/test_multi_device.py (GPU): one rank of an
instance-sharded crowd.
  * default (CPU, no GPU in the build container): the ORACLE stands in as the deformer -- what is under test is
    the N>1 plumbing: sharding, per-instance posing by global id, barriers, max-reduce, gather;
  * --hip (GPU box): the PRODUCT deforms -- every rank selects device LOCAL_RANK % n_devices, builds its own
    DeformModel there and runs its shard through mmdx_deform_batched with everything resident in HBM."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_mmd_renderer_amd import synth  # noqa: E402


def main():
    out_path, total = sys.argv[1], int(sys.argv[2])
    hip = "--hip" in sys.argv[3:]
    model = synth.make_model(1500, 40, 6, 100, seed=99)
    rates = synth.morph_weights(model.nm, 30)[0]
    if hip:
        # HIP first, torch (inside Rendezvous) second -- see bench.py
        from simple_mmd_renderer_amd import _capi as api
        from simple_mmd_renderer_amd.engine import DeformModel, DeviceBuffer, device_count, device_select
        ndev = device_count()
        assert ndev >= 1, "no HIP device visible"
        device = int(os.environ.get("LOCAL_RANK", "0")) % ndev
        device_select(device)
        DeviceBuffer(256).free()
    from simple_mmd_renderer_amd.crowd import Rendezvous, crowd_frames, shard_instances
    rv = Rendezvous()
    lo, hi = shard_instances(total, rv.world, rv.rank)
    pals = synth.make_palettes(model, crowd_frames(lo, hi))
    if hip:
        dm = DeformModel(model)
        assert dm.info.device_ordinal == device
        d_pal, d_w = DeviceBuffer.from_numpy(pals), DeviceBuffer.from_numpy(rates)
        sa, sb = dm.out_sizes(api.OUT_SOA, max(hi - lo, 1))
        d_a, d_b = DeviceBuffer(sa), DeviceBuffer(sb)
    else:
        from oracle.pyoracle import Oracle
        orc = Oracle()
        skin = orc.normalize(model)
        vimg = orc.morph(model, rates)
    if rv.rank == 1:
        time.sleep(0.3)                                      # a rank that is late for the barrier
    spins = rv.barrier_while(lambda: time.sleep(0.01))       # the others stay busy meanwhile
    rv.barrier()
    t0 = time.perf_counter()
    sums = []
    if hip:
        if hi > lo:
            flags = api.PALETTE_ON_DEVICE | api.WEIGHTS_ON_DEVICE | api.OUT_ON_DEVICE | api.WEIGHTS_SHARED
            dm.deform_batched_raw(hi - lo, d_w.ptr, d_pal.ptr, d_a.ptr, d_b.ptr, api.OUT_SOA, flags)
            dm.sync()
            pos = d_a.download((hi - lo, model.nv, 3), np.float32)
            nrm = d_b.download((hi - lo, model.nv, 3), np.float32)
            sums = [synth.checksum64(np.concatenate([pos[i].ravel(), nrm[i].ravel()])) for i in range(hi - lo)]
    else:
        for i in range(hi - lo):
            pos, nrm = orc.skin(model, pals[i], vimg, skin)
            sums.append(synth.checksum64(np.concatenate([pos.ravel(), nrm.ravel()])))
    elapsed = time.perf_counter() - t0 + 0.01 * rv.rank     # make the ranks' times differ
    rv.barrier()
    slowest = rv.max(elapsed)
    n_total = rv.sum(hi - lo)
    gathered = rv.gather_u64(sums)
    devices = rv.gather_u64([device if hip else -1])
    if rv.rank == 0:
        json.dump(dict(world=rv.world, ranges=[shard_instances(total, rv.world, r) for r in range(rv.world)],
                       checksums=[c for part in gathered for c in part], slowest=slowest,
                       rank0_elapsed=elapsed, n_total=n_total, rank0_busy_calls=spins,
                       devices=[d[0] for d in devices]), open(out_path, "w"))
    if hip:
        for b in (d_pal, d_w, d_a, d_b):
            b.free()
        dm.close()
    rv.close()


if __name__ == "__main__":
    main()
