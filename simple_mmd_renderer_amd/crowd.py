"""Instance-sharded crowds across the GPUs of one node: one process per GPU, no data-path collective.

Every (instance, vertex) output depends only on that instance's palette and morph weights and on the
static model, so the crowd shards embarrassingly: rank r of G deforms instances
[r*NI/G, (r+1)*NI/G) on its own GPU with its own copy of the static streams; outputs stay in that
GPU's HBM.  torch.distributed (gloo, over 127.0.0.1) is used for rendezvous only: start/stop barriers,
the max-over-ranks reduction of the elapsed time, and gathering per-rank checksums in tests.  There is
no RCCL traffic on the timed path because the path has no exchange step.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np


def shard_instances(total_instances: int, world: int, rank: int) -> Tuple[int, int]:
    """Half-open instance range of `rank`: [rank*NI/G, (rank+1)*NI/G).  Ranges tile [0, NI) exactly."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * total_instances) // world, ((rank + 1) * total_instances) // world


def crowd_frames(lo: int, hi: int) -> np.ndarray:
    """Animation phase of each instance of the synthetic crowd (a function of the GLOBAL instance id,
    so a sharded run poses every instance exactly as a single-GPU run would)."""
    return (np.arange(lo, hi) * 3) % 1801


class Rendezvous:
    """Thin wrapper over torch.distributed(gloo); a no-op for world size 1 (torch is not imported)."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self._dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            # gloo announces its connections on C++ stdout; callers (bench.py) own stdout for their
            # one JSON line, so point fd 1 at stderr while the process group comes up
            import sys
            sys.stdout.flush()
            saved = os.dup(1)
            try:
                os.dup2(2, 1)
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
                dist.barrier()
            finally:
                os.dup2(saved, 1)
                os.close(saved)
            self._dist = dist

    def barrier(self) -> None:
        if self._dist is not None:
            self._dist.barrier()

    def barrier_while(self, busy) -> int:
        """A barrier during which this rank keeps calling busy() (e.g. a few untimed steps) until every rank has
        arrived: ranks that finish their set-up early do not let their GPU idle -- and drop its clocks -- right
        before a timed region.  Returns how often busy() ran."""
        if self._dist is None:
            return 0
        work = self._dist.barrier(async_op=True)
        n = 0
        while not work.is_completed():
            busy()
            n += 1
        work.wait()
        return n

    def max(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def gather_u64(self, values: List[int]) -> List[List[int]]:
        """All ranks' lists of 64-bit checksums, on every rank (lists may differ in length)."""
        if self._dist is None:
            return [list(values)]
        out = [None] * self.world
        self._dist.all_gather_object(out, [int(v) for v in values])
        return out

    def close(self) -> None:
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None


class InProcessCrowd:
    """The in-process form of the sharded crowd: ONE process, one host thread + one model
    handle (with its own HIP stream) per shard, shard s on device s % n_devices.  Every worker thread selects its
    device (mmdx_device_select binds the calling thread, like hipSetDevice), creates its copy of the static
    streams there and deforms its slice [s*NI/S, (s+1)*NI/S) of the instances; no data crosses shards.  ctypes
    releases the GIL inside the library, so the shards' calls really run concurrently."""

    def __init__(self, flat, n_shards: int, **model_kw):
        import concurrent.futures as cf
        from .engine import DeformModel, device_count, device_select
        ndev = device_count()
        if ndev < 1:
            raise RuntimeError("InProcessCrowd: no HIP device visible (this engine has no CPU path)")
        self.n_shards = int(n_shards)
        self.devices = [s % ndev for s in range(self.n_shards)]
        self._pools = [cf.ThreadPoolExecutor(1) for _ in range(self.n_shards)]   # one persistent thread per shard

        def make(dev):
            device_select(dev)
            return DeformModel(flat, **model_kw)
        self.models = [f.result() for f in [p.submit(make, d) for p, d in zip(self._pools, self.devices)]]

    def deform(self, weights, palettes, shared_weights: bool = True, **kw):
        """Host arrays in, host arrays out; instance i is handled by the shard whose range holds i."""
        ni = palettes.shape[0]
        parts = [shard_instances(ni, self.n_shards, s) for s in range(self.n_shards)]

        def run(s):
            lo, hi = parts[s]
            if hi == lo:
                return None
            w = weights if shared_weights else weights[lo:hi]
            return self.models[s].deform_batched(w, palettes[lo:hi], shared_weights=shared_weights, **kw)
        outs = [f.result() for f in [self._pools[s].submit(run, s) for s in range(self.n_shards)]]
        outs = [o for o in outs if o is not None]
        if isinstance(outs[0], tuple):
            return tuple(np.concatenate([o[k] for o in outs]) for k in range(len(outs[0])))
        return np.concatenate(outs)

    def close(self) -> None:
        for p, m in zip(self._pools, self.models):
            p.submit(m.close).result()
            p.shutdown()
        self.models, self._pools = [], []
