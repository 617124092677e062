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
