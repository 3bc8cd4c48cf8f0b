"""Multi-GPU candidate sharding: one process per GPU.  On the GPU box the scores are collected by the ENGINE's own RCCL
communicator (rk_comm_*, llmrankers/_runtime.py); `torch.distributed` only launches the ranks and carries the RCCL id.
The helpers here are the rank lookup and the partitioning.  (The host-side gather of the CPU tests' doubles - a
torch.distributed all_gather on 'gloo' - lives with them in tests/_stub.py: no torch collective in the product package.)

The reference has no data parallelism at all (multi-GPU there = accelerate's device_map='auto' layer placement,
ref: llmrankers/pointwise.py:21; README.md:357).  Here the passages of one query are independent given the
query (pointwise scoring), so the candidate list is cut into `world` contiguous chunks, every rank scores its
chunk on its own engine (weights replicated: 1.5 GB fp16 for flan-t5-large against 288 GB of HBM), and ONE
all_gather of <= ceil(n/world) fp32 per rank collects the scores; every rank then holds the full score vector
and sorts identically.  Setwise heapsort is a dependency chain of compares: replicas only (shard QUERIES).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def world() -> Tuple[int, int]:
    """(rank, world_size) of the initialised default process group, (0, 1) when not distributed."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def shard_bounds(n_items: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous chunks, sizes differ by at most one, larger chunks first: 100 over 8 -> 13,13,13,13,12,12,12,12."""
    base, extra = divmod(n_items, world_size)
    out, s = [], 0
    for r in range(world_size):
        e = s + base + (1 if r < extra else 0)
        out.append((s, e))
        s = e
    return out
