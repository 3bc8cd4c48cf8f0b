"""Multi-GPU candidate sharding: one process per GPU.  On the GPU box the scores are collected by the ENGINE's own RCCL
communicator (rk_comm_*, llmrankers/_runtime.py); `torch.distributed` only launches the ranks and carries the RCCL id.
The helpers here are the partitioning and the host-side gather used when the runtime has no communicator (CPU tests on
'gloo').

The reference has no data parallelism at all (multi-GPU there = accelerate's device_map='auto' layer placement,
ref: llmrankers/pointwise.py:21; README.md:357).  Here the passages of one query are independent given the
query (pointwise scoring), so the candidate list is cut into `world` contiguous chunks, every rank scores its
chunk on its own engine (weights replicated: 1.5 GB fp16 for flan-t5-large against 288 GB of HBM), and ONE
all_gather of <= ceil(n/world) fp32 per rank collects the scores; every rank then holds the full score vector
and sorts identically.  Setwise heapsort is a dependency chain of compares: replicas only (shard QUERIES).
"""
from __future__ import annotations

from typing import List, Tuple

import os

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


def all_gather_flat(local: np.ndarray, width: int) -> np.ndarray:
    """[world, width] float32: every rank's `local` (<= width values, zero padded) through ONE torch.distributed
    all_gather on the process group's own backend — the host-side path used when the runtime has no engine-owned
    RCCL communicator (CPU tests on gloo; a GPU run goes through rk_comm_all_gather_slot instead)."""
    import torch
    import torch.distributed as dist
    rank, ws = world()
    local = np.asarray(local, dtype=np.float32).reshape(-1)
    if dist.get_backend() == "nccl":
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0") or 0))
    else:
        device = torch.device("cpu")
    buf = torch.zeros(width, dtype=torch.float32, device=device)
    buf[:len(local)] = torch.as_tensor(local, device=device)
    out = torch.empty(ws * width, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, buf)
    return out.cpu().numpy().reshape(ws, width)
