"""Multi-process glue for the bench / multi-GPU serving: one process per GPU, torch.distributed
(backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).  The data path has NO collective — streams
are sharded (sharding.py) — so the only communication is the bench's barrier and the
max-over-ranks reduction of the timed region."""
from __future__ import annotations

import os
from typing import Optional, Tuple


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str, device=None):
    """Returns torch.distributed or None for a single process."""
    rank, _, world = env_rank()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def barrier(dist, device_sync=None):
    if device_sync:
        device_sync()
    if dist is not None:
        dist.barrier()
    if device_sync:
        device_sync()


def max_over_ranks(dist, value: float, device: Optional[str] = None) -> float:
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_ints(dist, values, device: Optional[str] = None):
    """all_gather of equal-length int lists (used to verify shard coverage)."""
    if dist is None:
        return [list(values)]
    import torch
    t = torch.tensor(list(values), dtype=torch.int64, device=device or "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]
