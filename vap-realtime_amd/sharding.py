"""Stream -> GPU partitioning.  A dialogue stream's state (context ring, LSTM, carry) and compute
never touch another stream (process_vap has no cross-stream term), so multi-GPU is pure sharding:
one engine handle per device, no collective on the data path."""
from __future__ import annotations

from typing import List


def shard_streams(n_streams: int, world: int, rank: int) -> List[int]:
    """Contiguous block partition: rank r owns ids [r*ceil(n/world), ...).  Blocks (not
    round-robin) keep a front-end's per-GPU staging buffers contiguous."""
    per = (n_streams + world - 1) // world
    lo = min(rank * per, n_streams)
    hi = min(lo + per, n_streams)
    return list(range(lo, hi))


def owner_of(stream_id: int, n_streams: int, world: int) -> int:
    per = (n_streams + world - 1) // world
    return stream_id // per


def local_slot(stream_id: int, n_streams: int, world: int) -> int:
    per = (n_streams + world - 1) // world
    return stream_id % per
