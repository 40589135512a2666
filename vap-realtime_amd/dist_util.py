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
    if "MASTER_PORT" not in os.environ:
        raise RuntimeError("MASTER_PORT is not set: launch the ranks with torch.distributed.run (bench.py --gpus N does that itself)")
    if backend == "nccl":
        # RCCL for CUDA tensors (the barrier / max-over-ranks of the bench), gloo for CPU tensors (shard bookkeeping); if RCCL
        # cannot come up in this environment the control plane still works over gloo — there is no collective on the data path
        try:
            import datetime
            kw = {"device_id": device} if device is not None else {}
            # a finite timeout: a collective that cannot complete fails the run instead of hanging it
            dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600), **kw)
        except Exception as e:                                    # noqa: BLE001
            import sys
            print(f"[dist_util] RCCL init failed ({e}); falling back to gloo for the bench's barrier", file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def _parse_cpulist(txt: str):
    cpus = []
    for part in txt.strip().split(","):
        if "-" in part:
            lo, hi = part.split("-")
            cpus.extend(range(int(lo), int(hi) + 1))
        elif part:
            cpus.append(int(part))
    return cpus


def pin_rank_to_cores(local_rank: int, local_world: int, device_index=None, shared: bool = False):
    """Pin this process to host cores near its GPU: the NUMA node of the GPU's PCI function when sysfs exposes it
    (shared evenly by the ranks of that node), else an even split of the visible cores.  The host side of a shard (audio
    staging, result fan-out) then stays off the other ranks' cores.  ``shared`` = every rank uses device ``device_index``
    (``--share-gpu``: said explicitly, so that all ranks — rank 0 included, whose device index equals its rank — compute the SAME
    rank -> NUMA-node table and their core splits cannot overlap); otherwise rank r uses device r (clamped to the visible devices).
    Ranks on one node share its cores evenly.  Returns the core list or None (single rank / no affinity support)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        avail = sorted(os.sched_getaffinity(0))
        cores = None
        try:
            import torch
            if torch.cuda.is_available():
                nodes = []
                ndev = torch.cuda.device_count()
                for r in range(local_world):
                    d = int(device_index or 0) if shared else min(r, ndev - 1)
                    p = torch.cuda.get_device_properties(d)
                    bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
                    with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
                        nodes.append(int(f.read()))
                node = nodes[local_rank]
                if node >= 0:
                    with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                        ncpus = [c for c in _parse_cpulist(f.read()) if c in avail]
                    peers = [r for r in range(local_world) if nodes[r] == node]
                    k, m = peers.index(local_rank), len(peers)
                    per = len(ncpus) // m
                    if per >= 1:
                        cores = ncpus[k * per:(k + 1) * per]
        except Exception:
            cores = None
        if not cores:
            per = max(1, len(avail) // local_world)
            cores = avail[local_rank * per:(local_rank + 1) * per] or avail
        os.sched_setaffinity(0, cores)
        return [cores[0], cores[-1], len(cores)]
    except Exception:
        return None


def gpu_node_cores(device_index: int = 0):
    """Cores (of this process's affinity mask) on the NUMA node of GPU ``device_index``'s PCI function, or every allowed core when sysfs does
    not say.  Used to place the TCP front-end's threads next to the GPU they feed and load generators away from them."""
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read())
        if node >= 0:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                cores = [c for c in _parse_cpulist(f.read()) if c in avail]
            if cores:
                return cores
    except Exception:                                             # noqa: BLE001
        pass
    return avail


def _smt_siblings(cpu: int):
    try:
        with open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list") as f:
            return set(_parse_cpulist(f.read().replace("-", "-")))
    except Exception:                                             # noqa: BLE001
        return {cpu}


def front_end_placement(device_index: int, n_threads: int, skip: int = 0):
    """((first, count) for vapx_ingest_config / NativeServer(cores=...), cores left for everybody else): ``n_threads`` consecutive cores at
    the TOP of the longest run of the GPU's NUMA node (core 0 and its neighbours take the host's interrupts and housekeeping; ``skip`` cores
    below the top belong to front-ends placed earlier on the same node), and the allowed cores outside that range AND outside its SMT
    siblings — preferring other NUMA nodes' cores first — for load generators.  (None, all cores) when there are not even two cores to
    tell apart."""
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    node = gpu_node_cores(device_index)
    if len(avail) < 2 or not node:
        return None, avail
    runs, cur = [], [node[0]]
    for c in node[1:]:
        if c == cur[-1] + 1:
            cur.append(c)
        else:
            runs.append(cur)
            cur = [c]
    runs.append(cur)
    run = max(runs, key=len)                                      # (ties: the first = the lower-numbered = the physical cores, not their SMT twins)
    if skip >= len(run):
        skip = 0                                                  # (more front-ends than the node has cores for: share from the top)
    run = run[:len(run) - skip]
    count = max(1, min(len(run), n_threads, len(avail) - 1))
    mine = run[len(run) - count:]
    blocked = set()
    for c in mine:
        blocked |= _smt_siblings(c)
    blocked |= set(mine)
    others = [c for c in avail if c not in blocked and c not in node] + [c for c in node if c not in blocked and c in avail]
    return (mine[0], count), (others or avail)


def barrier(dist, device_sync=None):
    """Device synchronise, rendezvous of all ranks, device synchronise.  With a CUDA backend the rendezvous is an all-reduce
    of one CUDA scalar (RCCL over xGMI), otherwise a CPU barrier.  Either way the HOST returns only after every rank has
    arrived (the all-reduce result is read back), also without ``device_sync``."""
    if device_sync:
        device_sync()
    if dist is not None:
        done = False
        try:
            import torch
            if torch.cuda.is_available() and "nccl" in str(dist.get_backend()):
                t = torch.ones(1, device="cuda")
                dist.all_reduce(t)
                t.item()                                          # block the host: an enqueued collective is not a barrier yet
                done = True
        except Exception:                                         # noqa: BLE001
            done = False
        if not done:
            dist.barrier()
    if device_sync:
        device_sync()


def ranks_seen(dist, device: Optional[str] = None):
    """{"backend": ..., "n": ...}: an all-reduce (sum) of one 1 per rank over the backend that carries the bench's barrier — with "nccl"
    (RCCL) on a CUDA tensor, i.e. over xGMI: the record of a multi-GPU run then PROVES that N ranks met over RCCL, whatever the launcher
    claimed.  Single process: {"backend": "none", "n": 1}."""
    if dist is None:
        return {"backend": "none", "n": 1}
    import torch
    backend = str(dist.get_backend())
    dev = device or "cpu"
    if dev != "cpu" and "nccl" not in backend:
        dev = "cpu"
    t = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(t)
    return {"backend": "nccl" if (dev != "cpu" and "nccl" in backend) else "gloo", "n": int(round(float(t.item())))}


def max_over_ranks(dist, value: float, device: Optional[str] = None) -> float:
    if dist is None:
        return value
    import torch
    dev = device or "cpu"
    if dev != "cpu" and "nccl" not in str(dist.get_backend()):
        dev = "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_ints(dist, values, device: Optional[str] = None):
    """all_gather of equal-length int lists (used to verify shard coverage)."""
    if dist is None:
        return [list(values)]
    import torch
    if device and device != "cpu" and "nccl" not in str(dist.get_backend()):
        device = "cpu"
    t = torch.tensor(list(values), dtype=torch.int64, device=device or "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def gather_floats(dist, values, device: Optional[str] = None):
    """all_gather of equal-length float lists: one list per rank (the bench's per-rank rates next to the max-over-ranks time)."""
    if dist is None:
        return [list(values)]
    import torch
    if device and device != "cpu" and "nccl" not in str(dist.get_backend()):
        device = "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=device or "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]
