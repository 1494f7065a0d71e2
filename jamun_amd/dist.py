"""Multi-GPU plumbing: one process per GPU, walkers sharded across ranks, one gather per batch.

Walkers (graphs) are independent — no term of the path couples two graphs (``radius_graph`` is per batch segment,
``/root/reference/src/jamun/model/denoiser.py:140-149``) — so the data path needs NO collective.  The only exchange
is the gather of per-rank trajectory blocks, which mirrors what torchmetrics' ``dist_reduce_fx="cat"`` does for the
reference's callbacks (``src/jamun/metrics/_utils.py:40``).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the
CPU tests.
"""

from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def local_device() -> torch.device:
    if torch.cuda.is_available():
        lr = int(os.environ.get("LOCAL_RANK", 0))
        return torch.device("cuda", lr % max(torch.cuda.device_count(), 1))
    return torch.device("cpu")


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int]:
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  No-op for 1 rank."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_device())
        dist.init_process_group(backend=backend)
    return rank_world()


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block ``[lo, hi)`` of ``ceil(n/world)`` items per rank (the last ranks may get fewer / none)."""
    per = -(-n_items // world)
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def shard_range_balanced(costs: List[int], rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of ragged items balanced by cumulative ``costs`` (e.g. atoms x degree per walker)."""
    total = float(sum(costs))
    bounds = [0]
    acc, k = 0.0, 1
    for i, c in enumerate(costs):
        acc += c
        while k < world and acc >= total * k / world:
            bounds.append(i + 1)
            k += 1
    while len(bounds) < world + 1:
        bounds.append(len(costs))
    bounds[-1] = len(costs)
    return bounds[rank], bounds[rank + 1]


def gather_ragged(block: torch.Tensor, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """Gather per-rank blocks ``[n_rank, ...]`` with different leading sizes to ``dst`` (padded all-gather underneath).

    Returns the list of un-padded blocks on ``dst`` and ``None`` elsewhere.  One collective per call — the only
    communication of a sampling batch.
    """
    rank, world = rank_world()
    if world == 1:
        return [block]
    n = torch.tensor([block.shape[0]], dtype=torch.int64, device=block.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    pad = torch.zeros((nmax,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    pad[: block.shape[0]] = block
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    if rank != dst:
        return None
    return [o[:s] for o, s in zip(out, sizes)]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
