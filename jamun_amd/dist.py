"""Multi-GPU plumbing: one process per GPU, walkers sharded across ranks, one gather per batch.

Walkers (graphs) are independent — no term of the path couples two graphs (``radius_graph`` is per batch segment,
``/root/reference/src/jamun/model/denoiser.py:140-149``) — so the data path needs NO collective.  The only exchange
is the gather of per-rank trajectory blocks to rank 0, which mirrors what torchmetrics' ``dist_reduce_fx="cat"`` does for the
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


_MAX_NDIM = 8


def gather_ragged(block: Optional[torch.Tensor], dst: int = 0, group=None, device=None) -> Optional[List[torch.Tensor]]:
    """Gather per-rank blocks ``[n_rank, ...]`` with different leading sizes to ``dst``.

    A rank with nothing to contribute passes ``None`` (or a block with zero rows).  First every rank learns every rank's
    shape (one small all-gather of int64 metadata), so all ranks agree on the trailing shape and on who sends; then each
    non-empty rank sends exactly its rows to ``dst`` (batched point-to-point: no padding, nothing lands on the other ranks —
    an all-gather would put every rank's block on every rank).  Returns the list of non-empty blocks in rank order on
    ``dst`` and ``None`` elsewhere.  Mismatching trailing shapes raise on EVERY rank (before any payload moves), so no
    rank is left waiting in a collective."""
    rank, world = rank_world()
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return [block] if block is not None and block.shape[0] > 0 else []
    # (a one-rank group that IS initialised — torchrun --nproc-per-node 1 — runs the same metadata exchange as N ranks)
    backend = dist.get_backend(group)
    if device is None:
        # nccl (RCCL) moves device memory only: a CPU block (e.g. a trajectory kept on the host by cpu_offload) is staged on
        # this rank's GPU; gloo takes either
        if backend == "nccl":
            device = block.device if block is not None and block.is_cuda else local_device()
        else:
            device = block.device if block is not None else torch.device("cpu")
    if block is not None and block.device != device:
        block = block.to(device)
    # metadata: [ndim, shape..., dtype code, error code].  Local problems (too many dimensions, unsupported dtype) are
    # REPORTED through the all-gather and raised on every rank afterwards — raising here would leave the others in the collective
    codes = {torch.float32: 0, torch.float64: 1, torch.int64: 2, torch.int32: 3}
    meta = torch.zeros(_MAX_NDIM + 3, dtype=torch.int64)
    if block is not None and block.shape[0] > 0:
        if block.ndim > _MAX_NDIM:
            meta[-1] = 1
        elif block.dtype not in codes:
            meta[-1] = 2
        else:
            meta[0] = block.ndim
            meta[1 : 1 + block.ndim] = torch.tensor(list(block.shape), dtype=torch.int64)
            meta[-2] = codes[block.dtype]
    meta = meta.to(device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = [m.tolist() for m in metas]
    errs = {r: m[-1] for r, m in enumerate(metas) if m[-1] != 0}
    if errs:
        what = {1: f"more than {_MAX_NDIM} dimensions", 2: "unsupported dtype (float32, float64, int64, int32 only)"}
        raise ValueError("gather_ragged: " + "; ".join(f"rank {r}: {what[c]}" for r, c in errs.items()))
    shapes = [tuple(m[1 : 1 + m[0]]) if m[0] > 0 else None for m in metas]
    trailing = {sh[1:] for sh in shapes if sh is not None}
    dtypes = {int(m[-2]) for m, sh in zip(metas, shapes) if sh is not None}
    if len(trailing) > 1 or len(dtypes) > 1:
        raise ValueError(f"gather_ragged: ranks disagree on the block's trailing shape / dtype: {shapes}")
    senders = [r for r, sh in enumerate(shapes) if sh is not None and r != dst]
    ops, bufs = [], {}
    if rank == dst:
        dtype = [torch.float32, torch.float64, torch.int64, torch.int32][dtypes.pop()] if dtypes else torch.float32
        for r in senders:
            bufs[r] = torch.empty(shapes[r], dtype=dtype, device=device)
            ops.append(dist.P2POp(dist.irecv, bufs[r], r, group))
    elif rank in senders:
        ops.append(dist.P2POp(dist.isend, block.contiguous(), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return None
    out = []
    for r, sh in enumerate(shapes):
        if sh is None:
            continue
        out.append(block if r == dst else bufs[r])
    return out


def broadcast_object(obj, src: int = 0):
    """``obj`` of rank ``src`` on every rank (small Python objects: index lists)."""
    rank, world = rank_world()
    if not (dist.is_available() and dist.is_initialized()):
        return obj
    box = [obj if rank == src else None]
    dist.broadcast_object_list(box, src=src, device=local_device() if dist.get_backend() == "nccl" else None)
    return box[0]


def barrier():
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":  # name the device: without it RCCL guesses from the rank and warns (or picks a GPU another rank owns)
            dist.barrier(device_ids=[local_device().index])
        else:
            dist.barrier()
