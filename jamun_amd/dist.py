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


def init_process_group(backend: Optional[str] = None, force: bool = False) -> Tuple[int, int]:
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  No-op for 1 rank unless ``force``
    (a one-rank group exercises the backend's code path — RCCL on a single GPU — with nobody to talk to)."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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


def _gather_plan(block: Optional[torch.Tensor], dst: int, group, device):
    """The metadata exchange shared by both gathers: every rank learns every rank's block shape (one small all-gather of int64
    metadata) so that all ranks agree on the trailing shape, the dtype and on who sends.  Local problems (too many dimensions,
    unsupported dtype, trailing shapes that disagree) raise on EVERY rank, before any payload moves, so no rank is left waiting
    in a collective.  Returns (block on `device`, shapes per rank (None = nothing to send), dtype, senders, device)."""
    rank, world = rank_world()
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
    # metadata: [ndim, shape..., dtype code, error code].  Errors are REPORTED through the all-gather and raised on every rank afterwards
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
    dtype = [torch.float32, torch.float64, torch.int64, torch.int32][dtypes.pop()] if dtypes else torch.float32
    senders = [r for r, sh in enumerate(shapes) if sh is not None and r != dst]
    return block, shapes, dtype, senders, device


def gather_ragged(block: Optional[torch.Tensor], dst: int = 0, group=None, device=None) -> Optional[List[torch.Tensor]]:
    """Gather per-rank blocks ``[n_rank, ...]`` with different leading sizes to ``dst``.

    A rank with nothing to contribute passes ``None`` (or a block with zero rows).  After the metadata exchange (`_gather_plan`) each
    non-empty rank sends exactly its rows to ``dst`` (batched point-to-point: no padding, nothing lands on the other ranks —
    an all-gather would put every rank's block on every rank).  Returns the list of non-empty blocks in rank order on
    ``dst`` and ``None`` elsewhere.  All sender blocks are resident on ``dst``'s device at once: for trajectories use
    `gather_ragged_to_host`, which receives them one at a time."""
    rank, world = rank_world()
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return [block] if block is not None and block.shape[0] > 0 else []
    # (a one-rank group that IS initialised — torchrun --nproc-per-node 1 — runs the same metadata exchange as N ranks)
    block, shapes, dtype, senders, device = _gather_plan(block, dst, group, device)
    ops, bufs = [], {}
    if rank == dst:
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


class HostStager:
    """Staging memory of the trajectory gather, allocated once and reused for every block of every batch: ONE page-locked host
    buffer (device -> host copies at full PCIe rate) and, on ``dst`` under RCCL, ONE device receive buffer.  Both grow to the
    largest block seen and are never handed out: `to_numpy` returns a pageable COPY, so nothing page-locked is retained by the
    caller (a 20 000-step batch is ~1 GB per block; PyTorch's pinned allocator rounds each allocation up to a power of two and
    never returns it — eight fresh blocks per batch were ~80 GB of page-locked memory on rank 0 of an 8-GPU run)."""

    def __init__(self):
        self._pinned: Optional[torch.Tensor] = None
        self._recv: Optional[torch.Tensor] = None
        self.d2h_s = 0.0
        self.copy_s = 0.0

    def recv_buffer(self, shape, dtype, device) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        if self._recv is None or self._recv.numel() < nbytes or self._recv.device != device:
            self._recv = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._recv[:nbytes].view(dtype).view(shape)

    def to_numpy(self, t: torch.Tensor):
        import time

        import numpy as np

        t = t.detach()
        if not t.is_cuda:
            return np.array(t.contiguous().numpy(), copy=True)
        nbytes = t.numel() * t.element_size()
        if self._pinned is None or self._pinned.numel() < nbytes:
            self._pinned = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        stage = self._pinned[:nbytes].view(t.dtype).view(t.shape)
        t0 = time.perf_counter()
        stage.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        t1 = time.perf_counter()
        out = np.array(stage.numpy(), copy=True)  # pageable; the staging buffer is free for the next block
        self.d2h_s += t1 - t0
        self.copy_s += time.perf_counter() - t1
        return out


def gather_ragged_to_host(block: Optional[torch.Tensor], dst: int = 0, group=None, device=None, stager: Optional[HostStager] = None,
                          timings: Optional[dict] = None):
    """`gather_ragged` for trajectories: the blocks end up as pageable numpy arrays on ``dst`` and are received ONE AT A TIME —
    ``dst`` posts one receive into the stager's reusable device buffer, copies it to the host, then posts the next; a sender
    blocks in its send until its turn (the senders hold their own block anyway).  Peak extra device memory on ``dst`` = one
    block, not world - 1 blocks (7 x 2 GB at the 4AA shape with 20 000-step batches).  ``timings`` (optional dict) accumulates
    ``gather_s`` (wall time of the exchange on this rank, metadata + payload + host copies) and ``gather_bytes`` (payload bytes
    that crossed ranks, on ``dst``).  Returns the list of arrays in rank order on ``dst``, ``None`` elsewhere."""
    import time

    t0 = time.perf_counter()
    stager = stager if stager is not None else HostStager()
    rank, world = rank_world()
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        out = [stager.to_numpy(block)] if block is not None and block.shape[0] > 0 else []
        if timings is not None:
            timings["gather_s"] = timings.get("gather_s", 0.0) + time.perf_counter() - t0
            timings.setdefault("gather_bytes", 0)
        return out
    block, shapes, dtype, senders, device = _gather_plan(block, dst, group, device)
    out, moved = None, 0
    if rank == dst:
        out = []
        for r, sh in enumerate(shapes):
            if sh is None:
                continue
            if r == dst:
                out.append(stager.to_numpy(block))
                continue
            buf = stager.recv_buffer(sh, dtype, device)
            dist.recv(buf, src=r, group=group)
            moved += buf.numel() * buf.element_size()
            out.append(stager.to_numpy(buf))
    elif rank in senders:
        dist.send(block.contiguous(), dst=dst, group=group)
    if timings is not None:
        timings["gather_s"] = timings.get("gather_s", 0.0) + time.perf_counter() - t0
        timings["gather_bytes"] = timings.get("gather_bytes", 0) + moved
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
