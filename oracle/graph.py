"""Graph-side restatements: radius graph, scatter-mean, mean-centring, batch collation.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED for the
third-party ops (torch_cluster 1.6.3 / torch_geometric 2.6.1 / torch_scatter 2.1.2,
not vendored, not installed); the reference holds no fixtures for them.
"""

from __future__ import annotations

from typing import Dict, List, Sequence

import torch

MAX_NUM_NEIGHBORS = 32  # torch_geometric.nn.radius_graph default, not overridden at src/jamun/model/denoiser.py:149


def radius_graph(pos: torch.Tensor, r: float, batch: torch.Tensor, max_num_neighbors: int = MAX_NUM_NEIGHBORS) -> torch.Tensor:
    """``torch_geometric.nn.radius_graph(pos, r, batch)`` with torch_cluster's *CUDA* semantics.

    Call site: ``src/jamun/model/denoiser.py:149``.  ``radius_graph(loop=False)`` calls
    ``radius(x, x, r, batch, batch, max_num_neighbors + 1)`` and drops self pairs; the CUDA kernel
    scans each graph's atoms in index order and keeps a pair while ``d^2 < r^2`` until
    ``max_num_neighbors + 1`` are found.  Returned ``edge_index = [src = neighbour, dst = centre]``
    (``flow="source_to_target"``), ordered by centre then by neighbour index.
    (The CPU build of torch_cluster uses a kd-tree and keeps the first 33 matches in *tree* order; the
    two agree whenever no atom has more than 32 neighbours inside the cutoff.)
    """
    N = pos.shape[0]
    r2 = torch.as_tensor(r, dtype=pos.dtype) ** 2
    src: List[int] = []
    dst: List[int] = []
    batch_l = batch.tolist()
    # graph boundaries
    starts: Dict[int, int] = {}
    ends: Dict[int, int] = {}
    for i, b in enumerate(batch_l):
        starts.setdefault(b, i)
        ends[b] = i + 1
    for i in range(N):
        b = batch_l[i]
        lo, hi = starts[b], ends[b]
        d2 = ((pos[lo:hi] - pos[i]) ** 2).sum(-1)
        hit = (d2 < r2).nonzero().flatten().tolist()
        hit = hit[: max_num_neighbors + 1]
        for j in hit:
            if lo + j != i:
                src.append(lo + j)
                dst.append(i)
    return torch.tensor([src, dst], dtype=torch.long).reshape(2, -1)


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """``torch_scatter.scatter(src, index, dim=0, dim_size=dim_size, reduce="mean")``: sum / count.clamp(min=1).

    Call sites: ``src/jamun/e3tools/nn/_conv.py:117``, ``src/jamun/utils/mean_center.py:10``.
    """
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    out.index_add_(0, index, src)
    count = torch.zeros(dim_size, dtype=src.dtype)
    count.index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype))
    count = count.clamp(min=1)
    return out / count.reshape((-1,) + (1,) * (src.ndim - 1))


def mean_center(pos: torch.Tensor, batch: torch.Tensor, num_graphs: int) -> torch.Tensor:
    """``src/jamun/utils/mean_center.py:7-12``."""
    mean_pos = scatter_mean(pos, batch, num_graphs)
    return pos - mean_pos[batch]


def collate(graphs: Sequence[dict]) -> dict:
    """``Batch.from_data_list`` + ``DataWithResidueInformation.__inc__`` (``src/jamun/utils/data_with_residue_info.py:17-33``).

    Each graph is a dict with ``pos [n,3]``, the four per-atom int32 index tensors and ``bonds [2,B]`` (int64).
    Per-atom tensors are concatenated, ``bonds`` is offset by the cumulative atom count, ``batch``/``ptr`` are built.
    """
    out = {}
    for k in ("pos", "atom_type_index", "atom_code_index", "residue_code_index", "residue_sequence_index"):
        out[k] = torch.cat([g[k] for g in graphs], dim=0)
    ptr = [0]
    bonds = []
    batch = []
    for gi, g in enumerate(graphs):
        n = g["pos"].shape[0]
        bonds.append(g["bonds"].to(torch.long) + ptr[-1])
        batch.append(torch.full((n,), gi, dtype=torch.long))
        ptr.append(ptr[-1] + n)
    out["bonds"] = torch.cat(bonds, dim=1) if bonds else torch.zeros(2, 0, dtype=torch.long)
    out["batch"] = torch.cat(batch)
    out["ptr"] = torch.tensor(ptr, dtype=torch.long)
    out["num_graphs"] = len(graphs)
    return out
