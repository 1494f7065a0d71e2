"""Integer encodings and the walker-batch container used by the sampling path.

Mirrors ``/root/reference/src/jamun/utils/residue_metadata.py:7-32,62-83`` (encodings are part of the
checkpoint contract: embedding rows are indexed by them) and the collation rules of
``/root/reference/src/jamun/utils/data_with_residue_info.py:17-33`` + ``cmdline/sample.py:27-38``.
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass, field, replace
from typing import List, Optional, Sequence

import torch

_topology_ids = itertools.count(1)

ATOM_TYPES = ["C", "O", "N", "F", "S"]
ATOM_CODES = ["C", "O", "N", "S", "CA", "CB"]
RESIDUE_CODES = [
    "ALA", "ARG", "ASN", "ASP", "CYS", "GLU", "GLN", "GLY", "HIS", "ILE", "LEU",
    "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL", "ACE", "NME",
]  # fmt: skip


def _encode(x: str, table: List[str]) -> int:
    return table.index(x) if x in table else len(table)


def encode_atom_type(symbol: str) -> int:
    """Element symbol -> int; unknown -> ``len(ATOM_TYPES)`` (residue_metadata.py:62-67)."""
    return _encode(symbol, ATOM_TYPES)


def encode_atom_code(name: str) -> int:
    """PDB atom name -> int; unknown -> ``len(ATOM_CODES)`` (residue_metadata.py:70-75)."""
    return _encode(name, ATOM_CODES)


def encode_residue(name: str) -> int:
    """Residue name -> int; unknown -> ``len(RESIDUE_CODES)`` (residue_metadata.py:78-83)."""
    return _encode(name, RESIDUE_CODES)


@dataclass
class WalkerBatch:
    """A batch of independent walkers (graphs): the layout the reference gets from ``Batch.from_data_list``.

    ``pos [N,3] f32``; per-atom int32 index tensors ``[N]``; ``bonds [2,B] i64`` already offset by the
    cumulative atom count; ``batch [N] i64``; ``ptr [W+1] i64``.  Walkers of the same molecule are
    consecutive (``cmdline/sample.py:35-37``).
    """

    pos: torch.Tensor
    atom_type_index: torch.Tensor
    atom_code_index: torch.Tensor
    residue_code_index: torch.Tensor
    residue_sequence_index: torch.Tensor
    bonds: torch.Tensor
    batch: torch.Tensor
    ptr: torch.Tensor
    dataset_label: List[str] = field(default_factory=list)
    extras: Optional[List[dict]] = None  # per-walker passthrough (atom names etc.)
    topology_id: int = field(default_factory=lambda: next(_topology_ids))  # identifies the static part (everything but pos)

    def with_pos(self, pos: torch.Tensor) -> "WalkerBatch":
        """Same walkers, new coordinates (``ModelSamplingWrapper.positions_to_graph``, sampling_wrapper.py:36-47)."""
        assert pos.shape == self.pos.shape, "The number of positions and nodes should be the same"
        return replace(self, pos=pos)

    def clone(self) -> "WalkerBatch":
        return replace(self, pos=self.pos.clone())

    @property
    def num_graphs(self) -> int:
        return int(self.ptr.numel() - 1)

    @property
    def num_nodes(self) -> int:
        return int(self.pos.shape[0])

    @staticmethod
    def from_molecules(mols: Sequence[dict], labels: Optional[Sequence[str]] = None) -> "WalkerBatch":
        keys = ("pos", "atom_type_index", "atom_code_index", "residue_code_index", "residue_sequence_index")
        cat = {k: torch.cat([m[k] for m in mols], dim=0) for k in keys}
        ptr = [0]
        bonds, batch = [], []
        for gi, m in enumerate(mols):
            n = m["pos"].shape[0]
            bonds.append(m["bonds"].to(torch.long).reshape(2, -1) + ptr[-1])
            batch.append(torch.full((n,), gi, dtype=torch.long))
            ptr.append(ptr[-1] + n)
        return WalkerBatch(
            pos=cat["pos"].to(torch.float32),
            atom_type_index=cat["atom_type_index"].to(torch.int32),
            atom_code_index=cat["atom_code_index"].to(torch.int32),
            residue_code_index=cat["residue_code_index"].to(torch.int32),
            residue_sequence_index=cat["residue_sequence_index"].to(torch.int32),
            bonds=torch.cat(bonds, dim=1),
            batch=torch.cat(batch),
            ptr=torch.tensor(ptr, dtype=torch.long),
            dataset_label=list(labels) if labels is not None else [m.get("dataset_label", "mol") for m in mols],
            extras=[{k: v for k, v in m.items() if not torch.is_tensor(v)} for m in mols],
        )

    def slice_graphs(self, lo: int, hi: int) -> "WalkerBatch":
        """Contiguous block of walkers ``[lo, hi)`` (the multi-GPU shard)."""
        a, b = int(self.ptr[lo]), int(self.ptr[hi])
        bm = (self.bonds[0] >= a) & (self.bonds[0] < b)
        return WalkerBatch(
            pos=self.pos[a:b],
            atom_type_index=self.atom_type_index[a:b],
            atom_code_index=self.atom_code_index[a:b],
            residue_code_index=self.residue_code_index[a:b],
            residue_sequence_index=self.residue_sequence_index[a:b],
            bonds=self.bonds[:, bm] - a,
            batch=self.batch[a:b] - lo,
            ptr=self.ptr[lo : hi + 1] - a,
            dataset_label=self.dataset_label[lo:hi],
            extras=self.extras[lo:hi] if self.extras is not None else None,
        )

    def to(self, device) -> "WalkerBatch":
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if torch.is_tensor(v) else v
        return WalkerBatch(**kw)  # topology_id is preserved: same static graph on another device

    def as_topology(self) -> dict:
        """Plain-dict view (what ``oracle.graph.collate`` returns) for tests."""
        return dict(
            pos=self.pos, atom_type_index=self.atom_type_index, atom_code_index=self.atom_code_index,
            residue_code_index=self.residue_code_index, residue_sequence_index=self.residue_sequence_index,
            bonds=self.bonds, batch=self.batch, ptr=self.ptr, num_graphs=self.num_graphs,
        )  # fmt: skip
