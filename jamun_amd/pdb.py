"""PDB -> walker graph without mdtraj.

Restates what the sampling path needs of ``/root/reference/src/jamun/data/_mdtraj.py:56-89,169-256`` and
``data/_utils.py:217-237``: select "protein and not type H", encode atom type / atom name / residue name
(``utils/residue_metadata.py:62-83``), residue sequence index, and the bonded edges (each bond once, lower atom index
first, as mdtraj's ``Topology.add_bond`` orders them): standard-residue templates + peptide bonds + CONECT records
between selected atoms.  Coordinates are converted from Angstrom to nanometres (mdtraj convention).
"""

from __future__ import annotations

import os
import re
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .data import encode_atom_code, encode_atom_type, encode_residue

_BACKBONE = [("N", "CA"), ("CA", "C"), ("C", "O"), ("C", "OXT"), ("CA", "CB")]
_SIDECHAIN: Dict[str, List[Tuple[str, str]]] = {
    "ALA": [], "GLY": [],
    "ARG": [("CB", "CG"), ("CG", "CD"), ("CD", "NE"), ("NE", "CZ"), ("CZ", "NH1"), ("CZ", "NH2")],
    "ASN": [("CB", "CG"), ("CG", "OD1"), ("CG", "ND2")],
    "ASP": [("CB", "CG"), ("CG", "OD1"), ("CG", "OD2")],
    "CYS": [("CB", "SG")],
    "GLN": [("CB", "CG"), ("CG", "CD"), ("CD", "OE1"), ("CD", "NE2")],
    "GLU": [("CB", "CG"), ("CG", "CD"), ("CD", "OE1"), ("CD", "OE2")],
    "HIS": [("CB", "CG"), ("CG", "ND1"), ("CG", "CD2"), ("ND1", "CE1"), ("CD2", "NE2"), ("CE1", "NE2")],
    "ILE": [("CB", "CG1"), ("CB", "CG2"), ("CG1", "CD1")],
    "LEU": [("CB", "CG"), ("CG", "CD1"), ("CG", "CD2")],
    "LYS": [("CB", "CG"), ("CG", "CD"), ("CD", "CE"), ("CE", "NZ")],
    "MET": [("CB", "CG"), ("CG", "SD"), ("SD", "CE")],
    "PHE": [("CB", "CG"), ("CG", "CD1"), ("CG", "CD2"), ("CD1", "CE1"), ("CD2", "CE2"), ("CE1", "CZ"), ("CE2", "CZ")],
    "PRO": [("CB", "CG"), ("CG", "CD"), ("CD", "N")],
    "SER": [("CB", "OG")],
    "THR": [("CB", "OG1"), ("CB", "CG2")],
    "TRP": [("CB", "CG"), ("CG", "CD1"), ("CG", "CD2"), ("CD1", "NE1"), ("NE1", "CE2"), ("CD2", "CE2"), ("CD2", "CE3"),
            ("CE2", "CZ2"), ("CE3", "CZ3"), ("CZ2", "CH2"), ("CZ3", "CH2")],
    "TYR": [("CB", "CG"), ("CG", "CD1"), ("CG", "CD2"), ("CD1", "CE1"), ("CD2", "CE2"), ("CE1", "CZ"), ("CE2", "CZ"), ("CZ", "OH")],
    "VAL": [("CB", "CG1"), ("CB", "CG2")],
}  # fmt: skip
_ALIASES = {"HID": "HIS", "HIE": "HIS", "HIP": "HIS", "HSD": "HIS", "HSE": "HIS", "CYX": "CYS", "CYM": "CYS", "ASH": "ASP", "GLH": "GLU", "LYN": "LYS"}
_CAPS = {"ACE": [("CH3", "C"), ("C", "O")], "NME": [("N", "CH3"), ("N", "C")], "NMA": [("N", "CH3")], "NH2": []}
PROTEIN_RESIDUES = set(_SIDECHAIN) | set(_ALIASES) | set(_CAPS)


def _element(line: str, name: str) -> str:
    el = line[76:78].strip() if len(line) >= 78 else ""
    if el:
        return el.capitalize()
    n = name.lstrip("0123456789")
    return n[:1].upper() if n else "X"


def _chain_indices(chain_ids: List[str]) -> List[int]:
    """0-based index of each atom's chain, chains numbered in order of first appearance (mdtraj ``chain.index``)."""
    order: Dict[str, int] = {}
    return [order.setdefault(c, len(order)) for c in chain_ids]


def read_pdb(path: str) -> dict:
    """First model of a PDB file -> molecule dict (heavy protein atoms only)."""
    atoms = []  # (serial, name, resname, chain, resseq+icode, x, y, z, element)
    kept: List[int] = []  # index of each kept atom among ALL atom records of the model (trajectory arrays carry all atoms)
    n_records = 0
    conect: List[Tuple[int, int]] = []
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec in ("ATOM  ", "HETATM"):
                n_records += 1
                name = line[12:16].strip()
                resname = line[17:20].strip()
                if resname not in PROTEIN_RESIDUES:
                    continue
                el = _element(line, name)
                if el == "H" or el == "D":
                    continue
                atoms.append((int(line[6:11]), name, resname, line[21], line[22:27], float(line[30:38]), float(line[38:46]), float(line[46:54]), el))
                kept.append(n_records - 1)
            elif rec == "CONECT":
                f0 = line[6:].split()
                if len(f0) >= 2:
                    a0 = int(f0[0])
                    conect += [(a0, int(b)) for b in f0[1:]]
            elif rec == "ENDMDL":
                break
    if not atoms:
        raise ValueError(f"{path}: no protein heavy atoms found")
    # residues in file order
    res_keys: List[Tuple[str, str]] = []
    res_of_atom = []
    for a in atoms:
        key = (a[3], a[4])
        if not res_keys or res_keys[-1] != key:
            res_keys.append(key)
        res_of_atom.append(len(res_keys) - 1)
    n = len(atoms)
    by_res: Dict[int, Dict[str, int]] = {}
    for i, a in enumerate(atoms):
        by_res.setdefault(res_of_atom[i], {})[a[1]] = i
    bonds: List[Tuple[int, int]] = []
    seen = set()

    def add(i: Optional[int], j: Optional[int]):
        if i is None or j is None or i == j:
            return
        b = (min(i, j), max(i, j))  # mdtraj Topology.add_bond: lower index first
        if b not in seen:
            seen.add(b)
            bonds.append(b)

    for r in range(len(res_keys)):
        names = by_res[r]
        resname = atoms[next(iter(names.values()))][2]
        if r > 0 and res_keys[r][0] == res_keys[r - 1][0]:
            add(by_res[r - 1].get("C"), names.get("N"))  # peptide bond
        base = _ALIASES.get(resname, resname)
        templ = _CAPS[base] if base in _CAPS else _BACKBONE + _SIDECHAIN[base]
        for a, b in templ:
            add(names.get(a), names.get(b))
    serial_to_idx = {a[0]: i for i, a in enumerate(atoms)}
    for s0, s1 in conect:
        add(serial_to_idx.get(s0), serial_to_idx.get(s1))
    pos = torch.tensor([[a[5], a[6], a[7]] for a in atoms], dtype=torch.float32) / 10.0
    return dict(
        pos=pos,
        atom_type_index=torch.tensor([encode_atom_type(a[8]) for a in atoms], dtype=torch.int32),
        atom_code_index=torch.tensor([encode_atom_code(a[1]) for a in atoms], dtype=torch.int32),
        residue_code_index=torch.tensor([encode_residue(a[2]) for a in atoms], dtype=torch.int32),
        residue_sequence_index=torch.tensor(res_of_atom, dtype=torch.int32),
        bonds=torch.tensor(bonds, dtype=torch.long).reshape(-1, 2).T.contiguous(),
        atom_names=[a[1] for a in atoms],
        residues=[a[2] for a in atoms],
        elements=[a[8] for a in atoms],
        residue_ids=[int(a[4][:4]) for a in atoms],
        chain_index=_chain_indices([a[3] for a in atoms]),
        all_atom_index=torch.tensor(kept, dtype=torch.long),
        n_all_atoms=n_records,
    )


def save_pdb(path: str, mol: dict, frames: torch.Tensor) -> None:
    """Multi-model PDB in Angstrom, record for record what ``/root/reference/src/jamun/utils/mdtraj.py:26-60`` writes:
    ``MODEL <frame index from 0>``, ATOM records (serial, name, residue name, chain index as one digit, residue index + 1,
    coordinates x 10, element), a TER record, one CONECT record per atom listing its bonded partners, ``ENDMDL``; ``END``.
    ``frames`` is ``[T, n, 3]`` in nanometres; ``mol`` is a molecule dict of ``read_pdb``."""
    frames = frames.detach().cpu().float() if torch.is_tensor(frames) else torch.as_tensor(frames, dtype=torch.float32)
    if frames.ndim == 2:
        frames = frames[None]
    n = frames.shape[1]
    names, resn, els = mol["atom_names"], mol["residues"], mol["elements"]
    res_idx = [int(i) for i in mol["residue_sequence_index"]]
    chain = [int(c) for c in mol.get("chain_index", [0] * n)]
    partners = [[i + 1] for i in range(n)]
    seen = set()
    for a, b in mol["bonds"].T.tolist():
        if (a, b) not in seen:  # the reference de-duplicates (atom1, atom2) pairs
            seen.add((a, b))
            partners[a].append(b + 1)
            partners[b].append(a + 1)
    conect = "".join("CONECT" + "".join(f"{q:5d}" for q in ps) + "\n" for ps in partners)
    with open(path, "w") as f:
        for t in range(frames.shape[0]):
            f.write(f"MODEL        {t}\n")
            xyz = (frames[t] * 10).tolist()
            for i in range(n):
                x, y, z = xyz[i]
                f.write(f"ATOM  {i + 1:5d} {names[i]:<4s} {resn[i]:3s} {chain[i]:1d}{res_idx[i] + 1:4d}    {x:8.3f}{y:8.3f}{z:8.3f}  1.00  0.00          {els[i]:>2s}\n")
            f.write(f"TER   {n + 1:5d}      {resn[n - 1]:3s} {chain[n - 1]:1d}{res_idx[n - 1] + 1:4d}\n")
            f.write(conect)
            f.write("ENDMDL\n")
        f.write("END\n")


write_pdb = save_pdb  # older name


def save_dcd(path: str, frames) -> None:
    """CHARMM-format DCD (what mdtraj's ``Trajectory.save_dcd`` produces for a trajectory without a unit cell, called at
    ``/root/reference/src/jamun/metrics/_save_trajectory.py:56,92,96``): little-endian Fortran records — an 84-byte ``CORD``
    header (NSET frames, ISTART 0, NSAVC 1, DELTA 1.0 as float32, no unit cell, CHARMM version 24), a two-line title block,
    the atom count, then per frame three float32 records X, Y, Z in Angstrom.  ``frames`` is ``[T, n, 3]`` in nanometres."""
    import struct

    import numpy as np

    xyz = np.asarray(frames.detach().cpu() if torch.is_tensor(frames) else frames, dtype=np.float32)
    if xyz.ndim != 3 or xyz.shape[2] != 3:
        raise ValueError(f"expected [frames, atoms, 3], got {xyz.shape}")
    nset, natom = xyz.shape[0], xyz.shape[1]
    icntrl = [0] * 20
    icntrl[0], icntrl[1], icntrl[2], icntrl[3] = nset, 0, 1, nset  # NSET, ISTART, NSAVC, NSTEP
    icntrl[19] = 24
    head = b"CORD" + struct.pack("<9i", *icntrl[:9]) + struct.pack("<f", 1.0) + struct.pack("<10i", *icntrl[10:])
    assert len(head) == 84
    titles = [b"Created by jamun_amd (DCD, CHARMM format)".ljust(80), b"REMARKS coordinates in Angstrom".ljust(80)]
    with open(path, "wb") as f:
        f.write(struct.pack("<i", 84) + head + struct.pack("<i", 84))
        tb = struct.pack("<i", len(titles)) + b"".join(titles)
        f.write(struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(tb)))
        f.write(struct.pack("<3i", 4, natom, 4))
        # one buffer for all frames: per frame three Fortran records (marker, natom float32, marker) — a structured view instead of 3 T small writes
        ang = np.transpose(xyz * np.float32(10.0), (0, 2, 1))  # [T, 3, n]
        recs = np.empty((nset, 3), dtype=np.dtype([("a", "<i4"), ("v", "<f4", (natom,)), ("b", "<i4")]))
        recs["a"] = recs["b"] = 4 * natom
        recs["v"] = ang
        f.write(recs.tobytes())


class PDBDataset:
    """Single-structure dataset: what ``MDtrajDataset(root, trajfiles=[pdb], pdbfile=pdb, label)`` provides the sampler."""

    def __init__(self, root: str, pdbfile: str, label: str):
        self.root, self.pdbfile, self._label = root, pdbfile, label
        self.molecule = read_pdb(os.path.join(root, pdbfile))
        self.molecule["dataset_label"] = label

    def label(self) -> str:
        return self._label

    def __len__(self) -> int:
        return 1

    def __getitem__(self, idx: int) -> dict:
        if idx not in (0, -1):
            raise IndexError(idx)
        return self.molecule


class MDtrajDataset:
    """Frames of one molecule (``data/_mdtraj.py:155-237`` without mdtraj): topology from ``pdbfile``, coordinates from
    ``trajfiles`` — the PDB itself (one frame) or Timewarp-style ``*.npz`` / ``*.npy`` arrays whose ``positions`` are
    ``[frames, all atoms, 3]`` in nanometres (``_mdtraj.py:194-200``).  Frames ``start_frame : start_frame + num_frames :
    subsample``; atoms = the reference's selection "protein and not type H"; item = molecule dict with that frame."""

    def __init__(self, root: str, trajfiles: Sequence[str], pdbfile: str, label: str, num_frames: Optional[int] = None,
                 start_frame: Optional[int] = None, transform=None, subsample: Optional[int] = None, loss_weight: float = 1.0,
                 verbose: bool = False):
        import numpy as np

        self.root, self._label, self.transform, self.loss_weight = root, label, transform, loss_weight
        self.molecule = read_pdb(os.path.join(root, pdbfile))
        self.molecule["dataset_label"] = label
        keep = self.molecule["all_atom_index"]
        files = [os.path.join(root, f) for f in trajfiles]
        if files[0].endswith((".npz", ".npy")):
            chunks = []
            for f in files:
                arr = np.load(f)
                arr = arr["positions"] if hasattr(arr, "files") else arr
                if arr.ndim != 3 or arr.shape[1] != self.molecule["n_all_atoms"] or arr.shape[2] != 3:
                    raise ValueError(f"{f}: positions {arr.shape} do not match the {self.molecule['n_all_atoms']} atoms of {pdbfile}")
                chunks.append(np.asarray(arr, dtype=np.float32))
            xyz = torch.from_numpy(np.vstack(chunks))[:, keep]
        elif all(f.endswith(".pdb") for f in files):
            xyz = torch.stack([read_pdb(f)["pos"] for f in files])
        else:
            raise NotImplementedError(f"trajectory format of {files[0]} (supported: .pdb, .npz, .npy)")
        start = 0 if start_frame is None else start_frame
        n = xyz.shape[0] - start if num_frames in (-1, None) else num_frames
        step = 1 if not subsample else subsample
        self.xyz = xyz[start : start + n : step].contiguous()
        self.molecule["pos"] = self.xyz[0]

    def label(self) -> str:
        return self._label

    def __len__(self) -> int:
        return self.xyz.shape[0]

    def __getitem__(self, idx: int) -> dict:
        mol = dict(self.molecule)
        mol["pos"] = self.xyz[idx]
        return self.transform(mol) if self.transform else mol


def create_dataset_from_pdbs(pdbfiles: Sequence[str], label_prefix: Optional[str] = None) -> List[PDBDataset]:
    """``jamun.data.create_dataset_from_pdbs`` (``data/_utils.py:217-237``): one dataset per file, label = file stem."""
    if isinstance(pdbfiles, str):
        pdbfiles = [pdbfiles]
    out = []
    for p in pdbfiles:
        p = os.path.abspath(p)
        label = os.path.basename(p).split(".")[0]
        if label_prefix is not None:
            label = f"{label_prefix}{label}"
        out.append(PDBDataset(os.path.dirname(p), os.path.basename(p), label))
    return out


def parse_datasets_from_directory(root: str, traj_pattern: str, pdb_pattern: Optional[str] = None,
                                  pdb_file: Optional[Sequence[str]] = None, max_datasets: Optional[int] = None,
                                  max_datasets_offset: Optional[int] = None, filter_codes: Optional[Sequence[str]] = None,
                                  as_iterable: bool = False, **dataset_kwargs) -> List[MDtrajDataset]:
    """``jamun.data.parse_datasets_from_directory`` (``/root/reference/src/jamun/data/_utils.py:36-116``) — the
    ``init_datasets._target_`` of the reference's ``sample_uncapped_2AA / 4AA / mdgen`` experiment files: one
    ``MDtrajDataset`` per molecule code found under ``root``.

    ``traj_pattern`` / ``pdb_pattern`` are ``<sub-directory>/<regular expression>``; the expression is matched (``re.match``:
    anchored at the start) against the ENTRY NAMES of that directory and its group 1 is the code.  All trajectory files of a
    code belong to its dataset; ``pdb_pattern`` picks the topology per code (entries whose code has no trajectory are
    skipped), ``pdb_file`` is one topology shared by every code.  Then, in the reference's order: ``filter_codes``, sort,
    ``max_datasets_offset``, ``max_datasets``.  ``dataset_kwargs`` (``subsample``, ``num_frames``, ``start_frame`` ...) go to
    every dataset; label = code.  A code without a topology raises ``KeyError`` as the reference's dict lookup does.
    ``as_iterable`` (the reference's streaming ``MDtrajIterableDataset``, a training feature) is out of scope: rejected."""
    if pdb_file is not None and pdb_pattern is not None:
        raise ValueError("Exactly one of pdb_file and pdb_pattern should be provided.")
    if as_iterable:
        raise NotImplementedError("as_iterable=True (MDtrajIterableDataset, a streaming training dataset) is out of scope")
    traj_prefix, traj_name = os.path.split(traj_pattern)
    if "*" in traj_prefix or "?" in traj_prefix:
        raise ValueError("traj_prefix should not contain wildcards.")
    traj_re = re.compile(traj_name)
    traj_files: Dict[str, List[str]] = {}
    for entry in os.scandir(os.path.join(root, traj_prefix)):
        m = traj_re.match(entry.name)
        if m:
            traj_files.setdefault(m.group(1), []).append(os.path.join(traj_prefix, entry.name))
    if not traj_files:
        raise ValueError("No codes found in directory.")
    pdb_files: Dict[str, object] = {}
    if pdb_pattern is not None:
        pdb_prefix, pdb_name = os.path.split(pdb_pattern)
        if "*" in pdb_prefix or "?" in pdb_prefix:
            raise ValueError("pdb_prefix should not contain wildcards.")
        pdb_re = re.compile(pdb_name)
        for entry in os.scandir(os.path.join(root, pdb_prefix)):
            m = pdb_re.match(entry.name)
            if m and m.group(1) in traj_files:
                pdb_files[m.group(1)] = os.path.join(pdb_prefix, entry.name)
    else:
        pdb_files = {code: pdb_file for code in traj_files}
    codes = list(traj_files)
    if filter_codes is not None:
        wanted = set(filter_codes)
        codes = [c for c in codes if c in wanted]
    codes = sorted(codes)
    if max_datasets_offset is not None:
        codes = codes[max_datasets_offset:]
    if max_datasets is not None:
        codes = codes[:max_datasets]
    # os.scandir order is arbitrary; the reference keeps it (mdtraj concatenates the files in that order) — sorted here so that
    # a code's frames do not depend on the file system
    return [MDtrajDataset(root, trajfiles=sorted(traj_files[c]), pdbfile=pdb_files[c], label=c, **dataset_kwargs) for c in codes]

