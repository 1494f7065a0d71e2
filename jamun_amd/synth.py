"""Synthetic molecules and a seeded synthetic checkpoint (SURVEY.md §8 d).

No real datasets or published checkpoints are reachable offline, so benchmarks
and parity tests run on (a) self-avoiding random chains with the reference's
per-atom integer encodings and (b) a randomly initialised checkpoint that uses
the reference's state-dict names and tensor shapes
(``/root/reference/src/jamun/model/arch/e3conv.py:15-85``), so that the same
loader path serves real Lightning checkpoints.
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch


def default_arch(**over) -> dict:
    """Hyper-parameters of ``src/jamun/hydra_config/model/arch/e3conv.yaml:3-14``."""
    a = dict(
        irreps_out="1x1e",
        irreps_hidden="120x0e + 32x1e",
        irreps_sh="1x0e + 1x1e",
        n_layers=5,
        edge_attr_dim=64,
        atom_type_embedding_dim=8,
        atom_code_embedding_dim=8,
        residue_code_embedding_dim=32,
        residue_index_embedding_dim=8,
        use_residue_information=True,
        use_residue_sequence_index=False,
    )
    a.update(over)
    return a


def random_chain(n_atoms: int, seed: int = 0, bond: float = 0.15, min_dist: float = 0.24) -> dict:
    """Self-avoiding random chain: bonds (i-1, i), with every 4th atom branching off i-2.

    Returns a molecule dict: ``pos [n,3] f32`` (nm), the four per-atom int32 index tensors, ``bonds [2,B] i64``
    (each bond once, ``parent -> child``, as ``src/jamun/data/_mdtraj.py:73``).
    """
    rng = np.random.RandomState(seed)
    pos = np.zeros((n_atoms, 3), dtype=np.float64)
    parents = [-1]
    for i in range(1, n_atoms):
        p = i - 2 if (i % 4 == 3 and i >= 2) else i - 1
        parents.append(p)
        others = [j for j in range(i) if j != p]
        thr = min_dist
        for attempt in range(1, 100001):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            cand = pos[p] + bond * d
            if not others or np.min(np.linalg.norm(pos[others] - cand, axis=1)) >= thr:
                pos[i] = cand
                break
            if attempt % 500 == 0:  # crowded site: relax the exclusion radius a little and keep going
                thr *= 0.95
        else:  # pragma: no cover
            raise RuntimeError("could not place atom")
    bonds = np.array([[parents[i], i] for i in range(1, n_atoms)], dtype=np.int64).T.reshape(2, -1)
    idx = np.arange(n_atoms)
    return dict(
        pos=torch.tensor(pos, dtype=torch.float32),
        atom_type_index=torch.tensor(rng.randint(0, 3, size=n_atoms), dtype=torch.int32),
        atom_code_index=torch.tensor(rng.randint(0, 7, size=n_atoms), dtype=torch.int32),
        residue_code_index=torch.tensor((idx // 5) % 20, dtype=torch.int32),
        residue_sequence_index=torch.tensor(idx // 5, dtype=torch.int32),
        bonds=torch.tensor(bonds, dtype=torch.long),
    )


def ag_dipeptide() -> dict:
    """Hand-built uncapped Ala-Gly heavy-atom graph: N CA CB C O | N CA C O OXT.

    Encodings follow ``src/jamun/utils/residue_metadata.py:7-32,62-83``; coordinates (nm) are an extended
    conformation from ideal bond lengths/angles.
    """
    # atom names, elements, residue
    names = ["N", "CA", "CB", "C", "O", "N", "CA", "C", "O", "OXT"]
    elems = ["N", "C", "C", "C", "O", "N", "C", "C", "O", "O"]
    res = ["ALA"] * 5 + ["GLY"] * 5
    ATOM_TYPES = ["C", "O", "N", "F", "S"]
    ATOM_CODES = ["C", "O", "N", "S", "CA", "CB"]
    RES = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLU", "GLN", "GLY"]
    enc = lambda x, table: table.index(x) if x in table else len(table)
    # extended backbone built in the xy plane with CB / O / OXT off-axis (Angstrom)
    A = np.array(
        [
            [0.000, 0.000, 0.000],  # N
            [1.458, 0.000, 0.000],  # CA
            [1.994, 1.204, -0.773],  # CB
            [2.009, -0.105, 1.419],  # C
            [1.251, -0.332, 2.362],  # O
            [3.326, 0.060, 1.557],  # N
            [4.002, 0.000, 2.848],  # CA
            [5.510, 0.150, 2.690],  # C
            [6.020, 0.310, 1.560],  # O
            [6.200, 0.100, 3.740],  # OXT
        ]
    )
    bonds = [[0, 1], [1, 2], [1, 3], [3, 4], [3, 5], [5, 6], [6, 7], [7, 8], [7, 9]]
    return dict(
        pos=torch.tensor(A / 10.0, dtype=torch.float32),
        atom_type_index=torch.tensor([enc(e, ATOM_TYPES) for e in elems], dtype=torch.int32),
        atom_code_index=torch.tensor([enc(n, ATOM_CODES) for n in names], dtype=torch.int32),
        residue_code_index=torch.tensor([enc(r, RES) for r in res], dtype=torch.int32),
        residue_sequence_index=torch.tensor([0] * 5 + [1] * 5, dtype=torch.int32),
        bonds=torch.tensor(bonds, dtype=torch.long).T.contiguous(),
        atom_names=names,
        residues=res,
    )


AA_3CODES = {"A": "ALA", "R": "ARG", "N": "ASN", "D": "ASP", "C": "CYS", "E": "GLU", "Q": "GLN", "G": "GLY", "H": "HIS", "I": "ILE",
             "L": "LEU", "K": "LYS", "M": "MET", "F": "PHE", "P": "PRO", "S": "SER", "T": "THR", "W": "TRP", "Y": "TYR", "V": "VAL"}  # residue_metadata.py:34-55


def peptide(sequence: str, seed: int = 0, bond: float = 0.15, min_dist: float = 0.25) -> dict:
    """Uncapped peptide of the one-letter ``sequence`` as a heavy-atom molecule dict with REAL topology — atom names, elements,
    residue names, standard-residue bonds + peptide bonds + the C-terminal OXT, i.e. what ``pdb.read_pdb`` returns for a
    Timewarp ``*-traj-state0.pdb`` after the hydrogens are dropped — and SYNTHETIC coordinates: atoms are grown along the bond
    tree (self-avoiding, bond length 0.15 nm) and relaxed so that ring-closing bonds reach bond length and non-bonded pairs keep
    ``min_dist``.  Dipeptides have 9 (GG) ... 29 (WW) atoms, the uncapped-2AA range of SURVEY.md section 8."""
    from .data import encode_atom_code, encode_atom_type, encode_residue
    from .pdb import _BACKBONE, _SIDECHAIN

    names: List[str] = []
    res: List[str] = []
    res_idx: List[int] = []
    pairs: List[tuple] = []
    prev_c = None
    for r, aa in enumerate(sequence):
        rn = AA_3CODES[aa.upper()]
        order = ["N", "CA", "C", "O"]
        templ = _BACKBONE + _SIDECHAIN[rn]
        for a, b in templ:
            for x in (a, b):
                if x not in order and not (x == "OXT" and r != len(sequence) - 1) and not (x == "CB" and rn == "GLY"):
                    order.append(x)
        if "OXT" in order:  # the terminal oxygen closes the residue, as tleap writes it
            order.remove("OXT")
            order.append("OXT")
        base = len(names)
        idx = {x: base + i for i, x in enumerate(order)}
        names += order
        res += [rn] * len(order)
        res_idx += [r] * len(order)
        if prev_c is not None:
            pairs.append((prev_c, idx["N"]))
        for a, b in templ:
            if a in idx and b in idx:
                pairs.append((min(idx[a], idx[b]), max(idx[a], idx[b])))
        prev_c = idx["C"]
    pairs = sorted(set(pairs))
    n = len(names)
    rng = np.random.RandomState(seed)
    nbr = [[] for _ in range(n)]
    for a, b in pairs:
        nbr[a].append(b)
        nbr[b].append(a)
    pos = np.zeros((n, 3))
    placed = [0]
    seen = {0}
    queue = [0]
    while queue:  # breadth-first growth along the bond graph
        p = queue.pop(0)
        for c in nbr[p]:
            if c in seen:
                continue
            thr = min_dist
            for attempt in range(1, 20001):
                d = rng.normal(size=3)
                cand = pos[p] + bond * d / np.linalg.norm(d)
                others = [j for j in placed if j != p]
                if not others or np.min(np.linalg.norm(pos[others] - cand, axis=1)) >= thr:
                    break
                if attempt % 200 == 0:
                    thr *= 0.95
            pos[c] = cand
            seen.add(c)
            placed.append(c)
            queue.append(c)
    bonded = np.zeros((n, n), dtype=bool)
    for a, b in pairs:
        bonded[a, b] = bonded[b, a] = True
    ia, ib = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
    for _ in range(300):  # relaxation: bonds to their length (closes the rings), non-bonded pairs apart
        d = pos[:, None] - pos[None]
        r = np.linalg.norm(d, axis=-1) + np.eye(n)
        push = np.where(~bonded & (r < min_dist) & ~np.eye(n, dtype=bool), (min_dist - r) / r, 0.0)
        step = 0.25 * (push[..., None] * d).sum(1)
        db = pos[ia] - pos[ib]
        rb = np.linalg.norm(db, axis=-1, keepdims=True)
        f = 0.25 * (bond - rb) / rb * db
        np.add.at(step, ia, f)
        np.add.at(step, ib, -f)
        pos = pos + step
    els = [x[0] for x in names]
    return dict(
        pos=torch.tensor(pos - pos.mean(0), dtype=torch.float32),
        atom_type_index=torch.tensor([encode_atom_type(e) for e in els], dtype=torch.int32),
        atom_code_index=torch.tensor([encode_atom_code(x) for x in names], dtype=torch.int32),
        residue_code_index=torch.tensor([encode_residue(x) for x in res], dtype=torch.int32),
        residue_sequence_index=torch.tensor(res_idx, dtype=torch.int32),
        bonds=torch.tensor(pairs, dtype=torch.long).T.contiguous(),
        atom_names=names, residues=res, elements=els, residue_ids=[i + 1 for i in res_idx], chain_index=[0] * n,
    )


def all_dipeptides() -> List[str]:
    """The 400 two-letter codes in the alphabetical order of ``residue_metadata.py:34-55``'s one-letter keys."""
    aa = sorted(AA_3CODES)
    return [a + b for a in aa for b in aa]


def write_timewarp_tree(root: str, codes: Sequence[str], n_frames: int = 3, seed: int = 0) -> Dict[str, dict]:
    """A directory in the layout of Timewarp's ``2AA-1-large/test`` (what ``sample_uncapped_2AA.yaml:8-13`` points
    ``parse_datasets_from_directory`` at): per code ``<code>-traj-state0.pdb`` — ALL atoms, one hydrogen after every backbone N
    included, so the trajectory arrays carry more atoms than the model sees — and ``<code>-traj-arrays.npz`` with
    ``positions [n_frames, all atoms, 3]`` in nm (frame t = the structure displaced by a seeded 0.01 nm jitter).  Returns
    ``{code: heavy-atom molecule dict of frame 0}``."""
    import os

    os.makedirs(root, exist_ok=True)
    out = {}
    for ci, code in enumerate(codes):
        mol = peptide(code, seed=seed + ci)
        heavy = mol["pos"].double().numpy()
        recs, xyz = [], []
        for i, name in enumerate(mol["atom_names"]):
            recs.append((name, mol["residues"][i], mol["residue_ids"][i], mol["elements"][i]))
            xyz.append(heavy[i])
            if name == "N":
                recs.append(("H", mol["residues"][i], mol["residue_ids"][i], "H"))
                xyz.append(heavy[i] + np.array([0.0, 0.0, 0.1]))
        xyz = np.array(xyz)
        rng = np.random.RandomState(1000 + seed + ci)
        frames = np.stack([xyz + (0.01 * rng.normal(size=xyz.shape) if t else 0.0) for t in range(n_frames)]).astype(np.float32)
        with open(os.path.join(root, f"{code}-traj-state0.pdb"), "w") as f:
            for i, (name, rn, rid, el) in enumerate(recs):
                x, y, z = (frames[0, i] * 10).tolist()
                f.write(f"ATOM  {i + 1:5d} {name:<4s} {rn:>3s} A{rid:4d}    {x:8.3f}{y:8.3f}{z:8.3f}  1.00  0.00          {el:>2s}\n")
            f.write("END\n")
        np.savez(os.path.join(root, f"{code}-traj-arrays.npz"), positions=frames)
        out[code] = mol
    return out


def _irreps_muls(s: str):
    m0 = m1 = 0
    for part in s.split("+"):
        part = part.strip()
        mul, ir = part.split("x")
        if ir.strip() == "0e":
            m0 += int(mul)
        elif ir.strip() == "1e":
            m1 += int(mul)
        else:
            raise NotImplementedError(part)
    return m0, m1


def synthetic_state_dict(arch: Optional[dict] = None, seed: int = 0, output_gain: float = 0.5, tp_weight_scale: float = 13.0,
                         separable: bool = False) -> Dict[str, torch.Tensor]:
    """Random-init parameters with the reference's names/shapes (no ``g.`` prefix).

    ``o3.Linear`` weights ~ N(0,1) (e3nn default); ``nn.Linear`` / ``nn.Embedding`` use torch's default
    init; the radial net's output layer is scaled by ``tp_weight_scale`` so messages are O(1);
    noise-scaling last layers are ``(0, 1) + N(0, 0.1)`` (``noise_conditioning.py:39-42`` initialises to
    exactly (0, 1)); ``output_gain`` is non-zero (the reference initialises it to 0, ``e3conv.py:85``).
    """
    arch = arch or default_arch()
    g = torch.Generator().manual_seed(seed)
    m0, m1 = _irreps_muls(arch["irreps_hidden"])
    E = arch["edge_attr_dim"]
    emb_dims = [
        arch["atom_type_embedding_dim"],
        arch["atom_code_embedding_dim"],
        arch["residue_code_embedding_dim"],
        arch["residue_index_embedding_dim"],
    ]
    # src/jamun/model/atom_embedding.py:54-56 uses atom_type dim twice for irreps bookkeeping
    emb_irreps_muls = [emb_dims[0], emb_dims[0], emb_dims[2], emb_dims[3]]
    n_emb = sum(emb_irreps_muls)
    assert n_emb == sum(emb_dims), "atom_type/atom_code embedding dims must match (reference quirk, atom_embedding.py:54-56)"

    def randn(*shape):
        return torch.randn(*shape, generator=g)

    def lin(out_f, in_f, scale=1.0):
        bound = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound * scale
        b = (torch.rand(out_f, generator=g) * 2 - 1) * bound * scale
        return w, b

    sd: Dict[str, torch.Tensor] = {}
    sd["embed_bondedness.weight"] = randn(2, E // 2)
    sd["atom_embedder.atom_type_embedding.weight"] = randn(20, emb_dims[0])
    sd["atom_embedder.atom_code_embedding.weight"] = randn(10, emb_dims[1])
    sd["atom_embedder.residue_code_embedding.weight"] = randn(25, emb_dims[2])
    sd["atom_embedder.residue_index_embedding.weight"] = randn(10, emb_dims[3])

    def noise_mlp(prefix, k):
        w, b = lin(k, 1)
        sd[prefix + ".0.weight"], sd[prefix + ".0.bias"] = w, b
        sd[prefix + ".2.weight"] = 0.1 * randn(k, k) / math.sqrt(k)
        sd[prefix + ".2.bias"] = 1.0 + 0.1 * randn(k)

    def conv_block(prefix, mul_in_scalar_blocks: Sequence[int], mul1_in: int):
        n0_in = sum(mul_in_scalar_blocks)
        # skip: Linear(in -> hidden); self: Linear(hidden -> hidden)
        sd[prefix + ".gated_conv.skip_connection.weight"] = randn(n0_in * m0 + mul1_in * m1)
        sd[prefix + ".gated_conv.self_interaction.weight"] = randn(m0 * m0 + m1 * m1)
        # FCTP(in, 1x0e+1x1e, (m0+m1)x0e + m1x1e) weight count
        g0, g1 = m0 + m1, m1
        numel = 0
        for mul in mul_in_scalar_blocks:
            numel += mul * g0 + mul * g1
        if mul1_in:
            numel += mul1_in * g1 + mul1_in * g0 + mul1_in * g1
        lin_numel = numel
        if separable:
            # SeparableTensorProduct (e3tools/nn/_tensor_product.py:27-47): depth-wise "uvu" weights — one per input channel and kept
            # (in1 x sh -> out) triple — and the point-wise o3.Linear(irreps_out_dtp -> gate input irreps), whose weight count equals
            # the fully connected product's
            numel = 2 * n0_in + 3 * mul1_in
            sd[prefix + ".gated_conv.f.f.tp.lin.weight"] = randn(lin_numel)
        w0, b0 = lin(E, E)
        w3, b3 = lin(numel, E, scale=(1.0 if separable else tp_weight_scale) * (8.0 if separable else 1.0))
        sd[prefix + ".gated_conv.f.f.radial_nn.0.weight"], sd[prefix + ".gated_conv.f.f.radial_nn.0.bias"] = w0, b0
        sd[prefix + ".gated_conv.f.f.radial_nn.3.weight"], sd[prefix + ".gated_conv.f.f.radial_nn.3.bias"] = w3, b3

    noise_mlp("initial_noise_scaling.scale_predictor", n_emb)
    conv_block("initial_projector", emb_irreps_muls, 0)
    for i in range(arch["n_layers"]):
        conv_block(f"layers.{i}", [m0], m1)
        noise_mlp(f"noise_scalings.{i}.scale_predictor", m0 + m1)
        noise_mlp(f"skip_connections.{i}.weights.scale_predictor", m0 + m1)
        # skip weights go through a sigmoid: centre them at 0 instead of 1 for a balanced mix
        sd[f"skip_connections.{i}.weights.scale_predictor.2.bias"] -= 1.0
    sd["output_head.0.lin.weight"] = randn(m0 * (m0 + m1) + m1 * m1)
    sd["output_head.1.weight"] = randn(m1)
    sd["output_gain"] = torch.tensor(float(output_gain))
    return sd


def trained_like(sd: Dict[str, torch.Tensor], arch: Optional[dict] = None, seed: int = 1, log2_spread: float = 10.0,
                 outlier_fraction: float = 0.01, outlier_factor: float = 100.0) -> Dict[str, torch.Tensor]:
    """Re-shape the weight DISTRIBUTION of a synthetic state dict towards what training produces, keeping the function's overall
    size: (a) a per-channel gauge of the hidden features — channel u of ``irreps_hidden`` carries 2^t_u times its former values in
    every layer, t_u ~ U(-log2_spread, log2_spread) rounded to integers (exact powers of two), with the producers of the channel
    (skip / self-interaction output columns) multiplied by 2^t_u and its consumers (tensor-product weights, skip inputs, head
    inputs of the channel, and the per-channel noise scalings stay as they are) by 2^-t_u; (b) heavy tails — a fraction
    ``outlier_fraction`` of the entries of every large weight tensor times ``outlier_factor``, the tensor then rescaled to its former
    Frobenius norm.  The default checkpoints have N(0,1) / uniform weights of one scale per tensor, which is the easy case for
    the f16x3 kernels (one power-of-two scale per layer and operand); this preset is the hard one."""
    arch = arch or default_arch()
    g = torch.Generator().manual_seed(seed)
    m0, m1 = _irreps_muls(arch["irreps_hidden"])
    G0, G1 = m0 + m1, m1
    t0 = torch.randint(-int(log2_spread), int(log2_spread) + 1, (m0,), generator=g).double()
    t1 = torch.randint(-int(log2_spread), int(log2_spread) + 1, (m1,), generator=g).double()
    s0, s1 = torch.pow(2.0, t0), torch.pow(2.0, t1)
    out = {k: v.clone().double() for k, v in sd.items()}
    emb = [arch["atom_type_embedding_dim"], arch["atom_type_embedding_dim"], arch["residue_code_embedding_dim"], arch["residue_index_embedding_dim"]]
    n_emb = sum(emb)
    separable = any(k.endswith("tp.lin.weight") for k in sd)

    def gauge_linear_out(name, in0, in1):  # o3.Linear flat weight [in0 x m0 | in1 x m1]: output channel gauge
        w = out[name]
        a = w[: in0 * m0].reshape(in0, m0) * s0[None, :]
        b = w[in0 * m0 :].reshape(in1, m1) * s1[None, :] if in1 else w[in0 * m0 :]
        out[name] = torch.cat([a.reshape(-1), b.reshape(-1)])

    def gauge_linear_in(name, out0_, out1_):  # input channels are hidden features: rows times 2^-t
        w = out[name]
        a = w[: m0 * out0_].reshape(m0, out0_) / s0[:, None]
        b = w[m0 * out0_ :].reshape(m1, out1_) / s1[:, None]
        out[name] = torch.cat([a.reshape(-1), b.reshape(-1)])

    def gauge_tp_in(prefix):  # rows of radial_nn.3 (weight, bias) = tensor-product weights; input channel u of a hidden layer
        W, b = out[prefix + ".radial_nn.3.weight"], out[prefix + ".radial_nn.3.bias"]
        if separable:  # "uvu" weights: [A m0 | B m0 | C m1 | D m1 | E m1], one per input channel
            f = torch.cat([1 / s0, 1 / s0, 1 / s1, 1 / s1, 1 / s1])
        else:  # FCTP blocks in instruction order: (0e 0e->0e) m0 x G0, (0e 1e->1e) m0 x G1, (1e 0e->1e) m1 x G1, (1e 1e->0e) m1 x G0, (1e 1e->1e) m1 x G1
            f = torch.cat([(1 / s0)[:, None].expand(m0, G0).reshape(-1), (1 / s0)[:, None].expand(m0, G1).reshape(-1),
                           (1 / s1)[:, None].expand(m1, G1).reshape(-1), (1 / s1)[:, None].expand(m1, G0).reshape(-1),
                           (1 / s1)[:, None].expand(m1, G1).reshape(-1)])
        assert f.numel() == b.numel(), (f.numel(), b.numel())
        out[prefix + ".radial_nn.3.weight"], out[prefix + ".radial_nn.3.bias"] = W * f[:, None], b * f
        if separable:  # the point-wise Linear sees the depth-wise outputs, already in the un-gauged size: nothing to do
            pass

    gauge_linear_out("initial_projector.gated_conv.skip_connection.weight", n_emb, 0)
    gauge_linear_out("initial_projector.gated_conv.self_interaction.weight", m0, m1)
    for i in range(arch["n_layers"]):
        pre = f"layers.{i}.gated_conv"
        gauge_linear_in(pre + ".skip_connection.weight", m0, m1)
        gauge_linear_out(pre + ".skip_connection.weight", m0, m1)
        gauge_linear_out(pre + ".self_interaction.weight", m0, m1)
        gauge_tp_in(pre + ".f.f")
    gauge_linear_in("output_head.0.lin.weight", G0, m1)
    for k, v in out.items():  # heavy tails
        if v.numel() >= 1024 and "embedding" not in k and "embed_" not in k:
            nrm = v.norm()
            mask = torch.rand(v.shape, generator=g) < outlier_fraction
            v = torch.where(mask, v * outlier_factor, v)
            out[k] = v * (nrm / v.norm())
    return {k: v.to(torch.float32) for k, v in out.items()}


def synthetic_checkpoint(
    arch: Optional[dict] = None,
    seed: int = 0,
    max_radius: float = 1.0,
    average_squared_distance: float = 0.332,
    output_gain: float = 0.5,
    prefix: str = "g.",
    separable: bool = False,
    weights: str = "gaussian",
) -> dict:
    """A Lightning-shaped checkpoint dict: ``{"state_dict": {"g.<name>": ...}, "hyper_parameters": {...}}``.  ``weights``:
    "gaussian" (one scale per tensor) or "trained_like" (``trained_like``: per-channel scale spread 2^+-10, 1 % outliers x 100)."""
    arch = arch or default_arch()
    sd = synthetic_state_dict(arch, seed=seed, output_gain=output_gain, separable=separable)
    if weights == "trained_like":
        sd = trained_like(sd, arch, seed=seed + 1)
    elif weights != "gaussian":
        raise ValueError(weights)
    if separable:  # as e3conv_separable.yaml:14-19 reaches the checkpoint's hyper-parameters (a partial of ConvBlock with conv = SeparableConv)
        arch = dict(arch, hidden_layer_factory={"_target_": "jamun.e3tools.nn.ConvBlock", "_partial_": True,
                                                "conv": {"_target_": "jamun.e3tools.nn.SeparableConv", "_partial_": True}})
    return {
        "state_dict": {prefix + k: v for k, v in sd.items()},
        "hyper_parameters": dict(
            arch=dict(arch),
            max_radius=max_radius,
            average_squared_distance=average_squared_distance,
            mean_center=True,
            use_torch_compile=(prefix != "g."),
        ),
        "pytorch-lightning_version": "2.4.0",
    }


def replicate(mol: dict, walkers: int) -> List[dict]:
    """``repeat_init_samples`` walkers of the same molecule, consecutive (``src/jamun/cmdline/sample.py:35-37``)."""
    return [mol for _ in range(walkers)]
