"""Op-for-op CPU restatement of the reference denoiser forward (E3Conv + Denoiser.xhat/score).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED (e3nn /
torch_cluster / torch_scatter arithmetic restated in ``e3.py`` / ``graph.py``).

The algorithmic structure is kept exactly as the reference executes it on CPU:
per-edge tensor-product weights are *materialised* (``[E, weight_numel]``), the
tensor product is an einsum over them, aggregation is ``index_add_``-based
mean, the radius graph is rebuilt every forward.

``params`` is a plain ``{name: tensor}`` dict using the reference's state-dict
names without the ``g.`` / ``g._orig_mod.`` prefix (SURVEY.md §8 b);
``hp`` is a dict of hyper-parameters (``default_hparams``).
"""

from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import e3
from .graph import mean_center, radius_graph, scatter_mean


def default_hparams(**over) -> dict:
    """``src/jamun/hydra_config/model/arch/e3conv.yaml:3-14`` + ``model/denoiser.yaml:8-14``."""
    hp = dict(
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
        max_radius=1.0,
        average_squared_distance=0.332,
        mean_center=True,
        conv="conv",  # hidden_layer_factory.conv: "conv" = e3tools.nn.Conv (e3conv.yaml), "separable" = SeparableConv (e3conv_separable.yaml)
    )
    hp.update(over)
    return hp


def _mlp2(c: torch.Tensor, p: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """``Linear(1->k) . SELU . Linear(k->k)`` (``src/jamun/model/noise_conditioning.py:33-37``)."""
    h = F.linear(c, p[prefix + ".0.weight"], p[prefix + ".0.bias"])
    h = F.selu(h)
    return F.linear(h, p[prefix + ".2.weight"], p[prefix + ".2.bias"])


def noise_scaling(x, c_noise, p, prefix, irreps):
    """``NoiseConditionalScaling.forward`` (``src/jamun/model/noise_conditioning.py:50-54``)."""
    scales = _mlp2(c_noise.reshape(1, 1), p, prefix + ".scale_predictor")
    return e3.elementwise_scale(x, scales, irreps)


def noise_skip(x1, x2, c_noise, p, prefix, irreps):
    """``NoiseConditionalSkipConnection.forward`` (``src/jamun/model/noise_conditioning.py:69-73``)."""
    w = torch.sigmoid(_mlp2(c_noise.reshape(1), p, prefix + ".weights.scale_predictor"))
    return e3.elementwise_scale(x1, w, irreps) + e3.elementwise_scale(x2, 1 - w, irreps)


def atom_embedding(topo: dict, p, hp) -> torch.Tensor:
    """``AtomEmbeddingWithResidueInformation.forward`` (``src/jamun/model/atom_embedding.py:58-76``)."""
    feats = [
        F.embedding(topo["atom_type_index"].long(), p["atom_embedder.atom_type_embedding.weight"]),
        F.embedding(topo["atom_code_index"].long(), p["atom_embedder.atom_code_embedding.weight"]),
        F.embedding(topo["residue_code_index"].long(), p["atom_embedder.residue_code_embedding.weight"]),
    ]
    rsi = topo["residue_sequence_index"].long()
    if not hp["use_residue_sequence_index"]:
        rsi = torch.zeros_like(rsi)
    feats.append(F.embedding(rsi, p["atom_embedder.residue_index_embedding.weight"]))
    return torch.cat(feats, dim=-1)


def atom_embedding_irreps(hp) -> e3.Irreps:
    # src/jamun/model/atom_embedding.py:54-56 (atom_type dim used twice; harmless at 8/8)
    return [
        (hp["atom_type_embedding_dim"], 0),
        (hp["atom_type_embedding_dim"], 0),
        (hp["residue_code_embedding_dim"], 0),
        (hp["residue_index_embedding_dim"], 0),
    ]


def gate_in_irreps(irreps_out: e3.Irreps) -> e3.Irreps:
    """``e3nn.nn.Gate.irreps_in`` for ``irreps_out = a x0e + b x1e``: ``(a+b)x0e + b x1e`` (sorted + simplified)."""
    a = sum(m for m, l in irreps_out if l == 0)
    b = sum(m for m, l in irreps_out if l == 1)
    return [(a + b, 0), (b, 1)] if b else [(a, 0)]


def conv(x, edge_index, edge_attr, edge_sh, p, prefix, irreps_in, irreps_sh, irreps_out, kind="conv"):
    """``Conv.forward`` (``src/jamun/e3tools/nn/_conv.py:96-119``) with ``ScalarMLP(64->[64]->P)`` radial net (``_mlp.py:10-34``);
    ``kind="separable"``: ``SeparableConv`` (``_conv.py:122-135``) — the same forward with ``SeparableTensorProduct`` as ``self.tp``
    (its point-wise Linear is the parameter ``tp.lin.weight``)."""
    N = x.shape[0]
    src, dst = edge_index
    h = F.silu(F.linear(edge_attr, p[prefix + ".radial_nn.0.weight"], p[prefix + ".radial_nn.0.bias"]))
    w = F.linear(h, p[prefix + ".radial_nn.3.weight"], p[prefix + ".radial_nn.3.bias"])  # [E, weight_numel] materialised
    if kind == "separable":
        out_ij = e3.separable_tp(x[src], edge_sh, w, p[prefix + ".tp.lin.weight"], irreps_in, irreps_sh, irreps_out)
    else:
        out_ij = e3.fctp(x[src], edge_sh, w, irreps_in, irreps_sh, irreps_out)
    return scatter_mean(out_ij, dst, N)


def conv_block(x, edge_index, edge_attr, edge_sh, p, prefix, irreps_in, irreps_sh, irreps_out, kind="conv"):
    """``ConvBlock`` = ``LinearSelfInteraction(Gated(Conv))`` (``_conv.py:204-221``, ``_interaction.py:26-30``, ``_gate.py:107-110``)."""
    gin = gate_in_irreps(irreps_out)
    s = e3.linear(x, p[prefix + ".gated_conv.skip_connection.weight"], irreps_in, irreps_out)
    m = conv(x, edge_index, edge_attr, edge_sh, p, prefix + ".gated_conv.f.f", irreps_in, irreps_sh, gin, kind)
    mul0 = sum(mm for mm, l in irreps_out if l == 0)
    mul1 = sum(mm for mm, l in irreps_out if l == 1)
    g = e3.gate(m, mul0, mul1)
    y = e3.linear(g, p[prefix + ".gated_conv.self_interaction.weight"], irreps_out, irreps_out)
    return y + s


def e3conv_forward(pos_scaled, topo, edge_index, bond_mask, c_noise, radial_cutoff, p, hp, return_intermediates=False):
    """``E3Conv.forward`` (``src/jamun/model/arch/e3conv.py:87-138``)."""
    irreps_hidden = e3.parse_irreps(hp["irreps_hidden"])
    irreps_sh = e3.parse_irreps(hp["irreps_sh"])
    irreps_out = e3.parse_irreps(hp["irreps_out"])
    assert irreps_sh == [(1, 0), (1, 1)] and irreps_out == [(1, 1)]
    n_rad = (hp["edge_attr_dim"] + 1) // 2

    src, dst = edge_index
    edge_vec = pos_scaled[src] - pos_scaled[dst]
    edge_sh = e3.spherical_harmonics_01(edge_vec)
    bonded = F.embedding(bond_mask, p["embed_bondedness.weight"])
    radial = e3.soft_one_hot_linspace_gaussian(edge_vec.norm(dim=1), 0.0, radial_cutoff, n_rad)
    edge_attr = torch.cat((bonded, radial), dim=-1)

    inter = {}
    emb_irreps = atom_embedding_irreps(hp)
    x = atom_embedding(topo, p, hp)
    x = noise_scaling(x, c_noise, p, "initial_noise_scaling", emb_irreps)
    kind = hp.get("conv", "conv")
    x = conv_block(x, edge_index, edge_attr, edge_sh, p, "initial_projector", emb_irreps, irreps_sh, irreps_hidden, kind)
    inter["x0"] = x
    for i in range(hp["n_layers"]):
        xs = noise_scaling(x, c_noise, p, f"noise_scalings.{i}", irreps_hidden)
        y = conv_block(xs, edge_index, edge_attr, edge_sh, p, f"layers.{i}", irreps_hidden, irreps_sh, irreps_hidden, kind)
        x = noise_skip(x, y, c_noise, p, f"skip_connections.{i}", irreps_hidden)
        inter[f"x{i + 1}"] = x
    # EquivariantMLP head (src/jamun/e3tools/nn/_mlp.py:84-114)
    gin = gate_in_irreps(irreps_hidden)
    h = e3.linear(x, p["output_head.0.lin.weight"], irreps_hidden, gin)
    mul0 = sum(mm for mm, l in irreps_hidden if l == 0)
    mul1 = sum(mm for mm, l in irreps_hidden if l == 1)
    h = e3.gate(h, mul0, mul1)
    out = e3.linear(h, p["output_head.1.weight"], irreps_hidden, irreps_out)
    out = out * p["output_gain"]
    if return_intermediates:
        inter["edge_attr"] = edge_attr
        inter["edge_sh"] = edge_sh
        return out, inter
    return out


def normalization_factors(sigma, average_squared_distance, D=3, dtype=torch.float32):
    """``Denoiser.normalization_factors`` (``src/jamun/model/denoiser.py:116-126``), computed in ``dtype`` as the reference does."""
    sigma = torch.as_tensor(sigma, dtype=dtype)
    A = torch.as_tensor(average_squared_distance, dtype=dtype)
    B = torch.as_tensor(2 * D * sigma**2)
    c_in = 1.0 / torch.sqrt(A + B)
    c_skip = A / (A + B)
    c_out = torch.sqrt((A * B) / (A + B))
    c_noise = torch.log(sigma) / 4
    return c_in, c_skip, c_out, c_noise


def add_edges(pos, topo, radial_cutoff):
    """``Denoiser.add_edges`` (``src/jamun/model/denoiser.py:138-166``): radial edges then the dataset's bonded edges."""
    radial = radius_graph(pos, radial_cutoff, topo["batch"])
    bonds = topo["bonds"]
    edge_index = torch.cat((radial, bonds), dim=-1)
    bond_mask = torch.cat(
        (torch.zeros(radial.shape[1], dtype=torch.long), torch.ones(bonds.shape[1], dtype=torch.long)), dim=0
    )
    return edge_index, bond_mask


def xhat(y, topo, sigma, p, hp, return_intermediates=False):
    """``Denoiser.xhat`` (``src/jamun/model/denoiser.py:203-217``) -> denoised positions ``[N,3]``."""
    dtype = y.dtype
    B = topo["num_graphs"]
    if hp["mean_center"]:
        y = mean_center(y, topo["batch"], B)
    # xhat_normalized (src/jamun/model/denoiser.py:168-201)
    sig = torch.as_tensor(sigma, dtype=dtype)
    c_in, c_skip, c_out, c_noise = normalization_factors(sig, hp["average_squared_distance"], y.shape[-1], dtype)
    radial_cutoff = torch.sqrt(torch.as_tensor(hp["max_radius"] ** 2, dtype=dtype) + 6 * sig**2) / c_in
    edge_index, bond_mask = add_edges(y, topo, radial_cutoff)
    y_scaled = y * c_in
    g = e3conv_forward(y_scaled, topo, edge_index, bond_mask, c_noise, radial_cutoff, p, hp, return_intermediates)
    inter = None
    if return_intermediates:
        g, inter = g
        inter["edge_index"] = edge_index
        inter["bond_mask"] = bond_mask
        inter["y_centered"] = y
        inter["g"] = g
    x = c_skip * y + c_out * g
    if hp["mean_center"]:
        x = mean_center(x, topo["batch"], B)
    if return_intermediates:
        return x, inter
    return x


def score(y, topo, sigma, p, hp):
    """``Denoiser.score`` (``src/jamun/model/denoiser.py:111-114``): raw ``y`` in the numerator, centred ``xhat``."""
    sig = torch.as_tensor(sigma, dtype=y.dtype)
    return (xhat(y, topo, sigma, p, hp) - y) / (sig**2)
