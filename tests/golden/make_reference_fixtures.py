"""PIN KIT for SURVEY.md section 8 rows a7-a18: fixtures produced by the REFERENCE'S OWN classes on the inputs the oracle fixtures use.

    python tests/golden/make_reference_fixtures.py [case ...]        # needs e3nn 0.5.4, torch_geometric, torch_cluster, torch_scatter, lightning

The denoiser forward's arithmetic lives in third-party wheels (e3nn / torch_cluster / torch_scatter) that are neither vendored under
/root/reference nor installed in the build image, so `oracle/{e3,graph,denoiser}.py` are pinned only by known-answer properties
("PARITY UNPINNED").  This script is the one command that pins them the day such an environment exists; in THIS image it stops at
`import e3nn` (tests/test_host.py::test_reference_pin_kit_reaches_the_third_party_line asserts exactly that, which proves every
import and every line before the third-party one).  It runs in the build container only: nothing of it travels to the GPU box except
the `.npz` vectors it writes.

What it does — no reference source is copied, the reference package is imported from /root/reference/src:
  * instantiates `jamun.model.arch.E3Conv` with `jamun.e3tools.nn.ConvBlock` / `Conv` (or `SeparableConv`) / `EquivariantMLP` factories
    exactly as `hydra_config/model/arch/e3conv.yaml:3-33` does (the Hydra `_partial_` entries become `functools.partial`), inside
    `jamun.model.Denoiser` (`hydra_config/model/denoiser.yaml`, `use_torch_compile=False`);
  * `load_state_dict(jamun_amd.synth.synthetic_checkpoint(...)["state_dict"])` — the SAME seeded weights the oracle fixtures use; missing /
    unexpected keys other than e3nn's bookkeeping buffers abort the run (this is also the check of SURVEY section 8(b)'s key list);
  * runs `Denoiser.xhat` / `Denoiser.score` (`model/denoiser.py:111-217`) on the inputs of `make_oracle_fixtures.forward_case` (same
    molecules, same `torch.manual_seed(2)` noise) with forward hooks on the initial projector, every skip connection and the output head
    -> `reference_forward_<kind>.npz` with the keys of `oracle_forward_<kind>.npz` (`y, xhat, score, g, deg, n_edges, x0..x5`) plus the
    edge list of the run and which radius-graph implementation produced it (torch_cluster CPU = kd-tree order above 32 neighbours;
    with a GPU build also CUDA = index order — `oracle/graph.py:25-26`);
  * unit operators on seeded inputs -> `reference_ops.npz`: `o3.SphericalHarmonics("1x0e+1x1e", normalize=True, "component")`,
    `soft_one_hot_linspace(gaussian, cutoff=True)`, `o3.wigner_3j(l1,l2,l3)` for l <= 1, `o3.FullyConnectedTensorProduct` with
    per-sample weights, `o3.Linear`, `e3tools.nn.Gate`, `normalize2mom` constants, `radius_graph` (CPU, and CUDA when available),
    `torch_scatter.scatter(reduce="mean")`.
Consumers: tests/test_oracle.py::test_oracle_matches_reference_fixtures (CPU: oracle vs these) and
tests/test_gpu_parity.py::test_forward_matches_reference_fixtures (GPU: HIP path vs these); both skip while no `reference_*.npz` exists.
"""
import functools
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = os.environ.get("JAMUN_REFERENCE_SRC", "/root/reference/src")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from jamun_amd import synth  # noqa: E402  (inputs and weights: the same generators the oracle fixtures use)

# (kind, variant): the oracle_forward_* cases VERDICT r5 names + one SeparableConv and one trained-like case
FORWARD_CASES = {
    "ag4": ("ag4", None), "chain17x6": ("chain17x6", None), "ragged": ("ragged", None), "chig93x2": ("chig93x2", None),
    "dipep48": ("dipep48", None), "sep_ag4": ("ag4", "sep"), "trained_ragged": ("ragged", "trained"), "nl2_ragged": ("ragged", "nl2"),
}
THIRD_PARTY = ("e3nn", "torch_geometric", "torch_cluster", "torch_scatter", "lightning")


def import_reference():
    """Third-party wheels first (so that a missing one is reported by ITS name, not from inside the reference), then the reference."""
    import importlib

    mods = {}
    for name in THIRD_PARTY:  # e3nn is first: in the build image the script ends here with ModuleNotFoundError("No module named 'e3nn'")
        mods[name] = importlib.import_module(name)
    if not os.path.isdir(os.path.join(REF_SRC, "jamun")):
        raise FileNotFoundError(f"{REF_SRC}/jamun not found: run in the build container (or set JAMUN_REFERENCE_SRC)")
    sys.path.insert(0, REF_SRC)
    import jamun.e3tools.nn as e3nn_tools  # noqa: E402
    import jamun.model as jm  # noqa: E402
    import jamun.model.arch as arch  # noqa: E402
    import jamun.utils as ju  # noqa: E402

    return mods, jm, arch, e3nn_tools, ju


def build_reference_denoiser(jm, arch_mod, tools, ckpt):
    """`hydra_config/model/denoiser.yaml` + `arch/e3conv.yaml` (or `e3conv_separable.yaml`) as plain Python."""
    hp = ckpt["hyper_parameters"]
    a = dict(hp["arch"])
    separable = a.pop("hidden_layer_factory", None) is not None
    conv = tools.SeparableConv if separable else tools.Conv
    arch_factory = functools.partial(
        arch_mod.E3Conv,
        irreps_out=a.get("irreps_out", "1x1e"), irreps_hidden=a["irreps_hidden"], irreps_sh=a.get("irreps_sh", "1x0e + 1x1e"),
        n_layers=a["n_layers"], edge_attr_dim=a["edge_attr_dim"],
        atom_type_embedding_dim=a["atom_type_embedding_dim"], atom_code_embedding_dim=a["atom_code_embedding_dim"],
        residue_code_embedding_dim=a["residue_code_embedding_dim"], residue_index_embedding_dim=a["residue_index_embedding_dim"],
        use_residue_information=a.get("use_residue_information", True), use_residue_sequence_index=a.get("use_residue_sequence_index", False),
        hidden_layer_factory=functools.partial(tools.ConvBlock, conv=functools.partial(conv)),
        output_head_factory=functools.partial(tools.EquivariantMLP, irreps_hidden_list=[a["irreps_hidden"]]),
    )
    den = jm.Denoiser(
        arch=arch_factory, optim=None, sigma_distribution=None, max_radius=hp["max_radius"],
        average_squared_distance=hp["average_squared_distance"], add_fixed_noise=False, add_fixed_ones=False,
        align_noisy_input_during_training=True, align_noisy_input_during_evaluation=True, mean_center=hp.get("mean_center", True),
        mirror_augmentation_rate=0.0, use_torch_compile=False,
    )
    res = den.load_state_dict(ckpt["state_dict"], strict=False)
    bookkeeping = ("output_mask", "_w3j_", "_compiled_main", ".tp.weight", "sh._", "normalize2mom")
    missing = [k for k in res.missing_keys if not any(b in k for b in bookkeeping)]
    unexpected = [k for k in res.unexpected_keys if not any(b in k for b in bookkeeping)]
    if missing or unexpected:
        raise RuntimeError(f"synthetic checkpoint does not match the reference's parameter names: missing {missing[:8]} unexpected {unexpected[:8]}")
    return den.eval()


def to_batch(mods, ju, mols):
    """`Batch.from_data_list` of `DataWithResidueInformation` graphs (`utils/data_with_residue_info.py:17-33`; what `data/_mdtraj.py:75-89` builds)."""
    Batch = mods["torch_geometric"].data.Batch
    graphs = []
    for m in mols:
        n_res = int(m["residue_sequence_index"].max().item()) + 1
        graphs.append(ju.DataWithResidueInformation(
            pos=m["pos"].clone(), edge_index=m["bonds"].long(), atom_type_index=m["atom_type_index"].long(), atom_code_index=m["atom_code_index"].long(),
            residue_code_index=m["residue_code_index"].long(), residue_sequence_index=m["residue_sequence_index"].long(),
            residue_index=torch.arange(n_res), num_residues=n_res, loss_weight=1.0))
    return Batch.from_data_list(graphs)


def forward_case(mods, jm, arch_mod, tools, ju, kind, variant, device="cpu"):
    import make_oracle_fixtures as mof  # (molecules / VARIANTS / GAINS: one definition of the inputs for oracle and reference fixtures)

    mols = mof.molecules(kind)
    if variant == "sep":
        ckpt, sigma = synth.synthetic_checkpoint(output_gain=mof.GAINS["strong"], separable=True), mof.SIGMA
    elif variant is not None:
        ckpt, sigma = mof.variant_checkpoint(variant), mof.VARIANTS[variant]["sigma"]
    else:
        ckpt, sigma = synth.synthetic_checkpoint(output_gain=mof.GAINS["strong"]), mof.SIGMA
    den = build_reference_denoiser(jm, arch_mod, tools, ckpt).to(device)
    batch = to_batch(mods, ju, mols)
    torch.manual_seed(2)
    y_pos = batch.pos + sigma * torch.randn_like(batch.pos)  # (same draw as make_oracle_fixtures.forward_case: one randn_like over [N,3])
    y = batch.clone()
    y.pos = y_pos.clone()
    y = y.to(device)
    inter = {}
    g = den.g
    hooks = [g.initial_projector.register_forward_hook(lambda m, i, o: inter.__setitem__("x0", o.detach().cpu()))]
    for l, skip in enumerate(g.skip_connections):
        hooks.append(skip.register_forward_hook(lambda m, i, o, l=l: inter.__setitem__(f"x{l + 1}", o.detach().cpu())))
    hooks.append(g.output_head.register_forward_hook(lambda m, i, o: inter.__setitem__("head", o.detach().cpu())))
    edges = {}
    orig_add_edges = den.add_edges

    def add_edges_recording(yy, cutoff):  # (the edge list of THIS run: the CPU kd-tree order is data dependent above 32 neighbours)
        out = orig_add_edges(yy, cutoff)
        edges["edge_index"], edges["bond_mask"] = out.edge_index.detach().cpu(), out.bond_mask.detach().cpu()
        return out

    den.add_edges = add_edges_recording
    with torch.no_grad():
        xhat = den.xhat(y, sigma).pos.cpu()
        score = den.score(y, sigma).cpu()
    for h in hooks:
        h.remove()
    ei = edges["edge_index"]
    out = dict(y=y_pos, xhat=xhat, score=score, g=inter["head"] * g.output_gain.detach().cpu(), deg=torch.bincount(ei[1], minlength=y_pos.shape[0]),
               n_edges=torch.tensor(ei.shape[1]), edge_index=ei, bond_mask=edges["bond_mask"], sigma=torch.tensor(sigma),
               radius_graph_device=np.array(device))
    out.update({k: v for k, v in inter.items() if k.startswith("x")})
    return out


def unit_ops(mods, tools):
    e3nn = mods["e3nn"]
    o3 = e3nn.o3
    g = torch.Generator().manual_seed(11)
    out = {}
    vec = torch.randn(64, 3, generator=g)
    vec[0] = 0.0  # (F.normalize eps path)
    out["sh_in"] = vec
    out["sh_out"] = o3.SphericalHarmonics(o3.Irreps("1x0e + 1x1e"), normalize=True, normalization="component")(vec)
    d = torch.rand(128, generator=g) * 0.8
    out["rbf_in"] = d
    out["rbf_out"] = e3nn.math.soft_one_hot_linspace(d, 0.0, 0.58726, 32, basis="gaussian", cutoff=True)
    for l1 in (0, 1):
        for l2 in (0, 1):
            for l3 in (0, 1):
                if abs(l1 - l2) <= l3 <= l1 + l2:
                    out[f"w3j_{l1}{l2}{l3}"] = o3.wigner_3j(l1, l2, l3)
    for tag, i1, io in (("hidden", "120x0e + 32x1e", "152x0e + 32x1e"), ("init", "8x0e + 8x0e + 32x0e + 8x0e", "152x0e + 32x1e"), ("small", "3x0e + 2x1e", "4x0e + 3x1e")):
        tp = o3.FullyConnectedTensorProduct(o3.Irreps(i1), o3.Irreps("1x0e + 1x1e"), o3.Irreps(io), shared_weights=False, internal_weights=False)
        E = 5
        x1, x2, w = torch.randn(E, tp.irreps_in1.dim, generator=g), torch.randn(E, 4, generator=g), torch.randn(E, tp.weight_numel, generator=g)
        out[f"fctp_{tag}_x1"], out[f"fctp_{tag}_x2"], out[f"fctp_{tag}_w"], out[f"fctp_{tag}_out"] = x1, x2, w, tp(x1, x2, w)
    for tag, ii, io in (("self", "120x0e + 32x1e", "120x0e + 32x1e"), ("skip0", "8x0e + 8x0e + 32x0e + 8x0e", "120x0e + 32x1e"), ("head", "120x0e + 32x1e", "152x0e + 32x1e"), ("out", "120x0e + 32x1e", "1x1e")):
        lin = o3.Linear(o3.Irreps(ii), o3.Irreps(io))
        w = torch.randn(lin.weight_numel, generator=g)
        with torch.no_grad():
            lin.weight.copy_(w)
        x = torch.randn(7, lin.irreps_in.dim, generator=g)
        out[f"linear_{tag}_x"], out[f"linear_{tag}_w"], out[f"linear_{tag}_out"] = x, w, lin(x).detach()
    gate = tools.Gate(o3.Irreps("120x0e + 32x1e"))
    x = torch.randn(9, gate.irreps_in.dim, generator=g)
    out["gate_x"], out["gate_out"] = x, gate(x)
    out["gate_irreps_in"] = np.array(str(gate.irreps_in))
    from e3nn.math import normalize2mom

    out["n2m_leaky_relu"] = torch.tensor(normalize2mom(torch.nn.LeakyReLU()).cst)
    out["n2m_sigmoid"] = torch.tensor(normalize2mom(torch.nn.Sigmoid()).cst)
    # radius graph: a dense 48-atom blob (every atom above the 32-neighbour cap) and a ragged pair
    rg = mods["torch_geometric"].nn.radius_graph
    pos = torch.randn(48, 3, generator=g) * 0.25
    batch = torch.zeros(48, dtype=torch.long)
    out["rg_pos"], out["rg_r"] = pos, torch.tensor(0.587)
    out["rg_cpu"] = rg(pos, 0.587, batch)
    if torch.cuda.is_available():
        out["rg_cuda"] = rg(pos.cuda(), 0.587, batch.cuda()).cpu()
    src = torch.randn(40, 6, generator=g)
    idx = torch.randint(0, 9, (40,), generator=g)
    out["scatter_src"], out["scatter_index"] = src, idx
    out["scatter_mean"] = mods["torch_scatter"].scatter(src, idx, dim=0, dim_size=11, reduce="mean")
    return out


def main(argv):
    mods, jm, arch_mod, tools, ju = import_reference()
    only = set(argv)
    versions = {n: getattr(m, "__version__", "?") for n, m in mods.items()}
    print("third-party versions:", versions)
    if not only or "ops" in only:
        ops = unit_ops(mods, tools)
        np.savez_compressed(os.path.join(HERE, "reference_ops.npz"), versions=np.array(str(versions)),
                            **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in ops.items()})
        print("reference_ops", sorted(ops))
    for name, (kind, variant) in FORWARD_CASES.items():
        if only and name not in only:
            continue
        for device in ["cpu"] + (["cuda"] if torch.cuda.is_available() else []):
            out = forward_case(mods, jm, arch_mod, tools, ju, kind, variant, device)
            fn = f"reference_forward_{name}.npz" if device == "cpu" else f"reference_forward_{name}_cuda.npz"
            np.savez_compressed(os.path.join(HERE, fn), versions=np.array(str(versions)),
                                **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
            print(fn, {k: tuple(v.shape) for k, v in out.items() if torch.is_tensor(v)}, flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
