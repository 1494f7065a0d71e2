"""GPU parity tests (``-m gpu``): the HIP path, called through the C ABI, against the CPU oracle.

Tolerance: BASELINE.json's north_star states "denoised coordinates match the reference CPU path to <= 1e-4 Angstrom
RMSD on identical RNG seeds" = 1e-5 nm (the path works in nanometres).  Index/structure outputs (neighbour lists,
degrees, frame counts) are compared exactly.
"""
import json
import os

import numpy as np
import pytest
import torch

from jamun_amd import native  # (native.TUNING: the kernel-selection switches, jamun_tuning)

pytestmark = pytest.mark.gpu

RMSD_TOL_NM = 1e-5  # 1e-4 Angstrom


def rmsd(a, b):
    return ((a.double().cpu() - b.double().cpu()) ** 2).sum(-1).mean().sqrt().item()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def ckpt():
    from jamun_amd import synth

    return synth.synthetic_checkpoint()


def _mols(kind):
    from jamun_amd import synth

    if kind == "ag4":
        m = {k: v for k, v in synth.ag_dipeptide().items()}
        return [m] * 4
    if kind == "chain17x6":
        return [synth.random_chain(17, seed=0)] * 6
    if kind == "ragged":
        return [synth.random_chain(n, seed=s) for s, n in enumerate([5, 17, 33, 9, 57, 2, 1, 29])]
    if kind == "chain33x4":
        return [synth.random_chain(33, seed=0)] * 4
    if kind == "cfg4kinds":
        import random

        rng = random.Random(1234)
        return [synth.random_chain(rng.randint(17, 57), seed=100 + i) for i in range(8)]
    if kind == "dense70":  # > 32 neighbours inside the cutoff: exercises the neighbour cap
        return [synth.random_chain(70, seed=3, bond=0.12, min_dist=0.13)] * 2
    # BASELINE configs[4]: chignolin size, with hydrogens (~166 atoms) and as the reference feeds it (93 heavy atoms)
    if kind == "chig93x2":
        return [synth.random_chain(93, seed=5)] * 2
    if kind == "chig166x2":
        return [synth.random_chain(166, seed=5)] * 2
    if kind == "large_mix":  # (tests/golden/make_oracle_fixtures.py: molecules of 63..120 atoms next to small ones)
        return [synth.random_chain(n, seed=40 + i) for i, n in enumerate([63, 9, 101, 64, 120, 3, 77])]
    raise KeyError(kind)


def _init_path_of(mols):
    """The initial projector jamun_sampler_create picks on the tiles of k_conv_mf: up to 32 distinct embedding rows (atom type, atom
    name, residue — the sequence index is not used) k_conv_mfi (3), more k_conv_mfx (4)."""
    rows = set()
    for m in mols:
        rows |= set(zip(m["atom_type_index"].tolist(), m["atom_code_index"].tolist(), m["residue_code_index"].tolist()))
    return 3 if len(rows) <= 32 else 4


def _oracle_setup(mols, ckpt, dtype=torch.float32):
    from oracle import denoiser as od
    from oracle import graph as og

    tmols = [{k: v for k, v in m.items() if torch.is_tensor(v)} for m in mols]
    topo = og.collate(tmols)
    p = {k[2:]: v.to(dtype) for k, v in ckpt["state_dict"].items()}
    hp = od.default_hparams(max_radius=ckpt["hyper_parameters"]["max_radius"], average_squared_distance=ckpt["hyper_parameters"]["average_squared_distance"])
    return topo, p, hp


# ---- stand-alone operators ------------------------------------------------------------------------------------------


@pytest.mark.parametrize("kind", ["ag4", "ragged", "dense70", "chig93x2", "chig166x2"])
def test_mean_center_and_radius_graph(dev, kind):
    from jamun_amd import native
    from jamun_amd.data import WalkerBatch
    from oracle import graph as og

    batch = WalkerBatch.from_molecules(_mols(kind))
    torch.manual_seed(0)
    pos = batch.pos + 0.04 * torch.randn_like(batch.pos)
    out = native.mean_center(pos.to(dev), batch.ptr)
    ref = og.mean_center(pos, batch.batch, batch.num_graphs)
    assert torch.equal(out.cpu(), ref)  # same summation order -> bit-exact
    r = 0.58726
    ei = native.radius_graph_edge_index(ref.to(dev), r, batch.ptr).cpu()
    ei_ref = og.radius_graph(ref, r, batch.batch)
    assert torch.equal(ei, ei_ref)
    if kind in ("dense70", "chig93x2", "chig166x2"):
        deg = torch.bincount(ei_ref[1], minlength=batch.num_nodes)
        assert deg.max() >= 32  # the cap was actually hit


def test_radius_graph_empty_and_single():
    from jamun_amd import native

    dev = torch.device("cuda", 0)
    pos = torch.zeros(1, 3, device=dev)
    nbr, deg = native.radius_graph(pos, 1.0, torch.tensor([0, 1]))
    assert deg.tolist() == [0]
    # strict inequality at exactly r
    pos = torch.tensor([[0.0, 0, 0], [1.0, 0, 0]], device=dev)
    nbr, deg = native.radius_graph(pos, 1.0, torch.tensor([0, 2]))
    assert deg.tolist() == [0, 0]


@pytest.mark.parametrize("width", [248, 3, 64])
def test_scatter_mean(dev, width):
    from jamun_amd import native
    from oracle import graph as og

    torch.manual_seed(1)
    n_out = 300
    counts = torch.randint(0, 40, (n_out,))
    counts[7] = 0
    counts[n_out - 1] = 0
    seg = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    E = int(seg[-1])
    src = torch.randn(E, width)
    index = torch.repeat_interleave(torch.arange(n_out), counts)
    out = native.scatter_mean(src.to(dev), seg, n_out).cpu()
    ref = og.scatter_mean(src, index, n_out)
    assert torch.equal(out, ref)  # fixed row order on both sides -> bit-exact


def test_edge_geometry_export_matches_oracle(dev):
    """``jamun_edge_geometry`` (stand-alone C-ABI operator) vs the oracle's e3nn restatement of
    ``E3Conv.forward``'s edge features (``arch/e3conv.py:114-123``): spherical harmonics and the Gaussian radial basis."""
    from jamun_amd import native, synth
    from oracle import e3

    torch.manual_seed(0)
    mol = synth.random_chain(40, seed=4)
    pos = 1.7 * mol["pos"]  # "scaled" positions
    n = pos.shape[0]
    src = torch.randint(0, n, (700,))
    dst = torch.randint(0, n, (700,))
    src[5] = dst[5]  # a zero-length edge: v / max(|v|, 1e-12) = 0
    ei = torch.stack([src, dst])
    cutoff = 0.587
    sh, radial = native.edge_geometry(pos.to(dev), ei.to(dev), cutoff, 32)
    vec = pos[src] - pos[dst]
    sh_ref = e3.spherical_harmonics_01(vec)
    rad_ref = e3.soft_one_hot_linspace_gaussian(vec.norm(dim=1), 0.0, cutoff, 32)
    assert sh.shape == (700, 4) and radial.shape == (700, 32)
    assert (sh.cpu() - sh_ref).abs().max().item() <= 2e-6
    assert (radial.cpu() - rad_ref).abs().max().item() <= 2e-6
    assert torch.equal(sh[5].cpu(), torch.tensor([1.0, 0.0, 0.0, 0.0]))


@pytest.mark.parametrize("in0,in1,out0,out1", [(120, 32, 120, 32), (56, 0, 120, 32), (120, 32, 152, 32), (0, 32, 0, 1)])
def test_node_linear_export_matches_oracle(dev, in0, in1, out0, out1):
    """``jamun_node_linear`` vs the oracle's ``o3.Linear`` (flat e3nn weight, 1/sqrt(fan_in)): the shapes of the path — self
    interaction, the initial skip (scalars only in, the 1e half of the output stays zero), the head's Linear, the 32x1e -> 1x1e output."""
    from jamun_amd import native
    from oracle import e3

    torch.manual_seed(1)
    n = 77
    x = torch.randn(n, in0 + 3 * in1)
    numel = in0 * out0 + in1 * out1
    w = torch.randn(numel)
    irr = lambda a, b: [t for t in ((a, 0), (b, 1)) if t[0] > 0]
    ref = e3.linear(x, w, irr(in0, in1), irr(out0, out1))
    out = native.node_linear(x.to(dev), w.to(dev), in0, in1, out0, out1).cpu()
    if in1 == 0 and out1 > 0:  # no path into the 1e outputs: zeros (e3nn leaves outputs without an instruction at zero)
        assert torch.equal(out[:, out0:], torch.zeros(n, 3 * out1))
        ref = torch.cat([ref, torch.zeros(n, 3 * out1)], dim=1) if ref.shape[1] == out0 else ref
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1e-6)


@pytest.mark.parametrize("kind", ["ag4", "chain17x6"])
def test_conv_block_export_matches_oracle(dev, golden_dir, kind):
    """``jamun_build_edges`` + ``jamun_conv_block``: ONE block of the network on caller-owned features — fed with the ORACLE's
    features of the previous block it must reproduce the oracle's features of this block (cached per-layer outputs)."""
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    ref = _golden(golden_dir, f"oracle_forward_{kind}")
    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    smp = model.sampler_for(WalkerBatch.from_molecules(_mols(kind)).to(dev), 0.04)
    with pytest.raises(RuntimeError, match="no edge table"):
        smp.conv_block(1, ref["x0"].to(dev))
    smp.build_edges(ref["y"].to(dev))
    assert torch.equal(smp.debug_read(1).cpu().flatten().long(), ref["deg"])
    x0 = smp.conv_block(0)
    assert (x0.cpu() - ref["x0"]).abs().max().item() <= 2e-5 * ref["x0"].abs().max().item()
    for l in range(1, 6):
        xl = smp.conv_block(l, ref[f"x{l - 1}"].to(dev))
        assert (xl.cpu() - ref[f"x{l}"]).abs().max().item() <= 2e-5 * ref[f"x{l}"].abs().max().item(), l
    with pytest.raises(RuntimeError, match="x_in = NULL"):
        smp.conv_block(0, ref["x0"].to(dev))
    with pytest.raises(RuntimeError, match="layer out of range"):
        smp.conv_block(6, ref["x0"].to(dev))


@pytest.mark.parametrize("log2_scale", [-24, 24])
def test_matrix_formed_conv_block_at_extreme_feature_scales(dev, golden_dir, log2_scale, monkeypatch):
    """The f16x3 kernels scale every operand into the f16 range by powers of two measured on the data (span maxima, in-degrees,
    per-atom maxima in the node update): a ConvBlock on features 2^-24 / 2^24 times their usual size must agree with the general
    fp32 kernel on the same input (the block is not homogeneous — gates — so the comparison is kernel against kernel)."""
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    ref = _golden(golden_dir, "oracle_forward_chain17x6")
    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    batch = WalkerBatch.from_molecules(_mols("chain17x6")).to(dev)
    mf = NativeSampler(model._native, 0.04, batch, dev)
    assert mf.stats()["dg_mode"] == 4
    monkeypatch.setitem(native.TUNING, "no_dg", 1)
    monkeypatch.setitem(native.TUNING, "node_fp32", 1)
    general = NativeSampler(model._native, 0.04, batch, dev)
    assert general.stats()["conv_path"] == 0
    y = ref["y"].to(dev)
    mf.build_edges(y)
    general.build_edges(y)
    for l in (1, 3, 5):
        x_in = (ref[f"x{l - 1}"] * 2.0**log2_scale).to(dev)
        a, b = mf.conv_block(l, x_in).cpu(), general.conv_block(l, x_in).cpu()
        assert torch.isfinite(a).all() and torch.isfinite(b).all()
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item(), (l, log2_scale)


# ---- denoiser forward -----------------------------------------------------------------------------------------------


def _golden(golden_dir, name):
    return {k: torch.tensor(v) for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


def _ckpt(preset):
    from jamun_amd import synth

    return synth.synthetic_checkpoint(output_gain={"strong": 0.5, "stable": 0.05, "mid": 0.1, "g02": 0.2}[preset])


@pytest.mark.parametrize("kind", ["ag4", "chain17x6", "ragged", "dense70", "chig93x2", "chig166x2", "large_mix"])
def test_forward_matches_oracle(dev, golden_dir, kind):
    """One denoiser forward vs the cached CPU-oracle outputs (tests/golden/make_oracle_fixtures.py): edge structure
    exactly, node features after every block, network output, xhat (<= 1e-5 nm RMSD) and score."""
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    ref = _golden(golden_dir, f"oracle_forward_{kind}")
    sigma = 0.04
    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    batch = WalkerBatch.from_molecules(_mols(kind)).to(dev)
    smp = model.sampler_for(batch, sigma)
    y = ref["y"].to(dev)
    x = smp.xhat(y)
    assert torch.equal(smp.debug_read(1).cpu().flatten().long(), ref["deg"])
    assert smp.stats()["n_edges"] == int(ref["n_edges"])
    if kind.startswith("chig"):  # the 32-neighbour cap of torch_cluster (denoiser.py:149) must actually bind at this size
        assert int(ref["deg"].max()) >= 32 and (ref["deg"] >= 32).float().mean() > 0.05
    l = 0
    while f"x{l}" in ref:
        xl, r = smp.debug_read(0, l).cpu(), ref[f"x{l}"]
        err = (xl - r).abs().max().item() / max(r.abs().max().item(), 1e-6)
        assert err < 2e-5, (l, err)
        l += 1
    g = smp.debug_read(2).cpu()
    assert (g - ref["g"]).abs().max().item() < 2e-5 * max(ref["g"].abs().max().item(), 1.0)
    assert rmsd(x, ref["xhat"]) <= RMSD_TOL_NM, rmsd(x, ref["xhat"])
    s = smp.score(y)
    assert rmsd(s, ref["score"]) <= RMSD_TOL_NM / sigma**2  # score = (xhat - y)/sigma^2 amplifies by 625


def test_forward_matches_reference_fixtures(dev, golden_dir):
    """GPU twin of tests/test_oracle.py::test_oracle_matches_reference_fixtures: the HIP forward against `reference_forward_*.npz`, the
    outputs of the reference's own `Denoiser` (tests/golden/make_reference_fixtures.py — the pin kit for SURVEY section 8 rows a7-a18).
    Skips while no such fixture exists (they need e3nn / torch_cluster / torch_scatter, absent from this image).  A fixture whose radius
    graph came from torch_cluster's CPU kd-tree AND has an atom at the 32-neighbour cap is not compared: above the cap the CPU build keeps
    other neighbours than the CUDA build whose rule this path implements; its `_cuda` twin, when present, is."""
    import glob
    import importlib.util

    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    files = sorted(glob.glob(os.path.join(golden_dir, "reference_forward_*.npz")))
    if not files:
        pytest.skip("no tests/golden/reference_forward_*.npz (tests/golden/make_reference_fixtures.py needs e3nn / torch_cluster / torch_scatter)")
    spec = importlib.util.spec_from_file_location("make_oracle_fixtures", os.path.join(golden_dir, "make_oracle_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    cases = {"ag4": ("ag4", None), "chain17x6": ("chain17x6", None), "ragged": ("ragged", None), "chig93x2": ("chig93x2", None),
             "dipep48": ("dipep48", None), "sep_ag4": ("ag4", "sep"), "trained_ragged": ("ragged", "trained"), "nl2_ragged": ("ragged", "nl2")}
    compared = 0
    for f in files:
        name = os.path.basename(f)[len("reference_forward_"):-len(".npz")]
        cuda = name.endswith("_cuda")
        kind, variant = cases[name[:-5] if cuda else name]
        ref = {k: torch.tensor(v) for k, v in np.load(f).items() if v.dtype.kind in "fiu"}
        mols = mk.molecules(kind)
        if variant == "sep":
            ck, sigma = synth.synthetic_checkpoint(output_gain=mk.GAINS["strong"], separable=True), mk.SIGMA
        elif variant is not None:
            ck, sigma = mk.variant_checkpoint(variant), mk.VARIANTS[variant]["sigma"]
        else:
            ck, sigma = synth.synthetic_checkpoint(output_gain=mk.GAINS["strong"]), mk.SIGMA
        bonds_in = max(int(torch.bincount(m["bonds"][1], minlength=m["pos"].shape[0]).max()) if m["bonds"].numel() else 0 for m in mols)
        if not cuda and int(ref["deg"].max()) >= 32 + bonds_in:
            continue
        smp = Denoiser.from_checkpoint_dict(ck).to(dev).sampler_for(WalkerBatch.from_molecules(mols).to(dev), sigma)
        x = smp.xhat(ref["y"].to(dev))
        assert torch.equal(smp.debug_read(1).cpu().flatten().long(), ref["deg"]), name
        l = 0
        while f"x{l}" in ref:
            xl, r = smp.debug_read(0, l).cpu(), ref[f"x{l}"]
            assert (xl - r).abs().max().item() <= 2e-5 * max(r.abs().max().item(), 1e-6), (name, l)
            l += 1
        assert rmsd(x, ref["xhat"]) <= RMSD_TOL_NM, (name, rmsd(x, ref["xhat"]))
        assert rmsd(smp.score(ref["y"].to(dev)), ref["score"]) <= RMSD_TOL_NM / sigma**2, name
        compared += 1
    assert compared > 0


@pytest.mark.parametrize("kind", ["ag4", "chain17x6", "ragged_small", "ragged", "dense70", "chig93x2", "chig166x2"])
def test_conv_kernel_variants_agree(dev, golden_dir, kind, monkeypatch):
    """The conv kernels — jamun_conv_mf (A operand formed on the matrix cores), the four modes of jamun_conv_dg (formed on the vector
    ALUs) and the general k_conv — are independent implementations of the same contraction, and so are the initial projector's
    k_conv_mfi / k_conv_mfx, k_conv_init_v and k_conv: all must meet the oracle (where a cached oracle output exists) and each other
    on the same input.  Kernels are excluded one at a time through jamun_tuning (native.TUNING)."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    if kind == "ragged_small":  # ragged atom counts incl. 1- and 2-atom walkers, tile spans <= 60 atoms
        mols = [synth.random_chain(n, seed=100 + i) for i, n in enumerate([5, 9, 20, 1, 13, 2, 17, 20, 3, 11, 19, 7])]
        ref = None
        torch.manual_seed(5)
        batch = WalkerBatch.from_molecules(mols).to(dev)
        y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    else:
        ref = _golden(golden_dir, f"oracle_forward_{kind}")
        batch = WalkerBatch.from_molecules(_mols(kind)).to(dev)
        y = ref["y"].to(dev)
    monkeypatch.delitem(native.TUNING, "no_dg", raising=False)
    dg = NativeSampler(model._native, 0.04, batch, dev)  # default: the destination-grouped kernels on host-planned tiles
    assert dg.stats()["conv_path"] == 2
    monkeypatch.setitem(native.TUNING, "no_dg", 1)
    general = NativeSampler(model._native, 0.04, batch, dev)
    assert general.stats()["conv_path"] == 0 and general.stats()["init_path"] == 0
    xd, xg = dg.xhat(y), general.xhat(y)
    if ref is not None:
        assert rmsd(xd, ref["xhat"]) <= RMSD_TOL_NM, rmsd(xd, ref["xhat"])
        assert rmsd(xg, ref["xhat"]) <= RMSD_TOL_NM
    assert rmsd(xd, xg) <= RMSD_TOL_NM
    for l in range(6):
        b = general.debug_read(0, l).cpu()
        a = dg.debug_read(0, l).cpu()
        assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-6), l
    monkeypatch.delitem(native.TUNING, "no_dg", raising=False)
    # jamun_conv_dg.hip has four variants, chosen by the span of the tiles: single phase (small molecules), single phase with one
    # Y tile (mid-size spans), two phases with resident source rows, two passes (large molecules).  Switch the chosen one off
    # and the two-phase kernel must give the same features.
    mode = dg.stats()["dg_mode"]
    # Spans up to 62 source rows take jamun_conv_mf.hip (A operand formed on the matrix cores, dg_mode 4).  jamun_conv_dg.hip, the
    # VALU-forming kernel for everything larger, has four variants chosen by the span of the tiles: single phase (2), single phase
    # with one Y tile (3), two phases with resident source rows (0), two passes (large molecules, 1).  Each variant that can take
    # this batch is switched on in turn and must give the same features.
    # Molecules of 63..167 atoms take jamun_conv_ml.hip (dg_mode 5: matrix-core forming for large spans, two passes over the hidden units,
    # block-sparse forming); with it switched off (no_ml) the VALU-forming variants take over.
    expect = {"ag4": 4, "chain17x6": 4, "ragged_small": 4, "ragged": 4, "dense70": 5, "chig93x2": 5, "chig166x2": 5}[kind]
    assert mode == expect, (kind, mode)
    small = kind in ("ag4", "chain17x6", "ragged_small")  # spans within the single-phase budget of jamun_conv_dg.hip (~52 rows)
    variants = {4: [("no_mf",)], 5: [("no_ml",)], 1: [("dg_no_alt",)], 2: [], 3: []}[mode]
    if mode == 5:
        variants.append(("no_ml", "dg_no_sph") if kind == "dense70" else ("no_ml", "dg_no_alt"))
    if mode == 4:
        variants.append(("no_mf", "dg_no_sp") if small else ("no_mf", "dg_no_sph"))
        if small:
            variants.append(("no_mf", "dg_no_sp", "dg_no_sph"))
    if mode == 3:
        variants.append(("dg_no_sph",))
    seen = {mode}
    for envs in variants:
        other = NativeSampler(model._native, 0.04, batch, dev, tuning={e: 1 for e in envs})
        seen.add(other.stats()["dg_mode"])
        assert other.stats()["conv_path"] == 2 and other.stats()["dg_mode"] != mode, (envs, other.stats()["dg_mode"])
        assert rmsd(other.xhat(y), xg) <= RMSD_TOL_NM
        for l in range(6):
            a2, b2 = other.debug_read(0, l).cpu(), general.debug_read(0, l).cpu()
            assert (a2 - b2).abs().max().item() <= 2e-5 * max(b2.abs().max().item(), 1e-6), (envs, l)
    assert seen == {"ag4": {4, 2, 3, 0}, "chain17x6": {4, 2, 3, 0}, "ragged_small": {4, 2, 3, 0}, "ragged": {4, 3, 0}, "dense70": {5, 3, 0},
                    "chig93x2": {5, 1, 0}, "chig166x2": {5, 1, 0}}[kind], seen
    # The initial projector of the default path: on the tiles of k_conv_mf (spans up to 62 rows) k_conv_mfi (up to 32 distinct
    # embedding rows: one-hot selector, init_path 3) or k_conv_mfx (formed from the embedding rows, 4); on the tiles of the dg kernel
    # k_conv_init_v (edge by edge on the vector ALUs, 2); on the large-span tiles of jamun_conv_ml.hip k_conv_mlx (5).  Each is switched off in
    # turn: same features; with all of them off the general kernel takes the layer (0).
    expect_init = {"ag4": 3, "chain17x6": 3, "ragged_small": 4, "ragged": 4, "dense70": 5, "chig93x2": 5, "chig166x2": 5}[kind]
    assert dg.stats()["init_path"] == expect_init, (kind, dg.stats()["init_path"])
    a1 = dg.debug_read(0, 0).cpu()
    if expect_init >= 3:
        no_mfi = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_mfi": 1})
        assert no_mfi.stats()["init_path"] == 2, no_mfi.stats()["init_path"]
        assert rmsd(no_mfi.xhat(y), xd) <= RMSD_TOL_NM
        b1 = no_mfi.debug_read(0, 0).cpu()
        assert (a1 - b1).abs().max().item() <= 2e-5 * max(b1.abs().max().item(), 1e-6)
    no_v = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_mfi": 1, "no_init_v": 1})
    assert no_v.stats()["init_path"] == 0 and no_v.stats()["conv_path"] == 2
    assert rmsd(no_v.xhat(y), xd) <= RMSD_TOL_NM
    b1 = no_v.debug_read(0, 0).cpu()
    assert (a1 - b1).abs().max().item() <= 2e-5 * max(b1.abs().max().item(), 1e-6)


@pytest.mark.parametrize("seed", [0, 7])
def test_tiled_and_general_conv_agree_on_random_ragged_batches(dev, seed, monkeypatch):
    """Fuzz the tile plans of the destination-grouped kernels (multi-molecule tiles, partly filled tiles, source-row blocks of molecules
    above the per-tile budget, 1-atom walkers, tiny batches) against the general kernel: random walker counts and sizes."""
    import random

    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    rng = random.Random(seed)
    for trial in range(6):
        nw = rng.choice([1, 2, 3, 5, 8, 13, 40])
        hi = rng.choice([3, 9, 20, 33, 52, 53, 60, 104, 130])
        mols = [synth.random_chain(rng.randint(1, hi), seed=1000 * trial + i + 17 * seed) for i in range(nw)]
        batch = WalkerBatch.from_molecules(mols).to(dev)
        torch.manual_seed(trial)
        y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
        dg = NativeSampler(model._native, 0.04, batch, dev)
        vf = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_mf": 1, "no_mfi": 1})  # vector-ALU forming kernels on their own tile plans
        general = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_dg": 1})
        assert dg.stats()["conv_path"] == 2 and vf.stats()["conv_path"] == 2 and vf.stats()["dg_mode"] != 4 and general.stats()["conv_path"] == 0
        xd, xf, xg = dg.xhat(y), vf.xhat(y), general.xhat(y)
        assert torch.isfinite(xf).all() and torch.isfinite(xd).all()
        assert rmsd(xf, xg) <= RMSD_TOL_NM, (trial, nw, hi)
        assert rmsd(xd, xg) <= RMSD_TOL_NM, (trial, nw, hi)
        for l in range(6):
            b = general.debug_read(0, l).cpu()
            for other in (vf, dg):
                a = other.debug_read(0, l).cpu()
                assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-6), (trial, l)


@pytest.mark.parametrize("case", ["chain33x4", "tails_1_to_8", "doubled_bonds"])
def test_tail_tiles_of_the_matrix_formed_conv(dev, golden_dir, case, monkeypatch):
    """Tiles of k_conv_mf with few destinations (a 33-atom molecule cuts into 32 + 1) go through k_tail_form / k_tail_contract: formed with
    the hidden unit in the column index, parked, contracted 32 gathered destinations at a time.  Against the oracle (33-atom fixture),
    against the same tiles run as whole tiles (jamun_tuning.no_tail) and against the general kernel; tail sizes 1..8 (every
    columns-per-step variant), edge strides above 32, bonds listed twice in both directions (three edges per pair)."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    ref = None
    if case == "chain33x4":
        mols = _mols("chain33x4")
        ref = _golden(golden_dir, "oracle_forward_chain33x4")
    elif case == "tails_1_to_8":
        mols = [synth.random_chain(n, seed=50 + i) for i, n in enumerate([33, 34, 35, 36, 37, 38, 39, 40, 17, 33, 9, 40, 36])]
    else:
        mols = []
        for i, n in enumerate([33, 35, 34, 33, 40]):
            m = synth.random_chain(n, seed=70 + i)
            b = m["bonds"]
            m["bonds"] = torch.cat([b, b, b.flip(0), b.flip(0)], dim=1)
            mols.append(m)
    batch = WalkerBatch.from_molecules(mols).to(dev)
    if ref is not None:
        y = ref["y"].to(dev)
    else:
        torch.manual_seed(11)
        y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    tails = NativeSampler(model._native, 0.04, batch, dev)
    whole = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_tail": 1})
    general = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_dg": 1})
    assert tails.stats()["dg_mode"] == 4 and whole.stats()["dg_mode"] == 4 and general.stats()["conv_path"] == 0
    assert tails.stats()["n_tail_tiles"] >= 4 and whole.stats()["n_tail_tiles"] == 0
    xt, xw, xg = tails.xhat(y), whole.xhat(y), general.xhat(y)
    assert torch.isfinite(xt).all()
    assert rmsd(xt, xg) <= RMSD_TOL_NM and rmsd(xw, xg) <= RMSD_TOL_NM, (rmsd(xt, xg), rmsd(xw, xg))
    assert torch.equal(tails.xhat(y), xt)  # bit-reproducible
    if ref is not None:
        assert rmsd(xt, ref["xhat"]) <= RMSD_TOL_NM
    # (the initial projector without its matrix-formed kernels keeps the tail tiles on segment lists of its own: k_conv_init_v on all tiles,
    # the hidden layers' tails still through the tail kernels)
    mixed = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_mfi": 1})
    assert mixed.stats()["init_path"] == 2 and mixed.stats()["n_tail_tiles"] >= 4
    assert rmsd(mixed.xhat(y), xg) <= RMSD_TOL_NM
    for l in range(6):
        b = general.debug_read(0, l).cpu()
        for other in (tails, whole, mixed):
            a = other.debug_read(0, l).cpu()
            assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-6), (case, l)


def test_pair_table_with_more_than_four_bonds_into_one_atom(dev):
    """k_geom matches the edges of one ordered (source, destination) pair once per forward (the pair table the matrix-formed kernels read).
    For atoms with up to four bonds it notes the radial twin of each bond while it writes the radial edges; atoms with MORE bonds take a scan
    over their slots.  A hub with six bonded in-edges (all inside the cutoff: six pairs of two edges) and a second hub whose bonds are
    listed twice (three edges per pair) — against the general kernel, which treats every slot as an edge of its own."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    mols = []
    for i, n in enumerate([17, 21, 33]):
        m = synth.random_chain(n, seed=300 + i)
        hub = torch.tensor([[3, 5, 7, 9, 11, 13], [0] * 6])  # six more bonds INTO atom 0
        extra = torch.tensor([[2, 4, 6], [15] * 3])           # three bonds into atom 15, listed twice below
        m["bonds"] = torch.cat([m["bonds"], hub, extra, extra], dim=1)
        mols.append(m)
    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    batch = WalkerBatch.from_molecules(mols * 3).to(dev)
    torch.manual_seed(23)
    y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    fast = NativeSampler(model._native, 0.04, batch, dev)
    general = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_dg": 1})
    assert fast.stats()["dg_mode"] == 4 and general.stats()["conv_path"] == 0
    xf, xg = fast.xhat(y), general.xhat(y)
    assert torch.isfinite(xf).all() and rmsd(xf, xg) <= RMSD_TOL_NM, rmsd(xf, xg)
    for l in range(6):
        a, b = fast.debug_read(0, l).cpu(), general.debug_read(0, l).cpu()
        assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-6), l
    fast.stats()  # (synchronises: a kernel that saw more edges of a pair than an entry holds would raise here)


def test_geometry_of_a_molecule_above_the_lds_budget_of_k_geom(dev):
    """k_geom / k_finalize keep a walker's coordinates in LDS up to 1024 atoms and read global memory above that: a 1100-atom chain beside a
    small molecule — in-degrees against the stand-alone jamun_radius_graph (another kernel) plus the bonds, centred output, and the small
    molecule's x-hat independent of its big neighbour."""
    from jamun_amd import native, synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    big, small = synth.random_chain(1100, seed=7, min_dist=0.2), synth.random_chain(17, seed=8)
    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    batch = WalkerBatch.from_molecules([small, big, small]).to(dev)
    torch.manual_seed(29)
    y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    smp = model.sampler_for(batch, 0.04)
    x = smp.xhat(y)
    assert torch.isfinite(x).all()
    ptr = torch.tensor([0, 17, 1117, 1134], dtype=torch.int32)
    for lo, hi in ((0, 17), (17, 1117), (1117, 1134)):  # x-hat is centred per walker
        assert x[lo:hi].double().mean(0).abs().max().item() < 1e-5
    yc = native.mean_center(y, ptr.to(dev))
    # the sampler's cutoff on unscaled coordinates, fp32 op for op (jamun_sampler_create; denoiser.py:116-136,149): sqrt(r^2 + 6 s^2) / c_in
    f = np.float32
    s2 = f(0.04) * f(0.04)
    c_in = f(1.0) / np.sqrt(f(0.332) + f(6.0) * s2)
    r_cut = float(np.sqrt(f(1.0) + f(6.0) * s2) / c_in)
    _, deg_r = native.radius_graph(yc, r_cut, ptr.to(dev))
    bonded_in = torch.zeros(1134, dtype=torch.int32)
    for off, m in ((0, small), (17, big), (1117, small)):
        bonded_in += torch.bincount(m["bonds"][1] + off, minlength=1134).int()
    assert torch.equal(smp.debug_read(1).cpu().flatten().int(), deg_r.cpu() + bonded_in)
    alone = WalkerBatch.from_molecules([small]).to(dev)
    x_alone = model.xhat(alone.with_pos(y[:17]), 0.04).pos
    assert rmsd(x[:17], x_alone) <= RMSD_TOL_NM / 10


@pytest.mark.parametrize("case", ["chain33x4", "odd_starts_47", "mixed_40_46", "one_too_many"])
def test_short_forming_window_of_the_matrix_formed_conv(dev, golden_dir, case):
    """k_conv_mf<SPD, 3>: when every whole tile's sources lie in the first 48 rows of its 64-row window (one 33-atom molecule per tile — the
    4AA shape) the forming products run three K-steps instead of four (jamun_stats.mf_nks).  The dropped step multiplied zeros, so the result
    must be BIT-identical to the four-step instantiation (jamun_tuning.no_short_k); against the general kernel and the 33-atom oracle fixture
    besides.  47-atom molecules starting at odd atoms need exactly 48 rows; one 48-atom molecule at an odd atom needs 49 and must keep four."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    ref = None
    if case == "chain33x4":
        mols, ref, want = _mols("chain33x4"), _golden(golden_dir, "oracle_forward_chain33x4"), 3
    elif case == "odd_starts_47":
        mols, want = [synth.random_chain(n, seed=90 + i) for i, n in enumerate([47, 47, 47, 45, 47, 46])], 3
    elif case == "mixed_40_46":
        mols, want = [synth.random_chain(n, seed=120 + i) for i, n in enumerate([40, 46, 33, 44, 41, 35, 46, 43])], 3
    else:
        mols, want = [synth.random_chain(n, seed=150 + i) for i, n in enumerate([33, 48, 33, 40])], 4  # 48 atoms from atom 33: rows 32..80
    batch = WalkerBatch.from_molecules(mols).to(dev)
    if ref is not None:
        y = ref["y"].to(dev)
    else:
        torch.manual_seed(13)
        y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    short = NativeSampler(model._native, 0.04, batch, dev)
    full = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_short_k": 1})
    general = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_dg": 1})
    assert short.stats()["dg_mode"] == 4 and short.stats()["mf_nks"] == want, short.stats()
    assert full.stats()["mf_nks"] == 4 and general.stats()["mf_nks"] == 0
    xs, xf, xg = short.xhat(y), full.xhat(y), general.xhat(y)
    assert torch.equal(xs, xf)
    assert rmsd(xs, xg) <= RMSD_TOL_NM, rmsd(xs, xg)
    if ref is not None:
        assert rmsd(xs, ref["xhat"]) <= RMSD_TOL_NM
    for l in range(6):
        assert torch.equal(short.debug_read(0, l), full.debug_read(0, l)), (case, l)
    short.stats()  # (synchronises; a source outside the multiplied rows would have raised JAMUN_ERR_INVALID here)


@pytest.mark.parametrize("case", ["span62", "single62", "odd_start", "double_bonds", "rows100", "many_rows"])
def test_matrix_formed_conv_edge_cases(dev, case, monkeypatch):
    """k_conv_mf / k_conv_mfi (jamun_conv_mf.hip) against the general kernel on the shapes that stress their bookkeeping: a tile span of
    exactly 62 source rows over two molecules; one 62-atom molecule (edge stride above 32: all eight waves build); tiles whose span starts
    at an odd atom (the window of 64 rows starts one atom earlier: 8-byte T loads); every bond listed twice in both directions (a bonded
    pair inside the cutoff then has THREE edges sharing one coefficient entry); more than 64 distinct embedding rows (two selector tiles).
    A 63-atom molecule does not fit the window and must fall back to jamun_conv_dg.hip."""
    import dataclasses

    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    if case == "span62":
        mols = [synth.random_chain(31, seed=s) for s in (1, 2, 3, 4)]
    elif case == "single62":
        mols = [synth.random_chain(62, seed=5), synth.random_chain(62, seed=6)]
    elif case == "odd_start":
        mols = [synth.random_chain(n, seed=10 + i) for i, n in enumerate([1, 17, 3, 21, 5, 19, 17, 7, 25, 9])]
    elif case == "rows100":  # 65..128 distinct embedding rows: four selector tiles in k_conv_mfi
        mols = [synth.random_chain(40, seed=40 + i) for i in range(2)]
        mols[1]["residue_sequence_index"] = mols[1]["residue_sequence_index"] + 8
    elif case == "double_bonds":
        mols = []
        for i, n in enumerate([17, 20, 9, 17]):
            m = synth.random_chain(n, seed=20 + i)
            b = m["bonds"]
            m["bonds"] = torch.cat([b, b, b.flip(0), b.flip(0)], dim=1)  # both directions, each twice
            mols.append(m)
    else:  # many distinct (type, code, residue) rows: residue indices keep growing along the chains (> 128 rows in all)
        mols = [synth.random_chain(60, seed=30 + i) for i in range(3)]
        for i, m in enumerate(mols):
            m["residue_sequence_index"] = m["residue_sequence_index"] + 12 * i
            m["residue_code_index"] = (m["residue_code_index"] + 7 * i) % 20
    batch = WalkerBatch.from_molecules(mols).to(dev)
    torch.manual_seed(3)
    y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    mf = NativeSampler(model._native, 0.04, batch, dev)
    st = mf.stats()
    # (more than 32 distinct embedding rows: the initial projector is formed from the feature rows, k_conv_mfx)
    assert (st["conv_path"], st["dg_mode"], st["init_path"]) == (2, 4, _init_path_of(mols)), st
    if case in ("many_rows", "rows100"):
        assert st["init_path"] == 4
    monkeypatch.setitem(native.TUNING, "no_dg", 1)
    general = NativeSampler(model._native, 0.04, batch, dev)
    monkeypatch.delitem(native.TUNING, "no_dg", raising=False)
    assert general.stats()["conv_path"] == 0
    xm, xg = mf.xhat(y), general.xhat(y)
    assert torch.isfinite(xm).all() and rmsd(xm, xg) <= RMSD_TOL_NM, rmsd(xm, xg)
    assert torch.equal(mf.xhat(y), xm)  # bit-reproducible
    for l in range(6):
        a, b = mf.debug_read(0, l).cpu(), general.debug_read(0, l).cpu()
        assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-6), l
    if case == "single62":  # one atom more: the span no longer fits the K = 64 window
        big = WalkerBatch.from_molecules([synth.random_chain(63, seed=5)]).to(dev)
        assert NativeSampler(model._native, 0.04, big, dev).stats()["dg_mode"] != 4


@pytest.mark.parametrize("case", ["w96", "w128", "w168", "mixed", "odd_starts", "double_bonds", "stretched", "empty_and_single"])
def test_large_span_matrix_formed_conv(dev, case, monkeypatch):
    """jamun_conv_ml.hip (k_conv_ml: hidden layers, k_conv_mlx: initial projector; dg_mode 5 / init_path 5) against the general kernel and the
    VALU-forming kernels it replaces, on what stresses its bookkeeping: each window instantiation (96, 128 and 168 source rows, the last with
    the half block), spans of two molecules of different sizes, windows starting at odd atoms, every bond listed twice in both directions
    (three edges of one pair share a coefficient entry), a stretched chain whose tiles touch few 16-row source blocks (the block-sparse forming
    skips the others), a batch with 1- and 2-atom walkers between large ones.  Bit-reproducible (host-built segment lists, slabs summed in order;
    every split primitive a compiler-visible instruction: jamun_split.h on why inline asm in front of MFMAs is not)."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    window = None
    if case == "w96":
        mols, window = [synth.random_chain(93, seed=1)] * 3 + [synth.random_chain(80, seed=2)], 96
    elif case == "w128":
        mols, window = [synth.random_chain(120, seed=3)] * 2 + [synth.random_chain(128, seed=4)], 128
    elif case == "w168":
        mols, window = [synth.random_chain(166, seed=5), synth.random_chain(167, seed=6), synth.random_chain(150, seed=7)], 168
    elif case == "mixed":
        mols, window = [synth.random_chain(n, seed=50 + i) for i, n in enumerate([63, 70, 9, 101, 64, 33, 120, 77, 17])], 168
    elif case == "odd_starts":
        mols, window = [synth.random_chain(n, seed=60 + i) for i, n in enumerate([1, 93, 3, 95, 7, 65, 5, 91])], 168  # (a span of 95 + 7 + 65 atoms)
    elif case == "double_bonds":
        mols = []
        for i, n in enumerate([70, 93]):
            m = synth.random_chain(n, seed=70 + i)
            b = m["bonds"]
            m["bonds"] = torch.cat([b, b.flip(0)], dim=1)  # every bond a second time, reversed: with both directions present, each ordered pair twice
            mols.append(m)
    elif case == "stretched":
        m = synth.random_chain(160, seed=24)
        pos = m["pos"].clone()
        pos[:, 0] = 0.14 * torch.arange(160, dtype=pos.dtype)
        pos[:, 1:] *= 0.05
        mols, window = [dict(m, pos=pos)] * 2, 168
    else:
        mols = [synth.random_chain(n, seed=80 + i) for i, n in enumerate([1, 2, 100, 1, 64, 2])]
    batch = WalkerBatch.from_molecules(mols).to(dev)
    torch.manual_seed(13)
    y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    ml = NativeSampler(model._native, 0.04, batch, dev)
    x = ml.xhat(y)
    st = ml.stats()
    assert (st["conv_path"], st["dg_mode"], st["init_path"]) == (2, 5, 5), st
    if window is not None:
        assert st["ml_window"] == window, st
    assert torch.isfinite(x).all() and torch.equal(ml.xhat(y), x)
    # the block-sparse forming executes 57 MFMAs per occupied 16-row source block + 186 per (tile, hidden unit): on the stretched chain a tile's
    # sources lie in two or three of the window's eleven blocks
    if case == "stretched":
        dense = 10 * 65 * (57 * 11 + 186) * 32768  # 2 x 5 tiles
        assert 10 * 65 * 186 * 32768 < st["conv_flop_exec_launch"] < 0.55 * dense, (st["conv_flop_exec_launch"], dense)
    valu = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_ml": 1})
    assert valu.stats()["dg_mode"] in (0, 1, 3) and valu.stats()["init_path"] == 2
    monkeypatch.setitem(native.TUNING, "no_dg", 1)
    general = NativeSampler(model._native, 0.04, batch, dev)
    monkeypatch.delitem(native.TUNING, "no_dg", raising=False)
    assert general.stats()["conv_path"] == 0
    xv, xg = valu.xhat(y), general.xhat(y)
    assert rmsd(x, xg) <= RMSD_TOL_NM and rmsd(xv, xg) <= RMSD_TOL_NM, (rmsd(x, xg), rmsd(xv, xg))
    for l in range(6):
        a, b = ml.debug_read(0, l).cpu(), general.debug_read(0, l).cpu()
        assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-6), (case, l)
    # the initial projector alone against k_conv_init_v on the same tiles
    half = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_mfi": 1})
    assert half.stats()["dg_mode"] == 5 and half.stats()["init_path"] == 2
    assert rmsd(half.xhat(y), x) <= RMSD_TOL_NM


@pytest.mark.parametrize("atoms,walkers,dg_mode,init_path", [(17, 64, 4, 3), (33, 64, 4, 3), (57, 32, 4, 4), (70, 16, 5, 5), (93, 8, 5, 5), (166, 4, 5, 5)])
def test_kernel_variants_chosen_for_the_baseline_shapes(dev, atoms, walkers, dg_mode, init_path):
    """BASELINE configs[1..4] shapes (fewer walkers): which variant of the hidden-layer conv kernel (jamun_stats.dg_mode) and of
    the initial projector (init_path) the sampler picks, and that the forward through them is finite and rotation-equivariant."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    batch = WalkerBatch.from_molecules([synth.random_chain(atoms, seed=0)] * walkers).to(dev)
    smp = NativeSampler(model._native, 0.04, batch, dev)
    st = smp.stats()
    assert (st["conv_path"], st["dg_mode"], st["init_path"]) == (2, dg_mode, init_path), st
    torch.manual_seed(11)
    y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    x = smp.xhat(y)
    assert torch.isfinite(x).all()
    assert torch.equal(smp.xhat(y), x)  # fixed summation order (host-built segment lists, slabs summed in order): bit-reproducible
    # rotate every walker by the same proper rotation about its own centre: xhat must rotate with it
    q, _ = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    rot = q.to(torch.float32).to(dev)
    yw = y.view(walkers, atoms, 3)
    c = yw.mean(dim=1, keepdim=True)
    y_rot = ((yw - c) @ rot.T + c).reshape(-1, 3).contiguous()
    x_rot = smp.xhat(y_rot).view(walkers, atoms, 3)
    xw = x.view(walkers, atoms, 3)
    # (compared about the walkers' own centres: with mean_center the output is centred, without it it carries the input centre)
    x_exp = (xw - xw.mean(dim=1, keepdim=True)) @ rot.T
    x_got = x_rot - x_rot.mean(dim=1, keepdim=True)
    assert rmsd(x_got.reshape(-1, 3), x_exp.reshape(-1, 3)) <= 5 * RMSD_TOL_NM


def test_walkers_are_independent_at_baseline_batch_size(dev, golden_dir):
    """BASELINE configs[1] shape (17-atom molecule x 256 walkers = 136 tiles, ~2.5 segments per workgroup, 2-3 partial
    slabs per tile): the denoised coordinates of a walker must not depend on which walkers share its batch.  Walkers 0, 1,
    100 and 255 of the big batch are compared with the same four walkers sampled alone — and, through the 6-walker cached
    oracle case of the same molecule, with the CPU oracle."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    mol = synth.random_chain(17, seed=0)
    big = WalkerBatch.from_molecules([mol] * 256).to(dev)
    torch.manual_seed(3)
    y = big.pos + 0.04 * torch.randn(big.pos.shape).to(dev)
    x_big = model.xhat(big.with_pos(y), 0.04).pos.view(256, 17, 3)
    pick = [0, 1, 100, 255]
    small = WalkerBatch.from_molecules([mol] * len(pick)).to(dev)
    y_small = y.view(256, 17, 3)[pick].reshape(-1, 3)
    x_small = model.xhat(small.with_pos(y_small), 0.04).pos.view(len(pick), 17, 3)
    assert rmsd(x_big[pick].reshape(-1, 3), x_small.reshape(-1, 3)) <= RMSD_TOL_NM / 10
    # tie the big batch to the oracle: put the cached 6-walker oracle input into walkers 5..10 of the big batch
    ref = _golden(golden_dir, "oracle_forward_chain17x6")
    if _mols("chain17x6")[0]["pos"].shape[0] == 17:
        big6 = WalkerBatch.from_molecules(_mols("chain17x6") + [mol] * 250).to(dev)
        y6 = torch.cat([ref["y"].to(dev), y[6 * 17:]])
        x6 = model.xhat(big6.with_pos(y6), 0.04).pos[: 6 * 17]
        assert rmsd(x6, ref["xhat"]) <= RMSD_TOL_NM


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4", "cfg5"])
def test_full_size_batches_of_the_multi_gpu_configs_match_the_oracle(dev, golden_dir, cfg):
    """Per-GPU shapes of BASELINE configs[2..4] at FULL size — 33 atoms x 256 walkers, the ragged 17..57-atom batch of 256
    walkers, 166 atoms x 64 walkers: multi-segment work lists, 2-3 partial slabs per tile, every CU busy.  The cached oracle
    inputs are embedded as walkers of the big batch and must come out within 1e-5 nm of the oracle's denoised coordinates;
    walkers from the first, a middle and the last tile must not depend on the rest of the batch."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    if cfg == "cfg3":
        fixture, mols = "chain33x4", [synth.random_chain(33, seed=0)] * 256
        slots, expect_modes = [0, 1, 2, 3], (4,)
    elif cfg == "cfg4":  # bench.py's cfg4 workload: eight sequences, 32 consecutive walkers each
        kinds = _mols("cfg4kinds")
        fixture, mols = "cfg4kinds", [m for m in kinds for _ in range(32)]
        slots, expect_modes = [32 * i for i in range(8)], (4,)
    else:
        fixture, mols = "chig166x2", [synth.random_chain(166, seed=5)] * 64
        slots, expect_modes = [0, 1], (5,)
    ref = _golden(golden_dir, f"oracle_forward_{fixture}")
    big = WalkerBatch.from_molecules(mols).to(dev)
    smp = NativeSampler(model._native, 0.04, big, dev)
    st = smp.stats()
    assert st["conv_path"] == 2 and st["dg_mode"] in expect_modes, st
    ptr = [0]
    for m in mols:
        ptr.append(ptr[-1] + m["pos"].shape[0])
    torch.manual_seed(3)
    y = big.pos + 0.04 * torch.randn(big.pos.shape).to(dev)
    # the oracle's walkers (same molecules, the oracle's noisy input) at their slots of the big batch
    ofs = 0
    for w in slots:
        n = ptr[w + 1] - ptr[w]
        y[ptr[w] : ptr[w + 1]] = ref["y"][ofs : ofs + n].to(dev)
        ofs += n
    assert ofs == ref["y"].shape[0]
    x = smp.xhat(y)
    assert torch.isfinite(x).all()
    got = torch.cat([x[ptr[w] : ptr[w + 1]] for w in slots])
    assert rmsd(got, ref["xhat"]) <= RMSD_TOL_NM, rmsd(got, ref["xhat"])
    # batch independence across the work list: walkers of the first, a middle and the last tile, sampled alone
    W = len(mols)
    pick = sorted({0, W // 2 - 1, W // 2, W - 1})
    small = WalkerBatch.from_molecules([mols[w] for w in pick]).to(dev)
    y_small = torch.cat([y[ptr[w] : ptr[w + 1]] for w in pick])
    x_small = model.xhat(small.with_pos(y_small), 0.04).pos
    x_pick = torch.cat([x[ptr[w] : ptr[w + 1]] for w in pick])
    assert rmsd(x_pick, x_small) <= RMSD_TOL_NM / 10, rmsd(x_pick, x_small)


@pytest.mark.parametrize("case", ["alone200", "mixed260", "alone400", "sparse300"])
def test_source_row_blocks_of_the_default_conv_kernel(dev, monkeypatch, case):
    """Molecules above the span budget of k_conv_dg's large-molecule mode (~170 atoms) are cut into source ROW BLOCKS: every
    (destination chunk, row block) pair is a tile of its own, edges to sources outside the block are skipped, the blocks'
    results are extra partial slabs, and a tile without any edge writes a zero slab after a workgroup-wide vote.  200-, 260-
    and 400-atom molecules — alone, mixed with small molecules in one batch, and a stretched chain whose far row blocks share
    no edge with a destination chunk — against the general kernel (no tiling by molecule), in the two-pass mode and, with it
    switched off, in the resident two-phase mode."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.native import NativeSampler

    model = Denoiser.from_checkpoint_dict(_ckpt("strong")).to(dev)
    if case == "alone200":
        mols = [synth.random_chain(200, seed=21)]
    elif case == "mixed260":
        mols = [synth.random_chain(9, seed=1), synth.random_chain(260, seed=22), synth.random_chain(17, seed=2), synth.random_chain(40, seed=3)]
    elif case == "alone400":
        mols = [synth.random_chain(400, seed=23)] * 2
    else:  # a nearly straight chain: atoms 0..31 have no neighbour among atoms 200..299 -> edge-less (chunk, block) tiles
        m = synth.random_chain(300, seed=24)
        pos = m["pos"].clone()
        pos[:, 0] = 0.14 * torch.arange(300, dtype=pos.dtype)
        pos[:, 1:] *= 0.05
        m = dict(m, pos=pos)
        mols = [m]
    batch = WalkerBatch.from_molecules(mols).to(dev)
    torch.manual_seed(9)
    y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    monkeypatch.delitem(native.TUNING, "no_dg", raising=False)
    dg = NativeSampler(model._native, 0.04, batch, dev)
    st = dg.stats()
    assert st["conv_path"] == 2 and st["dg_mode"] == 1 and st["dg_row_blocks"] == 1, st
    monkeypatch.setitem(native.TUNING, "dg_no_alt", 1)
    dg0 = NativeSampler(model._native, 0.04, batch, dev)
    monkeypatch.delitem(native.TUNING, "dg_no_alt", raising=False)
    st0 = dg0.stats()
    assert st0["conv_path"] == 2 and st0["dg_mode"] == 0 and st0["dg_row_blocks"] == 1, st0
    monkeypatch.setitem(native.TUNING, "no_dg", 1)
    general = NativeSampler(model._native, 0.04, batch, dev)
    monkeypatch.delitem(native.TUNING, "no_dg", raising=False)
    assert general.stats()["conv_path"] == 0
    xg = general.xhat(y)
    for other in (dg, dg0):
        xo = other.xhat(y)
        assert torch.isfinite(xo).all()
        assert rmsd(xo, xg) <= RMSD_TOL_NM, (case, rmsd(xo, xg))
        for l in range(6):
            a, b = other.debug_read(0, l).cpu(), general.debug_read(0, l).cpu()
            assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-6), (case, l)
    if case == "sparse300":  # the stretched chain really has (chunk, block) tiles without edges
        deg = dg.debug_read(1).cpu().flatten()
        assert deg.max() < 12


@pytest.mark.parametrize("kind", ["ag4", "ragged", "chig93x2"])
def test_separable_conv_forward_matches_oracle(dev, golden_dir, kind):
    """The SeparableConv architecture (``hydra_config/model/arch/e3conv_separable.yaml``; ``e3tools/nn/_conv.py:122-135``,
    ``_tensor_product.py:8-58``): depth-wise "uvu" tensor product + point-wise ``o3.Linear`` in every ConvBlock.  A checkpoint of
    that architecture is recognised at load (factory name / ``tp.lin.weight`` parameters) and dispatched to
    ``jamun_sepconv.hip``; one forward against the cached oracle outputs: per-block features, xhat <= 1e-5 nm, score."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    ref = _golden(golden_dir, f"oracle_forward_sep_{kind}")
    model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint(output_gain=0.5, separable=True)).to(dev)
    assert model.arch["separable_conv"] is True
    smp = model.sampler_for(WalkerBatch.from_molecules(_mols(kind)).to(dev), 0.04)
    y = ref["y"].to(dev)
    x = smp.xhat(y)
    assert torch.equal(smp.debug_read(1).cpu().flatten().long(), ref["deg"])
    l = 0
    while f"x{l}" in ref:
        xl, r = smp.debug_read(0, l).cpu(), ref[f"x{l}"]
        assert (xl - r).abs().max().item() <= 2e-5 * max(r.abs().max().item(), 1e-6), l
        l += 1
    assert rmsd(x, ref["xhat"]) <= RMSD_TOL_NM, rmsd(x, ref["xhat"])
    assert rmsd(smp.score(y), ref["score"]) <= RMSD_TOL_NM / 0.04**2
    assert torch.equal(smp.xhat(y), x)  # fixed summation order
    # the state dict without the factory entry is still recognised by its parameters
    ck = synth.synthetic_checkpoint(output_gain=0.5, separable=True)
    ck["hyper_parameters"]["arch"].pop("hidden_layer_factory")
    m2 = Denoiser.from_checkpoint_dict(ck).to(dev)
    assert m2.arch["separable_conv"] is True
    assert rmsd(m2.sampler_for(WalkerBatch.from_molecules(_mols(kind)).to(dev), 0.04).xhat(y), ref["xhat"]) <= RMSD_TOL_NM


def test_forward_matches_live_oracle(dev, ckpt):
    """Same comparison with the oracle run live on this box (small case), so the cache is not the only witness."""
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from oracle import denoiser as od

    mols = _mols("ag4")
    topo, p, hp = _oracle_setup(mols, ckpt)
    torch.manual_seed(11)
    y = topo["pos"] + 0.04 * torch.randn_like(topo["pos"])
    x_ref = od.xhat(y, topo, 0.04, p, hp)
    model = Denoiser.from_checkpoint_dict(ckpt).to(dev)
    x = model.xhat(WalkerBatch.from_molecules(mols).to(dev).with_pos(y.to(dev)), 0.04).pos
    assert rmsd(x, x_ref) <= RMSD_TOL_NM


def test_forward_zero_gain_closed_form(dev):
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    ck = synth.synthetic_checkpoint(output_gain=0.0)
    mols = _mols("ag4")
    batch = WalkerBatch.from_molecules(mols)
    torch.manual_seed(0)
    y = batch.pos + 0.04 * torch.randn_like(batch.pos)
    model = Denoiser.from_checkpoint_dict(ck).to(dev)
    x = model.xhat(batch.to(dev).with_pos(y.to(dev)), 0.04).pos.cpu()
    A, B = 0.332, 6 * 0.04**2
    yc = y - torch.stack([y[batch.batch == g].mean(0) for g in range(4)])[batch.batch]
    xc = A / (A + B) * yc
    xc = xc - torch.stack([xc[batch.batch == g].mean(0) for g in range(4)])[batch.batch]
    assert (x - xc).abs().max().item() < 1e-6


def test_forward_equivariance_and_walker_permutation(dev, ckpt):
    from scipy.spatial.transform import Rotation

    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    model = Denoiser.from_checkpoint_dict(ckpt).to(dev)
    mols = _mols("ragged")
    batch = WalkerBatch.from_molecules(mols).to(dev)
    torch.manual_seed(3)
    y = (batch.pos.cpu() + 0.04 * torch.randn(batch.num_nodes, 3)).to(dev)
    x = model.xhat(batch.with_pos(y), 0.04).pos
    R = torch.tensor(Rotation.random(random_state=5).as_matrix(), dtype=torch.float32, device=dev)
    x2 = model.xhat(batch.with_pos(y @ R.T + torch.tensor([0.5, -1.0, 0.25], device=dev)), 0.04).pos
    assert (x2 - x @ R.T).abs().max().item() < 2e-6
    # reversing the order of walkers permutes the outputs
    rev = WalkerBatch.from_molecules(mols[::-1]).to(dev)
    ptr = batch.ptr.tolist()
    y_rev = torch.cat([y[ptr[i] : ptr[i + 1]] for i in reversed(range(len(mols)))])
    x_rev = model.xhat(rev.with_pos(y_rev), 0.04).pos
    x_back = torch.cat([x_rev[rev.ptr[len(mols) - 1 - i] : rev.ptr[len(mols) - i]] for i in range(len(mols))])
    assert (x_back - x).abs().max().item() < 2e-6


# ---- integrators ------------------------------------------------------------------------------------------------------


def _cpu_score(d):
    mu = torch.tensor(d["mu"])
    s, a, b = [float(v) for v in d["score_params"]]
    return lambda y: (-(y.cpu() - mu) / (s * s) + a * torch.sin(b * y.cpu())).to(y.device)


@pytest.mark.parametrize("name", ["baoab_default", "baoab_clip_mass", "baoab_noclip_notraj"])
def test_baoab_update_kernels_match_reference_golden(dev, golden_dir, name):
    """State-update kernels vs vectors produced by the reference's own baoab(); the score is evaluated on the CPU
    exactly as in the fixture, so the comparison is bit-exact."""
    from jamun_amd.sampling import BAOAB

    d = np.load(os.path.join(golden_dir, name + ".npz"))
    kw = json.load(open(os.path.join(golden_dir, name + ".json")))["kwargs"]
    noise = torch.tensor(d["noise"])
    v_init = kw.pop("v_init")
    # reference call order: v0 draw (if gaussian) comes first, then one draw per step
    if v_init == "gaussian":
        v0 = (kw["M"] ** -1) ** 0.5 * noise[0]
        step_noise = noise[1:]
    else:
        v0 = torch.zeros_like(noise[0])
        step_noise = noise
    import jamun_amd.sampling as S

    params = S.native.make_mcmc_params(kw["steps"], kw["delta"], kw["friction"], kw["M"], kw["inverse_temperature"], kw["score_fn_clip"],
                                       kw.get("save_every_n_steps", 1), kw.get("burn_in_steps", 0))
    y = torch.tensor(d["y0"]).to(dev).clone()
    v = v0.to(dev).clone()
    y_traj, score_traj, _ = S._python_walk("baoab", y, v, _cpu_score(d), params, step_noise[: kw["steps"] - 1].to(dev), 0, kw["save_trajectory"])
    assert torch.equal(y.cpu(), torch.tensor(d["y"]))
    assert torch.equal(v.cpu(), torch.tensor(d["v"]))
    if "y_traj" in d:
        assert torch.equal(y_traj.cpu(), torch.tensor(d["y_traj"]))
    assert torch.equal(score_traj.cpu(), torch.tensor(d["score_traj"]))


@pytest.mark.parametrize("name", ["aboba_default", "aboba_clip_mass"])
def test_aboba_update_kernels_match_reference_golden(dev, golden_dir, name):
    """k_aboba_a / k_aboba_b vs vectors produced by the reference's own aboba() (functional/_splitting.py:44-109): mass 2,
    clip 3, beta 0.8, burn-in 4, save_every 2 in the second case; T-1 score frames taken at the half-step positions.
    The score is evaluated on the CPU exactly as in the fixture, so the comparison is bit-exact."""
    import jamun_amd.sampling as S

    d = np.load(os.path.join(golden_dir, name + ".npz"))
    kw = json.load(open(os.path.join(golden_dir, name + ".json")))["kwargs"]
    noise = torch.tensor(d["noise"])
    if kw["v_init"] == "gaussian":  # reference call order: the v0 draw comes first, then one draw per step
        v0, step_noise = (kw["M"] ** -1) ** 0.5 * noise[0], noise[1:]
    else:
        v0, step_noise = torch.zeros_like(noise[0]), noise
    params = S.native.make_mcmc_params(kw["steps"], kw["delta"], kw["friction"], kw["M"], kw["inverse_temperature"], kw["score_fn_clip"],
                                       kw.get("save_every_n_steps", 1), kw.get("burn_in_steps", 0))
    y = torch.tensor(d["y0"]).to(dev).clone()
    v = v0.to(dev).clone()
    y_traj, score_traj, _ = S._python_walk("aboba", y, v, _cpu_score(d), params, step_noise[: kw["steps"] - 1].to(dev), 0, True)
    assert torch.equal(y.cpu(), torch.tensor(d["y"]))
    assert torch.equal(v.cpu(), torch.tensor(d["v"]))
    assert torch.equal(y_traj.cpu(), torch.tensor(d["y_traj"]))
    assert torch.equal(score_traj.cpu(), torch.tensor(d["score_traj"]))
    # and through the dataclass front end (ABOBA.__call__ -> _run_walk -> _python_walk), v_init given as a tensor
    mc = S.ABOBA(**{k: v_ for k, v_ in kw.items() if k != "v_init"}, rng="torch_cpu")
    torch.manual_seed(0)
    import unittest.mock as um

    with um.patch.object(torch, "randn", side_effect=[n_ for n_ in step_noise[: kw["steps"] - 1]]):
        y2, v2, yt2, st2 = mc(torch.tensor(d["y0"]).to(dev), _cpu_score(d), v_init=v0.to(dev))
    assert torch.equal(y2.cpu(), torch.tensor(d["y"])) and torch.equal(st2.cpu(), torch.tensor(d["score_traj"]))


def test_fused_baoab_without_trajectory_writes_one_score_frame(dev):
    """save_trajectory=False is the dataclass default: the reference keeps only the initial score then
    (functional/_splitting.py:155,168-170) and NativeSampler.walk allocates score_traj as [1, n, 3].  The fused walk must not
    write past it: guard tensors on both sides of a hand-laid buffer stay untouched, and the one frame is score(y0)."""
    from jamun_amd import _lib, native
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    import ctypes as C

    model = Denoiser.from_checkpoint_dict(_ckpt("stable")).to(dev)
    batch = WalkerBatch.from_molecules(_mols("ag4")).to(dev)
    smp = model.sampler_for(batch, 0.04)
    n = batch.num_nodes
    torch.manual_seed(1)
    y0 = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    steps = 12
    params = native.make_mcmc_params(steps, 0.04, 1.0, 1.0, 1.0, 100.0)
    arena = torch.full((1 + 1 + steps + 1, n, 3), 7.25, device=dev)  # [guard | score frame | room an overrun would hit | guard]
    y, v = y0.clone(), torch.zeros_like(y0)
    with torch.cuda.device(dev):
        _lib.check(smp._lib.jamun_walk_baoab(smp._h, y.data_ptr(), v.data_ptr(), C.byref(params), None, C.c_uint64(5), None,
                                             arena[1].data_ptr(), None, None, int(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(arena[0], torch.full_like(arena[0], 7.25)) and torch.equal(arena[2:], torch.full_like(arena[2:], 7.25))
    assert rmsd(arena[1], smp.score(y0)) == 0.0
    # the front end: BAOAB()(y, NativeScore) with the dataclass defaults returns y_traj None and ONE score frame
    from jamun_amd.sampling import BAOAB, ModelSamplingWrapper, NativeScore

    wrapped = ModelSamplingWrapper(model, batch, 0.04)
    torch.manual_seed(3)
    y1, v1, yt, st = BAOAB(steps=steps, delta=0.04)(y0, NativeScore(wrapped, 0.04))
    assert yt is None and st.shape == (1, n, 3) and torch.isfinite(y1).all() and not torch.equal(y1, y0)


def test_fused_integrator_halves_equal_standalone_update_kernels(dev):
    """The BAOAB halves fused into the first / last kernel of the forward (k_geom, k_finalize) against the stand-alone
    k_baoab_pre / k_baoab_post driven step by step around the same native score — the kernels that are pinned bit-exactly to
    the reference's baoab() above.  Mass 2, a clip that binds, inverse temperature 0.8, burn-in and save_every > 1: every
    output must be bit-identical."""
    import jamun_amd.sampling as S
    from jamun_amd import native
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    model = Denoiser.from_checkpoint_dict(_ckpt("stable")).to(dev)
    batch = WalkerBatch.from_molecules(_mols("ragged")).to(dev)
    smp = model.sampler_for(batch, 0.04)
    steps = 9
    params = native.make_mcmc_params(steps, 0.05, 0.7, 2.0, 0.8, 3.0, save_every_n_steps=2, burn_in_steps=3)
    torch.manual_seed(4)
    y0 = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
    v0 = torch.randn(batch.pos.shape).to(dev)
    noise = torch.randn(steps - 1, batch.num_nodes, 3).to(dev)
    ya, va = y0.clone(), v0.clone()
    yt_a, st_a, xt_a, xh_a = smp.walk("baoab", ya, va, params, noise, 0, True)
    yb, vb = y0.clone(), v0.clone()
    yt_b, st_b, _ = S._python_walk("baoab", yb, vb, lambda t: smp.score(t), params, noise, 0, True)
    assert torch.equal(ya, yb) and torch.equal(va, vb)
    assert torch.equal(yt_a, yt_b) and torch.equal(st_a, st_b) and yt_a.shape[0] == 3 and st_a.shape[0] == 4
    assert float((st_a[0].norm(dim=-1) > 3.0).float().mean()) > 0.5  # the clip actually binds
    assert torch.equal(xh_a, smp.xhat(ya))  # the jump of the final state comes out of the same forward
    # in-kernel Philox noise: fused walk == the same walk on a fresh sampler (determinism of the fused pre-update)
    yc1, vc1 = y0.clone(), v0.clone()
    r1 = smp.walk("baoab", yc1, vc1, params, None, 77, True)
    yc2, vc2 = y0.clone(), v0.clone()
    r2 = smp.walk("baoab", yc2, vc2, params, None, 77, True)
    assert torch.equal(yc1, yc2) and torch.equal(r1[0], r2[0]) and not torch.equal(yc1, ya)


@pytest.mark.parametrize("case,integrator,kind,preset", [
    ("oracle_walk_baoab_ag4_50", "baoab", "ag4", "stable"),  # cfg1 of BASELINE.json: AG dipeptide, 4 walkers x 50 steps
    ("oracle_walk_baoab_ag4_50_mid", "baoab", "ag4", "mid"),  # the same with twice the output gain (still contractive, see test_oracle)
    ("oracle_walk_baoab_ragged_12", "baoab", "ragged", "stable"),
    ("oracle_walk_aboba_ag4_20", "aboba", "ag4", "stable"),
    ("oracle_walk_baoab_chig93_6", "baoab", "chig93x2", "stable"),  # configs[4] size: 93 heavy atoms, neighbour cap active
    ("oracle_walk_baoab_chig166_4", "baoab", "chig166x2", "stable"),  # configs[4] with hydrogens: 166 atoms (two-pass conv kernel)
])
def test_fused_walk_matches_oracle(dev, golden_dir, case, integrator, kind, preset):
    """Fused native walk vs the oracle's walk_jump on the identical noise stream: every saved frame's denoised
    coordinates within 1e-5 nm RMSD (contractive checkpoint)."""
    from jamun_amd import native
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    ref = _golden(golden_dir, case)
    sigma = 0.04
    noise = ref["noise"]  # y0 draw, v0 draw, steps-1 step draws
    steps = noise.shape[0] - 1
    model = Denoiser.from_checkpoint_dict(_ckpt(preset)).to(dev)
    ns = model.sampler_for(WalkerBatch.from_molecules(_mols(kind)).to(dev), sigma)
    params = native.make_mcmc_params(steps, 0.04, 1.0, 1.0, 1.0, 100.0)
    y, v = ref["y0"].to(dev).clone(), noise[1].to(dev).clone()
    y_traj, score_traj, xhat_traj, xhat = ns.walk(integrator, y, v, params, noise[2 : 2 + steps - 1].to(dev).contiguous(), 0, True)
    T = ref["y_traj"].shape[0]
    assert y_traj.shape[0] == T and xhat_traj.shape[0] == T and score_traj.shape[0] == ref["score_traj"].shape[0]
    worst = max(rmsd(xhat_traj[t], ref["xhat_traj"][t]) for t in range(T))
    assert worst <= RMSD_TOL_NM, worst
    assert rmsd(xhat, ref["xhat"]) <= RMSD_TOL_NM
    assert rmsd(y, ref["y"]) <= RMSD_TOL_NM
    assert max(rmsd(y_traj[t], ref["y_traj"][t]) for t in range(T)) <= RMSD_TOL_NM
    assert rmsd(v, ref["v"]) <= 1e-3
    assert rmsd(score_traj[-1], ref["score_traj"][-1]) <= RMSD_TOL_NM / sigma**2


@pytest.mark.parametrize("case,preset,n_tight", [("oracle_walk_baoab_ag4_20_strong", "strong", 8), ("oracle_walk_baoab_ag4_50_g02", "g02", 10)])
def test_fused_walk_chaotic_checkpoint_tracks_oracle_early(dev, golden_dir, case, preset, n_tight):
    """With the strongly non-linear random checkpoint the dynamics amplifies ANY fp32 rounding difference by ~4x
    every 5 steps (the fp32 oracle itself leaves the fp64 oracle's trajectory at that rate),
    so the 1e-5 nm bar is asserted on the first frames only and the growth is bounded afterwards."""
    from jamun_amd import native
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    ref = _golden(golden_dir, case)
    noise = ref["noise"]
    steps = min(noise.shape[0] - 1, 24)
    model = Denoiser.from_checkpoint_dict(_ckpt(preset)).to(dev)
    ns = model.sampler_for(WalkerBatch.from_molecules(_mols("ag4")).to(dev), 0.04)
    params = native.make_mcmc_params(steps, 0.04, 1.0, 1.0, 1.0, 100.0)
    y, v = ref["y0"].to(dev).clone(), noise[1].to(dev).clone()
    _, _, xhat_traj, _ = ns.walk("baoab", y, v, params, noise[2 : 2 + steps - 1].to(dev).contiguous(), 0, True)
    errs = [rmsd(xhat_traj[t], ref["xhat_traj"][t]) for t in range(steps)]
    assert max(errs[:n_tight]) <= RMSD_TOL_NM, errs[:n_tight]
    assert max(errs) <= 1e-3, errs


def test_sampler_protocol_and_rng_modes(dev, golden_dir):
    """Sampler.sample hook order / output dict contract (sampling/_sampler.py:53-98, sampling_wrapper.py:49-83),
    reproducibility under torch.manual_seed for both RNG modes, and the torch_cpu mode against the oracle host loop."""
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.sampling import BAOAB, Sampler, SingleMeasurementSampler
    mols = _mols("ag4")
    ckpt = _ckpt("stable")
    sigma, steps = 0.04, 8
    kw = dict(steps=steps, delta=0.04, friction=1.0, M=1.0, inverse_temperature=1.0, score_fn_clip=100.0, save_trajectory=True)

    class Rec:
        def __init__(self):
            self.events, self.samples = [], []

        def on_sample_start(self, sampler):
            self.events.append("start")

        def on_after_sample_batch(self, sample, sampler):
            self.events.append(("batch", sampler.global_step))
            self.samples.append(sample)

        def on_sample_end(self, sampler):
            self.events.append("end")

    def run(rng, seed, continue_chain=True):
        rec = Rec()
        model = Denoiser.from_checkpoint_dict(ckpt)
        sampler = Sampler(devices=1, callbacks=[rec], rng=rng)
        bs = SingleMeasurementSampler(mcmc=BAOAB(**kw, cpu_offload=True, verbose=False, v_init="zero"), sigma=sigma)
        torch.manual_seed(seed)
        sampler.sample(model=model, batch_sampler=bs, init_graphs=WalkerBatch.from_molecules(mols), num_batches=2, continue_chain=continue_chain)
        return rec

    for rng in ("philox", "torch_cpu"):
        a, b, c = run(rng, 42), run(rng, 42), run(rng, 43)
        assert a.events == ["start", ("batch", 0), ("batch", 1), "end"]
        assert len(a.samples[0]) == 4
        s0 = a.samples[0][0]
        assert set(["xhat", "y", "v", "sample", "xhat_traj", "y_traj", "score_traj", "pos", "atom_type_index"]) <= set(s0)
        assert "t_traj" not in s0  # 1-D values are skipped by unbatch_samples
        assert s0["xhat_traj"].shape == (10, steps, 3) and s0["y"].shape == (10, 3)
        for k in ("xhat_traj", "y_traj"):
            assert torch.equal(a.samples[1][2][k].cpu(), b.samples[1][2][k].cpu())
            assert not torch.equal(a.samples[1][2][k].cpu(), c.samples[1][2][k].cpu())
        # continue_chain: last frame of batch 0 is frame 0 of batch 1
        assert torch.equal(a.samples[0][1]["y_traj"][:, -1].cpu(), a.samples[1][1]["y_traj"][:, 0].cpu())

    # torch_cpu mode == the reference host loop with torch.manual_seed(42): y0 draw, v0 draw, one draw per step
    ptr = WalkerBatch.from_molecules(mols).ptr
    for cc in (True, False):
        ref = _golden(golden_dir, f"oracle_sampler_cc{int(cc)}")
        rec = run("torch_cpu", 42, continue_chain=cc)
        for bi in range(2):
            per_walker = ref[f"xhat_traj_{bi}"].permute(1, 0, 2)
            for w in range(4):
                assert rmsd(rec.samples[bi][w]["xhat_traj"], per_walker[ptr[w] : ptr[w + 1]]) <= RMSD_TOL_NM


def test_missing_tensor_and_bad_config_fail_loudly(dev, ckpt):
    from jamun_amd.model import Denoiser

    bad = {"state_dict": {k: v for k, v in ckpt["state_dict"].items() if "layers.2.gated_conv.self_interaction" not in k}, "hyper_parameters": ckpt["hyper_parameters"]}
    from jamun_amd.data import WalkerBatch

    model = Denoiser.from_checkpoint_dict(bad).to(dev)
    batch = WalkerBatch.from_molecules(_mols("ag4")).to(dev)
    with pytest.raises(RuntimeError, match="missing checkpoint tensor"):
        model.score(batch, 0.04)
    with pytest.raises(RuntimeError):
        Denoiser.from_checkpoint_dict(ckpt).score(WalkerBatch.from_molecules(_mols("ag4")), 0.04)  # CPU tensors: no CPU path


@pytest.mark.parametrize("shape", ["17x256", "33x128", "ragged"])
def test_long_walks_are_bit_reproducible(dev, shape):
    """Soak test against intermittent hazards of the hand-scheduled kernels (a spill store under a stale exec mask, packed FMAs beside
    f16 MFMAs — both were met during development and showed up as ONE wrong row in about every second forward): the same walk of a
    few hundred forwards, twice, from the same state and seed, must agree bit for bit in every saved frame."""
    from jamun_amd import native, synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    if shape == "17x256":
        mols, steps = [synth.random_chain(17, seed=0)] * 256, 300
    elif shape == "33x128":
        mols, steps = [synth.random_chain(33, seed=0)] * 128, 80
    else:
        mols, steps = [synth.random_chain(n, seed=n) for n in (17, 57, 23, 41, 30, 19, 52, 36)] * 16, 80
    model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint(output_gain=0.05)).to(dev)
    batch = WalkerBatch.from_molecules(mols).to(dev)
    smp = model.sampler_for(batch, 0.04)
    assert smp.stats()["dg_mode"] == 4 and smp.stats()["init_path"] == _init_path_of(mols), smp.stats()
    assert (shape == "ragged") == (smp.stats()["init_path"] == 4)  # (the ragged batch has more than 32 distinct embedding rows)
    torch.manual_seed(0)
    y0 = batch.pos + 0.04 * torch.randn_like(batch.pos)
    params = native.make_mcmc_params(steps, delta=0.04, friction=1.0, M=1.0, inverse_temperature=1.0, score_fn_clip=100.0)
    runs = []
    for _ in range(2):
        y, v = y0.clone(), torch.zeros_like(y0)
        y_traj, score_traj, xhat_traj, _ = smp.walk("baoab", y, v, params, None, seed=77, save_trajectory=True)
        torch.cuda.synchronize()
        runs.append((xhat_traj.clone(), y_traj.clone(), score_traj.clone()))
    assert torch.isfinite(runs[0][0]).all()
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)
    # the last kernel of an iteration and the first of the next as ONE launch (k_finalize_geom, the default) or as two (k_finalize, k_geom):
    # the same arithmetic in the same order, so every frame, the final state and the host-noise variant agree bit for bit
    from jamun_amd.native import NativeSampler

    two = NativeSampler(model._native, 0.04, batch, dev, tuning={"no_fuse_geom": 1})
    y, v = y0.clone(), torch.zeros_like(y0)
    y_traj, score_traj, xhat_traj, xh = two.walk("baoab", y, v, params, None, seed=77, save_trajectory=True)
    for a, b in zip(runs[0], (xhat_traj, y_traj, score_traj)):
        assert torch.equal(a, b)
    short = native.make_mcmc_params(7, delta=0.04, friction=1.0, M=1.0, inverse_temperature=1.0, score_fn_clip=100.0)
    noise = torch.randn(6, *y0.shape, device=dev)
    outs = []
    for s_ in (smp, two):
        y, v = y0.clone(), torch.ones_like(y0) * 0.1
        o = s_.walk("baoab", y, v, short, noise, 0, True)
        outs.append((o[0].clone(), o[1].clone(), o[2].clone(), o[3].clone(), y.clone(), v.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_a_twenty_thousand_step_batch_through_the_sampler_is_bit_reproducible(dev):
    """The reference's shipped batch length (num_sampling_steps_per_batch = 20 000: configs/experiment/sample_uncapped_2AA.yaml:16-17) at the
    BASELINE configs[1] shape, through Sampler.sample / SingleMeasurementSampler / BAOAB as jamun_sample runs it: 3 x 1 GB of frames per batch,
    per-walker [n, T, 3] views for the callbacks.  Twice from the same seed: every frame of every walker bit-identical, the last frame finite."""
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd.sampling import BAOAB, Sampler, SingleMeasurementSampler

    steps = 20000
    mols = [dict(synth.random_chain(17, seed=0), dataset_label="m")] * 256
    batch = WalkerBatch.from_molecules(mols)
    model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint(output_gain=0.05)).to(dev)

    class Keep:
        def __init__(self):
            self.first, self.last, self.frames = None, None, 0

        def on_after_sample_batch(self, sample, sampler):
            assert len(sample) == 256 and tuple(sample[0]["xhat_traj"].shape) == (17, steps, 3)
            self.frames = sample[0]["xhat_traj"].shape[1]
            # (checksums on the device: the batch's frames are views of one [T, N, 3] tensor)
            full = torch.stack([s["xhat_traj"] for s in sample[:4]])
            self.first = full.clone()
            self.last = torch.stack([s["xhat_traj"][:, -1] for s in sample]).clone()
            self.sum = torch.stack([s["y_traj"].double().sum() for s in sample]).clone()

    outs = []
    for _ in range(2):
        keep = Keep()
        mcmc = BAOAB(steps=steps, save_trajectory=True, save_every_n_steps=1, v_init="gaussian", delta=0.04, friction=1.0, M=1.0,
                     inverse_temperature=1.0, score_fn_clip=100.0)
        torch.manual_seed(123)
        Sampler(callbacks=[keep]).sample(model, SingleMeasurementSampler(mcmc=mcmc, sigma=0.04), num_batches=1, init_graphs=batch)
        torch.cuda.synchronize()
        outs.append(keep)
    assert outs[0].frames == steps and torch.isfinite(outs[0].last).all()
    assert torch.equal(outs[0].first, outs[1].first) and torch.equal(outs[0].last, outs[1].last) and torch.equal(outs[0].sum, outs[1].sum)
