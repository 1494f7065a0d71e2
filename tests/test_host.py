"""CPU tests of the host-side logic: config composition, PDB reader, checkpoint lookup, sharding and the N>1 gather
(world_size-2 gloo processes)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pdb_round_trip_matches_hand_built_ag_dipeptide(tmp_path):
    from jamun_amd import pdb, synth

    mol = synth.ag_dipeptide()
    mol = dict(mol, elements=["N", "C", "C", "C", "O", "N", "C", "C", "O", "O"], residue_ids=[1] * 5 + [2] * 5)
    path = str(tmp_path / "uncapped_AG.pdb")
    pdb.write_pdb(path, mol, mol["pos"][None])
    # add hydrogens and a water that must be dropped ("protein and not type H", data/_mdtraj.py:60)
    lines = open(path).read().splitlines()
    extra = ["ATOM     11  H1  ALA A   1       0.500   0.900   0.100  1.00  0.00           H",
             "HETATM   12  O   HOH B   3       9.000   9.000   9.000  1.00  0.00           O"]
    lines = lines[:-2] + extra + lines[-2:]
    open(path, "w").write("\n".join(lines) + "\n")
    got = pdb.read_pdb(path)
    for k in ("atom_type_index", "atom_code_index", "residue_code_index", "residue_sequence_index"):
        assert torch.equal(got[k], mol[k]), k
    assert torch.allclose(got["pos"], mol["pos"], atol=1e-4)
    want = {tuple(b) for b in mol["bonds"].T.tolist()}
    have = {tuple(b) for b in got["bonds"].T.tolist()}
    assert want == have and got["bonds"].shape[1] == 9
    assert all(a < b for a, b in got["bonds"].T.tolist())  # lower index first, each bond once
    ds = pdb.create_dataset_from_pdbs([path])
    assert ds[0].label() == "uncapped_AG" and len(ds[0]) == 1


def test_config_composition_builtin_and_reference_style_overlay(tmp_path):
    from jamun_amd import cmdline
    from jamun_amd import config as C

    cfg = cmdline.compose(["--config-dir=" + os.path.join(ROOT, "configs"), "experiment=sample_custom", "++init_pdbs=[a.pdb]", "++checkpoint_dir=ck",
                           "num_sampling_steps_per_batch=50", "repeat_init_samples=4", "++trainer.accelerator=gpu"], cwd=str(tmp_path))
    r = C.resolve(cfg)
    assert r["batch_sampler"]["mcmc"]["steps"] == 50 and r["batch_sampler"]["sigma"] == 0.04
    assert r["batch_sampler"]["mcmc"]["_target_"] == "jamun.sampling.mcmc.BAOAB"
    assert r["init_pdbs"] == [str(tmp_path / "a.pdb")] and r["checkpoint_dir"] == str(tmp_path / "ck")
    assert r["repeat_init_samples"] == 4 and r["sampler"]["devices"] == 1 and r["trainer"]["accelerator"] == "gpu"
    assert r["paths"]["run_path"].startswith("./outputs/sample/dev/runs/")
    # an overlay in the reference's style: unknown keys, ??? values, group override by command line
    d = tmp_path / "cfgs" / "experiment"
    d.mkdir(parents=True)
    (d / "mine.yaml").write_text("# @package _global_\nsigma: 0.1\ndelta: ${sigma}\nfriction: 1.0\nM: 1.0\ninverse_temperature: 1.0\nscore_fn_clip: null\n"
                                 "num_sampling_steps_per_batch: 7\nnum_init_samples_per_dataset: 1\ninit_pdbs: ???\ncheckpoint_dir: x\n"
                                 "wandb_train_run_path: some/run\ninit_datasets:\n  _target_: jamun.data.create_dataset_from_pdbs\n  pdbfiles: ${init_pdbs}\n")
    cfg = cmdline.compose(["--config-dir", str(tmp_path / "cfgs"), "experiment=mine", "batch_sampler/mcmc=aboba", "wandb_train_run_path=null"], cwd=str(tmp_path))
    with pytest.raises(ValueError, match="Missing mandatory value"):
        C.resolve(cfg)
    cfg["init_pdbs"] = ["/x/y.pdb"]
    r = C.resolve(cfg)
    assert r["batch_sampler"]["mcmc"]["_target_"] == "jamun.sampling.mcmc.ABOBA" and r["batch_sampler"]["mcmc"]["delta"] == 0.1
    assert r["batch_sampler"]["mcmc"]["score_fn_clip"] is None and r["wandb_train_run_path"] is None
    with pytest.raises(KeyError):
        cmdline.compose(["no_such_key=1"], cwd=str(tmp_path))


def test_instantiate_maps_reference_targets():
    from jamun_amd import config as C
    from jamun_amd.sampling import BAOAB, SingleMeasurementSampler

    bs = C.instantiate({"_target_": "jamun.sampling.walkjump.SingleMeasurementSampler", "sigma": 0.04,
                        "mcmc": {"_target_": "jamun.sampling.mcmc.BAOAB", "steps": 5, "delta": 0.04, "v_init": "zero"}})
    assert isinstance(bs, SingleMeasurementSampler) and isinstance(bs.mcmc, BAOAB) and bs.mcmc.steps == 5
    with pytest.raises(RuntimeError, match="not in"):
        BAOAB(v_init="nope")
    p = C.instantiate({"_target_": "jamun.sampling.mcmc.BAOAB", "_partial_": True, "steps": 3})
    assert p().steps == 3


def test_find_checkpoint_rules(tmp_path):
    from jamun_amd.checkpoint import find_checkpoint

    for f in ["epoch=3-step=10.ckpt", "epoch=12-step=99.ckpt", "last.ckpt"]:
        (tmp_path / f).write_bytes(b"")
    d = str(tmp_path)
    assert find_checkpoint(checkpoint_dir=d, checkpoint_type="last").endswith("last.ckpt")
    assert find_checkpoint(checkpoint_dir=d, checkpoint_type="best_so_far").endswith("epoch=12-step=99.ckpt")
    assert find_checkpoint(checkpoint_dir=d, checkpoint_type="epoch=3-step=10.ckpt").endswith("epoch=3-step=10.ckpt")
    with pytest.raises(ValueError):
        find_checkpoint(wandb_train_run_path="a/b/c", checkpoint_dir=d, checkpoint_type="last")
    with pytest.raises(ValueError):
        find_checkpoint()
    with pytest.raises(ValueError):
        find_checkpoint(checkpoint_dir=d, checkpoint_type="bogus")


def test_checkpoint_file_round_trip_with_compile_prefix(tmp_path):
    """Lightning-shaped .ckpt with the torch.compile'd 'g._orig_mod.' prefix and a functools.partial arch."""
    import functools

    from jamun_amd import synth
    from jamun_amd.checkpoint import load_checkpoint_file
    from jamun_amd.model import _kw, strip_prefix

    ck = synth.synthetic_checkpoint(prefix="g._orig_mod.")
    ck["hyper_parameters"]["arch"] = functools.partial(dict, **ck["hyper_parameters"]["arch"])
    ck["state_dict"]["g._orig_mod.layers.0.gated_conv.f.gate.gate.mul.output_mask"] = torch.ones(3)  # e3nn bookkeeping buffer
    path = str(tmp_path / "last.ckpt")
    torch.save(ck, path)
    back = load_checkpoint_file(path)
    sd = strip_prefix(back["state_dict"])
    assert "layers.0.gated_conv.f.f.radial_nn.3.weight" in sd and sd["output_gain"].ndim == 0
    assert _kw(back["hyper_parameters"]["arch"])["n_layers"] == 5


def test_walker_batch_collation_and_slicing():
    from jamun_amd import synth
    from jamun_amd.data import WalkerBatch

    mols = [synth.random_chain(n, seed=s) for s, n in enumerate([5, 9, 3, 7])]
    b = WalkerBatch.from_molecules(mols, labels=list("abcd"))
    assert b.ptr.tolist() == [0, 5, 14, 17, 24] and b.num_graphs == 4
    assert torch.equal(b.batch, torch.repeat_interleave(torch.arange(4), torch.tensor([5, 9, 3, 7])))
    # bonds offset by the cumulative atom count (data_with_residue_info.py:27-28)
    assert b.bonds[:, 4:12].min() >= 5 and b.bonds[:, 4:12].max() < 14
    s = b.slice_graphs(1, 3)
    assert s.ptr.tolist() == [0, 9, 12] and s.dataset_label == ["b", "c"]
    assert torch.equal(s.pos, b.pos[5:17]) and s.bonds.min() >= 0 and s.bonds.max() < 12
    assert s.bonds.shape[1] == 8 + 2
    with pytest.raises(AssertionError):
        b.with_pos(torch.zeros(3, 3))


def test_shard_ranges_cover_everything():
    from jamun_amd import dist

    for n, w in [(2048, 8), (7, 8), (10, 3), (1, 1), (0, 4)]:
        parts = [dist.shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
    costs = [17 * 16] * 10 + [57 * 32] * 3 + [9 * 8] * 20
    parts = [dist.shard_range_balanced(costs, r, 4) for r in range(4)]
    assert parts[0][0] == 0 and parts[-1][1] == len(costs) and all(parts[i][1] == parts[i + 1][0] for i in range(3))
    loads = [sum(costs[a:b]) for a, b in parts]
    assert max(loads) <= sum(costs) / 4 + max(costs)  # a contiguous split cannot do better than one item of slack


_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from jamun_amd import dist, synth
from jamun_amd.callbacks import SaveTrajectoryCallback
from jamun_amd.data import WalkerBatch
rank, world = dist.init_process_group("gloo")
assert world == 2
# ragged gather: rank r contributes r+2 rows
blk = torch.full((rank + 2, 3, 2), float(rank))
got = dist.gather_ragged(blk, dst=0)
if rank == 0:
    assert [g.shape[0] for g in got] == [2, 3] and float(got[1].mean()) == 1.0
else:
    assert got is None
# the trajectory form of the gather: numpy arrays on rank 0, received one block at a time through ONE reusable staging object; what
# comes back are pageable copies (a second gather through the same stager must not change the first result)
stg, tm = dist.HostStager(), {}
got1 = dist.gather_ragged_to_host(blk, dst=0, stager=stg, timings=tm)
got2 = dist.gather_ragged_to_host(blk + 10.0, dst=0, stager=stg, timings=tm)
if rank == 0:
    assert [type(g).__name__ for g in got1] == ["ndarray", "ndarray"] and [g.shape for g in got1] == [(2, 3, 2), (3, 3, 2)]
    assert float(got1[1].mean()) == 1.0 and float(got2[1].mean()) == 11.0 and float(got1[0].mean()) == 0.0
    assert tm["gather_bytes"] == 2 * 3 * 3 * 2 * 4 and tm["gather_s"] > 0.0
else:
    assert got1 is None and got2 is None and tm["gather_bytes"] == 0
assert (dist.gather_ragged_to_host(None, dst=0) == []) if rank == 0 else (dist.gather_ragged_to_host(None, dst=0) is None)
# walker sharding + trajectory gather through the callback: 5 walkers of one molecule -> ranks get 3 and 2
mol = synth.random_chain(6, seed=0)
class DS:
    molecule = dict(mol)
    def label(self): return "m"
batch = WalkerBatch.from_molecules([mol] * 5, labels=["m"] * 5)
lo, hi = dist.shard_range(batch.num_graphs, rank, world)
local = batch.slice_graphs(lo, hi)
class FakeSampler:
    device = torch.device("cpu"); is_global_zero = rank == 0; world_size = world; global_step = 0
cb = SaveTrajectoryCallback([DS()], output_dir=os.path.join(sys.argv[2], "sampler"), write_pdb=False)
samples = [{"dataset_label": "m", "xhat_traj": torch.full((6, 4, 3), float(lo + w))} for w in range(local.num_graphs)]
cb.on_after_sample_batch(samples, FakeSampler())
cb.on_after_sample_batch(samples, FakeSampler())
cb.flush()  # (files are written on a side thread; on_sample_end / flush wait for it)
dist.barrier()
if rank == 0:
    d = os.path.join(sys.argv[2], "sampler", "m", "predicted_samples", "npy")
    j = np.load(os.path.join(d, "joined.npy"))
    assert j.shape == (6, 10 * 4, 3), j.shape
    firsts = [float(np.load(os.path.join(d, f"{i}.npy"))[0, 0, 0]) for i in range(10)]
    assert firsts == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4], firsts
torch.manual_seed(42 + rank)
print(json.dumps({"rank": rank, "draw": float(torch.randn(1))}))
torch.distributed.destroy_process_group()
'''


def test_two_process_gloo_gather_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    draws = [json.loads(o.strip().splitlines()[-1])["draw"] for o, _ in outs]
    assert draws[0] != draws[1]  # seed + rank: ranks generate different chains


def test_parameter_schedule_callbacks():
    """walkjump/_callbacks.py:10-77 — override / decay / interpolate integrator fields per measurement index."""
    import math

    from jamun_amd.sampling import (BAOAB, DeltaSqrtDecayCallback, InterpolateParametersCallback,
                                    MeasurementDependentParametersCallback)

    base = BAOAB(delta=0.04, friction=1.0, steps=50)
    cb = MeasurementDependentParametersCallback({2: {"delta": 0.01, "steps": 7}})
    assert cb.on_before_sample(base, 1) is base and cb.on_after_sample(base, 1) is base
    m2 = cb.on_before_sample(base, 2)
    assert (m2.delta, m2.steps, m2.friction) == (0.01, 7, 1.0) and isinstance(m2, BAOAB)
    back = cb.on_after_sample(m2, 2)
    assert (back.delta, back.steps) == (0.04, 50) and cb.previous_params is None
    dc = DeltaSqrtDecayCallback()
    m4 = dc.on_before_sample(base, 4)
    assert m4.delta == 0.04 / math.sqrt(4) and dc.on_after_sample(m4, 4).delta == 0.04
    ip = InterpolateParametersCallback({"delta": (0.04, 0.01), "steps": (10, 50)})
    m1 = ip.on_before_sample(base, 1)
    assert m1.delta == 0.04 and m1.steps == 10
    m4 = ip.on_before_sample(base, 4)  # f = 0.5
    assert abs(m4.delta - 0.025) < 1e-12 and m4.steps == 30 and isinstance(m4.steps, int)
    assert ip.on_after_sample(m4, 4) is m4


def test_separable_conv_checkpoint_is_recognised_and_experimental_rejected():
    """e3conv_separable.yaml swaps the ConvBlock's conv for SeparableConv: recognised at load by the factory's name (a pickled
    partial of the class) or by the ``tp.lin.weight`` parameters; a checkpoint that names SeparableConv but carries the default
    conv's parameters fails loudly when the native model is built (missing tensor); ExperimentalConv is refused by name."""
    import functools

    from jamun_amd import synth
    from jamun_amd.model import Denoiser

    class SeparableConv:  # stands in for jamun.e3tools.nn.SeparableConv inside the pickled partial
        pass

    class ExperimentalConv:
        pass

    ck = synth.synthetic_checkpoint(separable=True)
    ck["hyper_parameters"]["arch"]["hidden_layer_factory"] = functools.partial(dict, conv=functools.partial(SeparableConv))
    m = Denoiser.from_checkpoint_dict(ck)
    assert m.arch["separable_conv"] is True and m._native.hparams_struct.separable == 1
    assert Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint()).arch["separable_conv"] is False
    bad = synth.synthetic_checkpoint()  # default parameters under a SeparableConv factory: radial_nn.3 has the wrong size
    bad["hyper_parameters"]["arch"] = dict(bad["hyper_parameters"]["arch"], hidden_layer_factory=functools.partial(dict, conv=functools.partial(SeparableConv)))
    m_bad = Denoiser.from_checkpoint_dict(bad)  # (tensors are checked when a sampler packs them, on the GPU)
    assert m_bad.arch["separable_conv"] is True
    ex = synth.synthetic_checkpoint()
    ex["hyper_parameters"]["arch"] = dict(ex["hyper_parameters"]["arch"], hidden_layer_factory=functools.partial(dict, conv=functools.partial(ExperimentalConv)))
    with pytest.raises(NotImplementedError, match="Experimental"):
        Denoiser.from_checkpoint_dict(ex)


def test_trajectory_metric_callback_dispatches_by_label():
    """callbacks/sampler/_utils.py:22-56 — one meter per dataset label, samples routed by `dataset_label`, compute()
    results logged through the sampler, hooks forwarded; samples answer both key and attribute access."""
    from jamun_amd.callbacks import TrajectoryMetricCallback
    from jamun_amd.sampling import SampleGraph

    class DS:
        def __init__(self, label):
            self._l = label

        def label(self):
            return self._l

    class Meter:
        def __init__(self, dataset):
            self.dataset, self.seen, self.events = dataset, [], []

        def to(self, device):
            self.events.append(("to", device))

        def on_sample_start(self):
            self.events.append("start")

        def update(self, sample):
            self.seen.append(sample.xhat_traj)  # attribute access, as reference metrics do

        def compute(self):
            return {f"{self.dataset.label()}/n": len(self.seen)}

        def on_after_sample_batch(self):
            self.events.append("batch")

        def on_sample_end(self):
            self.events.append("end")

    class FakeSampler:
        def __init__(self):
            self.fabric, self.device, self.logged = self, "dev0", []

        def log_dict(self, m):
            self.logged.append(m)

    cb = TrajectoryMetricCallback([DS("b"), DS("a"), DS("b")], Meter)
    assert list(cb.meters) == ["a", "b"]
    smp = FakeSampler()
    cb.on_sample_start(smp)
    cb.on_after_sample_batch([SampleGraph(dataset_label="b", xhat_traj=1), SampleGraph(dataset_label="a", xhat_traj=2),
                              SampleGraph(dataset_label="b", xhat_traj=3)], smp)
    cb.on_sample_end(smp)
    assert cb.meters["b"].seen == [1, 3] and cb.meters["a"].seen == [2]
    assert smp.logged == [{"a/n": 1}, {"b/n": 2}]
    assert cb.meters["a"].events == [("to", "dev0"), "start", "batch", "end"]
    with pytest.raises(AttributeError):
        SampleGraph(a=1).missing


def test_mdtraj_dataset_reads_npz_frames(tmp_path):
    """data/_mdtraj.py:155-237 — topology from the PDB (heavy protein atoms), frames from Timewarp-style arrays that
    carry ALL atoms (hydrogens included): slicing by start_frame / num_frames / subsample, atom selection, item layout."""
    import numpy as np

    from jamun_amd.pdb import MDtrajDataset, read_pdb

    lines = []
    recs = [("N", "ALA", "N"), ("H", "ALA", "H"), ("CA", "ALA", "C"), ("HA", "ALA", "H"), ("C", "ALA", "C"), ("O", "ALA", "O"),
            ("CB", "ALA", "C"), ("O", "HOH", "O")]
    for i, (name, res, el) in enumerate(recs):
        lines.append(f"ATOM  {i + 1:5d} {name:<4s} {res:>3s} A   1    {i * 1.0:8.3f}{0.0:8.3f}{0.0:8.3f}  1.00  0.00          {el:>2s}")
    (tmp_path / "m.pdb").write_text("\n".join(lines) + "\nEND\n")
    mol = read_pdb(str(tmp_path / "m.pdb"))
    assert mol["all_atom_index"].tolist() == [0, 2, 4, 5, 6] and mol["n_all_atoms"] == 8
    frames = np.arange(6 * 8 * 3, dtype=np.float32).reshape(6, 8, 3) / 100.0
    np.savez(tmp_path / "m-traj-arrays.npz", positions=frames)
    ds = MDtrajDataset(str(tmp_path), ["m-traj-arrays.npz"], "m.pdb", "m", num_frames=4, start_frame=1, subsample=2)
    assert len(ds) == 2 and ds.label() == "m"
    g = ds[1]
    assert g["dataset_label"] == "m" and g["pos"].shape == (5, 3)
    assert torch.equal(g["pos"], torch.from_numpy(frames[3][[0, 2, 4, 5, 6]]))
    assert torch.equal(ds[0]["pos"], torch.from_numpy(frames[1][[0, 2, 4, 5, 6]]))
    assert g["atom_type_index"].shape == (5,) and g["bonds"].shape[0] == 2
    with pytest.raises(ValueError):
        np.savez(tmp_path / "bad.npz", positions=frames[:, :7])
        MDtrajDataset(str(tmp_path), ["bad.npz"], "m.pdb", "m")


def test_w3j_buffers_cross_check():
    """A checkpoint that carries e3nn's compiled Wigner-3j buffers confirms or flips the sign of the cross-product path."""
    import math

    from jamun_amd.checkpoint import w3j_111_sign_from_state_dict as f

    eps = torch.zeros(3, 3, 3)
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[i, j, k], eps[i, k, j] = 1.0, -1.0
    eps /= math.sqrt(6.0)
    pre = "layers.0.gated_conv.f.f.tp._compiled_main_left_right."
    assert f({"a.weight": torch.ones(3)}) == 1.0
    assert f({pre + "_w3j_1_1_1": eps.flatten(), pre + "_w3j_1_1_0": torch.eye(3) / math.sqrt(3.0), pre + "_w3j_0_0_0": torch.ones(1, 1, 1)}) == 1.0
    assert f({pre + "_w3j_1_1_1": -eps}) == -1.0
    with pytest.raises(ValueError):
        f({pre + "_w3j_1_1_1": eps.permute(1, 0, 2) * 0.5})
    with pytest.raises(ValueError):
        f({pre + "_w3j_1_0_1": torch.eye(3)})


def _read_dcd(path):
    """Independent minimal reader of the CHARMM DCD layout (Fortran records, little endian): returns [frames, atoms, 3] A."""
    import struct

    b = open(path, "rb").read()
    o = 0

    def rec():
        nonlocal o
        n = struct.unpack_from("<i", b, o)[0]
        body = b[o + 4 : o + 4 + n]
        assert struct.unpack_from("<i", b, o + 4 + n)[0] == n, "record length markers disagree"
        o += n + 8
        return body

    head = rec()
    assert len(head) == 84 and head[:4] == b"CORD"
    ic = struct.unpack("<20i", head[4:])
    nset, has_cell, version = ic[0], ic[10], ic[19]
    assert version == 24 and has_cell == 0 and ic[2] == 1
    title = rec()
    assert struct.unpack_from("<i", title)[0] * 80 + 4 == len(title)
    natom = struct.unpack("<i", rec())[0]
    frames = np.zeros((nset, natom, 3), dtype=np.float32)
    for t in range(nset):
        for c in range(3):
            frames[t, :, c] = np.frombuffer(rec(), dtype="<f4")
    assert o == len(b)
    return frames


def test_save_trajectory_writes_reference_file_set(tmp_path):
    """metrics/_save_trajectory.py:17-30,53-56,78-97: topology.pdb, <i>.npy/.pdb/.dcd per chain, joined.npy/.pdb/.dcd — and
    what analysis/load_trajectory.py:88-107 needs (dcd/joined.dcd + topology.pdb).  DCD parsed back by an independent reader."""
    from jamun_amd import pdb, synth
    from jamun_amd.callbacks import SaveTrajectoryCallback
    from jamun_amd.sampling import SampleGraph

    mol = dict(synth.ag_dipeptide(), elements=["N", "C", "C", "C", "O", "N", "C", "C", "O", "O"], residue_ids=[1] * 5 + [2] * 5)
    src = str(tmp_path / "uncapped_AG.pdb")
    pdb.save_pdb(src, mol, mol["pos"][None])
    ds = pdb.create_dataset_from_pdbs([src])

    class FakeSampler:
        device = torch.device("cpu"); is_global_zero = True; world_size = 1; global_step = 0

    out = str(tmp_path / "sampler")
    cb = SaveTrajectoryCallback(ds, output_dir=out)
    cb.on_sample_start(FakeSampler())
    torch.manual_seed(0)
    T = 5
    base = ds[0].molecule
    mk = lambda w: SampleGraph(dataset_label="uncapped_AG", atom_type_index=base["atom_type_index"],
                               xhat_traj=base["pos"][:, None, :] + 0.01 * torch.randn(10, T, 3) + w)
    batch0, batch1 = [mk(0), mk(1), mk(2)], [mk(3), mk(4), mk(5)]
    cb.on_after_sample_batch(batch0, FakeSampler())
    cb.on_after_sample_batch(batch1, FakeSampler())
    cb.on_sample_end(FakeSampler())
    root = os.path.join(out, "uncapped_AG")
    have = sorted(os.path.relpath(os.path.join(dp, f), root) for dp, _, fs in os.walk(root) for f in fs)
    want = ["topology.pdb"] + [f"predicted_samples/{e}/{i}.{e}" for e in ("dcd", "npy", "pdb") for i in list(range(6)) + ["joined"]]
    assert have == sorted(want)
    # DCD: Angstrom, frames x atoms; joined = chains' frames concatenated in chain order
    d3 = _read_dcd(os.path.join(root, "predicted_samples/dcd/3.dcd"))
    assert d3.shape == (T, 10, 3)
    assert np.allclose(d3, 10.0 * batch1[0]["xhat_traj"].permute(1, 0, 2).numpy(), atol=1e-5)
    dj = _read_dcd(os.path.join(root, "predicted_samples/dcd/joined.dcd"))
    nj = np.load(os.path.join(root, "predicted_samples/npy/joined.npy"))
    assert dj.shape == (6 * T, 10, 3) and nj.shape == (10, 6 * T, 3)
    assert np.allclose(dj, 10.0 * np.transpose(nj, (1, 0, 2)), atol=1e-5)
    assert np.array_equal(nj[:, 2 * T : 3 * T], batch0[2]["xhat_traj"].numpy())
    # PDB: the reference's record layout (utils/mdtraj.py:26-60), readable back as the same molecule
    txt = open(os.path.join(root, "predicted_samples/pdb/0.pdb")).read().splitlines()
    assert txt[0] == "MODEL        0" and txt[-1] == "END" and sum(l.startswith("MODEL") for l in txt) == T
    assert txt[1].startswith("ATOM      1 N    ALA 0   1    ") and txt[1].endswith(" N")
    assert sum(l.startswith("CONECT") for l in txt) == 10 * T and txt[11].startswith("TER      11      GLY 0   2")
    topo = pdb.read_pdb(os.path.join(root, "topology.pdb"))
    assert torch.equal(topo["atom_type_index"], base["atom_type_index"]) and topo["bonds"].shape[1] == 9
    assert torch.allclose(topo["pos"], base["pos"], atol=1e-4)
    # validate_sample (metrics/_utils.py:15-28): wrong atom types / unknown label are refused
    bad = mk(0)
    bad["atom_type_index"] = torch.zeros(10, dtype=torch.int32)
    with pytest.raises(ValueError, match="Atom types"):
        cb.on_after_sample_batch([bad], FakeSampler())
    with pytest.raises(KeyError):
        cb.on_after_sample_batch([SampleGraph(dataset_label="other", xhat_traj=torch.zeros(10, T, 3))], FakeSampler())


def test_save_trajectory_true_samples_and_reference_npy_numbering(tmp_path):
    """``save_true_trajectory`` writes the dataset's own frames as true_samples/{pdb,dcd}/0.* (_save_trajectory.py:22-26,58-62);
    ``npy_index_restarts_per_batch`` reproduces the reference's per-batch restart of the .npy numbering (:81 vs :89): after two
    batches of three chains the reference has 0..2.npy (holding batch 1) but 0..5.pdb / .dcd."""
    from jamun_amd import pdb, synth
    from jamun_amd.callbacks import SaveTrajectoryCallback
    from jamun_amd.sampling import SampleGraph

    mol = dict(synth.ag_dipeptide(), elements=["N", "C", "C", "C", "O", "N", "C", "C", "O", "O"], residue_ids=[1] * 5 + [2] * 5)
    src = str(tmp_path / "uncapped_AG.pdb")
    pdb.save_pdb(src, mol, mol["pos"][None])
    ds = pdb.create_dataset_from_pdbs([src])

    class FakeSampler:
        device = torch.device("cpu"); is_global_zero = True; world_size = 1; global_step = 0

    out = str(tmp_path / "sampler")
    cb = SaveTrajectoryCallback(ds, output_dir=out, save_true_trajectory=True, npy_index_restarts_per_batch=True)
    cb.on_sample_start(FakeSampler())
    base = ds[0].molecule
    mk = lambda w: SampleGraph(dataset_label="uncapped_AG", atom_type_index=base["atom_type_index"], xhat_traj=base["pos"][:, None, :] + torch.full((10, 2, 3), float(w)))
    cb.on_after_sample_batch([mk(0), mk(1), mk(2)], FakeSampler())
    cb.on_after_sample_batch([mk(3), mk(4), mk(5)], FakeSampler())
    cb.flush()
    root = os.path.join(out, "uncapped_AG")
    have = sorted(os.path.relpath(os.path.join(dp, f), root) for dp, _, fs in os.walk(root) for f in fs)
    want = (["topology.pdb", "true_samples/pdb/0.pdb", "true_samples/dcd/0.dcd"]
            + [f"predicted_samples/{e}/{i}.{e}" for e in ("dcd", "pdb") for i in list(range(6)) + ["joined"]]
            + [f"predicted_samples/npy/{i}.npy" for i in [0, 1, 2, "joined"]])
    assert have == sorted(want)
    assert np.allclose(np.load(os.path.join(root, "predicted_samples/npy/1.npy")), mk(4)["xhat_traj"].numpy())  # batch 1 overwrote batch 0
    assert np.load(os.path.join(root, "predicted_samples/npy/joined.npy")).shape == (10, 12, 3)
    true = _read_dcd(os.path.join(root, "true_samples/dcd/0.dcd"))
    assert true.shape == (1, 10, 3) and np.allclose(true[0], 10.0 * base["pos"].numpy(), atol=1e-4)


_WORKER2 = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from jamun_amd import dist, synth
from jamun_amd.callbacks import SaveTrajectoryCallback
from jamun_amd.data import WalkerBatch
from jamun_amd.sampling import Sampler
rank, world = dist.init_process_group("gloo")
assert world == 2
# ---- (1) gather with a rank that owns NO walker of a label: rank 1 contributes nothing for label "a"
got = dist.gather_ragged(torch.full((2, 3, 4, 3), 5.0) if rank == 0 else None, dst=0, device=torch.device("cpu"))
assert (got is None) if rank else ([tuple(g.shape) for g in got] == [(2, 3, 4, 3)])
got = dist.gather_ragged(None, dst=0, device=torch.device("cpu"))  # nobody has anything
assert (got is None) if rank else (got == [])
try:  # trailing shapes that disagree raise on EVERY rank, before any payload moves
    dist.gather_ragged(torch.zeros(1, 3 + rank), dst=0)
    raise SystemExit("expected ValueError")
except ValueError:
    pass
try:  # a problem only ONE rank can see (unsupported dtype on rank 1) is reported through the metadata exchange and raised on both
    dist.gather_ragged(torch.zeros(2, 3, dtype=torch.float16 if rank == 1 else torch.float32), dst=0)
    raise SystemExit("expected ValueError")
except ValueError as e:
    assert "rank 1" in str(e) and "dtype" in str(e), str(e)
try:  # ... and so are too many dimensions on rank 0
    dist.gather_ragged(torch.zeros([1] * (10 if rank == 0 else 2)), dst=0)
    raise SystemExit("expected ValueError")
except ValueError as e:
    assert "rank 0" in str(e), str(e)
dist.barrier()
# ---- (2) Sampler.sample(shard_walkers=True) end to end with a stub model / batch sampler, two dataset labels, ragged sizes
mol_a, mol_b = synth.random_chain(6, seed=0), synth.random_chain(9, seed=1)
class DS:
    def __init__(self, mol, label): self.molecule, self._l = dict(mol), label
    def label(self): return self._l
# walkers: a a a b b   -> the contiguous split balanced by cost (42 42 42 90 90) puts [a a a b] on rank 0 and [b] on rank 1:
# rank 1 has no walker of label "a"
batch = WalkerBatch.from_molecules([mol_a] * 3 + [mol_b] * 2, labels=["a"] * 3 + ["b"] * 2)
class StubModel:
    device = torch.device("cpu")
    def to(self, d): return self
    def eval(self): return self
class StubBatchSampler:
    sigma = 0.04
    mcmc = type("M", (), {"rng": "philox"})()
    def sample(self, model, y_init, v_init):
        T = 3
        r = torch.randn(1)  # rank-dependent stream (seed + rank)
        walker_id = torch.repeat_interleave(torch.arange(model.init_graphs.num_graphs), model.init_graphs.ptr.diff())
        xt = torch.zeros(T, y_init.shape[0], 3) + walker_id[None, :, None].float() + 100.0 * rank
        return {"xhat": y_init, "y": y_init, "v": torch.zeros_like(y_init), "sample": y_init, "xhat_traj": xt, "y_traj": xt.clone(),
                "score_traj": xt.clone(), "t_traj": torch.ones(T), "_draw": r}
out_dir = os.path.join(sys.argv[2], "sampler")
cb = SaveTrajectoryCallback([DS(mol_a, "a"), DS(mol_b, "b")], output_dir=out_dir, write_pdb=False)
sampler = Sampler(callbacks=[cb], shard_walkers=True)
torch.manual_seed(42 + sampler.fabric.global_rank)
sampler.sample(model=StubModel(), batch_sampler=StubBatchSampler(), num_batches=2, init_graphs=batch, continue_chain=True)
dist.barrier()
if rank == 0:
    a = os.path.join(out_dir, "a", "predicted_samples", "npy"); b = os.path.join(out_dir, "b", "predicted_samples", "npy")
    ja, jb = np.load(os.path.join(a, "joined.npy")), np.load(os.path.join(b, "joined.npy"))
    assert ja.shape == (6, 2 * 3 * 3, 3) and jb.shape == (9, 2 * 2 * 3, 3), (ja.shape, jb.shape)
    # global walker order within a label, per batch: rank 0's walkers first (local ids 0..2 are "a", 3 is "b"), then rank 1's
    # (its only walker, local id 0, marked + 100)
    assert [float(np.load(os.path.join(a, f"{i}.npy"))[0, 0, 0]) for i in range(6)] == [0, 1, 2, 0, 1, 2]
    assert [float(np.load(os.path.join(b, f"{i}.npy"))[0, 0, 0]) for i in range(4)] == [3, 100, 3, 100]
    assert os.path.exists(os.path.join(out_dir, "b", "predicted_samples", "dcd", "joined.dcd"))
# ---- (3) more ranks than walkers: the empty rank still takes part in the collectives
one = WalkerBatch.from_molecules([mol_a], labels=["a"])
cb2 = SaveTrajectoryCallback([DS(mol_a, "a")], output_dir=os.path.join(sys.argv[2], "sampler1"), write_pdb=False)
s2 = Sampler(callbacks=[cb2], shard_walkers=True)
s2.sample(model=StubModel(), batch_sampler=StubBatchSampler(), num_batches=1, init_graphs=one)
dist.barrier()
if rank == 0:
    assert np.load(os.path.join(sys.argv[2], "sampler1", "a", "predicted_samples", "npy", "joined.npy")).shape == (6, 3, 3)
print(json.dumps({"rank": rank, "ok": True}))
torch.distributed.destroy_process_group()
'''


def test_two_process_gloo_sharded_sampler_two_labels(tmp_path):
    """Sampler.sample(shard_walkers=True) on 2 gloo ranks through to the written files: a rank that owns no walker of a
    label, balanced ragged sharding, global walker order, seed + rank, and a rank with an empty shard."""
    script = tmp_path / "worker2.py"
    script.write_text(_WORKER2)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
        assert json.loads(o.strip().splitlines()[-1])["ok"]


def test_bench_self_launches_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher spawns the two ranks itself (before any GPU call) and relays ONE JSON
    line from rank 0; --dry-run keeps the rank plumbing (gloo) and skips the kernels."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 4
    # the N > 1 line proves its ranks (an all-reduce of ones over the group) and times north_star's one collective: Sampler.sample(
    # shard_walkers=True) + SaveTrajectoryCallback over both ranks — here with constant trajectories, the gather and the writer are real
    assert d["rccl"] == {"backend": "gloo", "world": 2, "ranks_seen": 2}
    sh = d["e2e_sharded"]
    assert sh["walkers_total"] == 7 and sh["num_batches"] == 2 and sh["steps_per_batch"] == 8
    assert sh["joined_shape"] == [17, 7 * 2 * 8, 3]  # every walker of both ranks, both batches, in the joined file on rank 0
    assert sh["gather_bytes"] == 2 * 3 * 17 * 8 * 3 * 4  # rank 1's three walkers crossed ranks, once per batch
    assert 0.0 < sh["gather_s"] <= sh["wall_s"] and sh["conformations_per_s"] > 0 and sh["files_written"] >= 2 * 8
    # under a launcher (WORLD_SIZE set) the script must NOT spawn again: a single rank with WORLD_SIZE=1 just runs
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                        capture_output=True, text=True, timeout=240)
    assert r1.returncode == 0 and json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 1


def test_bench_self_launch_supervises_its_ranks():
    """A rank that dies before the rendezvous must not leave the parent (or the surviving rank, which waits in the rendezvous)
    hanging: the parent notices the non-zero exit, terminates the other rank, relays the failed rank's stderr and returns its
    code within seconds."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--dry-run-fail-rank", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    dt = time.monotonic() - t0
    assert r.returncode != 0
    assert dt < 60, dt
    assert "rank 1 exited" in r.stderr and "simulated failure" in r.stderr, r.stderr[-1500:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_unpickler_reads_omegaconf_shaped_hparams(tmp_path):
    """A real JAMUN checkpoint nests omegaconf containers and partials of classes from modules that are not installed here
    ("Checkpoint format").  Build that shape with stand-in modules, save, REMOVE the modules, load."""
    import functools
    import types

    from jamun_amd import synth
    from jamun_amd.checkpoint import load_checkpoint_file
    from jamun_amd.model import Denoiser, _kw

    def module(name, **classes):
        m = types.ModuleType(name)
        for cname, cls in classes.items():
            cls.__module__, cls.__qualname__ = name, cname
            setattr(m, cname, cls)
        sys.modules[name] = m
        return m

    class Metadata:
        def __init__(self, **kw): self.__dict__.update(kw)

    class AnyNode:  # omegaconf value node: pickled state {_metadata, _parent, _val}
        def __init__(self, val, parent=None): self._metadata, self._parent, self._val = Metadata(optional=True), parent, val

    class DictConfig:  # container: {_metadata, _parent, _content: {key: node}} with back references to the parent
        def __init__(self, content, parent=None):
            self._metadata, self._parent, self._content = Metadata(key_type=str), parent, {}
            for k, v in content.items():
                self._content[k] = v if isinstance(v, (DictConfig, ListConfig)) else AnyNode(v, self)
                self._content[k]._parent = self

    class ListConfig:
        def __init__(self, content, parent=None):
            self._metadata, self._parent = Metadata(element_type=str), parent
            self._content = [AnyNode(v, self) for v in content]

    class AttributeDict(dict): pass
    class E3Conv: pass
    class ConvBlock: pass
    class Conv: pass
    class EquivariantMLP: pass
    class ConstantSigma:
        def __init__(self, sigma): self.sigma = sigma
    class Adam: pass

    names = ["omegaconf", "omegaconf.dictconfig", "omegaconf.listconfig", "omegaconf.nodes", "omegaconf.base", "jamun", "jamun.model", "jamun.model.arch",
             "jamun.model.arch.e3conv", "jamun.e3tools", "jamun.e3tools.nn", "jamun.e3tools.nn._conv", "jamun.e3tools.nn._mlp", "jamun.distributions",
             "jamun.distributions._distributions", "lightning_fake", "lightning_fake.fabric", "lightning_fake.fabric.utilities", "lightning_fake.fabric.utilities.data",
             "torch_fake_optim"]
    saved = {n: sys.modules.get(n) for n in names}
    try:
        for n in names:
            module(n)
        module("omegaconf.dictconfig", DictConfig=DictConfig)
        module("omegaconf.listconfig", ListConfig=ListConfig)
        module("omegaconf.nodes", AnyNode=AnyNode)
        module("omegaconf.base", Metadata=Metadata)
        module("jamun.model.arch.e3conv", E3Conv=E3Conv)
        module("jamun.e3tools.nn._conv", ConvBlock=ConvBlock, Conv=Conv)
        module("jamun.e3tools.nn._mlp", EquivariantMLP=EquivariantMLP)
        module("jamun.distributions._distributions", ConstantSigma=ConstantSigma)
        module("lightning_fake.fabric.utilities.data", AttributeDict=AttributeDict)
        module("torch_fake_optim", Adam=Adam)
        base = synth.synthetic_checkpoint(prefix="g._orig_mod.")
        a = base["hyper_parameters"]["arch"]
        arch = functools.partial(
            E3Conv, irreps_out=a["irreps_out"], irreps_hidden=a["irreps_hidden"], irreps_sh=a["irreps_sh"], n_layers=a["n_layers"],
            edge_attr_dim=a["edge_attr_dim"], atom_type_embedding_dim=8, atom_code_embedding_dim=8, residue_code_embedding_dim=32,
            residue_index_embedding_dim=8, use_residue_information=True, use_residue_sequence_index=False,
            hidden_layer_factory=functools.partial(ConvBlock, conv=functools.partial(Conv)),
            output_head_factory=functools.partial(EquivariantMLP, irreps_hidden_list=ListConfig([a["irreps_hidden"]])),
        )
        hp = AttributeDict(base["hyper_parameters"])
        hp.update(arch=arch, optim=functools.partial(Adam, lr=0.002), sigma_distribution=ConstantSigma(0.04),
                  torch_compile_kwargs=DictConfig({"fullgraph": True, "dynamic": True, "mode": "default", "nested": DictConfig({"k": 3})}),
                  use_torch_compile=True)
        path = str(tmp_path / "last.ckpt")
        torch.save({"state_dict": base["state_dict"], "hyper_parameters": hp, "pytorch-lightning_version": "2.4.0", "epoch": 7}, path)
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    assert "omegaconf" not in sys.modules or not hasattr(sys.modules["omegaconf"], "dictconfig")
    ck = load_checkpoint_file(path)
    from jamun_amd.checkpoint import to_plain

    hp2 = to_plain(ck["hyper_parameters"])
    assert isinstance(hp2, dict) and hp2["max_radius"] == base["hyper_parameters"]["max_radius"]
    kw = _kw(hp2["arch"])
    assert kw["n_layers"] == 5 and isinstance(kw["n_layers"], int) and kw["irreps_hidden"] == "120x0e + 32x1e"
    head = _kw(kw["output_head_factory"])
    assert head["irreps_hidden_list"] == ["120x0e + 32x1e"] and all(type(x) is str for x in head["irreps_hidden_list"])
    assert _kw(hp2["torch_compile_kwargs"]) == {"fullgraph": True, "dynamic": True, "mode": "default", "nested": {"k": 3}}
    conv = _kw(kw["hidden_layer_factory"])["conv"]
    assert isinstance(conv, functools.partial) and conv.func.__name__ == "Conv"
    model = Denoiser.load_from_checkpoint(path)  # the whole way: hparams -> arch dict -> native model handle
    assert model.arch["n_layers"] == 5 and model.max_radius == base["hyper_parameters"]["max_radius"]


def test_docs_cite_profile_files_that_exist():
    """Every `profiles/...` path quoted in DESIGN.md / README.md / INTEGRATION.md names files that are in the tree (`*` globs,
    `<cfg>` = any config name, `{a,b}` alternatives): stale references to removed evidence fail here, not at review time."""
    import glob
    import itertools
    import re

    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for ref in sorted(set(re.findall(r"profiles/[A-Za-z0-9_./*<>{},-]*", text))):
            ref = ref.rstrip(".,")
            if ref in ("profiles/", "profiles"):
                continue
            pats = [ref.replace("<cfg>", "cfg*")]
            m = re.search(r"\{([^}]*)\}", pats[0])
            if m:
                pats = [pats[0][: m.start()] + alt + pats[0][m.end():] for alt in m.group(1).split(",")]
            for pat in pats:
                if not glob.glob(os.path.join(ROOT, pat)):
                    missing.append(f"{doc}: {ref} ({pat})")
    assert not missing, missing


def test_quoted_counter_summaries_were_collected_from_this_build():
    """Every `profiles/*_pmc_traffic.json` that DESIGN.md or README.md names — the files `bench.py` takes `roofline.traffic` from — carries
    the digest of the library sources as they are NOW (`_build.source_digest`, written by `profiles/collect.sh` from
    `bench.py --signature`): an edit of any kernel after the counters were collected fails here until they are collected again, and
    `bench.py` itself refuses such a file (`traffic_stale`)."""
    import json
    import re

    from jamun_amd.csrc import build

    want = build._digest()
    cited = set()
    for doc in ("DESIGN.md", "README.md"):
        cited |= set(re.findall(r"profiles/([A-Za-z0-9_]+_pmc_traffic\.json)", open(os.path.join(ROOT, doc)).read()))
    assert cited, "DESIGN.md section 7 names the counter summaries its traffic figures come from"
    for name in sorted(cited):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d.get("_build", {}).get("source_digest") == want, f"{name} was collected from another build of the library"
    # and bench.py's reader applies the same rule: a summary with a foreign digest is reported as stale, never quoted
    import bench

    real = bench.build_signature

    try:
        bench.build_signature = lambda stats=None: {"source_digest": "0" * 64}
        r = bench._pmc_traffic("k_conv_mf<", "cfg2", "", None)
        assert r is None or r[0] is None
    finally:
        bench.build_signature = real


def test_parse_datasets_from_directory(tmp_path):
    """data/_utils.py:36-116 over a Timewarp-style tree (`<code>-traj-arrays.npz` + `<code>-traj-state0.pdb`): code = group 1 of
    the anchored regular expression, datasets sorted by code, filter -> offset -> max in the reference's order, dataset kwargs
    forwarded, topology per code (pdb_pattern) or shared (pdb_file), the reference's errors, and the Hydra target alias the
    reference's own sample_uncapped_2AA / 4AA / mdgen experiment files use."""
    import numpy as np

    from jamun_amd import config as C
    from jamun_amd import synth
    from jamun_amd.pdb import parse_datasets_from_directory

    root = tmp_path / "timewarp" / "2AA-1-large" / "test"
    codes = ["WG", "AG", "KR", "GA", "FY"]
    mols = synth.write_timewarp_tree(str(root), codes, n_frames=5)
    (root / "notes.txt").write_text("not a trajectory")
    (root / "ZZ-traj-state0.pdb").write_text((root / "AG-traj-state0.pdb").read_text())  # a topology without a trajectory: ignored
    kw = dict(root=str(root), traj_pattern="^(.*)-traj-arrays.npz", pdb_pattern="^(.*)-traj-state0.pdb")
    ds = parse_datasets_from_directory(**kw, subsample=2)
    assert [d.label() for d in ds] == sorted(codes)
    for d in ds:
        m = mols[d.label()]
        assert len(d) == 3  # frames 0, 2, 4
        g = d[0]
        assert g["dataset_label"] == d.label() and g["pos"].shape == m["pos"].shape  # hydrogens of the arrays dropped
        assert torch.equal(g["atom_code_index"], m["atom_code_index"]) and torch.equal(g["residue_code_index"], m["residue_code_index"])
        assert sorted(map(tuple, g["bonds"].T.tolist())) == sorted(map(tuple, m["bonds"].T.tolist()))  # template order vs sorted
        assert torch.allclose(g["pos"], m["pos"], atol=1e-6)
        arr = np.load(root / f"{d.label()}-traj-arrays.npz")["positions"]
        assert arr.shape[1] == m["pos"].shape[0] + 2  # one H per residue
    # filter, then sort, then offset, then max (data/_utils.py:91-99)
    lab = lambda **k: [d.label() for d in parse_datasets_from_directory(**kw, **k)]
    assert lab(filter_codes=["KR", "AG", "QQ"]) == ["AG", "KR"]
    assert lab(max_datasets=2) == ["AG", "FY"]
    assert lab(max_datasets_offset=3) == ["KR", "WG"]
    assert lab(max_datasets_offset=1, max_datasets=2) == ["FY", "GA"]
    assert lab(filter_codes=["WG", "GA", "AG"], max_datasets_offset=1, max_datasets=1) == ["GA"]
    ds = parse_datasets_from_directory(**kw, num_frames=2, start_frame=1)
    assert all(len(d) == 2 for d in ds)
    # a shared topology (pdb_file) for every code; sub-directory prefixes in the patterns
    sub = tmp_path / "tree"
    (sub / "trajs").mkdir(parents=True)
    (sub / "top").mkdir()
    for c in ("AG", "AG2"):
        np.savez(sub / "trajs" / f"{c}_sim.npz", positions=np.load(root / "AG-traj-arrays.npz")["positions"])
    (sub / "top" / "shared.pdb").write_text((root / "AG-traj-state0.pdb").read_text())
    ds = parse_datasets_from_directory(str(sub), "trajs/^(.*)_sim.npz", pdb_file="top/shared.pdb")
    assert [d.label() for d in ds] == ["AG", "AG2"] and len(ds[1]) == 5
    with pytest.raises(ValueError, match="Exactly one"):
        parse_datasets_from_directory(str(sub), "trajs/^(.*)_sim.npz", pdb_pattern="top/^(.*).pdb", pdb_file="top/shared.pdb")
    with pytest.raises(ValueError, match="wildcards"):
        parse_datasets_from_directory(str(sub), "tr*/^(.*)_sim.npz", pdb_file="top/shared.pdb")
    with pytest.raises(ValueError, match="No codes"):
        parse_datasets_from_directory(str(sub), "top/^(.*)_sim.npz", pdb_file="top/shared.pdb")
    with pytest.raises(KeyError):  # a code whose topology is missing (the reference's dict lookup)
        parse_datasets_from_directory(str(sub), "trajs/^(.*)_sim.npz", pdb_pattern="top/^(AG2).pdb")
    with pytest.raises(NotImplementedError):
        parse_datasets_from_directory(**kw, as_iterable=True)
    # the Hydra node of sample_uncapped_2AA.yaml:8-13 instantiates through the alias
    node = {"_target_": "jamun.data.parse_datasets_from_directory", "root": str(root), "traj_pattern": "^(.*)-traj-arrays.npz",
            "pdb_pattern": "^(.*)-traj-state0.pdb", "subsample": 1, "max_datasets": 3}
    ds = C.instantiate(node)
    assert [d.label() for d in ds] == ["AG", "FY", "GA"] and len(ds[0]) == 5


def test_bench_sharded_leg_cannot_cost_the_line_its_headline():
    """`bench._guarded_sharded_leg`: the exchange between GPUs has never run on a multi-GPU node, so an exception inside the leg becomes an
    `error` entry of the line, and a leg that does not return makes rank 0 print the line it has and every rank leave."""
    import subprocess
    import sys

    import bench

    out = {"metric": "m", "value": 1.0}
    res = bench._guarded_sharded_leg(out, 0, lambda: (_ for _ in ()).throw(RuntimeError("recv failed")))
    assert res == {"error": "RuntimeError: recv failed"}
    assert bench._guarded_sharded_leg(out, 0, lambda: {"wall_s": 1.0}) == {"wall_s": 1.0}
    code = (
        "import json, time, sys; sys.path.insert(0, %r); import bench\n"
        "out = {'metric': 'm', 'value': 2.5}\n"
        "bench._guarded_sharded_leg(out, 0, lambda: time.sleep(60), timeout_s=0.5)\n"
        "print('not reached')\n" % ROOT
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "not reached" not in r.stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 2.5 and "did not return" in line["e2e_sharded"]["error"]
