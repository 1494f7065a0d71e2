"""GPU test of BASELINE.json configs[0]: `jamun_sample experiment=sample_custom` on one AG dipeptide .pdb,
4 walkers x 50 walk-jump steps, through the config tree, checkpoint file, PDB reader, sampler and trajectory writer."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_jamun_sample_cfg1_end_to_end(tmp_path, monkeypatch):
    from jamun_amd import cmdline, pdb, synth

    mol = dict(synth.ag_dipeptide(), elements=["N", "C", "C", "C", "O", "N", "C", "C", "O", "O"], residue_ids=[1] * 5 + [2] * 5)
    pdb_path = str(tmp_path / "uncapped_AG.pdb")
    pdb.write_pdb(pdb_path, mol, mol["pos"][None])
    ck_dir = tmp_path / "ckpt"
    ck_dir.mkdir()
    torch.save(synth.synthetic_checkpoint(output_gain=0.05, prefix="g._orig_mod."), str(ck_dir / "epoch=7-step=100.ckpt"))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("JAMUN_ROOT_PATH", str(tmp_path))
    run_dir = cmdline.main(["--config-dir=" + os.path.join(ROOT, "configs"), "experiment=sample_custom", f"++init_pdbs=[{pdb_path}]",
                            f"++checkpoint_dir={ck_dir}", "checkpoint_type=best_so_far", "wandb_train_run_path=null", "finetune_on_init=null",
                            "num_sampling_steps_per_batch=50", "repeat_init_samples=4", "num_batches=2", "++sampler.rng=torch_cpu"])
    npy = os.path.join(run_dir, "sampler", "uncapped_AG", "predicted_samples", "npy")
    chains = [np.load(os.path.join(npy, f"{i}.npy")) for i in range(8)]  # 4 walkers x 2 batches
    assert all(c.shape == (10, 50, 3) and np.isfinite(c).all() for c in chains)
    joined = np.load(os.path.join(npy, "joined.npy"))
    assert joined.shape == (10, 8 * 50, 3)
    assert np.array_equal(joined[:, :50], chains[0])
    # walkers of the same molecule diverge from each other; samples stay near the molecule (nm scale)
    assert np.abs(chains[0] - chains[1]).max() > 1e-3 and np.abs(joined).max() < 5.0
    t = json.load(open(os.path.join(run_dir, "sampler", "timing.json")))
    assert len(t["batches"]) == 2 and t["batches"][0]["conformations"] == 4 * 50
    assert os.path.exists(os.path.join(run_dir, "sampler", "uncapped_AG", "predicted_samples", "pdb", "joined.pdb"))
    # accelerator=cpu is refused: there is no CPU path
    with pytest.raises(RuntimeError, match="no CPU path"):
        cmdline.main(["--config-dir=" + os.path.join(ROOT, "configs"), "experiment=sample_custom", f"++init_pdbs=[{pdb_path}]",
                      f"++checkpoint_dir={ck_dir}", "checkpoint_type=best_so_far", "num_sampling_steps_per_batch=3", "++trainer.accelerator=cpu"])


def test_jamun_sample_uncapped_2aa_directory_end_to_end(tmp_path, monkeypatch):
    """BASELINE.json configs[1] as the reference runs it (configs/experiment/sample_uncapped_2AA.yaml:8-19): init_datasets =
    jamun.data.parse_datasets_from_directory over a Timewarp-style test directory, ONE walker per distinct dipeptide (ragged
    9..29-atom batch), SaveTrajectory files per dataset label.  28 dipeptides covering all 20 residue types."""
    from jamun_amd import cmdline, synth

    codes = ["AG", "GG", "WW", "FY", "KR", "PH", "CM", "DE", "NQ", "ST", "IL", "VA", "RK", "HP", "MC", "ED", "QN", "TS", "LI", "YF",
             "GW", "WA", "AV", "SG", "PP", "KE", "DR", "HH"]
    root = tmp_path / "data" / "timewarp" / "2AA-1-large" / "test"
    mols = synth.write_timewarp_tree(str(root), codes, n_frames=2)
    assert len({a for c in codes for a in c}) == 20
    ck_dir = tmp_path / "ckpt"
    ck_dir.mkdir()
    torch.save(synth.synthetic_checkpoint(output_gain=0.05), str(ck_dir / "epoch=3-step=10.ckpt"))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("JAMUN_ROOT_PATH", str(tmp_path))
    monkeypatch.setenv("JAMUN_DATA_PATH", str(tmp_path / "data"))
    args = ["--config-dir=" + os.path.join(ROOT, "configs"), "experiment=sample_timewarp_2AA", f"++checkpoint_dir={ck_dir}",
            "num_sampling_steps_per_batch=12", "num_batches=2", "++sampler.rng=torch_cpu", "++init_datasets.num_frames=1"]
    run_dir = cmdline.main(args)
    sizes = {}
    for c in codes:
        d = os.path.join(run_dir, "sampler", c, "predicted_samples")
        a = np.load(os.path.join(d, "npy", "0.npy"))
        n = mols[c]["pos"].shape[0]
        assert a.shape == (n, 12, 3) and np.isfinite(a).all(), (c, a.shape)
        j = np.load(os.path.join(d, "npy", "joined.npy"))
        assert j.shape == (n, 24, 3)
        assert os.path.exists(os.path.join(d, "pdb", "joined.pdb")) and os.path.exists(os.path.join(d, "dcd", "joined.dcd"))
        assert os.path.exists(os.path.join(run_dir, "sampler", c, "topology.pdb"))
        sizes[c] = n
        # the chain starts at the data frame + sigma * noise and stays near it on this contractive checkpoint
        assert np.abs(a[:, 0] - mols[c]["pos"].numpy()).max() < 0.3
    assert min(sizes.values()) == 9 and max(sizes.values()) == 29
    t = json.load(open(os.path.join(run_dir, "sampler", "timing.json"))) if os.path.exists(os.path.join(run_dir, "sampler", "timing.json")) else None
    assert t is None or t["batches"][0]["conformations"] == len(codes) * 12
    # max_datasets (data/_utils.py:98-99) through the command line: the first three codes in sorted order, nothing else
    run2 = cmdline.main(args + ["++init_datasets.max_datasets=3", "run_key=second"])
    first = sorted(codes)[:3]
    assert sorted(x for x in os.listdir(os.path.join(run2, "sampler")) if x != "timing.json") == first
    for c in first:
        b = np.load(os.path.join(run2, "sampler", c, "predicted_samples", "npy", "0.npy"))
        assert b.shape == (sizes[c], 12, 3) and np.isfinite(b).all()


_RCCL_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import torch.distributed as td
from jamun_amd import dist, synth
from jamun_amd.callbacks import SaveTrajectoryCallback
from jamun_amd.data import WalkerBatch
from jamun_amd.model import Denoiser
from jamun_amd.sampling import BAOAB, Sampler, SingleMeasurementSampler
torch.cuda.set_device(0)
td.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[3], rank=0, world_size=1)
assert td.get_backend() == "nccl" and dist.rank_world() == (0, 1)
dev = dist.local_device()
# the collectives of the gather path on RCCL: device metadata all-gather, a CPU block staged on the GPU, an empty rank, the
# object broadcast and the barrier with a named device
got = dist.gather_ragged(torch.full((3, 2, 3), 7.0, device=dev))
assert len(got) == 1 and got[0].is_cuda and float(got[0].mean()) == 7.0
got = dist.gather_ragged(torch.ones(2, 3))  # cpu_offload keeps trajectories on the host
assert len(got) == 1 and got[0].is_cuda
assert dist.gather_ragged(None) == []
assert dist.broadcast_object([3, 1, 2]) == [3, 1, 2]
dist.barrier()
try:
    dist.gather_ragged(torch.zeros(2, 3, dtype=torch.float16, device=dev))
    raise SystemExit("expected ValueError")
except ValueError as e:
    assert "dtype" in str(e)
# Sampler.sample(shard_walkers=True) with the real denoiser inside the initialised group -> files
mol = synth.random_chain(9, seed=3)
class DS:
    molecule = dict(mol)
    def label(self): return "m"
model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint(output_gain=0.05))
batch = WalkerBatch.from_molecules([mol] * 3, labels=["m"] * 3)
out_dir = os.path.join(sys.argv[2], "sampler")
cb = SaveTrajectoryCallback([DS()], output_dir=out_dir, write_pdb=False)
bs = SingleMeasurementSampler(mcmc=BAOAB(delta=0.04, friction=1.0, steps=6, save_trajectory=True), sigma=0.04)
Sampler(callbacks=[cb], shard_walkers=True).sample(model=model, batch_sampler=bs, num_batches=2, init_graphs=batch, continue_chain=True)
dist.barrier()
j = np.load(os.path.join(out_dir, "m", "predicted_samples", "npy", "joined.npy"))
assert j.shape == (9, 3 * 2 * 6, 3) and np.isfinite(j).all(), j.shape
print(json.dumps({"ok": True}))
td.destroy_process_group()
'''


def test_sharded_sampler_inside_an_rccl_group(tmp_path):
    """The N > 1 plumbing on the backend the GPUs use: a one-rank RCCL ("nccl") group — a 1-GPU box cannot host two ranks,
    RCCL refuses two ranks on one device — runs the metadata all-gather, the host-block staging, the object broadcast, the
    device-named barrier and Sampler.sample(shard_walkers=True) through to the written files.  (The two-rank logic is covered on
    gloo in tests/test_host.py.)"""
    import socket
    import subprocess
    import sys

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path), str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["ok"], r.stdout[-1000:]


def test_bench_line_contract_and_roofline_invariants():
    """`python bench.py` (short run, no CPU baseline): ONE JSON line with the driver's keys; the roofline block prices the dominant kernel's
    EXECUTED f16 MFMA FLOPs against the dense f16 peak (at most 1 by construction), the useful part is a part of them, and the counted
    bytes stand beside the algorithmic ones."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--repeats", "3", "--no-cpu-baseline", "--no-secondary"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["steps"] == 4 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    assert abs(d["value"] - 256 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["peak"] == 2500.0 and rf["unit"] == "TFLOP/s"
    assert 0.05 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert 0.0 < rf["frac_useful"] < rf["frac"] and rf["flop_useful_per_launch"] < rf["flop_executed_per_launch"]
    assert rf["flop_executed_per_launch"] == 136 * 65 * 414 * 32768  # tiles x hidden units x MFMAs per (tile, unit) x FLOP per v_mfma_f32_32x32x16_f16
    assert rf["bytes_algorithmic_per_launch"] > 5e7 and (rf["traffic"] is None or rf["traffic_ratio"] > 1.0)
    # the shapes the reference actually runs ride in the same line: distinct dipeptides, the 93-heavy-atom chignolin shape, the opt-in f16x1 mode
    assert set(d["also"]) == {"cfg2r", "cfg5h", "cfg2_f16x1"}
    for name, leg in d["also"].items():
        assert leg["value"] > 0 and leg["timed_total_s"] >= 2.0 and leg["avg_launch_ms"] > 0, (name, leg)
    assert d["also"]["cfg2r"]["dominant_kernel"] == "k_conv_mf" and d["also"]["cfg2r"]["init_path"] == 4 and d["also"]["cfg5h"]["dominant_kernel"] == "k_conv_ml"
    assert d["also"]["cfg2_f16x1"]["dtype"] == "f16x1" and d["also"]["cfg2_f16x1"]["value"] > d["value"] and d["also"]["cfg2r"]["dtype"] == "f32"
    assert d["rccl"] == {"backend": None, "world": 1, "ranks_seen": 1}  # (no launcher: no process group)


def test_bench_sharded_leg_on_a_one_rank_rccl_group():
    """`bench.py --sharded-leg` creates a one-rank RCCL group and runs the leg every N > 1 line carries: Sampler.sample(shard_walkers=True) +
    SaveTrajectoryCallback with the trajectory gather (here: nobody to receive from, the own block through the pinned staging buffer), the
    all-reduce that proves the ranks, and the keys the 8-GPU line will have."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sharded-leg", "--steps", "4", "--warmup", "1", "--repeats", "2", "--no-cpu-baseline",
                        "--no-secondary", "--no-sweep", "--no-also", "--no-e2e"], env=dict(env, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["rccl"]["backend"].startswith("nccl") and d["rccl"]["world"] == 1 and d["rccl"]["ranks_seen"] == 1
    sh = d["e2e_sharded"]
    assert sh["walkers_total"] == 256 and sh["steps_per_batch"] == 1000 and sh["num_batches"] == 2 and sh["joined_shape"] == [17, 256 * 2 * 1000, 3]
    assert sh["gather_bytes"] == 0 and 0 < sh["gather_s"] < sh["wall_s"] and sh["conformations_per_s"] > 1e5
