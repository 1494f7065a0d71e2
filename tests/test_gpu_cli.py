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
