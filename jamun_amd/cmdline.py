"""``jamun_sample`` — entry point of the sampling path (mirror of ``/root/reference/src/jamun/cmdline/sample.py:41-138``).

    jamun_sample --config-dir=configs experiment=sample_custom ++init_pdbs=[x.pdb] ++checkpoint_dir=ckpts \
                 num_sampling_steps_per_batch=50 repeat_init_samples=4

Accepts the reference's command-line form (``--config-dir``, ``experiment=...``, ``key=value``, ``++key=value``) and its
config keys.  Differences, all explicit: no wandb (``wandb_train_run_path`` must be null; use ``checkpoint_dir``),
``finetune_on_init`` must be null/false (training is out of scope), ``sampler.accelerator=cpu`` /
``trainer.accelerator=cpu`` is rejected (there is no CPU path).
"""

from __future__ import annotations

import logging
import os
import sys
from typing import List, Optional, Sequence

import torch

from . import config as C
from . import dist
from .checkpoint import find_checkpoint
from .data import WalkerBatch

log = logging.getLogger("jamun")


def get_initial_graphs(datasets: Sequence, num_init_samples_per_dataset: int, repeat: int = 1) -> WalkerBatch:
    """``cmdline/sample.py:27-38``: per dataset ``randperm(len)[:k]`` frames, each repeated ``repeat`` times (consecutive)."""
    mols, labels = [], []
    for dataset in datasets:
        random_indices = torch.randperm(len(dataset))[:num_init_samples_per_dataset].tolist()
        # The reference draws this before any seed is set (sample.py:33 vs :86-88), so ranks may pick different frames.  Here
        # rank 0's draw is used on every rank: with sharded walkers the shards must partition ONE initial batch.
        random_indices = dist.broadcast_object(random_indices, src=0)
        for index in random_indices:
            g = dataset[int(index)]
            for _ in range(repeat):
                mols.append(g)
                labels.append(dataset.label())
    return WalkerBatch.from_molecules(mols, labels)


def run(cfg: dict):
    cfg = C.resolve(cfg, throw_on_missing=True)
    if cfg.get("finetune_on_init"):
        raise NotImplementedError("finetune_on_init is a training feature and is out of scope; pass finetune_on_init=null")
    if isinstance(cfg.get("trainer"), dict) and "accelerator" in cfg["trainer"]:
        cfg["sampler"]["accelerator"] = cfg["trainer"]["accelerator"]  # BASELINE.json's spelling of sampler.accelerator
    checkpoint_path = find_checkpoint(
        wandb_train_run_path=cfg.get("wandb_train_run_path"),
        checkpoint_dir=cfg.get("checkpoint_dir"),
        checkpoint_type=cfg.get("checkpoint_type"),
    )
    cfg["model"]["checkpoint_path"] = checkpoint_path
    model = C.instantiate(cfg["model"])
    init_datasets = C.instantiate(cfg["init_datasets"])
    callbacks = []
    for _, node in (cfg.get("callbacks") or {}).items():
        if isinstance(node, dict) and "_target_" in node:
            # ${init_datasets} resolved to the config node; hand the callback the instantiated datasets instead
            extra = {"datasets": init_datasets} if "datasets" in node else {}
            callbacks.append(C.instantiate({k: v for k, v in node.items() if k != "datasets"}, **extra))
    init_graphs = get_initial_graphs(init_datasets, cfg["num_init_samples_per_dataset"], cfg.get("repeat_init_samples", 1))
    sampler = C.instantiate(cfg["sampler"], callbacks=callbacks, loggers=None)
    batch_sampler = C.instantiate(cfg["batch_sampler"])
    if seed := cfg.get("seed"):
        torch.manual_seed(seed + sampler.fabric.global_rank)  # ranks generate different chains (sample.py:86-88)
    log.info("sampling %d walkers (%d atoms) on %s", init_graphs.num_graphs, init_graphs.num_nodes, sampler.device)
    sampler.sample(model=model, batch_sampler=batch_sampler, init_graphs=init_graphs, num_batches=cfg["num_batches"],
                   continue_chain=cfg["continue_chain"])
    return sampler


def compose(argv: List[str], cwd: Optional[str] = None) -> dict:
    cwd = cwd or os.getcwd()
    dirs, overrides = [], []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a.startswith("--config-dir="):
            dirs.append(a.split("=", 1)[1])
        elif a in ("--config-dir", "-cd"):
            dirs.append(argv[i + 1])
            i += 1
        elif a.startswith("--"):
            raise SystemExit(f"unsupported flag {a}")
        else:
            overrides.append(a)
        i += 1
    composer = C.Composer([os.path.abspath(os.path.join(cwd, d)) for d in dirs] + [C.BUILTIN_DIR])
    cfg = composer.compose("sample", overrides)
    # make user paths absolute before changing into the run directory (hydra job.chdir = true)
    if isinstance(cfg.get("init_pdbs"), list):
        cfg["init_pdbs"] = [p if os.path.isabs(p) or "${" in str(p) else os.path.join(cwd, p) for p in cfg["init_pdbs"]]
    ck = cfg.get("checkpoint_dir")
    if isinstance(ck, str) and ck != C.MISSING and not os.path.isabs(ck) and "${" not in ck:
        cfg["checkpoint_dir"] = os.path.join(cwd, ck)
    return cfg


def main(argv: Optional[List[str]] = None):
    argv = list(sys.argv[1:] if argv is None else argv)
    logging.basicConfig(level=logging.INFO)
    dist.init_process_group()
    cwd = os.getcwd()
    cfg = compose(argv, cwd)
    head = C.resolve({k: cfg[k] for k in ("paths", "task_name", "run_group", "run_key")})
    run_dir = head["paths"]["run_path"]
    run_dir = run_dir if os.path.isabs(run_dir) else os.path.join(cwd, run_dir)
    os.makedirs(run_dir, exist_ok=True)
    os.chdir(run_dir)
    try:
        run(cfg)
    finally:
        os.chdir(cwd)
    return run_dir


if __name__ == "__main__":
    main()
