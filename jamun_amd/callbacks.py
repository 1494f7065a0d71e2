"""Sampler callbacks: trajectory writer and timing.

``SaveTrajectoryCallback`` keeps the npy layout of the reference's ``SaveTrajectory`` metric
(``/root/reference/src/jamun/metrics/_save_trajectory.py:17-30,78-97``, ``metrics/_utils.py:84-97``): under
``sampler/<label>/predicted_samples/npy/`` one ``<chain>.npy`` of shape ``[n_atoms, frames, 3]`` per chain and
``joined.npy`` ``[n_atoms, chains*frames, 3]``; samples are dispatched by ``dataset_label``
(``callbacks/sampler/_utils.py:42-52``).  With several ranks the per-rank blocks are gathered to rank 0 once per batch
(one collective, ``jamun_amd.dist.gather_ragged``) — what torchmetrics' ``dist_reduce_fx="cat"`` does in the reference.
"""

from __future__ import annotations

import json
import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import dist


class SaveTrajectoryCallback:
    def __init__(self, datasets: Sequence, sample_key: str = "xhat_traj", output_dir: str = "sampler", write_pdb: bool = True, **_):
        labels = []
        self.datasets = {}
        for d in datasets:
            if d.label() not in self.datasets:
                labels.append(d.label())
                self.datasets[d.label()] = d
        self.labels = sorted(labels)
        self.sample_key = sample_key
        self.output_dir = output_dir
        self.write_pdb = write_pdb
        self.chains: Dict[str, List[np.ndarray]] = {l: [] for l in self.labels}  # per label: list of [n, T, 3]
        self.num_chains_seen = {l: 0 for l in self.labels}

    def _dir(self, label: str, ext: str) -> str:
        d = os.path.join(self.output_dir, label, "predicted_samples", ext)
        os.makedirs(d, exist_ok=True)
        return d

    def on_sample_start(self, sampler):
        pass

    def on_after_sample_batch(self, sample: Sequence[dict], sampler):
        for label in self.labels:
            mine = [s[self.sample_key] for s in sample if s.get("dataset_label") == label]
            unknown = [s.get("dataset_label") for s in sample if s.get("dataset_label") not in self.datasets]
            if unknown:
                raise KeyError(f"sample dataset label {unknown[0]!r} has no dataset")
            if mine:
                for m in mine:
                    if m.ndim != 3:
                        raise ValueError(f"Invalid sample shape: {tuple(m.shape)}, expected (num_atoms, num_frames, 3).")
                block = torch.stack(mine)  # [chains_local, n, T, 3]
            else:
                n = self.datasets[label].molecule["pos"].shape[0]
                block = torch.zeros(0, n, 1, 3, device=sampler.device)
            gathered = dist.gather_ragged(block.contiguous(), dst=0)
            if gathered is None:
                continue
            new = [c for g in gathered for c in g.detach().cpu().numpy()]
            start = len(self.chains[label])
            self.chains[label].extend(new)
            for i, arr in enumerate(new, start=start):
                np.save(os.path.join(self._dir(label, "npy"), f"{i}.npy"), arr)
            if self.chains[label]:
                joined = np.concatenate(self.chains[label], axis=1)  # "b n t c -> n (b t) c"
                np.save(os.path.join(self._dir(label, "npy"), "joined.npy"), joined)
            self.num_chains_seen[label] = len(self.chains[label])

    def on_sample_end(self, sampler):
        if not sampler.is_global_zero or not self.write_pdb:
            return
        from .pdb import write_pdb

        for label in self.labels:
            mol = getattr(self.datasets[label], "molecule", None)
            if mol is None or "atom_names" not in mol or not self.chains[label]:
                continue
            joined = np.concatenate(self.chains[label], axis=1)
            write_pdb(os.path.join(self._dir(label, "pdb"), "joined.pdb"), mol, torch.tensor(joined).permute(1, 0, 2))


class TrajectoryMetricCallback:
    """Feeds per-walker samples to one metric object per dataset label (``callbacks/sampler/_utils.py:22-56``).

    ``metric_fn(dataset=...)`` builds the meter of a dataset; a meter offers ``update(sample)``, ``compute() -> dict`` and
    the hooks ``on_sample_start / on_after_sample_batch / on_sample_end`` (and optionally ``to(device)``), i.e. the
    reference's ``TrajectoryMetric`` protocol, so metric classes written against it plug in unchanged.  Datasets are
    de-duplicated by label and ordered by label."""

    def __init__(self, datasets: Sequence, metric_fn):
        unique = {}
        for d in datasets:
            unique.setdefault(d.label(), d)
        self.meters = {label: metric_fn(dataset=unique[label]) for label in sorted(unique)}

    def on_sample_start(self, sampler):
        for meter in self.meters.values():
            if hasattr(meter, "to"):
                meter.to(sampler.fabric.device)
            meter.on_sample_start()

    def on_after_sample_batch(self, sample: Sequence, sampler):
        for sample_graph in sample:
            self.meters[sample_graph["dataset_label"]].update(sample_graph)
        for meter in self.meters.values():
            sampler.fabric.log_dict(meter.compute())
            meter.on_after_sample_batch()

    def on_sample_end(self, sampler):
        for meter in self.meters.values():
            meter.on_sample_end()


class MeasureSamplingTimeCallback:
    """Wall time per batch and per sampled conformation (one saved (walker, frame) pair — the reference's unit,
    ``callbacks/sampler/_measure_sampling_time.py:57,71``).  Writes ``sampler/timing.json`` on rank 0."""

    def __init__(self, output_dir: str = "sampler", **_):
        self.output_dir = output_dir
        self.t0: Optional[float] = None
        self.batches: List[dict] = []

    def on_sample_start(self, sampler):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.t0 = time.perf_counter()

    def on_after_sample_batch(self, sample: Sequence[dict], sampler):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_conf = sum(int(s["xhat_traj"].shape[1]) for s in sample if "xhat_traj" in s)
        self.batches.append({"batch": int(sampler.global_step), "seconds": t1 - self.t0, "conformations": n_conf})
        self.t0 = t1

    def on_sample_end(self, sampler):
        if not sampler.is_global_zero:
            return
        os.makedirs(self.output_dir, exist_ok=True)
        tot_s = sum(b["seconds"] for b in self.batches)
        tot_c = sum(b["conformations"] for b in self.batches)
        with open(os.path.join(self.output_dir, "timing.json"), "w") as f:
            json.dump({"batches": self.batches, "rank0_conformations_per_second": tot_c / tot_s if tot_s > 0 else None,
                       "ms_per_sample": 1e3 * tot_s / tot_c if tot_c else None, "world_size": sampler.world_size}, f, indent=1)
