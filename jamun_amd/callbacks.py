"""Sampler callbacks: trajectory writer, metric dispatch and timing.

``SaveTrajectoryCallback`` writes the file set of the reference's ``SaveTrajectory`` metric
(``/root/reference/src/jamun/metrics/_save_trajectory.py:17-30,53-56,78-97``, ``metrics/_utils.py:84-113``) under
``sampler/<label>/``:

    topology.pdb                                   first frame of the dataset            (on_sample_start)
    predicted_samples/npy/<i>.npy   [n, T, 3] nm   one per chain                         (every batch)
    predicted_samples/pdb/<i>.pdb, dcd/<i>.dcd     the same chain as PDB models / CHARMM DCD (Angstrom)
    predicted_samples/{npy,pdb,dcd}/joined.*       all chains so far, frames concatenated: [n, chains*T, 3]

``analysis/load_trajectory.py:88-107`` needs ``dcd/joined.dcd`` plus ``topology.pdb`` (or ``pdb/0.pdb``).  Samples are
dispatched by ``dataset_label`` (``callbacks/sampler/_utils.py:42-52``) and validated against the dataset
(``metrics/_utils.py:15-28``).  With several ranks the per-rank blocks are gathered to rank 0 once per batch and label
(``jamun_amd.dist.gather_ragged``) — what torchmetrics' ``dist_reduce_fx="cat"`` does in the reference.
"""

from __future__ import annotations

import json
import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import dist
from .data import ATOM_TYPES


def validate_sample(sample, dataset) -> None:
    """``metrics/_utils.py:15-28``: the sample's label must be the dataset's and its atom types must be the topology's."""
    label = sample["dataset_label"] if "dataset_label" in sample else None
    if label != dataset.label():
        raise ValueError(f"Sample dataset label {label} does not match expected label {dataset.label()}.")
    mol = getattr(dataset, "molecule", None)
    if mol is None or "atom_type_index" not in sample:
        return
    name = lambda i: ATOM_TYPES[i] if 0 <= i < len(ATOM_TYPES) else "?"
    expected = [name(int(i)) for i in mol["atom_type_index"]]
    actual = [name(int(i)) for i in sample["atom_type_index"]]
    if expected != actual:
        raise ValueError(f"Atom types in init_graph ({actual}) do not match expected atom types in structure ({expected}).")


class SaveTrajectoryCallback:
    """The file set of the reference's ``SaveTrajectory`` metric (``metrics/_save_trajectory.py:17-97``) under
    ``<output_dir>/<label>/``: ``topology.pdb``; ``predicted_samples/{npy,pdb,dcd}/<i>.*`` per chain and ``joined.*``; with
    ``save_true_trajectory`` also ``true_samples/{pdb,dcd}/0.*`` = the dataset's own frames (``:22-26,58-62``).

    One deliberate deviation: the reference numbers the ``.pdb`` / ``.dcd`` files of a batch from the running chain count but
    restarts the ``.npy`` numbering at 0 in every batch (``enumerate(samples_np)`` at ``:81`` against ``start=self.num_chains_seen``
    at ``:89``), so its ``<i>.npy`` files of batch b overwrite those of batch b-1 and only ``joined.npy`` keeps everything.  Here all
    three use the running index; ``npy_index_restarts_per_batch=True`` reproduces the reference's files exactly."""

    def __init__(self, datasets: Sequence, sample_key: str = "xhat_traj", output_dir: str = "sampler", write_pdb: bool = True,
                 write_dcd: bool = True, save_true_trajectory: bool = False, npy_index_restarts_per_batch: bool = False,
                 async_write: bool = True, **_):
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
        self.write_dcd = write_dcd
        self.save_true_trajectory = save_true_trajectory
        self.npy_index_restarts_per_batch = npy_index_restarts_per_batch
        self.chains: Dict[str, List[np.ndarray]] = {l: [] for l in self.labels}  # per label: list of [n, T, 3]
        self.num_chains_seen = {l: 0 for l in self.labels}
        # Files are written on ONE side thread, in submission order, while the next batch walks on the GPU (the walk is a single native call
        # that releases the GIL; numpy's file writes release it too): a 20 000-step batch of 256 dipeptides is 1 GB per key, and its files
        # take as long to write as the GPU needs for the walk.  on_sample_end (and flush) wait for the writer; wait_s is that time.
        self.async_write = bool(async_write)
        self._pool = None
        self._pending: list = []
        self.wait_s = 0.0
        from . import dist

        self._stager = dist.HostStager()  # one pinned + one device staging buffer for the whole run (dist.gather_ragged_to_host)
        self.gather_timings: Dict[str, float] = {}  # gather_s / gather_bytes accumulated over batches and labels (bench.py: e2e_sharded)

    def _dir(self, label: str, ext: str) -> str:
        d = os.path.join(self.output_dir, label, "predicted_samples", ext)
        os.makedirs(d, exist_ok=True)
        return d

    def _mol(self, label: str) -> Optional[dict]:
        mol = getattr(self.datasets[label], "molecule", None)
        return mol if (mol is not None and "atom_names" in mol) else None

    def filename_pred(self, label: str, trajectory_index, extension: str) -> str:
        if extension not in ("npy", "pdb", "dcd"):
            raise ValueError(f"Invalid extension: {extension}")
        return os.path.join(self._dir(label, extension), f"{trajectory_index}.{extension}")

    def on_sample_start(self, sampler):
        if not sampler.is_global_zero:
            return
        from .pdb import save_pdb

        for label in self.labels:
            for ext in ("npy", "pdb", "dcd"):
                self._dir(label, ext)
            mol = self._mol(label)
            if mol is not None and self.write_pdb:  # topology from the dataset's first frame (_save_trajectory.py:53-56)
                save_pdb(os.path.join(self.output_dir, label, "topology.pdb"), mol, mol["pos"][None])
            if self.save_true_trajectory:  # the dataset's own frames as true_samples/{pdb,dcd}/0.* (_save_trajectory.py:22-26,58-62)
                from .pdb import save_dcd

                ds = self.datasets[label]
                xyz = getattr(ds, "xyz", None)
                frames = (xyz if xyz is not None else mol["pos"][None]).detach().cpu().numpy() if mol is not None or xyz is not None else None
                if frames is not None:
                    for ext in ("pdb", "dcd"):
                        os.makedirs(os.path.join(self.output_dir, label, "true_samples", ext), exist_ok=True)
                    if mol is not None:
                        save_pdb(os.path.join(self.output_dir, label, "true_samples", "pdb", "0.pdb"), mol, frames)
                    save_dcd(os.path.join(self.output_dir, label, "true_samples", "dcd", "0.dcd"), frames)

    def _write_chain(self, label: str, index, arr: np.ndarray, npy_index=None) -> None:
        """arr [n, T, 3] nm -> <index>.npy / .pdb / .dcd"""
        from .pdb import save_dcd, save_pdb

        np.save(self.filename_pred(label, index if npy_index is None else npy_index, "npy"), arr)
        frames = np.transpose(arr, (1, 0, 2))  # "atoms frames coords -> frames atoms coords" (utils/mdtraj.py:17-21)
        mol = self._mol(label)
        if self.write_pdb and mol is not None:
            save_pdb(self.filename_pred(label, index, "pdb"), mol, frames)
        if self.write_dcd:
            save_dcd(self.filename_pred(label, index, "dcd"), frames)

    def _raise_finished_writer_errors(self) -> None:
        """A writer failure (disk full, permission, bad shape) must stop the run at the NEXT batch, not after the last one."""
        still = []
        for fut in self._pending:
            if fut.done():
                fut.result()
            else:
                still.append(fut)
        self._pending = still

    def on_after_sample_batch(self, sample: Sequence[dict], sampler):
        self._raise_finished_writer_errors()
        for s in sample:
            if s.get("dataset_label") not in self.datasets:
                raise KeyError(f"sample dataset label {s.get('dataset_label')!r} has no dataset")
        for label in self.labels:
            mine = [s for s in sample if s.get("dataset_label") == label]
            for s in mine:
                validate_sample(s, self.datasets[label])
                if s[self.sample_key].ndim != 3:
                    raise ValueError(f"Invalid sample shape: {tuple(s[self.sample_key].shape)}, expected (num_atoms, num_frames, 3).")
            # a rank without walkers of this label contributes nothing; gather_ragged agrees on the trailing shape first
            block = torch.stack([s[self.sample_key] for s in mine]).contiguous() if mine else None  # [chains_local, n, T, 3]
            # one block at a time through ONE reusable device receive buffer and ONE reusable pinned staging buffer; what the writer
            # thread (and self.chains) keep are pageable copies
            blocks = dist.gather_ragged_to_host(block, dst=0, device=sampler.device, stager=self._stager, timings=self.gather_timings)
            if blocks is None:
                continue
            start = self.num_chains_seen[label]
            self.num_chains_seen[label] = start + sum(int(b.shape[0]) for b in blocks)
            self._submit(self._write_batch, label, blocks, start)

    def _write_batch(self, label: str, blocks: List[np.ndarray], start: int) -> None:
        new = [c for b in blocks for c in b]
        self.chains[label].extend(new)
        for i, arr in enumerate(new, start=start):
            self._write_chain(label, i, arr, npy_index=(i - start) if self.npy_index_restarts_per_batch else None)
        if self.chains[label]:
            self._write_chain(label, "joined", np.concatenate(self.chains[label], axis=1))  # "b n t c -> n (b t) c"

    def _submit(self, fn, *args) -> None:
        if not self.async_write:
            fn(*args)
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="jamun-writer")
        self._pending.append(self._pool.submit(fn, *args))

    def flush(self) -> None:
        """Wait for the files of every batch handed over so far (errors of the writer thread surface here)."""
        t0 = time.perf_counter()
        pending, self._pending = self._pending, []
        for fut in pending:
            fut.result()
        self.wait_s += time.perf_counter() - t0

    def on_sample_end(self, sampler):
        try:
            self.flush()  # (the reference only uploads the joined files to wandb here, _save_trajectory.py:64-76: out of scope)
        finally:
            self.close()

    def close(self) -> None:
        """Stop the writer thread (also called by Sampler.sample when the loop raises: pending writes finish, their errors are not re-raised
        over the original one)."""
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None


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
        self.datasets = unique
        self.meters = {label: metric_fn(dataset=unique[label]) for label in sorted(unique)}

    def on_sample_start(self, sampler):
        for meter in self.meters.values():
            if hasattr(meter, "to"):
                meter.to(sampler.fabric.device)
            meter.on_sample_start()

    def on_after_sample_batch(self, sample: Sequence, sampler):
        for sample_graph in sample:
            validate_sample(sample_graph, self.datasets[sample_graph["dataset_label"]])  # TrajectoryMetric.update, metrics/_utils.py:64
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
