#!/usr/bin/env python3
"""Benchmark of the walk-jump hot path: sampled conformations / second, whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5]

One "step" = one pass of the hot path over the walker batch: one denoiser forward + the BAOAB state update + one saved
frame for every walker (save_every_n_steps = 1), i.e. one conformation per walker per step — the unit the reference calls
a "sample" (/root/reference/src/jamun/callbacks/sampler/_measure_sampling_time.py:57,71).  A walk with ``steps = K``
evaluates exactly K forwards and saves K frames (frame 0 is the initial state, functional/_splitting.py:136-155).

Workloads (per GPU; weak scaling: the walker count grows with the number of GPUs), default e3conv architecture,
sigma = delta = 0.04, friction 1, M 1, clip 100, synthetic molecules + seeded synthetic checkpoint (no datasets / published
checkpoints are reachable offline):
    cfg2  BASELINE.json configs[1]  uncapped-2AA shape: 17 heavy atoms x 256 walkers              (the default, the metric's config)
    cfg2r the same config as the reference runs it: one walker per DISTINCT dipeptide (9..29 atoms, real topology, 143 embedding rows)
    cfg3  configs[2]                uncapped-4AA shape: 33 atoms x 256 walkers per GPU (2048 on 8)
    cfg4  configs[3]                MDGen-4AA-like ragged batch: 256 walkers of 17..57 atoms
    cfg5  configs[4]                chignolin size with hydrogens: 166 atoms x 64 walkers per GPU (512 on 8)
    cfg5h configs[4] as the reference feeds it (heavy atoms only, data/_mdtraj.py:60,218): 93 atoms x 128 walkers per GPU
Arithmetic is fp32 end to end — VALU work in fp32, the dominant contraction either on v_mfma_f32_32x32x2_f32 (jamun_tuning.dg_fp32) or, by
default, as "f16x3": each fp32 operand split exactly into two f16 terms, three f16 MFMAs per product with fp32 accumulation, error at
the level of one fp32 rounding per product.  The reference's sampling precision is "32-true" and its bf16 mode is
undefined, and the 1e-5 nm parity bar needs fp32-level accuracy.

Timing: W untimed warm-up steps, then the K-step walk is timed R times back to back — each repeat bracketed by a barrier +
torch.cuda.synchronize() on both sides and taken as the MAX over ranks — until >= ~10 s of timed work have accumulated
(a 20-step walk lasts 20 ms, below what an external GPU-busy sampler resolves); ``ms_per_step`` and ``value`` are the
MEDIAN repeat.  ``--repeats R`` fixes R.  The CPU baseline (rank 0, N = 1) runs FIRST, so that the GPU legs are the tail of
the run and an outside observer sampling GPU activity sees them.

Multi-GPU: one process per GPU.  Either launch with ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``
or just ``python bench.py --gpus N``: without WORLD_SIZE in the environment the parent spawns the N ranks itself (before any
GPU call is made) and relays rank 0's JSON line.  Walkers are sharded, there is no data-path collective; value = all ranks'
conformations / max-over-ranks time.
"""
import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIGMA = 0.04
MCMC = dict(delta=0.04, friction=1.0, M=1.0, inverse_temperature=1.0, score_fn_clip=100.0)
F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
HBM_PEAK_GBS = 8000.0         # same guide, "HBM3E peak BW" (spec; ~6.3 TB/s measured on a float4 copy)
PROF_EVERY = 7  # HIP events around every 7th launch of the dominant kernel inside the timed region (jamun_profile_sample)
PROF_READ = 1  # ... and read back after every repeat (deferring the reads was slower: the event pool keeps growing inside the timed region)
MIN_TIMED_S = 10.0

CONFIGS = {
    "cfg2": dict(baseline="configs[1]", desc="uncapped-2AA-like 17-atom molecule", atoms=17, walkers=256),
    "cfg2r": dict(baseline="configs[1] as the reference runs it (sample_uncapped_2AA.yaml:18-19: ONE walker per distinct test peptide)",
                  desc="256 DISTINCT dipeptides with real topology (9..29 heavy atoms, all 20 residue types, 143 distinct embedding rows)", atoms=None, walkers=256),
    "cfg3": dict(baseline="configs[2]", desc="uncapped-4AA-like 33-atom molecule (2048 walkers over 8 GPUs)", atoms=33, walkers=256),
    "cfg4": dict(baseline="configs[3]", desc="MDGen-4AA-like ragged batch, 17..57 atoms per walker", atoms=None, walkers=256),
    "cfg5": dict(baseline="configs[4]", desc="chignolin-size 166-atom molecule with hydrogens (512 walkers over 8 GPUs)", atoms=166, walkers=64),
    "cfg5h": dict(baseline="configs[4] as the reference feeds it (data/_mdtraj.py:60,218: hydrogens stripped)", desc="chignolin-size 93-heavy-atom molecule", atoms=93, walkers=128),
}


def workload_molecules(cfg: str, walkers: int, atoms=None, rank: int = 0):
    """The walker batch of one rank.  Equal molecules are consecutive (as get_initial_graphs' repeat, cmdline/sample.py:36)."""
    from jamun_amd import synth

    c = CONFIGS[cfg]
    n = atoms if atoms is not None else c["atoms"]
    if cfg == "cfg2r":  # every walker its own dipeptide: codes spread evenly over the 400, a different stretch per rank
        codes = synth.all_dipeptides()
        return [synth.peptide(codes[(i * len(codes) // max(walkers, 1) + 7 * rank) % len(codes)], seed=i + 1000 * rank) for i in range(walkers)]
    if n is not None:
        return [synth.random_chain(n, seed=0)] * walkers
    import random

    rng = random.Random(1234 + rank)  # ragged: sizes ~ U{17..57}, 8 distinct sequences per rank, each repeated
    kinds = [synth.random_chain(rng.randint(17, 57), seed=100 + i + 16 * rank) for i in range(8)]
    per = -(-walkers // len(kinds))
    return [m for m in kinds for _ in range(per)][:walkers]


# ---- CPU baseline (rank 0, N = 1 only) ---------------------------------------------------------------------------------


def _cpu_threads():
    import torch

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, 16)  # these small-tensor ops get slower, not faster, with hundreds of threads (measured: 256 threads 40x slower than 8)
    torch.set_num_threads(cores)
    return cores


def _cpu_walk(mols, steps):
    import torch

    from jamun_amd import synth
    from oracle import denoiser as od
    from oracle import graph as og
    from oracle import walk as ow

    topo = og.collate([{k: v for k, v in m.items() if torch.is_tensor(v)} for m in mols])
    sd = synth.synthetic_state_dict()
    hp = od.default_hparams()
    noise = ow.TorchNoise(42)
    y0 = topo["pos"] + noise(topo["pos"]) * SIGMA
    score_fn = lambda y: od.score(y, topo, SIGMA, sd, hp)
    xhat_fn = lambda y: od.xhat(y, topo, SIGMA, sd, hp)
    score_fn(y0)  # warm-up (thread pools, allocator)
    t0 = time.perf_counter()
    ow.walk_jump(score_fn, xhat_fn, ow.baoab, y0, "gaussian", noise, steps=steps, save_trajectory=True, **MCMC)
    return time.perf_counter() - t0


def cpu_baseline(cfg: str, sample_walkers=8, frames=8, cfg1_exact=True):
    """The CPU oracle (op-for-op PyTorch restatement of the reference path, kind="port") timed on this box's host cores:
    a BOUNDED SAMPLE of the benchmarked workload — `sample_walkers` walkers (not the 256 of the GPU line: at ~2.5
    conformations/s the full batch would take half an hour) x `frames` walk-jump frames after a warm-up forward; each frame
    costs two forwards, as the reference — and, BASELINE.md section 2, BASELINE.json configs[0] EXACTLY: AG dipeptide,
    4 walkers x 50 walk-jump steps."""
    from jamun_amd import synth

    cores = _cpu_threads()
    mols = workload_molecules(cfg, sample_walkers)
    # bounded by ATOMS as well (measured: 136 atoms x 8 frames = 28 s; 1328 atoms of the 166-atom workload x 8 frames took 453 s): the
    # first walkers up to ~140 atoms in total — at least one — and 4 frames when that one walker alone is larger
    n_keep, tot = 0, 0
    for m in mols:
        if n_keep and tot + m["pos"].shape[0] > 140:
            break
        n_keep, tot = n_keep + 1, tot + m["pos"].shape[0]
    mols, sample_walkers = mols[:n_keep], n_keep
    if tot > 140 or (n_keep == 1 and tot > 80):
        frames = min(frames, 4)
    dt = _cpu_walk(mols, frames)
    out = {
        "value": sample_walkers * frames / dt,
        "unit": "conformations/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{sample_walkers} walkers x {frames} walk-jump frames of the {cfg} workload ({sum(m['pos'].shape[0] for m in mols)} atoms), fp32, {dt:.1f} s",
    }
    if cfg1_exact:
        n1 = 50 if os.environ.get("JAMUN_BENCH_CFG1_FULL") else 20
        dt1 = _cpu_walk([synth.ag_dipeptide()] * 4, n1)
        out["cfg1_exact"] = {"value": 4 * n1 / dt1, "unit": "conformations/s", "cores": cores, "kind": "port",
                             "sample": f"BASELINE configs[0]: AG dipeptide, 4 walkers x {n1} walk-jump steps"
                                       + (" (the configuration exactly)" if n1 == 50 else " (the first 20 of its 50 steps — every step costs the same two forwards; JAMUN_BENCH_CFG1_FULL=1 runs all 50)")
                                       + f", fp32, {dt1:.1f} s"}
    return out


# ---- secondary rooflines: the HBM-bound kernels north_star names --------------------------------------------------------


def _time_launches(fn, iters, warm=5):
    import torch

    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()  # the stand-alone operators launch on torch's current stream, which these events see
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def secondary_rooflines(dev):
    """HIP-event timing, in this process, of the HBM-bound kernels of the path at the 256 x 17 and 2048 x 17 walker shapes:
    `k_scatter_mean` (torch_scatter mean over [E,248] messages, _conv.py:117; algorithmic bytes = E*992 read + N*992 written +
    CSR pointers) and the Langevin update kernels `k_baoab_pre` / `k_baoab_post` (60 B/atom/step + 36 B/atom/saved frame,
    SURVEY.md section 8d).  Back-to-back launches; buffers above the 256 MiB Infinity Cache are cycled so reads come from HBM."""
    import torch

    from jamun_amd import native

    out = []
    params = native.make_mcmc_params(2, **MCMC)
    # the STAND-ALONE Langevin update operators (jamun_baoab_pre / jamun_baoab_post) at a size where they are bandwidth- and not
    # launch-bound.  They are exported and parity-pinned, but OFF the hot path: in the walk both halves run inside k_geom / k_finalize
    # (at the bench shape, 4352 atoms = 0.3 MB, a separate launch would be pure latency)
    n_big = 1 << 24  # 201 MB per [n,3] array: far beyond the 256 MiB Infinity Cache in total
    y, v, psi, R, sc = (torch.randn(n_big, 3, device=dev) for _ in range(5))
    dt = _time_launches(lambda: native.baoab_pre(y, v, psi, R, params), 20)
    b = n_big * 3 * 4 * 6
    out.append({"kernel": "k_baoab_pre", "shape": f"stand-alone operator (fused into k_geom in the walk), {n_big} atoms, host-supplied noise", "bound": "hbm", "bytes": b,
                "avg_launch_ms": dt * 1e3, "achieved": b / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b / dt / 1e9 / HBM_PEAK_GBS})
    dt = _time_launches(lambda: native.baoab_post(v, psi, sc, params), 20)
    b = n_big * 3 * 4 * 4
    out.append({"kernel": "k_baoab_post", "shape": f"stand-alone operator (fused into k_finalize in the walk), {n_big} atoms, no frame save", "bound": "hbm", "bytes": b,
                "avg_launch_ms": dt * 1e3, "achieved": b / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b / dt / 1e9 / HBM_PEAK_GBS})
    del y, v, psi, R, sc
    for walkers in (256, 2048):
        n, deg = walkers * 17, 17
        E = n * deg
        seg = torch.arange(0, E + 1, deg, dtype=torch.int32, device=dev)
        nbytes = E * 248 * 4 + n * 248 * 4 + (n + 1) * 4
        n_buf = max(1, min(8, int(math.ceil((300 << 20) / (E * 248 * 4)))))  # rotate sources so the set exceeds the Infinity Cache
        srcs = [torch.randn(E, 248, device=dev) for _ in range(n_buf)]
        it = [0]

        def run():
            native.scatter_mean(srcs[it[0] % n_buf], seg, n)
            it[0] += 1

        dt = _time_launches(run, 40 if walkers == 256 else 12)
        out.append({"kernel": "k_scatter_mean", "shape": f"{walkers} walkers x 17 atoms, in-degree {deg}: [E={E},248] f32 -> [{n},248]",
                    "bound": "hbm", "bytes": nbytes, "avg_launch_ms": dt * 1e3, "achieved": nbytes / dt / 1e9, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": nbytes / dt / 1e9 / HBM_PEAK_GBS})
        del srcs
    return out


def build_signature(stats=None):
    """What a committed counter summary must share with the running build to be quoted: the digest of the library's sources and flags
    (jamun_amd/csrc/build.py) and the kernel selection of the config (jamun_stats)."""
    from jamun_amd.csrc import build as b

    sig = {"source_digest": b._digest()}
    if stats is not None:
        sig.update({k: int(stats.get(k, 0)) for k in ("conv_path", "dg_mode", "init_path", "mf_nks", "ml_window", "dg_emu", "n_tail_tiles")})
    return sig


def _pmc_traffic(kernel: str, cfg: str = "cfg2", tag: str = "", stats=None):
    """HBM-side bytes per launch of `kernel` from the newest committed PMC summary of this config (profiles/<round>_<cfg><tag>_pmc_traffic.json,
    written by profiles/collect.sh) — only if that summary was collected from THIS build: its `_build` record must equal
    build_signature().  Returns (bytes, file) or (None, stale file) or None."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), key=lambda f: (os.path.getmtime(f), f))
    files = [f for f in files if f"_{cfg}{tag}_" in os.path.basename(f)]
    want = build_signature(stats)
    stale = None
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("_build") != want:
            stale = stale or os.path.basename(f)
            continue
        for k, v in d.items():
            if isinstance(v, dict) and kernel in k and "FETCH_SIZE_KB_per_dispatch" in v:
                return (2.0 * v["FETCH_SIZE_KB_per_dispatch"] + v.get("WRITE_SIZE_KB_per_dispatch", 0.0)) * 1024.0, os.path.basename(f)
    return (None, stale) if stale else None


# ---- the metric as SURVEY.md section 8(d) defines it: wall time of Sampler.sample -----------------------------------------


class _BenchDataset:
    """What SaveTrajectoryCallback needs of a dataset: a label and the molecule (synthetic chains carry no atom names: .npy + .dcd)."""

    def __init__(self, mol, label):
        self.molecule, self._label = dict(mol, dataset_label=label), label

    def label(self):
        return self._label


def e2e_legs(model, mols, dev, steps_list=(1000, 20000), writer_steps=(1000, 20000), num_batches=2):
    """`Sampler.sample(num_batches=2, continue_chain=True)` through SingleMeasurementSampler / BAOAB — the call the reference's
    MeasureSamplingTimeCallback brackets (callbacks/sampler/_measure_sampling_time.py:57-71, sampling/_sampler.py:53-98) — at
    num_sampling_steps_per_batch = 1000 and 20000 (the shipped value, configs/experiment/sample_uncapped_2AA.yaml:16-17): wall time
    around sample() including the initial-noise draw, both batches, unbatch_samples and every callback; with and without
    SaveTrajectoryCallback (.npy + .dcd per chain and joined, written to a temporary directory that is removed afterwards)."""
    import shutil
    import tempfile

    import torch

    from jamun_amd.callbacks import SaveTrajectoryCallback
    from jamun_amd.data import WalkerBatch
    from jamun_amd.sampling import BAOAB, Sampler, SingleMeasurementSampler

    label = "bench"
    batch = WalkerBatch.from_molecules([dict(m, dataset_label=label) for m in mols])
    out = {"num_batches": num_batches, "continue_chain": True, "legs": []}
    t0 = time.perf_counter()
    model.sampler_for(batch.to(dev), SIGMA)  # (cached per (topology, sigma): the walks below reuse it)
    torch.cuda.synchronize()
    out["sampler_create_s"] = time.perf_counter() - t0
    for steps in sorted(set(steps_list) | set(writer_steps)):
        for with_writer in (False, True):
            if (with_writer and steps not in writer_steps) or (not with_writer and steps not in steps_list):
                continue
            tmp = None
            cbs = []
            if with_writer:
                n_bytes = 12 * batch.num_nodes * steps * num_batches * 6  # per-chain + joined files, .npy and .dcd, with slack
                tmp = tempfile.mkdtemp(prefix="jamun_bench_")
                if shutil.disk_usage(tmp).free < 2 * n_bytes:
                    out["legs"].append({"steps_per_batch": steps, "writer": True, "skipped": f"less than {2 * n_bytes >> 20} MiB free in {tmp}"})
                    shutil.rmtree(tmp, ignore_errors=True)
                    continue
                cbs = [SaveTrajectoryCallback([_BenchDataset(mols[0], label)], output_dir=tmp)]
            mcmc = BAOAB(steps=steps, save_trajectory=True, save_every_n_steps=1, v_init="gaussian", **MCMC)
            bs = SingleMeasurementSampler(mcmc=mcmc, sigma=SIGMA)
            sampler = Sampler(callbacks=cbs)
            torch.manual_seed(7)
            torch.cuda.reset_peak_memory_stats(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sampler.sample(model, bs, num_batches=num_batches, init_graphs=batch, continue_chain=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            leg = {"steps_per_batch": steps, "writer": bool(with_writer), "wall_s": dt,
                   "conformations_per_s": batch.num_graphs * steps * num_batches / dt, "ms_per_sample": 1e3 * dt / (batch.num_graphs * steps * num_batches),
                   "peak_device_mem_GB": torch.cuda.max_memory_allocated(dev) / 1e9}
            if with_writer:
                leg["writer_files"] = ".npy + .dcd per chain and joined (synthetic chains carry no atom names: no .pdb text)"
                leg["writer_wait_s"] = getattr(cbs[0], "wait_s", None)
                shutil.rmtree(tmp, ignore_errors=True)
            out["legs"].append(leg)
    return out


def batch_sweep(model, cfg, atoms, dev, rank, walker_counts=(256, 512, 1024, 2048), steps=20, target_s=2.0):
    """conformations/s of the fused walk on ONE GPU at growing walker counts (north_star's strong-scaling curve has 2048 walkers in total:
    256 per GPU at N = 8, all 2048 on one at N = 1): rate(256) / rate(2048) is the strong-scaling efficiency a perfect 8-GPU run would show."""
    import torch

    from jamun_amd import native
    from jamun_amd.data import WalkerBatch

    rows = []
    for w in walker_counts:
        mols = workload_molecules(cfg, w, atoms, rank)
        batch = WalkerBatch.from_molecules(mols).to(dev)
        smp = model.sampler_for(batch, SIGMA)
        torch.manual_seed(42)
        y = batch.pos + SIGMA * torch.randn_like(batch.pos)
        v = torch.randn_like(y)
        params = native.make_mcmc_params(steps, **MCMC)
        run = lambda: smp.walk("baoab", y, v, params, None, seed=99, save_trajectory=True)
        run()
        torch.cuda.synchronize()
        dts = []
        t_all = time.perf_counter()
        while time.perf_counter() - t_all < target_s or len(dts) < 3:
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        dt = statistics.median(dts)
        rows.append({"walkers": w, "atoms_total": batch.num_nodes, "conformations_per_s": w * steps / dt, "ms_per_step": 1e3 * dt / steps, "repeats": len(dts)})
        del smp, batch, y, v
    out = {"rows": rows}
    r = {x["walkers"]: x["conformations_per_s"] for x in rows}
    if 256 in r and 2048 in r:
        out["predicted_strong_scaling_efficiency_1_to_8"] = r[256] / r[2048]
        out["note"] = "rate(256 walkers) / rate(2048 walkers) on one GPU: what 8 GPUs x 256 walkers would reach of 8 x the one-GPU rate on 2048 walkers; predicted, not measured on 8 GPUs"
    return out


def also_legs(model, dev, rank, steps=20, target_s=2.0):
    """The shapes the reference actually runs, in the driver's line (N = 1): 20-step fused walks, >= `target_s` of timed work each, same
    timing as the headline (median repeat, synchronised on both sides) with HIP events around every 7th launch of the dominant conv kernel:
      cfg2r       configs[1] as sample_uncapped_2AA.yaml:8-19 runs it: one walker per DISTINCT dipeptide (k_conv_mfx initial projector)
      cfg5h       configs[4] as data/_mdtraj.py:60,218 feeds it: 93 heavy atoms x 128 walkers (k_conv_ml<96>)
      cfg2_f16x1  the headline shape with the OPT-IN reduced-precision conv (Sampler(precision="bf16-true")): configs[1] says "bf16"
    """
    import torch

    from jamun_amd import native
    from jamun_amd.data import WalkerBatch
    from jamun_amd.native import NativeSampler

    out = {}
    for name, cfg, tuning in (("cfg2r", "cfg2r", None), ("cfg5h", "cfg5h", None), ("cfg2_f16x1", "cfg2", {"f16x1": 1})):
        mols = workload_molecules(cfg, CONFIGS[cfg]["walkers"], None, rank)
        batch = WalkerBatch.from_molecules(mols).to(dev)
        smp = NativeSampler(model._native, SIGMA, batch, dev, tuning=tuning)
        torch.manual_seed(42 + rank)
        y = batch.pos + SIGMA * torch.randn_like(batch.pos)
        v = torch.randn_like(y)
        params = native.make_mcmc_params(steps, **MCMC)
        run = lambda: smp.walk("baoab", y, v, params, None, seed=1234 + rank, save_trajectory=True)
        run()
        torch.cuda.synchronize()
        smp.profile_enable(True, classes=["conv0", "conv1"], every=PROF_EVERY)
        dts, t_all = [], time.perf_counter()
        while time.perf_counter() - t_all < target_s or len(dts) < 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o = run()
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        prof = smp.profile_read()
        smp.profile_enable(False)
        assert torch.isfinite(o[2]).all()
        st = smp.stats()
        dt = statistics.median(dts)
        ms0, c0 = prof["conv0"]
        kname = "k_conv_mf" if st.get("dg_mode") == 4 else "k_conv_ml" if st.get("dg_mode") == 5 else f"k_conv_dg<{st.get('dg_mode')}>"
        out[name] = {"workload": f"{CONFIGS[cfg]['baseline']}: {CONFIGS[cfg]['desc']}, {batch.num_graphs} walkers" + (" — opt-in f16x1 conv, NOT the metric's precision" if tuning else ""),
                     "value": batch.num_graphs * steps / dt, "unit": "conformations/s", "ms_per_step": 1e3 * dt / steps, "timed_repeats": len(dts), "timed_total_s": sum(dts),
                     "dtype": "f16x1" if st.get("dg_emu") == 2 else "f32", "dominant_kernel": kname, "avg_launch_ms": ms0 / max(c0, 1), "launches_timed": c0,
                     "frac": (st.get("conv_flop_exec_launch", 0) / (ms0 / max(c0, 1) * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS) if c0 else None,
                     "init_path": st.get("init_path"), "atoms_total": batch.num_nodes}
        del smp, batch, y, v, o
    return out


class _StubModel:
    """--dry-run: what Sampler.sample touches of a model, without kernels (the rank plumbing, the sharding, the callback's gather and the
    writer are the REAL code; only the walk is replaced by a constant trajectory)."""

    def __init__(self, dev):
        self.device = dev

    def to(self, d):
        return self

    def eval(self):
        return self


class _StubBatchSampler:
    sigma = SIGMA

    def __init__(self, steps):
        import types

        self.steps, self.mcmc = steps, types.SimpleNamespace(rng="philox")

    def sample(self, model, y_init, v_init):
        import torch

        T = self.steps
        xt = y_init[None].expand(T, -1, -1).clone()
        return {"xhat": y_init, "y": y_init, "v": torch.zeros_like(y_init), "sample": y_init, "xhat_traj": xt, "y_traj": xt, "score_traj": xt, "t_traj": torch.ones(T)}


def _guarded_sharded_leg(out, rank, leg, timeout_s=240.0):
    """Runs the trajectory-exchange leg so that it can never cost the line its headline: the send / receive pairs between GPUs have only ever
    run on gloo and on a one-rank RCCL group (no multi-GPU node was available to the builder), so an exception becomes
    `{"error": ...}` in the line, and an exchange that does not return within `timeout_s` makes rank 0 print the line it has — the walk's
    numbers are final before this leg starts — and every rank leave without a barrier."""
    import threading

    done = threading.Event()

    def watchdog():
        if done.wait(timeout_s):
            return
        if rank == 0 and out is not None:
            out["e2e_sharded"] = {"error": f"the sharded leg did not return within {timeout_s:.0f} s (exchange not completed); the timed walk above is unaffected"}
            print(json.dumps(out), flush=True)
        os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        res = leg()
    except Exception as e:  # noqa: BLE001 — reported in the line
        res = {"error": f"{type(e).__name__}: {e}"[:400]}
    done.set()
    return res


def e2e_sharded(model, cfg, dev, world, rank, steps=1000, num_batches=2, walkers_total=None, stub=False):
    """north_star's ONE collective inside a timed interval: `Sampler.sample(shard_walkers=True, num_batches=2)` over all ranks with
    `SaveTrajectoryCallback` — the walker batch is sharded (contiguous, cost balanced), every rank walks its share, and after each batch the
    per-rank `[chains, n, T, 3]` blocks go to rank 0 (`dist.gather_ragged_to_host`: one block at a time through one device receive buffer
    and one pinned staging buffer; RCCL send / recv on GPUs), which writes the files on its side thread.  Wall time is bracketed by barriers
    and taken as the MAX over ranks; `gather_s` is rank 0's time inside the exchange (metadata + receives + host copies; it includes
    waiting for the slowest rank to arrive), `gather_bytes` the payload that crossed ranks.  The reference's counterpart is torchmetrics'
    all-gather of the trajectories (metrics/_utils.py:40)."""
    import shutil
    import tempfile

    import torch

    from jamun_amd import dist
    from jamun_amd.callbacks import SaveTrajectoryCallback
    from jamun_amd.data import WalkerBatch
    from jamun_amd.sampling import BAOAB, Sampler, SingleMeasurementSampler

    label = "bench"
    total = walkers_total if walkers_total is not None else CONFIGS[cfg]["walkers"] * world
    mols = workload_molecules(cfg, total, None, 0)  # the WHOLE batch on every rank (as get_initial_graphs builds it); sample() takes this rank's share
    batch = WalkerBatch.from_molecules([dict(m, dataset_label=label) for m in mols])
    tmp = tempfile.mkdtemp(prefix="jamun_bench_sharded_") if rank == 0 else None
    cb = SaveTrajectoryCallback([_BenchDataset(mols[0], label)], output_dir=tmp if rank == 0 else os.path.join(tempfile.gettempdir(), "unused"), write_pdb=False)
    sampler = Sampler(callbacks=[cb], shard_walkers=True)
    if stub:
        bs = _StubBatchSampler(steps)
    else:
        bs = SingleMeasurementSampler(mcmc=BAOAB(steps=steps, save_trajectory=True, save_every_n_steps=1, v_init="gaussian", **MCMC), sigma=SIGMA)
    if not stub:  # untimed: the native sampler of this rank's shard (weights packed, work lists planned, create-time self-check) is built and cached
        Sampler(shard_walkers=True).sample(model, SingleMeasurementSampler(mcmc=BAOAB(steps=2, save_trajectory=True, save_every_n_steps=1, v_init="gaussian", **MCMC), sigma=SIGMA),
                                           num_batches=1, init_graphs=batch)
    torch.manual_seed(42 + rank)  # seed + rank (cmdline/sample.py:86-88)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    sampler.sample(model, bs, num_batches=num_batches, init_graphs=batch, continue_chain=True)
    sync()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    out = None
    if rank == 0:
        wall = float(t.item())
        n_files = sum(len(f) for _, _, f in os.walk(tmp))
        joined = os.path.join(tmp, label, "predicted_samples", "npy", "joined.npy")
        import numpy as np

        shape = list(np.load(joined, mmap_mode="r").shape) if os.path.exists(joined) else None
        out = {"steps_per_batch": steps, "num_batches": num_batches, "walkers_total": total, "wall_s": wall,
               "conformations_per_s": total * steps * num_batches / wall, "gather_s": cb.gather_timings.get("gather_s", 0.0),
               "gather_bytes": int(cb.gather_timings.get("gather_bytes", 0)), "gather_share_of_wall": cb.gather_timings.get("gather_s", 0.0) / wall,
               "writer_wait_s": cb.wait_s, "files_written": n_files, "joined_shape": shape,
               "note": "Sampler.sample(shard_walkers=True) + SaveTrajectoryCallback over all ranks; MAX-over-ranks wall time between two barriers; "
                       "gather_s includes waiting for the slowest rank" + ("; DRY RUN: constant trajectories, no kernels" if stub else "")}
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def rccl_record(dev, world):
    """Proof of the ranks behind an N > 1 line: backend name, world size and an all-reduce of ones over the group."""
    import torch

    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return {"backend": None, "world": 1, "ranks_seen": 1}
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(ones)
    return {"backend": torch.distributed.get_backend() + (" (RCCL)" if torch.distributed.get_backend() == "nccl" else ""),
            "world": world, "ranks_seen": int(round(float(ones.item())))}


# ---- self-launch ---------------------------------------------------------------------------------------------------------


def spawn_ranks(n: int, timeout_s: float = 1700.0) -> int:
    """`python bench.py --gpus N` without a launcher: start N copies of this script, one per GPU, with the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relay rank 0's stdout, return the first non-zero exit code.
    Called before this process has made any GPU call (children are ordinary subprocesses; nothing is exec'ed over a GPU
    process).  All children are supervised: when one exits with an error — or the whole job exceeds `timeout_s` — the others are
    terminated (a rank left alone would sit in the rendezvous or in its first collective for ever), and the stderr tail of the
    rank that failed is relayed."""
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        logs.append(tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=logs[r]))
    t0 = time.monotonic()
    rc, failed = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed, rc = bad[0], codes[bad[0]]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() - t0 > timeout_s:
            failed, rc = -1, 124
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        who = "timeout" if failed < 0 else f"rank {failed} exited with code {rc}"
        sys.stderr.write(f"bench.py: {who}; the other ranks were terminated\n")
        if failed >= 0:
            logs[failed].seek(0)
            sys.stderr.write(logs[failed].read().decode(errors="replace")[-4000:])
    else:
        logs[0].seek(0)
        sys.stderr.write(logs[0].read().decode(errors="replace"))
    out0.seek(0)
    for line in out0.read().decode().splitlines():  # stdout carries the ONE JSON line; library chatter (gloo / RCCL banners) goes to stderr
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2", help="workload (default: cfg2 = BASELINE configs[1], the metric's config)")
    ap.add_argument("--walkers", type=int, default=None, help="walkers per GPU (default: the config's)")
    ap.add_argument("--atoms", type=int, default=None, help="atoms per walker (default: the config's)")
    ap.add_argument("--repeats", type=int, default=0, help="timed repeats of the K-step walk (default: until >= 10 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg1-cpu", action="store_true", help="skip the exact configs[0] CPU run (~1 min) inside cpu_baseline")
    ap.add_argument("--no-secondary", action="store_true", help="skip the scatter-mean / Langevin-kernel bandwidth measurements")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--separable", action="store_true", help="SeparableConv architecture (e3conv_separable.yaml) instead of the default e3conv; not the metric's config")
    ap.add_argument("--dry-run", action="store_true", help="rank plumbing only (gloo, no kernels): used by the CPU test of the self-launch")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="with --dry-run: this rank exits with an error before the rendezvous (tests the supervision of the self-launch)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the Sampler.sample wall-time legs (1000 / 20000 steps per batch, with and without the trajectory writer)")
    ap.add_argument("--sharded-leg", action="store_true", help="run the e2e_sharded leg also at world size 1 when a process group exists (torch.distributed.run --nproc-per-node 1): the RCCL path on one GPU")
    ap.add_argument("--no-also", action="store_true", help="skip the 20-step walks of cfg2r / cfg5h / cfg2 f16x1 added to the cfg2 line")
    ap.add_argument("--no-sweep", action="store_true", help="skip the one-GPU walker-count sweep (256 / 512 / 1024 / 2048 walkers)")
    ap.add_argument("--signature", action="store_true", help="print the build signature of this config (source digest + kernel selection) as JSON and exit (profiles/collect.sh)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: the config's walker count x 8 (2048 for cfg2/3/4) is the TOTAL, split over the ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch

    from jamun_amd import dist

    if args.dry_run:
        if args.dry_run_fail_rank >= 0 and int(os.environ.get("RANK", 0)) == args.dry_run_fail_rank:
            raise SystemExit(f"rank {args.dry_run_fail_rank}: simulated failure before the rendezvous")
        rank, world = dist.init_process_group(backend="gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        cpu = torch.device("cpu")
        sharded = e2e_sharded(_StubModel(cpu), args.config, cpu, world, rank, steps=8, num_batches=2, walkers_total=3 * world + 1, stub=True)
        rec = rccl_record(cpu, world)
        if rank == 0:
            print(json.dumps({"metric": "dry-run", "value": 0.0, "n_gpus": world, "ranks_seen": int(t.item()), "steps": args.steps, "warmup": args.warmup,
                              "e2e_sharded": sharded, "rccl": rec}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    from jamun_amd import native
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser
    from jamun_amd import synth

    rank, world = dist.init_process_group(force=args.sharded_leg)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (jamun_amd has no CPU path)"
    dev = dist.local_device()
    torch.cuda.set_device(dev)

    cfg = CONFIGS[args.config]
    cpu_line = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_line = cpu_baseline(args.config, cfg1_exact=not args.no_cfg1_cpu)  # before the GPU legs (see the module docstring)
    walkers = args.walkers if args.walkers is not None else cfg["walkers"]
    total_walkers = None
    if args.strong:  # north_star's "2048 parallel walkers" at every N: the total is fixed, each rank takes a contiguous share
        total_walkers = args.walkers if args.walkers is not None else 8 * cfg["walkers"]
        lo, hi = dist.shard_range(total_walkers, rank, world)
        walkers = hi - lo
        if walkers < 1:
            raise SystemExit(f"--strong: rank {rank} of {world} has no walkers ({total_walkers} in total)")
    mols = workload_molecules(args.config, walkers, args.atoms, rank)
    batch = WalkerBatch.from_molecules(mols).to(dev)
    model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint(separable=args.separable)).to(dev)
    smp = model.sampler_for(batch, SIGMA)
    n = batch.num_nodes
    if args.signature:
        print(json.dumps(build_signature(smp.stats())), flush=True)
        return

    torch.manual_seed(42 + rank)  # seed + rank, as cmdline/sample.py:86-88
    y = batch.pos + SIGMA * torch.randn_like(batch.pos)
    v = torch.randn_like(y)

    def walk(steps, profile=None):
        """profile: None (no events), "dominant" (events around the conv launches only), "all" (every kernel class)."""
        params = native.make_mcmc_params(steps, **MCMC)
        if profile == "dominant":
            pass  # (armed once, in front of the timed region: below)
        elif profile == "all":
            smp.profile_enable(True)
        return smp.walk("baoab", y, v, params, None, seed=1234 + rank, save_trajectory=True)

    def timed(steps, profile):
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = walk(steps, profile)
        torch.cuda.synchronize()
        dist.barrier()
        return time.perf_counter() - t0, out

    if args.warmup > 0:
        walk(args.warmup)
    # HIP events only around the dominant kernel inside the timed region (timing all 16 launches of a step costs ~4 %)
    prof_mode = None if args.no_profile else "dominant"
    if prof_mode:
        # Every bracketed launch costs two event records (~3 % of the step with all five conv launches of a step timed): only the dominant
        # kernel, and every PROF_EVERY-th of its launches — coprime to the launches per step, so every layer is visited (same box: 277–285 k
        # with every launch timed, 287–290 k with every 7th, 293 k without events).  Armed ONCE, outside the timed region.
        smp.profile_enable(True, classes=["conv0", "conv1"], every=PROF_EVERY)
    dts = []
    dt, (y_traj, score_traj, xhat_traj, xhat) = timed(args.steps, prof_mode)
    assert xhat_traj.shape[0] == args.steps and torch.isfinite(xhat_traj).all()
    dts.append(dt)
    prof = smp.profile_read() if prof_mode else None  # events of the first repeat (the pool is re-armed per repeat)
    reps = args.repeats if args.repeats > 0 else None
    t_first = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t_first, op=torch.distributed.ReduceOp.MAX)
    if reps is None:  # the same count on every rank: derived from the max-over-ranks time of the first repeat
        reps = int(min(5000, max(1, math.ceil(MIN_TIMED_S / max(float(t_first.item()), 1e-6)))))
    for it in range(reps - 1):
        dt, _o = timed(args.steps, prof_mode)
        dts.append(dt)
        if prof_mode and ((it + 1) % PROF_READ == 0 or it == reps - 2):
            p2 = smp.profile_read()
            prof = {k: (prof[k][0] + p2[k][0], prof[k][1] + p2[k][1]) for k in prof}
    smp.profile_enable(False)
    prof_all = None
    if prof is not None and rank == 0 and world == 1:
        # per-kernel breakdown from a separate, untimed pass with every class timed
        walk(min(args.steps, 5), profile="all")
        torch.cuda.synchronize()
        prof_all = smp.profile_read()
        smp.profile_enable(False)
    stats = smp.stats()

    t = torch.tensor(dts, dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)  # per repeat: the slowest rank
    per_rep = [float(x) for x in t.tolist()]
    dt_med = statistics.median(per_rep)
    tot_walkers = torch.tensor([batch.num_graphs], dtype=torch.int64, device=dev)
    tot_atoms = torch.tensor([n], dtype=torch.int64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tot_walkers)
        torch.distributed.all_reduce(tot_atoms)

    rccl = rccl_record(dev, world)
    sharded = None
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()  # (under a launcher even one rank has a group: the RCCL path runs)
    # (the leg itself runs at the very end, when rank 0's line is complete but for it: `_guarded_sharded_leg`)
    sharded_wanted = ((world > 1 and not args.no_e2e) or (grouped and args.sharded_leg)) and args.walkers is None and args.atoms is None
    if sharded_wanted:
        y_traj = score_traj = xhat_traj = None  # (free the headline walk's frames)

    if rank == 0:
        total_conf = int(tot_walkers.item()) * args.steps
        sizes = sorted({int(m["pos"].shape[0]) for m in mols})
        out = {
            "metric": "sampled conformations/sec (whole node), uncapped-2AA" if args.config == "cfg2" else f"sampled conformations/sec (whole node), {args.config}",
            "value": total_conf / dt_med,
            "unit": "conformations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_med / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32" if stats.get("dg_emu", 0) != 2 else "f16x1 (opt-in reduced-precision conv, JAMUN_TUNING=f16x1: NOT the metric's precision)",
            "data": "synthetic",
            "timed_repeats": len(per_rep),
            "timed_total_s": sum(per_rep),
            "ms_per_step_min_max": [1e3 * min(per_rep) / args.steps, 1e3 * max(per_rep) / args.steps],
            "config": {
                "workload": f"BASELINE {cfg['baseline']} shape ({args.config}): {cfg['desc']}, {walkers} walkers per GPU, BAOAB walk-jump, "
                            f"sigma=delta=0.04, save_every_n_steps=1, default e3conv (1+5 ConvBlocks, 120x0e+32x1e); fp32 (reference precision 32-true)",
                "walkers_per_gpu": walkers,
                "atoms_per_walker": sizes[0] if len(sizes) == 1 else f"{sizes[0]}..{sizes[-1]} (ragged)",
                "atoms_total": int(tot_atoms.item()),
                "edges_per_forward": stats["n_edges"],
                "mean_in_degree": stats["n_edges"] / n,
                "parallelism": f"walkers sharded over {world} GPU(s), no data-path collective",
                "rng": "philox (in-kernel)",
                **({"arch": "SeparableConv (e3conv_separable.yaml) — NOT the metric's architecture"} if args.separable else {}),
            },
        }
        if prof is not None and args.separable:
            # SeparableConv: per-edge weights, per-destination sums and the point-wise Linear in two launches (k_sep_fused + k_sep_linear);
            # no big contraction: the layer is bound by memory traffic / latency, priced against the HBM roof
            ms0, c0 = prof["conv0"]
            avg0 = ms0 / max(c0, 1)
            nbytes = stats.get("conv_bytes_alg_launch", 0)
            out["roofline"] = {"kernel": "k_sep_fused + k_sep_linear (SeparableConv hidden layer: depth-wise weights formed and consumed in registers, per-destination sums, point-wise Linear)",
                               "bound": "hbm", "achieved": nbytes / (avg0 * 1e-3) / 1e9 if avg0 > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": nbytes / (avg0 * 1e-3) / 1e9 / HBM_PEAK_GBS if avg0 > 0 else 0.0, "traffic": None, "avg_launch_ms": avg0, "launches": c0,
                               "bytes_algorithmic_per_launch": nbytes,
                               "note": "both launches of a hidden layer timed together; algorithmic bytes = h~ of the layer + one feature row per edge + the per-destination sums written and read + slab + weights"}
        elif prof is not None:
            ms0, c0 = prof["conv0"]
            ms1, c1 = prof["conv1"]
            mst, ct = prof_all.get("tprod", (0.0, 0)) if prof_all is not None else (0.0, 0)  # (T pre-pass: from the separate untimed pass)
            avg0 = ms0 / max(c0, 1)                 # the conv kernel alone (k_conv_mf / k_conv_dg): the launch rocprofv3 lists under that name
            avg_pair = avg0 + mst / max(ct, 1)      # ... with the T pre-pass in front of it (k_tprod_h): one hidden layer's conv
            fused = stats["conv_path"] == 2  # one launch covers the scalar-row and the vector-row contraction of a hidden layer (conv1: tail tiles)
            flop = stats["conv0_flop_alg"] + (stats["conv1_flop_alg"] if fused else 0)
            kname = ("k_conv_mf" if stats.get("dg_mode") == 4 else "k_conv_ml" if stats.get("dg_mode") == 5 else f"k_conv_dg<mode {stats.get('dg_mode')}>") if stats["conv_path"] == 2 else \
                {1: "k_conv_fused"}.get(stats["conv_path"], "k_conv")
            f16x3 = stats.get("dg_emu", -1) in (1, 2) and fused and stats.get("conv_flop_exec_launch", 0) > 0  # (2: the opt-in f16x1 mode)
            f16x1 = stats.get("dg_emu", -1) == 2
            if f16x3 or (fused and stats.get("conv_flop_exec_launch", 0) > 0):
                # The roof is the one of the instructions the kernel issues: dense f16 MFMA (2.5 PFLOP/s) for the f16x3 scheme — every
                # fp32 product is three v_mfma_f32_32x32x16_f16 on operands split hi + lo, fp32 accumulate — or the
                # fp32 MFMA roof for the v_mfma_f32_32x32x2_f32 variant.  achieved = FLOPs of the MFMA instructions ONE launch executes
                # (jamun_stats.conv_flop_exec_launch = tiles x hidden units x MFMAs per (tile, unit) x 32768; reproducible as
                # PMC SQ_INSTS_MFMA per dispatch x 32768) / the launch's mean duration from HIP events on the launch stream
                peak = F16_MFMA_PEAK_TFLOPS if f16x3 else F32_MFMA_PEAK_TFLOPS
                ex = stats["conv_flop_exec_launch"] / (avg0 * 1e-3) / 1e12 if avg0 > 0 else 0.0
                useful = stats.get("conv_flop_useful_launch", 0) / (avg0 * 1e-3) / 1e12 if avg0 > 0 else 0.0
                out["roofline"] = {
                    "kernel": f"{kname} (hidden-layer conv: destination-grouped tensor-product contraction, scalar + vector rows in one launch"
                              + ("; A operand formed on the matrix cores" if stats.get("dg_mode") in (4, 5) else "; A operand formed on the vector ALUs")
                              + ("; two passes over the hidden units, block-sparse forming over the occupied 16-row source blocks" if stats.get("dg_mode") == 5 else "") + ")",
                    "bound": "mfma",
                    "mfma_dtype": "f16 (opt-in f16x1: operands rounded to f16, 1 MFMA per product, fp32 accumulate)" if f16x1 else
                                  "f16 (f16x3 emulation of fp32: operands split hi + lo, 3 MFMAs per product, fp32 accumulate)" if f16x3 else "f32",
                    "achieved": ex,
                    "peak": peak,
                    "unit": "TFLOP/s",
                    "frac": ex / peak,
                    "traffic": None,
                    "traffic_unit": "bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 read correction of MI355X_MICROARCH.md)",
                    "avg_launch_ms": avg0,
                    "launches": c0,
                    "flop_executed_per_launch": stats["conv_flop_exec_launch"],
                    "flop_useful_per_launch": stats.get("conv_flop_useful_launch", 0),
                    "frac_useful": useful / peak,
                    "bytes_algorithmic_per_launch": stats.get("conv_bytes_alg_launch", 0),
                    "frac_fp32_equiv": (flop / (avg_pair * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS) if avg_pair > 0 else 0.0,
                    "flop_fp32_algorithmic_per_layer": flop,
                    "avg_layer_ms_with_t_prepass": avg_pair,
                    "note": "achieved / frac: MFMA FLOPs the launch EXECUTES (padding, structural zeros of the forming GEMMs, 3 products per fp32 product) "
                            "against the dense peak of that MFMA dtype — at most 1 by construction; frac_useful: the part of them the "
                            "destination-grouped association needs (no padding, no zero blocks), same time, same peak; frac_fp32_equiv: the "
                            "layer's algorithmic fp32 FLOPs / (conv + T pre-pass time) against the 157.3 TFLOP/s fp32-MFMA roof (the figure "
                            "earlier rounds reported as frac; can exceed 1)",
                    "mfma_flop_executed_per_forward": stats["flop_executed"],
                }
            else:
                ach = flop / (avg0 * 1e-3) / 1e12 if avg0 > 0 else 0.0
                out["roofline"] = {"kernel": kname, "bound": "mfma", "mfma_dtype": "f32", "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": ach / F32_MFMA_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": avg0, "launches": c0, "flop_per_launch": flop,
                                   "traffic_unit": "bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 read correction of MI355X_MICROARCH.md)",
                                   "mfma_flop_executed_per_forward": stats["flop_executed"]}
            # HBM-side bytes per launch from the committed PMC passes of the cfg2 command (profiles/collect.sh); rocprofv3
            # cannot run inside the timed process, so the newest committed summary is quoted, with its file name
            if args.atoms is None and args.walkers is None and not args.strong:
                tr = _pmc_traffic(("k_conv_mf<" if stats.get("dg_mode") == 4 else "k_conv_ml<" if stats.get("dg_mode") == 5 else "k_conv_dg<") if stats["conv_path"] == 2 else "k_conv<",
                                  args.config, "", stats)
                if tr is not None and tr[0] is not None:
                    out["roofline"]["traffic"] = tr[0]
                    out["roofline"]["traffic_source"] = tr[1]
                    if out["roofline"].get("bytes_algorithmic_per_launch"):
                        out["roofline"]["traffic_ratio"] = tr[0] / out["roofline"]["bytes_algorithmic_per_launch"]
                elif tr is not None:  # a summary exists, but from another build (other sources / another kernel selection): not quoted
                    out["roofline"]["traffic_stale"] = tr[1]
        if prof is not None:
            if prof_all is not None:  # separate untimed pass (see above)
                tot = sum(ms for ms, _ in prof_all.values())
                out["kernel_time_share"] = {k: round(ms / tot, 4) for k, (ms, _) in prof_all.items()} if tot > 0 else {}
                out["kernel_avg_ms"] = {k: (ms / c if c else 0.0) for k, (ms, c) in prof_all.items()}
                out["kernel_breakdown_source"] = f"separate untimed pass of {min(args.steps, 5)} steps with every launch bracketed by HIP events"
            # the reference-association FLOP rate, for comparison with SURVEY.md section 8(d) (not a roofline fraction)
            out["config"]["ref_association_tflops_equiv"] = stats["flop_ref_assoc"] * args.steps / dt_med / 1e12
            if stats.get("n_tail_tiles"):
                out["config"]["tail_tiles"] = f"{stats['n_tail_tiles']} tiles / {stats['n_tail']} destinations through k_tail_form + k_tail_contract (kernel class conv1)"
        out["rccl"] = rccl
        if world == 1:
            del y_traj, score_traj, xhat_traj
        if not args.no_also and world == 1 and args.config == "cfg2" and args.walkers is None and args.atoms is None and not args.strong and not args.separable:
            out["also"] = also_legs(model, dev, rank)
        if not args.no_e2e and world == 1 and args.config in ("cfg2", "cfg2r") and args.walkers is None and args.atoms is None and not args.strong:
            # (the metric's config only: two 20 000-step batches of cfg2 are 35 s of GPU time per leg; other shapes take minutes)
            out["e2e"] = e2e_legs(model, mols, dev)
            ref = [l for l in out["e2e"]["legs"] if l.get("steps_per_batch") == 20000 and not l.get("writer") and "wall_s" in l]
            if ref:
                out["e2e"]["ratio_to_value_at_20000_steps_without_writer"] = ref[0]["conformations_per_s"] / out["value"]
        if not args.no_sweep and world == 1 and args.config in ("cfg2", "cfg2r") and args.walkers is None and not args.strong:
            out["batch_sweep"] = batch_sweep(model, args.config, args.atoms, dev, rank)
        if not args.no_secondary and world == 1:
            out["secondary_rooflines"] = secondary_rooflines(dev)
        if cpu_line is not None:
            out["cpu_baseline"] = cpu_line
    else:
        out = None
    if sharded_wanted:
        # every rank takes part; 1000 steps per batch: 0.85 s of walking per batch at cfg2, 13 MB per rank and batch to rank 0
        sharded = _guarded_sharded_leg(out, rank, lambda: e2e_sharded(model, args.config, dev, world, rank, steps=1000, num_batches=2,
                                                                      walkers_total=total_walkers if args.strong else None))
        if rank == 0:
            out["e2e_sharded"] = sharded
    if rank == 0:
        print(json.dumps(out), flush=True)
    if sharded_wanted and isinstance(sharded, dict) and "error" in sharded:
        os._exit(0)  # (an exchange that failed on this rank leaves the group in an unknown state: no barrier, no orderly shutdown)
    if world > 1:
        dist.barrier()  # (names the device under nccl)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
