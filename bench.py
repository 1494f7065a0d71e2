#!/usr/bin/env python3
"""Benchmark of the walk-jump hot path: sampled conformations / second, whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the hot path over the walker batch: one denoiser forward + the BAOAB state update + one saved
frame for every walker (save_every_n_steps = 1), i.e. one conformation per walker per step — the unit the reference calls
a "sample" (/root/reference/src/jamun/callbacks/sampler/_measure_sampling_time.py:57,71).  A walk with ``steps = K``
evaluates exactly K forwards and saves K frames (frame 0 is the initial state, functional/_splitting.py:136-155).

Workload (BASELINE.json configs[1]): uncapped-2AA shape — 17 heavy atoms per walker, 256 walkers per GPU (weak scaling:
2048 walkers on 8 GPUs), default e3conv architecture, sigma = delta = 0.04, friction 1, M 1, clip 100, synthetic
molecule + seeded synthetic checkpoint (no datasets / published checkpoints are reachable offline).  Arithmetic is fp32
end to end (exact-fp32 MFMA): the reference's sampling precision is "32-true" and its bf16 mode is undefined
(SURVEY.md Appendix C.12), and the 1e-5 nm parity bar needs fp32.

Multi-GPU: one process per GPU (torch.distributed.run), walkers sharded, no data-path collective; value = all ranks'
conformations / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_ATOMS = 17
WALKERS_PER_GPU = 256
SIGMA = 0.04
MCMC = dict(delta=0.04, friction=1.0, M=1.0, inverse_temperature=1.0, score_fn_clip=100.0)
F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def cpu_baseline(budget_walkers=8, frames=3):
    """The CPU oracle (op-for-op PyTorch restatement of the reference path, kind="port") timed on this box's host cores
    on a bounded sample of the same workload: `budget_walkers` walkers of the 17-atom molecule x `frames` walk-jump
    frames (each frame costs two forwards, as the reference)."""
    from jamun_amd import synth
    from oracle import denoiser as od
    from oracle import graph as og
    from oracle import walk as ow

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, 16)  # these small-tensor ops get slower, not faster, with hundreds of threads (measured: 256 threads 40x slower than 8)
    torch.set_num_threads(cores)
    mol = synth.random_chain(N_ATOMS, seed=0)
    topo = og.collate([{k: v for k, v in mol.items() if torch.is_tensor(v)}] * budget_walkers)
    sd = synth.synthetic_state_dict()
    hp = od.default_hparams()
    noise = ow.TorchNoise(42)
    y0 = topo["pos"] + noise(topo["pos"]) * SIGMA
    score_fn = lambda y: od.score(y, topo, SIGMA, sd, hp)
    xhat_fn = lambda y: od.xhat(y, topo, SIGMA, sd, hp)
    score_fn(y0)  # warm-up (thread pools, allocator)
    t0 = time.perf_counter()
    ow.walk_jump(score_fn, xhat_fn, ow.baoab, y0, "gaussian", noise, steps=frames, save_trajectory=True, **MCMC)
    dt = time.perf_counter() - t0
    # walk_jump evaluates xhat(y_final) once more on top of the 2 forwards per frame
    return {
        "value": budget_walkers * frames / dt,
        "unit": "conformations/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{budget_walkers} walkers x {frames} walk-jump frames of the {N_ATOMS}-atom workload molecule, fp32, {dt:.1f} s",
    }


def _pmc_traffic(kernel: str):
    import glob

    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.items():
            if isinstance(v, dict) and k.startswith(kernel) and "FETCH_SIZE_KB_per_dispatch" in v:
                return (2.0 * v["FETCH_SIZE_KB_per_dispatch"] + v.get("WRITE_SIZE_KB_per_dispatch", 0.0)) * 1024.0, os.path.basename(f)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--walkers", type=int, default=WALKERS_PER_GPU, help="walkers per GPU")
    ap.add_argument("--atoms", type=int, default=N_ATOMS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    args = ap.parse_args()

    from jamun_amd import dist, native, synth
    from jamun_amd.data import WalkerBatch
    from jamun_amd.model import Denoiser

    rank, world = dist.init_process_group()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU (jamun_amd has no CPU path)"
    dev = dist.local_device()
    torch.cuda.set_device(dev)

    mol = synth.random_chain(args.atoms, seed=0)
    batch = WalkerBatch.from_molecules([mol] * args.walkers).to(dev)
    model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint()).to(dev)
    smp = model.sampler_for(batch, SIGMA)
    n = batch.num_nodes

    torch.manual_seed(42 + rank)  # seed + rank, as cmdline/sample.py:86-88
    y = batch.pos + SIGMA * torch.randn_like(batch.pos)
    v = torch.randn_like(y)

    def walk(steps, profile=None):
        """profile: None (no events), "dominant" (events around the conv launches only), "all" (every kernel class)."""
        params = native.make_mcmc_params(steps, **MCMC)
        if profile == "dominant":
            smp.profile_enable(True, classes=["conv0", "conv1"])
        elif profile == "all":
            smp.profile_enable(True)
        out = smp.walk("baoab", y, v, params, None, seed=1234 + rank, save_trajectory=True)
        return out

    if args.warmup > 0:
        walk(args.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # HIP events only around the dominant kernel inside the timed region (timing all 16 launches of a step costs ~4 %)
    y_traj, score_traj, xhat_traj, xhat = walk(args.steps, profile=None if args.no_profile else "dominant")
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    assert xhat_traj.shape[0] == args.steps and torch.isfinite(xhat_traj).all()
    prof = smp.profile_read() if not args.no_profile else None
    smp.profile_enable(False)
    prof_all = None
    if prof is not None and rank == 0 and world == 1:
        # per-kernel breakdown from a separate, untimed pass with every class timed
        walk(min(args.steps, 5), profile="all")
        torch.cuda.synchronize()
        prof_all = smp.profile_read()
        smp.profile_enable(False)
    stats = smp.stats()

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt_max = float(t.item())

    if rank == 0:
        total_conf = args.walkers * world * args.steps
        out = {
            "metric": "sampled conformations/sec (whole node), uncapped-2AA",
            "value": total_conf / dt_max,
            "unit": "conformations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[1] shape: uncapped-2AA-like {args.atoms}-atom molecule, {args.walkers} walkers per GPU, "
                            f"BAOAB walk-jump, sigma=delta=0.04, save_every_n_steps=1, default e3conv (1+5 ConvBlocks, 120x0e+32x1e); "
                            f"fp32 (reference precision 32-true)",
                "walkers_per_gpu": args.walkers,
                "atoms_per_walker": args.atoms,
                "edges_per_forward": stats["n_edges"],
                "mean_in_degree": stats["n_edges"] / n,
                "parallelism": f"walkers sharded over {world} GPU(s), no data-path collective",
                "rng": "philox (in-kernel)",
            },
        }
        if prof is not None:
            ms0, c0 = prof["conv0"]
            ms1, c1 = prof["conv1"]
            avg0 = ms0 / max(c0, 1)
            fused = c1 == 0  # fused kernel: scalar-row and vector-row contractions of a hidden layer in ONE launch
            flop = stats["conv0_flop_alg"] + (stats["conv1_flop_alg"] if fused else 0)
            ach = flop / (avg0 * 1e-3) / 1e12 if avg0 > 0 else 0.0
            out["roofline"] = {
                "kernel": ("k_conv_fused (destination-grouped conv contraction of one hidden layer, scalar + vector rows)" if fused
                           else "k_conv<RC=1,NT=5,NK=5> (scalar-output conv contraction, hidden layers)"),
                "bound": "mfma",
                "achieved": ach,
                "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach / F32_MFMA_PEAK_TFLOPS,
                "traffic": None,
                "traffic_unit": "bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 read correction of MI355X_MICROARCH.md)",
                "avg_launch_ms": avg0,
                "launches": c0,
                "flop_per_launch": flop,
                "mfma_flop_executed_per_forward": stats["flop_executed"],  # all conv launches of one forward, padding included
            }
            # HBM-side bytes per launch from the committed PMC passes of this same command (profiles/collect.sh); rocprofv3
            # cannot run inside the timed process, so the newest committed summary is quoted, with its file name
            tr = _pmc_traffic("k_conv_fused" if fused else "k_conv")
            if tr is not None:
                out["roofline"]["traffic"] = tr[0]
                out["roofline"]["traffic_source"] = tr[1]
            if prof_all is not None:  # separate untimed pass (see above)
                tot = sum(ms for ms, _ in prof_all.values())
                out["kernel_time_share"] = {k: round(ms / tot, 4) for k, (ms, _) in prof_all.items()} if tot > 0 else {}
                out["kernel_avg_ms"] = {k: (ms / c if c else 0.0) for k, (ms, c) in prof_all.items()}
                out["kernel_breakdown_source"] = f"separate untimed pass of {min(args.steps, 5)} steps with every launch bracketed by HIP events"
            # the reference-association FLOP rate, for comparison with SURVEY.md section 8(d) (not a roofline fraction)
            out["config"]["ref_association_tflops_equiv"] = stats["flop_ref_assoc"] * args.steps / dt_max / 1e12
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
