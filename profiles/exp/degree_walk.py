"""In-degrees along the bench walk (run on the GPU box): does the walk push atoms above 32 in-edges?"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from jamun_amd import native, synth  # noqa: E402
from jamun_amd.data import WalkerBatch  # noqa: E402
from jamun_amd.model import Denoiser  # noqa: E402

dev = torch.device("cuda", 0)
model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint()).to(dev)
for cfg in ("cfg3", "cfg4"):
    batch = WalkerBatch.from_molecules(bench.workload_molecules(cfg, 256)).to(dev)
    smp = model.sampler_for(batch, bench.SIGMA)
    torch.manual_seed(42)
    y = batch.pos + bench.SIGMA * torch.randn_like(batch.pos)
    v = torch.randn_like(y)
    for steps in (1, 3, 10, 23):
        smp.walk("baoab", y, v, native.make_mcmc_params(steps, **bench.MCMC), None, seed=1234, save_trajectory=True)
        st = smp.stats()
        deg = smp.debug_read(1).cpu().flatten().long()
        print(cfg, "steps", steps, "max deg", int(deg.max()), "deg_over32", st["deg_over32"], "mean", float(deg.float().mean()))
