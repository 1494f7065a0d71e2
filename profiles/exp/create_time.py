# jamun_sampler_create wall time at a few batch shapes (on the GPU box: python3 profiles/exp/create_time.py); bench.py e2e reports the same as sampler_create_s
import sys, time, torch
sys.path.insert(0, '.')
from jamun_amd import synth
from jamun_amd.data import WalkerBatch
from jamun_amd.model import Denoiser
from jamun_amd.native import NativeSampler
dev = torch.device('cuda', 0)
model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint()).to(dev)
for n, w in [(17, 256), (17, 256), (57, 256), (10, 4)]:
    batch = WalkerBatch.from_molecules([synth.random_chain(n, seed=0)] * w).to(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = NativeSampler(model._native, 0.04, batch, dev)
    torch.cuda.synchronize(); print(f"sampler_create n={n} walkers={w}: {time.perf_counter()-t0:.3f} s")
