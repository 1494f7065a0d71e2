"""Does replaying the 20-step walk as ONE hipGraph change the step time?  (run on the GPU box)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from jamun_amd import native, synth
from jamun_amd.data import WalkerBatch
from jamun_amd.model import Denoiser
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint()).to(dev)
batch = WalkerBatch.from_molecules(bench.workload_molecules("cfg2", 256)).to(dev)
smp = model.sampler_for(batch, bench.SIGMA)
torch.manual_seed(42)
y = batch.pos + bench.SIGMA * torch.randn_like(batch.pos)
v = torch.randn_like(y)
steps = 20
params = native.make_mcmc_params(steps, **bench.MCMC)
def walk():
    return smp.walk("baoab", y, v, params, None, seed=1234, save_trajectory=True)
for _ in range(3): walk()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): walk()
torch.cuda.synchronize()
print("direct  ms/step", (time.perf_counter() - t0) / 50 / steps * 1e3)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    walk(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            out = walk()
    except Exception as e:
        print("capture failed:", repr(e)[:300]); sys.exit(0)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    print("graph   ms/step", (time.perf_counter() - t0) / 50 / steps * 1e3)
