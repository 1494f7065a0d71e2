import torch, sys, os
sys.path.insert(0,'.')
import bench
from jamun_amd import synth
from jamun_amd.data import WalkerBatch
from jamun_amd.model import Denoiser
dev=torch.device('cuda',0)
out={}
for cfg in ("cfg2","cfg2r","cfg3","cfg5h"):
    w = min(bench.CONFIGS[cfg]["walkers"], 64)
    batch = WalkerBatch.from_molecules(bench.workload_molecules(cfg, w)).to(dev)
    model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint()).to(dev)
    g=torch.Generator(device='cpu'); g.manual_seed(5)
    y = batch.pos + 0.04*torch.randn(batch.pos.shape, generator=g).to(dev)
    smp = model.sampler_for(batch, 0.04)
    out[cfg]=smp.xhat(y).cpu()
torch.save(out, sys.argv[1])
