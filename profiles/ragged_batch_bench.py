import sys, os, time, torch, random
sys.path.insert(0, '.')
from jamun_amd import synth, native
from jamun_amd.data import WalkerBatch
from jamun_amd.model import Denoiser
dev = torch.device('cuda', 0)
random.seed(0)
sizes = [random.randint(17, 57) for _ in range(256)]
mols = [synth.random_chain(n, seed=i) for i, n in enumerate(sizes)]
batch = WalkerBatch.from_molecules(mols).to(dev)
model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint()).to(dev)
smp = model.sampler_for(batch, 0.04)
torch.manual_seed(1)
y = batch.pos + 0.04 * torch.randn_like(batch.pos); v = torch.randn_like(y)
MCMC = dict(delta=0.04, friction=1.0, M=1.0, inverse_temperature=1.0, score_fn_clip=100.0)
def walk(steps):
    p = native.make_mcmc_params(steps, **MCMC)
    return smp.walk("baoab", y, v, p, None, seed=7, save_trajectory=True)
walk(2); torch.cuda.synchronize()
t0 = time.perf_counter(); walk(10); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("ragged 17..57 x256 walkers, atoms", batch.num_nodes, "conv_path", smp.stats()["conv_path"], "ms/step", round(dt / 10 * 1e3, 3), "conf/s", round(256 * 10 / dt))
