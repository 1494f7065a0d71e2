"""k_conv_ml<8> (window 128) determinism / accuracy probe: N forwards, equality between calls and deviation from the general kernel per layer."""
import sys, torch
sys.path.insert(0, ".")
from jamun_amd import synth, native
from jamun_amd.data import WalkerBatch
from jamun_amd.model import Denoiser
from jamun_amd.native import NativeSampler
dev = torch.device("cuda", 0)
model = Denoiser.from_checkpoint_dict(synth.synthetic_checkpoint(output_gain=0.5)).to(dev)
case = sys.argv[1] if len(sys.argv) > 1 else "w128"
mols = {"w128": [synth.random_chain(120, seed=3)] * 2 + [synth.random_chain(128, seed=4)],
        "w96": [synth.random_chain(93, seed=5)] * 3,
        "w168": [synth.random_chain(166, seed=5), synth.random_chain(167, seed=6), synth.random_chain(150, seed=7)]}[case]
batch = WalkerBatch.from_molecules(mols).to(dev)
torch.manual_seed(13)
y = batch.pos + 0.04 * torch.randn(batch.pos.shape).to(dev)
ml = NativeSampler(model._native, 0.04, batch, dev)
st = ml.stats()
native.TUNING["no_dg"] = 1
general = NativeSampler(model._native, 0.04, batch, dev)
native.TUNING.pop("no_dg")
xg = general.xhat(y)
ref_layers = [general.debug_read(0, l).clone() for l in range(6)]
xs, layers = [], []
for i in range(6):
    xs.append(ml.xhat(y).clone())
    layers.append([ml.debug_read(0, l).clone() for l in range(6)])
neq = sum(int(not torch.equal(xs[0], x)) for x in xs[1:])
dev_l = [max(((layers[i][l] - ref_layers[l]).abs().max() / ref_layers[l].abs().max()).item() for i in range(6)) for l in range(6)]
first_bad = next((l for l in range(6) if any(not torch.equal(layers[0][l], layers[i][l]) for i in range(1, 6))), None)
bad_rows = None
if first_bad is not None:
    d = torch.stack([(layers[i][first_bad] - layers[0][first_bad]).abs() for i in range(1, 6)]).amax(0)
    rows = (d.amax(1) > 0).nonzero().flatten().tolist()
    cols = (d.amax(0) > 0).nonzero().flatten().tolist()
    bad_rows = (len(rows), rows[:12], len(cols), cols[:8], cols[-4:])
print(f"{case} window {st['ml_window']}: calls differing from the first {neq}/5; first non-reproducible layer {first_bad}; rows/cols {bad_rows}; max rel dev from general per layer {['%.1e' % v for v in dev_l]}")
