"""Per-layer / per-channel error of every kernel combination against a cached oracle forward:  python profiles/diag_weight_presets.py trained chain17x6"""
import os, sys, importlib.util
sys.path.insert(0, os.getcwd())
import numpy as np, torch
spec = importlib.util.spec_from_file_location("mk", "tests/golden/make_oracle_fixtures.py"); mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
from jamun_amd.data import WalkerBatch
from jamun_amd.model import Denoiser
from jamun_amd.native import NativeSampler
dev = torch.device("cuda", 0)
variant, kind = sys.argv[1], sys.argv[2]
ref = {k: torch.tensor(v) for k, v in np.load(f"tests/golden/oracle_forward_{variant}_{kind}.npz").items()}
model = Denoiser.from_checkpoint_dict(mk.variant_checkpoint(variant)).to(dev)
batch = WalkerBatch.from_molecules(mk.molecules(kind)).to(dev)
sigma = mk.VARIANTS[variant]["sigma"]
combos = [[], ["node_fp32"], ["node_fp32", "edge_h_fp32"], ["node_fp32", "no_mfi"], ["node_fp32", "no_mfi", "no_mf"],
          ["node_fp32", "no_mfi", "dg_fp32"], ["no_dg", "node_fp32", "edge_h_fp32"]]
for envs in combos:
    smp = NativeSampler(model._native, sigma, batch, dev, tuning={e: 1 for e in envs})
    y = ref["y"].to(dev)
    x = smp.xhat(y)
    st = smp.stats()
    line = []
    l = 0
    while f"x{l}" in ref:
        xl, r = smp.debug_read(0, l).cpu(), ref[f"x{l}"]
        e_max = ((xl - r).abs().max() / r.abs().max()).item()
        e_ch = ((xl - r).abs().amax(0) / r.abs().amax(0).clamp_min(1e-30)).max().item()
        line.append(f"x{l}: {e_max:.1e}/{e_ch:.1e}")
        l += 1
    rm = ((x.cpu().double() - ref["xhat"].double()) ** 2).sum(-1).mean().sqrt().item()
    print(",".join(e.replace("JAMUN_", "") for e in envs) or "default", f"dg_mode {st['dg_mode']} init {st['init_path']} emu {st['dg_emu']} conv_path {st['conv_path']}", " ".join(line), f"xhat rmsd {rm:.2e}", flush=True)
