"""Generate golden vectors by IMPORTING the reference's own torch-only files.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

Imports, by file path, without copying any source:
  /root/reference/src/jamun/sampling/mcmc/functional/_splitting.py   (baoab, aboba)
  /root/reference/src/jamun/sampling/walkjump/_single_measurement.py (SingleMeasurementSampler)
  /root/reference/src/jamun/utils/residue_metadata.py                (integer encodings)
and writes small ``.npz`` / ``.json`` fixtures (inputs + expected outputs) next to this script.
"""
import importlib.util
import json
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src/jamun"
HERE = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synthetic_score(mu, s, a, b):
    """Deterministic smooth score used for the integrator fixtures: -(y-mu)/s^2 + a*sin(b*y)."""
    return lambda y: -(y - mu) / (s * s) + a * torch.sin(b * y)


def record_noise(seed, n_draws, shape):
    torch.manual_seed(seed)
    return torch.stack([torch.randn(shape) for _ in range(n_draws)])


def main():
    split = load(f"{REF}/sampling/mcmc/functional/_splitting.py", "ref_splitting")
    sms = load(f"{REF}/sampling/walkjump/_single_measurement.py", "ref_sms")
    resmeta = load(f"{REF}/utils/residue_metadata.py", "ref_resmeta")

    N = 40
    g = torch.Generator().manual_seed(7)
    y0 = torch.randn(N, 3, generator=g) * 0.3
    mu = torch.randn(N, 3, generator=g) * 0.2

    cases = {
        # name: (integrator, kwargs)
        "baoab_default": ("baoab", dict(steps=50, delta=0.04, friction=1.0, M=1.0, inverse_temperature=1.0, score_fn_clip=100.0, v_init="gaussian", save_trajectory=True)),
        "baoab_clip_mass": ("baoab", dict(steps=30, delta=0.05, friction=0.7, M=2.0, inverse_temperature=0.8, score_fn_clip=3.0, v_init="zero", save_trajectory=True, save_every_n_steps=3, burn_in_steps=6)),
        "baoab_noclip_notraj": ("baoab", dict(steps=20, delta=0.04, friction=1.0, M=1.0, inverse_temperature=1.0, score_fn_clip=None, v_init="gaussian", save_trajectory=False)),
        "aboba_default": ("aboba", dict(steps=50, delta=0.04, friction=1.0, M=1.0, inverse_temperature=1.0, score_fn_clip=100.0, v_init="gaussian", save_trajectory=True)),
        "aboba_clip_mass": ("aboba", dict(steps=30, delta=0.05, friction=0.7, M=2.0, inverse_temperature=0.8, score_fn_clip=3.0, v_init="zero", save_trajectory=True, save_every_n_steps=2, burn_in_steps=4)),
    }
    s, a, b = 0.35, 0.8, 3.0
    for name, (integ, kw) in cases.items():
        fn = getattr(split, integ)
        noise = record_noise(42, kw["steps"] + 1, (N, 3))
        torch.manual_seed(42)
        y, v, y_traj, score_traj = fn(y0.clone(), synthetic_score(mu, s, a, b), **kw)
        out = dict(y0=y0.numpy(), mu=mu.numpy(), score_params=np.array([s, a, b]), noise=noise.numpy(), y=y.numpy(), v=v.numpy())
        if y_traj is not None:
            out["y_traj"] = y_traj.numpy()
        out["score_traj"] = score_traj.numpy()
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        with open(os.path.join(HERE, f"{name}.json"), "w") as f:
            json.dump(dict(integrator=integ, kwargs=kw), f, indent=1)
        print(name, "y_traj", None if y_traj is None else tuple(y_traj.shape), "score_traj", tuple(score_traj.shape))

    # walk_jump through the reference's SingleMeasurementSampler with a stub model
    class StubModel:
        device = torch.device("cpu")

        def __init__(self, sigma):
            self.sigma = sigma
            self.sfn = synthetic_score(mu, s, a, b)

        def score(self, y, sigma):
            return self.sfn(y)

        def xhat(self, y, sigma):
            return y + (sigma**2) * self.sfn(y)

    import dataclasses

    @dataclasses.dataclass
    class RefBAOAB:  # same merge rule as src/jamun/sampling/mcmc/_splitting.py:56-58
        kw: dict

        def __call__(self, y, score_fn, **kwargs):
            return split.baoab(y, score_fn, **(self.kw | kwargs))

    sigma = 0.04
    kw = dict(steps=12, delta=0.04, friction=1.0, M=1.0, inverse_temperature=1.0, score_fn_clip=100.0, v_init="zero", save_trajectory=True, cpu_offload=True)
    sampler = sms.SingleMeasurementSampler(mcmc=RefBAOAB(kw), sigma=sigma)
    noise = record_noise(43, kw["steps"] + 1, (N, 3))
    torch.manual_seed(43)
    out = sampler.sample(model=StubModel(sigma), y_init=y0.clone(), v_init="gaussian")
    np.savez_compressed(
        os.path.join(HERE, "walkjump_baoab.npz"),
        y0=y0.numpy(), mu=mu.numpy(), score_params=np.array([s, a, b]), noise=noise.numpy(), sigma=np.array(sigma),
        **{k: v.numpy() for k, v in out.items() if torch.is_tensor(v)},
    )
    with open(os.path.join(HERE, "walkjump_baoab.json"), "w") as f:
        json.dump(dict(kwargs=kw, sigma=sigma), f, indent=1)
    print("walkjump keys", {k: tuple(v.shape) for k, v in out.items() if torch.is_tensor(v)})

    # integer encodings (src/jamun/utils/residue_metadata.py:62-83)
    enc = dict(
        atom_type={x: resmeta.encode_atom_type(x) for x in ["C", "O", "N", "F", "S", "H", "P", "Se"]},
        atom_code={x: resmeta.encode_atom_code(x) for x in ["C", "O", "N", "S", "CA", "CB", "OXT", "CG", "H"]},
        residue={x: resmeta.encode_residue(x) for x in resmeta.ResidueMetadata.RESIDUE_CODES + ["HOH", "UNK"]},
    )
    with open(os.path.join(HERE, "encodings.json"), "w") as f:
        json.dump(enc, f, indent=1)


if __name__ == "__main__":
    main()
