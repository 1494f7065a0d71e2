"""CPU restatement of the walk-jump sampling loop (integrators + sampler host loop).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PINNED: the integrators
and ``walk_jump`` are checked against ``tests/golden/baoab_*.npz`` /
``aboba_*.npz`` / ``walkjump_*.npz`` produced by importing the reference's own
``src/jamun/sampling/mcmc/functional/_splitting.py`` and
``src/jamun/sampling/walkjump/_single_measurement.py`` (``tests/golden/make_golden.py``).

Noise handling: the reference draws ``torch.randn_like`` from the global CPU
generator.  Here every function takes ``noise`` — a callable returning the
next ``[N,3]`` draw — so that the same stream can be replayed into the HIP
path.  ``TorchNoise(seed)`` reproduces the reference's call order exactly.
"""

from __future__ import annotations

import math
from typing import Callable, Optional, Union

import torch


class TorchNoise:
    """The reference's noise source: successive ``torch.randn`` draws after ``torch.manual_seed(seed)``."""

    def __init__(self, seed: int, dtype=torch.float32):
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        self.dtype = dtype

    def __call__(self, like: torch.Tensor) -> torch.Tensor:
        return torch.randn(like.shape, generator=self.gen, dtype=torch.float32).to(like.dtype)


class RecordedNoise:
    """Replays a recorded ``[K, N, 3]`` noise tensor draw by draw."""

    def __init__(self, tensor: torch.Tensor):
        self.t = tensor
        self.i = 0

    def __call__(self, like: torch.Tensor) -> torch.Tensor:
        r = self.t[self.i].to(like.dtype)
        self.i += 1
        return r


def process_score(orig_score: torch.Tensor, inverse_temperature: float, score_fn_clip: Optional[float]):
    """``create_score_fn`` (``src/jamun/sampling/mcmc/functional/_splitting.py:26-41``): per-atom norm clip, then x beta."""
    score = orig_score
    if score_fn_clip is not None:
        norm = torch.linalg.vector_norm(score, dim=-1, keepdim=True)
        clip = torch.min(norm, torch.ones_like(norm) * score_fn_clip)
        score = (score / norm) * clip
    return score * inverse_temperature, orig_score


def initialize_velocity(v_init, y, u, noise):
    """``initialize_velocity`` (``_splitting.py:11-23``)."""
    if isinstance(v_init, str):
        if v_init == "gaussian":
            return math.sqrt(u) * noise(y)
        if v_init == "zero":
            return torch.zeros_like(y)
        raise RuntimeError(f"{v_init} not in (gaussian, zero)")
    if isinstance(v_init, torch.Tensor):
        return v_init
    raise RuntimeError(f"{type(v_init)=} must be either `str` or `Tensor`.")


def baoab(
    y,
    score_fn: Callable,
    steps: int,
    noise: Callable,
    v_init: Union[str, torch.Tensor] = "zero",
    save_trajectory=False,
    save_every_n_steps=1,
    burn_in_steps=0,
    delta=1.0,
    friction=1.0,
    M=1.0,
    inverse_temperature=1.0,
    score_fn_clip=None,
    **_,
):
    """``baoab`` (``_splitting.py:112-178``).  Note the second half-kick has no ``u`` (``:166``)."""
    i = 0
    y_traj = [] if save_trajectory else None
    if y_traj is not None and i >= burn_in_steps:
        y_traj.append(y)
    u = pow(M, -1)
    zeta2 = math.sqrt(1 - math.exp(-2 * friction))
    v = initialize_velocity(v_init, y, u, noise)
    psi, orig = process_score(score_fn(y).to(y.dtype), inverse_temperature, score_fn_clip)
    score_traj = [orig]
    for i in range(1, steps):
        v = v + u * (delta / 2) * psi
        y = y + (delta / 2) * v
        R = noise(y)
        vhat = math.exp(-friction) * v + zeta2 * math.sqrt(u) * R
        y = y + (delta / 2) * vhat
        psi, orig = process_score(score_fn(y).to(y.dtype), inverse_temperature, score_fn_clip)
        v = vhat + (delta / 2) * psi
        if y_traj is not None and ((i % save_every_n_steps) == 0) and (i >= burn_in_steps):
            y_traj.append(y)
            score_traj.append(orig)
    if y_traj is not None:
        y_traj = torch.stack(y_traj)
    score_traj = torch.stack(score_traj)
    return y, v, y_traj, score_traj


def aboba(
    y,
    score_fn: Callable,
    steps: int,
    noise: Callable,
    v_init: Union[str, torch.Tensor] = "zero",
    save_trajectory=False,
    save_every_n_steps=1,
    burn_in_steps=0,
    delta=1.0,
    friction=1.0,
    M=1.0,
    inverse_temperature=1.0,
    score_fn_clip=None,
    **_,
):
    """``aboba`` (``_splitting.py:44-109``).  ``score_traj`` is one frame shorter than ``y_traj`` and
    holds half-step scores; ``save_trajectory=False`` raises in ``torch.stack([])`` as the reference does."""
    i = 0
    y_traj = [] if save_trajectory else None
    if y_traj is not None and i >= burn_in_steps:
        y_traj.append(y)
    u = pow(M, -1)
    zeta2 = math.sqrt(1 - math.exp(-2 * friction))
    v = initialize_velocity(v_init, y, u, noise)
    score_traj = []
    for i in range(1, steps):
        y = y + (delta / 2) * v
        psi, orig = process_score(score_fn(y).to(y.dtype), inverse_temperature, score_fn_clip)
        v = v + u * (delta / 2) * psi
        R = noise(y)
        vhat = math.exp(-friction) * v + zeta2 * math.sqrt(u) * R
        v = vhat + (delta / 2) * psi
        y = y + (delta / 2) * v
        if y_traj is not None and ((i % save_every_n_steps) == 0) and (i >= burn_in_steps):
            y_traj.append(y)
            score_traj.append(orig)
    if y_traj is not None:
        y_traj = torch.stack(y_traj)
    score_traj = torch.stack(score_traj)
    return y, v, y_traj, score_traj


def walk_jump(score_fn, xhat_fn, mcmc: Callable, y_init, v_init, noise, **mcmc_kwargs):
    """``SingleMeasurementSampler.walk_jump`` + ``sample`` (``src/jamun/sampling/walkjump/_single_measurement.py:21-89``).

    One extra denoiser forward per saved frame for ``xhat_traj`` (``:57-66``), as the reference.
    """
    y, v, y_traj, score_traj = mcmc(y_init, score_fn, noise=noise, v_init=v_init, **mcmc_kwargs)
    xhat = xhat_fn(y)
    out = {"xhat": xhat, "y": y, "v": v, "y_traj": y_traj, "score_traj": score_traj}
    if y_traj is not None:
        out["t_traj"] = torch.ones(y_traj.size(0), dtype=torch.long)
        out["xhat_traj"] = torch.stack([xhat_fn(y_traj[i]) for i in range(y_traj.size(0))], dim=0)
    else:
        out["t_traj"] = None
        out["xhat_traj"] = None
    out["sample"] = out["xhat"]
    return out


def sampler_loop(pos, score_fn, xhat_fn, mcmc, sigma, num_batches, continue_chain, noise, **mcmc_kwargs):
    """Host loop of ``Sampler.sample`` (``src/jamun/sampling/_sampler.py:53-98``) + ``ModelSamplingWrapper.sample_initial_noisy_positions``
    (``src/jamun/utils/sampling_wrapper.py:21-24``).  RNG call order: y0 draw, v0 draw, one draw per step
    (SURVEY.md Appendix A).  ``Sampler`` forces ``v_init="gaussian"`` on the first batch (``_sampler.py:73``).
    Returns the list of per-batch output dicts."""
    y_init = pos + noise(pos) * sigma
    v_init: Union[str, torch.Tensor] = "gaussian"
    outs = []
    for _ in range(num_batches):
        out = walk_jump(score_fn, xhat_fn, mcmc, y_init, v_init, noise, **mcmc_kwargs)
        outs.append(out)
        if continue_chain:
            y_init, v_init = out["y"], out["v"]
        else:
            y_init = pos + noise(pos) * sigma
            v_init = "gaussian"
    return outs


def unbatch(value: torch.Tensor, ptr: torch.Tensor):
    """``ModelSamplingWrapper.unbatch_samples`` for one key (``src/jamun/utils/sampling_wrapper.py:49-83``):
    2-D ``[N,3]`` split by graph; 3-D ``[T,N,3]`` -> ``[N,T,3]`` then split."""
    if value.ndim == 3:
        value = value.permute(1, 0, 2)
    return [value[ptr[i] : ptr[i + 1]] for i in range(len(ptr) - 1)]
