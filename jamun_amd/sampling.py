"""Walk-jump sampling: host-side mirror of the reference's sampler stack over the fused HIP walk.

Mirrors, with the same names, argument meaning and error behaviour:
  * ``BAOAB`` / ``ABOBA``            — ``/root/reference/src/jamun/sampling/mcmc/_splitting.py:11-58``
  * ``SingleMeasurementSampler``      — ``src/jamun/sampling/walkjump/_single_measurement.py:8-89``
  * ``ModelSamplingWrapper``          — ``src/jamun/utils/sampling_wrapper.py:9-83``
  * ``Sampler``                       — ``src/jamun/sampling/_sampler.py:15-98``

When the score function is the model's own (``NativeScore``), an integrator call runs as ONE native call
(``jamun_walk_baoab`` / ``jamun_walk_aboba``): every iteration's state update, denoiser forward and trajectory
write is enqueued on the stream with no host round trip.  With any other callable the integrators fall back to a
per-step loop that still uses the HIP update kernels (used by the integrator parity tests).

Noise: ``rng="philox"`` draws in-kernel (Philox4x32-10 keyed by a seed taken from torch's global CPU generator, so
``torch.manual_seed`` / ``seed_everything`` make runs reproducible); ``rng="torch_cpu"`` replays the reference CPU
path's stream — ``torch.randn`` draws from the global CPU generator in the reference's call order — for parity.
"""

from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass
from typing import Any, Callable, Dict, Iterable, List, Optional, Union

import torch
from torch import Tensor

from . import native
from .data import WalkerBatch


class NativeScore:
    """``lambda y: model.score(y, sigma)`` with the handles the fused walk needs."""

    def __init__(self, model: "ModelSamplingWrapper", sigma: float):
        self.model, self.sigma = model, float(sigma)

    def __call__(self, y: Tensor) -> Tensor:
        return self.model.score(y, self.sigma)

    def sampler(self) -> native.NativeSampler:
        return self.model.native_sampler(self.sigma)


def _seed_from_global_rng() -> int:
    return int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())


def _initialize_velocity(v_init, y: Tensor, u: float, rng: str) -> Tensor:
    """``initialize_velocity`` (``functional/_splitting.py:11-23``)."""
    if isinstance(v_init, str):
        if v_init == "gaussian":
            if rng == "torch_cpu":
                return (math.sqrt(u) * torch.randn(y.shape, dtype=y.dtype)).to(y.device)
            return math.sqrt(u) * torch.randn_like(y)
        if v_init == "zero":
            return torch.zeros_like(y)
        raise RuntimeError(f"{v_init} not in (gaussian, zero)")
    if isinstance(v_init, torch.Tensor):
        return v_init
    raise RuntimeError(f"{type(v_init)=} must be either `str` or `Tensor`.")


def _run_walk(integrator: str, y: Tensor, score_fn: Callable, *, steps: int, v_init="zero", save_trajectory=False,
              save_every_n_steps=1, burn_in_steps=0, verbose=False, cpu_offload=False, delta=1.0, friction=1.0, M=1.0,
              inverse_temperature=1.0, score_fn_clip=None, rng="philox", **_):
    if not y.is_cuda:
        raise RuntimeError("jamun_amd integrators run on the GPU only; move y to a cuda device")
    if integrator == "aboba" and not save_trajectory:
        # the reference stacks an empty score list here (functional/_splitting.py:106-107)
        raise RuntimeError("stack expects a non-empty TensorList")
    u = pow(M, -1)
    y = y.detach().to(torch.float32).clone().contiguous()
    v = _initialize_velocity(v_init, y, u, rng).detach().to(y.device, torch.float32).clone().contiguous()
    params = native.make_mcmc_params(steps, delta, friction, M, inverse_temperature, score_fn_clip, save_every_n_steps, burn_in_steps)
    n_iter = max(steps - 1, 0)
    noise = None
    seed = 0
    if rng == "torch_cpu":
        # one randn_like(y) per iteration from the global CPU generator, in order (functional/_splitting.py:93,161)
        noise = torch.stack([torch.randn(y.shape, dtype=torch.float32) for _ in range(n_iter)]) if n_iter else torch.zeros(0, *y.shape)
        noise = noise.to(y.device)
    elif rng == "philox":
        seed = _seed_from_global_rng()
    else:
        raise RuntimeError(f"rng={rng!r} not in (philox, torch_cpu)")

    if isinstance(score_fn, NativeScore):
        smp = score_fn.sampler()
        y_traj, score_traj, xhat_traj, xhat = smp.walk(integrator, y, v, params, noise, seed, save_trajectory)
        extras = {"xhat": xhat, "xhat_traj": xhat_traj}
    else:
        y_traj, score_traj, extras = _python_walk(integrator, y, v, score_fn, params, noise, seed, save_trajectory)
    if cpu_offload:
        y_traj = None if y_traj is None else y_traj.cpu()
        score_traj = None if score_traj is None else score_traj.cpu()
    return y, v, y_traj, score_traj, extras


def _python_walk(integrator, y, v, score_fn, params, noise, seed, save_trajectory):
    """Per-step loop for an arbitrary ``score_fn``: the state updates are still the HIP kernels (``k_baoab_pre/post``,
    ``k_aboba_a/b``) around a score evaluated by the caller — the form the integrator parity tests use."""
    if noise is None:
        g = torch.Generator(device=y.device).manual_seed(seed)
        noise = torch.randn((max(params.steps - 1, 0),) + tuple(y.shape), generator=g, device=y.device, dtype=torch.float32)
    saves = lambda i: (i % params.save_every_n_steps) == 0 and i >= params.burn_in_steps
    y_traj = None
    if save_trajectory:
        y_traj = [y.clone()] if params.burn_in_steps <= 0 else []  # frame 0 is kept iff 0 >= burn_in (_splitting.py:70-72,138-140)
    if integrator == "aboba":
        score_traj = []
        for i in range(1, params.steps):
            native.aboba_a(y, v, params)
            score = score_fn(y).to(torch.float32).contiguous()
            native.aboba_b(y, v, score, noise[i - 1].contiguous(), params)
            if y_traj is not None and saves(i):
                y_traj.append(y.clone())
                score_traj.append(score.clone())
        # the reference stacks the score list unconditionally (:106-107): empty without a trajectory -> it raises
        return (torch.stack(y_traj) if y_traj is not None else None), torch.stack(score_traj), {}
    psi = torch.empty_like(y)
    score = score_fn(y).to(torch.float32).contiguous()
    vv = torch.zeros_like(v)
    native.baoab_post(vv, psi, score, params)  # psi only (v update discarded)
    score_traj = [score.clone()]
    for i in range(1, params.steps):
        native.baoab_pre(y, v, psi, noise[i - 1].contiguous(), params)
        score = score_fn(y).to(torch.float32).contiguous()
        native.baoab_post(v, psi, score, params)
        if y_traj is not None and saves(i):
            y_traj.append(y.clone())
            score_traj.append(score.clone())
    return (torch.stack(y_traj) if y_traj is not None else None), torch.stack(score_traj), {}


@dataclass
class ABOBA:
    delta: float = 1.0
    friction: float = 1.0
    M: float = 1.0
    steps: int = 128
    save_trajectory: bool = False
    save_every_n_steps: int = 1
    burn_in_steps: int = 0
    verbose: bool = False
    cpu_offload: bool = False
    v_init: Union[str, Tensor] = "zero"
    inverse_temperature: float = 1.0
    score_fn_clip: Optional[float] = None
    rng: str = "philox"  # extension: "torch_cpu" replays the reference CPU noise stream

    def __post_init__(self):
        if isinstance(self.v_init, str):
            if self.v_init not in {"gaussian", "zero"}:
                raise RuntimeError(f"{self.v_init} not in (gaussian, zero)")

    def _kwargs(self, kwargs):
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)} | kwargs

    def replace(self, **changes):
        """New integrator with some fields changed — what the parameter-schedule callbacks below call on the
        integrator (``walkjump/_callbacks.py:20,44,63``)."""
        return dataclasses.replace(self, **changes)

    def __call__(self, y: torch.Tensor, score_fn: Callable, **kwargs):
        y, v, y_traj, score_traj, self.last_extras = _run_walk("aboba", y, score_fn, **self._kwargs(kwargs))
        return y, v, y_traj, score_traj


@dataclass
class BAOAB(ABOBA):
    def __call__(self, y: torch.Tensor, score_fn: Callable, **kwargs):
        y, v, y_traj, score_traj, self.last_extras = _run_walk("baoab", y, score_fn, **self._kwargs(kwargs))
        return y, v, y_traj, score_traj


# ---- parameter schedules for measurement-indexed sampling (``walkjump/_callbacks.py:10-77``).  Protocol:
# ``mcmc = cb.on_before_sample(mcmc, t)`` ... ``mcmc = cb.on_after_sample(mcmc, t)`` with t the 1-based measurement index.
class MeasurementDependentParametersCallback:
    """Overrides integrator fields for the measurements listed in ``parameters_by_measurement`` and restores them after."""

    def __init__(self, parameters_by_measurement: Optional[dict] = None, verbose: bool = False):
        self.parameters_by_measurement = parameters_by_measurement or {}
        self.verbose = verbose
        self.previous_params = None

    def on_before_sample(self, mcmc, t: int):
        override = self.parameters_by_measurement.get(t)
        if override:
            self.previous_params = {f.name: getattr(mcmc, f.name) for f in dataclasses.fields(mcmc)}
            mcmc = mcmc.replace(**(self.previous_params | override))
        return mcmc

    def on_after_sample(self, mcmc, t: int):
        if self.previous_params is not None:
            mcmc, self.previous_params = mcmc.replace(**self.previous_params), None
        return mcmc


class DeltaSqrtDecayCallback:
    """``delta / sqrt(t)`` for measurement t, original step size restored afterwards."""

    def __init__(self, verbose: bool = False):
        self.verbose = verbose
        self.delta_orig = None

    def on_before_sample(self, mcmc, t: int):
        self.delta_orig = mcmc.delta
        return mcmc.replace(delta=self.delta_orig / math.sqrt(t))

    def on_after_sample(self, mcmc, t: int):
        return mcmc.replace(delta=self.delta_orig)


class InterpolateParametersCallback:
    """Moves each listed field from ``params[name][0]`` to ``params[name][1]`` with weight ``1 - sqrt(1/t)``; the
    result keeps the type of the start value (ints stay ints).  Not restored after the sample (as the reference)."""

    def __init__(self, params: Dict[str, tuple], verbose: bool = False):
        self.params = params
        self.verbose = verbose

    def on_before_sample(self, mcmc, t: int):
        f = 1 - math.sqrt(1.0 / t)
        return mcmc.replace(**{k: type(v[0])((1 - f) * v[0] + f * v[1]) for k, v in self.params.items()})

    def on_after_sample(self, mcmc, t: int):
        return mcmc


class SingleMeasurementSampler:
    """Single Measurement Walk-Jump Sampler (``walkjump/_single_measurement.py:8-89``)."""

    def __init__(self, mcmc, sigma: float, y_init_distribution: Optional[torch.distributions.Distribution] = None):
        self.mcmc = mcmc
        self.sigma = float(sigma)
        self.y_init_distribution = y_init_distribution

    def walk(self, model, batch_size: Optional[int] = None, y_init: Optional[Tensor] = None, v_init: Union[str, Tensor] = "gaussian"):
        if y_init is None:
            if self.y_init_distribution is None:
                raise RuntimeError("either y_init and y_init_distribution must be supplied")
            y_init = self.y_init_distribution.sample(sample_shape=(batch_size,)).to(model.device)
        score_fn = NativeScore(model, self.sigma) if isinstance(model, ModelSamplingWrapper) else (lambda y: model.score(y, self.sigma))
        y, v, y_traj, score_traj = self.mcmc(y_init, score_fn, v_init=v_init)
        t_traj = torch.ones(y_traj.size(0), device=y_traj.device, dtype=int) if y_traj is not None else None
        return {"y": y, "v": v, "y_traj": y_traj, "t_traj": t_traj, "score_traj": score_traj}

    def walk_jump(self, model, batch_size: Optional[int] = None, y_init: Optional[Tensor] = None, v_init: Union[str, Tensor] = "gaussian"):
        out = self.walk(model, batch_size=batch_size, y_init=y_init, v_init=v_init)
        y, y_traj = out["y"], out["y_traj"]
        extras = getattr(self.mcmc, "last_extras", None) or {}
        # The fused walk already produced the jumps (xhat of the final y and of every saved frame) from the same
        # forwards that produced the scores; otherwise fall back to the reference's extra forwards (:57-66).
        xhat = extras.get("xhat")
        if xhat is None:
            xhat = model.xhat(y, sigma=self.sigma)
        xhat_traj = extras.get("xhat_traj")
        if y_traj is not None and xhat_traj is None:
            xhat_traj = torch.stack([model.xhat(y_traj[i].to(model.device), sigma=self.sigma) for i in range(y_traj.size(0))], dim=0)
        out.update({"xhat": xhat, "xhat_traj": xhat_traj})
        return {k: out[k] for k in ("xhat", "y", "v", "xhat_traj", "y_traj", "t_traj", "score_traj")}

    def sample(self, model, batch_size: Optional[int] = None, y_init: Optional[Tensor] = None, v_init: Union[str, Tensor] = "gaussian"):
        out = self.walk_jump(model, batch_size=batch_size, y_init=y_init, v_init=v_init)
        out["sample"] = out["xhat"]
        return out


class ModelSamplingWrapper:
    """Wrapper to sample positions from a model (``utils/sampling_wrapper.py:9-83``)."""

    def __init__(self, model, init_graphs: WalkerBatch, sigma: float, rng: str = "philox"):
        self._model = model
        self.init_graphs = init_graphs
        self.sigma = sigma
        self.rng = rng

    @property
    def device(self) -> torch.device:
        return self._model.device

    def sample_initial_noisy_positions(self) -> Tensor:
        pos = self.init_graphs.pos
        if self.rng == "torch_cpu":
            return pos + torch.randn(pos.shape, dtype=pos.dtype).to(pos.device) * self.sigma
        return pos + torch.randn_like(pos) * self.sigma

    def __getattr__(self, name):
        return getattr(self._model, name)

    def native_sampler(self, sigma: float) -> native.NativeSampler:
        return self._model.sampler_for(self.init_graphs, sigma)

    def score(self, y, sigma, *args, **kwargs):
        return self._model.score(self.positions_to_graph(y), sigma)

    def xhat(self, y, sigma, *args, **kwargs):
        return self._model.xhat(self.positions_to_graph(y), sigma).pos

    def positions_to_graph(self, positions: Tensor) -> WalkerBatch:
        assert len(positions) == self.init_graphs.num_nodes, "The number of positions and nodes should be the same"
        assert positions.shape[1] == 3, "Positions tensor should have a shape of (n, 3)"
        self.input_graphs = self.init_graphs.with_pos(positions)
        return self.input_graphs

    def unbatch_samples(self, samples: Dict[str, Tensor]) -> List[dict]:
        """Per-walker dicts: init-graph attributes plus every 2-D / 3-D sample tensor split by walker; ``[T,N,3]``
        becomes ``[n,T,3]`` (``sampling_wrapper.py:49-83``).  1-D values (``t_traj``) are skipped, as the reference."""
        g = self.init_graphs
        ptr = g.ptr.tolist()
        outs = []
        for w in range(g.num_graphs):
            a, b = ptr[w], ptr[w + 1]
            d = SampleGraph(
                pos=g.pos[a:b], atom_type_index=g.atom_type_index[a:b], atom_code_index=g.atom_code_index[a:b],
                residue_code_index=g.residue_code_index[a:b], residue_sequence_index=g.residue_sequence_index[a:b],
                dataset_label=g.dataset_label[w] if g.dataset_label else None, num_nodes=b - a,
            )  # fmt: skip
            outs.append(d)
        for key, value in samples.items():
            if value is None or value.ndim not in [2, 3]:
                continue
            if value.ndim == 3:
                value = value.permute(1, 0, 2)  # "num_frames atoms coords -> atoms num_frames coords"
            for w, d in enumerate(outs):
                if key in d:
                    raise ValueError(f"Key {key} already exists in the output graph.")
                chunk = value[ptr[w] : ptr[w + 1]]
                if chunk.shape[0] != d["num_nodes"]:
                    raise ValueError(f"Number of nodes in unbatched value ({chunk.shape[0]}) for key {key} does not match number of nodes in output graph ({d['num_nodes']}).")
                d[key] = chunk
        return outs


class SampleGraph(dict):
    """One walker's sample as handed to callbacks: a dict whose keys are also attributes (``sample.xhat_traj``,
    ``sample.dataset_label``), like the per-walker PyG ``Data`` objects of the reference (``sampling_wrapper.py:49-83``)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


class Sampler:
    """Host loop over sampling batches (``sampling/_sampler.py:15-98``) without Lightning Fabric.

    One process per GPU: ``devices`` / ``strategy`` / ``num_nodes`` are accepted for config compatibility; rank and world
    size come from ``torch.distributed`` when it is initialised (launch with ``torch.distributed.run``).  With
    ``shard_walkers=True`` (extension; the reference replicates) the walker batch is split in contiguous blocks across
    ranks and callbacks receive only the local walkers; see ``jamun_amd.dist``.
    """

    def __init__(self, accelerator: str = "auto", strategy: str = "auto", devices: Any = "auto", num_nodes: int = 1,
                 precision: Union[str, int] = "32-true", plugins: Any = None, callbacks: Optional[Iterable[Any]] = None,
                 loggers: Any = None, shard_walkers: bool = False, rng: str = "philox"):
        # "32-true" (the reference's sampling default, hydra_config/sample.yaml:27-28) is fp32-accurate arithmetic (f16x3).  The 16-bit names select
        # the OPT-IN reduced mode of the hidden-layer conv (one f16 MFMA per product, fp32 state / accumulation / integrator / node update):
        # x-hat ~1e-4 nm from the fp32 path — the level of the reference's own TF32 GPU path (float32_matmul_precision: high), not its bf16
        # autocast, whose effect on sampling is undefined (the Fabric wrapper is discarded, _sampler.py:62).
        p = str(precision)
        if p in ("32-true", "32"):
            self.reduced_precision = False
        elif p in ("bf16-mixed", "16-mixed", "bf16", "16"):
            # the reference discards the Fabric wrapper (_sampler.py:62): its autocast modes are numerically fp32 for sampling — and so here
            import warnings

            warnings.warn(f"precision={precision!r}: mixed-precision names have no numerical effect on the reference's sampling path "
                          "(sampling/_sampler.py:62) and run fp32-accurate here; the reduced-precision conv is opt-in by 'bf16-true' / '16-true'")
            self.reduced_precision = False
        elif p in ("bf16-true", "16-true"):
            import logging

            logging.getLogger("jamun").info("precision=%s: opt-in reduced-precision hidden-layer conv (f16x1; x-hat 2.5e-5 .. 7.6e-5 nm from the fp32 path)", p)
            self.reduced_precision = True
        else:
            raise NotImplementedError(f"precision={precision!r}: 32-true (default) or a 16-bit name (opt-in reduced-precision conv) are defined for sampling")
        if accelerator == "cpu":
            raise RuntimeError("jamun_amd has no CPU path: sampler.accelerator must be gpu/cuda/auto")
        from . import dist

        self.rank, self.world_size = dist.rank_world()
        self.global_rank = self.rank
        self.is_global_zero = self.rank == 0
        self.device = dist.local_device()
        self.callbacks = list(callbacks) if callbacks is not None else []
        self.loggers = loggers
        self.shard_walkers = shard_walkers
        self.rng = rng
        self.global_step = None
        self.logged: List[tuple] = []  # (step, metrics) pairs passed to log_dict
        self.fabric = self  # `sampler.fabric.global_rank` is read by cmdline/sample.py:86-88

    # ---- the slice of Fabric's logging surface that sampler callbacks use (``callbacks/sampler/_utils.py:49``)
    def log_dict(self, metrics: Dict[str, Any], step: Optional[int] = None) -> None:
        step = self.global_step if step is None else step
        self.logged.append((step, dict(metrics)))
        for lg in (self.loggers if isinstance(self.loggers, (list, tuple)) else ([self.loggers] if self.loggers else [])):
            if hasattr(lg, "log_metrics"):
                lg.log_metrics(dict(metrics), step=step)

    def log(self, name: str, value: Any, step: Optional[int] = None) -> None:
        self.log_dict({name: value}, step=step)

    def call(self, hook: str, **kwargs):
        for cb in self.callbacks:
            fn = getattr(cb, hook, None)
            if callable(fn):
                fn(**kwargs)

    def sample(self, model, batch_sampler, num_batches: int, init_graphs: WalkerBatch, continue_chain: bool = False):
        from . import dist

        model.to(self.device)
        model.eval()
        if hasattr(model, "reduced_precision"):
            model.reduced_precision = self.reduced_precision
        if self.shard_walkers and self.world_size > 1:
            # contiguous blocks of walkers, balanced by modelled cost = atoms x capped in-degree: equal
            # walkers give the even split, ragged batches (MDGen-4AA-like) are cut where the work is
            ptr = init_graphs.ptr.tolist()
            sizes = [ptr[w + 1] - ptr[w] for w in range(init_graphs.num_graphs)]
            lo, hi = dist.shard_range_balanced([n * (min(n - 1, 32) + 2) for n in sizes], self.rank, self.world_size)
            init_graphs = init_graphs.slice_graphs(lo, hi)
        if init_graphs.num_graphs == 0:
            # more ranks than walkers: this rank has nothing to walk but still takes part in every callback's collectives
            self.call("on_sample_start", sampler=self)
            for batch_idx in range(num_batches):
                self.global_step = batch_idx
                self.call("on_after_sample_batch", sample=[], sampler=self)
                self.log("sampler/global_step", batch_idx)
            self.call("on_sample_end", sampler=self)
            return
        init_graphs = init_graphs.to(self.device)
        model_wrapped = ModelSamplingWrapper(model=model, init_graphs=init_graphs, sigma=batch_sampler.sigma, rng=self.rng)
        if hasattr(batch_sampler.mcmc, "rng"):
            batch_sampler.mcmc.rng = self.rng

        y_init = model_wrapped.sample_initial_noisy_positions()
        v_init: Union[str, Tensor] = "gaussian"
        self.call("on_sample_start", sampler=self)
        try:
            self._sample_batches(model, model_wrapped, batch_sampler, num_batches, y_init, v_init, continue_chain)
        except BaseException:
            for cb in self.callbacks:  # writer threads must not outlive a failed run (their own errors do not mask this one)
                close = getattr(cb, "close", None)
                if callable(close):
                    try:
                        close()
                    except Exception:
                        pass
            raise
        self.call("on_sample_end", sampler=self)

    def _sample_batches(self, model, model_wrapped, batch_sampler, num_batches, y_init, v_init, continue_chain):
        with torch.inference_mode():
            for batch_idx in range(num_batches):
                self.global_step = batch_idx
                out = batch_sampler.sample(model=model_wrapped, y_init=y_init, v_init=v_init)
                # (the batch is about to be consumed: synchronise once and surface a conv kernel's error flag HERE, not at some later call)
                if hasattr(model, "sampler_for"):
                    model_wrapped.native_sampler(batch_sampler.sigma).check()
                samples = model_wrapped.unbatch_samples(out)
                if continue_chain:
                    y_init = out["y"].to(model_wrapped.device)
                    v_init = out["v"].to(model_wrapped.device)
                else:
                    y_init = model_wrapped.sample_initial_noisy_positions()
                    v_init = "gaussian"
                self.call("on_after_sample_batch", sample=samples, sampler=self)
                self.log("sampler/global_step", batch_idx)  # (_sampler.py:96)
