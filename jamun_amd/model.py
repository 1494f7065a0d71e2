"""``Denoiser`` — host-side mirror of ``jamun.model.Denoiser`` for the sampling path.

Same protocol as the reference (``/root/reference/src/jamun/model/denoiser.py:111-217`` and what
``utils/sampling_wrapper.py:17-34`` needs): ``load_from_checkpoint``, ``score(graph, sigma)``,
``xhat(graph, sigma)``, ``device``, ``eval()``, ``to()``.  All arithmetic runs in the HIP library; this class only
holds the checkpoint tensors and caches one native sampler per (sigma, walker batch).
"""

from __future__ import annotations

import functools
from typing import Dict, Optional, Tuple, Union

import torch

from .data import WalkerBatch
from .native import NativeModel, NativeSampler

_PREFIXES = ("g._orig_mod.", "g.")  # torch.compile'd (use_torch_compile=True, the default) and plain modules


def strip_prefix(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in state_dict.items():
        for p in _PREFIXES:
            if k.startswith(p):
                out[k[len(p) :]] = v
                break
    if not out:
        raise RuntimeError("checkpoint state_dict has no 'g.' / 'g._orig_mod.' keys: not a JAMUN Denoiser checkpoint")
    return out


def _kw(obj) -> dict:
    """Keyword dict of a hyper-parameter entry: plain dict, functools.partial (Hydra ``_partial_``) or DictConfig-like.
    Values are plain Python data (ints, strs, lists, nested dicts / partials), also when the checkpoint carried omegaconf
    containers that were unpickled without omegaconf (``checkpoint.to_plain``)."""
    from .checkpoint import to_plain

    obj = to_plain(obj)
    if isinstance(obj, functools.partial):
        return dict(obj.keywords)
    if isinstance(obj, dict):
        return dict(obj)
    if hasattr(obj, "keywords"):
        return {k: to_plain(v) for k, v in dict(obj.keywords).items()}
    if hasattr(obj, "items"):
        return {k: to_plain(v) for k, v in dict(obj.items()).items()}
    raise TypeError(f"cannot read hyper-parameters from {type(obj)}")


class Denoiser:
    """The denoiser of the walk-jump sampler (inference only)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], arch: dict, max_radius: float, average_squared_distance: float,
                 mean_center: bool = True, **_ignored):
        self.arch = dict(arch)
        self.max_radius = float(max_radius)
        self.average_squared_distance = float(average_squared_distance)
        self.mean_center = bool(mean_center)
        self.state_dict_ = {k: v.detach().to("cpu") for k, v in state_dict.items()}
        self._native = NativeModel(self.state_dict_, self.arch, self.max_radius, self.average_squared_distance, self.mean_center)
        self._device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self._samplers: Dict[Tuple[float, int, str, bool], NativeSampler] = {}
        # Opt-in reduced precision of the hidden-layer conv (``jamun_tuning.f16x1``: one f16 MFMA per product instead of three; everything else
        # stays fp32).  Set by ``Sampler(precision="bf16-true" | "16-true")``; never the default — x-hat then sits 2.5e-5 .. 7.6e-5 nm from the fp32 path on the test batches (asserted <= 1e-3 nm).
        self.reduced_precision = False

    # ---- construction ---------------------------------------------------------------------------------------
    @classmethod
    def from_checkpoint_dict(cls, ckpt: dict) -> "Denoiser":
        from .checkpoint import to_plain

        hp = to_plain(ckpt["hyper_parameters"])
        arch = _kw(hp["arch"])
        # the output-head factory is fixed (e3conv.yaml:24-33); the hidden-layer factory's conv is Conv (e3conv.yaml) or SeparableConv
        # (e3conv_separable.yaml)
        try:
            conv = _kw(arch.get("hidden_layer_factory", {})).get("conv")
            conv_name = repr(getattr(conv, "func", None) or (conv.get("_target_") if hasattr(conv, "get") else conv))
        except TypeError:
            conv_name = ""
        if "Experimental" in conv_name:
            raise NotImplementedError(f"hidden_layer_factory.conv = {conv_name}: jamun.e3tools.nn.Conv and SeparableConv are implemented")
        state = strip_prefix(ckpt["state_dict"])
        # SeparableConv (e3conv_separable.yaml:14-19; e3tools/nn/_conv.py:122-135): named by the factory and recognisable by the
        # point-wise Linear of its tensor product among the parameters
        separable = "Separable" in conv_name or any(k.endswith("gated_conv.f.f.tp.lin.weight") for k in state)
        arch = {k: v for k, v in arch.items() if k not in ("hidden_layer_factory", "output_head_factory", "_target_", "_partial_")}
        arch["separable_conv"] = bool(separable)
        return cls(
            state,
            arch=arch,
            max_radius=hp["max_radius"],
            average_squared_distance=hp["average_squared_distance"],
            mean_center=hp.get("mean_center", True),
        )

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, map_location=None, **_) -> "Denoiser":
        """``jamun.model.Denoiser.load_from_checkpoint`` (``hydra_config/model/denoiser_pretrained.yaml:1-2``)."""
        from .checkpoint import load_checkpoint_file

        return cls.from_checkpoint_dict(load_checkpoint_file(checkpoint_path))

    # ---- nn.Module-ish surface the Sampler touches -------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self._device

    def to(self, device) -> "Denoiser":
        self._device = torch.device(device)
        return self

    def eval(self) -> "Denoiser":
        return self

    def parameters(self):
        return iter(self.state_dict_.values())

    # ---- forward ----------------------------------------------------------------------------------------------
    def sampler_for(self, graph: WalkerBatch, sigma: float) -> NativeSampler:
        dev = graph.pos.device
        key = (float(sigma), graph.topology_id, str(dev), bool(self.reduced_precision))
        s = self._samplers.get(key)
        if s is None:
            s = NativeSampler(self._native, float(sigma), graph, dev, tuning={"f16x1": 1} if self.reduced_precision else None)
            if len(self._samplers) >= 4:  # a sampling run uses one sigma and one batch
                self._samplers.pop(next(iter(self._samplers)))
            self._samplers[key] = s
        return s

    def xhat(self, y: WalkerBatch, sigma: Union[float, torch.Tensor]) -> WalkerBatch:
        """Denoised graph (``denoiser.py:203-217``)."""
        sigma = float(sigma)
        return y.with_pos(self.sampler_for(y, sigma).xhat(y.pos))

    def score(self, y: WalkerBatch, sigma: Union[float, torch.Tensor]) -> torch.Tensor:
        """Score ``(xhat(y) - y) / sigma^2`` (``denoiser.py:111-114``)."""
        sigma = float(sigma)
        return self.sampler_for(y, sigma).score(y.pos)
