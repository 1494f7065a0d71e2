"""Checkpoint location and loading for the sampling path.

``find_checkpoint`` keeps the ``checkpoint_dir`` / ``checkpoint_type`` semantics of
``/root/reference/src/jamun/utils/checkpoint.py:25-68`` (the wandb lookup is out of scope offline).
``load_checkpoint_file`` reads a Lightning ``.ckpt`` (a ``torch.save``d dict with ``state_dict`` and
``hyper_parameters``).  Real JAMUN checkpoints pickle ``functools.partial`` / ``omegaconf`` objects that reference
modules which are not installed here (hydra, omegaconf, e3nn, lightning, jamun.*); those are resolved by a tolerant
unpickler that substitutes inert stand-ins and keeps only plain keyword data.
"""

from __future__ import annotations

import os
import pickle
import re
from typing import Optional

import torch


def find_checkpoint_in_directory(checkpoint_dir: str, checkpoint_type: str) -> str:
    """``utils/checkpoint.py:25-50``: 'last' -> last.ckpt, 'best_so_far' -> highest ``epoch=N-...ckpt``, or a file name."""
    if checkpoint_type.endswith(".ckpt"):
        path = os.path.join(checkpoint_dir, checkpoint_type)
        if not os.path.exists(path):
            raise ValueError(f"Checkpoint {path} not found")
        return path
    if checkpoint_type == "last":
        path = os.path.join(checkpoint_dir, "last.ckpt")
        if not os.path.exists(path):
            raise ValueError(f"Checkpoint {path} not found")
        return path
    if checkpoint_type == "best_so_far":
        best_epoch, best = -1, None
        for f in sorted(os.listdir(checkpoint_dir)):
            m = re.match(r"epoch=(\d+)-.*\.ckpt$", f)
            if m and int(m.group(1)) > best_epoch:
                best_epoch, best = int(m.group(1)), f
        if best is None:
            raise ValueError(f"No epoch=*.ckpt checkpoint found in {checkpoint_dir}")
        return os.path.join(checkpoint_dir, best)
    raise ValueError(f"Invalid checkpoint type: {checkpoint_type}")


def find_checkpoint(wandb_train_run_path: Optional[str] = None, checkpoint_dir: Optional[str] = None, checkpoint_type: Optional[str] = None) -> str:
    """``utils/checkpoint.py:53-68``: exactly one of ``wandb_train_run_path`` / ``checkpoint_dir``."""
    if wandb_train_run_path and checkpoint_dir:
        raise ValueError("Exactly one of wandb_train_run_path or checkpoint_dir must be provided.")
    if not wandb_train_run_path and not checkpoint_dir:
        raise ValueError("Must provide one of wandb_train_run_path or checkpoint_dir")
    if wandb_train_run_path:
        raise NotImplementedError("wandb run lookup is not available offline; pass checkpoint_dir=... wandb_train_run_path=null")
    return find_checkpoint_in_directory(checkpoint_dir, checkpoint_type or "last")


class _Inert:
    """Stand-in for a class from a module that is not installed: keeps constructor/state data, does nothing."""

    def __init__(self, *args, **kwargs):
        self._args, self._kwargs = args, kwargs

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    # dict / list subclasses (e.g. Lightning's AttributeDict) are rebuilt item by item
    def __setitem__(self, key, value):
        self.__dict__.setdefault("_items", {})[key] = value

    def append(self, value):
        self.__dict__.setdefault("_list", []).append(value)

    def extend(self, values):
        self.__dict__.setdefault("_list", []).extend(values)

    def items(self):
        plain = to_plain(self)
        return plain.items() if isinstance(plain, dict) else []


def to_plain(obj):
    """Plain Python data out of what the tolerant unpickler produced.  A Lightning checkpoint written under Hydra's default
    ``_convert_="none"`` nests ``omegaconf.DictConfig`` / ``ListConfig`` objects in its hyper-parameters, whose pickled state
    is ``{"_metadata", "_parent", "_content"}`` with ``_content`` a dict / list of NODE objects (``AnyNode``, ``IntegerNode``,
    ...: state ``{"_metadata", "_parent", "_val"}``).  Containers become dicts / lists, value nodes their ``_val``;
    ``functools.partial`` keeps its (possibly inert) callable and gets plain arguments; anything else is returned as is."""
    import functools

    if isinstance(obj, _Inert):
        st = obj.__dict__.get("_state")
        if isinstance(st, tuple) and len(st) == 2 and isinstance(st[0], dict):  # (dict state, slots state)
            st = st[0]
        if isinstance(st, dict):
            if "_content" in st:
                return to_plain(st["_content"])
            if "_val" in st:
                return to_plain(st["_val"])
        if "_items" in obj.__dict__:
            return to_plain(obj.__dict__["_items"])
        if "_list" in obj.__dict__:
            return to_plain(obj.__dict__["_list"])
        return obj
    if isinstance(obj, dict):
        return {to_plain(k): to_plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_plain(v) for v in obj)
    if isinstance(obj, functools.partial):
        return functools.partial(obj.func, *[to_plain(a) for a in obj.args], **{k: to_plain(v) for k, v in obj.keywords.items()})
    return obj


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except Exception:
            return type(name, (_Inert,), {"__module__": module})


class _TolerantPickle:
    """``pickle_module`` for ``torch.load``: unknown classes become inert stand-ins instead of import errors."""

    __name__ = "pickle"
    Unpickler = _TolerantUnpickler
    Pickler = pickle.Pickler

    @staticmethod
    def load(f, **kw):
        return _TolerantUnpickler(f, **kw).load()


def load_checkpoint_file(path: str) -> dict:
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    except (ModuleNotFoundError, AttributeError, ImportError):
        ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)
    if not isinstance(ckpt, dict) or "state_dict" not in ckpt or "hyper_parameters" not in ckpt:
        raise RuntimeError(f"{path} is not a Lightning checkpoint with state_dict + hyper_parameters")
    return ckpt


def w3j_111_sign_from_state_dict(state_dict) -> float:
    """Cross-check of the Clebsch-Gordan convention against a real checkpoint.

    e3nn keeps the real Wigner-3j tensors it compiled a tensor product with as buffers named ``..._w3j_{l1}_{l2}_{l3}``.
    This path assumes ``w3j(1,1,1) = +epsilon_ijk / sqrt(6)`` and ``w3j(0,1,1) = w3j(1,0,1) = w3j(1,1,0) = delta / sqrt(3)``
    (``oracle/e3.py``); a checkpoint that carries the buffers either confirms that (+1.0), asks for the opposite sign of the
    cross-product path (-1.0, passed to the kernels as ``jamun_hparams.w3j_111_sign``), or disagrees in a way this code does
    not understand (``ValueError``).  Without such buffers the assumed convention is returned."""
    import math

    import torch

    eps = torch.zeros(3, 3, 3, dtype=torch.float64)
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[i, j, k], eps[i, k, j] = 1.0, -1.0
    eps /= math.sqrt(6.0)
    delta = torch.eye(3, dtype=torch.float64) / math.sqrt(3.0)
    sign = None
    for name, value in state_dict.items():
        if not torch.is_tensor(value):
            continue
        tail = name.rsplit(".", 1)[-1]
        if not tail.startswith("_w3j_"):
            continue
        w = value.detach().to("cpu", torch.float64)
        if tail == "_w3j_1_1_1" and w.numel() == 27:
            w = w.reshape(3, 3, 3)
            if torch.allclose(w, eps, atol=1e-5):
                s = 1.0
            elif torch.allclose(w, -eps, atol=1e-5):
                s = -1.0
            else:
                raise ValueError(f"{name}: not +-epsilon/sqrt(6) in the assumed real basis")
            if sign is not None and s != sign:
                raise ValueError("inconsistent _w3j_1_1_1 buffers in the checkpoint")
            sign = s
        elif tail in ("_w3j_0_1_1", "_w3j_1_0_1", "_w3j_1_1_0") and w.numel() == 9:
            if not torch.allclose(w.reshape(3, 3), delta, atol=1e-5):
                raise ValueError(f"{name}: not delta/sqrt(3) in the assumed real basis")
        elif tail == "_w3j_0_0_0" and w.numel() == 1:
            if abs(float(w.reshape(())) - 1.0) > 1e-5:
                raise ValueError(f"{name}: expected 1")
    return 1.0 if sign is None else sign
