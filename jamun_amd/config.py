"""A small Hydra-compatible config composer and instantiator (hydra-core / omegaconf are not installed here).

Supports what the sampling entry point needs of Hydra 1.3 / OmegaConf 2.3 (SURVEY.md §5 "Config / flag system"):
defaults lists (``- group: option``, ``- group/option.yaml``, ``- _self_``, ``- override /group: [..]``, ``null`` options),
``# @package _global_`` overlays, an extra ``--config-dir`` searched before the built-in tree,
command-line overrides ``key=value`` / ``+key=value`` / ``++key=value`` / ``~key``, interpolation ``${a.b}``,
``${oc.env:VAR,default}``, ``${now:%fmt}``, and ``_target_`` / ``_partial_`` / ``_convert_`` instantiation.
``jamun.*`` targets (the reference's import paths) are mapped to this package's mirrors.
"""

from __future__ import annotations

import copy
import datetime
import functools
import importlib
import os
import re
from typing import Any, Dict, List, Optional, Tuple

import yaml

BUILTIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hydra_config")
MISSING = "???"

TARGET_ALIASES = {
    "jamun.model.Denoiser.load_from_checkpoint": "jamun_amd.model.Denoiser.load_from_checkpoint",
    "jamun.sampling.Sampler": "jamun_amd.sampling.Sampler",
    "jamun.sampling.walkjump.SingleMeasurementSampler": "jamun_amd.sampling.SingleMeasurementSampler",
    "jamun.sampling.mcmc.BAOAB": "jamun_amd.sampling.BAOAB",
    "jamun.sampling.mcmc.ABOBA": "jamun_amd.sampling.ABOBA",
    "jamun.data.create_dataset_from_pdbs": "jamun_amd.pdb.create_dataset_from_pdbs",
    "jamun.data.MDtrajDataset": "jamun_amd.pdb.MDtrajDataset",
    "jamun.data.parse_datasets_from_directory": "jamun_amd.pdb.parse_datasets_from_directory",
    "jamun.callbacks.sampler.SaveTrajectoryCallback": "jamun_amd.callbacks.SaveTrajectoryCallback",
    "jamun.callbacks.sampler.MeasureSamplingTimeCallback": "jamun_amd.callbacks.MeasureSamplingTimeCallback",
    "jamun.callbacks.sampler.TrajectoryMetricCallback": "jamun_amd.callbacks.TrajectoryMetricCallback",
    "jamun.sampling.walkjump.MeasurementDependentParametersCallback": "jamun_amd.sampling.MeasurementDependentParametersCallback",
    "jamun.sampling.walkjump.InterpolateParametersCallback": "jamun_amd.sampling.InterpolateParametersCallback",
    "jamun.utils.ModelSamplingWrapper": "jamun_amd.sampling.ModelSamplingWrapper",
}


def _parse_value(s: str) -> Any:
    try:
        return yaml.safe_load(s)
    except yaml.YAMLError:
        return s


def _load_yaml(path: str) -> Tuple[dict, bool]:
    with open(path) as f:
        text = f.read()
    is_global = bool(re.search(r"^#\s*@package\s+_global_", text, flags=re.M))
    return (yaml.safe_load(text) or {}), is_global


def _deep_merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(cfg: dict, path: str, value: Any, must_exist: Optional[bool]) -> None:
    keys = path.split(".")
    d = cfg
    for k in keys[:-1]:
        if not isinstance(d.get(k), dict):
            if must_exist:
                raise KeyError(f"Could not override '{path}': key '{k}' is not in the config (use +{path}=...)")
            d[k] = {}
        d = d[k]
    if must_exist is True and keys[-1] not in d:
        raise KeyError(f"Could not override '{path}'. Key '{keys[-1]}' is not in the config; to append use +{path}=...")
    if must_exist is False and keys[-1] in d:
        raise KeyError(f"Could not append '{path}': it is already in the config; to override use ++{path}=... or {path}=...")
    d[keys[-1]] = value


class Composer:
    def __init__(self, search_dirs: List[str]):
        self.search_dirs = search_dirs

    def _find(self, rel: str) -> Optional[str]:
        rel = rel if rel.endswith(".yaml") else rel + ".yaml"
        for d in self.search_dirs:
            p = os.path.join(d, rel)
            if os.path.exists(p):
                return p
        return None

    def _load_group(self, group: str, option: str, package: Optional[str], choices: Dict[str, Any]) -> dict:
        """Compose one config file (with its own defaults list) and place it at its package."""
        path = self._find(os.path.join(group, option) if group else option)
        if path is None:
            raise FileNotFoundError(f"config '{os.path.join(group, option)}' not found in {self.search_dirs}")
        raw, is_global = _load_yaml(path)
        defaults = raw.pop("defaults", None)
        body = raw
        out: dict = {}
        if defaults is None:
            defaults = ["_self_"]
        elif "_self_" not in defaults:
            defaults = list(defaults) + ["_self_"]
        overrides_here: List[Tuple[str, Any]] = []
        for entry in defaults:
            if entry == "_self_":
                _deep_merge(out, body)
                continue
            if isinstance(entry, str):  # "- sampler/save_trajectory.yaml": a file relative to this group
                sub = self._load_group(group, entry[:-5] if entry.endswith(".yaml") else entry, "", choices)
                _deep_merge(out, sub)
                continue
            (k, v), = entry.items()
            k = k.strip()
            if k.startswith("override "):
                overrides_here.append((k[len("override "):].strip().lstrip("/"), v))
                continue
            sub_group = k.lstrip("/")
            abs_group = sub_group if k.startswith("/") or not group else os.path.join(group, sub_group)
            choice = choices.get(abs_group, v)
            if choice is None or choice == "null":
                continue
            opts = choice if isinstance(choice, list) else [choice]
            for o in opts:
                o = o[:-5] if isinstance(o, str) and o.endswith(".yaml") else o
                sub = self._load_group(abs_group, o, None, choices)
                sub_path = self._find(os.path.join(abs_group, o))
                _, sub_global = _load_yaml(sub_path)
                if sub_global:
                    _deep_merge(out, sub)
                else:
                    node = out
                    for part in sub_group.split("/"):
                        node = node.setdefault(part, {})
                    _deep_merge(node, sub)
        self._pending_overrides = getattr(self, "_pending_overrides", []) + overrides_here
        return out

    def compose(self, config_name: str, overrides: List[str]) -> dict:
        choices: Dict[str, Any] = {}
        value_overrides: List[Tuple[str, str, str]] = []
        deletes: List[str] = []
        for o in overrides:
            if o.startswith("~"):
                deletes.append(o[1:])
                continue
            key, _, val = o.partition("=")
            mode = "++" if key.startswith("++") else ("+" if key.startswith("+") else "")
            key = key.lstrip("+")
            # group choice if a directory of that name exists in the search path
            if mode == "" and any(os.path.isdir(os.path.join(d, key.replace(".", "/"))) for d in self.search_dirs):
                choices[key.replace(".", "/")] = _parse_value(val)
            else:
                value_overrides.append((mode, key, val))
        # two passes: the experiment overlay may carry "override /group" entries that change earlier choices
        self._pending_overrides = []
        cfg = self._load_group("", config_name, "", choices)
        if self._pending_overrides:
            for g, v in self._pending_overrides:
                choices.setdefault(g, v)
            self._pending_overrides = []
            cfg = self._load_group("", config_name, "", choices)
        for mode, key, val in value_overrides:
            _set_path(cfg, key, _parse_value(val), must_exist=True if mode == "" else (False if mode == "+" else None))
        for key in deletes:
            d = cfg
            parts = key.split(".")
            for k in parts[:-1]:
                d = d.get(k, {})
            d.pop(parts[-1], None)
        return cfg


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def resolve(cfg: dict, now: Optional[datetime.datetime] = None, throw_on_missing: bool = True) -> dict:
    now = now or datetime.datetime.now()

    def lookup(path: str):
        d: Any = cfg
        for k in path.split("."):
            if isinstance(d, list):
                d = d[int(k)]
            else:
                if k not in d:
                    raise KeyError(f"interpolation key '{path}' not found")
                d = d[k]
        return d

    def res_str(s: str, depth=0):
        if depth > 50:
            raise RecursionError(f"interpolation too deep: {s}")
        m = _INTERP.fullmatch(s)
        if m:
            return res_expr(m.group(1).strip(), depth)
        while True:
            m = None
            for m in _INTERP.finditer(s):
                pass
            if m is None:
                return s
            s = s[: m.start()] + str(res_expr(m.group(1).strip(), depth)) + s[m.end():]

    def res_expr(expr: str, depth: int):
        if expr.startswith("oc.env:"):
            var, _, default = expr[len("oc.env:"):].partition(",")
            if var.strip() in os.environ:
                return os.environ[var.strip()]
            if not _:
                raise KeyError(f"environment variable {var} not set")
            default = default.strip().strip('"').strip("'")
            return res_any(_parse_value(default) if not _INTERP.search(default) else default, depth + 1)
        if expr.startswith("now:"):
            return now.strftime(expr[4:])
        return res_any(lookup(expr), depth + 1)

    def res_any(v, depth=0):
        if isinstance(v, str):
            if v == MISSING and throw_on_missing:
                raise ValueError("Missing mandatory value (???) in config")
            return res_str(v, depth) if "${" in v else v
        if isinstance(v, dict):
            return {k: res_any(x, depth) for k, x in v.items()}
        if isinstance(v, list):
            return [res_any(x, depth) for x in v]
        return v

    out = {}
    for k, v in cfg.items():
        try:
            out[k] = res_any(v)
        except ValueError as e:
            raise ValueError(f"{e} (under key '{k}')") from None
    return out


def _locate(target: str):
    target = TARGET_ALIASES.get(target, target)
    parts = target.split(".")
    for i in range(len(parts), 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:i]))
        except ModuleNotFoundError:
            continue
        for p in parts[i:]:
            obj = getattr(obj, p)
        return obj
    raise ImportError(f"cannot locate _target_ {target}")


def instantiate(node: Any, **kwargs):
    """``hydra.utils.instantiate`` for resolved plain containers."""
    if isinstance(node, list):
        return [instantiate(x) for x in node]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    fn = _locate(node["_target_"])
    args = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_", "_convert_", "_recursive_")}
    args.update(kwargs)
    if node.get("_partial_"):
        return functools.partial(fn, **args)
    return fn(**args)


def instantiate_dict_cfg(cfg: Optional[dict]) -> list:
    """``jamun.hydra.instantiate_dict_cfg`` (``hydra/utils.py:11-29``): instantiate every entry that has a ``_target_``."""
    out = []
    if not cfg:
        return out
    for _, v in cfg.items():
        if isinstance(v, dict) and "_target_" in v:
            out.append(instantiate(v))
    return out
