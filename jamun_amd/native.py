"""Thin object wrappers over the C ABI (``include/jamun_hip.h``): model / sampler handles and the stand-alone operators.

PyTorch is used only for device memory and streams: every call passes ``tensor.data_ptr()`` and
``torch.cuda.current_stream().cuda_stream`` to the HIP library.
"""

from __future__ import annotations

import ctypes as C
import os
import functools
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .data import WalkerBatch


@functools.lru_cache(maxsize=None)
def normalize2mom_const(name: str) -> float:
    """e3nn ``normalize2mom`` constant: E_{z~N(0,1)}[f(z)^2]^(-1/2) from 1e6 float64 draws of ``Generator().manual_seed(0)``.

    These Python floats are not stored in checkpoints (e3nn recomputes them at module construction), so the
    loader recomputes them the same way.
    """
    f = {"leaky_relu": lambda z: torch.nn.functional.leaky_relu(z, 0.01), "sigmoid": torch.sigmoid}[name]
    gen = torch.Generator(device="cpu").manual_seed(0)
    z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
    cst = f(z).pow(2).mean().pow(-0.5).item()
    return 1.0 if abs(cst - 1) < 1e-4 else cst


def parse_hidden_irreps(s: str) -> Tuple[int, int]:
    m0 = m1 = 0
    for part in str(s).split("+"):
        part = part.strip()
        if not part:
            continue
        mul, ir = part.split("x") if "x" in part else ("1", part)
        ir = ir.strip()
        if ir == "0e":
            m0 += int(mul)
        elif ir == "1e":
            m1 += int(mul)
        else:
            raise NotImplementedError(f"irreps_hidden term {part!r}: only 0e and 1e irreps are supported")
    return m0, m1


def _stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else int(t.data_ptr())


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); jamun_amd has no CPU path")
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t


class NativeModel:
    """Owns a ``jamun_model*`` built from a reference-format state dict (names without the ``g.`` prefix)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], arch: dict, max_radius: float, average_squared_distance: float, mean_center: bool = True):
        lib = _lib.load()
        if str(arch.get("irreps_sh", "1x0e + 1x1e")).replace(" ", "") not in ("1x0e+1x1e",):
            raise NotImplementedError("only irreps_sh = 1x0e + 1x1e is supported")
        if str(arch.get("irreps_out", "1x1e")).replace(" ", "") not in ("1x1e", "1e"):
            raise NotImplementedError("only irreps_out = 1x1e is supported")
        if not arch.get("use_residue_information", True):
            raise NotImplementedError("SimpleAtomEmbedding (use_residue_information=False) is not supported")
        m0, m1 = parse_hidden_irreps(arch["irreps_hidden"])
        hp = _lib.jamun_hparams()
        hp.n_layers = int(arch["n_layers"])
        hp.mul0, hp.mul1 = m0, m1
        hp.edge_attr_dim = int(arch["edge_attr_dim"])
        dims = [arch["atom_type_embedding_dim"], arch["atom_code_embedding_dim"], arch["residue_code_embedding_dim"], arch["residue_index_embedding_dim"]]
        names = ["atom_type", "atom_code", "residue_code", "residue_index"]
        for i in range(4):
            hp.emb_dim[i] = int(dims[i])
            hp.emb_rows[i] = int(state_dict[f"atom_embedder.{names[i]}_embedding.weight"].shape[0])
        hp.use_residue_sequence_index = int(bool(arch.get("use_residue_sequence_index", False)))
        hp.mean_center = int(bool(mean_center))
        hp.max_radius = float(max_radius)
        hp.average_squared_distance = float(average_squared_distance)
        hp.act_scalar_const = normalize2mom_const("leaky_relu")
        hp.act_gate_const = normalize2mom_const("sigmoid")
        from .checkpoint import w3j_111_sign_from_state_dict

        hp.w3j_111_sign = w3j_111_sign_from_state_dict(state_dict)  # +1 unless the checkpoint's e3nn buffers say otherwise
        hp.separable = int(bool(arch.get("separable_conv", False)))
        self.hparams_struct = hp
        keep = []  # keep host buffers alive during the call
        arr = (_lib.jamun_tensor * len(state_dict))()
        n = 0
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.is_floating_point():
                continue
            t = v.detach().to("cpu", torch.float32).contiguous()
            keep.append(t)
            arr[n].name = k.encode()
            arr[n].data = C.cast(t.data_ptr(), C.POINTER(C.c_float))
            arr[n].numel = t.numel()
            n += 1
        handle = C.c_void_p()
        _lib.check(lib.jamun_model_create(C.byref(hp), arr, n, C.byref(handle)))
        self._h = handle
        self._lib = lib

    @property
    def handle(self):
        return self._h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.jamun_model_destroy(h)


# Kernel-selection switches (``jamun_tuning`` in include/jamun_hip.h) applied to every sampler created in this process.  Empty in
# production; tests set entries (``monkeypatch.setitem(native.TUNING, "no_mf", 1)``) to run one implementation of a block against another,
# and profiling scripts may set ``JAMUN_TUNING="no_mf,dg_kgroups=2"`` — the ONLY environment variable that reaches kernel selection,
# read here, once, at import (the C library reads none).
TUNING: dict = {}
for _item in filter(None, (os.environ.get("JAMUN_TUNING") or "").split(",")):
    _k, _, _v = _item.strip().partition("=")
    TUNING[_k] = int(_v) if _v else 1


def make_tuning(extra: Optional[dict] = None) -> "_lib.jamun_tuning":
    t = _lib.jamun_tuning()
    names = {n for n, _ in _lib.jamun_tuning._fields_} - {"reserved"}
    for k, v in {**TUNING, **(extra or {})}.items():
        if k not in names:
            raise ValueError(f"unknown tuning switch {k!r} (known: {sorted(names)})")
        setattr(t, k, int(v))
    return t


class NativeSampler:
    """Owns a ``jamun_sampler*``: packed weights for one sigma + work buffers for one walker batch."""

    def __init__(self, model: NativeModel, sigma: float, batch: WalkerBatch, device: torch.device, tuning: Optional[dict] = None):
        lib = _lib.load()
        if device.type != "cuda":
            raise RuntimeError("jamun_amd needs a GPU device (cuda / ROCm); there is no CPU path")
        self.device = device
        self.sigma = float(sigma)
        self.n_atoms = batch.num_nodes
        self.n_graphs = batch.num_graphs
        topo = _lib.jamun_topology()
        i32 = lambda t: t.detach().to("cpu", torch.int32).contiguous()
        keep = [i32(batch.ptr), i32(batch.atom_type_index), i32(batch.atom_code_index), i32(batch.residue_code_index), i32(batch.residue_sequence_index)]
        b = batch.bonds.detach().to("cpu", torch.int64).reshape(2, -1)
        bs, bd = b[0].contiguous(), b[1].contiguous()
        topo.n_atoms, topo.n_graphs = self.n_atoms, self.n_graphs
        pi = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_int32))
        topo.ptr, topo.atom_type_index, topo.atom_code_index, topo.residue_code_index, topo.residue_sequence_index = [pi(t) for t in keep]
        topo.n_bonds = int(bs.numel())
        topo.bond_src = C.cast(bs.data_ptr(), C.POINTER(C.c_int64))
        topo.bond_dst = C.cast(bd.data_ptr(), C.POINTER(C.c_int64))
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.jamun_sampler_create(model.handle, C.c_float(self.sigma), C.byref(topo), C.byref(make_tuning(tuning)), C.byref(handle)))
        self._h = handle
        self._lib = lib
        self._model = model  # keep alive

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.jamun_sampler_destroy(h)

    def _new(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def xhat(self, y: torch.Tensor) -> torch.Tensor:
        y = _dev_f32(y, "y")
        assert y.shape == (self.n_atoms, 3), y.shape
        out = self._new(self.n_atoms, 3)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.jamun_xhat(self._h, _ptr(y), _ptr(out), _stream()))
        return out

    def score(self, y: torch.Tensor) -> torch.Tensor:
        y = _dev_f32(y, "y")
        assert y.shape == (self.n_atoms, 3), y.shape
        out = self._new(self.n_atoms, 3)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.jamun_score(self._h, _ptr(y), _ptr(out), _stream()))
        return out

    def walk(self, integrator: str, y: torch.Tensor, v: torch.Tensor, params: "_lib.jamun_mcmc_params", noise: Optional[torch.Tensor], seed: int,
             save_trajectory: bool, want_xhat_traj: bool = True, want_xhat: bool = True):
        """Runs ``steps-1`` fused iterations on the current stream.  ``y`` and ``v`` are updated in place."""
        assert integrator in ("baoab", "aboba")
        ny, nsb, nsa = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self._lib.jamun_num_frames(C.byref(params), C.byref(ny), C.byref(nsb), C.byref(nsa)))
        n = self.n_atoms
        y_traj = score_traj = xhat_traj = None
        ns = nsb.value if integrator == "baoab" else nsa.value
        if save_trajectory:
            y_traj = self._new(ny.value, n, 3)
            xhat_traj = self._new(ny.value, n, 3) if want_xhat_traj else None
            score_traj = self._new(ns, n, 3)
        elif integrator == "baoab":
            score_traj = self._new(1, n, 3)  # the reference always keeps the initial score (_splitting.py:155)
        xhat = self._new(n, 3) if want_xhat else None
        if noise is not None:
            noise = _dev_f32(noise, "noise")
            assert noise.shape == (max(params.steps - 1, 0), n, 3), noise.shape
        fn = self._lib.jamun_walk_baoab if integrator == "baoab" else self._lib.jamun_walk_aboba
        with torch.cuda.device(self.device):
            _lib.check(fn(self._h, _ptr(y), _ptr(v), C.byref(params), _ptr(noise), C.c_uint64(seed & (2**64 - 1)),
                          _ptr(y_traj), _ptr(score_traj), _ptr(xhat_traj), _ptr(xhat), _stream()))
        return y_traj, score_traj, xhat_traj, xhat

    def check(self) -> None:
        """Synchronise and raise if a conv kernel flagged an edge table it cannot represent (``jamun_sampler_check``)."""
        with torch.cuda.device(self.device):
            _lib.check(self._lib.jamun_sampler_check(self._h, _stream()))

    def stats(self) -> dict:
        st = _lib.jamun_stats()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.jamun_sampler_stats(self._h, C.byref(st), _stream()))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def profile_enable(self, on: bool = True, classes=None, every: int = 1) -> None:
        """Record HIP events around the forward's launches: all classes, or only the named ones (``_lib.PROF_CLASSES``); ``every`` > 1
        samples every ``every``-th launch of a class instead of all of them (``jamun_profile_sample``)."""
        code = int(bool(on))
        if on and classes is not None:
            code = 0
            for c in classes:
                code |= 1 << (_lib.PROF_CLASSES.index(c) + 1)
        _lib.check(self._lib.jamun_profile_sample(self._h, int(every)))
        _lib.check(self._lib.jamun_profile_enable(self._h, code))

    def profile_read(self) -> dict:
        """{class: (total_ms, launches)} from HIP events recorded on the launch stream; synchronises the stream."""
        n = len(_lib.PROF_CLASSES)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.jamun_profile_read(self._h, ms, cnt, _stream()))
        return {name: (ms[i], cnt[i]) for i, name in enumerate(_lib.PROF_CLASSES)}

    def build_edges(self, y: torch.Tensor) -> None:
        """``jamun_build_edges``: radius graph + bonded edges, edge geometry and the radial MLPs' hidden layer for positions
        ``y`` (``Denoiser.add_edges`` + ``E3Conv.forward`` up to the blocks); the edge table stays inside the sampler."""
        y = _dev_f32(y, "y")
        assert y.shape == (self.n_atoms, 3)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.jamun_build_edges(self._h, _ptr(y), _stream()))

    def conv_block(self, layer: int, x_in: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``jamun_conv_block``: one block of ``E3Conv`` on caller-owned node features and the current edge table — layer 0: the
        initial projector on the sampler's own noise-scaled embedding (``x_in`` must be None); layer l >= 1:
        ``w_l x + (1 - w_l) ConvBlock_l(s_l x)`` for ``x_in [n_atoms, mul0 + 3 mul1]``."""
        hp = self._model.hparams_struct
        out = self._new(self.n_atoms, hp.mul0 + 3 * hp.mul1)
        if x_in is not None:
            x_in = _dev_f32(x_in, "x_in")
            assert x_in.shape == out.shape, x_in.shape
        with torch.cuda.device(self.device):
            _lib.check(self._lib.jamun_conv_block(self._h, int(layer), _ptr(x_in), _ptr(out), _stream()))
        return out

    def debug_read(self, what: int, layer: int = 0) -> torch.Tensor:
        width = {0: None, 1: 1, 2: 3}[what]
        if what == 0:
            hp = self._model.hparams_struct
            width = hp.mul0 + 3 * hp.mul1
        out = self._new(self.n_atoms, width)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.jamun_debug_read(self._h, what, layer, _ptr(out), _stream()))
        return out


# ---- stand-alone operators --------------------------------------------------------------------------------------


def mean_center(pos: torch.Tensor, ptr: torch.Tensor) -> torch.Tensor:
    """``jamun.utils.mean_center`` on device (``src/jamun/utils/mean_center.py:7-12``)."""
    lib = _lib.load()
    pos = _dev_f32(pos, "pos")
    ptr = ptr.to(pos.device, torch.int32).contiguous()
    out = torch.empty_like(pos)
    with torch.cuda.device(pos.device):
        _lib.check(lib.jamun_mean_center(_ptr(pos), _ptr(ptr), ptr.numel() - 1, _ptr(out), _stream()))
    return out


def radius_graph(pos: torch.Tensor, r: float, ptr: torch.Tensor, stride: int = 33):
    """Neighbour table ``(nbr [N,stride] i32, deg [N] i32)``; see ``jamun_radius_graph``."""
    lib = _lib.load()
    pos = _dev_f32(pos, "pos")
    ptr = ptr.to(pos.device, torch.int32).contiguous()
    n = pos.shape[0]
    nbr = torch.full((n, stride), -1, dtype=torch.int32, device=pos.device)
    deg = torch.zeros(n, dtype=torch.int32, device=pos.device)
    with torch.cuda.device(pos.device):
        _lib.check(lib.jamun_radius_graph(_ptr(pos), _ptr(ptr), ptr.numel() - 1, n, C.c_float(r), stride, _ptr(nbr), _ptr(deg), _stream()))
    return nbr, deg


def radius_graph_edge_index(pos: torch.Tensor, r: float, ptr: torch.Tensor) -> torch.Tensor:
    """``edge_index [2,E]`` (src = neighbour, dst = centre), ordered by centre then neighbour — the reference layout."""
    nbr, deg = radius_graph(pos, r, ptr)
    n, s = nbr.shape
    mask = torch.arange(s, device=nbr.device)[None, :] < deg[:, None]
    dst = torch.arange(n, device=nbr.device)[:, None].expand(n, s)[mask]
    src = nbr[mask].long()
    return torch.stack([src, dst.long()])


def edge_geometry(pos_scaled: torch.Tensor, edge_index: torch.Tensor, radial_cutoff: float, n_basis: int = 32):
    """``(edge_sh [E,4], radial [E,n_basis])`` of ``E3Conv.forward`` (``arch/e3conv.py:114-123``) for an explicit edge list
    ``edge_index [2,E]`` (src, dst) and positions already scaled by ``c_in``; see ``jamun_edge_geometry``."""
    lib = _lib.load()
    pos = _dev_f32(pos_scaled, "pos_scaled")
    ei = edge_index.to(pos.device, torch.int64).contiguous()
    E = int(ei.shape[1])
    sh = torch.empty((E, 4), dtype=torch.float32, device=pos.device)
    radial = torch.empty((E, n_basis), dtype=torch.float32, device=pos.device)
    with torch.cuda.device(pos.device):
        _lib.check(lib.jamun_edge_geometry(_ptr(pos), int(pos.shape[0]), _ptr(ei[0].contiguous()), _ptr(ei[1].contiguous()), E,
                                           C.c_float(radial_cutoff), n_basis, _ptr(sh), _ptr(radial), _stream()))
    return sh, radial


def node_linear(x: torch.Tensor, weight: torch.Tensor, in0: int, in1: int, out0: int, out1: int) -> torch.Tensor:
    """e3nn ``o3.Linear`` between ``in0 x0e + in1 x1e`` and ``out0 x0e + out1 x1e`` with the flat e3nn weight; see ``jamun_node_linear``."""
    lib = _lib.load()
    x = _dev_f32(x, "x")
    w = _dev_f32(weight, "weight")
    assert x.shape[1] == in0 + 3 * in1, x.shape
    out = torch.empty((x.shape[0], out0 + 3 * out1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.jamun_node_linear(_ptr(x), x.shape[0], in0, in1, out0, out1, _ptr(w), int(w.numel()), _ptr(out), _stream()))
    return out


def philox_normal(n: int, seed: int, iteration: int, device, first_atom: int = 0) -> torch.Tensor:
    """``[n, 3]`` standard-normal draws of integrator iteration ``iteration`` under ``seed``: exactly what the walks use in place of
    the reference's ``torch.randn_like`` (``functional/_splitting.py:161``) when no noise tensor is passed; see ``jamun_philox_normal``."""
    lib = _lib.load()
    device = torch.device(device)
    out = torch.empty((n, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.jamun_philox_normal(_ptr(out), int(n), C.c_uint64(seed & (2**64 - 1)), C.c_uint32(iteration), C.c_uint32(first_atom), _stream()))
    return out


def scatter_mean(src: torch.Tensor, seg_ptr: torch.Tensor, n_out: int) -> torch.Tensor:
    """Mean of destination-sorted rows: ``out[d] = mean(src[seg_ptr[d]:seg_ptr[d+1]])`` (empty -> 0)."""
    lib = _lib.load()
    src = _dev_f32(src, "src")
    seg_ptr = seg_ptr.to(src.device, torch.int32).contiguous()
    assert seg_ptr.numel() == n_out + 1
    width = int(src.shape[1]) if src.ndim > 1 else 1
    out = torch.empty((n_out, width), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(lib.jamun_scatter_mean(_ptr(src), _ptr(seg_ptr), n_out, width, _ptr(out), _stream()))
    return out


def make_mcmc_params(steps, delta, friction, M, inverse_temperature, score_fn_clip, save_every_n_steps=1, burn_in_steps=0):
    p = _lib.jamun_mcmc_params()
    p.steps = int(steps)
    p.save_every_n_steps = int(save_every_n_steps)
    p.burn_in_steps = int(burn_in_steps)
    p.has_clip = int(score_fn_clip is not None)
    p.delta, p.friction, p.M = float(delta), float(friction), float(M)
    p.inverse_temperature = float(inverse_temperature)
    p.score_fn_clip = float(score_fn_clip) if score_fn_clip is not None else 0.0
    return p


def baoab_pre(y, v, psi, noise, params):
    lib = _lib.load()
    with torch.cuda.device(y.device):
        _lib.check(lib.jamun_baoab_pre(_ptr(y), _ptr(v), _ptr(psi), _ptr(noise), y.shape[0], C.byref(params), _stream()))


def baoab_post(v, psi, score, params):
    lib = _lib.load()
    with torch.cuda.device(v.device):
        _lib.check(lib.jamun_baoab_post(_ptr(v), _ptr(psi), _ptr(score), v.shape[0], C.byref(params), _stream()))


def aboba_a(y, v, params):
    lib = _lib.load()
    with torch.cuda.device(y.device):
        _lib.check(lib.jamun_aboba_a(_ptr(y), _ptr(v), y.shape[0], C.byref(params), _stream()))


def aboba_b(y, v, score, noise, params):
    lib = _lib.load()
    with torch.cuda.device(y.device):
        _lib.check(lib.jamun_aboba_b(_ptr(y), _ptr(v), _ptr(score), _ptr(noise), y.shape[0], C.byref(params), _stream()))
