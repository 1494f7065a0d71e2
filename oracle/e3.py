"""Restatement of the e3nn 0.5.4 arithmetic the reference path calls.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED:
e3nn 0.5.4 (``/root/reference/env/requirements.txt:37``) is a third-party
dependency that is not vendored in ``/root/reference`` and not installed in
this image, so each function restates e3nn's published algorithm and cites the
reference call site that uses it.

Irreps are lists of ``(mul, l)`` with even parity and ``l <= 1`` — all the
default architecture (``src/jamun/hydra_config/model/arch/e3conv.yaml:3-6``)
needs.  Feature layout follows e3nn: irreps concatenated, each block stored
``[mul, 2l+1]`` with ``m`` fastest; the l=1 component order is the input xyz
order.
"""

from __future__ import annotations

import math
from fractions import Fraction
from functools import lru_cache
from math import factorial
from typing import List, Sequence, Tuple

import numpy as np
import torch

Irreps = List[Tuple[int, int]]  # (mul, l)


def parse_irreps(s) -> Irreps:
    """``"120x0e + 32x1e"`` -> ``[(120, 0), (32, 1)]`` (e3nn ``o3.Irreps`` string form)."""
    if not isinstance(s, str):
        return [(int(m), int(l)) for m, l in s]
    out = []
    for part in s.split("+"):
        part = part.strip()
        if not part:
            continue
        if "x" in part:
            mul, ir = part.split("x")
            mul = int(mul)
        else:
            mul, ir = 1, part
        ir = ir.strip()
        if ir[-1] != "e":
            raise NotImplementedError(f"only even-parity irreps are supported by the oracle, got {part!r}")
        l = int(ir[:-1])
        if l > 1:
            raise NotImplementedError(f"only l <= 1 is supported by the oracle, got {part!r}")
        out.append((mul, l))
    return out


def irreps_dim(irreps: Irreps) -> int:
    return sum(m * (2 * l + 1) for m, l in irreps)


def irreps_num(irreps: Irreps) -> int:
    """e3nn ``Irreps.num_irreps`` = sum of multiplicities."""
    return sum(m for m, _ in irreps)


def irreps_slices(irreps: Irreps):
    off = 0
    out = []
    for m, l in irreps:
        d = m * (2 * l + 1)
        out.append(slice(off, off + d))
        off += d
    return out


# ---------------------------------------------------------------------------
# Wigner 3j in e3nn's real basis (e3nn/o3/_wigner.py: _su2_clebsch_gordan_coeff,
# change_basis_real_to_complex, _so3_clebsch_gordan).  Used by
# FullyConnectedTensorProduct at src/jamun/e3tools/nn/_conv.py:76-83.
# ---------------------------------------------------------------------------


def _su2_cg_coeff(j1, m1, j2, m2, j3, m3) -> float:
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max(-j1 + j2 + m3, -j1 + m1, 0))
    vmax = int(min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3))

    def f(n):
        return factorial(round(n))

    C = (
        (2.0 * j3 + 1.0)
        * Fraction(
            f(j3 + j1 - j2) * f(j3 - j1 + j2) * f(j1 + j2 - j3) * f(j3 + m3) * f(j3 - m3),
            f(j1 + j2 + j3 + 1) * f(j1 - m1) * f(j1 + m1) * f(j2 - m2) * f(j2 + m2),
        )
    ) ** 0.5
    S = 0
    for v in range(vmin, vmax + 1):
        S += (-1) ** int(v + j2 + m2) * Fraction(
            f(j2 + j3 + m1 - v) * f(j1 - m1 + v),
            f(v) * f(j3 - j1 + j2 - v) * f(j3 + m3 - v) * f(v + j1 - j2 - m3),
        )
    return float(C * S)


def _su2_cg(j1: int, j2: int, j3: int) -> np.ndarray:
    mat = np.zeros((2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1), dtype=np.float64)
    if abs(j1 - j2) <= j3 <= j1 + j2:
        for m1 in range(-j1, j1 + 1):
            for m2 in range(-j2, j2 + 1):
                if abs(m1 + m2) <= j3:
                    mat[j1 + m1, j2 + m2, j3 + m1 + m2] = _su2_cg_coeff(j1, m1, j2, m2, j3, m1 + m2)
    return mat


def _real_to_complex(l: int) -> np.ndarray:
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / 2**0.5
        q[l + m, l - abs(m)] = -1j / 2**0.5
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / 2**0.5
        q[l + m, l - abs(m)] = 1j * (-1) ** m / 2**0.5
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def _wigner_3j_np(l1: int, l2: int, l3: int) -> np.ndarray:
    Q1, Q2, Q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    C = _su2_cg(l1, l2, l3).astype(np.complex128)
    C = np.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, np.conj(Q3.T), C)
    assert np.all(np.abs(C.imag) < 1e-5)
    C = C.real
    return C / np.linalg.norm(C)


def wigner_3j(l1: int, l2: int, l3: int, dtype=torch.float32) -> torch.Tensor:
    """Real-basis Wigner 3j, unit Frobenius norm (e3nn ``o3.wigner_3j``)."""
    return torch.tensor(_wigner_3j_np(l1, l2, l3), dtype=dtype)


# ---------------------------------------------------------------------------
# Spherical harmonics / radial basis  (call sites: src/jamun/model/arch/e3conv.py:41,116,119-126)
# ---------------------------------------------------------------------------


def spherical_harmonics_01(vec: torch.Tensor) -> torch.Tensor:
    """``o3.SphericalHarmonics("1x0e+1x1e", normalize=True, normalization="component")``.

    ``Y0 = 1``; ``Y1 = sqrt(3) * v / max(|v|, 1e-12)`` (``F.normalize`` eps) in xyz order.
    """
    n = torch.nn.functional.normalize(vec, dim=-1)  # eps = 1e-12, as e3nn
    return torch.cat([torch.ones_like(vec[..., :1]), math.sqrt(3.0) * n], dim=-1)


def soft_one_hot_linspace_gaussian(x: torch.Tensor, start, end, number: int) -> torch.Tensor:
    """``e3nn.math.soft_one_hot_linspace(x, start, end, number, basis="gaussian", cutoff=True)``."""
    values = torch.linspace(float(start), float(end), number + 2, dtype=x.dtype)
    step = values[1] - values[0]
    values = values[1:-1]
    diff = (x[..., None] - values) / step
    return diff.pow(2).neg().exp().div(1.12)


# ---------------------------------------------------------------------------
# o3.Linear  (call sites: src/jamun/e3tools/nn/_interaction.py:23-24, _mlp.py:69,109)
# ---------------------------------------------------------------------------


def linear_instructions(irreps_in: Irreps, irreps_out: Irreps):
    """(i_in, i_out, weight offset, alpha) in e3nn's order: ``for i_in for i_out if ir_in == ir_out``.

    ``path_normalization="element"``: ``alpha = 1 / sum(mul_in over instructions sharing i_out)``.
    """
    ins = [
        (i_in, i_out)
        for i_in, (_, l_in) in enumerate(irreps_in)
        for i_out, (_, l_out) in enumerate(irreps_out)
        if l_in == l_out
    ]
    out = []
    off = 0
    for i_in, i_out in ins:
        fan_in = sum(irreps_in[a][0] for a, b in ins if b == i_out)
        n = irreps_in[i_in][0] * irreps_out[i_out][0]
        out.append((i_in, i_out, off, 1.0 / fan_in))
        off += n
    return out, off


def linear_weight_numel(irreps_in: Irreps, irreps_out: Irreps) -> int:
    return linear_instructions(irreps_in, irreps_out)[1]


def linear(x: torch.Tensor, weight: torch.Tensor, irreps_in: Irreps, irreps_out: Irreps) -> torch.Tensor:
    """e3nn ``o3.Linear`` forward, no bias, shared flat ``weight``."""
    ins, numel = linear_instructions(irreps_in, irreps_out)
    assert weight.numel() == numel, (weight.numel(), numel)
    sl_in, sl_out = irreps_slices(irreps_in), irreps_slices(irreps_out)
    out = x.new_zeros(x.shape[0], irreps_dim(irreps_out))
    for i_in, i_out, off, alpha in ins:
        mul_in, l = irreps_in[i_in]
        mul_out, _ = irreps_out[i_out]
        d = 2 * l + 1
        w = weight[off : off + mul_in * mul_out].reshape(mul_in, mul_out)
        xi = x[:, sl_in[i_in]].reshape(-1, mul_in, d)
        yo = torch.einsum("uw,zui->zwi", w, xi) * math.sqrt(alpha)
        out[:, sl_out[i_out]] += yo.reshape(-1, mul_out * d)
    return out


# ---------------------------------------------------------------------------
# FullyConnectedTensorProduct(shared_weights=False, internal_weights=False)
# (call site: src/jamun/e3tools/nn/_conv.py:76-83,94)
# ---------------------------------------------------------------------------


def fctp_instructions(irreps_in1: Irreps, irreps_in2: Irreps, irreps_out: Irreps):
    """(i1, i2, i_out, weight offset, coefficient); e3nn order ``for i1 for i2 for i_out``; mode "uvw".

    ``irrep_normalization="component"``, ``path_normalization="element"``:
    ``alpha = dim(ir_out) / sum_{instr -> same i_out} mul1*mul2``; coefficient ``sqrt(alpha)``.
    """
    ins = []
    for i1, (_, l1) in enumerate(irreps_in1):
        for i2, (_, l2) in enumerate(irreps_in2):
            for io, (_, lo) in enumerate(irreps_out):
                if abs(l1 - l2) <= lo <= l1 + l2:
                    ins.append((i1, i2, io))
    out = []
    off = 0
    for i1, i2, io in ins:
        x = sum(irreps_in1[a][0] * irreps_in2[b][0] for a, b, c in ins if c == io)
        alpha = (2 * irreps_out[io][1] + 1) / x
        n = irreps_in1[i1][0] * irreps_in2[i2][0] * irreps_out[io][0]
        out.append((i1, i2, io, off, math.sqrt(alpha)))
        off += n
    return out, off


def fctp_weight_numel(irreps_in1: Irreps, irreps_in2: Irreps, irreps_out: Irreps) -> int:
    return fctp_instructions(irreps_in1, irreps_in2, irreps_out)[1]


def fctp(
    x1: torch.Tensor,
    x2: torch.Tensor,
    weight: torch.Tensor,
    irreps_in1: Irreps,
    irreps_in2: Irreps,
    irreps_out: Irreps,
) -> torch.Tensor:
    """Per-sample-weight tensor product: ``weight`` is ``[batch, weight_numel]`` (materialised, as the reference)."""
    ins, numel = fctp_instructions(irreps_in1, irreps_in2, irreps_out)
    assert weight.shape[-1] == numel, (weight.shape, numel)
    s1, s2, so = irreps_slices(irreps_in1), irreps_slices(irreps_in2), irreps_slices(irreps_out)
    Z = x1.shape[0]
    out = x1.new_zeros(Z, irreps_dim(irreps_out))
    for i1, i2, io, off, coef in ins:
        m1, l1 = irreps_in1[i1]
        m2, l2 = irreps_in2[i2]
        mo, lo = irreps_out[io]
        w = weight[:, off : off + m1 * m2 * mo].reshape(Z, m1, m2, mo)
        a = x1[:, s1[i1]].reshape(Z, m1, 2 * l1 + 1)
        b = x2[:, s2[i2]].reshape(Z, m2, 2 * l2 + 1)
        C = wigner_3j(l1, l2, lo, dtype=x1.dtype)
        r = torch.einsum("zuvw,ijk,zui,zvj->zwk", w, C, a, b) * coef
        out[:, so[io]] += r.reshape(Z, mo * (2 * lo + 1))
    return out


# ---------------------------------------------------------------------------
# SeparableTensorProduct = depth-wise o3.TensorProduct("uvu") + point-wise o3.Linear
# (src/jamun/e3tools/nn/_tensor_product.py:8-58; used by SeparableConv, _conv.py:122-135)
# ---------------------------------------------------------------------------


def separable_instructions(irreps_in1: Irreps, irreps_in2: Irreps, irreps_out: Irreps):
    """``_tensor_product.py:27-37``: for i (in1) for j (in2) for ir_out in ir_in1 * ir_in2 (l = |l1-l2| .. l1+l2), kept when
    ``ir_out`` occurs in ``irreps_out`` or is 0e; every kept triple gets its OWN output block ``(mul_in1, ir_out)`` and the
    instruction ``(i, j, k, "uvu", has_weight=True)``.  Returns ([(i, j, l_out, weight offset, path coefficient)], irreps_out_dtp,
    weight_numel).  e3nn normalisation (component / element): ``alpha = dim(ir_out) / (num_elements = mul_in2)`` per output block
    — each block has exactly one instruction — so the coefficient is ``sqrt((2 l_out + 1) / mul_in2)``."""
    ls_out = {l for _, l in irreps_out}
    ins, irreps_dtp, off = [], [], 0
    for i, (mul, l1) in enumerate(irreps_in1):
        for j, (mul2, l2) in enumerate(irreps_in2):
            for lo in range(abs(l1 - l2), l1 + l2 + 1):
                if lo in ls_out or lo == 0:
                    ins.append((i, j, lo, off, math.sqrt((2 * lo + 1) / mul2)))
                    irreps_dtp.append((mul, lo))
                    off += mul * mul2  # "uvu" weights: [mul_in1, mul_in2]
    return ins, irreps_dtp, off


def separable_weight_numel(irreps_in1: Irreps, irreps_in2: Irreps, irreps_out: Irreps) -> int:
    return separable_instructions(irreps_in1, irreps_in2, irreps_out)[2]


def separable_tp(x1, x2, weight, lin_weight, irreps_in1: Irreps, irreps_in2: Irreps, irreps_out: Irreps) -> torch.Tensor:
    """``SeparableTensorProduct.forward``: ``lin(dtp(x, y, weight))`` with per-sample ``weight [batch, weight_numel]`` for the
    depth-wise product and the shared flat ``lin_weight`` of the point-wise ``o3.Linear(irreps_out_dtp -> irreps_out)``."""
    ins, irreps_dtp, numel = separable_instructions(irreps_in1, irreps_in2, irreps_out)
    assert weight.shape[-1] == numel, (weight.shape, numel)
    s1, s2, sd = irreps_slices(irreps_in1), irreps_slices(irreps_in2), irreps_slices(irreps_dtp)
    Z = x1.shape[0]
    mid = x1.new_zeros(Z, irreps_dim(irreps_dtp))
    for k, (i, j, lo, off, coef) in enumerate(ins):
        m1, l1 = irreps_in1[i]
        m2, l2 = irreps_in2[j]
        w = weight[:, off : off + m1 * m2].reshape(Z, m1, m2)
        a = x1[:, s1[i]].reshape(Z, m1, 2 * l1 + 1)
        b = x2[:, s2[j]].reshape(Z, m2, 2 * l2 + 1)
        C = wigner_3j(l1, l2, lo, dtype=x1.dtype)
        r = torch.einsum("zuv,ijk,zui,zvj->zuk", w, C, a, b) * coef
        mid[:, sd[k]] = r.reshape(Z, m1 * (2 * lo + 1))
    return linear(mid, lin_weight, irreps_dtp, irreps_out)


# ---------------------------------------------------------------------------
# nn.Gate with normalize2mom activations (call site: src/jamun/e3tools/nn/_gate.py:53-64)
# ---------------------------------------------------------------------------


@lru_cache(maxsize=None)
def normalize2mom_const(name: str) -> float:
    """``e3nn.math.normalize2mom``: ``E_{z~N(0,1)}[f(z)^2]^(-1/2)`` from 1e6 float64 samples, generator seed 0."""
    f = {
        "leaky_relu": lambda z: torch.nn.functional.leaky_relu(z, 0.01),
        "sigmoid": torch.sigmoid,
        "tanh": torch.tanh,
        "silu": torch.nn.functional.silu,
    }[name]
    gen = torch.Generator(device="cpu").manual_seed(0)
    z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
    cst = f(z).pow(2).mean().pow(-0.5).item()
    if abs(cst - 1) < 1e-4:
        return 1.0
    return cst


def gate(x: torch.Tensor, mul_scalars: int, mul_gated: int) -> torch.Tensor:
    """``e3tools Gate(irreps_out = mul_scalars x0e + mul_gated x1e)`` applied to ``(mul_scalars+mul_gated)x0e + mul_gated x1e``.

    scalars -> LeakyReLU(0.01) * c_L; gates -> sigmoid * c_S; gated vectors * gate
    (``ElementwiseTensorProduct`` nets to coefficient 1).
    """
    cL = normalize2mom_const("leaky_relu")
    cS = normalize2mom_const("sigmoid")
    Z = x.shape[0]
    scalars = x[:, :mul_scalars]
    gates = x[:, mul_scalars : mul_scalars + mul_gated]
    gated = x[:, mul_scalars + mul_gated :].reshape(Z, mul_gated, 3)
    scalars = torch.nn.functional.leaky_relu(scalars, 0.01) * cL
    if mul_gated == 0:
        return scalars
    gates = torch.sigmoid(gates) * cS
    gated = gated * gates[:, :, None]
    return torch.cat([scalars, gated.reshape(Z, mul_gated * 3)], dim=-1)


def elementwise_scale(x: torch.Tensor, scales: torch.Tensor, irreps: Irreps) -> torch.Tensor:
    """``o3.ElementwiseTensorProduct(irreps, "Kx0e")`` with K = num_irreps (coefficient 1).

    Call sites: ``src/jamun/model/noise_conditioning.py:46-48,65-67``.
    """
    reps = torch.tensor([2 * l + 1 for m, l in irreps for _ in range(m)])
    s = torch.repeat_interleave(scales.reshape(-1), reps)
    return x * s
