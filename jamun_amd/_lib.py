"""ctypes binding of ``libjamun_hip.so`` (C ABI declared in ``include/jamun_hip.h``).

The product path has no CPU fallback: if the shared library is missing or does not load, every
operator raises ``RuntimeError`` (build it with ``python jamun_amd/csrc/build.py``).
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjamun_hip.so")


class jamun_tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


class jamun_hparams(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32),
        ("mul0", C.c_int32),
        ("mul1", C.c_int32),
        ("edge_attr_dim", C.c_int32),
        ("emb_dim", C.c_int32 * 4),
        ("emb_rows", C.c_int32 * 4),
        ("use_residue_sequence_index", C.c_int32),
        ("mean_center", C.c_int32),
        ("max_radius", C.c_float),
        ("average_squared_distance", C.c_float),
        ("act_scalar_const", C.c_float),
        ("act_gate_const", C.c_float),
        ("w3j_111_sign", C.c_float),
        ("separable", C.c_int32),
    ]


class jamun_topology(C.Structure):
    _fields_ = [
        ("n_atoms", C.c_int32),
        ("n_graphs", C.c_int32),
        ("ptr", C.POINTER(C.c_int32)),
        ("atom_type_index", C.POINTER(C.c_int32)),
        ("atom_code_index", C.POINTER(C.c_int32)),
        ("residue_code_index", C.POINTER(C.c_int32)),
        ("residue_sequence_index", C.POINTER(C.c_int32)),
        ("n_bonds", C.c_int32),
        ("bond_src", C.POINTER(C.c_int64)),
        ("bond_dst", C.POINTER(C.c_int64)),
    ]


class jamun_tuning(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("no_dg", "no_mf", "dg_fp32", "dg_no_alt", "dg_no_sp", "dg_no_sph", "no_mfi", "no_init_v", "node_fp32",
                                         "edge_h_fp32", "dg_kgroups", "no_tail", "no_short_k", "no_ml", "seg_cost_tenths", "f16x1", "no_fuse_geom", "selfcheck", "no_tprod_t")]


class jamun_mcmc_params(C.Structure):
    _fields_ = [
        ("steps", C.c_int32),
        ("save_every_n_steps", C.c_int32),
        ("burn_in_steps", C.c_int32),
        ("has_clip", C.c_int32),
        ("delta", C.c_float),
        ("friction", C.c_float),
        ("M", C.c_float),
        ("inverse_temperature", C.c_float),
        ("score_fn_clip", C.c_float),
    ]


class jamun_stats(C.Structure):
    _fields_ = [
        ("n_edges", C.c_int64),
        ("flop_ref_assoc", C.c_int64),
        ("flop_executed", C.c_int64),
        ("conv_k0", C.c_int64),
        ("conv_k1", C.c_int64),
        ("conv0_flop_alg", C.c_int64),
        ("conv1_flop_alg", C.c_int64),
        ("edge_stride", C.c_int32),
        ("n_slices", C.c_int32),
        ("conv_path", C.c_int32),
        ("dg_mode", C.c_int32),
        ("init_path", C.c_int32),
        ("dg_row_blocks", C.c_int32),
        ("dg_emu", C.c_int32),
        ("conv_flop_exec_launch", C.c_int64),
        ("conv_flop_useful_launch", C.c_int64),
        ("n_tail_tiles", C.c_int32),
        ("n_tail", C.c_int32),
        ("conv_bytes_alg_launch", C.c_int64),
        ("mf_nks", C.c_int32),
        ("ml_window", C.c_int32),
    ]


# every symbol include/jamun_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "jamun_last_error": (C.c_char_p, []),
    "jamun_version": (C.c_int, []),
    "jamun_model_create": (C.c_int, [C.POINTER(jamun_hparams), C.POINTER(jamun_tensor), C.c_int32, C.POINTER(_P)]),
    "jamun_model_destroy": (None, [_P]),
    "jamun_sampler_create": (C.c_int, [_P, C.c_float, C.POINTER(jamun_topology), C.POINTER(jamun_tuning), C.POINTER(_P)]),
    "jamun_sampler_destroy": (None, [_P]),
    "jamun_xhat": (C.c_int, [_P, _P, _P, _P]),
    "jamun_score": (C.c_int, [_P, _P, _P, _P]),
    "jamun_walk_baoab": (C.c_int, [_P, _P, _P, C.POINTER(jamun_mcmc_params), _P, C.c_uint64, _P, _P, _P, _P, _P]),
    "jamun_walk_aboba": (C.c_int, [_P, _P, _P, C.POINTER(jamun_mcmc_params), _P, C.c_uint64, _P, _P, _P, _P, _P]),
    "jamun_num_frames": (C.c_int, [C.POINTER(jamun_mcmc_params), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "jamun_mean_center": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "jamun_radius_graph": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P, _P, _P]),
    "jamun_scatter_mean": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "jamun_baoab_pre": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(jamun_mcmc_params), _P]),
    "jamun_baoab_post": (C.c_int, [_P, _P, _P, C.c_int32, C.POINTER(jamun_mcmc_params), _P]),
    "jamun_aboba_a": (C.c_int, [_P, _P, C.c_int32, C.POINTER(jamun_mcmc_params), _P]),
    "jamun_aboba_b": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(jamun_mcmc_params), _P]),
    "jamun_edge_geometry": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, C.c_float, C.c_int32, _P, _P, _P]),
    "jamun_node_linear": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64, _P, _P]),
    "jamun_philox_normal": (C.c_int, [_P, C.c_int32, C.c_uint64, C.c_uint32, C.c_uint32, _P]),
    "jamun_build_edges": (C.c_int, [_P, _P, _P]),
    "jamun_conv_block": (C.c_int, [_P, C.c_int32, _P, _P, _P]),
    "jamun_sampler_stats": (C.c_int, [_P, C.POINTER(jamun_stats), _P]),
    "jamun_sampler_check": (C.c_int, [_P, _P]),
    "jamun_profile_enable": (C.c_int, [_P, C.c_int32]),
    "jamun_profile_sample": (C.c_int, [_P, C.c_int32]),
    "jamun_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), _P]),
    "jamun_debug_stamps": (C.c_int, [C.POINTER(C.c_uint64)]),
    "jamun_debug_read": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
}

PROF_CLASSES = ["geom", "edge_h", "conv0_init", "conv1_init", "conv0", "conv1", "node_update", "head_finalize", "tprod"]

ABI_VERSION = 6  # jamun_version() of the library this binding was written for (struct layouts and signatures above)

_lib: Optional[C.CDLL] = None


def _check_fresh() -> None:
    """Refuse to run a library that was built from other sources than the ones in the tree: ``csrc/build.py`` stamps the
    binary with a digest of its sources and flags; after an edit the stamp no longer matches.  With ``hipcc`` at hand the
    library is rebuilt (seconds); without it — or with ``JAMUN_NO_REBUILD=1`` — a stale binary is an error, never silently
    used."""
    from .csrc import build as b

    stamp = LIB_PATH + ".sha256"
    if not os.path.exists(LIB_PATH) or not os.path.exists(stamp):
        return  # missing library: reported by load(); a library without a stamp was built by hand (explicit hipcc command)
    if open(stamp).read().strip() == b._digest():
        return
    if os.environ.get("JAMUN_NO_REBUILD") or not os.path.exists(b.HIPCC):
        raise RuntimeError(f"{LIB_PATH} is stale: its sources changed since it was built. Run `python jamun_amd/csrc/build.py`.")
    b.build(verbose=False)


def load() -> C.CDLL:
    """Load the HIP library or raise.  There is deliberately no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    _check_fresh()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python jamun_amd/csrc/build.py` "
            "(hipcc --offload-arch=gfx950). jamun_amd has no CPU fallback."
        )
    # PyTorch first: its wheel carries its own HIP runtime, and the library's HIP calls must land in THAT copy (the one that owns the
    # tensors' memory and streams).  Loaded the other way round — the library before torch, as `build()` followed by `smoke()` in one
    # process did — the library binds /opt/rocm's runtime and its first call fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401

    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"could not load {LIB_PATH}: {e}") from e
    # (the version first: an older library may lack a symbol declared below, and "rebuild" is the helpful message, not AttributeError)
    lib.jamun_version.restype = C.c_int
    lib.jamun_version.argtypes = []
    ver = int(lib.jamun_version())
    if ver != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports ABI version {ver}, this binding is written for version {ABI_VERSION} "
                           "(struct layouts differ): rebuild with `python jamun_amd/csrc/build.py --force`")
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        msg = load().jamun_last_error()
        raise RuntimeError(f"jamun_hip error {code}: {msg.decode() if msg else '?'}")
