"""CPU tests: the C-ABI library loads and exports every symbol include/jamun_hip.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "jamun_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(jamun_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_match_binding_table():
    from jamun_amd import _lib

    assert _declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_builds_loads_and_exports_all_symbols():
    from jamun_amd.csrc import build as b
    from jamun_amd import _lib

    b.build(verbose=False)
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.jamun_version() == _lib.ABI_VERSION == 5  # the binding refuses any other version (struct layouts)


def test_num_frames_matches_reference_counts():
    # frame counts of the golden fixtures produced by the reference integrators
    import ctypes as C
    import json

    from jamun_amd import _lib, native

    lib = _lib.load()
    for name, ny_exp, ns_exp in [("baoab_default", 50, 50), ("baoab_clip_mass", 8, 9), ("aboba_default", 50, 49), ("aboba_clip_mass", 13, 13)]:
        meta = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
        kw = meta["kwargs"]
        p = native.make_mcmc_params(kw["steps"], kw["delta"], kw["friction"], kw["M"], kw["inverse_temperature"], kw["score_fn_clip"],
                                    kw.get("save_every_n_steps", 1), kw.get("burn_in_steps", 0))
        ny, nb, na = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(lib.jamun_num_frames(C.byref(p), C.byref(ny), C.byref(nb), C.byref(na)))
        assert ny.value == ny_exp
        assert (nb.value if meta["integrator"] == "baoab" else na.value) == ns_exp


def test_error_reporting_without_gpu():
    import ctypes as C

    from jamun_amd import _lib

    lib = _lib.load()
    h = C.c_void_p()
    code = lib.jamun_model_create(None, None, 0, C.byref(h))
    assert code != 0
    assert b"null" in lib.jamun_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(code)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "jamun_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_library_reads_no_environment_and_tuning_is_validated():
    """Kernel selection goes through jamun_tuning only (include/jamun_hip.h): no getenv in the native sources; the Python side rejects
    switches the struct does not have."""
    csrc = os.path.join(ROOT, "jamun_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    from jamun_amd import _lib, native

    t = native.make_tuning({"no_mf": 1, "dg_kgroups": 4})
    assert t.no_mf == 1 and t.dg_kgroups == 4 and t.no_dg == 0
    with pytest.raises(ValueError, match="unknown tuning switch"):
        native.make_tuning({"no_such_kernel": 1})
    hdr = open(os.path.join(ROOT, "include", "jamun_hip.h")).read()
    body = hdr[hdr.index("typedef struct jamun_tuning {") : hdr.index("} jamun_tuning;")]
    fields = re.findall(r"int32_t\s+(\w+)", body)
    assert fields == [n for n, _ in _lib.jamun_tuning._fields_]  # the binding mirrors the header field for field
