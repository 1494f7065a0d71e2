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
    assert lib.jamun_version() == _lib.ABI_VERSION == 6  # the binding refuses any other version (struct layouts)


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


def _device_isa(src: str) -> str:
    """gfx950 ISA text of one kernel source with the library's own flags (hipcc -S --cuda-device-only), cached under csrc/build/ by the same
    digest as the objects (source + shared headers + flags)."""
    import subprocess

    from jamun_amd.csrc import build as b

    os.makedirs(b.OBJ_DIR, exist_ok=True)
    out = os.path.join(b.OBJ_DIR, f"{os.path.splitext(src)[0]}.{b._obj_digest(src)}.s")
    if not os.path.exists(out):
        for old in os.listdir(b.OBJ_DIR):
            if old.startswith(os.path.splitext(src)[0] + ".") and old.endswith(".s"):
                os.unlink(os.path.join(b.OBJ_DIR, old))
        tmp = out + f".tmp{os.getpid()}"
        subprocess.run([b.HIPCC] + b.CFLAGS + b.EXTRA + ["-S", "--cuda-device-only", os.path.join(b.HERE, src), "-o", tmp], check=True, capture_output=True)
        os.replace(tmp, out)
    return out


def test_no_vector_instruction_sits_inside_an_mfma_hazard_window_of_the_shipped_build():
    """The tripwire behind jamun_split.h.  On gfx950 the dependencies between a v_mfma and the vector instructions around it are software
    managed (wait states); hipcc inserts them for instructions it selects and NOT for the body of an inline-asm statement.  Round 5 shipped
    asm `v_cvt_pk_f16_f32` results read by an MFMA one wait state later (two are needed): right by luck in three kernels, irreproducible in a
    fourth.  profiles/tools/mfma_hazard_scan.py re-derives the rules (RAW / WAW / WAR behind an MFMA, VALU -> MFMA) and walks the ISA of
    every kernel: no instruction, compiler-selected or asm, may violate them — and the split primitives must not be asm again."""
    import importlib.util
    from concurrent.futures import ThreadPoolExecutor

    from jamun_amd.csrc import build as b

    spec = importlib.util.spec_from_file_location("mfma_hazard_scan", os.path.join(ROOT, "profiles", "tools", "mfma_hazard_scan.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    assert "JAMUN_SPLIT_ASM_REPRO" not in " ".join(b.CFLAGS + b.EXTRA)
    hips = [s for s in b.SOURCES if s.endswith(".hip")]
    with ThreadPoolExecutor(max_workers=min(len(hips), os.cpu_count() or 4)) as ex:
        isas = list(ex.map(_device_isa, hips))
    n_kernels = n_mfma = 0
    for src, isa in zip(hips, isas):
        for name, (n_ins, n_m, n_asm, violations) in scan.scan(isa, want_all=True).items():
            n_kernels += 1
            n_mfma += n_m
            assert not violations, (src, name, [(k, ws, need, p.text, c.text) for k, ws, need, p, c, _ in violations[:4]])
    assert n_kernels >= 60 and n_mfma >= 3000  # (the scan saw the library: ~80 kernels, ~4800 MFMA instructions)
    for f in os.listdir(b.HERE):  # one split implementation: no kernel source carries its own asm v_cvt_pk / v_fma_mix
        if f.endswith((".hip", ".h")) and f != "jamun_split.h":
            txt = open(os.path.join(b.HERE, f)).read()
            assert not re.search(r'asm[^;]*"v_(cvt_pk_f16_f32|fma_mix_f32)', txt), f


def test_hazard_scanner_flags_the_round5_pattern_and_accepts_the_compiler_forms(tmp_path):
    """The tripwire itself: on hand-written ISA the scanner must flag (a) an inline-asm v_cvt_pk_f16_f32 one wait state in front of the MFMA
    that reads it (the round-5 bug: `s_nop 0` is all hipcc puts behind an asm definition), (b) a vector read of an MFMA's destination inside
    its latency, (c) a vector write to an in-flight MFMA's SrcC — and accept the same sequences at the distances hipcc keeps for visible
    instructions, across a branch edge too."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("mfma_hazard_scan", os.path.join(ROOT, "profiles", "tools", "mfma_hazard_scan.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)

    def run(body):
        f = tmp_path / "k.s"
        f.write_text("_Z1kv:\n" + body + "\n\ts_endpgm\n.Lfunc_end0:\n")
        (name, (n, n_mfma, n_asm, viol)), = scan.scan(str(f), want_all=True).items()
        return [(k, ws, need) for k, ws, need, p, c, tag in viol]

    mfma = "\tv_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]\n"
    asm_cvt = "\t;;#ASMSTART\n\tv_cvt_pk_f16_f32 v20, v30, v31\n\t;;#ASMEND\n"
    assert run(asm_cvt + "\ts_nop 0\n" + mfma) == [("V2M", 1, 2)]                      # the bug as shipped
    assert run(asm_cvt + "\ts_nop 1\n" + mfma) == []                                  # two wait states: fine
    assert run("\tv_cvt_pk_f16_f32 v20, v30, v31\n\ts_nop 0\n" + mfma) == [("V2M", 1, 2)]  # a visible instruction that close would be a compiler bug
    assert run("\tv_cvt_pk_f16_f32 v20, v30, v31\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 0\n" + mfma) == []
    assert run(mfma + "\ts_nop 7\n\tv_mul_f32_e32 v40, v0, v41\n") == [("RAW", 8, 12)]  # 8-pass MFMA result read after 8 wait states (12 needed)
    assert run(mfma + "\ts_nop 11\n\tv_mul_f32_e32 v40, v0, v41\n") == []
    assert ("WAR", 3, 7) in run(mfma + "\ts_nop 2\n\tv_mov_b32_e32 v5, 0\n")          # write to SrcC (= vDst) of an MFMA in flight
    assert run(mfma + "\tv_mov_b32_e32 v16, 0\n") == []                               # SrcA / SrcB: safe against write-after-read (microbenchmark)
    # through a back edge: the MFMA at the bottom of a loop, its consumer at the top
    loop = ".LBB0_1:\n\tv_mul_f32_e32 v40, v0, v41\n\ts_nop 3\n" + mfma + "\ts_cbranch_scc1 .LBB0_1\n"
    assert ("RAW", 1, 12) in run(loop)
    # the 4-pass and the fp32-input (non-XDL) opcodes have their own distances
    assert run("\tv_mfma_f32_16x16x32_f16 v[0:3], v[16:19], v[20:23], v[0:3]\n\ts_nop 6\n\tv_mul_f32_e32 v40, v0, v41\n") == [("RAW", 7, 8)]
    assert run("\tv_mfma_f32_32x32x2_f32 v[0:15], v16, v20, v[0:15]\n\ts_nop 15\n\ts_nop 1\n\tv_mul_f32_e32 v40, v0, v41\n") == []


def test_wait_scan_counts_requests_that_are_waited_for_at_once(tmp_path):
    """profiles/tools/wait_scan.py: a request followed by `s_waitcnt vmcnt(0)` within a few instructions is one serialised
    round trip; requests that are issued together and waited for once are not."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("wait_scan", os.path.join(ROOT, "profiles", "tools", "wait_scan.py"))
    ws = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ws)
    text = "\n".join([
        "_Z4k_ab:",
        "\tglobal_load_dword v0, v[2:3], off",
        "\ts_waitcnt vmcnt(0)",
        "\tv_add_f32 v1, v0, v0",
        "\tglobal_load_dword v4, v[2:3], off",
        "\ts_waitcnt vmcnt(0)",
        "\t.size\t_Z4k_ab, .Lfunc_end0-_Z4k_ab",
        "_Z4k_ok:",
        "\tglobal_load_dword v0, v[2:3], off",
        "\tglobal_load_dword v4, v[2:3], off",
    ] + ["\tv_mov_b32 v9, v9"] * 20 + [
        "\ts_waitcnt vmcnt(0)",
        "\t.size\t_Z4k_ok, .Lfunc_end1-_Z4k_ok",
    ])
    f = tmp_path / "k.s"
    f.write_text(text)
    ks = ws.kernels(str(f))
    assert set(ks) == {"_Z4k_ab", "_Z4k_ok"}

    def tight(name):
        ins = ws.instrs(ks[name])
        return sum(1 for i, x in enumerate(ins) if ws.is_load(x) and any(y.startswith("s_waitcnt") and "vmcnt(0)" in y for y in ins[i + 1 : i + ws.W]))

    assert tight("_Z4k_ab") == 2 and tight("_Z4k_ok") == 0
