"""Build libjamun_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python jamun_amd/csrc/build.py [--force]

The shared library is written in-tree (jamun_amd/libjamun_hip.so) so that it travels to the GPU box.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libjamun_hip.so")
SOURCES = ["jamun_kernels.hip", "jamun_conv.hip", "jamun_conv_fused.hip", "jamun_conv_init.hip", "jamun_conv_initv.hip", "jamun_conv_dg.hip", "jamun_api.cpp"]
DEPS = SOURCES + ["jamun_internal.h", os.path.join(ROOT, "include", "jamun_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "-ffp-contract=off", "-fno-slp-vectorize",
         "-Wno-unused-result", "-Wno-unused-value"]


def _digest() -> str:
    h = hashlib.sha256()
    for f in DEPS:
        with open(f if os.path.isabs(f) else os.path.join(HERE, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    stamp = OUT + ".sha256"
    dig = _digest()
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return OUT
    cmd = [HIPCC] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
