"""Build libjamun_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python jamun_amd/csrc/build.py [--force]

The shared library is written in-tree (jamun_amd/libjamun_hip.so) so that it travels to the GPU box.  Sources are compiled
to objects in parallel (``csrc/build/``, keyed by a digest of the source, the shared headers and the flags, so an edit of one
kernel file recompiles that file only) and linked; the library and its stamp are replaced atomically, under an exclusive
file lock, so that several ranks importing the package at once never see a half-written binary or compile twice.
"""
import fcntl
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libjamun_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
SOURCES = ["jamun_kernels.hip", "jamun_conv.hip", "jamun_conv_initv.hip", "jamun_conv_dg.hip", "jamun_conv_mf.hip", "jamun_conv_ml.hip", "jamun_node.hip", "jamun_sepconv.hip",
           "jamun_api.cpp"]
HEADERS = ["jamun_internal.h", "jamun_mf_dev.h", "jamun_split.h", os.path.join(ROOT, "include", "jamun_hip.h")]
DEPS = SOURCES + HEADERS
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-ffp-contract=off", "-fno-slp-vectorize",
          "-Wno-unused-result", "-Wno-unused-value"]
EXTRA = os.environ.get("JAMUN_EXTRA_CFLAGS", "").split()  # diagnostic builds, e.g. -DJAMUN_STAMP
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]
FLAGS = CFLAGS + LDFLAGS  # (kept for the digest)


def _read(f: str) -> bytes:
    with open(f if os.path.isabs(f) else os.path.join(HERE, f), "rb") as fh:
        return fh.read()


def _digest() -> str:
    h = hashlib.sha256()
    for f in DEPS:
        h.update(_read(f))
    h.update(" ".join(FLAGS + EXTRA).encode())
    return h.hexdigest()


def _obj_digest(src: str) -> str:
    h = hashlib.sha256()
    h.update(_read(src))
    for f in HEADERS:
        h.update(_read(f))
    h.update(" ".join(CFLAGS + EXTRA).encode())
    return h.hexdigest()[:24]


def _compile(src: str, verbose: bool) -> str:
    obj = os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{_obj_digest(src)}.o")
    if os.path.exists(obj):
        return obj
    for old in os.listdir(OBJ_DIR):  # objects of older versions of this source
        if old.startswith(os.path.splitext(src)[0] + ".") and old.endswith(".o"):
            os.unlink(os.path.join(OBJ_DIR, old))
    tmp = obj + f".tmp{os.getpid()}"
    cmd = [HIPCC] + CFLAGS + EXTRA + ["-c", os.path.join(HERE, src), "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, obj)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    stamp = OUT + ".sha256"
    dig = _digest()

    def fresh() -> bool:
        return os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read().strip() == dig

    if not force and fresh():
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)  # one builder at a time; the others wait here and find a fresh library
        if not force and fresh():
            return OUT
        if force:
            for old in os.listdir(OBJ_DIR):
                if old.endswith(".o"):
                    os.unlink(os.path.join(OBJ_DIR, old))
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
            objs = list(ex.map(lambda s: _compile(s, verbose), SOURCES))
        tmp = OUT + f".tmp{os.getpid()}"
        cmd = [HIPCC] + LDFLAGS + objs + ["-o", tmp]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)
        with open(stamp + f".tmp{os.getpid()}", "w") as f:
            f.write(dig)
        os.replace(stamp + f".tmp{os.getpid()}", stamp)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
