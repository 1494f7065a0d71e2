"""Build a variant of the library with extra compiler flags into scratch/libs/<name>.so (own object cache; the in-tree library is untouched).

    python profiles/exp/build_variant.py mf_split_c -DMF_SPLIT_C
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jamun_amd.csrc import build as b  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "scratch", "libs")
os.makedirs(out_dir, exist_ok=True)
b.OUT = os.path.join(out_dir, name + ".so")
b.OBJ_DIR = os.path.join(ROOT, "scratch", "obj_" + name)
b.EXTRA = flags
print(b.build(force=False, verbose=False))
