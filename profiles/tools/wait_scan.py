"""Where a kernel's vector-memory requests are waited for — the serialised round trips the COMPILER builds (round 6).

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o k.s jamun_amd/csrc/jamun_node.hip
    python profiles/tools/wait_scan.py k.s                 # one line per kernel: waits, `load .. s_waitcnt vmcnt(0)` within 14 instructions
    python profiles/tools/wait_scan.py k.s <kernel symbol>  # that kernel's requests, waits, branches and barriers in program order

What it found in this library (profiles/EXPERIMENTS.md, round 6): a load under a per-lane predicate (`in ? a.esrc[slot] : 0`,
`if (row < rows) v = x[row]`) becomes a branch of its own and the waits land inside the branches; a select on a loaded value
(`ok ? f(v) : 0`) sinks the load into a branch; a prefetch whose only consumer sits behind one side of a later branch is sunk to that
side; a register ring indexed by a run-time loop is rotated through copies behind `s_waitcnt vmcnt(0)`; operands requested early and first
used inside a predicated store block are waited for with vmcnt(0) in EVERY such block (stores count in vmcnt).  The cures are in the
kernels' comments: unconditional requests at clamped addresses masked by a product, compile-time ring slots, `sched_barrier` behind a
block of requests, consuming early operands once in unconditional code.
"""
import re
import sys

W = 14


def kernels(path):
    lines = open(path).read().split("\n")
    out, name, body = {}, None, []
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, body = m.group(1), []
            out[name] = body
            continue
        if name and l.strip().startswith(".size"):
            name = None
            continue
        if name is not None:
            body.append(l)
    return out


def instrs(body):
    return [x.strip() for x in body if x.strip() and not x.strip().startswith(";") and not x.strip().startswith(".")]


def is_load(x):
    return x.startswith("global_load") or x.startswith("buffer_load")


def summary(path):
    for k, b in kernels(path).items():
        ins = instrs(b)
        n0 = sum(1 for x in ins if x.startswith("s_waitcnt") and "vmcnt(0)" in x)
        tight = 0
        for i, x in enumerate(ins):
            if is_load(x) and any(y.startswith("s_waitcnt") and "vmcnt(0)" in y for y in ins[i + 1 : i + W]):
                tight += 1
        nl = sum(1 for x in ins if is_load(x))
        ns = sum(1 for x in ins if x.startswith("global_store") or x.startswith("buffer_store"))
        print(f"{k[:72]:72s} vmcnt(0) {n0:3d}  load..wait0 {tight:3d}  loads {nl:3d}  stores {ns:3d}  instructions {len(ins)}")


def sequence(path, kname):
    body = kernels(path)[kname]
    m = v = d = 0

    def flush():
        nonlocal m, v, d
        if m or v or d:
            print(f"   [{m} mfma, {v} valu, {d} ds]")
        m = v = d = 0

    for l in body:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith("v_mfma"):
            m += 1
        elif t.startswith("ds_"):
            d += 1
        elif t.startswith("v_"):
            v += 1
        elif t.startswith(("global_load", "buffer_load", "global_store", "s_barrier", ".LBB", "s_cbranch", "s_branch")) or (t.startswith("s_waitcnt") and "vmcnt" in t):
            flush()
            print(t[:78])
    flush()


if __name__ == "__main__":
    if len(sys.argv) == 2:
        summary(sys.argv[1])
    else:
        sequence(sys.argv[1], sys.argv[2])
