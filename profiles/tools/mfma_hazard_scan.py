#!/usr/bin/env python3
"""Static scan of gfx950 ISA text (hipcc -S output) for MFMA <-> VALU register hazards the hardware does NOT interlock.

On gfx90a and later the matrix pipe writes ordinary VGPRs and the dependency between an XDL (MFMA) instruction and a later
vector / memory instruction of the SAME wave on the MFMA's destination registers is software managed: the compiler's hazard
recogniser inserts `s_nop`s.  It does so for instructions it can see.  The body of an inline-asm statement is opaque to it
(LLVM `GCNHazardRecognizer::checkInlineAsmHazards` covers only the legacy 12-dword store and dst-sel forwarding cases, and
`getWaitStatesSince` counts an INLINEASM as zero wait states), so a `v_cvt_pk_f16_f32` / `v_fma_mix_f32` written as inline asm
that reads an accumulator too early reads whatever the register held — data dependent AND timing dependent.

Rules checked (wait states = instructions issued by the wave between the two, `s_nop N` = N + 1; P = passes of the MFMA: PASSES below,
8 for v_mfma_f32_32x32x16_f16 — LLVM 'GFX940_XDL_N_Pass...' tables, +1 on gfx950; --passes is the default for opcodes not listed):
  RAW  MFMA writes vDst -> VALU / LDS / VMEM reads it          P + 3 + 1  (12)
  WAW  MFMA writes vDst -> VALU writes it                       P + 1 + 1  (10)
  WAR  MFMA reads SrcC  -> VALU writes it                       {2: 1, 4: 3, 8: 7, 16: 13}[P]
  V2M  VALU writes a VGPR -> MFMA reads it (SrcA / SrcB / SrcC) 2      (hipcc -S of `x = f(..); mfma(x, ..)` shows the compiler keeping two wait
       states there on gfx942 and gfx950; behind an inline-asm definition it keeps ONE — its generic rule for an asm that defines a VGPR — which
       is what made k_conv_ml<8> with the asm `v_cvt_pk_f16_f32` irreproducible: profiles/EXPERIMENTS.md, round 6)
  (MFMA -> MFMA dependencies are all compiler-visible and not checked here.)

Usage:  mfma_hazard_scan.py file.s [--kernel SUBSTR] [--passes 8] [--all]
Reports every violation; with --all also those on compiler-visible instructions (expected: none — that validates the table).
Exit status 1 if an inline-asm instruction violates a rule.
"""
import argparse
import re
import sys

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


NO_DST = ("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store", "s_", "v_cmpx", "v_nop", "global_atomic", "buffer_atomic")
SGPR_DST = ("v_cmp", "v_readfirstlane", "v_readlane")


# passes (4 cycles each) of the MFMA opcodes this library issues, gfx950 (LLVM SISchedule: 32x32x16 f16 8, 16x16x32 f16 4, 32x32x2 f32 16, 16x16x4 f32 8)
PASSES = {"v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_32x32x16_bf16": 8, "v_mfma_f32_16x16x32_f16": 4, "v_mfma_f32_16x16x32_bf16": 4,
          "v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_16x16x4_f32": 8}


class Ins:
    __slots__ = ("line", "text", "op", "dst", "src", "asm", "ws", "mfma", "srcc", "passes")

    def __init__(self, line, text, asm):
        self.line, self.text, self.asm = line, text, asm
        parts = text.split(None, 1)
        self.op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        self.mfma = self.op.startswith("v_mfma") or self.op.startswith("v_smfmac")
        self.passes = PASSES.get(self.op.split("_e64")[0], None)
        self.ws = 1
        if self.op == "s_nop":
            self.ws = int(ops[0], 0) + 1
        self.dst, self.src, self.srcc = set(), set(), set()
        if not ops:
            return
        if self.op.startswith(NO_DST):
            for o in ops:
                self.src |= regs_of(o)
        elif self.op.startswith(SGPR_DST):
            for o in ops[1:]:
                self.src |= regs_of(o)
        else:
            self.dst = regs_of(ops[0])
            for o in ops[1:]:
                self.src |= regs_of(o)
            if self.mfma and len(ops) >= 4:
                self.srcc = regs_of(ops[3])
        # v_fma_mix / v_pk with dst also read? no: plain three-address forms only.


def parse(lines):
    ins, labels, asm = [], {}, False
    for n, raw in enumerate(lines, 1):
        s = raw.split(";;#")[0] if ";;#" in raw and not raw.strip().startswith(";;#") else raw
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            asm = True
            continue
        if t.startswith(";;#ASMEND"):
            asm = False
            continue
        t = t.split(";")[0].strip()
        if not t or t.startswith(".") and not t.endswith(":") or t.startswith("//"):
            continue
        if t.endswith(":"):
            labels[t[:-1]] = len(ins)
            continue
        if t.startswith("."):
            continue
        ins.append(Ins(n, t, asm))
    return ins, labels


def check(ins, labels, passes, want_all):
    """Backward walk over the control-flow graph (fall-through + branch edges) from every candidate consumer."""
    def need(p):  # (RAW, WAW, WAR) wait states behind MFMA p
        n = p.passes or passes
        if p.op.count("f32") == 2:  # fp32-input MFMAs are not XDL ops on gfx940+: LLVM's SMFMA tables (N + 2, N + 2)
            return n + 2, n + 2, {2: 1, 4: 3, 8: 7, 16: 13}[n]
        return n + 4, n + 2, {2: 1, 4: 3, 8: 7, 16: 13}[n]

    horizon = 20
    jump_preds = {}
    for j, c in enumerate(ins):
        if c.op.startswith("s_cbranch") or c.op == "s_branch":
            tgt = labels.get(c.text.split()[-1])
            if tgt is not None:
                jump_preds.setdefault(tgt, []).append(j)
    uncond = ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64")

    def preds(i):
        out = list(jump_preds.get(i, ()))
        if i > 0 and ins[i - 1].op not in uncond:
            out.append(i - 1)
        return out

    out = []
    # V2M: the consumer is the (compiler-visible) MFMA, the producer a vector ALU instruction — inline asm or not
    for j, c in enumerate(ins):
        if not c.mfma:
            continue
        stack, seen = [(i, 0, i != j - 1) for i in preds(j)], set()
        while stack:
            i, ws, jumped = stack.pop()
            if (i, ws) in seen:
                continue
            seen.add((i, ws))
            p = ins[i]
            if p.op.startswith("v_") and not p.mfma and (p.dst & c.src) and (p.asm or want_all) and ws < 2:
                out.append(("V2M", ws, 2, p, c, " (path through a branch edge)" if jumped else ""))
            ws += 0 if p.asm else p.ws
            if ws < 2:
                for q in preds(i):
                    stack.append((q, ws, jumped or q != i - 1))
    for j, c in enumerate(ins):
        if c.mfma or c.op.startswith("s_") or not (c.asm or want_all) or not (c.dst or c.src):
            continue
        stack, seen = [(i, 0, i != j - 1) for i in preds(j)], set()
        while stack:
            i, ws, jumped = stack.pop()
            if (i, ws) in seen:
                continue
            seen.add((i, ws))
            p = ins[i]
            if p.mfma:
                tag = " (path through a branch edge)" if jumped else ""
                raw_ws, waw_ws, war_ws = need(p)
                if ws < raw_ws and (c.src & p.dst):
                    out.append(("RAW", ws, raw_ws, p, c, tag))
                if ws < waw_ws and (c.dst & p.dst):
                    out.append(("WAW", ws, waw_ws, p, c, tag))
                if ws < war_ws and (c.dst & p.srcc):
                    out.append(("WAR", ws, war_ws, p, c, tag))
            ws += 0 if p.asm else p.ws  # LLVM counts an inline-asm statement as zero wait states: so does this scan
            if ws < horizon:
                for q in preds(i):
                    stack.append((q, ws, jumped or q != i - 1))
    return out


def scan(path, kernel=None, passes=8, want_all=False):
    lines = open(path).read().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    res = {}
    for idx, (i0, name) in enumerate(starts):
        if kernel and kernel not in name:
            continue
        i1 = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        ins, labels = parse(lines[i0 + 1:i1])
        for x in ins:
            x.line += i0 + 1
        out = check(ins, labels, passes, want_all)
        res[name] = (len(ins), sum(1 for x in ins if x.mfma), sum(1 for x in ins if x.asm), out)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--kernel")
    ap.add_argument("--passes", type=int, default=8)
    ap.add_argument("--all", action="store_true")
    a = ap.parse_args()
    bad = 0
    for name, (n, nm, na, out) in scan(a.file, a.kernel, a.passes, a.all).items():
        n_asm = sum(1 for v in out if (v[3].asm if v[0] == "V2M" else v[4].asm))
        print(f"{name}: {n} instructions, {nm} MFMA, {na} inline-asm; violations: {n_asm} inline-asm, {len(out) - n_asm} compiler-visible")
        for kind, ws, need, p, c, tag in out:
            print(f"   {kind} {ws}/{need} wait states{tag}: line {p.line}{' [asm]' if p.asm else ''}: {p.text}\n        -> line {c.line}{' [asm]' if c.asm else ''}: {c.text}")
        bad += n_asm
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
