#!/usr/bin/env python3
"""Static check of the gfx950 ISA hipcc emits for the hand-written-asm kernels: the software-visible hazards hipcc's own
hazard recogniser does NOT cover once an instruction sits inside an `asm` statement (it treats the statement as opaque).

Round 2 found two such bugs dynamically: an asm MFMA read its C operand right behind the VALU instructions
that unpacked the bias (stale register: wrong bytes), and an asm `global_atomic_add` took its address SGPRs straight from a
`v_readlane` (stale address: a GPU memory fault).  Both are caught here from the text of the ISA; `build()` runs this over
k_mfma16x.hip and k_mfma16p.hip and fails on a violation.

Rules (wait states = instructions issued in between, `s_nop N` counts N + 1; numbers from the gfx940-family tables of LLVM's
GCNHazardRecognizer, the same silicon rules as gfx950; P = passes of the MFMA: 4 for 16x16x64 i8, 8 for 32x32x32 i8):
  A  MFMA writes D            -> any non-MFMA instruction reading D, any VALU writing D             needs P + 3
  B  MFMA writes D            -> MFMA reading D as SrcA / SrcB                                     needs P + 3
  C  MFMA writes D            -> MFMA whose SrcC overlaps D but is not exactly D                   needs P + 1
                                 (SrcC == D exactly, the accumulate chain, may issue back to back)
  D  VALU writes a VGPR/AGPR  -> MFMA reading it (SrcA / SrcB / SrcC)                              needs 2
  E  MFMA reads SrcC != D     -> VALU writing SrcC                                                 needs P + 3 (conservative)
  F  ds_read* writes v[..]    -> first use needs an s_waitcnt lgkmcnt(k) in between that covers it (LDS returns in order;
                                 any scalar load in flight makes anything but lgkmcnt(0) insufficient)
  G  VALU writes an SGPR (v_readlane, v_readfirstlane, v_cmp .. s[..]) -> VMEM / SMEM / LDS-DMA using it    needs 5
  H  s_mov / s_add .. m0      -> LDS-DMA (buffer_load .. lds, global_load_lds) or ds_* with M0 use  needs 1
Control flow: every function is checked in textual order, and every backward branch additionally with the loop tail glued to
the loop head.  Calls do not occur in these kernels.

usage: isa_hazards.py file.s [file.s ...]       exit status 1 on a violation
       isa_hazards.py --selftest                the two round-2 bugs as ISA snippets: both must be flagged"""
from __future__ import annotations

import re
import sys
from dataclasses import dataclass, field
from typing import List, Optional, Set, Tuple

WINDOW = 24   # instructions looked back (max requirement: 16-pass MFMA + 3 = 19)

REG = re.compile(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b|\b(vcc|exec|m0|scc)\b")


def regs_of(op: str) -> Set[Tuple[str, int]]:
    out: Set[Tuple[str, int]] = set()
    for m in REG.finditer(op):
        if m.group(1):
            for i in range(int(m.group(2)), int(m.group(3)) + 1):
                out.add((m.group(1), i))
        elif m.group(4):
            out.add((m.group(4), int(m.group(5))))
        else:
            out.add((m.group(6), 0))
    return out


@dataclass
class Ins:
    line: int
    text: str
    mnem: str
    ops: List[str]
    defs: Set[Tuple[str, int]] = field(default_factory=set)
    uses: Set[Tuple[str, int]] = field(default_factory=set)
    in_asm: bool = False

    @property
    def is_mfma(self):
        return self.mnem.startswith(("v_mfma", "v_smfmac"))

    @property
    def is_valu(self):
        return self.mnem.startswith("v_") and not self.is_mfma

    @property
    def is_lds(self):
        return self.mnem.startswith("ds_")

    @property
    def is_vmem(self):
        return self.mnem.startswith(("buffer_", "global_", "flat_", "scratch_", "tbuffer_"))

    @property
    def is_smem(self):
        return self.mnem.startswith(("s_load", "s_buffer_load", "s_store", "s_atomic", "s_dcache"))

    @property
    def is_lds_dma(self):
        return self.is_vmem and ("lds" in self.text.split("//")[0].split(";")[0].split()[-3:] or "_lds_" in self.mnem)

    @property
    def wait_states(self):
        if self.mnem == "s_nop":
            try:
                return int(self.ops[0], 0) + 1
            except (ValueError, IndexError):
                return 1
        return 1

    def passes(self):
        m = re.search(r"_(\d+)x(\d+)x(\d+)", self.mnem)
        if not m:
            return 16
        a, b, k = map(int, m.groups())
        if "f64" in self.mnem:
            return 16
        # cycles = 2 a b k / (ops per clock per SIMD); i8 / fp8: 2048, bf16 / f16: 1024 on gfx950; 4 cycles per pass
        rate = 2048 if ("_i8" in self.mnem or "fp8" in self.mnem or "bf8" in self.mnem) else 1024 if ("bf16" in self.mnem or "f16" in self.mnem) else 256
        return max(2, min(16, (2 * a * b * k // rate) // 4))


def parse(text: str):
    """-> {function name: [Ins]}; labels kept as Ins with mnem '.label'."""
    funcs = {}
    cur: Optional[List[Ins]] = None
    in_asm = False
    for n, raw in enumerate(text.splitlines(), 1):
        line = raw.split("//")[0]
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
        if m:
            name = m.group(1)
            if not name.startswith((".L", "BB", ".Ltmp")):
                cur = funcs.setdefault(name, [])
            elif cur is not None:
                cur.append(Ins(n, s, ".label", [name]))
            continue
        if s.startswith(".") or cur is None:
            if s.startswith(".Lfunc_end") or s.startswith(".size"):
                pass
            continue
        parts = s.split(None, 1)
        mnem = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ins = Ins(n, s, mnem, ops, in_asm=in_asm)
        classify(ins)
        cur.append(ins)
    return funcs


def classify(i: Ins):
    ops = i.ops
    R = [regs_of(o) for o in ops]
    allr = set().union(*R) if R else set()
    m = i.mnem
    if m.startswith(("s_nop", "s_waitcnt", "s_barrier", "s_endpgm", "s_branch", "s_cbranch", "s_sleep", "s_setprio", "s_sethalt")):
        if m.startswith("s_cbranch_vcc"):
            i.uses = {("vcc", 0)}
        elif m.startswith("s_cbranch_exec"):
            i.uses = {("exec", 0)}
        return
    if i.is_mfma:
        i.defs, i.uses = R[0], set().union(*R[1:4])
        return
    if i.is_lds:
        if m.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or "_rtn" in m:
            i.defs, i.uses = R[0], set().union(*R[1:]) if len(R) > 1 else set()
        else:
            i.uses = allr
        return
    if i.is_vmem:
        is_load = "_load" in m or ("atomic" in m and (" sc0" in i.text or " glc" in i.text))
        if i.is_lds_dma:
            i.uses = allr | {("m0", 0)}
        elif is_load:
            i.defs, i.uses = R[0], set().union(*R[1:]) if len(R) > 1 else set()
            if "atomic" in m:
                i.uses |= R[0]
        else:
            i.uses = allr
        return
    if i.is_smem:
        if "load" in m:
            i.defs, i.uses = R[0], set().union(*R[1:]) if len(R) > 1 else set()
        else:
            i.uses = allr
        return
    if m.startswith("v_"):
        if m.startswith("v_cmp") or m.startswith("v_cmpx"):
            if m.endswith("_e64") or (ops and regs_of(ops[0]) and next(iter(regs_of(ops[0])))[0] in ("s", "vcc") and len(ops) == 3):
                i.defs, i.uses = R[0], set().union(*R[1:])
            else:
                i.defs, i.uses = {("vcc", 0)}, allr
            if m.startswith("v_cmpx"):
                i.defs |= {("exec", 0)}
            return
        ndef = 1
        if m.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")):
            ndef = 2
        i.defs = set().union(*R[:ndef]) if R else set()
        i.uses = set().union(*R[ndef:]) if len(R) > ndef else set()
        if m.startswith(("v_cndmask_b32_e32", "v_addc_co_u32_e32", "v_subb_co_u32_e32")):
            i.uses |= {("vcc", 0)}
        if "sdwa" in m and "UNUSED_PRESERVE" in i.text:
            i.uses |= i.defs
        if m.startswith(("v_readlane", "v_readfirstlane")):
            pass   # defs = the SGPR: as parsed
        return
    if m.startswith("s_"):
        if m.startswith(("s_cmp", "s_bitcmp")):
            i.defs, i.uses = {("scc", 0)}, allr
        elif m.startswith(("s_cselect", "s_addc", "s_subb")):
            i.defs, i.uses = R[0], set().union(*R[1:]) | {("scc", 0)}
        else:
            i.defs, i.uses = (R[0] if R else set()), (set().union(*R[1:]) if len(R) > 1 else set())
            if m.startswith(("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec")):
                i.defs |= {("exec", 0)}
                i.uses |= {("exec", 0)}
        return


def check_sequence(seq: List[Ins], fname: str, out: List[str], start_at: int = 0):
    """Hazards A-E, G, H over a linear sequence; only pairs whose SECOND instruction has index >= start_at are reported (the
    glued loop check reports the head only)."""
    for j in range(start_at, len(seq)):
        cur = seq[j]
        if cur.mnem == ".label":
            continue
        ws = 0   # wait states between the candidate producer and cur
        for k in range(j - 1, max(-1, j - 1 - 4 * WINDOW), -1):
            prev = seq[k]
            if prev.mnem == ".label":
                continue
            if ws >= WINDOW:
                break

            def flag(rule, need, what):
                if ws < need and (prev.in_asm or cur.in_asm):
                    out.append(f"{fname}: rule {rule}: line {prev.line} `{prev.text}` -> line {cur.line} `{cur.text}`: "
                               f"{ws} wait state(s), {need} needed ({what})")

            if prev.is_mfma:
                P = prev.passes()
                D = prev.defs
                if cur.is_mfma:
                    a_b = set().union(*[regs_of(o) for o in cur.ops[1:3]])
                    c = regs_of(cur.ops[3]) if len(cur.ops) > 3 else set()
                    if D & a_b:
                        flag("B", P + 3, "MFMA result read as SrcA / SrcB")
                    if (D & c) and c != D:
                        flag("C", P + 1, "MFMA result partly overlaps the next MFMA's SrcC")
                elif D & (cur.uses | (cur.defs if cur.is_valu else set())):
                    # (a memory LOAD into D is not a hazard: its data returns long after the MFMA has written)
                    flag("A", P + 3, "MFMA result touched by a non-MFMA instruction")
                c_prev = regs_of(prev.ops[3]) if len(prev.ops) > 3 else set()
                if cur.is_valu and c_prev and c_prev != D and (c_prev & cur.defs):
                    flag("E", P + 3, "VALU overwrites the SrcC an MFMA may still be reading")
            if prev.is_valu and cur.is_mfma:
                vg = {r for r in prev.defs if r[0] in ("v", "a")}
                if vg & cur.uses:
                    flag("D", 2, "VALU result read by an MFMA")
            if prev.is_valu and (cur.is_vmem or cur.is_smem):
                sg = {r for r in prev.defs if r[0] in ("s", "vcc")}
                if sg & cur.uses:
                    flag("G", 5, "VALU-written SGPR used by a memory instruction")
            if prev.mnem.startswith("s_") and ("m0", 0) in prev.defs and (cur.is_lds_dma or (cur.is_lds and "gws" in cur.mnem)):
                flag("H", 1, "M0 written right in front of an LDS-DMA")
            ws += prev.wait_states


def check_lgkm(seq: List[Ins], fname: str, out: List[str]):
    """Rule F: every register written by an LDS read must be covered by an s_waitcnt lgkmcnt before its first use."""
    pending: List[Tuple[int, Set[Tuple[str, int]], Ins]] = []   # (sequence number among lgkm ops, regs, ins)
    issued = 0
    smem_in_flight = False
    for ins in seq:
        if ins.mnem == ".label":
            continue
        if ins.mnem == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ins.text)
            k = None
            if m:
                k = int(m.group(1))
            elif re.fullmatch(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)", ins.text):
                k = (int(ins.text.split()[1], 0) >> 8) & 0xF
            if k is not None:
                if k == 0:
                    pending, smem_in_flight = [], False
                elif not smem_in_flight:
                    pending = [p for p in pending if p[0] >= issued - k]
            continue
        if pending:
            touched = ins.uses | ins.defs
            for n, regs, src in pending:
                if regs & touched and not (ins is src):
                    out.append(f"{fname}: rule F: line {src.line} `{src.text}` -> line {ins.line} `{ins.text}`: LDS result used "
                               f"before an s_waitcnt lgkmcnt covers it")
            pending = [p for p in pending if not (p[1] & touched)]
        if ins.is_lds or ins.is_smem:
            if ins.is_smem:
                smem_in_flight = True
            if ins.defs and ins.is_lds:
                pending.append((issued, set(ins.defs), ins))
            issued += 1


def check_function(name: str, seq: List[Ins]) -> List[str]:
    out: List[str] = []
    check_sequence(seq, name, out)
    check_lgkm([i for i in seq], name, out)
    # backward branches: loop tail + loop head
    labels = {i.ops[0]: n for n, i in enumerate(seq) if i.mnem == ".label"}
    for n, i in enumerate(seq):
        if i.mnem.startswith(("s_branch", "s_cbranch")) and i.ops and i.ops[0] in labels and labels[i.ops[0]] < n:
            head = labels[i.ops[0]]
            tail = seq[max(0, n - 4 * WINDOW):n + 1]
            glued = tail + seq[head:head + 4 * WINDOW]
            check_sequence(glued, name + " (loop back-edge)", out, start_at=len(tail))
    # one report per pair
    seen, uniq = set(), []
    for o in out:
        key = re.sub(r" \(loop back-edge\)", "", o)
        if key not in seen:
            seen.add(key)
            uniq.append(o)
    return uniq


SELFTEST = {
    "bias_c_operand (round 2: k_mfma16p FIRST pass without its s_nop)": ("""
f:
	v_bfe_i32 v10, v2, 0, 8
	v_bfe_i32 v11, v2, 8, 8
	v_bfe_i32 v12, v2, 16, 8
	v_ashrrev_i32_e32 v13, 24, v2
	;;#ASMSTART
	v_mfma_i32_16x16x64_i8 v[20:23], v[30:33], v[40:43], v[10:13]
	;;#ASMEND
	s_endpgm
""", "D"),
    "atomic_address_from_readlane (round 2: k_conv_pp ticket without its s_nop)": ("""
f:
	v_readlane_b32 s4, v195, 23
	v_readlane_b32 s5, v195, 24
	;;#ASMSTART
	global_atomic_add v1, v0, v1, s[4:5] sc0 sc1
	;;#ASMEND
	s_endpgm
""", "G"),
    "accumulator_read_behind_mfma (round 3: C = 0 MFMA with an output-only operand)": ("""
f:
	;;#ASMSTART
	v_mfma_i32_16x16x64_i8 a[28:31], v[76:79], v[4:7], 0
	;;#ASMEND
	v_pk_max_i16 v2, v2, s29
	v_accvgpr_read_b32 v151, a31
	s_endpgm
""", "A"),
    "lds_fragment_without_wait": ("""
f:
	;;#ASMSTART
	ds_read_b128 v[4:7], v32 offset:32
	;;#ASMEND
	;;#ASMSTART
	v_mfma_i32_16x16x64_i8 a[0:3], v[4:7], v[8:11], a[0:3]
	;;#ASMEND
	s_endpgm
""", "F"),
    "clean accumulate chain": ("""
f:
	;;#ASMSTART
	ds_read_b128 v[4:7], v32 offset:32
	;;#ASMEND
	s_waitcnt lgkmcnt(0)
	;;#ASMSTART
	v_mfma_i32_16x16x64_i8 a[0:3], v[4:7], v[8:11], a[0:3]
	;;#ASMEND
	;;#ASMSTART
	v_mfma_i32_16x16x64_i8 a[0:3], v[4:7], v[8:11], a[0:3]
	;;#ASMEND
	s_nop 7
	v_accvgpr_read_b32 v1, a0
	s_endpgm
""", None),
}


def selftest() -> int:
    bad = 0
    for name, (text, want) in SELFTEST.items():
        found = []
        for fn, seq in parse(text).items():
            found += check_function(fn, seq)
        rules = {re.search(r"rule (\w)", f).group(1) for f in found}
        ok = (want in rules) if want else not found
        print(("ok   " if ok else "FAIL ") + name + (": flagged " + ", ".join(sorted(rules)) if rules else ": clean"))
        bad += not ok
    return bad


def main(argv):
    if argv[1:] == ["--selftest"]:
        return 1 if selftest() else 0
    total = 0
    for path in argv[1:]:
        funcs = parse(open(path).read())
        n_asm = 0
        for fn, seq in funcs.items():
            n_asm += sum(1 for i in seq if i.in_asm)
            for v in check_function(fn, seq):
                print(v)
                total += 1
        print(f"{path}: {len(funcs)} functions, {n_asm} instructions inside asm statements, {total} violation(s) so far", file=sys.stderr)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
