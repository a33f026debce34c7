#!/usr/bin/env python3
"""Build-time guard: the software-visible data hazards of gfx950 (the gfx90a / gfx940
family's "manually inserted wait states") around and inside inline-asm blocks.

hipcc's hazard recogniser inserts the s_nop a hazard needs between the instructions
IT schedules; it does not look inside an asm statement (GCNHazardRecognizer::
checkInlineAsmHazards covers store-data hazards only).  The kernels here carry
several dozen hand-written blocks, so every hazard whose producer or consumer lies
inside an asm block is ours to keep.  This script walks the device assembly of a
translation unit (hipcc -save-temps, the object that ships) and counts wait states
between producer and consumer (every instruction is one, `s_nop N` is N + 1):

  H1  VALU writes an SGPR / VCC (v_readlane, v_readfirstlane, v_cmp, carry-out,
      v_div_scale)  ->  v_readlane / v_writelane with THAT register as lane select   4
  H2  VALU writes an SGPR  ->  vector memory instruction that reads it (address,
      offset)                                                                        5
  H3  VALU writes a VGPR  ->  DPP instruction that reads it                           2
  H4  VALU writes EXEC (v_cmpx)  ->  DPP instruction                                  5
  H5  VALU writes EXEC (v_cmpx)  ->  v_readlane / v_readfirstlane / v_writelane       4
  H6  VALU writes VCC  ->  v_div_fmas                                                 4
  H7  SALU writes M0  ->  LDS add-tid / LDS-DMA / s_sendmsg / s_movrel                1
  H8  transcendental VALU (v_exp, v_log, v_rcp, v_rsq, v_sqrt, v_sin, v_cos) writes
      a VGPR  ->  non-transcendental VALU that reads it                               1
  H10 VALU writes an SGPR or VCC  ->  VALU that reads it as an operand -- an SGPR source,
      v_cndmask's mask, a carry-in (LLVM: VALUWriteSGPRVALURead; the compiler puts `s_nop 1`
      between a v_cmp and the v_cndmask on its VCC).  Until round 3 the walk of the LZ4 mix
      kernel had several hundred v_cmp / v_cndmask pairs back to back on VCC (never wrong in
      any parity run); they are compares into SGPR pairs now, four windows at a time, with
      the selects behind them (lz4_mix.hiph) -- which also runs 1.6 % faster               2
  H9  VALU writes a VGPR  ->  v_readlane / v_readfirstlane / v_writelane that reads
      it (LLVM: VALUWriteVGPRReadlaneRead; met on the GPU in round 3: a v_readlane
      opening an asm block read the value its VGPR held BEFORE the v_or3_b32 in
      front of it)                                                                   1

A violation with producer or consumer inside ";;#ASMSTART .. ;;#ASMEND" fails the
build.  Pairs that lie wholly in compiler-scheduled code are reported as a
calibration figure (the compiler keeps them: the count must be 0, else this script
misreads the assembly).  The assembly is read in layout order (the fall-through path) AND along
the taken path of every branch to a label of the same function, forward or backward -- the loops
inside asm statements (`1: ... s_cbranch_scc1 1b`: the decoders' walks, the encoders' chain walks)
have the last instructions of their body as predecessors of the first ones.  A taken branch is
counted as the one wait state its instruction is (that is how the compiler counts: calibration).

usage: check_asm_hazards.py file.s [file.s ...]   exit 0 = fine
"""
import re
import sys

SREG = re.compile(r"^(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0)$")
VREG = re.compile(r"^(v\d+|v\[\d+:\d+\])$")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
VMEM_PREFIX = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_")


def regs_of(tok):
    """register token -> set of names like s4, v7, vcc_lo, exec_lo, m0"""
    tok = tok.strip()
    m = re.match(r"^([sv])\[(\d+):(\d+)\]$", tok)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    if re.match(r"^[sv]\d+$", tok):
        return {tok}
    if tok == "vcc":
        return {"vcc_lo", "vcc_hi"}
    if tok == "exec":
        return {"exec_lo", "exec_hi"}
    if tok in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi", "m0"):
        return {tok}
    return set()


def parse(line):
    code = line.split(";", 1)[0].strip()
    if not code or code.endswith(":") or code.startswith("."):
        return None
    parts = code.split(None, 1)
    mn = parts[0]
    ops = []
    if len(parts) > 1:
        # operands separated by commas; modifiers (row_shr:1, offset:4, sc0 ...) trail the last one
        for chunk in parts[1].split(","):
            ops.append(chunk.strip())
    return mn, ops


def operand_regs(op):
    """registers named in one operand text (also "v3 row_shr:1" or "s[2:3] offset:16")"""
    out = set()
    for tok in re.split(r"[\s]+", op):
        tok = tok.strip()
        out |= regs_of(re.sub(r"^[-|]|\|$", "", tok))
        m = re.match(r"^(?:neg|abs|sext)\((.*)\)$", tok)
        if m:
            out |= regs_of(m.group(1))
    return out


def is_dpp(mn, ops):
    text = " ".join(ops)
    return "_dpp" in mn or re.search(r"\b(quad_perm|row_shl|row_shr|row_ror|row_bcast|row_mirror|row_half_mirror|"
                                     r"row_newbcast|wave_shl|wave_shr|wave_rol|wave_ror|row_share|row_xmask)\b", text)


class Inst:
    __slots__ = ("mn", "ops", "raw", "in_asm", "states", "read", "lane_select", "is_vmem", "dpp", "is_valu", "target")


def analyse(mn, ops, raw, in_asm):
    """what an instruction reads, and as what (the consumer side of every rule)"""
    it = Inst()
    it.mn, it.ops, it.raw, it.in_asm = mn, ops, raw.strip(), in_asm
    it.states = 1
    if mn == "s_nop":
        try:
            it.states = int(ops[0], 0) + 1
        except (ValueError, IndexError):
            it.states = 1
    read = set()
    # (second operand of a VALU with carry-out / 64-bit multiply-add: the SGPR pair it WRITES)
    second_is_dst = mn.startswith("v_") and ("_co_" in mn or "div_scale" in mn or mn.startswith(("v_mad_u64_u32", "v_mad_i64_i32")))
    for i, op in enumerate(ops):
        if i == 0 and not (mn.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "ds_write",
                                           "s_cmp", "v_cmpx", "s_bitcmp", "global_atomic", "s_waitcnt", "s_cbranch"))):
            continue  # destination
        if i == 1 and second_is_dst:
            continue
        read |= operand_regs(op)
    it.read = read
    it.lane_select = set()
    if mn.startswith(("v_readlane_b32", "v_writelane_b32")) and len(ops) >= 3:
        it.lane_select = operand_regs(ops[2])
    it.is_vmem = mn.startswith(VMEM_PREFIX)
    it.dpp = bool(is_dpp(mn, ops)) and mn.startswith("v_")
    it.is_valu = mn.startswith("v_")
    it.target = None
    if mn.startswith("s_cbranch") or mn == "s_branch":
        it.target = ops[-1].strip() if ops else None
    return it


def producers_of(it):
    """the producer records an instruction leaves behind (kind, regs)"""
    out = []
    mn, ops = it.mn, it.ops
    if mn.startswith("v_") and ops:
        dst = operand_regs(ops[0]) if not mn.startswith("v_cmpx") else set()
        sdst = {r for r in dst if r[0] == "s" or r.startswith(("vcc", "exec"))}
        if mn.startswith("v_cmpx"):
            out.append(("valu_exec", {"exec_lo", "exec_hi"}))
        if ("_co_" in mn or "div_scale" in mn or mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64_i32")) and len(ops) > 1:
            sdst |= {r for r in operand_regs(ops[1]) if not r.startswith("v") or r.startswith("vcc")}
        if sdst:
            out.append(("valu_exec" if sdst <= {"exec_lo", "exec_hi"} else "valu_sgpr", sdst))
        vdst = {r for r in dst if re.match(r"^v\d+$", r)}
        if vdst:
            out.append(("trans_vgpr" if mn.startswith(TRANS) else "valu_vgpr", vdst))
    elif mn.startswith("s_") and ops and "m0" in operand_regs(ops[0]) and not mn.startswith(("s_cmp", "s_waitcnt")):
        out.append(("salu_m0", {"m0"}))
    return out


def rule_for(pr, it):
    """(wait states needed, rule text) for producer record pr in front of instruction it, or (0, None)"""
    mn, read = it.mn, it.read
    if pr["kind"] == "valu_sgpr" and it.lane_select & pr["regs"]:
        return 4, "H1 VALU-written SGPR as lane select"
    if pr["kind"] == "valu_sgpr" and it.is_vmem and (read & pr["regs"]) and not (pr["regs"] <= {"exec_lo", "exec_hi"}):
        return 5, "H2 VALU-written SGPR read by vector memory"
    if pr["kind"] == "valu_vgpr" and it.dpp and (read & pr["regs"]):
        return 2, "H3 VALU-written VGPR read by DPP"
    if pr["kind"] == "valu_exec" and it.dpp:
        return 5, "H4 VALU-written EXEC before DPP"
    if pr["kind"] == "valu_exec" and mn.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return 4, "H5 VALU-written EXEC before lane access"
    if pr["kind"] == "valu_sgpr" and mn.startswith("v_div_fmas") and (pr["regs"] & {"vcc_lo", "vcc_hi"}):
        return 4, "H6 VALU-written VCC before v_div_fmas"
    if pr["kind"] == "salu_m0" and (("addtid" in mn) or mn.startswith(("s_sendmsg", "s_movrel"))
                                    or (it.is_vmem and any("lds" in o for o in it.ops))):
        return 1, "H7 SALU-written M0"
    if pr["kind"] == "trans_vgpr" and it.is_valu and not mn.startswith(TRANS) and (read & pr["regs"]):
        return 1, "H8 transcendental result read by VALU"
    if pr["kind"] == "valu_vgpr" and mn.startswith(("v_readlane", "v_readfirstlane", "v_writelane")) and (read & pr["regs"]):
        return 1, "H9 VALU-written VGPR read by a lane access"
    if pr["kind"] == "valu_sgpr" and it.is_valu and (read & pr["regs"]) and not (it.lane_select & pr["regs"]):
        return 2, "H10 VALU-written SGPR / VCC read by VALU"
    return 0, None


HORIZON = 6  # no rule asks for more wait states than 5


def functions_of(text):
    """[(name, items)]; an item is an Inst or a label name (str)"""
    funcs, items, in_asm = [], None, False
    for raw in text.splitlines():
        m = re.match(r"^(_Z\S+|[A-Za-z_]\w*):\s*(;.*)?$", raw)
        if m and not raw.startswith(".L"):
            items = []
            funcs.append((m.group(1), items))
            in_asm = False
            continue
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        if items is None:
            continue
        code = raw.split(";", 1)[0].strip()
        m = re.match(r"^(\.L\w+|\d+):$", code)
        if m:
            items.append(m.group(1))
            continue
        p = parse(raw)
        if p is None:
            continue
        items.append(analyse(p[0], p[1], raw, in_asm))
    return funcs


def resolve(items, labels, at, target):
    """index of the label a branch at index `at` names: `.Lxyz`, or a numeric local label `1b` / `2f`
    (the nearest definition before / after the branch)"""
    if target is None:
        return None
    m = re.match(r"^(\d+)([bf])$", target)
    if m:
        defs = labels.get(m.group(1), [])
        if m.group(2) == "b":
            before = [i for i in defs if i < at]
            return before[-1] if before else None
        after = [i for i in defs if i > at]
        return after[0] if after else None
    defs = labels.get(target, [])
    return defs[0] if defs else None


def check(path):
    text = open(path).read()
    failures, calibration = [], 0
    seen = set()

    def consume(func, recent, it, via):
        nonlocal calibration
        for pr in recent:
            need, why = rule_for(pr, it)
            if why and pr["age"] < need:
                key = (func, pr["text"], it.raw, why, via)
                if key in seen:
                    continue
                seen.add(key)
                if pr["in_asm"] or it.in_asm:
                    failures.append(f"{path}: {func}: {why}: `{pr['text']}` then `{it.raw}` after "
                                    f"{pr['age']} wait state(s){via}, {need} needed")
                else:
                    calibration += 1

    for func, items in functions_of(text):
        labels = {}
        for i, it in enumerate(items):
            if isinstance(it, str):
                labels.setdefault(it, []).append(i)
        recent = []  # producers in flight: dicts(kind, regs, age, in_asm, text)
        for i, it in enumerate(items):
            if isinstance(it, str):
                continue
            if it.mn in ("s_endpgm", "s_setpc_b64"):
                recent = []
                continue
            consume(func, recent, it, "")
            for pr in recent:
                pr["age"] += it.states
            recent = [pr for pr in recent if pr["age"] < HORIZON]
            for kind, regs in producers_of(it):
                recent.append(dict(kind=kind, regs=regs, age=0, in_asm=it.in_asm, text=it.raw))
            # The taken path of a branch, forward or backward (a loop inside an asm statement: the last
            # instructions of its body are the predecessors of its first ones): what is in flight at the
            # branch meets the instructions at its target.  The branch instruction itself is one wait state
            # like any other (it took its place in `age` above) -- the compiler's own count: read as none,
            # 21 pairs of compiler-scheduled code in lz4_kernels come out one state short, read as one, none.
            if it.target is not None:
                t = resolve(items, labels, i, it.target)
                if t is not None:
                    flight = [dict(pr) for pr in recent]
                    j = t
                    while flight and j < len(items):
                        nx = items[j]
                        j += 1
                        if isinstance(nx, str):
                            continue
                        if nx.mn in ("s_endpgm", "s_setpc_b64"):
                            break
                        consume(func, flight, nx, f" across the branch `{it.raw}`")
                        for pr in flight:
                            pr["age"] += nx.states
                        flight = [pr for pr in flight if pr["age"] < HORIZON]
                        if nx.mn == "s_branch":
                            break
                if it.mn == "s_branch":
                    recent = []
    return failures, calibration


def main(paths):
    bad = 0
    for p in paths:
        failures, calibration = check(p)
        for f in failures:
            sys.stderr.write("check_asm_hazards: " + f + "\n")
        if calibration:
            sys.stderr.write(f"check_asm_hazards: {p}: {calibration} pair(s) in compiler-scheduled code read as too close "
                             "(calibration: expected 0)\n")
        bad += len(failures)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
