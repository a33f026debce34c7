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
misreads the assembly).  Straight-line reading in layout order: a hazard across a
taken branch into the middle of an asm block is not something these kernels do.

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


def check(path):
    text = open(path).read()
    failures, calibration = [], 0
    func, in_asm = None, False
    # recent producers: list of dicts(kind, regs, age, in_asm, text)
    recent = []
    for raw in text.splitlines():
        m = re.match(r"^(_Z\S+|[A-Za-z_]\w*):\s*(;.*)?$", raw)
        if m and not raw.startswith(".L"):
            func, recent, in_asm = m.group(1), [], False
            continue
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        p = parse(raw)
        if p is None or func is None:
            continue
        mn, ops = p
        if mn in ("s_endpgm", "s_setpc_b64", "s_branch"):
            recent = []
            continue
        states = 1
        if mn == "s_nop":
            try:
                states = int(ops[0], 0) + 1
            except (ValueError, IndexError):
                states = 1
        # ---- consumer checks against what is in flight
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
        lane_select = set()
        if mn.startswith(("v_readlane_b32", "v_writelane_b32")) and len(ops) >= 3:
            lane_select = operand_regs(ops[2])
        is_vmem = mn.startswith(VMEM_PREFIX)
        dpp = bool(is_dpp(mn, ops)) and mn.startswith("v_")
        is_valu = mn.startswith("v_")
        for pr in recent:
            need = 0
            why = None
            if pr["kind"] == "valu_sgpr" and lane_select & pr["regs"]:
                need, why = 4, "H1 VALU-written SGPR as lane select"
            elif pr["kind"] == "valu_sgpr" and is_vmem and (read & pr["regs"]) and not (pr["regs"] <= {"exec_lo", "exec_hi"}):
                need, why = 5, "H2 VALU-written SGPR read by vector memory"
            elif pr["kind"] == "valu_vgpr" and dpp and (read & pr["regs"]):
                need, why = 2, "H3 VALU-written VGPR read by DPP"
            elif pr["kind"] == "valu_exec" and dpp:
                need, why = 5, "H4 VALU-written EXEC before DPP"
            elif pr["kind"] == "valu_exec" and mn.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
                need, why = 4, "H5 VALU-written EXEC before lane access"
            elif pr["kind"] == "valu_sgpr" and mn.startswith("v_div_fmas") and (pr["regs"] & {"vcc_lo", "vcc_hi"}):
                need, why = 4, "H6 VALU-written VCC before v_div_fmas"
            elif pr["kind"] == "salu_m0" and (("addtid" in mn) or mn.startswith(("s_sendmsg", "s_movrel"))
                                                or (is_vmem and any("lds" in o for o in ops))):
                need, why = 1, "H7 SALU-written M0"
            elif pr["kind"] == "trans_vgpr" and is_valu and not mn.startswith(TRANS) and (read & pr["regs"]):
                need, why = 1, "H8 transcendental result read by VALU"
            elif pr["kind"] == "valu_vgpr" and mn.startswith(("v_readlane", "v_readfirstlane", "v_writelane")) \
                    and (read & pr["regs"]):
                need, why = 1, "H9 VALU-written VGPR read by a lane access"
            elif pr["kind"] == "valu_sgpr" and is_valu and (read & pr["regs"]) and not (lane_select & pr["regs"]):
                need, why = 2, "H10 VALU-written SGPR / VCC read by VALU"
            if why and pr["age"] < need:
                if pr["in_asm"] or in_asm:
                    failures.append(f"{path}: {func}: {why}: `{pr['text']}` then `{raw.strip()}` after "
                                    f"{pr['age']} wait state(s), {need} needed")
                else:
                    calibration += 1
        # ---- age, then register this instruction as a producer
        for pr in recent:
            pr["age"] += states
        recent = [pr for pr in recent if pr["age"] < 6]
        if mn.startswith("v_") and ops:
            dst = operand_regs(ops[0]) if not mn.startswith("v_cmpx") else set()
            sdst = {r for r in dst if r[0] == "s" or r.startswith(("vcc", "exec"))}
            if mn.startswith("v_cmpx"):
                recent.append(dict(kind="valu_exec", regs={"exec_lo", "exec_hi"}, age=0, in_asm=in_asm, text=raw.strip()))
            if ("_co_" in mn or "div_scale" in mn or mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64_i32")) and len(ops) > 1:
                sdst |= {r for r in operand_regs(ops[1]) if not r.startswith("v") or r.startswith("vcc")}
            if sdst:
                kind = "valu_exec" if sdst <= {"exec_lo", "exec_hi"} else "valu_sgpr"
                recent.append(dict(kind=kind, regs=sdst, age=0, in_asm=in_asm, text=raw.strip()))
            vdst = {r for r in dst if re.match(r"^v\d+$", r)}
            if vdst:
                recent.append(dict(kind="trans_vgpr" if mn.startswith(TRANS) else "valu_vgpr", regs=vdst, age=0,
                                   in_asm=in_asm, text=raw.strip()))
        elif mn.startswith("s_") and ops and "m0" in operand_regs(ops[0]) and not mn.startswith(("s_cmp", "s_waitcnt")):
            recent.append(dict(kind="salu_m0", regs={"m0"}, age=0, in_asm=in_asm, text=raw.strip()))
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
