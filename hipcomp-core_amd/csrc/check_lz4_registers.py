#!/usr/bin/env python3
"""Build-time guard of the LZ4 encoder kernels, run by csrc/Makefile on the device
assembly of the very object that ships (hipcc -save-temps=obj, same flags).

The "mix" kernels keep loads in flight in accumulation registers a0..a23 that only
their inline asm names (lz4_mix.hiph, HC_WALK_AGPRS); the compiler does not know
they are live between two asm statements.  That is sound only while the compiler
itself never touches an AGPR in those kernels -- not to spill a vector register
(v_accvgpr_write / v_accvgpr_mov), not as the destination of a load it allocated
(gfx90a+ lets it), not through scratch.  So, per kernel:

  mix   every mention of an AGPR lies INSIDE an inline-asm block (";;#ASMSTART" ..
        ";;#ASMEND"), .num_agpr == 24, no scratch, <= 256 VGPRs;
  pair  the same with .num_agpr == 26 (a24, a25: the early look at the pair's token) and
        VGPRs + AGPRs <= 256: two waves per SIMD;
  far   no AGPR at all, no scratch, <= 64 VGPRs (eight waves per SIMD);
  all   no *_d16 loads (table entries are read zero-extended: walk_probe relies on it).

usage: check_lz4_registers.py <device .s>     exit 0 = fine, 1 = message on stderr
"""
import re
import sys

AGPR = re.compile(r"(?<![\w.$])a(?:\d+|\[\d+(?::\d+)?\])(?![\w])")


def kernels(text):
    """name -> list of source lines of its body"""
    out = {}
    name, body = None, []
    for ln in text.splitlines():
        m = re.match(r"^(_ZN5hcamd\S*lz4_\w+kernel\S*):\s*(;.*)?$", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            if re.match(r"^\.Lfunc_end\d+:", ln):
                out[name] = body
                name = None
            else:
                body.append(ln)
    return out


def main(path):
    text = open(path).read()
    errors = []
    ks = kernels(text)
    sets = {}
    for m in re.finditer(r"\.set (\S+)\.(num_agpr|num_vgpr|private_seg_size), (\d+)", text):
        sets.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
    mix = [k for k in ks if "lz4_compress_kernel_mix" in k]
    pair = [k for k in ks if "lz4_compress_kernel_pair" in k]
    far = [k for k in ks if "lz4_compress_kernel_far" in k or "lz4_compress_kernel_near" in k
           or "lz4_compress_kernel_both" in k]
    if len(mix) != 3:
        errors.append(f"expected 3 mix kernels (element size 1, 2, 4), found {len(mix)}")
    if len(far) < 9:
        errors.append(f"expected at least 9 far kernels (3 element sizes x lean, lean with chains, wide), found {len(far)}")
    if len(pair) != 3:
        errors.append(f"expected 3 pair kernels (element size 1, 2, 4), found {len(pair)}")
    for k in mix + pair + far:
        s = sets.get(k, {})
        in_asm = False
        for ln in ks[k]:
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            code = ln.split(";", 1)[0]
            if not in_asm and AGPR.search(code):
                errors.append(f"{k}: the compiler uses an accumulation register: {code.strip()}")
                break
        if s.get("private_seg_size", -1) != 0:
            errors.append(f"{k}: scratch memory in use ({s.get('private_seg_size')})")
        if k in mix:
            if s.get("num_agpr") != 24:
                errors.append(f"{k}: num_agpr {s.get('num_agpr')} != 24")
            if s.get("num_vgpr", 999) > 256:
                errors.append(f"{k}: num_vgpr {s.get('num_vgpr')} > 256")
        elif k in pair:
            # two waves per SIMD: vector + accumulation registers within 256 (the ring a0..a23 and the
            # token look a24, a25)
            if s.get("num_agpr") != 26:
                errors.append(f"{k}: num_agpr {s.get('num_agpr')} != 26")
            if (s.get("num_vgpr", 999) + 7) // 8 * 8 + 26 > 256:
                errors.append(f"{k}: num_vgpr {s.get('num_vgpr')} + 26 accumulation registers > 256 (two waves per SIMD)")
        else:
            if s.get("num_agpr") != 0:
                errors.append(f"{k}: num_agpr {s.get('num_agpr')} != 0")
            limit = 128 if "kernel_both" in k or "kernel_near" in k else 64
            if s.get("num_vgpr", 999) > limit:
                errors.append(f"{k}: num_vgpr {s.get('num_vgpr')} > {limit}")
    if re.search(r"\b(?:ds_read|global_load|buffer_load)\w*_d16", text):
        errors.append("a *_d16 load: table entries must be read zero-extended")
    if "v_accvgpr_write" in text or "v_accvgpr_mov" in text:
        errors.append("v_accvgpr_write / v_accvgpr_mov present")
    for e in errors:
        sys.stderr.write("check_lz4_registers: " + e + "\n")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
