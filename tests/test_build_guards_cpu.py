"""Build-time guards of the LZ4 encoder (no GPU): its walk keeps loads in flight
in accumulation registers a0..a23 that it names in inline asm (lz4_mix.hiph,
HC_WALK_AGPRS).  That is only sound while the compiler itself never touches an
AGPR in those kernels.  The guard lives in the build (csrc/Makefile runs
csrc/check_lz4_registers.py on the device assembly of the very object that goes
into libhipcomp.so and refuses to keep an object that fails); here: the guard
accepts the shipped build's assembly, and it does catch what it is there for."""
import importlib.util
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hipcomp-core_amd", "csrc")
ASM = os.path.join(CSRC, "build", "lz4_kernels.gfx950.s")
CHECK = os.path.join(CSRC, "check_lz4_registers.py")


def _run(path):
    return subprocess.run([sys.executable, CHECK, path], capture_output=True, text=True)


def test_shipped_object_passed_the_register_guard():
    assert os.path.exists(ASM), "csrc/Makefile keeps the checked device assembly next to the object: run build()"
    obj = os.path.join(CSRC, "build", "lz4_kernels.hip.o")
    assert abs(os.path.getmtime(obj) - os.path.getmtime(ASM)) < 300   # same make rule, same compile
    r = _run(ASM)
    assert r.returncode == 0, r.stderr
    text = open(ASM).read()
    agprs = dict(re.findall(r"\.set (\S*lz4_compress_kernel_\S*)\.num_agpr, (\d+)", text))
    assert {v for k, v in agprs.items() if "kernel_mix" in k} == {"24"}
    assert {v for k, v in agprs.items() if "kernel_pair" in k} == {"26"}   # (+ a24, a25: the early look at the pair's token)
    assert {v for k, v in agprs.items() if "kernel_mix" not in k and "kernel_pair" not in k} == {"0"}
    # the pair kernels run two waves per SIMD: vector + accumulation registers within 256
    vgprs = dict(re.findall(r"\.set (\S*lz4_compress_kernel_pair\S*)\.num_vgpr, (\d+)", text))
    assert len(vgprs) == 3 and all((int(v) + 7) // 8 * 8 + 26 <= 256 for v in vgprs.values()), vgprs


def test_makefile_runs_the_guard_on_the_object_it_ships():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    rule = mk[mk.index("$(OBJDIR)/%.hip.o:"):]
    assert "-save-temps=obj" in rule and "check_lz4_registers.py" in rule
    assert rule.index("check_lz4_registers.py $(OBJDIR)") < rule.index("mv $(OBJDIR)/$*_temps/$*.hip.o")
    assert "$(HIPCC) $(CXXFLAGS)" in rule.split("\n")[2]          # the flags of every object, EXTRA included


def test_guard_catches_compiler_use_of_accumulation_registers(tmp_path):
    text = open(ASM).read()
    # (1) a compiler-made AGPR access outside any inline-asm block of a mix kernel
    k = re.search(r"^(_ZN5hcamd\S*lz4_compress_kernel_mixILi1E\S*):", text, re.M)
    at = text.index("\n", k.end()) + 1
    kp = re.search(r"^(_ZN5hcamd\S*lz4_compress_kernel_pairILi1E\S*):", text, re.M)
    vp = re.search(re.escape(kp.group(1)) + r"\.num_vgpr, (\d+)", text).group(1)
    for doctored, what in (
        # a pair kernel that no longer fits twice on a SIMD
        (text.replace(kp.group(1) + ".num_vgpr, " + vp, kp.group(1) + ".num_vgpr, 236"), "two waves per SIMD"),
        (text[:at] + "\tv_accvgpr_read_b32 v1, a7\n" + text[at:], "uses an accumulation register"),
        (text[:at] + "\tglobal_load_dword a[3], v1, s[2:3]\n" + text[at:], "uses an accumulation register"),
        (text.replace(k.group(1) + ".num_agpr, 24", k.group(1) + ".num_agpr, 32"), "num_agpr 32"),
        (text.replace(k.group(1) + ".private_seg_size, 0", k.group(1) + ".private_seg_size, 16"), "scratch"),
        (text[:at] + "\tv_accvgpr_write_b32 a30, v1\n" + text[at:], "v_accvgpr_write"),
    ):
        p = tmp_path / "x.s"
        p.write_text(doctored)
        r = _run(str(p))
        assert r.returncode == 1 and what in r.stderr, (what, r.stderr[-300:])


# ---- the hazard guard (csrc/check_asm_hazards.py) -------------------------------------------
HAZ = os.path.join(CSRC, "check_asm_hazards.py")


def _haz(path):
    return subprocess.run([sys.executable, HAZ, path], capture_output=True, text=True)


def test_every_shipped_kernel_object_passed_the_hazard_guard():
    """csrc/Makefile keeps the checked device assembly of every kernel object; the guard accepts all
    of it and reads compiler-scheduled code as hazard-free (its calibration)."""
    import glob
    files = sorted(glob.glob(os.path.join(CSRC, "build", "*.gfx950.s")))
    assert {os.path.basename(f).split(".")[0] for f in files} >= {"lz4_kernels", "snappy_kernels", "cascaded_kernels",
                                                                   "hlif", "primitives"}
    for f in files:
        r = _haz(f)
        assert r.returncode == 0 and "calibration" not in r.stderr, (f, r.stderr[-500:])
    mk = open(os.path.join(CSRC, "Makefile")).read()
    rule = mk[mk.index("$(OBJDIR)/%.hip.o:"):]
    assert rule.index("check_asm_hazards.py $(OBJDIR)") < rule.index("mv $(OBJDIR)/$*_temps/$*.hip.o")


def _snippet(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text("\t.text\nkernel_x:\n" + body + "\ts_endpgm\n.Lfunc_end0:\n")
    return str(p)


def test_hazard_guard_catches_what_it_is_there_for(tmp_path):
    cases = {
        # the one it found in round 3: the ticket counter's address back from a spill lane, then
        # four instructions of the asm statement, then the atomic (five wait states needed)
        "H2": ("\tv_readlane_b32 s15, v54, 9\n\t;;#ASMSTART\n\ts_mov_b64 s[20:21], exec\n\ts_and_b64 exec, s[20:21], 1\n"
               "\tv_mov_b32_e32 v20, 0\n\tv_mov_b32_e32 v21, s3\n\tglobal_atomic_add v12, v20, v21, s[14:15] sc0\n\t;;#ASMEND\n"),
        "H1": ("\t;;#ASMSTART\n\tv_readfirstlane_b32 s4, v1\n\tv_readlane_b32 s5, v2, s4\n\t;;#ASMEND\n"),
        "H3": ("\t;;#ASMSTART\n\tv_add_u32_e32 v3, v1, v2\n\tv_mov_b32_dpp v4, v3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t;;#ASMEND\n"),
        "H5": ("\t;;#ASMSTART\n\tv_cmpx_eq_u32_e32 v1, v2\n\tv_readfirstlane_b32 s4, v1\n\t;;#ASMEND\n"),
        # met on the GPU in round 3 (the chain walk of the LZ4 encoder wrote a sequence twice): the v_readlane that
        # opens an asm statement right behind the compiler's instruction that made its source register
        # a VALU-written SGPR as a VALU source two instructions too early (the mix kernel's v_readlane of window
        # lane 31's slot in front of the compare of its insert rule, until round 3)
        "H10 ": ("\tv_readlane_b32 s10, v91, 63\n\t;;#ASMSTART\n\tv_cmp_ne_u32_e32 vcc, s10, v91\n\t;;#ASMEND\n"),
        "H9": ("\tv_or3_b32 v17, v42, v17, v15\n\t;;#ASMSTART\n\tv_readlane_b32 s27, v17, 0\n\ts_cmp_lt_i32 s27, 0\n\t;;#ASMEND\n"),
    }
    for code, body in cases.items():
        r = _haz(_snippet(tmp_path, body))
        assert r.returncode == 1 and code in r.stderr, (code, r.stderr)
    fine = {
        "H2 with the s_nop": cases["H2"].replace("\tglobal_atomic_add", "\ts_nop 0\n\tglobal_atomic_add"),
        "H1 four states on": ("\t;;#ASMSTART\n\tv_readfirstlane_b32 s4, v1\n\ts_nop 3\n\tv_readlane_b32 s5, v2, s4\n\t;;#ASMEND\n"),
        "H9 one state on": cases["H9"].replace("\tv_readlane_b32 s27", "\ts_mov_b64 s[38:39], 1\n\tv_readlane_b32 s27"),
        "SALU-made lane select": ("\t;;#ASMSTART\n\ts_add_u32 s4, s6, 1\n\tv_readlane_b32 s5, v2, s4\n\t;;#ASMEND\n"),
    }
    for what, body in fine.items():
        r = _haz(_snippet(tmp_path, body))
        assert r.returncode == 0, (what, r.stderr)
    # v_cmp -> v_cndmask on VCC back to back inside asm (the mix kernel's walk until round 3): the same rule
    r = _haz(_snippet(tmp_path, "\t;;#ASMSTART\n\tv_cmp_ne_u32_e32 vcc, 0xffff, v2\n\tv_cndmask_b32_e32 v0, v1, v2, vcc\n\t;;#ASMEND\n"))
    assert r.returncode == 1 and "H10" in r.stderr, r.stderr
    r = _haz(_snippet(tmp_path, "\t;;#ASMSTART\n\tv_cmp_ne_u32_e64 s[4:5], s6, v2\n\tv_xor_b32_e32 v3, v4, v5\n\tv_xor_b32_e32 v6, v4, v5\n"
                                "\tv_cndmask_b32_e64 v0, v1, v2, s[4:5]\n\t;;#ASMEND\n"))
    assert r.returncode == 0, r.stderr
    # the same pair wholly in compiler-scheduled code is not ours: reported as calibration, not as a failure
    r = _haz(_snippet(tmp_path, cases["H1"].replace("\t;;#ASMSTART\n", "").replace("\t;;#ASMEND\n", "")))
    assert r.returncode == 0 and "calibration" in r.stderr


def test_hazard_guard_follows_branches_inside_asm_statements(tmp_path):
    """The loops inside asm statements (`1: ... s_cbranch_scc1 1b`): the last instructions of the body are
    predecessors of the first ones, and a forward branch that is taken skips the instructions between."""
    loop_h10 = ("\t;;#ASMSTART\n1:\n\tv_cndmask_b32_e64 v0, v1, v2, s[4:5]\n\ts_add_u32 s6, s6, 1\n\ts_cmp_lt_u32 s6, 8\n"
                "\tv_cmp_ne_u32_e64 s[4:5], s6, v2\n\ts_cbranch_scc1 1b\n\t;;#ASMEND\n")
    r = _haz(_snippet(tmp_path, loop_h10))
    assert r.returncode == 1 and "H10" in r.stderr and "across the branch" in r.stderr, r.stderr
    # the compare one instruction earlier in the body: two wait states (the s_cmp, the branch) -- fine
    r = _haz(_snippet(tmp_path, loop_h10.replace("\ts_cmp_lt_u32 s6, 8\n\tv_cmp_ne_u32_e64 s[4:5], s6, v2\n",
                                                 "\tv_cmp_ne_u32_e64 s[4:5], s6, v2\n\ts_cmp_lt_u32 s6, 8\n")))
    assert r.returncode == 0, r.stderr
    # loop-carried H9's sibling with a lane select (H1, four states): the walks' shape with the select made by a VALU instruction
    loop_h1 = ("\t;;#ASMSTART\n1:\n\tv_readlane_b32 s7, v9, s4\n\ts_cmp_lt_i32 s7, 0\n\tv_readfirstlane_b32 s4, v3\n"
               "\ts_cbranch_scc1 1b\n\t;;#ASMEND\n")
    r = _haz(_snippet(tmp_path, loop_h1))
    assert r.returncode == 1 and "H1 " in r.stderr, r.stderr
    # a DPP read at the loop head of a register the loop tail's VALU instruction wrote (H3, two states)
    loop_h3 = ("\t;;#ASMSTART\n1:\n\tv_mov_b32_dpp v4, v3 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_add_u32 s6, s6, 1\n"
               "\ts_cmp_lt_u32 s6, 8\n\tv_add_u32_e32 v3, v3, v4\n\ts_cbranch_scc1 1b\n\t;;#ASMEND\n")
    r = _haz(_snippet(tmp_path, loop_h3))
    assert r.returncode == 1 and "H3" in r.stderr, r.stderr
    # forward: the fall-through path has its s_nop, the taken path jumps past it
    fwd = ("\t;;#ASMSTART\n\tv_readfirstlane_b32 s4, v1\n\ts_cbranch_scc0 2f\n\ts_nop 3\n2:\n\tv_readlane_b32 s5, v2, s4\n\t;;#ASMEND\n")
    r = _haz(_snippet(tmp_path, fwd))
    assert r.returncode == 1 and "H1 " in r.stderr and "across the branch" in r.stderr, r.stderr
    r = _haz(_snippet(tmp_path, fwd.replace("\ts_cbranch_scc0 2f\n\ts_nop 3\n", "\ts_nop 3\n\ts_cbranch_scc0 2f\n")))
    assert r.returncode == 0, r.stderr
    # a compiler label as the target (.LBB...), producer in compiler code, consumer in an asm statement
    r = _haz(_snippet(tmp_path, ".LBB0_1:\n\t;;#ASMSTART\n\tv_readlane_b32 s7, v9, 0\n\t;;#ASMEND\n\tv_or_b32_e32 v9, v1, v2\n\ts_cbranch_vccnz .LBB0_1\n"))
    assert r.returncode == 0, r.stderr   # (the branch is the one state H9 asks for)
    r = _haz(_snippet(tmp_path, ".LBB0_1:\n\t;;#ASMSTART\n\tv_cndmask_b32_e64 v0, v1, v2, s[4:5]\n\t;;#ASMEND\n\tv_cmp_ne_u32_e64 s[4:5], s6, v2\n\ts_cbranch_vccnz .LBB0_1\n"))
    assert r.returncode == 1 and "H10" in r.stderr, r.stderr


# ---- the library that ships has no knobs; the knobs build is the same device code ------------------
def test_product_library_reads_nothing_from_the_environment_and_knobs_build_is_the_same_device_code():
    """lib/libhipcomp.so: no HIPCOMP_* string, no getenv import.  lib/libhipcomp_knobs.so (csrc/Makefile
    VARIANT=knobs: -DHC_MEASUREMENT_KNOBS, read by the tests that force a launch shape): the device
    assembly of every kernel object is the product's, line for line (the macro touches host code only;
    hipcc names a compilation unit after a hash of its command line: __hip_cuid_<hash>)."""
    import glob
    lib = os.path.join(ROOT, "hipcomp-core_amd", "lib", "libhipcomp.so")
    knobs = os.path.join(ROOT, "hipcomp-core_amd", "lib", "libhipcomp_knobs.so")
    assert os.path.exists(lib) and os.path.exists(knobs), "run build()"
    blob = open(lib, "rb").read()
    assert b"HIPCOMP_LZ4" not in blob and b"HIPCOMP_CASCADED" not in blob
    und = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert "getenv" not in und, und
    assert b"HIPCOMP_LZ4_SHAPE" in open(knobs, "rb").read()

    def device_code(path):
        out = []
        for line in open(path):
            line = line.split(";", 1)[0].rstrip()
            if line:
                out.append(re.sub(r"__hip_cuid_[0-9a-f]+", "__hip_cuid_X", line))
        return out
    files = sorted(glob.glob(os.path.join(CSRC, "build", "*.gfx950.s")))
    assert len(files) >= 5
    for f in files:
        k = os.path.join(CSRC, "build_knobs", os.path.basename(f))
        assert os.path.exists(k), k
        assert device_code(f) == device_code(k), os.path.basename(f)


def test_committed_profiles_belong_to_the_build_at_hand():
    """bench.py attaches the committed rocprof numbers (profiles/rNN_rows.json) to its rows only when they were
    taken on THIS build of the kernels: the same kernel source files, or -- host code or comments of such a file
    changed since -- the same device assembly (csrc/build/*.gfx950.s, which build() leaves there)."""
    import glob
    import json
    sys.path.insert(0, ROOT)
    import bench
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_rows.json")), reverse=True)
    assert tables, "no profiles/rNN_rows.json"
    newest = json.load(open(tables[0]))
    asm = bench.device_asm_id()
    assert asm is not None, "csrc/build/*.gfx950.s missing: run __graft_entry__.build()"
    assert newest.get("kernel_source_sha16") == bench.kernel_source_id() or newest.get("device_asm_sha16") == asm, (
        f"{tables[0]} was measured on other kernels (sources {newest.get('kernel_source_sha16')}, assembly "
        f"{newest.get('device_asm_sha16')}; here {bench.kernel_source_id()}, {asm}): collect the rows again "
        "(scripts/collect_profiles.sh) or bench.py reports roofline.traffic = null")
    bench.PROFILED_ROWS = None
    assert bench.profiled("lz4/uniform/char/100000", "compress") is not None
