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
    assert os.path.getmtime(obj) >= os.path.getmtime(ASM) - 1.0   # same make rule, same compile
    r = _run(ASM)
    assert r.returncode == 0, r.stderr
    text = open(ASM).read()
    agprs = dict(re.findall(r"\.set (\S*lz4_compress_kernel_\S*)\.num_agpr, (\d+)", text))
    assert {v for k, v in agprs.items() if "kernel_mix" in k} == {"24"}
    assert {v for k, v in agprs.items() if "kernel_mix" not in k} == {"0"}


def test_makefile_runs_the_guard_on_the_object_it_ships():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    rule = mk[mk.index("$(OBJDIR)/lz4_kernels.hip.o:"):]
    assert "-save-temps=obj" in rule and "check_lz4_registers.py" in rule
    assert rule.index("check_lz4_registers.py $(OBJDIR)") < rule.index("mv $(OBJDIR)/lz4_temps/lz4_kernels.hip.o")
    assert "$(CXXFLAGS)" in rule.split("\n")[2]                   # the flags of every other object, EXTRA included


def test_guard_catches_compiler_use_of_accumulation_registers(tmp_path):
    text = open(ASM).read()
    # (1) a compiler-made AGPR access outside any inline-asm block of a mix kernel
    k = re.search(r"^(_ZN5hcamd\S*lz4_compress_kernel_mixILi1E\S*):", text, re.M)
    at = text.index("\n", k.end()) + 1
    for doctored, what in (
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
