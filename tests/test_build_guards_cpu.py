"""Build-time guards of the LZ4 encoder (no GPU): its walk keeps loads in flight
in accumulation registers a0..a23 that it names in inline asm (lz4_kernels.hip,
HC_WALK_AGPRS).  That is only sound while the compiler itself never writes an
AGPR in those kernels (it would, to spill vector registers): the kernels'
device assembly must hold no v_accvgpr_write and declare exactly those 24."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.timeout(600)
def test_compiler_leaves_the_accumulation_registers_alone(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path / "lz4_kernels.s"
    src = os.path.join(ROOT, "hipcomp-core_amd", "csrc", "lz4_kernels.hip")
    subprocess.run([HIPCC, "-std=c++17", "-O3", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "hipcomp-core_amd", "csrc"), "--offload-arch=gfx950", "-S",
                    "--cuda-device-only", src, "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    assert "v_accvgpr_write" not in text
    assert "_d16" not in text                    # table entries are read zero-extended (walk_probe relies on it)
    agprs = dict(re.findall(r"\.set (\S*lz4_compress_kernel_\S*)\.num_agpr, (\d+)", text))
    vgprs = dict(re.findall(r"\.set (\S*lz4_compress_kernel_\S*)\.num_vgpr, (\d+)", text))
    mix = [k for k in agprs if "kernel_mix" in k]
    far = [k for k in agprs if "kernel_far" in k]
    assert len(mix) == 3 and len(far) == 6       # element size 1, 2, 4 (far: lean and wide form)
    assert {agprs[k] for k in mix} == {"24"}     # the walk's own, nothing of the compiler's
    assert {agprs[k] for k in far} == {"0"}      # no walk
    assert all(int(vgprs[k]) <= 64 for k in far)     # 32 waves per CU = eight on one SIMD
    assert all(int(vgprs[k]) <= 256 for k in mix)
    spills = re.findall(r"\.set \S*lz4_compress_kernel_\S*\.private_seg_size, (\d+)", text)
    assert spills and set(spills) == {"0"}       # nothing spilled to scratch memory either
