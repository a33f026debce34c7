"""CPU-side checks of the drop-in boundary: the shared library loads, exports
every symbol include/hipcomp/*.h declares (and nothing else), and the pure
host size queries follow the reference's formulas."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in ("lz4.h", "snappy.h", "cascaded.h"):
        text = open(os.path.join(ROOT, "include", "hipcomp", h)).read()
        names |= set(re.findall(r"hipcompStatus_t\s+(hipcompBatched\w+)\s*\(", text))
    return names


def _declared_primitives():
    text = open(os.path.join(ROOT, "include", "hipcomp", "primitives.h")).read()
    return set(re.findall(r"hipcompStatus_t\s+(hipcomp\w+)\s*\(", text))


def _declared_hlif():
    text = open(os.path.join(ROOT, "include", "hipcomp", "hlif.h")).read()
    return set(re.findall(r"hipcompStatus_t\s+(hipcompHlif\w+)\s*\(", text))


def _declared_interop():
    text = open(os.path.join(ROOT, "include", "hipcomp", "lz4_interop.h")).read()
    return set(re.findall(r"(?:hipcompStatus_t|size_t)\s+(hipcompLZ4Frame\w+)\s*\(", text))


def test_library_exports_exactly_the_declared_abi(hc):
    lib = hc.default_library()
    out = subprocess.run(["nm", "-D", "--defined-only", "-C", lib.path], capture_output=True, text=True, check=True).stdout
    exported = {ln.split(" T ", 1)[1].strip() for ln in out.splitlines() if " T " in ln}
    declared = _declared_symbols()
    built = {s for s in declared if any(s.startswith(f"hipcompBatched{c}") for c in lib.codecs)}
    assert built <= exported, sorted(built - exported)
    assert len(declared) == 18
    prims = _declared_primitives()
    assert len(prims) == 7 and prims <= exported
    # the C++ classes of hipcomp/primitives.hpp (reference src/{RunLengthEncodeGPU,DeltaGPU,BitPackGPU}.h)
    classes = {e.split("(")[0] for e in exported if e.startswith("hipcomp::")}
    managers = ("LZ4Manager", "SnappyManager", "CascadedManager")
    hlif_classes = {c for c in classes if c.split("::")[1] in managers + ("CompressionConfig", "DecompressionConfig", "create_manager")}
    for m in managers:                                    # the reference's manager surface, per format
        assert {f"hipcomp::{m}::{f}" for f in (m, "compress", "decompress", "configure_compression", "configure_decompression",
                                                 "get_compressed_output_size", "set_scratch_buffer",
                                                 "get_required_scratch_buffer_size")} <= hlif_classes
    assert {"hipcomp::CompressionConfig::get_status", "hipcomp::DecompressionConfig::get_status",
            "hipcomp::create_manager"} <= hlif_classes
    classes -= hlif_classes
    # the layer below the batched Snappy API (include/hipcomp/snappy_kernels.hpp; reference src/lowlevel/SnappyBatchKernels.h)
    snappy_layer = {"hipcomp::gpu_snap", "hipcomp::gpu_unsnap", "hipcomp::gpu_get_uncompressed_sizes"}
    header = open(os.path.join(ROOT, "include", "hipcomp", "snappy_kernels.hpp")).read()
    assert snappy_layer <= classes and all(n.split("::")[1] + "(" in header for n in snappy_layer)
    classes -= snappy_layer
    assert classes == {"hipcomp::RunLengthEncodeGPU::compress", "hipcomp::RunLengthEncodeGPU::compressDownstream",
                       "hipcomp::RunLengthEncodeGPU::requiredWorkspaceSize", "hipcomp::DeltaGPU::compress",
                       "hipcomp::DeltaGPU::requiredWorkspaceSize", "hipcomp::BitPackGPU::compress",
                       "hipcomp::BitPackGPU::requiredWorkspaceSize"}
    interop = _declared_interop()
    assert len(interop) == 3 and interop <= exported
    hlif = _declared_hlif()
    assert len(hlif) == 13 and hlif <= exported
    # this library's own extension of the batched Cascaded API (include/hipcomp/cascaded_select.h; the header says so)
    select = set(re.findall(r"hipcompStatus_t\s+(hipcompBatchedCascadedSelect\w+)\s*\(",
                            open(os.path.join(ROOT, "include", "hipcomp", "cascaded_select.h")).read()))
    assert select == {"hipcompBatchedCascadedSelectOpts", "hipcompBatchedCascadedSelectOptsGetTempSize"} and select <= exported
    others = {e for e in exported if not e.startswith("hipcomp::") and not e.startswith(("vtable for", "typeinfo"))}
    assert others <= declared | prims | interop | hlif | select, sorted(others - declared - prims - interop - hlif - select)


def test_headers_compile_as_c(tmp_path):
    """The public headers are C-clean (the reference proves this with its C harness)."""
    src = tmp_path / "t.c"
    src.write_text('#include "hipcomp/lz4.h"\n#include "hipcomp/snappy.h"\n#include "hipcomp/cascaded.h"\n#include "hipcomp/primitives.h"\n#include "hipcomp/lz4_interop.h"\n#include "hipcomp/hlif.h"\n'
                   "int main(void){hipcompBatchedLZ4Opts_t o = hipcompBatchedLZ4DefaultOpts;"
                   "hipcompBatchedCascadedOpts_t c = hipcompBatchedCascadedDefaultOpts;"
                   "hipcompBatchedSnappyOpts_t s = hipcompBatchedSnappyDefaultOpts;"
                   "return (int)o.data_type + c.num_RLEs - 2 + s.reserved;}\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                        "-I", "/opt/rocm/include", "-c", str(src), "-o", str(tmp_path / "t.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("n", [0, 1, 100, 255, 256, 65535, 65536, 1 << 20, 1 << 24])
def test_lz4_size_queries(hc, oracle, n):
    lib = hc.default_library()
    opts = hc.LZ4Opts(hc.hipcompType.CHAR)
    assert lib.max_output_chunk_size("LZ4", n, opts) == (n + 1 + (n + 254) // 255 + 7) // 8 * 8
    assert lib.max_output_chunk_size("LZ4", n, opts) == oracle.lz4_max_compressed_size(n)
    ht = 1
    while ht < n:
        ht *= 2
    ht = min(ht, 16384)
    assert oracle.lz4_hash_table_size(n) == ht
    for batch in (1, 7, 100000):
        assert lib.compress_temp_size("LZ4", batch, n, opts) == ht * 2 * batch
        assert lib.decompress_temp_size("LZ4", batch, n) == (24 * batch + 7) // 8 * 8


def test_lz4_size_query_errors(hc):
    lib = hc.default_library()
    out = ctypes.c_size_t(0)
    opts = hc.LZ4Opts(0)
    assert lib.hipcompBatchedLZ4CompressGetTempSize(1, (1 << 24) + 1, opts, ctypes.byref(out)) == 10
    assert lib.hipcompBatchedLZ4CompressGetMaxOutputChunkSize((1 << 24) + 1, opts, ctypes.byref(out)) == 10
    assert lib.hipcompBatchedLZ4CompressGetTempSize(1, 100, opts, None) == 10
    assert lib.hipcompBatchedLZ4CompressGetMaxOutputChunkSize(100, opts, None) == 10
    assert lib.hipcompBatchedLZ4DecompressGetTempSize(1, 100, None) == 10
    assert lib.hipcompBatchedLZ4GetDecompressSizeAsync(None, None, None, 1, None) == 10
    # 65536-byte chunks: the numbers SURVEY.md 8(a) L1/L2 quote
    assert lib.compress_temp_size("LZ4", 100000, 65536, opts) == 3276800000
    assert lib.max_output_chunk_size("LZ4", 65536, opts) == 65800


@pytest.mark.parametrize("n", [0, 1, 5, 6, 65536, 1 << 24])
def test_snappy_size_queries(hc, oracle, n):
    lib = hc.default_library()
    opts = hc.SnappyOpts(0)
    assert lib.max_output_chunk_size("Snappy", n, opts) == 32 + n + n // 6 == oracle.snappy_max_compressed_size(n)
    assert lib.compress_temp_size("Snappy", 1000, n, opts) == 0
    assert lib.decompress_temp_size("Snappy", 1000, n) == 0
    assert lib.max_output_chunk_size("Snappy", 65536, opts) == 76490 or n != 65536


def test_snappy_null_arguments(hc):
    lib = hc.default_library()
    opts = hc.SnappyOpts(0)
    assert lib.hipcompBatchedSnappyCompressGetTempSize(1, 1, opts, None) == 10
    assert lib.hipcompBatchedSnappyCompressGetMaxOutputChunkSize(1, opts, None) == 10
    assert lib.hipcompBatchedSnappyDecompressGetTempSize(1, 1, None) == 10
    assert lib.hipcompBatchedSnappyCompressAsync(None, None, 0, 1, None, 0, None, None, opts, None) == 10
    assert lib.hipcompBatchedSnappyDecompressAsync(None, None, None, None, 1, None, 0, None, None, None) == 10
    assert lib.hipcompBatchedSnappyGetDecompressSizeAsync(None, None, None, 1, None) == 10


@pytest.mark.parametrize("n", [0, 1, 3, 4, 65536, 65537])
def test_cascaded_size_queries(hc, oracle, n):
    lib = hc.default_library()
    opts = hc.CascadedOpts(4096, hc.hipcompType.UINT, 2, 1, 1)
    assert lib.max_output_chunk_size("Cascaded", n, opts) == (n + 3) // 4 * 4 + 8 == oracle.cascaded_max_compressed_size(n)
    assert lib.compress_temp_size("Cascaded", 10, n, opts) == 0
    assert lib.decompress_temp_size("Cascaded", 10, n) == 0


def test_cascaded_null_arguments(hc):
    lib = hc.default_library()
    opts = hc.CascadedOpts(4096, 5, 2, 1, 1)
    assert lib.hipcompBatchedCascadedCompressGetTempSize(1, 1, opts, None) == 10
    assert lib.hipcompBatchedCascadedCompressGetMaxOutputChunkSize(1, opts, None) == 10
    assert lib.hipcompBatchedCascadedDecompressGetTempSize(1, 1, None) == 10
    assert lib.hipcompBatchedCascadedCompressAsync(None, None, 0, 1, None, 0, None, None, opts, None) == 10
    assert lib.hipcompBatchedCascadedGetDecompressSizeAsync(None, None, None, 1, None) == 10
    assert lib.max_output_chunk_size("Cascaded", 65536, opts) == 65544
