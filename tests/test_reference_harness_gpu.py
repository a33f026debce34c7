"""The reference's OWN C harness against the product library: reference
tests/test_{lz4batch,snappy_batch,cascadedbatch}_c_api.c (five-line stubs over
tests/test_batch_c_api.h:225-790 -- six batches of 1 .. 10 025 chunks, the nullptr
forms of the API, CRASH_SAFE decompression of raw input) compiled as C99 from where
they lie against THIS repo's include/ and linked with hipcomp-core_amd/lib/libhipcomp.so
(oracle/Makefile, binaries under the git-ignored oracle/_ref/).  Each runs as a child
process and must exit 0: the drop-in claim of INTEGRATION.md, proved with the caller
the reference ships."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["lz4batch", "snappy_batch", "cascadedbatch"])
def test_reference_c_harness_passes_on_the_product_library(cuda, name):
    exe = os.path.join(ROOT, "oracle", "_ref", "harness_" + name)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} absent (built by oracle/Makefile where /root/reference exists)")
    # the binary must have bound the product library, not another libhipcomp.so
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    bound = [ln for ln in ldd.splitlines() if "libhipcomp.so" in ln]
    assert bound and os.path.realpath(bound[0].split("=>")[1].split("(")[0].strip()) == os.path.realpath(
        os.path.join(ROOT, "hipcomp-core_amd", "lib", "libhipcomp.so")), ldd
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "failed" not in r.stdout.lower(), r.stdout[-2000:]


@pytest.mark.parametrize("name", ["test_cascaded_batch", "test_lz4", "test_random_lz4", "test_cascaded"])
def test_reference_cpp_tests_pass_on_the_product_library(cuda, name):
    """The reference's C++ callers: tests/test_cascaded_batch.cpp (the batched Cascaded C API with the
    wire-layout known answers of verify_compression_output, :213-379, all element types),
    tests/test_lz4.cpp and tests/test_random_lz4.cpp (LZ4Manager through hipcomp/lz4.hpp),
    tests/test_cascaded.cpp (CascadedManager through hipcomp/cascaded.hpp) -- Catch programs compiled
    UNCHANGED from where they lie against this repo's include/ and linked with the product library
    (oracle/Makefile: _ref/cpp_*).  They must exit 0 with every assertion passed."""
    exe = os.path.join(ROOT, "oracle", "_ref", "cpp_" + name)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} absent (built by oracle/Makefile where /root/reference exists)")
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    bound = [ln for ln in ldd.splitlines() if "libhipcomp.so" in ln]
    assert bound and os.path.realpath(bound[0].split("=>")[1].split("(")[0].strip()) == os.path.realpath(
        os.path.join(ROOT, "hipcomp-core_amd", "lib", "libhipcomp.so")), ldd
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "All tests passed" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("name", ["RunLengthEncodeGPU_test", "DeltaGPU_test", "BitPackGPU_test",
                                  "SnappyLargeTokens_test", "test_snappy_app"])
def test_reference_unit_tests_pass_on_the_product_library(cuda, name):
    """the reference's unit tests of its whole-array classes (src/test/*_test.cpp), compiled
    UNCHANGED from where they lie -- hipcomp.hpp from this repo's include/, the class headers they name
    (RunLengthEncodeGPU.h ...) the reference's own -- and linked with the product library
    (oracle/Makefile: _ref/cpp_unit_*): the library exports the classes those headers declare, and they
    behave as the reference's tests demand.  The same for the two programs written against the layer below
    the batched Snappy API (hipcomp::gpu_snap / gpu_unsnap): src/test/SnappyLargeTokens_test.cpp (known
    answers for literals and matches of 2^16 .. 2^24 bytes, both directions) and tests/test_snappy_app.cpp
    (the decoder's golden vectors)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "cpp_unit_" + name)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} absent (built by oracle/Makefile where /root/reference exists)")
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    bound = [ln for ln in ldd.splitlines() if "libhipcomp.so" in ln]
    assert bound and os.path.realpath(bound[0].split("=>")[1].split("(")[0].strip()) == os.path.realpath(
        os.path.join(ROOT, "hipcomp-core_amd", "lib", "libhipcomp.so")), ldd
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "All tests passed" in r.stdout, r.stdout[-2000:]
