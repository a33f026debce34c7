// snappy_batch.cpp -- C ABI of the batched Snappy codec (include/hipcomp/snappy.h).
//
// Host-side mirror of the reference's src/lowlevel/SnappyBatch.cpp:84-245:
// same argument meaning and status codes; the pointer arguments are checked for
// null exactly where the reference checks them (it does not call
// hipPointerGetAttributes on this path).
#include "hipcomp/snappy.h"
#include "hipcomp/snappy_kernels.hpp"

#include "host_common.hpp"
#include "snappy_launch.hpp"

using namespace hcamd;

extern "C" {

hipcompStatus_t hipcompBatchedSnappyDecompressGetTempSize(
    size_t /*num_chunks*/, size_t /*max_uncompressed_chunk_size*/, size_t* temp_bytes)
{
  static const char* fn = "hipcompBatchedSnappyDecompressGetTempSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, temp_bytes);
  *temp_bytes = 0;
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedSnappyGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes, size_t* device_uncompressed_bytes,
    size_t batch_size, hipStream_t stream)
{
  static const char* fn = "hipcompBatchedSnappyGetDecompressSizeAsync()";
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_ptrs);
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_bytes);
  if (batch_size == 0)
    return hipcompSuccess;
  snappy_launch_get_sizes(
      reinterpret_cast<const uint8_t* const*>(device_compressed_ptrs),
      device_compressed_bytes, device_uncompressed_bytes, batch_size, stream);
  std::string why;
  if (!launch_ok("Failed to run Snappy kernel gpu_get_uncompressed_sizes", why))
    return fail(fn, why);
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedSnappyDecompressAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    const size_t* device_uncompressed_bytes,
    size_t* device_actual_uncompressed_bytes, size_t batch_size,
    void* const /*device_temp_ptr*/, const size_t /*temp_bytes*/,
    void* const* device_uncompressed_ptrs, hipcompStatus_t* device_statuses,
    hipStream_t stream)
{
  static const char* fn = "hipcompBatchedSnappyDecompressAsync()";
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_ptrs);
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_ptrs);
  if (batch_size == 0)
    return hipcompSuccess;
  snappy_launch_decompress(
      reinterpret_cast<const uint8_t* const*>(device_compressed_ptrs),
      device_compressed_bytes, device_uncompressed_bytes, batch_size,
      reinterpret_cast<uint8_t* const*>(device_uncompressed_ptrs),
      device_actual_uncompressed_bytes, device_statuses, stream);
  std::string why;
  if (!launch_ok("Failed to launch Snappy decompression HIP kernel", why))
    return fail(fn, why);
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedSnappyCompressGetTempSize(
    size_t /*batch_size*/, size_t /*max_chunk_size*/,
    hipcompBatchedSnappyOpts_t /*format_opts*/, size_t* temp_bytes)
{
  static const char* fn = "hipcompBatchedSnappyCompressGetTempSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, temp_bytes);
  *temp_bytes = 0;
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedSnappyCompressGetMaxOutputChunkSize(
    size_t max_chunk_size, hipcompBatchedSnappyOpts_t /*format_opts*/,
    size_t* max_compressed_size)
{
  static const char* fn = "hipcompBatchedSnappyCompressGetOutputSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, max_compressed_size);
  *max_compressed_size = 32 + max_chunk_size + max_chunk_size / 6;
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedSnappyCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t /*max_uncompressed_chunk_bytes*/, size_t batch_size,
    void* /*device_temp_ptr*/, size_t /*temp_bytes*/,
    void* const* device_compressed_ptrs, size_t* device_compressed_bytes,
    hipcompBatchedSnappyOpts_t /*format_opts*/, hipStream_t stream)
{
  static const char* fn = "hipcompBatchedSnappyCompressAsync()";
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_ptrs);
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_ptrs);
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_bytes);
  if (batch_size == 0) // the reference skips the launch too (SnappyBatchKernels.hip:172)
    return hipcompSuccess;
  snappy_launch_compress(
      reinterpret_cast<const uint8_t* const*>(device_uncompressed_ptrs),
      device_uncompressed_bytes,
      reinterpret_cast<uint8_t* const*>(device_compressed_ptrs),
      device_compressed_bytes, batch_size, stream);
  std::string why;
  if (!launch_ok("Failed to launch Snappy compression HIP kernel", why))
    return fail(fn, why);
  return hipcompSuccess;
}

} // extern "C"

// ---------------------------------------------------------------------------
// The reference's INTERNAL Snappy entry points (src/lowlevel/SnappyBatchKernels.h:80-148), the layer its
// batched C functions call and its own tests/test_snappy_app.cpp and src/test/SnappyLargeTokens_test.cpp
// are written against: exported under the same C++ names so that those programs compile unchanged (with
// the reference's header, used in place) and link with this library (oracle/Makefile: _ref/cpp_unit_Snappy*).
// Thin: the kernels are the batched API's.  Nothing is launched for count <= 0 (as there, :172, :192).
// ---------------------------------------------------------------------------
namespace hipcomp {

void gpu_snap(
    const void* const* device_in_ptr, const size_t* device_in_bytes, void* const* device_out_ptr,
    const size_t* device_out_available_bytes, gpu_snappy_status_s* outputs, size_t* device_out_bytes, int count,
    hipStream_t stream)
{
  if (count <= 0)
    return;
  hcamd::snappy_launch_compress(
      reinterpret_cast<const uint8_t* const*>(device_in_ptr), device_in_bytes,
      reinterpret_cast<uint8_t* const*>(device_out_ptr), device_out_bytes, (size_t)count, stream,
      device_out_available_bytes, reinterpret_cast<uint32_t*>(outputs));
}

void gpu_unsnap(
    const void* const* device_in_ptr, const size_t* device_in_bytes, void* const* device_out_ptr,
    const size_t* device_out_available_bytes, hipcompStatus_t* outputs, size_t* device_out_bytes, int count,
    hipStream_t stream)
{
  if (count <= 0)
    return;
  hcamd::snappy_launch_decompress(
      reinterpret_cast<const uint8_t* const*>(device_in_ptr), device_in_bytes, device_out_available_bytes,
      (size_t)count, reinterpret_cast<uint8_t* const*>(device_out_ptr), device_out_bytes, outputs, stream);
}

void gpu_get_uncompressed_sizes(
    const void* const* device_in_ptr, const size_t* device_in_bytes, size_t* device_out_bytes, int count,
    hipStream_t stream)
{
  if (count <= 0)
    return;
  hcamd::snappy_launch_get_sizes(
      reinterpret_cast<const uint8_t* const*>(device_in_ptr), device_in_bytes, device_out_bytes, (size_t)count, stream);
}

} // namespace hipcomp
