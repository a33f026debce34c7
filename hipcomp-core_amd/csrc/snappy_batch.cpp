// snappy_batch.cpp -- C ABI of the batched Snappy codec (include/hipcomp/snappy.h).
//
// Host-side mirror of the reference's src/lowlevel/SnappyBatch.cpp:84-245:
// same argument meaning and status codes; the pointer arguments are checked for
// null exactly where the reference checks them (it does not call
// hipPointerGetAttributes on this path).
#include "hipcomp/snappy.h"

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
