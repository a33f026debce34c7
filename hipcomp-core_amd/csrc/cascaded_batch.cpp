// cascaded_batch.cpp -- C ABI of the batched Cascaded codec
// (include/hipcomp/cascaded.h).
//
// Host-side mirror of the reference's src/lowlevel/CascadedBatch.hip:306-462.
// Differences, deliberate: null output pointers of the size queries and null
// array arguments return hipcompErrorInvalidValue (the reference dereferences
// them); option sets whose chunk metadata would not fit the format's 64-byte
// image (num_RLEs/num_deltas too large) are rejected up front instead of
// tripping a device-side assert.
#include "hipcomp/cascaded.h"

#include "cascaded_launch.hpp"
#include "host_common.hpp"


using namespace hcamd;

namespace {

bool elem_size_of(hipcompType_t t, int& s)
{
  switch (t) { // reference type_macros.h:219-249 (HIPCOMP_TYPE_ONE_SWITCH)
  case HIPCOMP_TYPE_CHAR:
  case HIPCOMP_TYPE_UCHAR:
    s = 1;
    return true;
  case HIPCOMP_TYPE_SHORT:
  case HIPCOMP_TYPE_USHORT:
    s = 2;
    return true;
  case HIPCOMP_TYPE_INT:
  case HIPCOMP_TYPE_UINT:
    s = 4;
    return true;
  case HIPCOMP_TYPE_LONGLONG:
  case HIPCOMP_TYPE_ULONGLONG:
    s = 8;
    return true;
  default:
    return false;
  }
}

} // namespace

extern "C" {

hipcompStatus_t hipcompBatchedCascadedCompressGetTempSize(
    size_t /*batch_size*/, size_t /*max_uncompressed_chunk_bytes*/,
    hipcompBatchedCascadedOpts_t /*format_opts*/, size_t* temp_bytes)
{
  static const char* fn = "hipcompBatchedCascadedCompressGetTempSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, temp_bytes);
  *temp_bytes = 0;
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedCascadedCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes,
    hipcompBatchedCascadedOpts_t /*format_opts*/, size_t* max_compressed_bytes)
{
  static const char* fn = "hipcompBatchedCascadedCompressGetMaxOutputChunkSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, max_compressed_bytes);
  *max_compressed_bytes = round_up_to(max_uncompressed_chunk_bytes, 4) + 8;
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedCascadedCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t /*max_uncompressed_chunk_bytes*/, size_t batch_size,
    void* /*device_temp_ptr*/, size_t /*temp_bytes*/,
    void* const* device_compressed_ptrs, size_t* device_compressed_bytes,
    const hipcompBatchedCascadedOpts_t format_opts, hipStream_t stream)
{
  static const char* fn = "hipcompBatchedCascadedCompressAsync()";
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_ptrs);
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_ptrs);
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_bytes);
  int s = 0;
  if (!elem_size_of(format_opts.type, s))
    return fail(fn, "Unknown type: " + std::to_string((int)format_opts.type));
  const int R = format_opts.num_RLEs, D = format_opts.num_deltas;
  if (R < 0 || D < 0 || R > 255 || D > 255
      || round_up_to(4 + 4 * (size_t)(R + 1), s) + round_up_to((size_t)s * D, 4) > 64)
    return fail(fn, "num_RLEs / num_deltas do not fit the 64-byte chunk metadata");
  if (batch_size == 0)
    return hipcompSuccess;
  cascaded_launch_compress(
      reinterpret_cast<const uint8_t* const*>(device_uncompressed_ptrs),
      device_uncompressed_bytes,
      reinterpret_cast<uint8_t* const*>(device_compressed_ptrs),
      device_compressed_bytes, batch_size, (int)format_opts.type, s, R, D,
      format_opts.use_bp ? 1 : 0,
      // (format_opts.chunk_size: the reference ignores the field -- cascaded.h:93-100 -- and cuts partitions
      // into 4096-byte sub-chunks whatever it says, and so does this)
      stream);
  std::string why;
  if (!launch_ok("cascaded compression kernel", why))
    return fail(fn, why);
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedCascadedDecompressGetTempSize(
    size_t /*num_chunks*/, size_t /*max_uncompressed_chunk_bytes*/, size_t* temp_bytes)
{
  static const char* fn = "hipcompBatchedCascadedDecompressGetTempSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, temp_bytes);
  *temp_bytes = 0;
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedCascadedDecompressAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    const size_t* device_uncompressed_bytes,
    size_t* device_actual_uncompressed_bytes, size_t batch_size,
    void* const /*device_temp_ptr*/, size_t /*temp_bytes*/,
    void* const* device_uncompressed_ptrs, hipcompStatus_t* device_statuses,
    hipStream_t stream)
{
  static const char* fn = "hipcompBatchedCascadedDecompressAsync()";
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_ptrs);
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_ptrs);
  // required by the reference too (cascaded.h:262-271: not nullable)
  HCAMD_REQUIRE_NOT_NULL(fn, device_actual_uncompressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_statuses);
  if (batch_size == 0)
    return hipcompSuccess;
  const hipError_t e = cascaded_launch_decompress(
      reinterpret_cast<const uint8_t* const*>(device_compressed_ptrs),
      device_compressed_bytes, device_uncompressed_bytes, batch_size,
      reinterpret_cast<uint8_t* const*>(device_uncompressed_ptrs),
      device_actual_uncompressed_bytes, device_statuses, stream);
  if (e != hipSuccess)
    return fail(fn, std::string("cascaded decompress launch: ") + hipGetErrorString(e));
  std::string why;
  if (!launch_ok("cascaded decompression kernel", why))
    return fail(fn, why);
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedCascadedGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes, size_t* device_uncompressed_bytes,
    size_t batch_size, hipStream_t stream)
{
  static const char* fn = "hipcompBatchedCascadedGetDecompressSizeAsync()";
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_ptrs);
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_bytes);
  if (batch_size == 0)
    return hipcompSuccess;
  cascaded_launch_get_sizes(
      reinterpret_cast<const uint8_t* const*>(device_compressed_ptrs),
      device_compressed_bytes, device_uncompressed_bytes, batch_size, stream);
  std::string why;
  if (!launch_ok("cascaded get-size kernel", why))
    return fail(fn, why);
  return hipcompSuccess;
}

} // extern "C"
