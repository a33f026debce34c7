// lz4_batch.cpp -- C ABI of the batched LZ4 codec (include/hipcomp/lz4.h).
//
// Host-side mirror of the reference's src/lowlevel/LZ4Batch.cpp:71-224 and
// the size math of src/lowlevel/LZ4CompressionKernels.hip:142-156,287-313:
// same argument meaning, same status codes.  Differences, all deliberate:
//   * null size-query outputs return hipcompErrorInvalidValue instead of
//     letting a C++ exception escape through the C ABI (reference
//     LZ4Batch.cpp:76,160,177 call CHECK_NOT_NULL outside the try block);
//   * batch_size == 0 is a successful no-op (the reference launches a
//     zero-sized grid and reports the resulting HIP error);
//   * of device_temp_ptr the first 64 bytes hold the chunk ticket counters of
//     the persistent compress kernels, list lengths and sample totals, behind
//     them lie the routing kernel's chunk lists (16 bytes per chunk) and hash
//     tables for data that compresses (one per resident device-table wave, not
//     one per chunk: lz4_far.hiph); the other tables live in LDS
//     (lz4_launch.hpp).  temp_bytes is checked against the contract size, so
//     callers sized for the reference keep working and callers that
//     under-allocate keep failing the same way.  As with the reference, one
//     temp buffer serves one compress call at a time.
#include "hipcomp/lz4.h"

#include "host_common.hpp"
#include "lz4_launch.hpp"

using namespace hcamd;

namespace {

constexpr size_t kMaxChunk = size_t(1) << 24; // reference LZ4Kernels.hiph:174
constexpr size_t kMaxHashTable = size_t(1) << 14; // reference :151

size_t hash_table_size(size_t max_chunk)
{
  size_t p = 1;
  while (p < max_chunk)
    p *= 2;
  return p < kMaxHashTable ? p : kMaxHashTable;
}

bool elem_size_of(hipcompType_t t, int& s)
{
  switch (t) { // reference LZ4CompressionKernels.hip:185-219
  case HIPCOMP_TYPE_BITS:
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
  default:
    return false;
  }
}

} // namespace

extern "C" {

hipcompStatus_t hipcompBatchedLZ4CompressGetTempSize(
    size_t batch_size, size_t max_uncompressed_chunk_bytes,
    hipcompBatchedLZ4Opts_t /*format_opts*/, size_t* temp_bytes)
{
  static const char* fn = "hipcompBatchedLZ4CompressGetTempSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, temp_bytes);
  if (max_uncompressed_chunk_bytes > kMaxChunk)
    return fail(fn, "Maximum chunk size for LZ4 is " + std::to_string(kMaxChunk));
  *temp_bytes = hash_table_size(max_uncompressed_chunk_bytes) * sizeof(uint16_t)
                * batch_size;
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedLZ4CompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes, hipcompBatchedLZ4Opts_t /*format_opts*/,
    size_t* max_compressed_bytes)
{
  static const char* fn = "hipcompBatchedLZ4CompressGetOutputSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, max_compressed_bytes);
  if (max_uncompressed_chunk_bytes > kMaxChunk)
    return fail(fn, "Maximum chunk size for LZ4 is " + std::to_string(kMaxChunk));
  const size_t n = max_uncompressed_chunk_bytes;
  *max_compressed_bytes = round_up_to(n + 1 + round_up_div(n, 255), sizeof(size_t));
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedLZ4CompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes, size_t batch_size,
    void* device_temp_ptr, size_t temp_bytes,
    void* const* device_compressed_ptrs, size_t* device_compressed_bytes,
    hipcompBatchedLZ4Opts_t format_opts, hipStream_t stream)
{
  static const char* fn = "hipcompBatchedLZ4CompressAsync()";
  HCAMD_DEVICE_POINTER(fn, device_uncompressed_ptrs);
  HCAMD_DEVICE_POINTER(fn, device_uncompressed_bytes);
  HCAMD_DEVICE_POINTER(fn, device_compressed_ptrs);
  HCAMD_DEVICE_POINTER(fn, device_compressed_bytes);

  const size_t ht = hash_table_size(max_uncompressed_chunk_bytes);
  const size_t need = batch_size * ht * sizeof(uint16_t);
  if (temp_bytes < need)
    return fail(fn, "Insufficient temp space: got " + std::to_string(temp_bytes)
                        + " bytes, but need " + std::to_string(need) + " bytes.");
  int s = 0;
  if (!elem_size_of(format_opts.data_type, s))
    return fail(fn, "Unsupported input data type");
  if (batch_size == 0)
    return hipcompSuccess;
  if (batch_size > 0x7FFFFFFFull) // the ticket counter runs past batch_size by up to waves x 64
    return fail(fn, "batch_size must be below 2^31");
  HCAMD_DEVICE_POINTER(fn, device_temp_ptr);
  const hipError_t e = lz4_launch_compress(
      reinterpret_cast<const uint8_t* const*>(device_uncompressed_ptrs),
      device_uncompressed_bytes,
      reinterpret_cast<uint8_t* const*>(device_compressed_ptrs),
      device_compressed_bytes, (uint32_t)ht, batch_size, s, device_temp_ptr, temp_bytes,
      max_uncompressed_chunk_bytes,
      lz4_mode_from_environment(), stream);
  if (e != hipSuccess)
    return fail(fn, std::string("lz4 compress launch: ") + hipGetErrorString(e));
  std::string why;
  if (!launch_ok("lz4 compress kernel", why))
    return fail(fn, why);
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedLZ4DecompressGetTempSize(
    size_t num_chunks, size_t /*max_uncompressed_chunk_bytes*/, size_t* temp_bytes)
{
  static const char* fn = "hipcompBatchedLZ4DecompressGetTempSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, temp_bytes);
  // reference: sizeof(chunk_header) = 24 bytes per chunk, rounded to 8
  *temp_bytes = round_up_to(24 * num_chunks, sizeof(size_t));
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedLZ4DecompressAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    const size_t* device_uncompressed_bytes,
    size_t* device_actual_uncompressed_bytes, size_t batch_size,
    void* const device_temp_ptr, size_t temp_bytes,
    void* const* device_uncompressed_ptrs, hipcompStatus_t* device_statuses,
    hipStream_t stream)
{
  static const char* fn = "hipcompBatchedLZ4DecompressAsync()";
  HCAMD_DEVICE_POINTER(fn, device_compressed_ptrs);
  HCAMD_DEVICE_POINTER(fn, device_compressed_bytes);
  HCAMD_DEVICE_POINTER(fn, device_uncompressed_bytes);
  // the reference validates the (unused) temp pointer too: LZ4Batch.cpp:112
  void* temp = device_temp_ptr;
  HCAMD_DEVICE_POINTER(fn, temp);
  HCAMD_DEVICE_POINTER(fn, device_uncompressed_ptrs);
  if (device_actual_uncompressed_bytes)
    HCAMD_DEVICE_POINTER(fn, device_actual_uncompressed_bytes);
  if (device_statuses)
    HCAMD_DEVICE_POINTER(fn, device_statuses);
  if (batch_size == 0)
    return hipcompSuccess;

  if (lz4_launch_decompress(
          reinterpret_cast<const uint8_t* const*>(device_compressed_ptrs),
          device_compressed_bytes, device_uncompressed_bytes, batch_size,
          reinterpret_cast<uint8_t* const*>(device_uncompressed_ptrs),
          device_actual_uncompressed_bytes, device_statuses, true, stream, device_temp_ptr, temp_bytes)
      != hipSuccess)
    return fail(fn, "could not zero the chunk ticket counter in the temp buffer");
  std::string why;
  if (!launch_ok("lz4 decompress kernel", why))
    return fail(fn, why);
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedLZ4GetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes, size_t* device_uncompressed_bytes,
    size_t batch_size, hipStream_t stream)
{
  static const char* fn = "hipcompBatchedLZ4GetDecompressSizeAsync()";
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_ptrs);
  HCAMD_REQUIRE_NOT_NULL(fn, device_compressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_bytes);
  HCAMD_DEVICE_POINTER(fn, device_compressed_ptrs);
  HCAMD_DEVICE_POINTER(fn, device_compressed_bytes);
  HCAMD_DEVICE_POINTER(fn, device_uncompressed_bytes);
  if (batch_size == 0)
    return hipcompSuccess;

  (void)lz4_launch_decompress( // (no temp buffer in this call: one wave per chunk by position, nothing to fail)
      reinterpret_cast<const uint8_t* const*>(device_compressed_ptrs),
      device_compressed_bytes, nullptr, batch_size, nullptr,
      device_uncompressed_bytes, nullptr, false, stream);
  std::string why;
  if (!launch_ok("lz4 decompress-size kernel", why))
    return fail(fn, why);
  return hipcompSuccess;
}

} // extern "C"
