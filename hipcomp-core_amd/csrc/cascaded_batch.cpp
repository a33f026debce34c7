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
#include "hipcomp/cascaded_select.h"

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

// ---------------------------------------------------------------------------
// hipcomp/cascaded_select.h: options picked by measuring (an API of this library's own)
// ---------------------------------------------------------------------------
namespace {

constexpr size_t kSelectParts = 64;            // sample partitions at most
constexpr size_t kSelectBytes = 16384;         // of each: its first 16 KiB
constexpr size_t kSelectSlot = kSelectBytes + 8; // max compressed size of a sample partition
struct Candidate
{
  int R, D, bp;
};
// (in the order ties are broken: the fewest layers first)
constexpr Candidate kCandidates[] = {{0, 0, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}, {2, 0, 1},
                                     {0, 2, 1}, {2, 1, 1}, {1, 2, 1}, {2, 2, 1}};
constexpr size_t kNumCandidates = sizeof(kCandidates) / sizeof(kCandidates[0]);

// temp layout: [in ptrs][in bytes][out ptrs][out bytes x candidates][totals x (candidates + 1)][sample outputs]
struct SelectLayout
{
  size_t in_ptrs, in_bytes, out_ptrs, out_bytes, totals, slots, end;
};
SelectLayout select_layout()
{
  SelectLayout l;
  l.in_ptrs = 0;
  l.in_bytes = l.in_ptrs + kSelectParts * sizeof(void*);
  l.out_ptrs = l.in_bytes + kSelectParts * sizeof(size_t);
  l.out_bytes = l.out_ptrs + kSelectParts * sizeof(void*);
  l.totals = l.out_bytes + kNumCandidates * kSelectParts * sizeof(size_t);
  l.slots = round_up_to(l.totals + (kNumCandidates + 1) * sizeof(unsigned long long), 16);
  l.end = l.slots + kSelectParts * kSelectSlot;
  return l;
}

} // namespace

namespace hcamd {
// (cascaded_kernels.hip)
void cascaded_launch_select_sample(
    const uint8_t* const* in_ptrs, const size_t* in_bytes, size_t batch, size_t parts, size_t clip, int elem_size,
    const uint8_t** s_in_ptrs, size_t* s_in_bytes, uint8_t** s_out_ptrs, uint8_t* slots, size_t slot_bytes,
    hipStream_t stream);
void cascaded_launch_select_totals(
    const size_t* s_in_bytes, const size_t* s_out_bytes, size_t parts, size_t candidates, size_t stride,
    unsigned long long* totals, hipStream_t stream);
} // namespace hcamd

extern "C" {

hipcompStatus_t hipcompBatchedCascadedSelectOptsGetTempSize(size_t* temp_bytes)
{
  static const char* fn = "hipcompBatchedCascadedSelectOptsGetTempSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, temp_bytes);
  *temp_bytes = select_layout().end;
  return hipcompSuccess;
}

hipcompStatus_t hipcompBatchedCascadedSelectOpts(
    const void* const* device_uncompressed_ptrs, const size_t* device_uncompressed_bytes, size_t batch_size,
    hipcompType_t type, void* device_temp_ptr, size_t temp_bytes, hipcompBatchedCascadedOpts_t* opts_out,
    double* estimated_ratio, hipStream_t stream)
{
  static const char* fn = "hipcompBatchedCascadedSelectOpts()";
  HCAMD_REQUIRE_NOT_NULL(fn, opts_out);
  int s = 0;
  if (!elem_size_of(type, s))
    return fail(fn, "Unknown type: " + std::to_string((int)type));
  *opts_out = hipcompBatchedCascadedDefaultOpts;
  opts_out->type = type;
  if (estimated_ratio)
    *estimated_ratio = 1.0;
  if (batch_size == 0)
    return hipcompSuccess;
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_ptrs);
  HCAMD_REQUIRE_NOT_NULL(fn, device_uncompressed_bytes);
  HCAMD_REQUIRE_NOT_NULL(fn, device_temp_ptr);
  const SelectLayout l = select_layout();
  if (temp_bytes < l.end || (reinterpret_cast<uintptr_t>(device_temp_ptr) & 7u))
    return fail(fn, "temp space too small or not 8-byte aligned");
  uint8_t* t = static_cast<uint8_t*>(device_temp_ptr);
  const size_t parts = batch_size < kSelectParts ? batch_size : kSelectParts;
  const uint8_t** s_in = reinterpret_cast<const uint8_t**>(t + l.in_ptrs);
  size_t* s_in_bytes = reinterpret_cast<size_t*>(t + l.in_bytes);
  uint8_t** s_out = reinterpret_cast<uint8_t**>(t + l.out_ptrs);
  size_t* s_out_bytes = reinterpret_cast<size_t*>(t + l.out_bytes);
  unsigned long long* totals = reinterpret_cast<unsigned long long*>(t + l.totals);
  cascaded_launch_select_sample(
      reinterpret_cast<const uint8_t* const*>(device_uncompressed_ptrs), device_uncompressed_bytes, batch_size, parts,
      kSelectBytes, s, s_in, s_in_bytes, s_out, t + l.slots, kSelectSlot, stream);
  for (size_t c = 0; c < kNumCandidates; ++c)
    cascaded_launch_compress(s_in, s_in_bytes, s_out, s_out_bytes + c * kSelectParts, parts, (int)type, s,
                             kCandidates[c].R, kCandidates[c].D, kCandidates[c].bp, stream);
  cascaded_launch_select_totals(s_in_bytes, s_out_bytes, parts, kNumCandidates, kSelectParts, totals, stream);
  std::string why;
  if (!launch_ok("cascaded option selection kernels", why))
    return fail(fn, why);
  unsigned long long host[kNumCandidates + 1] = {};
  hipError_t e = hipMemcpyAsync(host, totals, sizeof(host), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize(stream);
  if (e != hipSuccess)
    return fail(fn, std::string("reading the sample sizes back: ") + hipGetErrorString(e), hipcompErrorCudaError);
  size_t best = 0;
  for (size_t c = 1; c < kNumCandidates; ++c)
    if (host[c] < host[best])
      best = c;
  opts_out->num_RLEs = kCandidates[best].R;
  opts_out->num_deltas = kCandidates[best].D;
  opts_out->use_bp = kCandidates[best].bp;
  if (estimated_ratio && host[best] > 0)
    *estimated_ratio = (double)host[kNumCandidates] / (double)host[best];
  return hipcompSuccess;
}

} // extern "C"
