// lz4_launch.hpp -- host-callable launchers of the LZ4 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#include "hipcomp/shared_types.h"

namespace hcamd {

// LDS bytes of one chunk's tables: positions (u16 per slot) and, with `tags`,
// the tag table (u8 per slot) that spares the encoder most candidate fetches.
size_t lz4_compress_lds_bytes(uint32_t ht_size, bool tags);

// Most waves (= chunks in flight) one compression workgroup holds.  Four (one
// per SIMD): the kernel may then use up to 256 vector registers and keeps
// clear of the accumulation registers, which its walk uses by name
// (lz4_kernels.hip, HC_WALK_AGPRS; tests/test_abi_cpu.py checks the build).
constexpr int kLz4MaxWavesPerGroup = 4;

// Launch shape of the compression kernel: `waves` chunks in flight per
// workgroup, each with its own `table_stride` bytes of LDS.
struct Lz4CompressShape
{
  uint32_t waves;
  uint32_t table_stride;
  uint32_t lds_bytes;
  uint32_t groups;
};
Lz4CompressShape lz4_compress_shape(uint32_t ht_size, size_t batch, bool tags);

// `ticket` is one zero-initialised-by-the-launcher uint32 in device memory
// (in the caller's temp buffer) from which the waves of the persistent
// workgroups draw chunk numbers; nullptr = one chunk per wave, as many
// workgroups as that takes.  batch must be > 0 and < 2^31.
hipError_t lz4_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, uint32_t ht_size,
    size_t batch, int elem_size, uint32_t* ticket, size_t max_chunk_bytes, bool tags, hipStream_t stream);

// write_out == false: parse-only pass that reports sizes.
void lz4_launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    const size_t* out_caps, size_t batch, uint8_t* const* out_ptrs,
    size_t* actual_bytes, hipcompStatus_t* statuses, bool write_out,
    hipStream_t stream);

} // namespace hcamd
