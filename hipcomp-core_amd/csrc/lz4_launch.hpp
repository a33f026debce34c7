// lz4_launch.hpp -- host-callable launchers of the LZ4 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#include "hipcomp/shared_types.h"

namespace hcamd {

// Dynamic LDS one compression workgroup (= one chunk) needs.
size_t lz4_compress_lds_bytes(uint32_t ht_size);

void lz4_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, uint32_t ht_size,
    size_t batch, int elem_size, hipStream_t stream);

// write_out == false: parse-only pass that reports sizes.
void lz4_launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    const size_t* out_caps, size_t batch, uint8_t* const* out_ptrs,
    size_t* actual_bytes, hipcompStatus_t* statuses, bool write_out,
    hipStream_t stream);

} // namespace hcamd
