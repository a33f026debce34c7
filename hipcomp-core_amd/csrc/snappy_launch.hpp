// snappy_launch.hpp -- host-callable launchers of the Snappy kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#include "hipcomp/shared_types.h"
#include "placement.hpp"

namespace hcamd {

void snappy_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, size_t batch,
    hipStream_t stream, const size_t* out_available = nullptr, uint32_t* statuses = nullptr);

// The high-level manager's compress (placement.hpp): `ticket` = one word of device memory of the call's own, ZERO when the kernel starts (the caller zeroes it on the stream),
// place.slots = snappy_placement_slots() slots.  out_ptrs does not exist here.
size_t snappy_placement_slots();
hipError_t snappy_launch_compress_placed(
    const uint8_t* const* in_ptrs, const size_t* in_bytes, size_t* out_bytes, size_t batch,
    uint32_t* ticket, const Placement& place, hipStream_t stream);

void snappy_launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    const size_t* out_caps, size_t batch, uint8_t* const* out_ptrs,
    size_t* actual_bytes, hipcompStatus_t* statuses, hipStream_t stream);

void snappy_launch_get_sizes(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    size_t* out_sizes, size_t batch, hipStream_t stream);

} // namespace hcamd
