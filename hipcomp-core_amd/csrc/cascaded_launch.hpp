// cascaded_launch.hpp -- host-callable launchers of the Cascaded kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#include "hipcomp/shared_types.h"
#include "placement.hpp"

namespace hcamd {

void cascaded_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, size_t batch, int type_tag,
    int elem_size, int num_rles, int num_deltas, int use_bp, hipStream_t stream);

// The high-level manager's compress (placement.hpp): `ticket` = one word of device memory of the call's own, ZERO when the kernel starts (the caller zeroes it on the stream),
// place.slots = cascaded_placement_slots(elem_size) slots.  out_ptrs does not exist here.
size_t cascaded_placement_slots(int elem_size);
hipError_t cascaded_launch_compress_placed(
    const uint8_t* const* in_ptrs, const size_t* in_bytes, size_t* out_bytes, size_t batch, int type_tag,
    int elem_size, int num_rles, int num_deltas, int use_bp, uint32_t* ticket, const Placement& place,
    hipStream_t stream);

// an error: a launch could not be set up (nothing was decoded by it; the caller fails the call)
hipError_t cascaded_launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    const size_t* out_caps, size_t batch, uint8_t* const* out_ptrs,
    size_t* actual_bytes, hipcompStatus_t* statuses, hipStream_t stream);

void cascaded_launch_get_sizes(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    size_t* out_sizes, size_t batch, hipStream_t stream);

} // namespace hcamd
