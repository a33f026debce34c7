// placement.hpp -- how the high-level managers (csrc/hlif.hip) have the batched encoders put a chunk
// straight into the container; the batched API leaves it empty.
//
// A chunk is not written where out_ptrs says but into a slot of the wave's own, and when it is done -- its
// size known -- the wave takes its place in the container with one atomic add on the container's byte count,
// copies it there and notes where: chunk data in completion order, the reference's scheme (each CTA of its
// persistent loop compresses into its scratch slot, claims room with an atomic on comp_data_size and copies:
// reference src/hipcomp_common_deps/hlif_shared.hiph:165-232).  One slot per resident wave instead of one per
// chunk, and no pass over the compressed bytes afterwards.  out_ptrs is not read then (may be null).
#pragma once

#include <cstddef>
#include <cstdint>

namespace hcamd {

struct Placement
{
  uint8_t* slots = nullptr;              // nullptr: off
  unsigned long long slot_bytes = 0;     // at least the maximal compressed chunk
  uint8_t* data = nullptr;               // where the container's chunks begin
  unsigned long long* cursor = nullptr;  // the container's byte count so far (8-byte aligned)
  unsigned long long* offsets = nullptr; // per chunk of the batch: where it went, relative to `data`
  uint32_t align = 1;                    // chunk starts are multiples of it
};

} // namespace hcamd
