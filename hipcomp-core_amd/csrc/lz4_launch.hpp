// lz4_launch.hpp -- host-callable launchers of the LZ4 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#include "hipcomp/shared_types.h"
#include "placement.hpp"

namespace hcamd {

// compute units of the calling thread's current device (cached per device; lz4_kernels.hip)
int num_cus_of_current_device();

// (The "pair" shape -- lz4_mix.hiph, lz4_compress_kernel_pair: two waves per chunk, one chunk per workgroup -- is
// what data without matches meets in chunks of 16 .. 64 KiB and batches of several thousand; else "mix":)
// Most waves (= chunks in flight) one compression workgroup of the "mix" shape
// holds.  Four (one per SIMD): the kernel may then use up to 256 vector
// registers and keeps clear of the accumulation registers, which its walk uses
// by name (lz4_kernels.hip, HC_WALK_AGPRS; tests/test_build_guards_cpu.py
// checks the build).
constexpr int kLz4MaxWavesPerGroup = 4;

// Launch shape of a compression kernel: per workgroup `tagged` waves whose
// chunk has a tag table behind its position table and `plain` waves without.
struct Lz4CompressShape
{
  uint32_t tagged, plain;
  uint32_t stride_tagged, stride_plain; // LDS bytes of one wave's tables
  uint32_t lds_bytes;                   // of the workgroup
  uint32_t groups;                      // persistent workgroups
  uint32_t waves() const { return tagged + plain; }
};
// the shape with the tables in LDS
Lz4CompressShape lz4_compress_shape_mix(uint32_t ht_size, size_t batch);

enum class Lz4Mode { Auto, Mix, Far, FarSparse, FarWide };
// Auto in the library that ships.  The measurement / test build (-DHC_MEASUREMENT_KNOBS,
// lib/libhipcomp_knobs.so) reads HIPCOMP_LZ4_SHAPE = auto | mix | far | fars | farw at every call;
// the compressed bytes do not depend on it.  auto: a routing kernel
// puts every chunk on the list of the shape its data calls for; the others run every chunk
// through one shape (far / fars: the lean form with the launch geometry for dense / sparse data).
Lz4Mode lz4_mode_from_environment();

// `temp` / `temp_bytes`: the caller's temp buffer (hipcompBatchedLZ4CompressGetTempSize bytes by
// contract), used while the call runs as far as it goes: 64 words of header -- a chunk ticket
// counter and a list length per launch shape, sample totals (lz4_far.hiph, kHeaderWords) -- zeroed by the
// launcher on the stream, the routing kernel's lists (4 x batch words), and hash tables for the far kernel's
// device-table waves (max(ht_size, 8) uint16 each, 16-byte aligned).  Too small for the lists:
// no routing, the LDS shape for all; too small for the header: one chunk per wave.  nullptr / 0
// is accepted (the same).  batch must be > 0 and < 2^31.
// Placement: see placement.hpp.
typedef Placement Lz4Placement;
// how many slots a launch can ask for (the most waves any of the compress kernels holds on the device)
size_t lz4_placement_slots();

hipError_t lz4_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, uint32_t ht_size,
    size_t batch, int elem_size, void* temp, size_t temp_bytes,
    size_t max_chunk_bytes, Lz4Mode mode, hipStream_t stream, const Lz4Placement* place = nullptr);
// what the launcher makes use of at most (for callers that size their own scratch: hlif.hip)
size_t lz4_compress_temp_bytes_used(uint32_t ht_size, size_t batch);

// write_out == false: parse-only pass that reports sizes.
// `temp` / `temp_bytes`: the caller's temp buffer (hipcompBatchedLZ4DecompressGetTempSize bytes by contract).  A call
// with more chunks than the chip holds waves uses ONE 4-byte word of it as its chunk ticket counter (zeroed on the
// stream), a different word for every call of the process -- calls in flight at once may share the buffer; nullptr /
// too small: accepted, one wave per chunk by its position in the grid.  Returns the error of a launch that failed.
hipError_t lz4_launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    const size_t* out_caps, size_t batch, uint8_t* const* out_ptrs,
    size_t* actual_bytes, hipcompStatus_t* statuses, bool write_out,
    hipStream_t stream, void* temp = nullptr, size_t temp_bytes = 0);

} // namespace hcamd
