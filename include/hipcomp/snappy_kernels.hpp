/* hipcomp/snappy_kernels.hpp -- the entry points one layer below the batched Snappy C API.
 *
 * These replace the reference's INTERNAL interface src/lowlevel/SnappyBatchKernels.h:80-148 (with
 * gpu_snappy_status_s of src/snappy/types.h:60-62): the functions its hipcompBatchedSnappy* wrappers call
 * (src/lowlevel/SnappyBatch.cpp:84-245) and its own tests/test_snappy_app.cpp and
 * src/test/SnappyLargeTokens_test.cpp are written against.  Same namespace, names, argument order and
 * meaning, so that a caller of that layer -- those two programs, compiled unchanged -- links with this
 * library.  Everything is asynchronous on `stream`; all pointers are device pointers; count <= 0 launches
 * nothing.
 *
 *  gpu_snap    device_out_available_bytes may be null (every buffer holds the worst case,
 *              32 + n + n / 6 bytes); where it is given and a buffer is smaller than the worst case for
 *              its input, the chunk is not compressed: status 1, size 0 (the reference lets its encoder
 *              run and reports an output that did not fit, compression.hiph:307-311, :383).  outputs may
 *              be null.
 *  gpu_unsnap  device_out_available_bytes null or an entry of 0: the buffer holds whatever the stream
 *              says (decompression.hiph:148-149).  outputs and device_out_bytes may be null.
 */
#ifndef HIPCOMP_SNAPPY_KERNELS_HPP
#define HIPCOMP_SNAPPY_KERNELS_HPP

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "hipcomp/shared_types.h"

namespace hipcomp {

struct gpu_snappy_status_s
{
  uint32_t status; /* non-zero: an error */
};

void gpu_snap(
    const void* const* device_in_ptr, const size_t* device_in_bytes, void* const* device_out_ptr,
    const size_t* device_out_available_bytes, gpu_snappy_status_s* outputs, size_t* device_out_bytes, int count,
    hipStream_t stream);

void gpu_unsnap(
    const void* const* device_in_ptr, const size_t* device_in_bytes, void* const* device_out_ptr,
    const size_t* device_out_available_bytes, hipcompStatus_t* outputs, size_t* device_out_bytes, int count,
    hipStream_t stream);

void gpu_get_uncompressed_sizes(
    const void* const* device_in_ptr, const size_t* device_in_bytes, size_t* device_out_bytes, int count,
    hipStream_t stream);

} /* namespace hipcomp */

#endif
