/*
 * hipcomp/snappy.h -- batched raw-Snappy block codec, C ABI.
 *
 * Each entry point replaces the same-named function of the reference
 * (declarations: reference include/hipcomp/snappy.h:80-195; definitions:
 * reference src/lowlevel/SnappyBatch.cpp:84-245).  Same ownership, async and
 * device-resident-array contract as hipcomp/lz4.h.
 */
#ifndef HIPCOMP_SNAPPY_H
#define HIPCOMP_SNAPPY_H

#include "hipcomp.h"
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference snappy.h:62-67 */
typedef struct
{
  int reserved;
} hipcompBatchedSnappyOpts_t;

static const hipcompBatchedSnappyOpts_t hipcompBatchedSnappyDefaultOpts = {0};

/* temp_bytes = 0.  (reference SnappyBatch.cpp:84-102) */
hipcompStatus_t hipcompBatchedSnappyDecompressGetTempSize(
    size_t num_chunks, size_t max_uncompressed_chunk_size, size_t* temp_bytes);

/* Varint preamble of every chunk; 0 for an empty / oversized (>= 2^31) one.
 * (reference SnappyBatch.cpp:104-130) */
hipcompStatus_t hipcompBatchedSnappyGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream);

/* Decompress; actual-bytes and statuses arrays may be NULL.
 * (reference SnappyBatch.cpp:132-166) */
hipcompStatus_t hipcompBatchedSnappyDecompressAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    const size_t* device_uncompressed_bytes,
    size_t* device_actual_uncompressed_bytes,
    size_t batch_size,
    void* const device_temp_ptr,
    const size_t temp_bytes,
    void* const* device_uncompressed_ptrs,
    hipcompStatus_t* device_statuses,
    hipStream_t stream);

/* temp_bytes = 0.  (reference SnappyBatch.cpp:168-187) */
hipcompStatus_t hipcompBatchedSnappyCompressGetTempSize(
    size_t batch_size,
    size_t max_chunk_size,
    hipcompBatchedSnappyOpts_t format_opts,
    size_t* temp_bytes);

/* max_compressed_size = 32 + n + n/6.  (reference SnappyBatch.cpp:189-206) */
hipcompStatus_t hipcompBatchedSnappyCompressGetMaxOutputChunkSize(
    size_t max_chunk_size,
    hipcompBatchedSnappyOpts_t format_opts,
    size_t* max_compressed_size);

/* Compress; output identical to the reference's wave64 encoder.
 * (reference SnappyBatch.cpp:208-245) */
hipcompStatus_t hipcompBatchedSnappyCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    hipcompBatchedSnappyOpts_t format_opts,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif
