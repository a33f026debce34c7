/*
 * hipcomp/cascaded.h -- batched Cascaded (RLE -> Delta -> BitPack) codec, C ABI.
 *
 * Each entry point replaces the same-named function of the reference
 * (declarations: reference include/hipcomp/cascaded.h:142-295; definitions:
 * reference src/lowlevel/CascadedBatch.hip:306-462).  Same ownership, async
 * and device-resident-array contract as hipcomp/lz4.h.  Input and output
 * buffers must be 4-byte aligned and aligned to the element type.
 */
#ifndef HIPCOMP_CASCADED_H
#define HIPCOMP_CASCADED_H

#include "hipcomp.h"
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference cascaded.h:90-125.  chunk_size is carried but, as in the
 * reference, the internal sub-chunk is fixed at 4096 bytes whatever it says,
 * and the streams are the reference's.  (An extension of this library, off by
 * default: with HIPCOMP_CASCADED_CHUNK_SIZE=honour in the environment the
 * values 8192 and 16384 are honoured; such streams carry the size in the high
 * nibble of header byte 2 and are NOT readable by the reference -- its decoder
 * takes the byte for use_bp and walks 4096-byte sub-chunks.  This library's
 * decoder reads all three sizes with or without the variable.) */
typedef struct
{
  size_t chunk_size;
  hipcompType_t type;
  int num_RLEs;
  int num_deltas;
  int use_bp;
} hipcompBatchedCascadedOpts_t;

static const hipcompBatchedCascadedOpts_t hipcompBatchedCascadedDefaultOpts
    = {4096, HIPCOMP_TYPE_INT, 2, 1, 1};

/* temp_bytes = 0.  (reference CascadedBatch.hip:306-316) */
hipcompStatus_t hipcompBatchedCascadedCompressGetTempSize(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    hipcompBatchedCascadedOpts_t format_opts,
    size_t* temp_bytes);

/* max_compressed_bytes = roundUp4(n) + 8.  (reference CascadedBatch.hip:318-327) */
hipcompStatus_t hipcompBatchedCascadedCompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes,
    hipcompBatchedCascadedOpts_t format_opts,
    size_t* max_compressed_bytes);

/* (reference CascadedBatch.hip:329-357) */
hipcompStatus_t hipcompBatchedCascadedCompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    const hipcompBatchedCascadedOpts_t format_opts,
    hipStream_t stream);

/* temp_bytes = 0.  (reference CascadedBatch.hip:359-364) */
hipcompStatus_t hipcompBatchedCascadedDecompressGetTempSize(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes);

/* Actual-bytes and statuses arrays are required (not nullable).
 * (reference CascadedBatch.hip:366-436) */
hipcompStatus_t hipcompBatchedCascadedDecompressAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    const size_t* device_uncompressed_bytes,
    size_t* device_actual_uncompressed_bytes,
    size_t batch_size,
    void* const device_temp_ptr,
    size_t temp_bytes,
    void* const* device_uncompressed_ptrs,
    hipcompStatus_t* device_statuses,
    hipStream_t stream);

/* (reference CascadedBatch.hip:438-462) */
hipcompStatus_t hipcompBatchedCascadedGetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif
