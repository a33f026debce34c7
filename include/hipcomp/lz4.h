/*
 * hipcomp/lz4.h -- batched LZ4 block codec, C ABI.
 *
 * Each entry point replaces the same-named function of the reference
 * (declarations: reference include/hipcomp/lz4.h:106-243; definitions:
 * reference src/lowlevel/LZ4Batch.cpp:71-224).  Contract, unchanged:
 *   - the library allocates nothing; all buffers (incl. temp) are the caller's;
 *   - pointer arrays and size arrays live in device-accessible memory and are
 *     dereferenced on the device;
 *   - every call is asynchronous on `stream` and never synchronises;
 *   - chunk i is an independent raw LZ4 block (no frame);
 *   - compressed bytes are identical to the reference's wave64 encoder.
 */
#ifndef HIPCOMP_LZ4_H
#define HIPCOMP_LZ4_H

#include "hipcomp.h"
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference lz4.h:79-84: the element width the matcher works in
 * (CHAR/UCHAR/BITS -> 1 B, SHORT/USHORT -> 2 B, INT/UINT -> 4 B). */
typedef struct
{
  hipcompType_t data_type;
} hipcompBatchedLZ4Opts_t;

static const hipcompBatchedLZ4Opts_t hipcompBatchedLZ4DefaultOpts
    = {HIPCOMP_TYPE_CHAR};

/* temp_bytes = min(pow2ceil(max_chunk),16384) * 2 * batch_size; InvalidValue
 * if max_chunk > 16 MiB.  (reference LZ4Batch.cpp:154-170) */
hipcompStatus_t hipcompBatchedLZ4CompressGetTempSize(
    size_t batch_size,
    size_t max_uncompressed_chunk_bytes,
    hipcompBatchedLZ4Opts_t format_opts,
    size_t* temp_bytes);

/* max_compressed_bytes = roundUp8(n + 1 + ceil(n/255)).
 * (reference LZ4Batch.cpp:172-187) */
hipcompStatus_t hipcompBatchedLZ4CompressGetMaxOutputChunkSize(
    size_t max_uncompressed_chunk_bytes,
    hipcompBatchedLZ4Opts_t format_opts,
    size_t* max_compressed_bytes);

/* Compress batch_size chunks.  max_uncompressed_chunk_bytes sizes the hash
 * table exactly as the reference does (LZ4CompressionKernels.hip:171), so it
 * takes part in the bit-exact result.  device_temp_ptr: device memory of at
 * least hipcompBatchedLZ4CompressGetTempSize bytes; it holds the call's chunk
 * counter and, for data that compresses, its hash tables, so ONE temp buffer
 * serves ONE compress call at a time (calls in flight on different streams
 * need a buffer each).  batch_size < 2^31.  (reference LZ4Batch.cpp:189-224) */
hipcompStatus_t hipcompBatchedLZ4CompressAsync(
    const void* const* device_uncompressed_ptrs,
    const size_t* device_uncompressed_bytes,
    size_t max_uncompressed_chunk_bytes,
    size_t batch_size,
    void* device_temp_ptr,
    size_t temp_bytes,
    void* const* device_compressed_ptrs,
    size_t* device_compressed_bytes,
    hipcompBatchedLZ4Opts_t format_opts,
    hipStream_t stream);

/* temp_bytes = roundUp8(24 * num_chunks).  (reference LZ4Batch.cpp:71-87) */
hipcompStatus_t hipcompBatchedLZ4DecompressGetTempSize(
    size_t num_chunks, size_t max_uncompressed_chunk_bytes, size_t* temp_bytes);

/* Decompress.  device_actual_uncompressed_bytes and device_statuses may be
 * NULL.  An undecodable chunk gets size 0 and hipcompErrorCannotDecompress.
 * device_temp_ptr: as in the reference a valid device pointer; one 4-byte word
 * of it (another one for every call) may be written while the call runs, so
 * calls in flight at once may share one buffer.
 * (reference LZ4Batch.cpp:89-125) */
hipcompStatus_t hipcompBatchedLZ4DecompressAsync(
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

/* Parse-only pass: uncompressed size of every chunk.
 * (reference LZ4Batch.cpp:127-152) */
hipcompStatus_t hipcompBatchedLZ4GetDecompressSizeAsync(
    const void* const* device_compressed_ptrs,
    const size_t* device_compressed_bytes,
    size_t* device_uncompressed_bytes,
    size_t batch_size,
    hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif
