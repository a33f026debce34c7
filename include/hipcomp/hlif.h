/*
 * hipcomp/hlif.h -- C binding of the high-level interface's managers (hipcomp/lz4.hpp,
 * snappy.hpp, cascaded.hpp, hipcompManagerFactory.hpp; the reference offers them in C++
 * only).  A manager is an opaque handle; every call returns a status instead of throwing.
 */
#ifndef HIPCOMP_HLIF_H
#define HIPCOMP_HLIF_H

#include "hipcomp.h"
#include "hipcomp/cascaded.h"

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hipcompHlifManager hipcompHlifManager_t;

hipcompStatus_t hipcompHlifLZ4ManagerCreate(
    size_t uncomp_chunk_size, hipcompType_t data_type, hipStream_t stream, hipcompHlifManager_t** manager);
hipcompStatus_t hipcompHlifSnappyManagerCreate(
    size_t uncomp_chunk_size, hipStream_t stream, hipcompHlifManager_t** manager);
/* options.chunk_size is the chunk size of the container */
hipcompStatus_t hipcompHlifCascadedManagerCreate(
    hipcompBatchedCascadedOpts_t options, hipStream_t stream, hipcompHlifManager_t** manager);
/* the manager that reads the container at device_container (synchronises the stream) */
hipcompStatus_t hipcompHlifManagerCreateFromContainer(
    const void* device_container, hipStream_t stream, hipcompHlifManager_t** manager);
hipcompStatus_t hipcompHlifManagerDestroy(hipcompHlifManager_t* manager);

/* *max_compressed_bytes: size to give the container buffer; *num_chunks: chunks it will hold */
hipcompStatus_t hipcompHlifConfigureCompression(
    hipcompHlifManager_t* manager, size_t uncompressed_bytes, size_t* max_compressed_bytes, size_t* num_chunks);
/* asynchronous on the manager's stream; the outcome is read with hipcompHlifGetLastStatus */
hipcompStatus_t hipcompHlifCompress(
    hipcompHlifManager_t* manager, const void* device_uncompressed, size_t uncompressed_bytes,
    void* device_container);
/* reads the container header (synchronises the stream) */
hipcompStatus_t hipcompHlifGetDecompressedSize(
    hipcompHlifManager_t* manager, const void* device_container, size_t* uncompressed_bytes, size_t* num_chunks);
hipcompStatus_t hipcompHlifGetCompressedSize(
    hipcompHlifManager_t* manager, const void* device_container, size_t* container_bytes);
hipcompStatus_t hipcompHlifDecompress(
    hipcompHlifManager_t* manager, const void* device_container, void* device_uncompressed);
/* status of the manager's last compress / decompress; synchronises the stream */
hipcompStatus_t hipcompHlifGetLastStatus(hipcompHlifManager_t* manager, hipcompStatus_t* status);
hipcompStatus_t hipcompHlifGetRequiredScratchBytes(hipcompHlifManager_t* manager, size_t* scratch_bytes);
/* hipcompManagerBase::set_scratch_buffer: the caller's buffer of hipcompHlifGetRequiredScratchBytes bytes
   (device memory, 16-byte aligned) replaces the one the manager would allocate itself */
hipcompStatus_t hipcompHlifSetScratchBuffer(hipcompHlifManager_t* manager, void* device_scratch);

#ifdef __cplusplus
}
#endif

#endif
