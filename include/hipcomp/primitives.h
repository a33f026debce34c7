/*
 * hipcomp/primitives.h -- C binding of the whole-array run-length / delta /
 * bit-packing primitives of hipcomp/primitives.hpp (reference
 * src/RunLengthEncodeGPU.h, src/DeltaGPU.h, src/BitPackGPU.h; the reference
 * exposes them as C++ classes only).  Same arguments as the class methods;
 * a failure (bad type, workspace too small, launch error) is a status instead
 * of an exception.
 */
#ifndef HIPCOMP_PRIMITIVES_H
#define HIPCOMP_PRIMITIVES_H

#include "hipcomp.h"

#include <hip/hip_runtime_api.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

hipcompStatus_t hipcompRunLengthEncodeGetWorkspaceSize(
    size_t num, hipcompType_t valueType, hipcompType_t countType, size_t* workspace_bytes);

hipcompStatus_t hipcompRunLengthEncodeCompress(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void* outValues,
    hipcompType_t countType, void* outCounts, size_t* numOutDevice, const void* in, size_t num,
    hipStream_t stream);

hipcompStatus_t hipcompRunLengthEncodeCompressDownstream(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void** outValuesPtr,
    hipcompType_t countType, void** outCountsPtr, size_t* numOutDevice, const void* in,
    const size_t* numDevice, size_t maxNum, hipStream_t stream);

hipcompStatus_t hipcompDeltaGetWorkspaceSize(size_t num, hipcompType_t type, size_t* workspace_bytes);

hipcompStatus_t hipcompDeltaCompress(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void** outValuesPtr,
    const void* inValues, const size_t* numDevice, size_t maxNum, hipStream_t stream);

hipcompStatus_t hipcompBitPackGetWorkspaceSize(size_t num, hipcompType_t type, size_t* workspace_bytes);

hipcompStatus_t hipcompBitPackCompress(
    void* workspace, size_t workspaceSize, hipcompType_t inType, void* const* outPtr, const void* in,
    const size_t* numDevice, size_t maxNum, void* const* minValueDevicePtr,
    unsigned char* const* numBitsDevicePtr, hipStream_t stream);

#ifdef __cplusplus
}
#endif

#endif
