// prims_ref_shim.cpp -- test infrastructure, not product.
//
// C entry points over the REFERENCE's whole-array primitives (its classes
// hipcomp::RunLengthEncodeGPU / DeltaGPU / BitPackGPU, reference
// src/RunLengthEncodeGPU.h:67-131, src/DeltaGPU.h:66-86, src/BitPackGPU.h:66-110),
// with the signatures of this repo's include/hipcomp/primitives.h but the prefix
// `ref_`, so that tests/test_primitives_gpu.py can call the reference build and the
// product side by side through ctypes.  Compiled together with the reference's own
// unmodified sources into oracle/_ref/libhipcomp_prims_ref.so (oracle/Makefile);
// includes the reference's headers, so it is built only where /root/reference is.
#include "BitPackGPU.h"
#include "DeltaGPU.h"
#include "RunLengthEncodeGPU.h"

#include <exception>

using namespace hipcomp;

#define REF_TRY(stmt)                 \
  try {                               \
    stmt;                             \
    return 0;                         \
  } catch (const std::exception&) {   \
    return 10;                        \
  }

extern "C" {

int ref_hipcompRunLengthEncodeGetWorkspaceSize(size_t num, hipcompType_t vt, hipcompType_t ct, size_t* out)
{
  REF_TRY(*out = RunLengthEncodeGPU::requiredWorkspaceSize(num, vt, ct))
}

int ref_hipcompRunLengthEncodeCompress(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void* outValues, hipcompType_t countType,
    void* outCounts, size_t* numOutDevice, const void* in, size_t num, hipStream_t stream)
{
  REF_TRY(RunLengthEncodeGPU::compress(workspace, workspaceSize, valueType, outValues, countType, outCounts,
                                       numOutDevice, in, num, stream))
}

int ref_hipcompRunLengthEncodeCompressDownstream(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void** outValuesPtr, hipcompType_t countType,
    void** outCountsPtr, size_t* numOutDevice, const void* in, const size_t* numDevice, size_t maxNum,
    hipStream_t stream)
{
  REF_TRY(RunLengthEncodeGPU::compressDownstream(workspace, workspaceSize, valueType, outValuesPtr, countType,
                                                 outCountsPtr, numOutDevice, in, numDevice, maxNum, stream))
}

int ref_hipcompDeltaGetWorkspaceSize(size_t num, hipcompType_t type, size_t* out)
{
  REF_TRY(*out = DeltaGPU::requiredWorkspaceSize(num, type))
}

int ref_hipcompDeltaCompress(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void** outValuesPtr, const void* inValues,
    const size_t* numDevice, size_t maxNum, hipStream_t stream)
{
  REF_TRY(DeltaGPU::compress(workspace, workspaceSize, valueType, outValuesPtr, inValues, numDevice, maxNum, stream))
}

int ref_hipcompBitPackGetWorkspaceSize(size_t num, hipcompType_t type, size_t* out)
{
  REF_TRY(*out = BitPackGPU::requiredWorkspaceSize(num, type))
}

int ref_hipcompBitPackCompress(
    void* workspace, size_t workspaceSize, hipcompType_t inType, void* const* outPtr, const void* in,
    const size_t* numDevice, size_t maxNum, void* const* minValueDevicePtr, unsigned char* const* numBitsDevicePtr,
    hipStream_t stream)
{
  REF_TRY(BitPackGPU::compress(workspace, workspaceSize, inType, outPtr, in, numDevice, maxNum, minValueDevicePtr,
                               numBitsDevicePtr, stream))
}

} // extern "C"
