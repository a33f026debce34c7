/*
 * hipcomp/primitives.hpp -- whole-array run-length / delta / bit-packing
 * primitives (C++), the building blocks the Cascaded scheme is made of.
 *
 * Same classes, static methods, argument order and meaning as the reference's
 * internal headers (reference src/RunLengthEncodeGPU.h:60-131,
 * src/DeltaGPU.h:60-96, src/BitPackGPU.h:55-104), which its unit tests link
 * (src/test/{RunLengthEncodeGPU,DeltaGPU,BitPackGPU}_test.cpp).  The batched
 * Cascaded calls do not go through them (there the layers run fused in LDS);
 * they are exported for callers that used the classes directly.  Errors are
 * reported the reference's way, as std::runtime_error.  A C binding of the
 * same operations is in hipcomp/primitives.h.
 *
 * Sizes and output pointers that the reference keeps in device memory
 * (`numDevice`, `outValuesPtr`, ...) are read on the device here too: nothing
 * synchronises with the host.
 */
#ifndef HIPCOMP_PRIMITIVES_HPP
#define HIPCOMP_PRIMITIVES_HPP

#include "hipcomp.h"

#include <hip/hip_runtime_api.h>
#include <cstddef>

namespace hipcomp
{

class RunLengthEncodeGPU
{
public:
  /* in[0, num) -> outValues / outCounts (one entry per run), *numOutDevice = runs */
  static void compress(
      void* workspace, size_t workspaceSize, hipcompType_t valueType, void* outValues,
      hipcompType_t countType, void* outCounts, size_t* numOutDevice, const void* in, size_t num,
      hipStream_t stream);

  /* the same with the element count and both output addresses resident on the device */
  static void compressDownstream(
      void* workspace, size_t workspaceSize, hipcompType_t valueType, void** outValuesPtr,
      hipcompType_t countType, void** outCountsPtr, size_t* numOutDevice, const void* in,
      const size_t* numDevice, size_t maxNum, hipStream_t stream);

  static size_t requiredWorkspaceSize(size_t num, hipcompType_t valueType, hipcompType_t countType);
};

class DeltaGPU
{
public:
  /* (*outValuesPtr)[i] = in[i] - in[i-1], in[-1] taken as 0, for i < *numDevice */
  static void compress(
      void* workspace, size_t workspaceSize, hipcompType_t valueType, void** outValuesPtr,
      const void* inValues, const size_t* numDevice, size_t maxNum, hipStream_t stream);

  static size_t requiredWorkspaceSize(size_t num, hipcompType_t type);
};

class BitPackGPU
{
public:
  /* **minValueDevicePtr = min(in), **numBitsDevicePtr = bits(max - min); (in[i] - min) packed
     numBits apiece, value i at bit i * numBits, into 32-bit words (64-bit for 8-byte types) */
  static void compress(
      void* workspace, size_t workspaceSize, hipcompType_t inType, void* const* outPtr, const void* in,
      const size_t* numDevice, size_t maxNum, void* const* minValueDevicePtr,
      unsigned char* const* numBitsDevicePtr, hipStream_t stream);

  static size_t requiredWorkspaceSize(size_t num, hipcompType_t type);
};

} // namespace hipcomp

#endif
