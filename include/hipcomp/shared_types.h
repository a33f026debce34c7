/*
 * shared_types.h -- status codes of the batched codec C ABI.
 *
 * Drop-in for the reference's include/hipcomp/shared_types.h:52-66 (same
 * enumerator names and numeric values, including the nvcomp* aliases, so
 * callers compiled against either header see one ABI).
 */
#ifndef HIPCOMP_SHARED_TYPES_H
#define HIPCOMP_SHARED_TYPES_H

typedef enum hipcompStatus_t
{
  hipcompSuccess = 0,
  hipcompErrorInvalidValue = 10,     /* bad argument / host-side failure   */
  hipcompErrorNotSupported = 11,
  hipcompErrorCannotDecompress = 12, /* per-chunk: stream is not decodable */
  hipcompErrorCudaError = 1000,      /* HIP runtime error                  */
  hipcompErrorInternal = 10000,
  nvcompSuccess = hipcompSuccess,
  nvcompErrorInvalidValue = hipcompErrorInvalidValue,
  nvcompErrorNotSupported = hipcompErrorNotSupported,
  nvcompErrorCannotDecompress = hipcompErrorCannotDecompress,
  nvcompErrorCudaError = hipcompErrorCudaError,
  nvcompErrorInternal = hipcompErrorInternal
} hipcompStatus_t;

#endif
