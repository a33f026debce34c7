/*
 * hipcomp.h -- element-type tags and version of the batched codec C ABI.
 *
 * Drop-in for the reference's include/hipcomp.h:64-80.  Only the batched
 * low-level interface (hipcomp/lz4.h, hipcomp/snappy.h, hipcomp/cascaded.h)
 * is provided by this library; the six deprecated hipcompDecompress* entry
 * points the reference declares at hipcomp.h:106-186 have no definition in
 * the reference either (src/hipcomp_api.cpp is empty) and are not declared.
 */
#ifndef HIPCOMP_H
#define HIPCOMP_H

#include <stddef.h>
#include <hip/hip_runtime_api.h>
#include "hipcomp/shared_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HIPCOMP_MAJOR_VERSION 2
#define HIPCOMP_MINOR_VERSION 2
#define HIPCOMP_PATCH_VERSION 0

/* Element type of the data inside a chunk (reference hipcomp.h:69-80). */
typedef enum hipcompType_t
{
  HIPCOMP_TYPE_CHAR = 0,      /* 1 byte  */
  HIPCOMP_TYPE_UCHAR = 1,     /* 1 byte  */
  HIPCOMP_TYPE_SHORT = 2,     /* 2 bytes */
  HIPCOMP_TYPE_USHORT = 3,    /* 2 bytes */
  HIPCOMP_TYPE_INT = 4,       /* 4 bytes */
  HIPCOMP_TYPE_UINT = 5,      /* 4 bytes */
  HIPCOMP_TYPE_LONGLONG = 6,  /* 8 bytes */
  HIPCOMP_TYPE_ULONGLONG = 7, /* 8 bytes */
  HIPCOMP_TYPE_BITS = 0xff    /* opaque bytes */
} hipcompType_t;

#ifdef __cplusplus
}
#endif

#endif
