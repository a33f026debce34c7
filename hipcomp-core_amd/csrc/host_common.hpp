// host_common.hpp -- host-side glue shared by the three batched codecs.
//
// Mirrors the *contract* of the reference's support layer, re-derived rather
// than copied: pointer validation (reference src/HipUtils.hip:91-108,150-170),
// error -> status mapping with one stderr line (reference src/Check.cpp:80-99)
// and the round-up helpers (reference src/common.h).  No exceptions cross the
// C ABI: every entry point returns a hipcompStatus_t.
#pragma once

#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>

#include "hipcomp/shared_types.h"

namespace hcamd {

inline size_t round_up_div(size_t a, size_t b) { return (a + b - 1) / b; }
inline size_t round_up_to(size_t a, size_t b) { return round_up_div(a, b) * b; }

// One line on stderr, same shape as the reference ("ERROR: In <fn>: <what>").
inline hipcompStatus_t fail(const char* fn, const std::string& what,
                            hipcompStatus_t code = hipcompErrorInvalidValue)
{
  std::fprintf(stderr, "ERROR: In %s: %s\n", fn, what.c_str());
  return code;
}

// Translate `p` to a device-accessible address (identity for hipMalloc
// memory, the mapped address for registered/pinned host memory).  Returns
// false when the current GPU cannot dereference it (incl. nullptr).
template <typename T>
inline bool device_pointer(T*& p, std::string& why)
{
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, (const void*)p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    char buf[160];
    std::snprintf(buf, sizeof buf,
                  "Encountered Hip Error: %d: '%s': Failed to get pointer "
                  "attributes for pointer: %p.",
                  (int)e, hipGetErrorString(e), (const void*)p);
    why = buf;
    return false;
  }
  if (!attr.devicePointer) {
    char buf[96];
    std::snprintf(buf, sizeof buf,
                  "Memory location is not accessible by the current GPU: %p",
                  (const void*)p);
    why = buf;
    return false;
  }
  p = reinterpret_cast<T*>(attr.devicePointer);
  return true;
}

inline bool launch_ok(const char* what, std::string& why)
{
  hipError_t e = hipGetLastError();
  if (e == hipSuccess)
    return true;
  char buf[200];
  std::snprintf(buf, sizeof buf, "Encountered Hip Error: %d: '%s': %s.", (int)e,
                hipGetErrorString(e), what);
  why = buf;
  return false;
}

} // namespace hcamd

#define HCAMD_REQUIRE_NOT_NULL(fn, p)                                          \
  do {                                                                         \
    if ((p) == nullptr)                                                        \
      return ::hcamd::fail(fn, "'" #p "' must not be null.");                  \
  } while (0)

#define HCAMD_DEVICE_POINTER(fn, p)                                            \
  do {                                                                         \
    std::string why__;                                                         \
    if (!::hcamd::device_pointer(p, why__))                                    \
      return ::hcamd::fail(fn, why__);                                         \
  } while (0)
