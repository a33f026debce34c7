/* hipcomp.hpp -- the C++ umbrella header of the library's public interface
 * (reference include/hipcomp.hpp:49-259): the status-carrying exception the
 * C++ callers of the reference catch, the element-type mapping TypeOf<T>() and
 * the throw-on-error helper, on top of the C declarations.  Header-only,
 * written fresh; the abstract Compressor / Decompressor interfaces of the
 * reference (:93-210) have no implementation in its tree for the open formats
 * (the managers of hipcompManager.hpp superseded them) and none here either,
 * but they are declared so that code that names them keeps compiling. */
#ifndef HIPCOMP_API_HPP
#define HIPCOMP_API_HPP

#include "hipcomp.h"
#include "hipcomp/lz4.h"

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>

namespace hipcomp
{

/* reference hipcomp.hpp:65-90 */
class HipCompException : public std::runtime_error
{
public:
  HipCompException(hipcompStatus_t err, const std::string& msg)
      : std::runtime_error(msg + " : code=" + std::to_string(err) + "."), m_err(err)
  {
  }
  hipcompStatus_t get_error() const { return m_err; }

private:
  hipcompStatus_t m_err;
};

/* reference hipcomp.hpp:93-144 (interface only) */
class Compressor
{
public:
  virtual ~Compressor() = default;
  virtual void configure(const size_t in_bytes, size_t* temp_bytes, size_t* out_bytes) = 0;
  virtual void compress_async(
      const void* in_ptr, const size_t in_bytes, void* temp_ptr, const size_t temp_bytes, void* out_ptr,
      size_t* out_bytes, hipStream_t stream)
      = 0;
};

/* reference hipcomp.hpp:150-210 (interface only) */
class Decompressor
{
public:
  virtual ~Decompressor() = default;
  virtual void configure(
      const void* in_ptr, const size_t in_bytes, size_t* temp_bytes, size_t* out_bytes, hipStream_t stream)
      = 0;
  virtual void decompress_async(
      const void* in_ptr, const size_t in_bytes, void* temp_ptr, const size_t temp_bytes, void* out_ptr,
      const size_t out_bytes, hipStream_t stream)
      = 0;
};

/* the hipcompType_t of an integer type (reference hipcomp.hpp:217-241); anything else throws */
template <typename T>
inline hipcompType_t TypeOf()
{
  if (std::is_same<T, int8_t>::value)
    return HIPCOMP_TYPE_CHAR;
  if (std::is_same<T, uint8_t>::value)
    return HIPCOMP_TYPE_UCHAR;
  if (std::is_same<T, int16_t>::value)
    return HIPCOMP_TYPE_SHORT;
  if (std::is_same<T, uint16_t>::value)
    return HIPCOMP_TYPE_USHORT;
  if (std::is_same<T, int32_t>::value)
    return HIPCOMP_TYPE_INT;
  if (std::is_same<T, uint32_t>::value)
    return HIPCOMP_TYPE_UINT;
  if (std::is_same<T, int64_t>::value)
    return HIPCOMP_TYPE_LONGLONG;
  if (std::is_same<T, uint64_t>::value)
    return HIPCOMP_TYPE_ULONGLONG;
  throw HipCompException(hipcompErrorNotSupported, "hipcomp does not support the given type.");
}

/* reference hipcomp.hpp:250-255 */
inline void throwExceptionIfError(hipcompStatus_t error, const std::string& msg)
{
  if (error != hipcompSuccess)
    throw HipCompException(error, msg);
}

} // namespace hipcomp

#endif
