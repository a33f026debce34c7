/*
 * hipcomp/hipcompManagerFactory.hpp -- the manager that reads a given container (reference
 * include/hipcomp/hipcompManagerFactory.hpp:50-62, src/highlevel/hipcompManagerFactory.cpp):
 * looks at the container's headers (copies them to the host: synchronises the stream) and
 * constructs the LZ4, Snappy or Cascaded manager with the parameters recorded there.
 * Containers of the closed formats (ANS, GDeflate, Bitcomp) are refused with an exception.
 */
#ifndef HIPCOMP_MANAGER_FACTORY_HPP
#define HIPCOMP_MANAGER_FACTORY_HPP

#include "cascaded.hpp"
#include "hipcompManager.hpp"
#include "lz4.hpp"
#include "snappy.hpp"

namespace hipcomp
{

std::shared_ptr<hipcompManagerBase> create_manager(
    const uint8_t* comp_buffer, hipStream_t stream = 0, const int device_id = 0);

} // namespace hipcomp

#endif
