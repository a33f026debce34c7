/*
 * hipcomp/cascaded.hpp -- Cascaded manager of the high-level interface (reference
 * include/hipcomp/cascaded.hpp:55-68): the input is cut into chunks of options.chunk_size
 * bytes (the CONTAINER's chunk size; 4096 by default), each compressed as one partition of
 * the batched Cascaded codec with options.{type, num_RLEs, num_deltas, use_bp}
 * (hipcomp/cascaded.h).  The input buffer has to be aligned like its elements.
 */
#ifndef HIPCOMP_CASCADED_HPP
#define HIPCOMP_CASCADED_HPP

#include "cascaded.h"
#include "hipcompManager.hpp"

namespace hipcomp
{

struct CascadedFormatSpecHeader
{
  hipcompBatchedCascadedOpts_t options;
};

struct CascadedManager : hipcompManagerBase
{
  /* device_id must be the current device */
  CascadedManager(
      const hipcompBatchedCascadedOpts_t& options = hipcompBatchedCascadedDefaultOpts, hipStream_t user_stream = 0,
      int device_id = 0);
  ~CascadedManager() override;
  CascadedManager(const CascadedManager&) = delete;
  CascadedManager& operator=(const CascadedManager&) = delete;

  CompressionConfig configure_compression(const size_t decomp_buffer_size) override;
  void compress(const uint8_t* decomp_buffer, uint8_t* comp_buffer, const CompressionConfig& comp_config) override;
  DecompressionConfig configure_decompression(const uint8_t* comp_buffer) override;
  DecompressionConfig configure_decompression(const CompressionConfig& comp_config) override;
  void decompress(uint8_t* decomp_buffer, const uint8_t* comp_buffer, const DecompressionConfig& decomp_config) override;
  void set_scratch_buffer(uint8_t* new_scratch_buffer) override;
  size_t get_required_scratch_buffer_size() override;
  size_t get_compressed_output_size(uint8_t* comp_buffer) override;

private:
  struct Impl;
  std::unique_ptr<Impl> impl;
};

} // namespace hipcomp

#endif
