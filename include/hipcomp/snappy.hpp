/*
 * hipcomp/snappy.hpp -- Snappy manager of the high-level interface (reference
 * include/hipcomp/snappy.hpp:58-69): the input is cut into chunks of uncomp_chunk_size
 * bytes, each compressed as a raw Snappy block (hipcomp/snappy.h).
 */
#ifndef HIPCOMP_SNAPPY_HPP
#define HIPCOMP_SNAPPY_HPP

#include "hipcompManager.hpp"

namespace hipcomp
{

struct SnappyFormatSpecHeader
{
  /* empty, as in the reference: one byte of the container */
};

struct SnappyManager : hipcompManagerBase
{
  /* device_id must be the current device */
  SnappyManager(size_t uncomp_chunk_size, hipStream_t user_stream = 0, int device_id = 0);
  ~SnappyManager() override;
  SnappyManager(const SnappyManager&) = delete;
  SnappyManager& operator=(const SnappyManager&) = delete;

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
