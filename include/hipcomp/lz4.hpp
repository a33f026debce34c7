/*
 * hipcomp/lz4.hpp -- LZ4 manager of the high-level interface (reference
 * include/hipcomp/lz4.hpp:58-70): the input is cut into chunks of uncomp_chunk_size bytes,
 * each compressed as an LZ4 block with elements of data_type (hipcomp/lz4.h).
 */
#ifndef HIPCOMP_LZ4_HPP
#define HIPCOMP_LZ4_HPP

#include "hipcompManager.hpp"

namespace hipcomp
{

struct LZ4FormatSpecHeader
{
  hipcompType_t data_type;
};

struct LZ4Manager : hipcompManagerBase
{
  /* uncomp_chunk_size at most 16 MiB (the LZ4 block limit); device_id must be the current device */
  LZ4Manager(size_t uncomp_chunk_size, hipcompType_t data_type, hipStream_t user_stream = 0, const int device_id = 0);
  ~LZ4Manager() override;
  LZ4Manager(const LZ4Manager&) = delete;
  LZ4Manager& operator=(const LZ4Manager&) = delete;

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
