/*
 * hipcomp/hipcompManager.hpp -- the high-level interface ("HLIF"): one call compresses a
 * whole device buffer into ONE self-describing container, one call decompresses it.
 *
 * Same public surface as the reference (include/hipcomp/hipcompManager.hpp:50-303):
 * CompressionConfig / DecompressionConfig with the same public fields and get_status(),
 * hipcompManagerBase with the same eight virtuals; same container layout
 * (src/hipcomp_common_deps/hlif_shared_types.hpp:68-84, src/highlevel/BatchManager.hpp:
 * 108-112, 245-251), so a container written here is read by the reference's managers and
 * the other way round:
 *
 *   CommonHeader (64 B) | format header | pad to 8 | chunk offsets u64 x n |
 *   chunk sizes u64 x n | 2 x checksums u32 x n (unused, as in the reference) | chunk data
 *
 * Behind it are the batched codecs of this library (the chunks of a container are exactly
 * the streams hipcompBatched*CompressAsync produces).  Differences by design: chunk data
 * lies in chunk order (the reference places chunks in completion order; both record the
 * offsets), configs own their status word instead of borrowing it from a pool, and
 * max_compressed_buffer_size counts the size array at its true width.
 */
#ifndef HIPCOMP_MANAGER_HPP
#define HIPCOMP_MANAGER_HPP

#include "hipcomp.h"

#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>
#include <memory>

namespace hipcomp
{

struct CompressionConfig
{
  size_t uncompressed_buffer_size;
  size_t max_compressed_buffer_size;
  size_t num_chunks;
  explicit CompressionConfig(size_t uncompressed_buffer_size);
  /* the status word of the operation: pinned host memory the device writes (read it after
     synchronising the stream) */
  hipcompStatus_t* get_status() const;

private:
  std::shared_ptr<hipcompStatus_t> status;
};

struct DecompressionConfig
{
  size_t decomp_data_size;
  uint32_t num_chunks;
  DecompressionConfig();
  hipcompStatus_t* get_status() const;

private:
  std::shared_ptr<hipcompStatus_t> status;
};

struct hipcompManagerBase
{
  /* sizes the result buffer for an input of decomp_buffer_size bytes */
  virtual CompressionConfig configure_compression(const size_t decomp_buffer_size) = 0;
  /* asynchronous on the manager's stream; both buffers device-accessible */
  virtual void compress(const uint8_t* decomp_buffer, uint8_t* comp_buffer, const CompressionConfig& comp_config) = 0;
  /* reads the container's header (synchronises the stream) */
  virtual DecompressionConfig configure_decompression(const uint8_t* comp_buffer) = 0;
  /* from the config the buffer was compressed with (no synchronisation) */
  virtual DecompressionConfig configure_decompression(const CompressionConfig& comp_config) = 0;
  virtual void decompress(uint8_t* decomp_buffer, const uint8_t* comp_buffer, const DecompressionConfig& decomp_config) = 0;
  /* the manager allocates its scratch space on first use unless one is set */
  virtual void set_scratch_buffer(uint8_t* new_scratch_buffer) = 0;
  virtual size_t get_required_scratch_buffer_size() = 0;
  /* bytes of the container at comp_buffer (copies its header to the host) */
  virtual size_t get_compressed_output_size(uint8_t* comp_buffer) = 0;
  virtual ~hipcompManagerBase() = default;
};

} // namespace hipcomp

#endif
