// hlif.hip -- the high-level interface (include/hipcomp/hipcompManager.hpp, lz4.hpp, snappy.hpp,
// cascaded.hpp, hipcompManagerFactory.hpp, hlif.h): one container per buffer, written and read
// on the device over the batched kernels of the three codecs.
//
// Reference: src/highlevel/{ManagerBase,BatchManager,LZ4Manager,SnappyManager,CascadedManager}.hpp,
// hipcompManagerFactory.cpp, the persistent-CTA
// loop of src/hipcomp_common_deps/hlif_shared.hiph:165-232 (each CTA compresses a chunk into
// its scratch slot, claims room in the container with an atomic on comp_data_size and copies
// the chunk there: chunk data in completion order) and :293-345.
// Here: the same scheme inside the batched encoders of all three codecs -- a wave compresses into a slot of its
// own (one per RESIDENT wave, not per chunk), takes the chunk's room in the container with an atomic add on
// comp_data_size when the size is known, and copies it there (placement.hpp); an LZ4 chunk that ends in one
// long literal run (data that does not compress) takes its room before that run is written and writes it in
// place, once (lz4_common.hiph: reserve_place).  Chunk data in completion order, as in the reference.  No
// stream, event or thread of the manager's own: everything runs on the caller's stream.  The scratch space is a
// constant of the manager, as in the reference.
#include "cascaded_launch.hpp"
#include "host_common.hpp"
#include "lz4_launch.hpp"
#include "snappy_launch.hpp"
#include "wave_utils.hpp"

#include "hipcomp/cascaded.h"
#include "hipcomp/cascaded.hpp"
#include "hipcomp/hipcompManagerFactory.hpp"
#include "hipcomp/hlif.h"
#include "hipcomp/lz4.h"
#include "hipcomp/lz4.hpp"
#include "hipcomp/snappy.h"
#include "hipcomp/snappy.hpp"

#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

namespace hcamd {
namespace hlif {

// reference src/hipcomp_common_deps/hlif_shared_types.hpp:58-84 (layout by the C++ ABI: 64 bytes)
enum FormatType : uint8_t { kLZ4 = 0, kSnappy = 1, kANS = 2, kGDeflate = 3, kCascaded = 4, kBitcomp = 5 };
struct CommonHeader
{
  uint32_t magic_number;
  uint8_t major_version;
  uint8_t minor_version;
  uint8_t format;
  uint64_t comp_data_size;
  uint64_t decomp_data_size;
  uint64_t num_chunks;
  bool include_chunk_starts;
  uint32_t full_comp_buffer_checksum;
  uint32_t decomp_buffer_checksum;
  bool include_per_chunk_comp_buffer_checksums;
  bool include_per_chunk_decomp_buffer_checksums;
  uint64_t uncomp_chunk_size;
  uint32_t comp_data_offset;
};
static_assert(sizeof(CommonHeader) == 64, "container header layout");
static_assert(offsetof(CommonHeader, comp_data_size) == 8 && offsetof(CommonHeader, num_chunks) == 24
                  && offsetof(CommonHeader, full_comp_buffer_checksum) == 36
                  && offsetof(CommonHeader, uncomp_chunk_size) == 48 && offsetof(CommonHeader, comp_data_offset) == 56,
              "container header layout");

// format headers behind the common one (reference include/hipcomp/{lz4,snappy,cascaded}.hpp):
// LZ4 {hipcompType_t} 4 bytes, Snappy {} 1 byte (an empty struct), Cascaded {options} 24 bytes
constexpr size_t kMaxFormatHeaderBytes = 24;
struct FormatHeader
{
  uint8_t bytes[kMaxFormatHeaderBytes];
};

// where the arrays and the chunk data of a container of n chunks start (bytes from its start;
// the reference aligns the ADDRESS behind the two headers to 8 -- containers are at least
// 8-byte aligned, so that is an offset)
struct Layout
{
  size_t offsets, sizes, comp_checksums, decomp_checksums, data;
};
inline Layout layout_of(size_t n, size_t format_header_bytes)
{
  Layout l;
  l.offsets = (sizeof(CommonHeader) + format_header_bytes + 7) & ~size_t(7);
  l.sizes = l.offsets + 8 * n;
  l.comp_checksums = l.sizes + 8 * n;
  l.decomp_checksums = l.comp_checksums + 4 * n;
  l.data = l.decomp_checksums + 4 * n;
  return l;
}

constexpr int kBlock = 256;

// ---- compression ------------------------------------------------------------------------
__global__ void header_kernel(
    uint8_t* container, uint64_t decomp_bytes, uint64_t num_chunks, uint64_t chunk_bytes, uint32_t data_offset,
    uint8_t format, FormatHeader format_header, uint32_t format_header_bytes, hipcompStatus_t* status)
{
  CommonHeader* h = reinterpret_cast<CommonHeader*>(container);
  h->magic_number = 0; // reference fill_common_header, hlif_shared.hiph:113-131
  h->major_version = 2;
  h->minor_version = 2;
  h->format = format;
  h->comp_data_size = 0;
  h->decomp_data_size = decomp_bytes;
  h->num_chunks = num_chunks;
  h->include_chunk_starts = true;
  h->full_comp_buffer_checksum = 0;
  h->decomp_buffer_checksum = 0;
  h->include_per_chunk_comp_buffer_checksums = false;
  h->include_per_chunk_decomp_buffer_checksums = false;
  h->uncomp_chunk_size = chunk_bytes;
  h->comp_data_offset = data_offset;
  for (uint32_t i = 0; i < format_header_bytes; ++i)
    container[sizeof(CommonHeader) + i] = format_header.bytes[i];
  if (status)
    *status = hipcompSuccess;
}

// chunk list of one pass: inputs are slices of the buffer; the pass's chunk ticket counter = 0 (the Snappy /
// Cascaded kernels draw their chunks from it; zeroed here, by a kernel on the stream, not by hipMemsetAsync:
// lz4_kernels.hip, lz4_launch_decompress, says what was seen with a memset node in a captured graph)
__global__ void slab_inputs_kernel(
    const uint8_t* decomp, uint64_t decomp_bytes, uint64_t chunk_bytes, uint64_t first, uint32_t count,
    const uint8_t** in_ptrs, size_t* in_bytes, uint32_t* ticket)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i == 0)
    *ticket = 0;
  if (i >= count)
    return;
  const uint64_t at = (first + i) * chunk_bytes;
  in_ptrs[i] = decomp + at;
  in_bytes[i] = (size_t)(decomp_bytes - at < chunk_bytes ? decomp_bytes - at : chunk_bytes);
}

// ---- decompression ------------------------------------------------------------------------
// The header is the container's own word about itself and may be corrupt or hostile: the
// chunk list of a slab is made only if it agrees with what the manager and the caller's
// configuration say (format, chunk size, chunk count, size of the output, where the data
// starts) -- otherwise every chunk of the slab gets an empty output slice and a one-byte
// stream, fails, and the verdict kernel reports hipcompErrorCannotDecompress.  Either way no
// output slice reaches past decomp_bytes.
__global__ void slab_streams_kernel(
    const uint8_t* container, uint64_t offsets_at, uint8_t* decomp, uint64_t decomp_bytes,
    uint64_t chunk_bytes, uint64_t first, uint32_t count, uint64_t num_chunks, uint32_t format, uint64_t data_at,
    const uint8_t** comp_ptrs, uint8_t** out_ptrs, size_t* caps, hipcompStatus_t* status)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count)
    return;
  const CommonHeader* h = reinterpret_cast<const CommonHeader*>(container);
  const bool header_ok = h->format == format && h->uncomp_chunk_size == chunk_bytes
                         && h->num_chunks == num_chunks && h->decomp_data_size == decomp_bytes
                         && h->comp_data_offset == data_at
                         && num_chunks == (decomp_bytes + chunk_bytes - 1) / chunk_bytes;
  const uint64_t off = reinterpret_cast<const uint64_t*>(container + offsets_at)[first + i];
  const uint64_t at = (first + i) * chunk_bytes;
  if (!header_ok) {
    comp_ptrs[i] = container;
    out_ptrs[i] = decomp;
    caps[i] = 0;
    *status = hipcompErrorCannotDecompress;
    return;
  }
  comp_ptrs[i] = container + data_at + off;
  out_ptrs[i] = decomp + at;
  caps[i] = at >= decomp_bytes ? 0 : (size_t)(decomp_bytes - at < chunk_bytes ? decomp_bytes - at : chunk_bytes);
}

// a chunk that failed, or did not fill its slice, fails the whole buffer
__global__ void slab_verdict_kernel(
    const hipcompStatus_t* statuses, const size_t* actual, const size_t* caps, uint32_t count, hipcompStatus_t* status)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < count && (statuses[i] != hipcompSuccess || actual[i] != caps[i]))
    *status = hipcompErrorCannotDecompress;
}

__global__ void set_status_kernel(hipcompStatus_t* status, hipcompStatus_t value) { *status = value; }

inline void check(hipError_t e, const char* what)
{
  if (e != hipSuccess)
    throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// The status word of a configuration (pinned host memory: kernels write it, the host reads it).  A
// configuration is made per call in the reference's API; hipHostMalloc + hipHostFree per call were a fifth of a
// 3 ms compress.  Words come from pages of 64 that are handed out again and never given back (a process that
// configures holds a few hundred bytes of pinned memory until it ends; the list is leaked on purpose: a
// configuration in a static of the caller's may outlive this library's own statics).
struct StatusPool
{
  std::mutex lock;
  std::vector<hipcompStatus_t*> spare;
  hipcompStatus_t* take()
  {
    std::lock_guard<std::mutex> g(lock);
    if (spare.empty()) {
      hipcompStatus_t* page = nullptr;
      check(hipHostMalloc((void**)&page, 64 * sizeof(hipcompStatus_t), hipHostMallocDefault), "hipHostMalloc(status)");
      for (int i = 0; i < 64; ++i)
        spare.push_back(page + i);
    }
    hipcompStatus_t* p = spare.back();
    spare.pop_back();
    return p;
  }
  void give(hipcompStatus_t* p)
  {
    std::lock_guard<std::mutex> g(lock);
    spare.push_back(p);
  }
};
inline StatusPool& status_pool()
{
  static StatusPool* pool = new StatusPool;
  return *pool;
}
inline std::shared_ptr<hipcompStatus_t> new_status()
{
  hipcompStatus_t* p = status_pool().take();
  *p = hipcompSuccess;
  return std::shared_ptr<hipcompStatus_t>(p, [](hipcompStatus_t* q) { status_pool().give(q); });
}

// What the three managers share: the container, the slabs, the scratch space.  The codec is
// a small set of hooks.
struct Core
{
  enum Codec { LZ4, Snappy, Cascaded } codec;
  size_t chunk_bytes = 0;
  hipStream_t stream = nullptr;
  size_t slot_bytes = 0;   // hipcompBatched*CompressGetMaxOutputChunkSize(chunk_bytes), 16-byte multiple
  uint32_t slab = 0;       // chunks per pass of decompress
  uint8_t format = kLZ4;
  FormatHeader format_header = {};
  uint32_t format_header_bytes = 0;
  uint32_t place_align = 1; // chunk starts in the container
  // LZ4
  hipcompType_t lz4_type = HIPCOMP_TYPE_CHAR;
  int lz4_elem = 1;
  uint32_t ht_size = 0;
  // Cascaded
  hipcompBatchedCascadedOpts_t cascaded_opts = {};
  int cascaded_elem = 4;

  uint8_t* scratch = nullptr;
  bool own_scratch = false;
  CommonHeader* header_host = nullptr; // pinned
  Core(Codec c, size_t chunk, hipStream_t st, int device_id, const char* who) : codec(c), chunk_bytes(chunk), stream(st)
  {
    int dev = -1;
    check(hipGetDevice(&dev), "hipGetDevice");
    if (dev != device_id)
      throw std::runtime_error(std::string(who) + ": device_id " + std::to_string(device_id) + " is not the current device");
    check(hipHostMalloc((void**)&header_host, sizeof(CommonHeader), hipHostMallocDefault), "hipHostMalloc(header)");
  }
  ~Core()
  {
    if (own_scratch)
      (void)hipFree(scratch);
    (void)hipHostFree(header_host);
  }
  Core(const Core&) = delete;
  Core& operator=(const Core&) = delete;

  void finish_init(size_t max_compressed_chunk)
  {
    slot_bytes = (max_compressed_chunk + 15) & ~size_t(15);
    slab = (uint32_t)kPlacedSlab;
  }

  Layout layout(size_t n) const { return layout_of(n, format_header_bytes); }

  // decompress: the chunk lists of a pass (and 64 spare bytes: the LZ4 decoder's ticket counter)
  size_t lists_bytes() const { return (size_t)slab * (8 + 8 + 8 + 8 + 8 + 4) + 64; }
  // compress: the encoders place the chunks themselves (placement.hpp) -- a slot per resident wave, no scan, no
  // gather, no second stream; a pass takes up to kPlacedSlab chunks (what bounds it is the chunk lists and the
  // LZ4 launcher's routing lists, 40 bytes per chunk).  Scratch: the pass's input list, 64 spare bytes (the
  // Snappy / Cascaded kernels' ticket counter), the slots, and the LZ4 launcher's own temp space (its header,
  // routing lists and hash tables: lz4_launch.hpp).
  static constexpr size_t kPlacedSlab = 262144;
  size_t lz4_temp_bytes(size_t chunks) const { return codec == LZ4 ? lz4_compress_temp_bytes_used(ht_size, chunks) : 0; }
  size_t placement_slots() const
  {
    const size_t n = codec == LZ4      ? lz4_placement_slots()
                     : codec == Snappy ? snappy_placement_slots()
                                       : cascaded_placement_slots(cascaded_elem);
    if (n == 0)
      throw std::runtime_error("compress: cannot find out how many workgroups the device holds");
    return n;
  }
  size_t placed_scratch_bytes(size_t chunks) const
  {
    return 16 * chunks + 64 + placement_slots() * slot_bytes + 16 + lz4_temp_bytes(chunks);
  }
  // what a caller's own scratch buffer must hold
  size_t scratch_bytes() const
  {
    const size_t a = placed_scratch_bytes(kPlacedSlab), b = lists_bytes();
    return a > b ? a : b;
  }
  // The manager's own scratch is as large as the calls so far needed (a buffer of a few chunks does not
  // pay for the lists of a full pass); the caller's is scratch_bytes() by contract.
  size_t own_capacity = 0;
  uint8_t* ensure_scratch(size_t need)
  {
    if (scratch && !own_scratch)
      return scratch;
    if (!scratch || own_capacity < need) {
      if (scratch)
        (void)hipFree(scratch); // (waits for whatever still uses it)
      scratch = nullptr;
      own_capacity = 0;
      check(hipMalloc((void**)&scratch, need), "hipMalloc(scratch)");
      own_scratch = true;
      own_capacity = need;
    }
    return scratch;
  }

  hipcomp::CompressionConfig configure_compression(size_t decomp_buffer_size) const
  {
    hipcomp::CompressionConfig c(decomp_buffer_size);
    c.num_chunks = (decomp_buffer_size + chunk_bytes - 1) / chunk_bytes;
    c.max_compressed_buffer_size = layout(c.num_chunks).data + c.num_chunks * slot_bytes;
    return c;
  }

  void compress(const uint8_t* decomp_buffer, uint8_t* comp_buffer, const hipcomp::CompressionConfig& cfg)
  {
    if (reinterpret_cast<uintptr_t>(comp_buffer) & 7)
      throw std::runtime_error("compress: the container buffer must be 8-byte aligned");
    const int R = cascaded_opts.num_RLEs, D = cascaded_opts.num_deltas;
    // (what hipcompBatchedCascadedCompressAsync refuses: cascaded_batch.cpp)
    if (codec == Cascaded
        && (R < 0 || D < 0 || R > 255 || D > 255
            || round_up_to(4 + 4 * (size_t)(R + 1), cascaded_elem) + round_up_to((size_t)cascaded_elem * D, 4) > 64))
      throw std::runtime_error("CascadedManager::compress: num_RLEs / num_deltas do not fit the 64-byte chunk metadata");
    const size_t n = cfg.num_chunks;
    const size_t per_pass = n < kPlacedSlab ? (n ? n : 1) : kPlacedSlab;
    uint8_t* const s = ensure_scratch(placed_scratch_bytes(per_pass));
    const Layout lay = layout(n);
    const uint8_t** in_ptrs = reinterpret_cast<const uint8_t**>(s);
    size_t* in_bytes = reinterpret_cast<size_t*>(s + per_pass * 8);
    uint32_t* const ticket = reinterpret_cast<uint32_t*>(s + per_pass * 16);
    uint8_t* const slots = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(s + per_pass * 16 + 64) + 15) & ~uintptr_t(15));
    uint8_t* lz4_temp = reinterpret_cast<uint8_t*>(
        (reinterpret_cast<uintptr_t>(slots + placement_slots() * slot_bytes) + 15) & ~uintptr_t(15));
    header_kernel<<<1, 1, 0, stream>>>(comp_buffer, cfg.uncompressed_buffer_size, n, chunk_bytes, (uint32_t)lay.data,
                                       format, format_header, format_header_bytes, cfg.get_status());
    for (size_t first = 0; first < n; first += per_pass) {
      const uint32_t count = (uint32_t)(n - first < per_pass ? n - first : per_pass);
      slab_inputs_kernel<<<(count + kBlock - 1) / kBlock, kBlock, 0, stream>>>(
          decomp_buffer, cfg.uncompressed_buffer_size, chunk_bytes, first, count, in_ptrs, in_bytes, ticket);
      Placement place;
      place.slots = slots;
      place.slot_bytes = slot_bytes;
      place.data = comp_buffer + lay.data;
      place.cursor = reinterpret_cast<unsigned long long*>(comp_buffer + offsetof(CommonHeader, comp_data_size));
      place.offsets = reinterpret_cast<unsigned long long*>(comp_buffer + lay.offsets) + first;
      place.align = place_align;
      // sizes go straight into the container's size array
      size_t* sizes = reinterpret_cast<size_t*>(comp_buffer + lay.sizes) + first;
      switch (codec) {
      case LZ4:
        check(lz4_launch_compress(in_ptrs, in_bytes, nullptr, sizes, ht_size, count, lz4_elem, lz4_temp,
                                  lz4_temp_bytes(per_pass), chunk_bytes, lz4_mode_from_environment(), stream, &place),
              "LZ4Manager::compress");
        break;
      case Snappy:
        check(snappy_launch_compress_placed(in_ptrs, in_bytes, sizes, count, ticket, place, stream),
              "SnappyManager::compress");
        break;
      case Cascaded:
        check(cascaded_launch_compress_placed(in_ptrs, in_bytes, sizes, count, (int)cascaded_opts.type, cascaded_elem, R, D,
                                              cascaded_opts.use_bp ? 1 : 0, ticket, place, stream),
              "CascadedManager::compress");
        break;
      }
    }
    check(hipGetLastError(), "compress kernels");
  }

  const CommonHeader& read_header(const uint8_t* comp_buffer)
  {
    check(hipMemcpyAsync(header_host, comp_buffer, sizeof(CommonHeader), hipMemcpyDeviceToHost, stream), "read header");
    check(hipStreamSynchronize(stream), "read header");
    return *header_host;
  }

  // A header that contradicts this manager (another format or chunk size, a chunk count that
  // is not that of its own sizes, data that does not start behind its own tables) configures
  // nothing: status hipcompErrorCannotDecompress, zero bytes, zero chunks -- decompress() then
  // launches nothing.
  bool header_fits(const CommonHeader& h) const
  {
    if (h.format != format || h.uncomp_chunk_size != chunk_bytes || chunk_bytes == 0)
      return false;
    const uint64_t want_chunks = (h.decomp_data_size + chunk_bytes - 1) / chunk_bytes;
    if (h.num_chunks != want_chunks || want_chunks > 0xFFFFFFFFull)
      return false;
    return h.comp_data_offset == layout((size_t)h.num_chunks).data;
  }

  hipcomp::DecompressionConfig configure_decompression(const uint8_t* comp_buffer)
  {
    hipcomp::DecompressionConfig d;
    const CommonHeader& h = read_header(comp_buffer);
    if (!header_fits(h)) {
      *d.get_status() = hipcompErrorCannotDecompress;
      d.decomp_data_size = 0;
      d.num_chunks = 0;
      return d;
    }
    d.decomp_data_size = (size_t)h.decomp_data_size;
    d.num_chunks = (uint32_t)h.num_chunks;
    return d;
  }

  void decompress(uint8_t* decomp_buffer, const uint8_t* comp_buffer, const hipcomp::DecompressionConfig& cfg)
  {
    uint8_t* const s = ensure_scratch(lists_bytes());
    const size_t n = cfg.num_chunks;
    const Layout lay = layout(n);
    const uint8_t** comp_ptrs = reinterpret_cast<const uint8_t**>(s);
    size_t* caps = reinterpret_cast<size_t*>(s + (size_t)slab * 8);
    uint8_t** out_ptrs = reinterpret_cast<uint8_t**>(s + (size_t)slab * 16);
    size_t* actual = reinterpret_cast<size_t*>(s + (size_t)slab * 24);
    hipcompStatus_t* statuses = reinterpret_cast<hipcompStatus_t*>(s + (size_t)slab * 40);
    // (a configuration refused by configure_decompression: nothing to do, its status stands)
    if (n == 0 && cfg.decomp_data_size == 0 && *cfg.get_status() == hipcompErrorCannotDecompress)
      return;
    set_status_kernel<<<1, 1, 0, stream>>>(cfg.get_status(), hipcompSuccess);
    for (size_t first = 0; first < n; first += slab) {
      const uint32_t count = (uint32_t)(n - first < slab ? n - first : slab);
      slab_streams_kernel<<<(count + kBlock - 1) / kBlock, kBlock, 0, stream>>>(
          comp_buffer, lay.offsets, decomp_buffer, cfg.decomp_data_size, chunk_bytes, first, count, n, format, lay.data,
          comp_ptrs, out_ptrs, caps, cfg.get_status());
      const size_t* sizes = reinterpret_cast<const size_t*>(comp_buffer + lay.sizes) + first;
      switch (codec) {
      case LZ4:
        // (the 64 spare bytes behind the chunk lists hold the decoder's chunk ticket counter)
        check(lz4_launch_decompress(comp_ptrs, sizes, caps, count, out_ptrs, actual, statuses, true, stream,
                                    s + (size_t)slab * 44, 64),
              "LZ4Manager::decompress");
        break;
      case Snappy:
        if (hipcompBatchedSnappyDecompressAsync(reinterpret_cast<const void* const*>(comp_ptrs), sizes, caps, actual, count,
                                                nullptr, 0, reinterpret_cast<void* const*>(out_ptrs), statuses, stream)
            != hipcompSuccess)
          throw std::runtime_error("SnappyManager::decompress: batched decompress failed");
        break;
      case Cascaded:
        if (hipcompBatchedCascadedDecompressAsync(reinterpret_cast<const void* const*>(comp_ptrs), sizes, caps, actual,
                                                  count, nullptr, 0, reinterpret_cast<void* const*>(out_ptrs), statuses,
                                                  stream)
            != hipcompSuccess)
          throw std::runtime_error("CascadedManager::decompress: batched decompress failed");
        break;
      }
      slab_verdict_kernel<<<(count + kBlock - 1) / kBlock, kBlock, 0, stream>>>(statuses, actual, caps, count,
                                                                                  cfg.get_status());
    }
    check(hipGetLastError(), "decompress kernels");
  }

  void set_scratch_buffer(uint8_t* new_scratch_buffer)
  {
    if (own_scratch)
      (void)hipFree(scratch);
    own_scratch = false;
    own_capacity = 0;
    scratch = new_scratch_buffer;
  }

  size_t compressed_output_size(const uint8_t* comp_buffer)
  {
    const CommonHeader& h = read_header(comp_buffer);
    return (size_t)(h.comp_data_size + h.comp_data_offset);
  }
};

hipcomp::DecompressionConfig config_of(const hipcomp::CompressionConfig& comp_config)
{
  hipcomp::DecompressionConfig d;
  d.decomp_data_size = comp_config.uncompressed_buffer_size;
  d.num_chunks = (uint32_t)comp_config.num_chunks;
  return d;
}

} // namespace hlif
} // namespace hcamd

namespace hipcomp
{
using namespace hcamd;
using namespace hcamd::hlif;

CompressionConfig::CompressionConfig(size_t n)
    : uncompressed_buffer_size(n), max_compressed_buffer_size(0), num_chunks(0), status(new_status())
{
}
hipcompStatus_t* CompressionConfig::get_status() const { return status.get(); }
DecompressionConfig::DecompressionConfig() : decomp_data_size(0), num_chunks(0), status(new_status()) {}
hipcompStatus_t* DecompressionConfig::get_status() const { return status.get(); }

// the eight virtuals of a manager, all on its core
#define HCAMD_MANAGER_METHODS(M)                                                                                    \
  M::~M() {}                                                                                                        \
  CompressionConfig M::configure_compression(const size_t n) { return impl->core.configure_compression(n); }       \
  void M::compress(const uint8_t* in, uint8_t* out, const CompressionConfig& c) { impl->core.compress(in, out, c); } \
  DecompressionConfig M::configure_decompression(const uint8_t* comp) { return impl->core.configure_decompression(comp); } \
  DecompressionConfig M::configure_decompression(const CompressionConfig& c) { return config_of(c); }              \
  void M::decompress(uint8_t* out, const uint8_t* comp, const DecompressionConfig& c) { impl->core.decompress(out, comp, c); } \
  void M::set_scratch_buffer(uint8_t* p) { impl->core.set_scratch_buffer(p); }                                      \
  size_t M::get_required_scratch_buffer_size() { return impl->core.scratch_bytes(); }                               \
  size_t M::get_compressed_output_size(uint8_t* comp) { return impl->core.compressed_output_size(comp); }

struct LZ4Manager::Impl
{
  Core core;
  Impl(size_t chunk, hipStream_t st, int dev) : core(Core::LZ4, chunk, st, dev, "LZ4Manager") {}
};
struct SnappyManager::Impl
{
  Core core;
  Impl(size_t chunk, hipStream_t st, int dev) : core(Core::Snappy, chunk, st, dev, "SnappyManager") {}
};
struct CascadedManager::Impl
{
  Core core;
  Impl(size_t chunk, hipStream_t st, int dev) : core(Core::Cascaded, chunk, st, dev, "CascadedManager") {}
};

LZ4Manager::LZ4Manager(size_t uncomp_chunk_size, hipcompType_t data_type, hipStream_t user_stream, const int device_id)
{
  if (uncomp_chunk_size == 0 || uncomp_chunk_size > (size_t(1) << 24))
    throw std::runtime_error("LZ4Manager: uncomp_chunk_size must be in [1, 16 MiB]");
  size_t slot = 0;
  if (hipcompBatchedLZ4CompressGetMaxOutputChunkSize(uncomp_chunk_size, hipcompBatchedLZ4Opts_t{data_type}, &slot)
      != hipcompSuccess)
    throw std::runtime_error("LZ4Manager: bad chunk size");
  int elem = 0;
  switch (data_type) {
  case HIPCOMP_TYPE_BITS: case HIPCOMP_TYPE_CHAR: case HIPCOMP_TYPE_UCHAR: elem = 1; break;
  case HIPCOMP_TYPE_SHORT: case HIPCOMP_TYPE_USHORT: elem = 2; break;
  case HIPCOMP_TYPE_INT: case HIPCOMP_TYPE_UINT: elem = 4; break;
  default: throw std::runtime_error("LZ4Manager: unsupported data type");
  }
  impl.reset(new Impl(uncomp_chunk_size, user_stream, device_id));
  Core& m = impl->core;
  m.format = kLZ4;
  m.format_header_bytes = sizeof(LZ4FormatSpecHeader);
  const uint32_t t = (uint32_t)data_type;
  std::memcpy(m.format_header.bytes, &t, sizeof(t));
  m.lz4_type = data_type;
  m.lz4_elem = elem;
  size_t p = 1;
  while (p < uncomp_chunk_size)
    p *= 2;
  m.ht_size = (uint32_t)(p < 16384 ? p : 16384);
  m.finish_init(slot);
}
HCAMD_MANAGER_METHODS(LZ4Manager)

SnappyManager::SnappyManager(size_t uncomp_chunk_size, hipStream_t user_stream, int device_id)
{
  size_t slot = 0;
  if (uncomp_chunk_size == 0
      || hipcompBatchedSnappyCompressGetMaxOutputChunkSize(uncomp_chunk_size, hipcompBatchedSnappyDefaultOpts, &slot)
             != hipcompSuccess)
    throw std::runtime_error("SnappyManager: bad chunk size");
  impl.reset(new Impl(uncomp_chunk_size, user_stream, device_id));
  Core& m = impl->core;
  m.format = kSnappy;
  m.format_header_bytes = sizeof(SnappyFormatSpecHeader); // 1: an empty struct
  m.finish_init(slot);
}
HCAMD_MANAGER_METHODS(SnappyManager)

// Every chunk of the container is one partition of the batched Cascaded codec, cut into the
// reference's 4096-byte sub-chunks whatever options.chunk_size says (that is the CONTAINER's
// chunk size here, as in the reference's CascadedManager.hpp:52-57), so the reference reads it.
CascadedManager::CascadedManager(const hipcompBatchedCascadedOpts_t& options, hipStream_t user_stream, int device_id)
{
  size_t slot = 0;
  if (options.chunk_size == 0
      || hipcompBatchedCascadedCompressGetMaxOutputChunkSize(options.chunk_size, hipcompBatchedCascadedDefaultOpts, &slot)
             != hipcompSuccess)
    throw std::runtime_error("CascadedManager: bad chunk size");
  size_t elem = 0;
  switch (options.type) {
  case HIPCOMP_TYPE_CHAR: case HIPCOMP_TYPE_UCHAR: elem = 1; break;
  case HIPCOMP_TYPE_SHORT: case HIPCOMP_TYPE_USHORT: elem = 2; break;
  case HIPCOMP_TYPE_INT: case HIPCOMP_TYPE_UINT: elem = 4; break;
  case HIPCOMP_TYPE_LONGLONG: case HIPCOMP_TYPE_ULONGLONG: elem = 8; break;
  default: throw std::runtime_error("CascadedManager: unsupported data type");
  }
  if (options.chunk_size % elem)
    throw std::runtime_error("CascadedManager: chunk_size must be a multiple of the element size");
  impl.reset(new Impl(options.chunk_size, user_stream, device_id));
  Core& m = impl->core;
  m.format = kCascaded;
  m.format_header_bytes = sizeof(CascadedFormatSpecHeader);
  static_assert(sizeof(CascadedFormatSpecHeader) <= kMaxFormatHeaderBytes, "format header");
  std::memcpy(m.format_header.bytes, &options, sizeof(options));
  m.cascaded_opts = options;
  m.cascaded_opts.chunk_size = 4096;
  m.cascaded_elem = (int)elem;
  m.place_align = 8;
  m.finish_init(slot);
}
HCAMD_MANAGER_METHODS(CascadedManager)

// reference hipcompManagerFactory.cpp:44-148 (synchronises the stream, as there)
std::shared_ptr<hipcompManagerBase> create_manager(const uint8_t* comp_buffer, hipStream_t stream, const int device_id)
{
  struct Heads
  {
    CommonHeader common;
    FormatHeader format;
  } heads;
  check(hipMemcpyAsync(&heads, comp_buffer, sizeof(heads), hipMemcpyDeviceToHost, stream), "create_manager: read headers");
  check(hipStreamSynchronize(stream), "create_manager: read headers");
  switch (heads.common.format) {
  case kLZ4: {
    uint32_t t;
    std::memcpy(&t, heads.format.bytes, sizeof(t));
    return std::make_shared<LZ4Manager>((size_t)heads.common.uncomp_chunk_size, (hipcompType_t)t, stream, device_id);
  }
  case kSnappy:
    return std::make_shared<SnappyManager>((size_t)heads.common.uncomp_chunk_size, stream, device_id);
  case kCascaded: {
    hipcompBatchedCascadedOpts_t o;
    std::memcpy(&o, heads.format.bytes, sizeof(o));
    if (o.chunk_size != heads.common.uncomp_chunk_size)
      throw std::runtime_error("create_manager: Cascaded options do not match the container's chunk size");
    return std::make_shared<CascadedManager>(o, stream, device_id);
  }
  default:
    throw std::runtime_error("create_manager: format " + std::to_string((int)heads.common.format)
                             + " is not supported (LZ4, Snappy and Cascaded are)");
  }
}

} // namespace hipcomp

// ---- C binding -----------------------------------------------------------------------------
struct hipcompHlifManager
{
  std::shared_ptr<hipcomp::hipcompManagerBase> lz4; // (any of the managers)
  hipStream_t stream = nullptr;
  std::unique_ptr<hipcomp::CompressionConfig> last_comp;
  std::unique_ptr<hipcomp::DecompressionConfig> last_decomp;
  bool last_was_compress = true;
};

namespace {
template <typename F>
hipcompStatus_t guarded(const char* fn, F&& f)
{
  try {
    f();
    return hipcompSuccess;
  } catch (const std::exception& e) {
    return hcamd::fail(fn, e.what());
  }
}
} // namespace

extern "C" {

hipcompStatus_t hipcompHlifLZ4ManagerCreate(
    size_t uncomp_chunk_size, hipcompType_t data_type, hipStream_t stream, hipcompHlifManager_t** manager)
{
  static const char* fn = "hipcompHlifLZ4ManagerCreate()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  return guarded(fn, [&] {
    int dev = 0;
    hcamd::hlif::check(hipGetDevice(&dev), "hipGetDevice");
    std::unique_ptr<hipcompHlifManager> h(new hipcompHlifManager);
    h->lz4 = std::make_shared<hipcomp::LZ4Manager>(uncomp_chunk_size, data_type, stream, dev);
    h->stream = stream;
    *manager = h.release();
  });
}

hipcompStatus_t hipcompHlifSnappyManagerCreate(size_t uncomp_chunk_size, hipStream_t stream, hipcompHlifManager_t** manager)
{
  static const char* fn = "hipcompHlifSnappyManagerCreate()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  return guarded(fn, [&] {
    int dev = 0;
    hcamd::hlif::check(hipGetDevice(&dev), "hipGetDevice");
    std::unique_ptr<hipcompHlifManager> h(new hipcompHlifManager);
    h->lz4 = std::make_shared<hipcomp::SnappyManager>(uncomp_chunk_size, stream, dev);
    h->stream = stream;
    *manager = h.release();
  });
}

hipcompStatus_t hipcompHlifCascadedManagerCreate(
    hipcompBatchedCascadedOpts_t options, hipStream_t stream, hipcompHlifManager_t** manager)
{
  static const char* fn = "hipcompHlifCascadedManagerCreate()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  return guarded(fn, [&] {
    int dev = 0;
    hcamd::hlif::check(hipGetDevice(&dev), "hipGetDevice");
    std::unique_ptr<hipcompHlifManager> h(new hipcompHlifManager);
    h->lz4 = std::make_shared<hipcomp::CascadedManager>(options, stream, dev);
    h->stream = stream;
    *manager = h.release();
  });
}

hipcompStatus_t hipcompHlifManagerCreateFromContainer(
    const void* device_container, hipStream_t stream, hipcompHlifManager_t** manager)
{
  static const char* fn = "hipcompHlifManagerCreateFromContainer()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  HCAMD_REQUIRE_NOT_NULL(fn, device_container);
  return guarded(fn, [&] {
    int dev = 0;
    hcamd::hlif::check(hipGetDevice(&dev), "hipGetDevice");
    std::unique_ptr<hipcompHlifManager> h(new hipcompHlifManager);
    h->lz4 = hipcomp::create_manager(static_cast<const uint8_t*>(device_container), stream, dev);
    h->stream = stream;
    *manager = h.release();
  });
}

hipcompStatus_t hipcompHlifManagerDestroy(hipcompHlifManager_t* manager)
{
  delete manager;
  return hipcompSuccess;
}

hipcompStatus_t hipcompHlifConfigureCompression(
    hipcompHlifManager_t* manager, size_t uncompressed_bytes, size_t* max_compressed_bytes, size_t* num_chunks)
{
  static const char* fn = "hipcompHlifConfigureCompression()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  HCAMD_REQUIRE_NOT_NULL(fn, max_compressed_bytes);
  return guarded(fn, [&] {
    const hipcomp::CompressionConfig c = manager->lz4->configure_compression(uncompressed_bytes);
    *max_compressed_bytes = c.max_compressed_buffer_size;
    if (num_chunks)
      *num_chunks = c.num_chunks;
  });
}

hipcompStatus_t hipcompHlifCompress(
    hipcompHlifManager_t* manager, const void* device_uncompressed, size_t uncompressed_bytes, void* device_container)
{
  static const char* fn = "hipcompHlifCompress()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  HCAMD_REQUIRE_NOT_NULL(fn, device_container);
  return guarded(fn, [&] {
    // (a configuration owns a pinned status word: made anew only when the size changes -- hipHostMalloc /
    // hipHostFree per call were a fifth of a 3 ms compress)
    if (!manager->last_comp || manager->last_comp->uncompressed_buffer_size != uncompressed_bytes)
      manager->last_comp.reset(new hipcomp::CompressionConfig(manager->lz4->configure_compression(uncompressed_bytes)));
    manager->last_was_compress = true;
    manager->lz4->compress(static_cast<const uint8_t*>(device_uncompressed), static_cast<uint8_t*>(device_container),
                           *manager->last_comp);
  });
}

hipcompStatus_t hipcompHlifGetDecompressedSize(
    hipcompHlifManager_t* manager, const void* device_container, size_t* uncompressed_bytes, size_t* num_chunks)
{
  static const char* fn = "hipcompHlifGetDecompressedSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  HCAMD_REQUIRE_NOT_NULL(fn, device_container);
  HCAMD_REQUIRE_NOT_NULL(fn, uncompressed_bytes);
  return guarded(fn, [&] {
    const hipcomp::DecompressionConfig d = manager->lz4->configure_decompression(static_cast<const uint8_t*>(device_container));
    *uncompressed_bytes = d.decomp_data_size;
    if (num_chunks)
      *num_chunks = d.num_chunks;
  });
}

hipcompStatus_t hipcompHlifGetCompressedSize(
    hipcompHlifManager_t* manager, const void* device_container, size_t* container_bytes)
{
  static const char* fn = "hipcompHlifGetCompressedSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  HCAMD_REQUIRE_NOT_NULL(fn, device_container);
  HCAMD_REQUIRE_NOT_NULL(fn, container_bytes);
  return guarded(fn, [&] {
    *container_bytes = manager->lz4->get_compressed_output_size(static_cast<uint8_t*>(const_cast<void*>(device_container)));
  });
}

hipcompStatus_t hipcompHlifDecompress(hipcompHlifManager_t* manager, const void* device_container, void* device_uncompressed)
{
  static const char* fn = "hipcompHlifDecompress()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  HCAMD_REQUIRE_NOT_NULL(fn, device_container);
  return guarded(fn, [&] {
    manager->last_decomp.reset(new hipcomp::DecompressionConfig(
        manager->lz4->configure_decompression(static_cast<const uint8_t*>(device_container))));
    manager->last_was_compress = false;
    manager->lz4->decompress(static_cast<uint8_t*>(device_uncompressed), static_cast<const uint8_t*>(device_container),
                             *manager->last_decomp);
  });
}

hipcompStatus_t hipcompHlifGetLastStatus(hipcompHlifManager_t* manager, hipcompStatus_t* status)
{
  static const char* fn = "hipcompHlifGetLastStatus()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  HCAMD_REQUIRE_NOT_NULL(fn, status);
  return guarded(fn, [&] {
    hcamd::hlif::check(hipStreamSynchronize(manager->stream), "hipStreamSynchronize");
    *status = hipcompSuccess;
    if (manager->last_was_compress && manager->last_comp)
      *status = *manager->last_comp->get_status();
    if (!manager->last_was_compress && manager->last_decomp)
      *status = *manager->last_decomp->get_status();
  });
}

hipcompStatus_t hipcompHlifGetRequiredScratchBytes(hipcompHlifManager_t* manager, size_t* scratch_bytes)
{
  static const char* fn = "hipcompHlifGetRequiredScratchBytes()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  HCAMD_REQUIRE_NOT_NULL(fn, scratch_bytes);
  *scratch_bytes = manager->lz4->get_required_scratch_buffer_size();
  return hipcompSuccess;
}

hipcompStatus_t hipcompHlifSetScratchBuffer(hipcompHlifManager_t* manager, void* device_scratch)
{
  static const char* fn = "hipcompHlifSetScratchBuffer()";
  HCAMD_REQUIRE_NOT_NULL(fn, manager);
  HCAMD_REQUIRE_NOT_NULL(fn, device_scratch);
  manager->lz4->set_scratch_buffer(static_cast<uint8_t*>(device_scratch));
  return hipcompSuccess;
}

} // extern "C"
