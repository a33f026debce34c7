// hlif.hip -- the high-level interface (include/hipcomp/hipcompManager.hpp, lz4.hpp, hlif.h):
// one container per buffer, written and read on the device over the batched LZ4 kernels.
//
// Reference: src/highlevel/{ManagerBase,BatchManager,LZ4Manager}.hpp, the persistent-CTA
// loop of src/hipcomp_common_deps/hlif_shared.hiph:165-232 (each CTA compresses a chunk into
// its scratch slot, claims room in the container with an atomic on comp_data_size and copies
// the chunk there: chunk data in completion order) and :293-345.  Here the chunk list goes
// through the batched kernels a slab of chunks at a time: compress the slab into scratch
// slots, scan the slab's sizes on top of comp_data_size, copy each chunk to its offset --
// chunk data in chunk order, the same header, offsets and sizes arrays.  The scratch space is
// one slab (a constant of the manager, as in the reference).
#include "host_common.hpp"
#include "lz4_launch.hpp"
#include "wave_utils.hpp"

#include "hipcomp/hlif.h"
#include "hipcomp/lz4.h"
#include "hipcomp/lz4.hpp"

#include <new>
#include <stdexcept>
#include <string>

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

constexpr size_t kFormatHeaderBytes = 4; // LZ4FormatSpecHeader

// where the arrays and the chunk data of a container of n chunks start (bytes from its start;
// the reference aligns the ADDRESS behind the two headers to 8 -- containers are at least
// 8-byte aligned, so that is an offset)
struct Layout
{
  size_t offsets, sizes, comp_checksums, decomp_checksums, data;
};
inline Layout layout_of(size_t n)
{
  Layout l;
  l.offsets = (sizeof(CommonHeader) + kFormatHeaderBytes + 7) & ~size_t(7);
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
    uint32_t data_type, hipcompStatus_t* status)
{
  CommonHeader* h = reinterpret_cast<CommonHeader*>(container);
  h->magic_number = 0; // reference fill_common_header, hlif_shared.hiph:113-131
  h->major_version = 2;
  h->minor_version = 2;
  h->format = kLZ4;
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
  *reinterpret_cast<uint32_t*>(container + sizeof(CommonHeader)) = data_type;
  if (status)
    *status = hipcompSuccess;
}

// chunk list of one slab: inputs are slices of the buffer, outputs the scratch slots
__global__ void slab_inputs_kernel(
    const uint8_t* decomp, uint64_t decomp_bytes, uint64_t chunk_bytes, uint64_t first, uint32_t count,
    uint8_t* slots, uint64_t slot_bytes, const uint8_t** in_ptrs, size_t* in_bytes, uint8_t** out_ptrs)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count)
    return;
  const uint64_t at = (first + i) * chunk_bytes;
  in_ptrs[i] = decomp + at;
  in_bytes[i] = (size_t)(decomp_bytes - at < chunk_bytes ? decomp_bytes - at : chunk_bytes);
  out_ptrs[i] = slots + (uint64_t)i * slot_bytes;
}

// one workgroup: offsets of the slab's chunks = running total of the container + exclusive
// scan of their sizes; the running total moves on
__global__ __launch_bounds__(kBlock) void slab_place_kernel(
    uint8_t* container, uint64_t sizes_at, uint64_t offsets_at, uint64_t first, uint32_t count)
{
  __shared__ uint64_t wave_sums[kBlock / 64];
  CommonHeader* h = reinterpret_cast<CommonHeader*>(container);
  const uint64_t* sizes = reinterpret_cast<const uint64_t*>(container + sizes_at) + first;
  uint64_t* offsets = reinterpret_cast<uint64_t*>(container + offsets_at) + first;
  uint64_t carry = h->comp_data_size;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t i0 = 0; i0 < count; i0 += kBlock) {
    const uint32_t i = i0 + threadIdx.x;
    const uint64_t v = i < count ? sizes[i] : 0;
    const uint64_t incl = wave_scan_add_u64(v);
    if (lane == 63)
      wave_sums[wave] = incl;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (int w = 0; w < kBlock / 64; ++w) {
      before += w < wave ? wave_sums[w] : 0;
      all += wave_sums[w];
    }
    __syncthreads();
    if (i < count)
      offsets[i] = carry + before + incl - v;
    carry += all;
  }
  if (threadIdx.x == 0)
    h->comp_data_size = carry;
}

// one wave per chunk: scratch slot -> its place in the container
__global__ __launch_bounds__(kBlock) void slab_gather_kernel(
    uint8_t* container, uint64_t sizes_at, uint64_t offsets_at, uint64_t data_at, uint64_t first, uint32_t count,
    const uint8_t* slots, uint64_t slot_bytes)
{
  const uint32_t i = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (i >= count)
    return;
  const uint64_t size = reinterpret_cast<const uint64_t*>(container + sizes_at)[first + i];
  const uint64_t off = reinterpret_cast<const uint64_t*>(container + offsets_at)[first + i];
  wave_copy(to_global(container + data_at + off), to_global(slots + (uint64_t)i * slot_bytes), (uint32_t)size,
            (int)(threadIdx.x & 63));
}

// ---- decompression ------------------------------------------------------------------------
__global__ void slab_streams_kernel(
    const uint8_t* container, uint64_t offsets_at, uint8_t* decomp, uint64_t decomp_bytes,
    uint64_t chunk_bytes, uint64_t first, uint32_t count, const uint8_t** comp_ptrs, uint8_t** out_ptrs, size_t* caps)
{
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count)
    return;
  // the chunk data starts where the header says (a reference-written container's too)
  const uint64_t data_at = reinterpret_cast<const CommonHeader*>(container)->comp_data_offset;
  const uint64_t off = reinterpret_cast<const uint64_t*>(container + offsets_at)[first + i];
  const uint64_t at = (first + i) * chunk_bytes;
  comp_ptrs[i] = container + data_at + off;
  out_ptrs[i] = decomp + at;
  caps[i] = (size_t)(decomp_bytes - at < chunk_bytes ? decomp_bytes - at : chunk_bytes);
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

inline std::shared_ptr<hipcompStatus_t> new_status()
{
  hipcompStatus_t* p = nullptr;
  check(hipHostMalloc((void**)&p, sizeof(hipcompStatus_t), hipHostMallocDefault), "hipHostMalloc(status)");
  *p = hipcompSuccess;
  return std::shared_ptr<hipcompStatus_t>(p, [](hipcompStatus_t* q) { (void)hipHostFree(q); });
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

struct LZ4Manager::Impl
{
  size_t chunk_bytes;
  hipcompType_t data_type;
  int elem;
  hipStream_t stream;
  size_t slot_bytes;   // hipcompBatchedLZ4CompressGetMaxOutputChunkSize(chunk_bytes)
  uint32_t ht_size;
  uint32_t slab;       // chunks per pass
  uint8_t* scratch = nullptr;
  bool own_scratch = false;
  CommonHeader* header_host = nullptr; // pinned

  // scratch: chunk lists of a slab, three words for the compress kernels, the
  // slots, hash tables for the compress kernel's "far" shape (one per chunk of a slab)
  size_t lists_bytes() const { return (size_t)slab * (8 + 8 + 8 + 8 + 8 + 4) + 64; }
  size_t table_bytes() const { return (size_t)(ht_size < 8 ? 8 : ht_size) * sizeof(uint16_t); }
  size_t scratch_bytes() const { return lists_bytes() + (size_t)slab * slot_bytes + 16 + (size_t)slab * table_bytes(); }
  uint8_t* ensure_scratch()
  {
    if (!scratch) {
      check(hipMalloc((void**)&scratch, scratch_bytes()), "hipMalloc(scratch)");
      own_scratch = true;
    }
    return scratch;
  }
};

LZ4Manager::LZ4Manager(size_t uncomp_chunk_size, hipcompType_t data_type, hipStream_t user_stream, const int device_id)
    : impl(new Impl)
{
  int dev = -1;
  check(hipGetDevice(&dev), "hipGetDevice");
  if (dev != device_id)
    throw std::runtime_error("LZ4Manager: device_id " + std::to_string(device_id) + " is not the current device");
  if (uncomp_chunk_size == 0 || uncomp_chunk_size > (size_t(1) << 24))
    throw std::runtime_error("LZ4Manager: uncomp_chunk_size must be in [1, 16 MiB]");
  size_t slot = 0;
  if (hipcompBatchedLZ4CompressGetMaxOutputChunkSize(uncomp_chunk_size, hipcompBatchedLZ4Opts_t{data_type}, &slot)
      != hipcompSuccess)
    throw std::runtime_error("LZ4Manager: bad chunk size");
  switch (data_type) {
  case HIPCOMP_TYPE_BITS: case HIPCOMP_TYPE_CHAR: case HIPCOMP_TYPE_UCHAR: impl->elem = 1; break;
  case HIPCOMP_TYPE_SHORT: case HIPCOMP_TYPE_USHORT: impl->elem = 2; break;
  case HIPCOMP_TYPE_INT: case HIPCOMP_TYPE_UINT: impl->elem = 4; break;
  default: throw std::runtime_error("LZ4Manager: unsupported data type");
  }
  impl->chunk_bytes = uncomp_chunk_size;
  impl->data_type = data_type;
  impl->stream = user_stream;
  impl->slot_bytes = (slot + 15) & ~size_t(15);
  size_t p = 1;
  while (p < uncomp_chunk_size)
    p *= 2;
  impl->ht_size = (uint32_t)(p < 16384 ? p : 16384);
  // a slab: about 512 MiB of slots, 256 .. 8192 chunks
  size_t slab = (size_t(512) << 20) / impl->slot_bytes;
  impl->slab = (uint32_t)(slab < 256 ? 256 : (slab > 8192 ? 8192 : slab));
  check(hipHostMalloc((void**)&impl->header_host, sizeof(CommonHeader), hipHostMallocDefault), "hipHostMalloc(header)");
}

LZ4Manager::~LZ4Manager()
{
  if (impl->own_scratch)
    (void)hipFree(impl->scratch);
  (void)hipHostFree(impl->header_host);
}

CompressionConfig LZ4Manager::configure_compression(const size_t decomp_buffer_size)
{
  CompressionConfig c(decomp_buffer_size);
  c.num_chunks = (decomp_buffer_size + impl->chunk_bytes - 1) / impl->chunk_bytes;
  c.max_compressed_buffer_size = layout_of(c.num_chunks).data + c.num_chunks * impl->slot_bytes;
  return c;
}

void LZ4Manager::compress(const uint8_t* decomp_buffer, uint8_t* comp_buffer, const CompressionConfig& cfg)
{
  Impl& m = *impl;
  if (reinterpret_cast<uintptr_t>(comp_buffer) & 7)
    throw std::runtime_error("LZ4Manager::compress: the container buffer must be 8-byte aligned");
  uint8_t* const s = m.ensure_scratch();
  const size_t n = cfg.num_chunks;
  const Layout lay = layout_of(n);
  const uint8_t** in_ptrs = reinterpret_cast<const uint8_t**>(s);
  size_t* in_bytes = reinterpret_cast<size_t*>(s + (size_t)m.slab * 8);
  uint8_t** out_ptrs = reinterpret_cast<uint8_t**>(s + (size_t)m.slab * 16);
  uint32_t* words = reinterpret_cast<uint32_t*>(s + (size_t)m.slab * 44);
  uint8_t* slots = s + m.lists_bytes();
  uint16_t* tables = reinterpret_cast<uint16_t*>(
      (reinterpret_cast<uintptr_t>(slots + (size_t)m.slab * m.slot_bytes) + 15) & ~uintptr_t(15));
  header_kernel<<<1, 1, 0, m.stream>>>(comp_buffer, cfg.uncompressed_buffer_size, n, m.chunk_bytes, (uint32_t)lay.data,
                                       (uint32_t)m.data_type, cfg.get_status());
  for (size_t first = 0; first < n; first += m.slab) {
    const uint32_t count = (uint32_t)(n - first < m.slab ? n - first : m.slab);
    slab_inputs_kernel<<<(count + kBlock - 1) / kBlock, kBlock, 0, m.stream>>>(
        decomp_buffer, cfg.uncompressed_buffer_size, m.chunk_bytes, first, count, slots, m.slot_bytes, in_ptrs,
        in_bytes, out_ptrs);
    // sizes go straight into the container's size array
    check(lz4_launch_compress(in_ptrs, in_bytes, out_ptrs, reinterpret_cast<size_t*>(comp_buffer + lay.sizes) + first,
                              m.ht_size, count, m.elem, words, tables, m.slab, m.chunk_bytes, lz4_mode_from_environment(),
                              m.stream),
          "LZ4Manager::compress");
    slab_place_kernel<<<1, kBlock, 0, m.stream>>>(comp_buffer, lay.sizes, lay.offsets, first, count);
    slab_gather_kernel<<<(count + 3) / 4, kBlock, 0, m.stream>>>(comp_buffer, lay.sizes, lay.offsets, lay.data, first,
                                                               count, slots, m.slot_bytes);
  }
  check(hipGetLastError(), "LZ4Manager::compress kernels");
}

DecompressionConfig LZ4Manager::configure_decompression(const uint8_t* comp_buffer)
{
  Impl& m = *impl;
  DecompressionConfig d;
  check(hipMemcpyAsync(m.header_host, comp_buffer, sizeof(CommonHeader), hipMemcpyDeviceToHost, m.stream), "read header");
  check(hipStreamSynchronize(m.stream), "read header");
  d.decomp_data_size = (size_t)m.header_host->decomp_data_size;
  d.num_chunks = (uint32_t)m.header_host->num_chunks;
  return d;
}

DecompressionConfig LZ4Manager::configure_decompression(const CompressionConfig& comp_config)
{
  DecompressionConfig d;
  d.decomp_data_size = comp_config.uncompressed_buffer_size;
  d.num_chunks = (uint32_t)comp_config.num_chunks;
  return d;
}

void LZ4Manager::decompress(uint8_t* decomp_buffer, const uint8_t* comp_buffer, const DecompressionConfig& cfg)
{
  Impl& m = *impl;
  uint8_t* const s = m.ensure_scratch();
  const size_t n = cfg.num_chunks;
  const Layout lay = layout_of(n);
  const uint8_t** comp_ptrs = reinterpret_cast<const uint8_t**>(s);
  size_t* caps = reinterpret_cast<size_t*>(s + (size_t)m.slab * 8);
  uint8_t** out_ptrs = reinterpret_cast<uint8_t**>(s + (size_t)m.slab * 16);
  size_t* actual = reinterpret_cast<size_t*>(s + (size_t)m.slab * 24);
  hipcompStatus_t* statuses = reinterpret_cast<hipcompStatus_t*>(s + (size_t)m.slab * 40);
  set_status_kernel<<<1, 1, 0, m.stream>>>(cfg.get_status(), hipcompSuccess);
  for (size_t first = 0; first < n; first += m.slab) {
    const uint32_t count = (uint32_t)(n - first < m.slab ? n - first : m.slab);
    slab_streams_kernel<<<(count + kBlock - 1) / kBlock, kBlock, 0, m.stream>>>(
        comp_buffer, lay.offsets, decomp_buffer, cfg.decomp_data_size, m.chunk_bytes, first, count, comp_ptrs,
        out_ptrs, caps);
    lz4_launch_decompress(comp_ptrs, reinterpret_cast<const size_t*>(comp_buffer + lay.sizes) + first, caps, count,
                          out_ptrs, actual, statuses, true, m.stream);
    slab_verdict_kernel<<<(count + kBlock - 1) / kBlock, kBlock, 0, m.stream>>>(statuses, actual, caps, count,
                                                                                cfg.get_status());
  }
  check(hipGetLastError(), "LZ4Manager::decompress kernels");
}

void LZ4Manager::set_scratch_buffer(uint8_t* new_scratch_buffer)
{
  if (impl->own_scratch)
    (void)hipFree(impl->scratch);
  impl->own_scratch = false;
  impl->scratch = new_scratch_buffer;
}

size_t LZ4Manager::get_required_scratch_buffer_size() { return impl->scratch_bytes(); }

size_t LZ4Manager::get_compressed_output_size(uint8_t* comp_buffer)
{
  Impl& m = *impl;
  check(hipMemcpyAsync(m.header_host, comp_buffer, sizeof(CommonHeader), hipMemcpyDeviceToHost, m.stream), "read header");
  check(hipStreamSynchronize(m.stream), "read header");
  return (size_t)(m.header_host->comp_data_size + m.header_host->comp_data_offset);
}

} // namespace hipcomp

// ---- C binding -----------------------------------------------------------------------------
struct hipcompHlifManager
{
  hipcomp::LZ4Manager* lz4 = nullptr;
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
    h->lz4 = new hipcomp::LZ4Manager(uncomp_chunk_size, data_type, stream, dev);
    h->stream = stream;
    *manager = h.release();
  });
}

hipcompStatus_t hipcompHlifManagerDestroy(hipcompHlifManager_t* manager)
{
  if (manager) {
    delete manager->lz4;
    delete manager;
  }
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

} // extern "C"
