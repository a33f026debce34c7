// lz4_interop.cpp -- LZ4 frame <-> block list on the host (include/hipcomp/lz4_interop.h).
//
// Frame layout (LZ4 Frame Format Description v1.6.x, the public specification of
// liblz4's lz4frame): magic 0x184D2204 | FLG | BD | [content size, 8 bytes] |
// [dictionary id, 4] | HC | { block size (bit 31 = stored raw) | data | [block
// checksum, 4] }* | end mark 0 | [content checksum, 4].  HC is the second byte
// of the XXH32 (seed 0) of the descriptor bytes from FLG on.
#include "host_common.hpp"

#include "hipcomp/lz4_interop.h"

#include <cstring>

namespace {

constexpr uint32_t kMagic = 0x184D2204u;

uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// XXH32 (public algorithm, xxHash specification), only ever run on the few
// descriptor bytes of a frame header
uint32_t xxh32(const uint8_t* p, size_t len, uint32_t seed)
{
  const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
  const uint8_t* const end = p + len;
  uint32_t h;
  if (len >= 16) {
    uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    do {
      v1 = rotl(v1 + rd32(p) * P2, 13) * P1; p += 4;
      v2 = rotl(v2 + rd32(p) * P2, 13) * P1; p += 4;
      v3 = rotl(v3 + rd32(p) * P2, 13) * P1; p += 4;
      v4 = rotl(v4 + rd32(p) * P2, 13) * P1; p += 4;
    } while (p + 16 <= end);
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
  } else {
    h = seed + P5;
  }
  h += (uint32_t)len;
  while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
  while (p < end) { h = rotl(h + *p * P5, 11) * P1; ++p; }
  h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
  return h;
}

const size_t kBlockSizes[8] = {0, 0, 0, 0, 64u << 10, 256u << 10, 1u << 20, 4u << 20};

} // namespace

extern "C" {

size_t hipcompLZ4FrameBound(size_t num_chunks, size_t total_compressed_bytes)
{
  return 15 + 4 * num_chunks + total_compressed_bytes + 4;
}

hipcompStatus_t hipcompLZ4FrameFromBlocks(
    const void* const* host_compressed_ptrs, const size_t* host_compressed_bytes,
    const size_t* host_uncompressed_bytes, size_t num_chunks, void* frame, size_t frame_capacity,
    size_t* frame_bytes)
{
  static const char* fn = "hipcompLZ4FrameFromBlocks()";
  HCAMD_REQUIRE_NOT_NULL(fn, frame);
  HCAMD_REQUIRE_NOT_NULL(fn, frame_bytes);
  if (num_chunks) {
    HCAMD_REQUIRE_NOT_NULL(fn, host_compressed_ptrs);
    HCAMD_REQUIRE_NOT_NULL(fn, host_compressed_bytes);
    HCAMD_REQUIRE_NOT_NULL(fn, host_uncompressed_bytes);
  }
  size_t largest = 0, total = 0;
  uint64_t content = 0;
  for (size_t i = 0; i < num_chunks; ++i) {
    if (host_compressed_bytes[i] && !host_compressed_ptrs[i])
      return hcamd::fail(fn, "null chunk pointer");
    // (an LZ4 block of incompressible data is a little larger than the data:
    // the declared block size has to hold the stored block as well)
    largest = host_uncompressed_bytes[i] > largest ? host_uncompressed_bytes[i] : largest;
    largest = host_compressed_bytes[i] > largest ? host_compressed_bytes[i] : largest;
    total += host_compressed_bytes[i];
    content += host_uncompressed_bytes[i];
  }
  int bd = 4;
  while (bd < 7 && kBlockSizes[bd] < largest)
    ++bd;
  if (largest > kBlockSizes[7])
    return hcamd::fail(fn, "a chunk is larger than 4 MiB, the largest block an LZ4 frame can declare");
  if (frame_capacity < hipcompLZ4FrameBound(num_chunks, total))
    return hcamd::fail(fn, "frame buffer too small: see hipcompLZ4FrameBound()");
  uint8_t* p = static_cast<uint8_t*>(frame);
  wr32(p, kMagic);
  p[4] = (uint8_t)((1u << 6) | (1u << 5) | (1u << 3)); // version 01, independent blocks, content size
  p[5] = (uint8_t)(bd << 4);
  for (int k = 0; k < 8; ++k)
    p[6 + k] = (uint8_t)(content >> (8 * k));
  p[14] = (uint8_t)(xxh32(p + 4, 10, 0) >> 8);
  p += 15;
  for (size_t i = 0; i < num_chunks; ++i) {
    if (host_uncompressed_bytes[i] == 0)
      continue; // (an empty chunk has no block; a zero size would read as the end mark)
    if (host_compressed_bytes[i] >= 0x80000000ull)
      return hcamd::fail(fn, "block too large");
    wr32(p, (uint32_t)host_compressed_bytes[i]);
    std::memcpy(p + 4, host_compressed_ptrs[i], host_compressed_bytes[i]);
    p += 4 + host_compressed_bytes[i];
  }
  wr32(p, 0);
  p += 4;
  *frame_bytes = (size_t)(p - static_cast<uint8_t*>(frame));
  return hipcompSuccess;
}

hipcompStatus_t hipcompLZ4FrameToBlocks(
    const void* frame, size_t frame_bytes, size_t* num_blocks, size_t* block_max_bytes,
    uint64_t* content_bytes, size_t* block_offsets, size_t* block_bytes, int* block_is_raw,
    size_t blocks_capacity)
{
  static const char* fn = "hipcompLZ4FrameToBlocks()";
  HCAMD_REQUIRE_NOT_NULL(fn, frame);
  HCAMD_REQUIRE_NOT_NULL(fn, num_blocks);
  HCAMD_REQUIRE_NOT_NULL(fn, block_max_bytes);
  const uint8_t* const base = static_cast<const uint8_t*>(frame);
  *num_blocks = 0;
  if (frame_bytes < 7 + 4 || rd32(base) != kMagic)
    return hcamd::fail(fn, "not an LZ4 frame", hipcompErrorCannotDecompress);
  const uint8_t flg = base[4], bdb = base[5];
  if ((flg >> 6) != 1 || (bdb >> 4) < 4 || (bdb >> 4) > 7)
    return hcamd::fail(fn, "unsupported frame version or block size", hipcompErrorCannotDecompress);
  if (!(flg & (1u << 5)))
    return hcamd::fail(fn, "blocks of this frame depend on each other (linked mode): cannot be decoded as a batch",
                       hipcompErrorNotSupported);
  if (flg & 1u)
    return hcamd::fail(fn, "frame needs a dictionary", hipcompErrorNotSupported);
  const bool has_size = flg & (1u << 3), block_sums = flg & (1u << 4);
  size_t at = 6;
  uint64_t content = 0;
  if (has_size) {
    if (frame_bytes < at + 8)
      return hcamd::fail(fn, "truncated frame header", hipcompErrorCannotDecompress);
    for (int k = 0; k < 8; ++k)
      content |= (uint64_t)base[at + k] << (8 * k);
    at += 8;
  }
  if (frame_bytes < at + 1 || base[at] != (uint8_t)(xxh32(base + 4, at - 4, 0) >> 8))
    return hcamd::fail(fn, "bad frame header checksum", hipcompErrorCannotDecompress);
  ++at;
  if (content_bytes)
    *content_bytes = content;
  *block_max_bytes = kBlockSizes[bdb >> 4];
  size_t n = 0;
  for (;;) {
    if (frame_bytes < at + 4)
      return hcamd::fail(fn, "truncated frame (no end mark)", hipcompErrorCannotDecompress);
    const uint32_t word = rd32(base + at);
    at += 4;
    if (word == 0)
      break;
    const size_t sz = word & 0x7FFFFFFFu;
    if (sz > *block_max_bytes || frame_bytes < at + sz + (block_sums ? 4 : 0))
      return hcamd::fail(fn, "truncated or oversized block", hipcompErrorCannotDecompress);
    if (block_offsets || block_bytes || block_is_raw) {
      if (n >= blocks_capacity)
        return hcamd::fail(fn, "block arrays too small");
      if (block_offsets) block_offsets[n] = at;
      if (block_bytes) block_bytes[n] = sz;
      if (block_is_raw) block_is_raw[n] = (int)(word >> 31);
    }
    ++n;
    at += sz + (block_sums ? 4 : 0);
  }
  *num_blocks = n;
  return hipcompSuccess;
}

} // extern "C"
