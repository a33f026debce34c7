/*
 * snappy_oracle.c -- CPU restatement of the reference's batched Snappy codec.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Parity pinning: (1) the reference's own decoder golden vectors
 * (tests/test_snappy_app.cpp:210-223, extracted as data into
 * tests/golden/snappy_app_vectors.json) and encoder known answers
 * (src/test/SnappyLargeTokens_test.cpp:381-447, restated in
 * tests/test_snappy_oracle_cpu.py); (2) output of the reference build
 * oracle/_ref/libhipcomp_ref.so run on MI355X, committed as
 * tests/golden/snappy_reference.json.
 *
 * Encoder restated (line numbers: /root/reference/src/snappy/compression.hiph):
 *   snap_hash            :57-60    (v * 0x102A6B) >> 20, 12 bits
 *   HashMatchAny         :157-172  mask of ALL 64 lanes with the same hash
 *                                  (lanes past the end carry hash 0)
 *   FindFourByteMatch    :190-246  64-byte windows, <= 256 literal bytes
 *   Match60              :251-269
 *   StoreLiterals        :73-117, StoreCopy :129-151
 *   do_snap              :281-385  (wave 0 emits what wave 1 found one step
 *                                  earlier: logically sequential)
 * Hardware-defined spot: the hash-map update mask `(2ULL << literal_cnt) - 1`
 * with literal_cnt == 64 (:240) is a shift by the type width; gfx9's
 * v_lshlrev_b64 uses shift & 63, so the mask is 1 and only lane 0 updates the
 * map for a window without a match (SURVEY.md finding 3, confirmed in the ISA
 * and by the reference build on MI355X).
 *
 * Decoder: the Snappy format as the reference decodes it
 * (decompression.hiph:106-211, decompression_decode.hiph:72-150, :207-289),
 * with three tightenings where the reference has undefined behaviour:
 * reads of the compressed stream are bounded by its length, a copy may not
 * reach before the start of the output (the reference checks the offset
 * against the position AFTER the copy, decompression_decode.hiph:112), and an
 * empty stream reports 0 bytes.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define W 64
#define HASH_BITS 12
#define MAX_LITERAL_LENGTH 256
#define MAX_COPY_DISTANCE 32768

size_t oracle_snappy_max_compressed_size(size_t n)
{
  return 32 + n + n / 6; /* SnappyBatch.cpp:72-76 */
}

static uint32_t load32(const uint8_t* p)
{
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)
         | ((uint32_t)p[3] << 24);
}

static uint32_t snap_hash(uint32_t v)
{
  return (uint32_t)(v * ((1u << 20) + 0x2a00u + 0x6au + 1u)) >> (32 - HASH_BITS);
}

/* StoreLiterals :73-117 (all stores bounds-checked against `end`) */
static uint8_t* store_literals(
    uint8_t* dst, uint8_t* end, const uint8_t* src, uint32_t len_minus1)
{
  if (len_minus1 < 60) {
    if (dst < end) dst[0] = (uint8_t)(len_minus1 << 2);
    dst += 1;
  } else if (len_minus1 <= 0xff) {
    if (dst + 1 < end) { dst[0] = 60 << 2; dst[1] = (uint8_t)len_minus1; }
    dst += 2;
  } else if (len_minus1 <= 0xffff) {
    if (dst + 2 < end) { dst[0] = 61 << 2; dst[1] = (uint8_t)len_minus1; dst[2] = (uint8_t)(len_minus1 >> 8); }
    dst += 3;
  } else if (len_minus1 <= 0xffffff) {
    if (dst + 3 < end) {
      dst[0] = 62 << 2; dst[1] = (uint8_t)len_minus1; dst[2] = (uint8_t)(len_minus1 >> 8);
      dst[3] = (uint8_t)(len_minus1 >> 16);
    }
    dst += 4;
  } else {
    if (dst + 4 < end) {
      dst[0] = 63 << 2; dst[1] = (uint8_t)len_minus1; dst[2] = (uint8_t)(len_minus1 >> 8);
      dst[3] = (uint8_t)(len_minus1 >> 16); dst[4] = (uint8_t)(len_minus1 >> 24);
    }
    dst += 5;
  }
  for (uint32_t i = 0; i <= len_minus1; ++i)
    if (dst + i < end) dst[i] = src[i];
  return dst + len_minus1 + 1;
}

/* StoreCopy :129-151 */
static uint8_t* store_copy(uint8_t* dst, uint8_t* end, uint32_t copy_len, uint32_t distance)
{
  if (copy_len < 12 && distance < 2048) {
    if (dst + 2 <= end) {
      dst[0] = (uint8_t)(((distance & 0x700) >> 3) | ((copy_len - 4) << 2) | 0x01);
      dst[1] = (uint8_t)distance;
    }
    return dst + 2;
  }
  if (dst + 3 <= end) {
    dst[0] = (uint8_t)(((copy_len - 1) << 2) | 0x2);
    dst[1] = (uint8_t)distance;
    dst[2] = (uint8_t)(distance >> 8);
  }
  return dst + 3;
}

/*
 * FindFourByteMatch :190-246 at 64 lanes.  Returns the literal count; sets
 * *copy_length (0 or 4) and *copy_distance; updates hash_map.
 */
static uint32_t find_four_byte_match(
    uint16_t* hash_map, const uint8_t* src, uint32_t len, uint32_t pos0,
    uint32_t* copy_length, uint32_t* copy_distance)
{
  uint32_t pos = pos0;
  const uint32_t maxpos = pos0 + MAX_LITERAL_LENGTH - (W - 1);
  uint32_t literal_cnt;
  *copy_length = 0;
  do {
    int valid4[W];
    uint32_t data32[W], hash[W], offset[W];
    uint64_t local_match[W];
    int match[W];
    for (int t = 0; t < W; ++t) {
      valid4[t] = (pos + (uint32_t)t + 4 <= len);
      data32[t] = valid4[t] ? load32(src + pos + t) : 0;
      hash[t] = valid4[t] ? snap_hash(data32[t]) : 0;
    }
    /* HashMatchAny: every lane takes part, valid or not */
    for (int t = 0; t < W; ++t) {
      uint64_t m = 0;
      for (int u = 0; u < W; ++u)
        if (hash[u] == hash[t]) m |= 1ull << u;
      local_match[t] = m;
    }
    for (int t = 0; t < W; ++t) {
      if (valid4[t]) {
        const uint64_t below = local_match[t] & ((1ull << t) - 1);
        const uint32_t lml = below ? (uint32_t)(63 - __builtin_clzll(below)) : 0xFFFFFFFFu;
        const uint32_t lmd = data32[lml < (uint32_t)t ? lml : (uint32_t)t];
        if (lml < (uint32_t)t && lmd == data32[t]) {
          match[t] = 1;
          offset[t] = pos + lml;
        } else {
          uint32_t off = (pos & ~0xffffu) | hash_map[hash[t]];
          if (off >= pos) off = (off >= 0x10000u) ? off - 0x10000u : pos;
          offset[t] = off;
          match[t] = (off < pos && off + MAX_COPY_DISTANCE >= pos + (uint32_t)t
                      && load32(src + off) == data32[t]);
        }
      } else {
        match[t] = 0;
        local_match[t] = 0;
        offset[t] = pos + (uint32_t)t;
      }
    }
    literal_cnt = W;
    for (int t = 0; t < W; ++t)
      if (match[t]) { literal_cnt = (uint32_t)t; break; }
    if (literal_cnt < W) {
      *copy_distance = pos + literal_cnt - offset[literal_cnt];
      *copy_length = 4;
    }
    /* hash-map update :240-242; (2ULL << 64) behaves as (2ULL << 0) on gfx9 */
    const uint64_t upd = (2ull << (literal_cnt & 63)) - 1;
    for (int t = 0; t < W; ++t) {
      const uint64_t m = local_match[t] & upd;
      if ((uint32_t)t <= literal_cnt && m && t == 63 - __builtin_clzll(m))
        hash_map[hash[t]] = (uint16_t)(pos + (uint32_t)t);
    }
    pos += literal_cnt;
  } while (literal_cnt == W && pos < maxpos);
  return (pos < len ? pos : len) - pos0;
}

/* do_snap :281-385.  dst must hold oracle_snappy_max_compressed_size(len). */
int oracle_snappy_compress(const uint8_t* src, size_t len64, uint8_t* dst, size_t* out_len)
{
  static __thread uint16_t hash_map[1 << HASH_BITS]; /* per thread, see cascaded_oracle.c */
  const uint32_t len = (uint32_t)len64;
  uint8_t* const base = dst;
  uint8_t* const end = dst + oracle_snappy_max_compressed_size(len);
  uint32_t v = len;
  while (v > 0x7f) {
    if (dst < end) dst[0] = (uint8_t)(v | 0x80);
    dst++;
    v >>= 7;
  }
  if (dst < end) dst[0] = (uint8_t)v;
  dst++;
  memset(hash_map, 0, sizeof hash_map);
  uint32_t pos = 0;
  while (pos < len) {
    uint32_t copy_len, distance = 0;
    const uint32_t lit = find_four_byte_match(hash_map, src, len, pos, &copy_len, &distance);
    if (copy_len) {
      const uint32_t match_pos = pos + lit + copy_len; /* copy_len == 4 */
      uint32_t n = len - match_pos;
      if (n > 64 - copy_len) n = 64 - copy_len;
      uint32_t k = 0;
      while (k < n && src[match_pos + k] == src[match_pos - distance + k]) ++k;
      copy_len += k;
    }
    if (lit > 0) {
      dst = store_literals(dst, end, src + pos, lit - 1);
      pos += lit;
    }
    if (copy_len > 0) {
      dst = store_copy(dst, end, copy_len, distance);
      pos += copy_len;
    }
  }
  *out_len = (size_t)(dst - base);
  return 0;
}

/* get_uncompressed_sizes_kernel, SnappyBatchKernels.hip:84-134 */
size_t oracle_snappy_uncompressed_size(const uint8_t* comp, size_t comp_len)
{
  const uint8_t* cur = comp;
  const uint8_t* end = comp + comp_len;
  uint32_t n = 0;
  if (cur < end) {
    n = *cur++;
    if (n > 0x7f) {
      uint32_t c = (cur < end) ? *cur++ : 0;
      n = (n & 0x7f) | (c << 7);
      if (n >= (0x80u << 7)) {
        c = (cur < end) ? *cur++ : 0;
        n = (n & 0x3fff) | (c << 14);
        if (n >= (0x80u << 14)) {
          c = (cur < end) ? *cur++ : 0;
          n = (n & 0x1fffff) | (c << 21);
          if (n >= (0x80u << 21)) {
            c = (cur < end) ? *cur++ : 0;
            n = (c < 0x8) ? ((n & 0xfffffff) | (c << 28)) : 0;
          }
        }
      }
    }
  }
  return n;
}

/*
 * do_unsnap.  cap == 0 means "capacity = the stream's own size" as in the
 * reference (decompression.hiph:148-149).  Returns the status the reference
 * reports (0 / 12); *actual = uncompressed_size - bytes_left (:197-198).
 */
int oracle_snappy_decompress(
    const uint8_t* comp, size_t comp_len, uint8_t* out, size_t cap, size_t* actual)
{
  *actual = 0;
  if (comp_len == 0)
    return 12;
  const uint32_t end = (uint32_t)comp_len;
  uint32_t cur = 0;
  /* decode_uncompressed_size, decompression.hiph:70-104 */
  int err = 0;
  uint32_t usize = comp[cur++];
  if (usize > 0x7f) {
    uint32_t c = (cur < end) ? comp[cur++] : 0;
    usize = (usize & 0x7f) | (c << 7);
    if (usize >= (0x80u << 7)) {
      c = (cur < end) ? comp[cur++] : 0;
      usize = (usize & 0x3fff) | (c << 14);
      if (usize >= (0x80u << 14)) {
        c = (cur < end) ? comp[cur++] : 0;
        usize = (usize & 0x1fffff) | (c << 21);
        if (usize >= (0x80u << 21)) {
          c = (cur < end) ? comp[cur++] : 0;
          if (c < 0x8)
            usize = (usize & 0xfffffff) | (c << 28);
          else
            err = 1;
        }
      }
    }
  }
  const size_t dst_size = cap ? cap : usize;
  if ((cur >= end && usize != 0) || usize > dst_size)
    err = 1;
  if (err)
    return 12;
  uint32_t bytes_left = usize, dst_pos = 0;
  while (bytes_left > 0) {
    if (cur >= end) break;
    const uint32_t b0 = comp[cur];
    uint32_t blen, offset;
    if (b0 & 3) {
      if (!(b0 & 2)) { /* xxxxxx01.oooooooo */
        if (end - cur < 2) break;
        offset = ((b0 & 0xe0) << 3) | comp[cur + 1];
        blen = ((b0 >> 2) & 7) + 4;
        cur += 2;
      } else if (b0 & 1) { /* 4-byte offset */
        if (end - cur < 5) break;
        offset = load32(comp + cur + 1);
        blen = (b0 >> 2) + 1;
        cur += 5;
      } else { /* 2-byte offset */
        if (end - cur < 3) break;
        offset = (uint32_t)comp[cur + 1] | ((uint32_t)comp[cur + 2] << 8);
        blen = (b0 >> 2) + 1;
        cur += 3;
      }
      if (offset == 0 || offset > dst_pos || bytes_left < blen) break;
      for (uint32_t i = 0; i < blen; ++i)
        out[dst_pos + i] = out[dst_pos - offset + i];
    } else {
      blen = b0 >> 2;
      cur += 1;
      if (blen >= 60) {
        const uint32_t nb = blen - 59;
        if (end - cur < nb) break;
        blen = 0;
        for (uint32_t i = 0; i < nb; ++i)
          blen |= (uint32_t)comp[cur + i] << (8 * i);
        cur += nb;
      }
      blen += 1;
      if (blen == 0 || bytes_left < blen || end - cur < blen) break;
      memcpy(out + dst_pos, comp + cur, blen);
      cur += blen;
    }
    dst_pos += blen;
    bytes_left -= blen;
  }
  *actual = usize - bytes_left;
  return bytes_left ? 12 : 0;
}
