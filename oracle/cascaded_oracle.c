/*
 * cascaded_oracle.c -- CPU restatement of the reference's batched Cascaded
 * codec (fused RLE -> Delta -> BitPack over 4096-byte sub-chunks).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Parity pinning: (1) the wire-layout known answers of the reference's own
 * test (tests/test_cascaded_batch.cpp:91-150, 337-378, restated as data in
 * tests/test_cascaded_oracle_cpu.py); (2) output of the reference build
 * oracle/_ref/libhipcomp_ref.so run on MI355X, committed as
 * tests/golden/cascaded_reference.json (compared under the don't-care mask).
 *
 * Restated (line numbers: /root/reference/src/CascadedKernels.hiph):
 *   get_chunk_metadata_size        :101-106
 *   block_rle_compress             :124-241
 *   block_delta_compress           :317-328
 *   get_for_bitwidth/block_bitpack :394-553
 *   block_write                    :646-680
 *   do_cascaded_compression_kernel :761-1058
 *   cascaded_decompression_fcn     :1106-1435 (+ block_read :702-737,
 *     block_bitunpack :563-618, block_delta_decompress :343-377,
 *     block_rle_decompress :255-305)
 *   get_decompress_size_kernel     src/lowlevel/CascadedBatch.hip:262-281
 *
 * Don't-care bytes.  The reference copies whole 32-bit words out of shared
 * memory, so some output bytes are whatever the LDS held (SURVEY.md App.
 * C.4), and some alignment gaps are never written at all.  The oracle writes
 * 0 there and clears the byte in `mask` (0xFF = meaningful).  Such bytes:
 * the tail of a raw (bp=0) array whose length is not a multiple of 4; the
 * gap between the frame of reference and the bit-width word of a bit-packed
 * array narrower than 4 bytes and the gap after that word for 8-byte types;
 * chunk-metadata and delta-header padding; alignment gaps before/after the
 * final array for 8-byte types; the FOR/bit-width of an EMPTY bit-packed
 * array (uninitialised in the reference, App. C.5; 0/0 here).
 *
 * Deliberate deviations: a delta layer that would run on zero
 * elements makes the reference loop on `input_size - 1` of an unsigned 0
 * (:323) -- undefined; here the partition falls back to the raw layout.
 * The decoder validates array lengths/run totals against the 4096-byte
 * sub-chunk instead of writing past its buffers.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define CHUNK_BYTES 4096 /* the reference's (hard-wired) sub-chunk, CascadedKernels.hiph:88; opts.chunk_size is ignored */
#define PART_META 8

static size_t ru(size_t a, size_t b) { return (a + b - 1) / b * b; }

static uint64_t ld(const uint8_t* p, int s)
{
  uint64_t v = 0;
  memcpy(&v, p, (size_t)s);
  return v;
}
static void st(uint8_t* p, uint64_t v, int s) { memcpy(p, &v, (size_t)s); }
static int64_t sx(uint64_t v, int s) /* sign extend s bytes */
{
  const int sh = 64 - 8 * s;
  return (int64_t)(v << sh) >> sh;
}
static uint64_t trunc_s(uint64_t v, int s)
{
  return s == 8 ? v : (v & ((1ull << (8 * s)) - 1));
}

size_t oracle_cascaded_max_compressed_size(size_t n) { return ru(n, 4) + 8; }

/* :101-106 */
static int chunk_metadata_size(int s, int R, int D)
{
  return (int)(ru((size_t)(4 + 4 * (R + 1)), (size_t)s) + ru((size_t)(s * D), 4));
}

typedef struct
{
  uint8_t* out;
  uint8_t* mask;
} sink_t;

static void put(sink_t* k, size_t off, const void* src, size_t n, int meaningful)
{
  if (meaningful)
    memcpy(k->out + off, src, n);
  else
    memset(k->out + off, 0, n);
  memset(k->mask + off, meaningful ? 0xFF : 0x00, n);
}

/*
 * block_write :646-680 of `n` elements of `es` bytes.  Returns the byte
 * length the reference records (out_bytes); *padded = bytes actually copied.
 * With sink == NULL only the sizes are computed.
 */
static size_t write_array(
    sink_t* k, size_t off, const uint64_t* v, size_t n, int es, int bp,
    size_t* padded)
{
  if (!bp) {
    const size_t ob = n * (size_t)es;
    *padded = ru(ob, 4);
    if (k) {
      for (size_t i = 0; i < n; ++i) {
        uint64_t x = v[i];
        put(k, off + i * (size_t)es, &x, (size_t)es, 1);
      }
      uint64_t z = 0;
      if (*padded > ob)
        put(k, off + ob, &z, *padded - ob, 0);
    }
    return ob;
  }
  /* get_for_bitwidth :394-471: min/max under the SIGNED interpretation */
  int64_t mn = 0, mx = 0;
  for (size_t i = 0; i < n; ++i) {
    const int64_t x = sx(v[i], es);
    if (i == 0 || x < mn) mn = x;
    if (i == 0 || x > mx) mx = x;
  }
  uint32_t bw;
  if (es > 4) {
    const uint64_t range = (uint64_t)mx - (uint64_t)mn;
    bw = range ? (uint32_t)(64 - __builtin_clzll(range)) : 0;
  } else {
    const uint32_t range = (uint32_t)mx - (uint32_t)mn;
    bw = range ? (uint32_t)(32 - __builtin_clz(range)) : 0;
  }
  const size_t words = (n * bw + 31) / 32;
  const size_t hdr = ru((size_t)es + 4, es > 4 ? (size_t)es : 4);
  const size_t ob = hdr + 4 * words;
  *padded = ob;
  if (!k)
    return ob;
  uint64_t z = 0;
  const uint64_t fr = trunc_s((uint64_t)mn, es);
  put(k, off, &fr, (size_t)es, n > 0);                 /* frame of reference */
  const size_t w_off = ru((size_t)es, 4);
  if (w_off > (size_t)es)
    put(k, off + (size_t)es, &z, w_off - (size_t)es, 0);
  const uint32_t word = (bw << 16) | (uint32_t)n;
  if (n > 0) {
    put(k, off + w_off, &word, 4, 1);
  } else { /* count = 0 is meaningful, the bit width is not */
    put(k, off + w_off, &word, 2, 1);
    put(k, off + w_off + 2, &z, 2, 0);
  }
  if (hdr > w_off + 4)
    put(k, off + w_off + 4, &z, hdr - w_off - 4, 0);
  /* block_bitpack :523-552: LSB-first, (x - FOR) in bw bits */
  for (size_t w = 0; w < words; ++w) {
    uint32_t acc = 0;
    const size_t b0 = w * 32;
    for (size_t i = b0 / bw; i * bw < b0 + 32; ++i) {
      uint64_t x = 0;
      if (i < n)
        x = trunc_s(v[i] - fr, es);
      const long sh = (long)(i * bw) - (long)b0;
      if (sh > 0)
        acc |= (uint32_t)(es > 4 ? x << sh : (uint64_t)((uint32_t)x << sh));
      else
        acc |= (uint32_t)(es > 4 ? x >> (-sh) : (uint64_t)((uint32_t)x >> (-sh)));
    }
    put(k, off + hdr + 4 * w, &acc, 4, 1);
  }
  return ob;
}

/*
 * One partition.  `type` is the hipcompType_t tag (byte 3 of the header),
 * s its size in bytes.  out/mask must hold oracle_cascaded_max_compressed_size
 * (in_bytes) bytes (and are fully initialised: untouched bytes get mask 0).
 */
int oracle_cascaded_compress(
    const uint8_t* in, size_t in_bytes, int type, int s, int R, int D, int bp,
    uint8_t* out, uint8_t* mask, size_t* out_bytes)
{
  static __thread uint64_t a[CHUNK_BYTES], b[CHUNK_BYTES], cnt[CHUNK_BYTES]; /* per thread: timed from a thread pool */
  const size_t cb = CHUNK_BYTES;
  const size_t cap = oracle_cascaded_max_compressed_size(in_bytes);
  memset(out, 0, cap);
  memset(mask, 0, cap);
  *out_bytes = 0;
  if (in_bytes == 0)
    return 0;                                            /* :856-860 */
  if (R < 0 || D < 0 || chunk_metadata_size(s, R, D) > 64)
    return -1;
  sink_t k = {out, mask};
  const size_t N = in_bytes / (size_t)s;
  const size_t limit = 4 * (2 + (in_bytes + 3) / 4);     /* :852-854, bytes */
  int use = !(R == 0 && D == 0 && bp == 0);              /* :868-870 */
  size_t cur = ru(PART_META, (size_t)s);                 /* :873-876 */
  const size_t ce = cb / (size_t)s;
  const size_t nchunks = (N + ce - 1) / ce;
  const int msz = chunk_metadata_size(s, R, D);
  const size_t dh_off = ru((size_t)(4 + 4 * (R + 1)), (size_t)s);

  for (size_t c = 0; c < nchunks && use; ++c) {
    const size_t chunk_start = cur;
    uint32_t meta[16] = {0};
    uint64_t dhead[16] = {0};
    cur += (size_t)msz;
    size_t n = N - c * ce < ce ? N - c * ce : ce;
    uint64_t* x = a;
    uint64_t* y = b;
    for (size_t i = 0; i < n; ++i)
      x[i] = ld(in + (c * ce + i) * (size_t)s, s);
    int rr = R, dr = D;
    const int layers = R > D ? R : D;
    for (int l = 0; l < layers && use; ++l) {
      if (rr > 0) {                                      /* :913-953 */
        size_t m = 0;
        for (size_t i = 0; i < n;) {
          size_t j = i + 1;
          while (j < n && x[j] == x[i]) ++j;
          y[m] = x[i];
          cnt[m] = j - i;
          ++m;
          i = j;
        }
        size_t padded;
        const size_t ob = write_array(NULL, 0, cnt, m, 2, bp, &padded);
        if (cur + ru(ob, 4) > limit) { use = 0; break; } /* :668-671 */
        write_array(&k, cur, cnt, m, 2, bp, &padded);
        cur += ru(ob, 4);
        meta[R - rr + 1] = (uint32_t)ob;
        uint64_t* t = x; x = y; y = t;
        n = m;
        --rr;
      }
      if (dr > 0) {                                      /* :955-977 */
        if (n == 0) { use = 0; break; }                  /* deviation, see top */
        dhead[D - dr] = x[0];
        for (size_t i = 0; i + 1 < n; ++i)
          y[i] = trunc_s(x[i + 1] - x[i], s);
        uint64_t* t = x; x = y; y = t;
        n -= 1;
        --dr;
      }
    }
    if (!use) break;
    const size_t fin = ru(cur, (size_t)s);               /* :983-984 */
    size_t padded;
    const size_t ob = write_array(NULL, 0, x, n, s, bp, &padded);
    if (fin + ru(ob, 4) > limit) { use = 0; break; }
    write_array(&k, fin, x, n, s, bp, &padded);
    cur = ru(fin + ru(ob, 4), (size_t)s);                /* :999-1001 */
    meta[0] = (uint32_t)(cur - chunk_start);
    meta[R + 1] = (uint32_t)ob;
    /* chunk metadata :1004-1014: sizes, pad, delta heads, pad */
    uint64_t z = 0;
    put(&k, chunk_start, meta, (size_t)(4 * (R + 2)), 1);
    if (dh_off > (size_t)(4 * (R + 2)))
      put(&k, chunk_start + (size_t)(4 * (R + 2)), &z, dh_off - (size_t)(4 * (R + 2)), 0);
    for (int i = 0; i < D; ++i)
      put(&k, chunk_start + dh_off + (size_t)(i * s), &dhead[i], (size_t)s, 1);
    if ((size_t)msz > dh_off + (size_t)(s * D))
      put(&k, chunk_start + dh_off + (size_t)(s * D), &z, (size_t)msz - dh_off - (size_t)(s * D), 0);
  }

  uint8_t hdr[8];
  if (use) {
    hdr[0] = (uint8_t)R; hdr[1] = (uint8_t)D; hdr[2] = (uint8_t)(bp ? 1 : 0);
    *out_bytes = cur;
  } else {                                               /* :1019-1053 */
    memset(out, 0, cap);
    memset(mask, 0, cap);
    const size_t raw = ru(PART_META, (size_t)s);
    put(&k, raw, in, N * (size_t)s, 1);
    hdr[0] = hdr[1] = hdr[2] = 0;
    *out_bytes = raw + ru(N * (size_t)s, 4);
  }
  hdr[3] = (uint8_t)type;
  const uint32_t ub = (uint32_t)(N * (size_t)s);
  memcpy(hdr + 4, &ub, 4);
  put(&k, 0, hdr, 8, 1);
  return 0;
}

/* get_decompress_size_kernel, CascadedBatch.hip:262-281 */
size_t oracle_cascaded_decompressed_size(const uint8_t* comp, size_t comp_bytes)
{
  if (comp_bytes < PART_META)
    return 0;
  uint32_t v;
  memcpy(&v, comp + 4, 4);
  return v;
}

/* block_read + block_bitunpack; returns element count or -1 */
static long read_array(
    const uint8_t* comp, size_t comp_words_end, size_t off, size_t nbytes,
    int es, int bp, uint64_t* dst, size_t max_elems)
{
  if (off % 4 || (off + ru(nbytes, 4)) / 4 > comp_words_end)   /* :712-713 */
    return -1;
  if (!bp) {
    const size_t n = nbytes / (size_t)es;
    if (n > max_elems) return -1;
    for (size_t i = 0; i < n; ++i)
      dst[i] = ld(comp + off + i * (size_t)es, es);
    return (long)n;
  }
  const size_t w_off = ru((size_t)es, 4);
  const size_t hdr = ru((size_t)es + 4, es > 4 ? (size_t)es : 4);
  if (nbytes < hdr) return -1;
  const uint64_t fr = ld(comp + off, es);
  uint32_t word;
  memcpy(&word, comp + off + w_off, 4);
  const uint32_t bw = word >> 16;
  const size_t n = word & 0xFFFF;
  if (n == 0) return 0;           /* FOR / bit width of an empty array are don't-care */
  if (n > max_elems || bw > (uint32_t)(8 * es)) return -1;
  if (hdr + 4 * ((n * bw + 31) / 32) > ru(nbytes, 4)) return -1;
  const uint8_t* data = comp + off + hdr;
  for (size_t i = 0; i < n; ++i) {
    uint64_t x = 0;
    for (uint32_t bit = 0; bit < bw; ++bit) {
      const size_t p = i * bw + bit;
      if ((data[p / 8] >> (p % 8)) & 1)
        x |= 1ull << bit;
    }
    dst[i] = trunc_s(x + fr, es);
  }
  return (long)n;
}

/*
 * cascaded_decompression_fcn :1106-1435.  The element size comes from the
 * partition's own header byte 3.  Returns the status (0 / 12); *actual as the
 * reference writes it (0 on failure).
 */
int oracle_cascaded_decompress(
    const uint8_t* comp, size_t comp_bytes, uint8_t* out, size_t cap, size_t* actual)
{
  static __thread uint64_t a[CHUNK_BYTES], b[CHUNK_BYTES], cnt[CHUNK_BYTES]; /* per thread: timed from a thread pool */
  static const int sizes[8] = {1, 1, 2, 2, 4, 4, 8, 8};
  *actual = 0;
  if (comp_bytes < PART_META)
    return 12;
  const int R = comp[0], D = comp[1], bp = comp[2] & 0x0F, type = comp[3];
  if (comp[2] >> 4) return 12;   /* byte 2 is use_bp, 0 or 1 (rounds 2-3 marked larger sub-chunks, an extension that is gone, here) */
  const size_t cb = CHUNK_BYTES;
  if (type > 7) return 12;
  const int s = sizes[type];
  uint32_t ub;
  memcpy(&ub, comp + 4, 4);
  const size_t N = ub / (size_t)s;
  if (cap < N * (size_t)s)
    return 12;                                           /* :1214-1223 */
  if (R == 0 && D == 0 && bp == 0) {                     /* :1225-1254 */
    if (comp_bytes < ru(PART_META, (size_t)s) + N * (size_t)s)
      return 12;
    memcpy(out, comp + ru(PART_META, (size_t)s), N * (size_t)s);
    *actual = N * (size_t)s;
    return 0;
  }
  if (R > 7 || chunk_metadata_size(s, R, D) > 64)
    return 12;
  const size_t end_w = comp_bytes / 4;                   /* partition_end_ptr */
  const int msz = chunk_metadata_size(s, R, D);
  const size_t dh_off = ru((size_t)(4 + 4 * (R + 1)), (size_t)s);
  const size_t ce = cb / (size_t)s;
  size_t pos = ru(PART_META, (size_t)s), done = 0;
  int ok = 1;
  while (pos / 4 < end_w) {                              /* :1268 */
    if ((pos + (size_t)msz) / 4 > end_w) { ok = 0; break; }
    uint32_t meta[16];
    memcpy(meta, comp + pos, (size_t)(4 * (R + 2)));
    /* A sub-chunk must lie inside the partition and move the cursor on: with
     * meta[0] in 1..3 the reference's cursor (:1412-1413) stands still and a
     * sub-chunk that decodes to zero elements would be read forever. */
    if (meta[0] < 4 || meta[0] > comp_bytes - pos) { ok = 0; break; }
    size_t offs[16];
    offs[0] = 0;
    if (R > 0) {
      for (int i = 0; i < R - 1; ++i)
        offs[i + 1] = ru(offs[i] + meta[i + 1], 4);
      offs[R] = ru(offs[R - 1] + meta[R], s > 4 ? (size_t)s : 4);
    }
    const size_t base = pos + (size_t)msz;
    uint64_t* x = a;
    uint64_t* y = b;
    long n = read_array(comp, end_w, base + offs[R], meta[1 + R], s, bp, x, ce);
    if (n < 0) { ok = 0; break; }
    /* Undo the layers in the exact reverse of the encoder (layer l = RLE_l
     * then Delta_l).  For num_RLEs >= num_deltas this is the order the
     * reference decoder uses (:1332-1391); for num_deltas > num_RLEs >= 1 the
     * reference's `remaining` comparisons undo RLE_0 before Delta_0 and cannot
     * decode its own encoder's output -- the true inverse is used here
     * (deliberate deviations). */
    const int layers = R > D ? R : D;
    for (int l = layers - 1; l >= 0 && ok; --l) {
      if (l < D) {                                       /* :1334-1352 */
        if ((size_t)n + 1 > ce) { ok = 0; break; }
        uint64_t acc = ld(comp + pos + dh_off + (size_t)(l * s), s);
        for (long i = 0; i < n; ++i) {
          y[i] = acc;
          acc = trunc_s(acc + x[i], s);
        }
        y[n] = acc;
        uint64_t* t = x; x = y; y = t;
        ++n;
      }
      if (l < R) {                                       /* :1354-1390 */
        long m = read_array(comp, end_w, base + offs[l], meta[l + 1], 2, bp, cnt, ce);
        if (m < 0 || m != n) { ok = 0; break; }
        size_t tot = 0;
        for (long i = 0; i < n; ++i) {
          if (tot + cnt[i] > ce) { ok = 0; break; }
          for (uint64_t j = 0; j < cnt[i]; ++j)
            y[tot + j] = x[i];
          tot += cnt[i];
        }
        if (!ok) break;
        uint64_t* t = x; x = y; y = t;
        n = (long)tot;
      }
    }
    if (!ok) break;
    if (done + (size_t)n > N) { ok = 0; break; }         /* :1395-1402 */
    for (long i = 0; i < n; ++i)
      st(out + (done + (size_t)i) * (size_t)s, x[i], s);
    done += (size_t)n;
    pos = ru(pos + (meta[0] / 4) * 4, (size_t)s);        /* :1412-1413 */
  }
  if (done != N) ok = 0;                                 /* :1417-1422 */
  *actual = ok ? N * (size_t)s : 0;
  return ok ? 0 : 12;
}
