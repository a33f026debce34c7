/*
 * cpu_baseline.c -- host-core LZ4 round trip for bench.py's `cpu_baseline`.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * BASELINE.json configs[0]: LZ4 round trip over 64 KiB chunks with the system
 * liblz4 (dlopen("liblz4.so.1"): the image ships the runtime library without
 * headers, so the three prototypes are declared here), pthreads with a static
 * contiguous partition of the chunk list, best of `reps`.  When liblz4 is not
 * present the caller falls back to the C restatement (oracle_lz4_compress).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*compress_fn)(const char*, char*, int, int);
typedef int (*decompress_fn)(const char*, char*, int, int);
typedef int (*bound_fn)(int);

typedef struct
{
  const uint8_t* in;
  uint8_t* comp;
  uint8_t* out;
  int* csize;
  size_t lo, hi, chunk, bound;
  int phase;
  compress_fn c;
  decompress_fn d;
} job_t;

static void* worker(void* p)
{
  job_t* j = (job_t*)p;
  for (size_t i = j->lo; i < j->hi; ++i) {
    if (j->phase == 0)
      j->csize[i] = j->c((const char*)j->in + i * j->chunk, (char*)j->comp + i * j->bound,
                         (int)j->chunk, (int)j->bound);
    else
      j->d((const char*)j->comp + i * j->bound, (char*)j->out + i * j->chunk, j->csize[i],
           (int)j->chunk);
  }
  return NULL;
}

static double now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* returns 0 ok, 1 liblz4 unavailable, 2 round trip mismatch */
int cpu_liblz4_roundtrip(
    const uint8_t* data, size_t n_chunks, size_t chunk, int threads, int reps,
    double* t_comp, double* t_decomp, size_t* comp_total)
{
  void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h)
    return 1;
  compress_fn c = (compress_fn)dlsym(h, "LZ4_compress_default");
  decompress_fn d = (decompress_fn)dlsym(h, "LZ4_decompress_safe");
  bound_fn b = (bound_fn)dlsym(h, "LZ4_compressBound");
  if (!c || !d || !b)
    return 1;
  const size_t bound = (size_t)b((int)chunk);
  uint8_t* comp = (uint8_t*)malloc(n_chunks * bound);
  uint8_t* out = (uint8_t*)malloc(n_chunks * chunk);
  int* csize = (int*)calloc(n_chunks, sizeof(int));
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * (size_t)threads);
  *t_comp = 1e30;
  *t_decomp = 1e30;
  for (int r = 0; r < reps; ++r) {
    for (int phase = 0; phase < 2; ++phase) {
      const size_t per = (n_chunks + (size_t)threads - 1) / (size_t)threads;
      const double t0 = now();
      for (int k = 0; k < threads; ++k) {
        size_t lo = (size_t)k * per, hi = lo + per;
        if (lo > n_chunks) lo = n_chunks;
        if (hi > n_chunks) hi = n_chunks;
        job_t j = {data, comp, out, csize, lo, hi, chunk, bound, phase, c, d};
        jobs[k] = j;
        pthread_create(&th[k], NULL, worker, &jobs[k]);
      }
      for (int k = 0; k < threads; ++k)
        pthread_join(th[k], NULL);
      const double dt = now() - t0;
      if (phase == 0 && dt < *t_comp) *t_comp = dt;
      if (phase == 1 && dt < *t_decomp) *t_decomp = dt;
    }
  }
  size_t tot = 0;
  for (size_t i = 0; i < n_chunks; ++i)
    tot += (size_t)csize[i];
  *comp_total = tot;
  const int bad = memcmp(out, data, n_chunks * chunk) != 0;
  free(comp); free(out); free(csize); free(th); free(jobs);
  return bad ? 2 : 0;
}

/* ------------------------------------------------------------------------
 * Snappy (BASELINE.json configs[3]) and Cascaded (configs[2]) on the host
 * cores, same harness: pthreads, static contiguous partition, best of `reps`.
 *   codec 1: Snappy -- system libsnappy's C API (dlopen("libsnappy.so.1"):
 *            snappy_compress / snappy_uncompress) when present, else a plain
 *            scalar encoder of libsnappy's kind (scalar_snappy_compress) and
 *            the C restatement's decoder (oracle_snappy_decompress);
 *   codec 2: Cascaded {type UINT, RLE 2, Delta 1, bit-packing} -- no third-party
 *            CPU equivalent exists, the C restatement is timed.
 * *used_lib: 1 = system library, 0 = restatement.
 * returns 0 ok, 2 round trip mismatch, 3 bad codec.
 * ------------------------------------------------------------------------ */
int oracle_snappy_compress(const uint8_t*, size_t, uint8_t*, size_t*);
int oracle_snappy_decompress(const uint8_t*, size_t, uint8_t*, size_t, size_t*);
size_t oracle_snappy_max_compressed_size(size_t);
int oracle_cascaded_compress(const uint8_t*, size_t, int, int, int, int, int, uint8_t*, uint8_t*, size_t*);
int oracle_cascaded_decompress(const uint8_t*, size_t, uint8_t*, size_t, size_t*);
size_t oracle_cascaded_max_compressed_size(size_t);

typedef int (*snappy_c_fn)(const char*, size_t, char*, size_t*);

/* A plain scalar Snappy encoder in the manner of libsnappy's (greedy, one
 * 16 Ki-entry table of 4-byte hashes, copies of up to 64 bytes, 2-byte
 * offsets): what a host would run when libsnappy itself is not installed.
 * It is NOT the reference's GPU encoder (that is oracle_snappy_compress, a
 * 64-lane emulation and accordingly slow); its streams are checked by
 * decoding them with oracle_snappy_decompress.  Chunks < 64 KiB + 1. */
static uint8_t* scalar_snappy_literal(uint8_t* op, const uint8_t* lit, size_t n)
{
  if (n == 0) return op;
  const size_t m = n - 1;
  if (m < 60) *op++ = (uint8_t)(m << 2);
  else if (m < 256) { *op++ = 60 << 2; *op++ = (uint8_t)m; }
  else { *op++ = 61 << 2; *op++ = (uint8_t)m; *op++ = (uint8_t)(m >> 8); }
  memcpy(op, lit, n);
  return op + n;
}

static int scalar_snappy_compress(const char* src_, size_t n, char* dst_, size_t* out_len)
{
  const uint8_t* src = (const uint8_t*)src_;
  uint8_t* op = (uint8_t*)dst_;
  static __thread uint16_t table[1 << 14];
  memset(table, 0, sizeof(table));
  { size_t v = n; while (v >= 128) { *op++ = (uint8_t)(v | 128); v >>= 7; } *op++ = (uint8_t)v; }
  size_t ip = 0, lit = 0;
  if (n >= 15) {
    const size_t limit = n - 4;
    ip = 1;
    while (ip <= limit) {
      uint32_t w; memcpy(&w, src + ip, 4);
      const uint32_t h = (w * 0x1e35a7bdu) >> 18;
      const size_t cand = table[h];
      table[h] = (uint16_t)ip;
      uint32_t cw; memcpy(&cw, src + cand, 4);
      if (cand < ip && cw == w) {
        op = scalar_snappy_literal(op, src + lit, ip - lit);
        size_t len = 4;
        while (ip + len < n && src[cand + len] == src[ip + len]) ++len;
        const size_t off = ip - cand;
        size_t left = len;
        while (left) { /* copy elements of 4..64 bytes, never leaving a rest below 4 */
          size_t piece = left > 64 ? (left - 64 < 4 ? 60 : 64) : left;
          if (piece < 12 && off < 2048) { *op++ = (uint8_t)(1 | ((piece - 4) << 2) | ((off >> 8) << 5)); *op++ = (uint8_t)off; }
          else { *op++ = (uint8_t)(2 | ((piece - 1) << 2)); *op++ = (uint8_t)off; *op++ = (uint8_t)(off >> 8); }
          left -= piece;
        }
        ip += len;
        lit = ip;
      } else {
        ++ip;
      }
    }
  }
  op = scalar_snappy_literal(op, src + lit, n - lit);
  *out_len = (size_t)(op - (uint8_t*)dst_);
  return 0;
}

typedef struct
{
  const uint8_t* in;
  uint8_t* comp;
  uint8_t* mask;
  uint8_t* out;
  size_t* csize;
  size_t lo, hi, chunk, bound;
  int phase, codec;
  snappy_c_fn sc, sd;
} cjob_t;

static void* codec_worker(void* p)
{
  cjob_t* j = (cjob_t*)p;
  for (size_t i = j->lo; i < j->hi; ++i) {
    const uint8_t* in = j->in + i * j->chunk;
    uint8_t* comp = j->comp + i * j->bound;
    uint8_t* out = j->out + i * j->chunk;
    if (j->codec == 1) {
      if (j->phase == 0) {
        if (j->sc) { size_t n = j->bound; j->sc((const char*)in, j->chunk, (char*)comp, &n); j->csize[i] = n; }
        else scalar_snappy_compress((const char*)in, j->chunk, (char*)comp, &j->csize[i]);
      } else {
        size_t n = j->chunk;
        if (j->sd) j->sd((const char*)comp, j->csize[i], (char*)out, &n);
        else oracle_snappy_decompress(comp, j->csize[i], out, j->chunk, &n);
      }
    } else {
      if (j->phase == 0)
        oracle_cascaded_compress(in, j->chunk, 5, 4, 2, 1, 1, comp, j->mask + i * j->bound, &j->csize[i]);
      else {
        size_t n = 0;
        oracle_cascaded_decompress(comp, j->csize[i], out, j->chunk, &n);
      }
    }
  }
  return NULL;
}

int cpu_codec_roundtrip(
    int codec, const uint8_t* data, size_t n_chunks, size_t chunk, int threads, int reps,
    double* t_comp, double* t_decomp, size_t* comp_total, int* used_lib)
{
  if (codec != 1 && codec != 2)
    return 3;
  snappy_c_fn sc = NULL, sd = NULL;
  size_t bound = codec == 1 ? oracle_snappy_max_compressed_size(chunk) : oracle_cascaded_max_compressed_size(chunk);
  *used_lib = 0;
  if (codec == 1) {
    void* h = dlopen("libsnappy.so.1", RTLD_NOW | RTLD_LOCAL);
    if (h) {
      sc = (snappy_c_fn)dlsym(h, "snappy_compress");
      sd = (snappy_c_fn)dlsym(h, "snappy_uncompress");
      size_t (*mx)(size_t) = (size_t (*)(size_t))dlsym(h, "snappy_max_compressed_length");
      if (sc && sd && mx) { *used_lib = 1; if (mx(chunk) > bound) bound = mx(chunk); }
      else sc = sd = NULL;
    }
  }
  uint8_t* comp = (uint8_t*)malloc(n_chunks * bound);
  uint8_t* mask = codec == 2 ? (uint8_t*)malloc(n_chunks * bound) : NULL;
  uint8_t* out = (uint8_t*)malloc(n_chunks * chunk);
  size_t* csize = (size_t*)calloc(n_chunks, sizeof(size_t));
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  cjob_t* jobs = (cjob_t*)malloc(sizeof(cjob_t) * (size_t)threads);
  *t_comp = 1e30;
  *t_decomp = 1e30;
  for (int r = 0; r < reps; ++r) {
    for (int phase = 0; phase < 2; ++phase) {
      const size_t per = (n_chunks + (size_t)threads - 1) / (size_t)threads;
      const double t0 = now();
      for (int k = 0; k < threads; ++k) {
        size_t lo = (size_t)k * per, hi = lo + per;
        if (lo > n_chunks) lo = n_chunks;
        if (hi > n_chunks) hi = n_chunks;
        cjob_t j = {data, comp, mask, out, csize, lo, hi, chunk, bound, phase, codec, sc, sd};
        jobs[k] = j;
        pthread_create(&th[k], NULL, codec_worker, &jobs[k]);
      }
      for (int k = 0; k < threads; ++k)
        pthread_join(th[k], NULL);
      const double dt = now() - t0;
      if (phase == 0 && dt < *t_comp) *t_comp = dt;
      if (phase == 1 && dt < *t_decomp) *t_decomp = dt;
    }
  }
  size_t tot = 0;
  for (size_t i = 0; i < n_chunks; ++i)
    tot += csize[i];
  *comp_total = tot;
  const int bad = memcmp(out, data, n_chunks * chunk) != 0;
  free(comp); free(mask); free(out); free(csize); free(th); free(jobs);
  return bad ? 2 : 0;
}
