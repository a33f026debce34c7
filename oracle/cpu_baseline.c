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
