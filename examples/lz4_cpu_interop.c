/*
 * lz4_cpu_interop.c -- the batched GPU LZ4 codec and liblz4 on the CPU read each
 * other's data (what the reference lineage shipped as examples/lz4_cpu_compression
 * and lz4_cpu_decompression; reference CHANGELOG.md:65-66).
 *
 *   1. chunks compressed on the GPU (hipcompBatchedLZ4CompressAsync) are decoded by
 *      liblz4's LZ4_decompress_safe, block by block;
 *   2. the same blocks wrapped into ONE LZ4 frame (hipcompLZ4FrameFromBlocks) are
 *      decoded by liblz4's frame API, i.e. what the `lz4` command line tool reads;
 *   3. chunks compressed by liblz4 (LZ4_compress_default) are decompressed on the GPU
 *      (hipcompBatchedLZ4DecompressAsync).
 *
 * Build: make -C examples      Run: examples/lz4_cpu_interop [chunks] [chunk_bytes]
 * Plain C against the C API; liblz4 is loaded with dlopen (the image has the runtime
 * library, not its headers).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "hipcomp/lz4.h"
#include "hipcomp/lz4_interop.h"

#define HIP(call)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)
#define OK(call)                                                       \
  do {                                                                 \
    hipcompStatus_t s_ = (call);                                       \
    if (s_ != hipcompSuccess) {                                        \
      fprintf(stderr, "%s:%d %s -> status %d\n", __FILE__, __LINE__, #call, (int)s_); \
      return 3;                                                        \
    }                                                                  \
  } while (0)

typedef int (*lz4_decompress_safe_fn)(const char*, char*, int, int);
typedef int (*lz4_compress_default_fn)(const char*, char*, int, int);
typedef int (*lz4_bound_fn)(int);
typedef size_t (*lz4f_create_fn)(void**, unsigned);
typedef size_t (*lz4f_free_fn)(void*);
typedef size_t (*lz4f_decompress_fn)(void*, void*, size_t*, const void*, size_t*, const void*);
typedef unsigned (*lz4f_iserror_fn)(size_t);

int main(int argc, char** argv)
{
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 64;
  const size_t chunk = argc > 2 ? (size_t)atol(argv[2]) : 65536;
  void* lz4 = dlopen("liblz4.so.1", RTLD_NOW);
  if (!lz4) {
    fprintf(stderr, "liblz4.so.1 not found\n");
    return 77;
  }
  lz4_decompress_safe_fn lz4_decompress_safe = (lz4_decompress_safe_fn)dlsym(lz4, "LZ4_decompress_safe");
  lz4_compress_default_fn lz4_compress_default = (lz4_compress_default_fn)dlsym(lz4, "LZ4_compress_default");
  lz4_bound_fn lz4_bound = (lz4_bound_fn)dlsym(lz4, "LZ4_compressBound");
  lz4f_create_fn lz4f_create = (lz4f_create_fn)dlsym(lz4, "LZ4F_createDecompressionContext");
  lz4f_free_fn lz4f_free = (lz4f_free_fn)dlsym(lz4, "LZ4F_freeDecompressionContext");
  lz4f_decompress_fn lz4f_decompress = (lz4f_decompress_fn)dlsym(lz4, "LZ4F_decompress");
  lz4f_iserror_fn lz4f_iserror = (lz4f_iserror_fn)dlsym(lz4, "LZ4F_isError");

  /* some text-like data */
  const size_t total = n * chunk;
  uint8_t* host = (uint8_t*)malloc(total);
  uint32_t x = 12345;
  for (size_t i = 0; i < total; ++i) {
    x = x * 1664525u + 1013904223u;
    host[i] = (uint8_t)("the quick brown fox jumps over the lazy dog, 0123456789 "[(i * 7 + (x >> 29)) % 56]);
  }

  /* ---- 1. GPU compress ---- */
  size_t max_out = 0, temp_bytes = 0;
  OK(hipcompBatchedLZ4CompressGetMaxOutputChunkSize(chunk, hipcompBatchedLZ4DefaultOpts, &max_out));
  OK(hipcompBatchedLZ4CompressGetTempSize(n, chunk, hipcompBatchedLZ4DefaultOpts, &temp_bytes));
  uint8_t *d_in, *d_comp, *d_temp, *d_out;
  void **d_in_ptrs, **d_comp_ptrs, **d_out_ptrs;
  size_t *d_in_bytes, *d_comp_bytes, *d_caps, *d_actual;
  hipcompStatus_t* d_status;
  HIP(hipMalloc((void**)&d_in, total));
  HIP(hipMalloc((void**)&d_comp, n * max_out));
  HIP(hipMalloc((void**)&d_out, total));
  HIP(hipMalloc((void**)&d_temp, temp_bytes ? temp_bytes : 8));
  HIP(hipMalloc((void**)&d_in_ptrs, n * sizeof(void*)));
  HIP(hipMalloc((void**)&d_comp_ptrs, n * sizeof(void*)));
  HIP(hipMalloc((void**)&d_out_ptrs, n * sizeof(void*)));
  HIP(hipMalloc((void**)&d_in_bytes, n * sizeof(size_t)));
  HIP(hipMalloc((void**)&d_comp_bytes, n * sizeof(size_t)));
  HIP(hipMalloc((void**)&d_caps, n * sizeof(size_t)));
  HIP(hipMalloc((void**)&d_actual, n * sizeof(size_t)));
  HIP(hipMalloc((void**)&d_status, n * sizeof(hipcompStatus_t)));
  void** h_ptrs = (void**)malloc(n * sizeof(void*));
  size_t* h_sizes = (size_t*)malloc(n * sizeof(size_t));
  for (size_t i = 0; i < n; ++i) { h_ptrs[i] = d_in + i * chunk; h_sizes[i] = chunk; }
  HIP(hipMemcpy(d_in_ptrs, h_ptrs, n * sizeof(void*), hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_in_bytes, h_sizes, n * sizeof(size_t), hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_caps, h_sizes, n * sizeof(size_t), hipMemcpyHostToDevice));
  for (size_t i = 0; i < n; ++i) h_ptrs[i] = d_comp + i * max_out;
  HIP(hipMemcpy(d_comp_ptrs, h_ptrs, n * sizeof(void*), hipMemcpyHostToDevice));
  for (size_t i = 0; i < n; ++i) h_ptrs[i] = d_out + i * chunk;
  HIP(hipMemcpy(d_out_ptrs, h_ptrs, n * sizeof(void*), hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_in, host, total, hipMemcpyHostToDevice));
  OK(hipcompBatchedLZ4CompressAsync((const void* const*)d_in_ptrs, d_in_bytes, chunk, n, d_temp, temp_bytes,
                                    d_comp_ptrs, d_comp_bytes, hipcompBatchedLZ4DefaultOpts, 0));
  HIP(hipDeviceSynchronize());
  uint8_t* h_comp = (uint8_t*)malloc(n * max_out);
  size_t* h_comp_bytes = (size_t*)malloc(n * sizeof(size_t));
  HIP(hipMemcpy(h_comp, d_comp, n * max_out, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(h_comp_bytes, d_comp_bytes, n * sizeof(size_t), hipMemcpyDeviceToHost));

  /* liblz4 decodes every block */
  uint8_t* back = (uint8_t*)malloc(chunk);
  size_t total_comp = 0;
  for (size_t i = 0; i < n; ++i) {
    const int got = lz4_decompress_safe((const char*)h_comp + i * max_out, (char*)back, (int)h_comp_bytes[i], (int)chunk);
    if (got != (int)chunk || memcmp(back, host + i * chunk, chunk) != 0) {
      fprintf(stderr, "chunk %zu: LZ4_decompress_safe -> %d\n", i, got);
      return 1;
    }
    total_comp += h_comp_bytes[i];
  }
  printf("1. %zu GPU-compressed chunks (ratio %.2f) decoded by LZ4_decompress_safe: OK\n", n, (double)total / total_comp);

  /* ---- 2. one LZ4 frame out of the blocks, decoded by liblz4's frame API ---- */
  const void** blk = (const void**)malloc(n * sizeof(void*));
  for (size_t i = 0; i < n; ++i) blk[i] = h_comp + i * max_out;
  const size_t frame_cap = hipcompLZ4FrameBound(n, total_comp);
  uint8_t* frame = (uint8_t*)malloc(frame_cap);
  size_t frame_bytes = 0;
  OK(hipcompLZ4FrameFromBlocks(blk, h_comp_bytes, h_sizes, n, frame, frame_cap, &frame_bytes));
  void* dctx = NULL;
  if (lz4f_iserror(lz4f_create(&dctx, 100))) return 1;
  uint8_t* all = (uint8_t*)malloc(total);
  size_t in_at = 0, out_at = 0;
  while (in_at < frame_bytes) {
    size_t dst = total - out_at, src = frame_bytes - in_at;
    const size_t r = lz4f_decompress(dctx, all + out_at, &dst, frame + in_at, &src, NULL);
    if (lz4f_iserror(r)) { fprintf(stderr, "LZ4F_decompress failed\n"); return 1; }
    in_at += src;
    out_at += dst;
    if (r == 0) break;
  }
  lz4f_free(dctx);
  if (out_at != total || memcmp(all, host, total) != 0) { fprintf(stderr, "frame round trip differs\n"); return 1; }
  printf("2. frame of %zu bytes decoded by LZ4F_decompress: OK\n", frame_bytes);

  /* ---- 3. liblz4 compresses, the GPU decompresses ---- */
  const size_t bound = (size_t)lz4_bound((int)chunk);
  uint8_t* h_cpu = (uint8_t*)malloc(n * bound);
  uint8_t* d_cpu;
  HIP(hipMalloc((void**)&d_cpu, n * bound));
  for (size_t i = 0; i < n; ++i) {
    h_comp_bytes[i] = (size_t)lz4_compress_default((const char*)host + i * chunk, (char*)h_cpu + i * bound, (int)chunk, (int)bound);
    h_ptrs[i] = d_cpu + i * bound;
  }
  HIP(hipMemcpy(d_cpu, h_cpu, n * bound, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_comp_ptrs, h_ptrs, n * sizeof(void*), hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_comp_bytes, h_comp_bytes, n * sizeof(size_t), hipMemcpyHostToDevice));
  size_t dtemp_bytes = 0;
  OK(hipcompBatchedLZ4DecompressGetTempSize(n, chunk, &dtemp_bytes));
  uint8_t* d_dtemp;
  HIP(hipMalloc((void**)&d_dtemp, dtemp_bytes ? dtemp_bytes : 8));
  HIP(hipMemset(d_out, 0, total));
  OK(hipcompBatchedLZ4DecompressAsync((const void* const*)d_comp_ptrs, d_comp_bytes, d_caps, d_actual, n, d_dtemp,
                                      dtemp_bytes, d_out_ptrs, d_status, 0));
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(all, d_out, total, hipMemcpyDeviceToHost));
  hipcompStatus_t* h_status = (hipcompStatus_t*)malloc(n * sizeof(hipcompStatus_t));
  HIP(hipMemcpy(h_status, d_status, n * sizeof(hipcompStatus_t), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i)
    if (h_status[i] != hipcompSuccess) { fprintf(stderr, "chunk %zu: status %d\n", i, (int)h_status[i]); return 1; }
  if (memcmp(all, host, total) != 0) { fprintf(stderr, "GPU decompression of liblz4 blocks differs\n"); return 1; }
  printf("3. %zu liblz4-compressed chunks decompressed on the GPU: OK\n", n);
  return 0;
}
