// hlif_ref_tool.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
//
// Drives the REFERENCE's high-level managers (LZ4, Snappy, Cascaded; compiled unmodified from
// /root/reference into oracle/_ref/libhipcomp_hlif_ref.so, recipe in oracle/Makefile) on
// files, so that tests/test_hlif_gpu.py can check that containers written by this library
// are read by the reference and the other way round:
//   hlif_ref_tool compress lz4 <chunk_bytes> <hipcompType_t> <in_file> <out_container>
//   hlif_ref_tool compress snappy <chunk_bytes> <in_file> <out_container>
//   hlif_ref_tool compress cascaded <chunk_bytes> <hipcompType_t> <rles> <deltas> <bp> <in_file> <out_container>
//   hlif_ref_tool decompress <in_container> <out_file>      (manager by the container's format byte)
// Built only where /root/reference exists (its headers are needed); the binary travels.
#include "hipcomp/cascaded.hpp"
#include "hipcomp/lz4.hpp"
#include "hipcomp/snappy.hpp"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#define HIP(call)                                                                 \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));             \
      return 2;                                                                   \
    }                                                                             \
  } while (0)

static std::vector<uint8_t> slurp(const char* path)
{
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

template <class Manager>
static int compress_with(Manager& m, const std::vector<uint8_t>& in, const char* out_path)
{
  uint8_t *d_in, *d_out;
  HIP(hipMalloc((void**)&d_in, in.size() ? in.size() : 1));
  HIP(hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice));
  hipcomp::CompressionConfig cfg = m.configure_compression(in.size());
  HIP(hipMalloc((void**)&d_out, cfg.max_compressed_buffer_size + 4096));
  m.compress(d_in, d_out, cfg);
  HIP(hipDeviceSynchronize());
  const size_t bytes = m.get_compressed_output_size(d_out);
  std::vector<uint8_t> out(bytes);
  HIP(hipMemcpy(out.data(), d_out, bytes, hipMemcpyDeviceToHost));
  std::ofstream(out_path, std::ios::binary).write((const char*)out.data(), (std::streamsize)bytes);
  std::printf("compressed %zu -> %zu bytes, %zu chunks, status %d\n", in.size(), bytes, cfg.num_chunks,
              (int)*cfg.get_status());
  return 0; // (the reference only ever writes the status word on failure)
}

template <class Manager>
static int decompress_with(Manager& m, const std::vector<uint8_t>& in, const char* out_path)
{
  uint8_t *d_in, *d_out;
  HIP(hipMalloc((void**)&d_in, in.size()));
  HIP(hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice));
  hipcomp::DecompressionConfig cfg = m.configure_decompression(d_in);
  HIP(hipMalloc((void**)&d_out, cfg.decomp_data_size ? cfg.decomp_data_size : 1));
  m.decompress(d_out, d_in, cfg);
  HIP(hipDeviceSynchronize());
  std::vector<uint8_t> out(cfg.decomp_data_size);
  HIP(hipMemcpy(out.data(), d_out, out.size(), hipMemcpyDeviceToHost));
  std::ofstream(out_path, std::ios::binary).write((const char*)out.data(), (std::streamsize)out.size());
  std::printf("decompressed %zu -> %zu bytes, status %d\n", in.size(), out.size(), (int)*cfg.get_status());
  return *cfg.get_status() == hipcompErrorCannotDecompress ? 1 : 0;
}

int main(int argc, char** argv)
{
  if (argc >= 7 && std::strcmp(argv[1], "compress") == 0 && std::strcmp(argv[2], "lz4") == 0) {
    hipcomp::LZ4Manager m((size_t)std::atoll(argv[3]), (hipcompType_t)std::atoi(argv[4]), 0, 0);
    return compress_with(m, slurp(argv[5]), argv[6]);
  }
  if (argc >= 6 && std::strcmp(argv[1], "compress") == 0 && std::strcmp(argv[2], "snappy") == 0) {
    hipcomp::SnappyManager m((size_t)std::atoll(argv[3]), 0, 0);
    return compress_with(m, slurp(argv[4]), argv[5]);
  }
  if (argc >= 10 && std::strcmp(argv[1], "compress") == 0 && std::strcmp(argv[2], "cascaded") == 0) {
    hipcompBatchedCascadedOpts_t o = hipcompBatchedCascadedDefaultOpts;
    o.chunk_size = (size_t)std::atoll(argv[3]);
    o.type = (hipcompType_t)std::atoi(argv[4]);
    o.num_RLEs = std::atoi(argv[5]);
    o.num_deltas = std::atoi(argv[6]);
    o.use_bp = std::atoi(argv[7]);
    hipcomp::CascadedManager m(o, 0, 0);
    return compress_with(m, slurp(argv[8]), argv[9]);
  }
  if (argc >= 4 && std::strcmp(argv[1], "decompress") == 0) {
    const std::vector<uint8_t> in = slurp(argv[2]);
    if (in.size() < 72)
      return 1;
    uint64_t chunk;
    std::memcpy(&chunk, in.data() + 48, 8); // CommonHeader::uncomp_chunk_size
    switch (in[6]) {                        // CommonHeader::format
    case 0: {
      uint32_t type;
      std::memcpy(&type, in.data() + 64, 4); // LZ4FormatSpecHeader::data_type
      hipcomp::LZ4Manager m((size_t)chunk, (hipcompType_t)type, 0, 0);
      return decompress_with(m, in, argv[3]);
    }
    case 1: {
      hipcomp::SnappyManager m((size_t)chunk, 0, 0);
      return decompress_with(m, in, argv[3]);
    }
    case 4: {
      hipcompBatchedCascadedOpts_t o;
      std::memcpy(&o, in.data() + 64, sizeof(o)); // CascadedFormatSpecHeader::options
      hipcomp::CascadedManager m(o, 0, 0);
      return decompress_with(m, in, argv[3]);
    }
    default:
      std::fprintf(stderr, "format %d\n", (int)in[6]);
      return 3;
    }
  }
  std::fprintf(stderr,
               "usage: hlif_ref_tool compress lz4 <chunk> <type> <in> <out> | compress snappy <chunk> <in> <out>\n"
               "       | compress cascaded <chunk> <type> <rles> <deltas> <bp> <in> <out> | decompress <in> <out>\n");
  return 64;
}
