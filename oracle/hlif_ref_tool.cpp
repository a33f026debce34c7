// hlif_ref_tool.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
//
// Drives the REFERENCE's high-level LZ4 manager (compiled unmodified from
// /root/reference into oracle/_ref/libhipcomp_hlif_ref.so, recipe in oracle/Makefile) on
// files, so that tests/test_hlif_gpu.py can check that containers written by this library
// are read by the reference and the other way round:
//   hlif_ref_tool compress <chunk_bytes> <hipcompType_t> <in_file> <out_container>
//   hlif_ref_tool decompress <in_container> <out_file>
// Built only where /root/reference exists (its headers are needed); the binary travels.
#include "hipcomp/lz4.hpp"

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

int main(int argc, char** argv)
{
  if (argc >= 6 && std::strcmp(argv[1], "compress") == 0) {
    const size_t chunk = (size_t)std::atoll(argv[2]);
    const hipcompType_t type = (hipcompType_t)std::atoi(argv[3]);
    const std::vector<uint8_t> in = slurp(argv[4]);
    uint8_t *d_in, *d_out;
    HIP(hipMalloc((void**)&d_in, in.size() ? in.size() : 1));
    HIP(hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice));
    hipcomp::LZ4Manager m(chunk, type, 0, 0);
    hipcomp::CompressionConfig cfg = m.configure_compression(in.size());
    HIP(hipMalloc((void**)&d_out, cfg.max_compressed_buffer_size + 4096));
    m.compress(d_in, d_out, cfg);
    HIP(hipDeviceSynchronize());
    const size_t bytes = m.get_compressed_output_size(d_out);
    std::vector<uint8_t> out(bytes);
    HIP(hipMemcpy(out.data(), d_out, bytes, hipMemcpyDeviceToHost));
    std::ofstream(argv[5], std::ios::binary).write((const char*)out.data(), (std::streamsize)bytes);
    std::printf("compressed %zu -> %zu bytes, %zu chunks, status %d\n", in.size(), bytes, cfg.num_chunks,
                (int)*cfg.get_status());
    return 0; // (the reference only ever writes the status word on failure)
  }
  if (argc >= 4 && std::strcmp(argv[1], "decompress") == 0) {
    const std::vector<uint8_t> in = slurp(argv[2]);
    if (in.size() < 72)
      return 1;
    uint64_t chunk;
    uint32_t type;
    std::memcpy(&chunk, in.data() + 48, 8); // CommonHeader::uncomp_chunk_size
    std::memcpy(&type, in.data() + 64, 4);  // LZ4FormatSpecHeader::data_type
    uint8_t *d_in, *d_out;
    HIP(hipMalloc((void**)&d_in, in.size()));
    HIP(hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice));
    hipcomp::LZ4Manager m((size_t)chunk, (hipcompType_t)type, 0, 0);
    hipcomp::DecompressionConfig cfg = m.configure_decompression(d_in);
    HIP(hipMalloc((void**)&d_out, cfg.decomp_data_size ? cfg.decomp_data_size : 1));
    m.decompress(d_out, d_in, cfg);
    HIP(hipDeviceSynchronize());
    std::vector<uint8_t> out(cfg.decomp_data_size);
    HIP(hipMemcpy(out.data(), d_out, out.size(), hipMemcpyDeviceToHost));
    std::ofstream(argv[3], std::ios::binary).write((const char*)out.data(), (std::streamsize)out.size());
    std::printf("decompressed %zu -> %zu bytes, status %d\n", in.size(), out.size(), (int)*cfg.get_status());
    return *cfg.get_status() == hipcompErrorCannotDecompress ? 1 : 0;
  }
  std::fprintf(stderr, "usage: hlif_ref_tool compress <chunk> <type> <in> <out> | decompress <in> <out>\n");
  return 64;
}
