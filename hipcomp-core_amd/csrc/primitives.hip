// primitives.hip -- whole-array run-length / delta / bit-packing primitives
// (include/hipcomp/primitives.hpp, primitives.h).
//
// Reference: src/RunLengthEncodeGPU.hip:574-633 (hipCUB DeviceRunLengthEncode or
// three kernels + two device scans), src/DeltaGPU.hip:78-164 (one kernel with a
// 1025-entry LDS tile), src/BitPackGPU.hip:184-601 (two reduction kernels + a
// packing kernel that loops over LDS tiles).  Here every operation is a few
// plain passes over the array with wave-level scans (DPP) inside a workgroup:
// these entry points are not on the batched Cascaded path (its layers run fused
// in LDS, cascaded_kernels.hip) and exist for callers of the classes.
#include "wave_utils.hpp"
#include "host_common.hpp"

#include "hipcomp/primitives.h"
#include "hipcomp/primitives.hpp"

#include <stdexcept>
#include <string>
#include <type_traits>

namespace hcamd {
namespace prim {

constexpr int kBlock = 256;          // threads per workgroup
constexpr int kPerThread = 4;
constexpr int kTile = kBlock * kPerThread; // elements per workgroup

inline size_t tiles_of(size_t num) { return num == 0 ? 1 : (num + kTile - 1) / kTile; }

// ---- workgroup-wide exclusive scan of one uint32 per thread (256 threads) ----
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t& total, uint32_t* wave_sums)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t incl = wave_scan_add_u32(v);
  if (lane == 63)
    wave_sums[wave] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kBlock / 64; ++w) {
    const uint32_t s = wave_sums[w];
    before += w < wave ? s : 0u;
    all += s;
  }
  __syncthreads();
  total = all;
  return before + incl - v;
}

// ---- run-length encoding -----------------------------------------------------
// pass 1: runs that start in each tile
template <typename T>
__global__ __launch_bounds__(kBlock) void rle_count_kernel(
    const T* __restrict__ in, const size_t* __restrict__ num_dev, size_t num_host, uint32_t* __restrict__ tile_runs)
{
  __shared__ uint32_t wave_sums[kBlock / 64];
  const size_t num = num_dev ? *num_dev : num_host;
  const size_t base = (size_t)blockIdx.x * kTile;
  uint32_t starts = 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const size_t i = base + (size_t)threadIdx.x * kPerThread + k;
    if (i < num && (i == 0 || in[i] != in[i - 1]))
      ++starts;
  }
  uint32_t total;
  block_exclusive_scan(starts, total, wave_sums);
  if (threadIdx.x == 0)
    tile_runs[blockIdx.x] = total;
}

// pass 2 (one workgroup): exclusive scan of the tile counts -> tile offsets, total -> *num_out
__global__ __launch_bounds__(kBlock) void rle_offsets_kernel(
    const uint32_t* __restrict__ tile_runs, uint64_t* __restrict__ tile_offsets, size_t tiles, size_t* __restrict__ num_out)
{
  __shared__ uint32_t wave_sums[kBlock / 64];
  uint64_t carry = 0;
  for (size_t t0 = 0; t0 < tiles; t0 += kBlock) {
    const size_t t = t0 + threadIdx.x;
    const uint32_t v = t < tiles ? tile_runs[t] : 0u;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(v, total, wave_sums);
    if (t < tiles)
      tile_offsets[t] = carry + excl;
    carry += total;
  }
  if (threadIdx.x == 0)
    *num_out = (size_t)carry;
}

// pass 3: values and start positions of the runs
template <typename T>
__global__ __launch_bounds__(kBlock) void rle_write_kernel(
    const T* __restrict__ in, const size_t* __restrict__ num_dev, size_t num_host,
    const uint64_t* __restrict__ tile_offsets, T* out_values_host, T* const* out_values_dev,
    uint64_t* __restrict__ run_starts)
{
  __shared__ uint32_t wave_sums[kBlock / 64];
  const size_t num = num_dev ? *num_dev : num_host;
  T* const out_values = out_values_dev ? *out_values_dev : out_values_host;
  const size_t base = (size_t)blockIdx.x * kTile;
  bool st[kPerThread];
  uint32_t starts = 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const size_t i = base + (size_t)threadIdx.x * kPerThread + k;
    st[k] = i < num && (i == 0 || in[i] != in[i - 1]);
    starts += st[k] ? 1u : 0u;
  }
  uint32_t total;
  uint64_t at = tile_offsets[blockIdx.x] + block_exclusive_scan(starts, total, wave_sums);
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const size_t i = base + (size_t)threadIdx.x * kPerThread + k;
    if (st[k]) {
      out_values[at] = in[i];
      run_starts[at] = i;
      ++at;
    }
  }
}

// pass 4: run lengths from the start positions
template <typename C>
__global__ __launch_bounds__(kBlock) void rle_lengths_kernel(
    const uint64_t* __restrict__ run_starts, const size_t* __restrict__ num_out, const size_t* __restrict__ num_dev,
    size_t num_host, C* out_counts_host, C* const* out_counts_dev)
{
  const size_t num = num_dev ? *num_dev : num_host;
  const size_t runs = *num_out;
  C* const out_counts = out_counts_dev ? *out_counts_dev : out_counts_host;
  for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < runs; j += (size_t)gridDim.x * kBlock)
    out_counts[j] = (C)((j + 1 < runs ? run_starts[j + 1] : (uint64_t)num) - run_starts[j]);
}

// ---- delta ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void delta_kernel(
    T* const* __restrict__ out_ptr, const T* __restrict__ in, const size_t* __restrict__ num_dev)
{
  typedef typename std::make_unsigned<T>::type U;
  const size_t num = *num_dev;
  T* const out = *out_ptr;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < num; i += (size_t)gridDim.x * kBlock)
    out[i] = (T)((U)in[i] - (i ? (U)in[i - 1] : (U)0)); // wrap-around arithmetic
}

// ---- bit packing ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void minmax_tiles_kernel(
    const T* __restrict__ in, const size_t* __restrict__ num_dev, T* __restrict__ tile_min, T* __restrict__ tile_max)
{
  __shared__ T smin[kBlock], smax[kBlock];
  const size_t num = *num_dev;
  const size_t base = (size_t)blockIdx.x * kTile;
  if (base >= num)
    return;
  T lo = in[base], hi = lo;
  for (int k = 0; k < kPerThread; ++k) {
    const size_t i = base + (size_t)k * kBlock + threadIdx.x;
    if (i < num) {
      const T v = in[i];
      lo = v < lo ? v : lo;
      hi = v > hi ? v : hi;
    }
  }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  for (int d = kBlock / 2; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) {
      smin[threadIdx.x] = smin[threadIdx.x + d] < smin[threadIdx.x] ? smin[threadIdx.x + d] : smin[threadIdx.x];
      smax[threadIdx.x] = smax[threadIdx.x + d] > smax[threadIdx.x] ? smax[threadIdx.x + d] : smax[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tile_min[blockIdx.x] = smin[0];
    tile_max[blockIdx.x] = smax[0];
  }
}

// one workgroup: min / max over the tiles, frame of reference and bit width
// (reference bitPackConfigFinalizeKernel, BitPackGPU.hip:280-294: the width of
// the range taken in 32 bits, in 64 for 8-byte types)
template <typename T>
__global__ __launch_bounds__(kBlock) void minmax_final_kernel(
    const T* __restrict__ tile_min, const T* __restrict__ tile_max, const size_t* __restrict__ num_dev,
    T* const* __restrict__ min_out, unsigned char* const* __restrict__ bits_out)
{
  __shared__ T smin[kBlock], smax[kBlock];
  const size_t num = *num_dev;
  const size_t tiles = (num + kTile - 1) / kTile;
  if (tiles == 0) {
    if (threadIdx.x == 0) {
      **min_out = 0;
      **bits_out = 0;
    }
    return;
  }
  T lo = tile_min[0], hi = tile_max[0];
  for (size_t t = threadIdx.x; t < tiles; t += kBlock) {
    lo = tile_min[t] < lo ? tile_min[t] : lo;
    hi = tile_max[t] > hi ? tile_max[t] : hi;
  }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  for (int d = kBlock / 2; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) {
      smin[threadIdx.x] = smin[threadIdx.x + d] < smin[threadIdx.x] ? smin[threadIdx.x + d] : smin[threadIdx.x];
      smax[threadIdx.x] = smax[threadIdx.x + d] > smax[threadIdx.x] ? smax[threadIdx.x + d] : smax[threadIdx.x];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    **min_out = smin[0];
    if (sizeof(T) > 4) {
      const uint64_t range = (uint64_t)smax[0] - (uint64_t)smin[0];
      **bits_out = (unsigned char)(range ? 64 - __builtin_clzll(range) : 0);
    } else {
      const uint32_t range = (uint32_t)smax[0] - (uint32_t)smin[0];
      **bits_out = (unsigned char)(range ? 32 - __builtin_clz(range) : 0);
    }
  }
}

// value i at bit i * bits of the output, least significant bit first
// (reference bitPackKernel, BitPackGPU.hip:296-386)
template <typename T, typename W>
__global__ __launch_bounds__(kBlock) void bitpack_kernel(
    const unsigned char* const* __restrict__ bits_ptr, const T* const* __restrict__ min_ptr,
    W* const* __restrict__ out_ptr, const T* __restrict__ in, const size_t* __restrict__ num_dev)
{
  typedef typename std::make_unsigned<T>::type U;
  constexpr uint64_t B = sizeof(W) * 8;
  const size_t num = *num_dev;
  const uint64_t bits = **bits_ptr;
  if (bits == 0)
    return; // every value equals the frame of reference: nothing to store
  const T ref = **min_ptr;
  W* const out = *out_ptr;
  const uint64_t words = ((uint64_t)num * bits + B - 1) / B;
  for (uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x; w < words; w += (uint64_t)gridDim.x * kBlock) {
    const uint64_t bit0 = w * B;
    uint64_t i = bit0 / bits;
    W val = 0;
    for (; i < num && i * bits < bit0 + B; ++i) {
      const W v = (W)(U)((U)in[i] - (U)ref);
      const int64_t off = (int64_t)(i * bits) - (int64_t)bit0;
      val |= off >= 0 ? (W)(v << off) : (W)(v >> -off);
    }
    out[w] = val;
  }
}

// ---- host side -------------------------------------------------------------------
inline int type_size(hipcompType_t t)
{
  switch (t) {
  case HIPCOMP_TYPE_CHAR: case HIPCOMP_TYPE_UCHAR: return 1;
  case HIPCOMP_TYPE_SHORT: case HIPCOMP_TYPE_USHORT: return 2;
  case HIPCOMP_TYPE_INT: case HIPCOMP_TYPE_UINT: return 4;
  case HIPCOMP_TYPE_LONGLONG: case HIPCOMP_TYPE_ULONGLONG: return 8;
  default: return 0;
  }
}

// f.template operator()<T>() for the C type behind a hipcompType_t
template <typename F>
void with_type(hipcompType_t t, F&& f)
{
  switch (t) {
  case HIPCOMP_TYPE_CHAR: f((signed char)0); break;
  case HIPCOMP_TYPE_UCHAR: f((unsigned char)0); break;
  case HIPCOMP_TYPE_SHORT: f((short)0); break;
  case HIPCOMP_TYPE_USHORT: f((unsigned short)0); break;
  case HIPCOMP_TYPE_INT: f((int)0); break;
  case HIPCOMP_TYPE_UINT: f((unsigned int)0); break;
  case HIPCOMP_TYPE_LONGLONG: f((long long)0); break;
  case HIPCOMP_TYPE_ULONGLONG: f((unsigned long long)0); break;
  default: throw std::runtime_error("Unknown type: " + std::to_string((int)t));
  }
}

inline void check_launch(const char* what)
{
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

inline unsigned grid_for(size_t n)
{
  const size_t b = (n + kBlock - 1) / kBlock;
  return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

// workspace of the run-length encoder: tile counts (u32), tile offsets (u64), run starts (u64)
struct RleSpace
{
  uint32_t* tile_runs;
  uint64_t* tile_offsets;
  uint64_t* run_starts;
  size_t bytes;
};
inline RleSpace rle_space(void* base, size_t num)
{
  const size_t tiles = tiles_of(num);
  uintptr_t p = ((uintptr_t)base + 7) & ~(uintptr_t)7;
  RleSpace s;
  s.tile_offsets = (uint64_t*)p;
  p += tiles * 8;
  s.run_starts = (uint64_t*)p;
  p += (num ? num : 1) * 8;
  s.tile_runs = (uint32_t*)p;
  p += tiles * 4;
  s.bytes = (size_t)(p - (uintptr_t)base) + 8;
  return s;
}

void rle_run(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void* outValues, void* const* outValuesPtr,
    hipcompType_t countType, void* outCounts, void* const* outCountsPtr, size_t* numOutDevice, const void* in,
    const size_t* numDevice, size_t num, hipStream_t stream)
{
  if (type_size(countType) == 0)
    throw std::runtime_error("Unknown type: " + std::to_string((int)countType));
  const size_t need = hipcomp::RunLengthEncodeGPU::requiredWorkspaceSize(num, valueType, countType);
  if (workspaceSize < need)
    throw std::runtime_error("Invalid workspace size: " + std::to_string(workspaceSize) + ", need at least "
                             + std::to_string(need));
  const RleSpace sp = rle_space(workspace, num);
  const size_t tiles = tiles_of(num);
  with_type(valueType, [&](auto v) {
    typedef decltype(v) T;
    rle_count_kernel<T><<<(unsigned)tiles, kBlock, 0, stream>>>((const T*)in, numDevice, num, sp.tile_runs);
    rle_offsets_kernel<<<1, kBlock, 0, stream>>>(sp.tile_runs, sp.tile_offsets, tiles, numOutDevice);
    rle_write_kernel<T><<<(unsigned)tiles, kBlock, 0, stream>>>(
        (const T*)in, numDevice, num, sp.tile_offsets, (T*)outValues, (T* const*)outValuesPtr, sp.run_starts);
  });
  with_type(countType, [&](auto c) {
    typedef decltype(c) C;
    rle_lengths_kernel<C><<<grid_for(num), kBlock, 0, stream>>>(
        sp.run_starts, numOutDevice, numDevice, num, (C*)outCounts, (C* const*)outCountsPtr);
  });
  check_launch("run-length encoding kernels");
}

} // namespace prim
} // namespace hcamd

namespace hipcomp
{
using namespace hcamd::prim;

void RunLengthEncodeGPU::compress(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void* outValues, hipcompType_t countType,
    void* outCounts, size_t* numOutDevice, const void* in, size_t num, hipStream_t stream)
{
  rle_run(workspace, workspaceSize, valueType, outValues, nullptr, countType, outCounts, nullptr, numOutDevice, in,
          nullptr, num, stream);
}

void RunLengthEncodeGPU::compressDownstream(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void** outValuesPtr, hipcompType_t countType,
    void** outCountsPtr, size_t* numOutDevice, const void* in, const size_t* numDevice, size_t maxNum,
    hipStream_t stream)
{
  rle_run(workspace, workspaceSize, valueType, nullptr, outValuesPtr, countType, nullptr, outCountsPtr, numOutDevice,
          in, numDevice, maxNum, stream);
}

size_t RunLengthEncodeGPU::requiredWorkspaceSize(size_t num, hipcompType_t valueType, hipcompType_t countType)
{
  if (type_size(valueType) == 0 || type_size(countType) == 0)
    throw std::runtime_error("Unknown type: " + std::to_string((int)(type_size(valueType) ? countType : valueType)));
  return rle_space(nullptr, num).bytes;
}

void DeltaGPU::compress(
    void* /*workspace*/, size_t /*workspaceSize*/, hipcompType_t valueType, void** outValuesPtr,
    const void* inValues, const size_t* numDevice, size_t maxNum, hipStream_t stream)
{
  with_type(valueType, [&](auto v) {
    typedef decltype(v) T;
    delta_kernel<T><<<grid_for(maxNum), kBlock, 0, stream>>>((T* const*)outValuesPtr, (const T*)inValues, numDevice);
  });
  check_launch("delta kernel");
}

size_t DeltaGPU::requiredWorkspaceSize(size_t /*num*/, hipcompType_t type)
{
  if (type_size(type) == 0)
    throw std::runtime_error("Unknown type: " + std::to_string((int)type));
  return 0; // reference DeltaGPU.hip:166-170: none needed
}

void BitPackGPU::compress(
    void* workspace, size_t workspaceSize, hipcompType_t inType, void* const* outPtr, const void* in,
    const size_t* numDevice, size_t maxNum, void* const* minValueDevicePtr, unsigned char* const* numBitsDevicePtr,
    hipStream_t stream)
{
  const size_t need = requiredWorkspaceSize(maxNum, inType);
  if (workspaceSize < need)
    throw std::runtime_error("Insufficient workspace size: " + std::to_string(workspaceSize) + ", need "
                             + std::to_string(need));
  const size_t tiles = tiles_of(maxNum);
  with_type(inType, [&](auto v) {
    typedef decltype(v) T;
    typedef typename std::conditional<(sizeof(T) > 4), uint64_t, uint32_t>::type W;
    T* const tile_min = (T*)(((uintptr_t)workspace + 7) & ~(uintptr_t)7);
    T* const tile_max = tile_min + tiles;
    minmax_tiles_kernel<T><<<(unsigned)tiles, kBlock, 0, stream>>>((const T*)in, numDevice, tile_min, tile_max);
    minmax_final_kernel<T><<<1, kBlock, 0, stream>>>(tile_min, tile_max, numDevice, (T* const*)minValueDevicePtr,
                                                    numBitsDevicePtr);
    bitpack_kernel<T, W><<<grid_for(maxNum), kBlock, 0, stream>>>(
        numBitsDevicePtr, (const T* const*)minValueDevicePtr, (W* const*)outPtr, (const T*)in, numDevice);
  });
  check_launch("bit packing kernels");
}

size_t BitPackGPU::requiredWorkspaceSize(size_t num, hipcompType_t type)
{
  const int s = type_size(type);
  if (s == 0)
    throw std::runtime_error("Unknown type: " + std::to_string((int)type));
  return 2 * tiles_of(num) * (size_t)s + 16;
}

} // namespace hipcomp

// ---- C binding ---------------------------------------------------------------------
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

hipcompStatus_t hipcompRunLengthEncodeGetWorkspaceSize(
    size_t num, hipcompType_t valueType, hipcompType_t countType, size_t* workspace_bytes)
{
  static const char* fn = "hipcompRunLengthEncodeGetWorkspaceSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, workspace_bytes);
  return guarded(fn, [&] { *workspace_bytes = hipcomp::RunLengthEncodeGPU::requiredWorkspaceSize(num, valueType, countType); });
}

hipcompStatus_t hipcompRunLengthEncodeCompress(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void* outValues, hipcompType_t countType,
    void* outCounts, size_t* numOutDevice, const void* in, size_t num, hipStream_t stream)
{
  return guarded("hipcompRunLengthEncodeCompress()", [&] {
    hipcomp::RunLengthEncodeGPU::compress(workspace, workspaceSize, valueType, outValues, countType, outCounts,
                                          numOutDevice, in, num, stream);
  });
}

hipcompStatus_t hipcompRunLengthEncodeCompressDownstream(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void** outValuesPtr, hipcompType_t countType,
    void** outCountsPtr, size_t* numOutDevice, const void* in, const size_t* numDevice, size_t maxNum,
    hipStream_t stream)
{
  return guarded("hipcompRunLengthEncodeCompressDownstream()", [&] {
    hipcomp::RunLengthEncodeGPU::compressDownstream(workspace, workspaceSize, valueType, outValuesPtr, countType,
                                                    outCountsPtr, numOutDevice, in, numDevice, maxNum, stream);
  });
}

hipcompStatus_t hipcompDeltaGetWorkspaceSize(size_t num, hipcompType_t type, size_t* workspace_bytes)
{
  static const char* fn = "hipcompDeltaGetWorkspaceSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, workspace_bytes);
  return guarded(fn, [&] { *workspace_bytes = hipcomp::DeltaGPU::requiredWorkspaceSize(num, type); });
}

hipcompStatus_t hipcompDeltaCompress(
    void* workspace, size_t workspaceSize, hipcompType_t valueType, void** outValuesPtr, const void* inValues,
    const size_t* numDevice, size_t maxNum, hipStream_t stream)
{
  return guarded("hipcompDeltaCompress()", [&] {
    hipcomp::DeltaGPU::compress(workspace, workspaceSize, valueType, outValuesPtr, inValues, numDevice, maxNum, stream);
  });
}

hipcompStatus_t hipcompBitPackGetWorkspaceSize(size_t num, hipcompType_t type, size_t* workspace_bytes)
{
  static const char* fn = "hipcompBitPackGetWorkspaceSize()";
  HCAMD_REQUIRE_NOT_NULL(fn, workspace_bytes);
  return guarded(fn, [&] { *workspace_bytes = hipcomp::BitPackGPU::requiredWorkspaceSize(num, type); });
}

hipcompStatus_t hipcompBitPackCompress(
    void* workspace, size_t workspaceSize, hipcompType_t inType, void* const* outPtr, const void* in,
    const size_t* numDevice, size_t maxNum, void* const* minValueDevicePtr, unsigned char* const* numBitsDevicePtr,
    hipStream_t stream)
{
  return guarded("hipcompBitPackCompress()", [&] {
    hipcomp::BitPackGPU::compress(workspace, workspaceSize, inType, outPtr, in, numDevice, maxNum, minValueDevicePtr,
                                  numBitsDevicePtr, stream);
  });
}

} // extern "C"
