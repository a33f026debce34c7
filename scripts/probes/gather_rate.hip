// Probe: vector-memory instruction throughput per CU on gfx950 for the access
// shapes of the LZ4 encoder: coalesced window loads, 64-line gathers (candidate
// verification), 2-byte gathers / scatters (a hash table kept in global memory).
// Build: hipcc -O3 --offload-arch=gfx950 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef const __attribute__((address_space(1))) uint8_t* cg;
typedef __attribute__((address_space(1))) uint8_t* gp;

// K: 0 coalesced aligned dword, 1 byte-stride unaligned dword (window words),
//    2 dword gather over 64 KiB per wave (64 distinct 128-B lines), 3 dword gather over
//    8 KiB per wave, 4 ushort gather over 32 KiB, 5 short scatter over 32 KiB,
//    6 dword gather where only 16 lanes are active, 7 strided gather (1 KiB stride: a pathological case)
//    (2, 4, 5, 6: pseudo-random offsets inside the region, as a hash table / candidate positions give)
template <int K>
__global__ __launch_bounds__(256) void k_rate(uint8_t* buf, unsigned long long* out, int iters, uint32_t region)
{
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  gp base = (gp)buf + (size_t)wave * region;
  uint32_t acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t j = (uint32_t)it * 8u + u;
      uint32_t off;
      if (K == 0) off = lane * 4 + ((j * 256) & (region - 1));
      if (K == 1) off = lane + ((j * 61) & (region - 256));
      uint32_t hh = (lane + 64u * j) * 2654435761u; hh ^= hh >> 15; hh *= 0x2c1b3c6du; hh ^= hh >> 12;
      if (K == 2) off = hh & (region - 1) & ~3u;
      if (K == 3) off = lane * 128 + ((j * 4) & 124);
      if (K == 4) off = hh & (region - 1) & ~1u;
      if (K == 5) off = hh & (region - 1) & ~1u;
      if (K == 6) off = hh & (region - 1) & ~3u;
      if (K == 7) off = (lane * 8 + (j & 7)) * 128 + ((j * 4) & 124);
      if (K == 4) {
        acc += *(const __attribute__((address_space(1))) uint16_t*)(base + off);
      } else if (K == 5) {
        *(__attribute__((address_space(1))) uint16_t*)(base + off) = (uint16_t)j;
      } else if (K == 6) {
        if ((lane & 3) == 0) acc += *(const __attribute__((address_space(1))) uint32_t*)(base + off);
      } else {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        acc += *(const __attribute__((address_space(1))) u32u*)(base + off);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[wave] = t1 - t0;
  if (acc == 0xdeadbeef) out[1 << 20] = acc;
}

template <int K> void run(const char* name, uint8_t* buf, unsigned long long* d, int waves_per_cu, uint32_t region)
{
  const int iters = 4000;
  const int blocks = 256 * waves_per_cu / 4;
  k_rate<K><<<blocks, 256>>>(buf, d, 100, region);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k_rate<K><<<blocks, 256>>>(buf, d, iters, region);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const double n_inst_per_cu = (double)waves_per_cu * iters * 8;
  printf("%-44s %2d waves/CU: %8.3f ms  %7.1f ns per wave-instr per CU (= %6.1f cyc @2.4GHz); wave0 ticks/instr %.1f\n",
         name, waves_per_cu, ms, ms * 1e6 / n_inst_per_cu, ms * 1e6 / n_inst_per_cu * 2.4, (double)h / (iters * 8));
}

int main()
{
  setvbuf(stdout, NULL, _IONBF, 0);
  const size_t bytes = (size_t)256 * 32 * 65536 + 4096;
  uint8_t* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
  unsigned long long* d; hipMalloc(&d, ((1 << 20) + 16) * 8);
  for (int w : {5, 8, 16, 32}) {
    run<0>("coalesced aligned dword", buf, d, w, 65536);
    run<1>("byte-stride unaligned dword (window words)", buf, d, w, 65536);
    run<2>("dword gather, random in a 64 KiB region", buf, d, w, 65536);
    run<7>("dword gather, 1 KiB lane stride, 64 KiB", buf, d, w, 65536);
    run<3>("dword gather, 64 lines of an 8 KiB region", buf, d, w, 8192);
    run<6>("dword gather, random, 16 active lanes", buf, d, w, 65536);
    run<4>("ushort gather, random in 32 KiB", buf, d, w, 32768);
    run<5>("short scatter, random in 32 KiB", buf, d, w, 32768);
  }
  return 0;
}
