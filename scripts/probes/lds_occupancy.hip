// Probe: how much LDS can one workgroup get on gfx950, and how many workgroups of a given
// LDS size are co-resident on one CU?  (scratch experiment for DESIGN.md section 3.5)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_resident(unsigned* counter, unsigned* max_seen, unsigned* cu_ids, int spin)
{
  extern __shared__ unsigned char smem[];
  smem[threadIdx.x] = 1;
  if (threadIdx.x == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // CU id bits [11:8], SE id [15:13] (gfx9)
    unsigned key = ((xcc & 0xF) << 8) | (((hwid >> 13) & 7) << 4) | ((hwid >> 8) & 0xF);
    cu_ids[blockIdx.x] = key;
    unsigned now = atomicAdd(&counter[key], 1u) + 1;
    atomicMax(&max_seen[key], now);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
    atomicSub(&counter[key], 1u);
  }
}

int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu optin %zu perMP %zu CUs %d\n", p.sharedMemPerBlock, p.sharedMemPerBlockOptin,
         p.maxSharedMemoryPerMultiProcessor, p.multiProcessorCount);
  unsigned *counter, *max_seen, *cu;
  const int grid = 4096;
  hipMalloc(&counter, 4096 * 4); hipMalloc(&max_seen, 4096 * 4); hipMalloc(&cu, grid * 4);
  struct Cfg { int threads; int lds; };
  Cfg cfgs[] = {{64, 32768}, {64, 32000}, {64, 32256}, {64, 31744}, {320, 163840}, {320, 160000}, {256, 131072}, {128, 65536}, {128, 66560},
                {64, 36864}, {64, 40960}, {64, 33280}};
  for (auto c : cfgs) {
    hipMemset(counter, 0, 4096 * 4); hipMemset(max_seen, 0, 4096 * 4);
    hipError_t e = hipFuncSetAttribute((const void*)k_resident, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds);
    k_resident<<<grid, c.threads, c.lds>>>(counter, max_seen, cu, 200000);
    hipError_t e2 = hipGetLastError();
    hipError_t e3 = hipDeviceSynchronize();
    std::vector<unsigned> m(4096);
    hipMemcpy(m.data(), max_seen, 4096 * 4, hipMemcpyDeviceToHost);
    unsigned mx = 0, ncu = 0; unsigned long long sum = 0;
    for (auto v : m) { if (v) { ++ncu; sum += v; if (v > mx) mx = v; } }
    printf("threads %3d lds %6d: attr=%d launch=%d sync=%d | CUs seen %u, max resident WGs/CU %u, mean %.2f\n", c.threads, c.lds,
           (int)e, (int)e2, (int)e3, ncu, mx, ncu ? (double)sum / ncu : 0.0);
  }
  return 0;
}
