// Which XCD / SE / CU the workgroups of a launch land on, by position in the grid (gfx950):
//   hipcc --offload-arch=gfx950 -O2 where_blocks_go.hip -o bin/where_blocks_go && bin/where_blocks_go [threads=256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void where(uint32_t* out)
{
  if (threadIdx.x == 0) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}
int main(int argc, char** argv)
{
  const int threads = argc > 1 ? atoi(argv[1]) : 256, blocks = 96;
  uint32_t* d;
  hipMalloc(&d, blocks * 8);
  where<<<blocks, threads>>>(d);
  std::vector<uint32_t> h(2 * blocks);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  printf("threads per workgroup %d: position -> xcc (HW_ID: se, sh, cu, simd, wave)\n", threads);
  for (int i = 0; i < blocks; ++i) {
    const uint32_t hw = h[2 * i + 1];
    printf("%3d -> xcc %u  se %u sh %u cu %2u simd %u wave %u\n", i, h[2 * i] & 15u, (hw >> 13) & 7u, (hw >> 12) & 1u, (hw >> 8) & 15u,
           (hw >> 4) & 3u, hw & 15u);
  }
  return 0;
}
