// Probe: instruction throughput of a FULL CU (32 waves) on gfx950: scalar vs vector vs mixed.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__global__ __launch_bounds__(1024) void k_rate(unsigned long long* out, int iters)
{
  unsigned v = threadIdx.x, w = 1, a = 2, x = 3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (K == 0) asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                             "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n" : "+v"(v), "+v"(w), "+v"(a), "+v"(x));
    if (K == 1) asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                             "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n" ::: "s20", "s21", "s22", "s23", "scc");
    if (K == 2) asm volatile("v_add_u32 %0, %0, 1\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, 1\n s_add_u32 s21, s21, 1\n"
                             "v_add_u32 %2, %2, 1\n s_add_u32 s22, s22, 1\n v_add_u32 %3, %3, 1\n s_add_u32 s23, s23, 1\n" : "+v"(v), "+v"(w), "+v"(a), "+v"(x) :: "s20", "s21", "s22", "s23", "scc");
    if (K == 3) asm volatile("v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 6\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 8\n"
                             "v_readlane_b32 s20, %0, 9\n v_readlane_b32 s21, %1, 1\n v_readlane_b32 s22, %2, 2\n v_readlane_b32 s23, %3, 3\n" :: "v"(v), "v"(w), "v"(a), "v"(x) : "s20", "s21", "s22", "s23");
    if (K == 4) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0\n"
                             "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0\n" :: "v"(v), "v"(w), "v"(a), "v"(x) : "vcc");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (v + w + a + x == 0xdeadbeef) out[100000] = v;
}

template <int K> void run(const char* name, unsigned long long* d, int blocks, int threads)
{
  const int iters = 2000000;
  k_rate<K><<<blocks, threads>>>(d, iters); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k_rate<K><<<blocks, threads>>>(d, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("[%.3f ms, %.0f ticks -> %.3f ticks/ns] ", ms, (double)h, (double)h / (ms * 1e6));
  const double per_iter = (double)h / iters;
  const int waves_per_cu = threads / 64 * (blocks > 256 ? blocks / 256 : 1);
  printf("%-28s %2d waves/CU: %7.1f ticks per 8 instr per wave -> %5.2f instr/tick/CU\n", name, waves_per_cu, per_iter,
         8.0 * waves_per_cu / per_iter);
}

int main()
{
  setvbuf(stdout, NULL, _IONBF, 0);
  unsigned long long* d; hipMalloc(&d, 200000 * 8);
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int blocks = cfg == 2 ? 512 : 256, threads = cfg == 0 ? 256 : 1024;
    run<0>("v_add independent", d, blocks, threads);
    run<1>("s_add independent", d, blocks, threads);
    run<2>("v_add / s_add alternating", d, blocks, threads);
    run<3>("v_readlane", d, blocks, threads);
    run<4>("v_cmp", d, blocks, threads);
  }
  return 0;
}
