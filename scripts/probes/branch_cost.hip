// Probe: what does a taken branch cost a lone wave on a SIMD (gfx950)?
// Each loop iteration runs 8 dependent v_add with K taken s_branch between them.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__global__ void k_branches(unsigned long long* out, int iters)
{
  unsigned v = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (K == 0)
      asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n"
                   "v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n" : "+v"(v));
    if (K == 4)
      asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n s_branch 1f\n 1:\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n s_branch 2f\n 2:\n"
                   "v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n s_branch 3f\n 3:\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n s_branch 4f\n 4:\n" : "+v"(v));
    if (K == 8)
      asm volatile("v_add_u32 %0, %0, 1\n s_branch 1f\n 1:\n v_add_u32 %0, %0, 1\n s_branch 2f\n 2:\n v_add_u32 %0, %0, 1\n s_branch 3f\n 3:\n v_add_u32 %0, %0, 1\n s_branch 4f\n 4:\n"
                   "v_add_u32 %0, %0, 1\n s_branch 5f\n 5:\n v_add_u32 %0, %0, 1\n s_branch 6f\n 6:\n v_add_u32 %0, %0, 1\n s_branch 7f\n 7:\n v_add_u32 %0, %0, 1\n s_branch 8f\n 8:\n" : "+v"(v));
    if (K == 108) // 8 far jumps: over 64 bytes of s_nop each
      asm volatile(
          "v_add_u32 %0, %0, 1\n s_branch 1f\n .fill 16, 4, 0xbf800000\n 1:\n v_add_u32 %0, %0, 1\n s_branch 2f\n .fill 16, 4, 0xbf800000\n 2:\n"
          "v_add_u32 %0, %0, 1\n s_branch 3f\n .fill 16, 4, 0xbf800000\n 3:\n v_add_u32 %0, %0, 1\n s_branch 4f\n .fill 16, 4, 0xbf800000\n 4:\n"
          "v_add_u32 %0, %0, 1\n s_branch 5f\n .fill 16, 4, 0xbf800000\n 5:\n v_add_u32 %0, %0, 1\n s_branch 6f\n .fill 16, 4, 0xbf800000\n 6:\n"
          "v_add_u32 %0, %0, 1\n s_branch 7f\n .fill 16, 4, 0xbf800000\n 7:\n v_add_u32 %0, %0, 1\n s_branch 8f\n .fill 16, 4, 0xbf800000\n 8:\n" : "+v"(v));
    if (K == 208) // 8 not-taken conditional branches
      asm volatile("s_cmp_eq_u32 0, 1\n"
          "v_add_u32 %0, %0, 1\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, 1\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, 1\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, 1\n s_cbranch_scc1 1f\n"
          "v_add_u32 %0, %0, 1\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, 1\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, 1\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, 1\n s_cbranch_scc1 1f\n 1:\n" : "+v"(v) :: "scc");
    if (K == 308) // 8 exec save/restore pairs (predication instead of branching)
      asm volatile(
          "v_add_u32 %0, %0, 1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n v_add_u32 %0, %0, 1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n"
          "v_add_u32 %0, %0, 1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n v_add_u32 %0, %0, 1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n"
          "v_add_u32 %0, %0, 1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n v_add_u32 %0, %0, 1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n"
          "v_add_u32 %0, %0, 1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n v_add_u32 %0, %0, 1\n s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n"
          : "+v"(v) :: "s20", "s21", "vcc");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (v == 0xdeadbeef) out[1000] = v;
}

template <int K> void run(const char* name, int threads)
{
  unsigned long long* d; hipMalloc(&d, 8192 * 8);
  const int iters = 100000;
  k_branches<K><<<1, threads>>>(d, iters); hipDeviceSynchronize();
  k_branches<K><<<1, threads>>>(d, iters); hipDeviceSynchronize();
  unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-44s threads %4d: %7.2f ticks/iteration\n", name, threads, (double)h / iters);
  hipFree(d);
}

int main()
{
  for (int threads : {64, 256, 512}) {
    run<0>("8 v_add", threads);
    run<4>("8 v_add + 4 taken s_branch (next instr)", threads);
    run<8>("8 v_add + 8 taken s_branch (next instr)", threads);
    run<108>("8 v_add + 8 taken s_branch (+64 B)", threads);
    run<208>("8 v_add + 8 untaken s_cbranch", threads);
    run<308>("8 v_add + 8 saveexec/restore pairs", threads);
  }
  return 0;
}
