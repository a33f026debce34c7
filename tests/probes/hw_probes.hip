// hw_probes.hip -- tiny kernels that measure hardware-defined behaviour the
// bit-exact LZ4 path depends on.  Test
// infrastructure: built into tests/probes/libhwprobes.so by
// __graft_entry__.build(), loaded only by tests/test_hw_probes.py.
//
//  (1) several lanes of one wave execute ONE global_store_short to the same
//      address: which lane's value survives?  (the reference's
//      insertHashTableWarp does this, src/LZ4Kernels.hiph:734-737)
//  (2) the same for ONE ds_write_b16 (this library keeps the table in LDS);
//  (3) the same for ONE ds_write_b8 (the tag table next to it must keep the
//      tag of the very lane whose position survives in (2)).
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void k_global_store_short(uint16_t* out, const int* slot, unsigned long long mask)
{
  const int t = threadIdx.x;
  const int s = slot[t];
  if ((mask >> t) & 1ull)
    out[s] = (uint16_t)(1000 + t);
}

__global__ void k_lds_store_short(uint16_t* out, const int* slot, unsigned long long mask, int nslots)
{
  extern __shared__ uint16_t tab[];
  const int t = threadIdx.x;
  for (int i = t; i < nslots; i += 64)
    tab[i] = 0xFFFF;
  __syncthreads();
  const int s = slot[t];
  if ((mask >> t) & 1ull)
    tab[s] = (uint16_t)(1000 + t);
  __syncthreads();
  for (int i = t; i < nslots; i += 64)
    out[i] = tab[i];
}

__global__ void k_lds_store_byte(uint16_t* out, const int* slot, unsigned long long mask, int nslots)
{
  extern __shared__ uint8_t tabb[];
  const int t = threadIdx.x;
  for (int i = t; i < nslots; i += 64)
    tabb[i] = 0xFF;
  __syncthreads();
  const int s = slot[t];
  if ((mask >> t) & 1ull)
    tabb[s] = (uint8_t)(100 + t);
  __syncthreads();
  for (int i = t; i < nslots; i += 64)
    out[i] = tabb[i] == 0xFF ? (uint16_t)0xFFFF : (uint16_t)(tabb[i] + 900); // same scale as the short probes
}

extern "C" int probe_lds_store_byte(uint16_t* out, const int* slot, unsigned long long mask, int nslots, hipStream_t st)
{
  k_lds_store_byte<<<1, 64, nslots, st>>>(out, slot, mask, nslots);
  return (int)hipGetLastError();
}

// (4) LDS stores to an address beyond everything a workgroup can own (the
//     encoder's walk sends the stores of lanes that must not store there when
//     its tables fill the whole 160 KiB): dropped, nothing else changes; a read
//     from there returns 0.
__global__ void k_lds_out_of_range(uint32_t* out, int nwords, uint32_t far_addr)
{
  extern __shared__ uint32_t words[];
  const int t = threadIdx.x;
  for (int i = t; i < nwords; i += 64)
    words[i] = 0xA5000000u + (uint32_t)i;
  __syncthreads();
  const uint32_t a = far_addr + 2u * (uint32_t)t;
  uint32_t back = 0x12345678u;
  asm volatile("ds_write_b16 %1, %2\n\tds_write_b8 %1, %2 offset:1\n\tds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)"
               : "=v"(back) : "v"(a), "v"(0xBEEFu + (uint32_t)t) : "memory");
  __syncthreads();
  for (int i = t; i < nwords; i += 64)
    out[i] = words[i];
  out[nwords + t] = back;
}

extern "C" int probe_lds_out_of_range(uint32_t* out, int nwords, uint32_t far_addr, hipStream_t st)
{
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_out_of_range),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess)
    return (int)e;
  k_lds_out_of_range<<<1, 64, nwords * 4, st>>>(out, nwords, far_addr);
  return (int)hipGetLastError();
}

extern "C" int probe_global_store_short(uint16_t* out, const int* slot, unsigned long long mask, hipStream_t st)
{
  k_global_store_short<<<1, 64, 0, st>>>(out, slot, mask);
  return (int)hipGetLastError();
}

extern "C" int probe_lds_store_short(uint16_t* out, const int* slot, unsigned long long mask, int nslots, hipStream_t st)
{
  k_lds_store_short<<<1, 64, nslots * 2, st>>>(out, slot, mask, nslots);
  return (int)hipGetLastError();
}
