// wave_utils.hpp -- wave64 building blocks shared by the gfx950 codec kernels.
//
// Everything here assumes a 64-lane wavefront (CDNA4) and is written for it
// directly: 64-bit ballots, v_readlane with a scalar lane index, unaligned
// dword / dwordx4 global accesses (legal on gfx950 in the HSA default
// unaligned-access mode) and 16-byte-per-lane copies.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace hcamd {

constexpr int kWave = 64;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Chunk buffers are reached through pointers loaded from device arrays, so
// the compiler cannot tell they are global memory and would emit flat_*
// accesses (counted on both vmcnt and lgkmcnt).  All data pointers are
// therefore moved to the global address space explicitly.
#define HC_GLOBAL __attribute__((address_space(1)))
typedef HC_GLOBAL uint8_t* gptr;
typedef const HC_GLOBAL uint8_t* cgptr;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;

__device__ __forceinline__ gptr to_global(uint8_t* p) { return (gptr)p; }
__device__ __forceinline__ cgptr to_global(const uint8_t* p) { return (cgptr)p; }

__device__ __forceinline__ uint32_t load_u32_any(cgptr p)
{
  return *reinterpret_cast<const HC_GLOBAL u32_unaligned*>(p);
}

__device__ __forceinline__ u32x4 load_u128_any(cgptr p)
{
  return *reinterpret_cast<const HC_GLOBAL u32x4_unaligned*>(p);
}

__device__ __forceinline__ int lane_id()
{
  return (int)(threadIdx.x & (kWave - 1));
}

// Tell the compiler a value is the same in every lane (it came from a load
// at a wave-uniform address, which lands in a VGPR): the value moves to an
// SGPR and everything computed from it -- loop bounds, branch conditions,
// addresses -- becomes scalar.  Without this, loops steered by loaded values
// are compiled as DIVERGENT control flow: every `if` turns into exec-mask
// save/restore sequences on the CU's single scalar ALU.
__device__ __forceinline__ uint32_t uniform(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t uniform(uint64_t v)
{
  return (uint64_t)uniform((uint32_t)v) | ((uint64_t)uniform((uint32_t)(v >> 32)) << 32);
}
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p)
{
  return reinterpret_cast<T*>(uniform((uint64_t)reinterpret_cast<uintptr_t>(p)));
}

__device__ __forceinline__ uint32_t read_lane(uint32_t v, int lane)
{
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// Compiler-only fence for LDS traffic that travels BETWEEN lanes of the wave
// (one lane stores, another lane loads the same address in the next
// instruction).  LDS operations of a wave execute in order, so no hardware
// wait is needed, but the compiler must neither forward this lane's own store
// to its load nor reorder the accesses.  (A `volatile` pointer would do that
// too, but it turns the ds_* accesses into flat_* ones with full waits.)
__device__ __forceinline__ void lds_lane_exchange_fence()
{
  asm volatile("" ::: "memory");
}

// 64-bit ballot straight from the condition (HIP's __ballot goes through a
// VGPR 0/1 value and a second compare).
__device__ __forceinline__ uint64_t wave_ballot(bool p)
{
  return __builtin_amdgcn_ballot_w64(p);
}

// lanes [0, n), n in [0, MAXN]; the n == 64 case is only compiled in when the
// caller can actually pass it
template <int MAXN>
__device__ __forceinline__ uint64_t lanes_below(int n)
{
  if (MAXN >= 64)
    return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
  return (1ull << n) - 1ull;
}

__device__ __forceinline__ uint64_t low_lanes_mask(int n) // lanes [0, n)
{
  return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
}

// ---- wave64 inclusive scans on the DPP crossbar (no LDS, no ds_bpermute) ---
// row_shr:1,2,4,8 scans each 16-lane row, row_bcast:15 / row_bcast:31 carry
// the row totals into the following rows (gfx9 DPP controls).  Lanes shifted
// in from outside a row read 0, the identity of both operations on unsigned
// values.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}

__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v)
{
  v += dpp_u32<0x111, 0xF>(v); // row_shr:1
  v += dpp_u32<0x112, 0xF>(v); // row_shr:2
  v += dpp_u32<0x114, 0xF>(v); // row_shr:4
  v += dpp_u32<0x118, 0xF>(v); // row_shr:8
  v += dpp_u32<0x142, 0xA>(v); // row_bcast:15 -> rows 1, 3
  v += dpp_u32<0x143, 0xC>(v); // row_bcast:31 -> rows 2, 3
  return v;
}

__device__ __forceinline__ uint32_t wave_scan_max_u32(uint32_t v)
{
  uint32_t u;
  u = dpp_u32<0x111, 0xF>(v); v = u > v ? u : v;
  u = dpp_u32<0x112, 0xF>(v); v = u > v ? u : v;
  u = dpp_u32<0x114, 0xF>(v); v = u > v ? u : v;
  u = dpp_u32<0x118, 0xF>(v); v = u > v ? u : v;
  u = dpp_u32<0x142, 0xA>(v); v = u > v ? u : v;
  u = dpp_u32<0x143, 0xC>(v); v = u > v ? u : v;
  return v;
}

__device__ __forceinline__ uint64_t wave_scan_add_u64(uint64_t v)
{
  // 64-bit sums: two 32-bit DPP moves per step
#define HC_STEP(CTRL, MASK)                                                      \
  {                                                                              \
    const uint64_t u = (uint64_t)dpp_u32<CTRL, MASK>((uint32_t)v)                \
                       | ((uint64_t)dpp_u32<CTRL, MASK>((uint32_t)(v >> 32)) << 32); \
    v += u;                                                                      \
  }
  HC_STEP(0x111, 0xF)
  HC_STEP(0x112, 0xF)
  HC_STEP(0x114, 0xF)
  HC_STEP(0x118, 0xF)
  HC_STEP(0x142, 0xA)
  HC_STEP(0x143, 0xC)
#undef HC_STEP
  return v;
}

// Copy n bytes dst <- src with all 64 lanes; any alignment on either side,
// regions must not overlap.  16-byte aligned stores, unaligned 16-byte loads.
__device__ __forceinline__ void wave_copy(
    gptr __restrict__ dst, cgptr __restrict__ src, uint32_t n, int lane)
{
  uint32_t head = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
  if (head > n)
    head = n;
  if ((uint32_t)lane < head)
    dst[lane] = src[lane];
  dst += head;
  src += head;
  n -= head;
  const uint32_t nvec = n >> 4;
  uint32_t i = (uint32_t)lane;
  // 4 x 16 B in flight per lane
  for (; i + 3 * kWave < nvec; i += 4 * kWave) {
    u32x4 a = load_u128_any(src + 16u * i);
    u32x4 b = load_u128_any(src + 16u * (i + kWave));
    u32x4 c = load_u128_any(src + 16u * (i + 2 * kWave));
    u32x4 d = load_u128_any(src + 16u * (i + 3 * kWave));
    *reinterpret_cast<HC_GLOBAL u32x4*>(dst + 16u * i) = a;
    *reinterpret_cast<HC_GLOBAL u32x4*>(dst + 16u * (i + kWave)) = b;
    *reinterpret_cast<HC_GLOBAL u32x4*>(dst + 16u * (i + 2 * kWave)) = c;
    *reinterpret_cast<HC_GLOBAL u32x4*>(dst + 16u * (i + 3 * kWave)) = d;
  }
  for (; i < nvec; i += kWave)
    *reinterpret_cast<HC_GLOBAL u32x4*>(dst + 16u * i) = load_u128_any(src + 16u * i);
  const uint32_t tail = n & 15u;
  if ((uint32_t)lane < tail)
    dst[(nvec << 4) + lane] = src[(nvec << 4) + lane];
}

// j mod m for j < 64 and any m >= 1 (m wave-uniform or not): quotient from a
// float reciprocal (exact to within one for these ranges), then one
// correction step either way.  About 7 instructions; the compiler's generic
// 32-bit modulo is about 25.
__device__ __forceinline__ uint32_t small_mod(uint32_t j, uint32_t m)
{
  const float r = __builtin_amdgcn_rcpf((float)m);
  const uint32_t q = (uint32_t)((float)j * r);
  int32_t rem = (int32_t)(j - q * m);
  rem = rem < 0 ? rem + (int32_t)m : rem;
  rem = rem >= (int32_t)m ? rem - (int32_t)m : rem;
  return (uint32_t)rem;
}

// A wave-uniform value moved into a vector register, where it and whatever is
// computed from it stay: the compiler computes anything it can prove uniform
// on the scalar unit, of which a CU has one for all its waves.
__device__ __forceinline__ uint32_t in_vector_register(uint32_t x)
{
  asm("" : "+v"(x));
  return x;
}

// ---------------------------------------------------------------------------
// 256 bytes of a byte stream in registers: lane t holds the dword at stream
// index base + 4t (any alignment).  A decoder reads its tags / tokens /
// offsets from it with v_readlane instead of taking a memory round trip for
// each of them.
// ---------------------------------------------------------------------------
struct StreamWindow
{
  uint32_t words = 0; // per lane
  // stream index of lane 0's dword (wave-uniform); initially out of reach of
  // any position below 2^31, i.e. the first ensure() fills the window
  uint32_t base = 0x80000000u;

  // Makes the window cover [pos, pos + reach + 8).  Needs pos < end, end >= 4
  // and reach <= 64: `reach` is how far past pos the caller will ask for bytes
  // before calling ensure() the next time.
  __device__ __forceinline__ void ensure(cgptr stream, uint32_t pos, uint32_t end, uint32_t reach, int lane)
  {
    if (pos - base > 256u - 8u - reach) {
      base = pos;
      // Lanes whose dword would reach past the end of the stream load the
      // last dword of the stream instead (end >= 4 here) and shift it into
      // place: bytes past the end read as 0.
      const uint32_t at = pos + 4u * (uint32_t)lane;
      const uint32_t over = at > end - 4u ? at - (end - 4u) : 0u; // bytes
      const uint32_t raw = load_u32_any(stream + (at - over));
      words = over >= 4u ? 0u : raw >> (8u * over);
      // wait here: left to the compiler the wait lands after the branch, where
      // it would also wait for the caller's last store every time
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    }
  }

  // 4 bytes at byte index idx (< 252) of the window; idx is wave-uniform.
  __device__ __forceinline__ uint32_t bytes_at(uint32_t idx) const
  {
    const uint32_t q = idx >> 2;
    const uint32_t lo = read_lane(words, (int)q), hi = read_lane(words, (int)q + 1);
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> ((idx & 3u) * 8u));
  }
};

// The same window for a decoder that keeps its stream position in a vector
// register (wave-uniform all the same, see in_vector_register): the window is
// read with ds_bpermute instead of v_readlane, so that no step of the parse
// runs on the scalar unit.
struct StreamWindowV
{
  uint32_t words = 0;          // per lane
  uint32_t base = 0x80000000u; // stream index of lane 0's dword, same in all lanes

  // Makes the window cover [pos, pos + reach + 8); needs pos < end, end >= 4, reach <= 64.
  __device__ __forceinline__ void ensure(cgptr stream, uint32_t pos, uint32_t end, uint32_t reach, int lane)
  {
    if (wave_ballot(pos - base > 256u - 8u - reach) != 0) {
      base = pos;
      const uint32_t at = pos + 4u * (uint32_t)lane;
      const uint32_t over = at > end - 4u ? at - (end - 4u) : 0u; // bytes
      const uint32_t raw = load_u32_any(stream + (at - over));
      words = over >= 4u ? 0u : raw >> (8u * over);
    }
  }

  // 4 bytes at byte index idx (< 252) of the window
  __device__ __forceinline__ uint32_t bytes_at(uint32_t idx) const
  {
    const uint32_t q4 = idx & ~3u;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)q4, (int)words);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(q4 + 4u), (int)words);
    return __builtin_amdgcn_alignbyte(hi, lo, idx & 3u);
  }
};

} // namespace hcamd
