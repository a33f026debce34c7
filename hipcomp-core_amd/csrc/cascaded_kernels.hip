// cascaded_kernels.hip -- gfx950 kernels of the batched Cascaded codec
// (RLE -> Delta -> BitPack fused over 4096-byte sub-chunks, all in LDS).
//
// Wire format and layer semantics: reference src/CascadedKernels.hiph:101-1435
// (restated in oracle/cascaded_oracle.c).  Mechanism:
//
//   reference                                 here
//   ----------------------------------------  -------------------------------
//   one 128-thread block per partition,       one WAVE per partition and workgroup:
//   6+ __syncthreads per layer, hipCUB        no barrier at all
//   BlockScan / BlockReduce
//   RLE: count pass, scan, scatter pass,      ONE pass in a register-blocked layout
//   adjacent-difference pass, a thread        (round 4): a lane holds 16 CONSECUTIVE
//   per element                               elements of the sub-chunk, finds its run
//                                             ends with 16 compares, one wave scan ranks
//                                             them, every run end stores its value and
//                                             its length itself -- 1/5 of the
//                                             instructions of the element-per-lane form
//                                             of rounds 1-3 (one ballot, one rank and
//                                             one store group per 64 elements)
//   bit-packed arrays built in LDS, then      packed words go straight from the LDS
//   copied out word by word                   element array to HBM
//   decompress: 4 launches, type taken from   4 launches (LDS sized per width), type
//   partition 0, RLE expand = 1 thread/run    taken from EACH partition; RLE expand =
//   (a 4096-long run = 4096 serial stores)    run markers + a running maximum and delta
//                                             = a prefix sum, both in the blocked layout:
//                                             16 steps inside the lane, one wave scan
//                                             across the lanes
//
// Bytes the reference leaves undefined (stale LDS / unwritten gaps,
// SURVEY.md App. C.4) are written as 0 here, so the output is deterministic.

#include "cascaded_launch.hpp"
#include "lz4_launch.hpp" // num_cus_of_current_device
#include "placement.hiph"
#include "wave_utils.hpp"

#include <atomic>
#include <type_traits>

#define HC_LDS __attribute__((address_space(3)))

namespace hcamd {

namespace {

// Sub-chunks are the reference's 4096 bytes (CascadedKernels.hiph:88, hard-wired there; its
// opts.chunk_size is "not currently used", cascaded.h:93-100, and is ignored here as well).
// (Rounds 2-3 honoured 8192 / 16384 as an opt-in extension of the format: ratio +1 %, -26 ... -56 %
// throughput, eight more decompress launches -- measured useless and removed in round 4.)
constexpr uint32_t kPartMeta = 8;
constexpr uint32_t CB = 4096;
#ifndef HC_CASC_STOP_AFTER
#define HC_CASC_STOP_AFTER 0 // (measurement builds only: see the encoder's fast path)
#endif
#ifndef HC_CASC_WAVES
#define HC_CASC_WAVES 1 // 6 KiB of LDS per wave: separate blocks pack 25 per CU
#endif
constexpr int kWavesPerBlock = HC_CASC_WAVES;

template <int S> struct UIntOf;
template <> struct UIntOf<1> { typedef uint8_t type; typedef int8_t stype; };
template <> struct UIntOf<2> { typedef uint16_t type; typedef int16_t stype; };
template <> struct UIntOf<4> { typedef uint32_t type; typedef int32_t stype; };
template <> struct UIntOf<8> { typedef uint64_t type; typedef int64_t stype; };

__device__ __forceinline__ uint32_t ru(uint32_t a, uint32_t b) { return (a + b - 1) / b * b; }

// An array of `nbytes` bytes at byte offset pos + rel of a stream of end_words words: 4-byte aligned and inside
// (reference block_read :712-713) -- in arithmetic that cannot wrap: nbytes comes straight from the stream, and
// 0xFFFFF000 must not pass for a short array (the 32-bit sum did).
__device__ __forceinline__ bool array_inside(uint32_t pos, uint32_t rel, uint32_t nbytes, uint32_t end_words)
{
  const uint64_t off = (uint64_t)pos + rel;
  return !(off & 3u) && off / 4u + ((uint64_t)nbytes + 3u) / 4u <= end_words;
}
// ... and within the first stage_bytes bytes of its sub-chunk (the staged image)
__device__ __forceinline__ bool array_staged(uint32_t rel, uint32_t nbytes, uint32_t stage_bytes)
{
  return (uint64_t)rel + (((uint64_t)nbytes + 3u) & ~3ull) <= stage_bytes;
}

// ---------------------------------------------------------------------------
// The register-blocked layout.  A ROUND is 64 x E consecutive elements of an LDS array, lane t
// holding elements [t E, (t + 1) E) of it in registers: E = 16 (8 for 8-byte elements), i.e. a
// round is 1024 elements (512) and a 4096-byte sub-chunk is 4 / 2 / 1 / 1 rounds of 1- / 2- /
// 4- / 8-byte elements.  What an element-per-lane layout does with a ballot or a wave scan per
// 64 elements is E steps inside the lane here plus ONE wave scan per round.
// ---------------------------------------------------------------------------
template <int S>
struct Blocked
{
  static constexpr int E = S == 8 ? 8 : 16;    // elements per lane and round
  static constexpr int W = E * S / 4;          // 32-bit words per lane and round: 4, 8, 16, 16
  static constexpr uint32_t ROUND = kWave * E; // elements per round
};

// encoder: behind the element buffer lie the E + 1 sentinels of wave_rle
__host__ __device__ constexpr uint32_t enc_buf_bytes() { return CB + 80; }
// decoder: one element more (the head of a delta layer)
__host__ __device__ constexpr uint32_t dec_buf_bytes() { return CB + 16; }

template <int S>
__host__ __device__ constexpr uint32_t wave_lds_bytes()
{
  // encoder: one element buffer (every layer works in place) + run-length
  // array + 64-byte metadata image; 4-byte elements (the fast path, see rle16): the two padded arrays
  return S == 4 ? (1024 + 32) * 4 + (1024 * 2 + 32 * 4) : enc_buf_bytes() + (CB / S) * 2 + 64;
}

// decoder: the head of the compressed sub-chunk (metadata + arrays) is staged
// in LDS, kStageWords 32-bit words, kStagePerLane per lane, loaded one
// sub-chunk ahead into registers.  Arrays that do not fit are read from HBM
// directly.  1 KiB holds the whole sub-chunk of a column that compresses 4x or
// better and lets a CU hold 14 waves of the 4-byte launch (2 KiB: 12;
// measured 960 vs 843 GB/s at ratio 5.3, 770 vs 840 GB/s at ratio 2.2).
constexpr uint32_t kStagePerLane = 4;
constexpr uint32_t kStageWords = kStagePerLane * kWave;
template <int S>
__host__ __device__ constexpr uint32_t dec_lds_bytes()
{
  // two element buffers + run markers + staged sub-chunk; 4-byte elements (the fast path): padded buffers
  return S == 4 ? 1024 + (1024 + 32) * 4 + 2048 + 16 : 2 * dec_buf_bytes() + (CB / S) * 2 + kStageWords * 4;
}

// E elements p[0 .. E) of an LDS array (p 16-byte aligned) into registers
template <typename UT>
__device__ __forceinline__ void load_block(const UT* p, UT (&v)[Blocked<sizeof(UT)>::E])
{
  constexpr int S = sizeof(UT), E = Blocked<S>::E, W = Blocked<S>::W;
  uint32_t w[W];
  const u32x4* q = reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) {
    const u32x4 t = q[i];
    w[4 * i] = t.x;
    w[4 * i + 1] = t.y;
    w[4 * i + 2] = t.z;
    w[4 * i + 3] = t.w;
  }
#pragma unroll
  for (int k = 0; k < E; ++k) {
    if (S == 8)
      v[k] = (UT)((uint64_t)w[(2 * k) % W] | ((uint64_t)w[(2 * k + 1) % W] << 32));
    else if (S == 4)
      v[k] = (UT)w[k % W];
    else if (S == 2)
      v[k] = (UT)(w[(k / 2) % W] >> (16 * (k & 1)));
    else
      v[k] = (UT)(w[(k / 4) % W] >> (8 * (k & 3)));
  }
}

// ... and back
template <typename UT>
__device__ __forceinline__ void store_block(UT* p, const UT (&v)[Blocked<sizeof(UT)>::E])
{
  constexpr int S = sizeof(UT), E = Blocked<S>::E, W = Blocked<S>::W;
  uint32_t w[W];
#pragma unroll
  for (int i = 0; i < W; ++i) {
    if (S == 8)
      w[i] = (uint32_t)((uint64_t)v[(i / 2) % E] >> (32 * (i & 1)));
    else if (S == 4)
      w[i] = (uint32_t)v[i % E];
    else if (S == 2)
      w[i] = (uint32_t)v[(2 * i) % E] | ((uint32_t)v[(2 * i + 1) % E] << 16);
    else
      w[i] = (uint32_t)v[(4 * i) % E] | ((uint32_t)v[(4 * i + 1) % E] << 8) | ((uint32_t)v[(4 * i + 2) % E] << 16)
             | ((uint32_t)v[(4 * i + 3) % E] << 24);
  }
  u32x4* q = reinterpret_cast<u32x4*>(p);
#pragma unroll
  for (int i = 0; i < W / 4; ++i) {
    u32x4 t = {w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
    q[i] = t;
  }
}

// the value of the lane below (lane 0: `first`), by the DPP crossbar
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v, uint32_t first, int lane)
{
  const uint32_t u = dpp_u32<0x138, 0xF>(v); // wave_shr:1
  return lane == 0 ? first : u;
}
__device__ __forceinline__ uint64_t from_lane_below(uint64_t v, uint64_t first, int lane)
{
  const uint64_t u = (uint64_t)dpp_u32<0x138, 0xF>((uint32_t)v) | ((uint64_t)dpp_u32<0x138, 0xF>((uint32_t)(v >> 32)) << 32);
  return lane == 0 ? first : u;
}

// ---- wave reductions (64 lanes) -------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_min(T v)
{
  for (int o = 32; o > 0; o >>= 1) {
    T u = __shfl_xor(v, o);
    v = u < v ? u : v;
  }
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
  for (int o = 32; o > 0; o >>= 1) {
    T u = __shfl_xor(v, o);
    v = u > v ? u : v;
  }
  return v;
}

// ---- RLE of n elements: values, run lengths, number of runs ----------------
// (reference block_rle_compress :124-241).  In place: the values of the runs are
// compacted to x[0 .. m), their lengths go to cnts[0 .. m).  Register-blocked: a
// lane compares its E elements with their right neighbours (the last one with the
// first element of the lane above: one more LDS read), a wave scan of the lanes'
// counts ranks the run ends, a running maximum of "where the last run ended below
// me" gives the first one its length, and every run end then stores value and
// length at its rank -- a round's loads all come before its stores (LDS operations
// of a wave execute in order) and a round writes at or below what it has read.
// E + 1 sentinels unlike x[n - 1] behind the last element make the last run end
// like any other and keep the lanes beyond n quiet without a compare per element.
template <typename UT>
__device__ __forceinline__ uint32_t wave_rle(UT* x, uint32_t n, uint16_t* cnts, int lane)
{
  constexpr int S = sizeof(UT), E = Blocked<S>::E;
  if (n == 0)
    return 0;
  {
    const UT last = x[n - 1];
    lds_lane_exchange_fence();
    if (lane <= E)
      x[n + (uint32_t)lane] = (UT)~last;
    lds_lane_exchange_fence();
  }
  uint32_t m = 0, prev_end = 0;
  for (uint32_t base = 0; base < n; base += Blocked<S>::ROUND) {
    const uint32_t pos0 = base + (uint32_t)lane * E;
    const bool active = pos0 < n;
    UT v[E];
    UT nx = 0;
    uint32_t cnt = 0, last_end = 0;
    bool f[E];
    if (active) {
      load_block(x + pos0, v);
      nx = x[pos0 + E];
    }
#pragma unroll
    for (int k = 0; k < E; ++k) {
      f[k] = active && v[k] != (k + 1 < E ? v[(k + 1) % E] : nx);
      cnt += f[k] ? 1u : 0u;
      last_end = f[k] ? pos0 + (uint32_t)k + 1u : last_end;
    }
    const uint32_t incl = wave_scan_add_u32(cnt);
    const uint32_t ends = wave_scan_max_u32(last_end); // (inclusive; positions grow with the lane)
    uint32_t rank = m + incl - cnt;
    // where the run that ends first in this lane began: behind the last end of the lanes below
    uint32_t start = max(prev_end, dpp_u32<0x138, 0xF>(ends));
    lds_lane_exchange_fence();
#pragma unroll
    for (int k = 0; k < E; ++k) {
      if (f[k]) {
        const uint32_t end = pos0 + (uint32_t)k + 1u;
        x[rank] = v[k];
        cnts[rank] = (uint16_t)(end - start);
        start = end;
        ++rank;
      }
    }
    lds_lane_exchange_fence();
    m += read_lane(incl, 63);
    prev_end = max(prev_end, read_lane(ends, 63));
  }
  return m;
}

// ---- delta of n >= 1 elements in place: x[i] = x[i + 1] - x[i], i < n - 1 ----
// (reference block_delta_compress :317-328; wrap-around arithmetic).  Blocked as above;
// what lands at and behind n - 1 is not used.
template <typename UT>
__device__ __forceinline__ void wave_delta(UT* x, uint32_t n, int lane)
{
  constexpr int S = sizeof(UT), E = Blocked<S>::E;
  for (uint32_t base = 0; base + 1 < n; base += Blocked<S>::ROUND) {
    const uint32_t pos0 = base + (uint32_t)lane * E;
    if (pos0 + 1 < n) {
      UT v[E], d[E];
      load_block(x + pos0, v);
      const UT nx = x[pos0 + E];
      lds_lane_exchange_fence();
#pragma unroll
      for (int k = 0; k < E; ++k)
        d[k] = (UT)((k + 1 < E ? v[(k + 1) % E] : nx) - v[k]);
      store_block(x + pos0, d);
    }
    lds_lane_exchange_fence();
  }
}

// ---- one array to HBM (reference block_write :646-680 / block_bitpack) ----
// Returns the byte length the format records; words are written at
// out + off (4-byte aligned).  `limit` is the partition's output limit in
// bytes; returns 0xFFFFFFFF when the array does not fit (reference :668-671).
template <typename ET>
__device__ __forceinline__ uint32_t wave_write_array(
    gptr out, uint32_t off, uint32_t limit, const ET* v, uint32_t n, int bp, int lane)
{
  typedef typename std::make_signed<ET>::type SET;
  constexpr uint32_t ES = sizeof(ET);
  HC_GLOBAL uint32_t* dst = reinterpret_cast<HC_GLOBAL uint32_t*>(out + off);
  if (!bp) {
    const uint32_t ob = n * ES;
    const uint32_t words = (ob + 3) / 4;
    if (off + words * 4 > limit)
      return 0xFFFFFFFFu;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(v);
    for (uint32_t w = (uint32_t)lane; w < words; w += kWave) {
      uint32_t x = src[w];
      const uint32_t valid_bytes = ob - 4 * w; // < 4 only in the last word
      if (valid_bytes < 4)
        x &= (1u << (8 * valid_bytes)) - 1u;
      dst[w] = x;
    }
    return ob;
  }
  // frame of reference = minimum under the SIGNED interpretation, bit width
  // from max - min (reference get_for_bitwidth :394-471)
  SET mn = 0, mx = 0;
  if (n > 0) {
    mn = (SET)v[0];
    mx = mn;
    for (uint32_t i = (uint32_t)lane; i < n; i += kWave) {
      const SET x = (SET)v[i];
      mn = x < mn ? x : mn;
      mx = x > mx ? x : mx;
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (ES > 4) {
      mn = (SET)uniform((uint64_t)mn);
      mx = (SET)uniform((uint64_t)mx);
    } else {
      mn = (SET)uniform((uint32_t)(int32_t)mn);
      mx = (SET)uniform((uint32_t)(int32_t)mx);
    }
  }
  uint32_t bw;
  if (ES > 4) {
    const uint64_t range = (uint64_t)mx - (uint64_t)mn;
    bw = range ? 64u - (uint32_t)__builtin_clzll(range) : 0u;
  } else {
    const uint32_t range = (uint32_t)(int32_t)mx - (uint32_t)(int32_t)mn;
    bw = range ? 32u - (uint32_t)__builtin_clz(range) : 0u;
  }
  const uint32_t words = (n * bw + 31) / 32;
  constexpr uint32_t HDR = ES > 4 ? 16 : 8; // roundUp(ES + 4, max(4, ES))
  const uint32_t ob = HDR + 4 * words;
  if (off + ob > limit)
    return 0xFFFFFFFFu;
  const ET fr = (ET)mn;
  if (lane == 0) {
    // [FOR][pad to 4][bitwidth<<16 | n][pad to ES]; pads written as 0
    if (ES > 4) {
      dst[0] = (uint32_t)((uint64_t)fr);
      dst[1] = (uint32_t)((uint64_t)fr >> 32);
      dst[2] = (bw << 16) | n;
      dst[3] = 0;
    } else {
      dst[0] = (uint32_t)fr; // zero-extended: pad bytes are 0
      dst[1] = (bw << 16) | n;
    }
  }
  HC_GLOBAL uint32_t* data = dst + HDR / 4;
  if (ES <= 4) {
    // LSB-first packing of (x - FOR) in bw bits (reference :523-552), a lane per
    // output word: the first element of the word from a float reciprocal (exact
    // to within one for these sizes, then corrected), each element moved into
    // place by one 64-bit shift
    const float rbw = __builtin_amdgcn_rcpf((float)bw); // (words > 0: bw >= 1)
    for (uint32_t w = (uint32_t)lane; w < words; w += kWave) {
      const uint32_t b0 = w * 32;
      uint32_t i = (uint32_t)((float)b0 * rbw);
      int32_t sh = (int32_t)(i * bw) - (int32_t)b0; // bit of the word at which element i starts: (-bw, 0] for the first one
      if (sh > 0) {
        --i;
        sh -= (int32_t)bw;
      }
      if (sh <= -(int32_t)bw) {
        ++i;
        sh += (int32_t)bw;
      }
      uint32_t acc = 0;
      for (; sh < 32 && i < n; ++i, sh += (int32_t)bw) {
        const uint32_t x = (uint32_t)(ET)(v[i] - fr);
        acc |= (uint32_t)(((uint64_t)x << (uint32_t)(sh + 32)) >> 32);
      }
      data[w] = acc;
    }
    return ob;
  }
  for (uint32_t w = (uint32_t)lane; w < words; w += kWave) {
    // LSB-first packing of (x - FOR) in bw bits (reference :523-552)
    const uint32_t b0 = w * 32;
    uint32_t acc = 0;
    for (uint32_t i = b0 / bw; i * bw < b0 + 32 && i < n; ++i) {
      const ET x = (ET)(v[i] - fr);
      const int sh = (int)(i * bw) - (int)b0;
      if (ES > 4) {
        const uint64_t xx = (uint64_t)x;
        acc |= (uint32_t)(sh > 0 ? xx << sh : xx >> (-sh));
      } else {
        const uint32_t xx = (uint32_t)x;
        acc |= sh > 0 ? xx << sh : xx >> (-sh);
      }
    }
    data[w] = acc;
  }
  return ob;
}

template <int S>
__device__ __forceinline__ uint32_t chunk_metadata_size(int R, int D)
{
  return ru((uint32_t)(4 + 4 * (R + 1)), (uint32_t)S) + ru((uint32_t)(S * D), 4u);
}

// ---------------------------------------------------------------------------
// 4-byte elements: the encoder's fast path (round 4).  PMC on the round-3 kernel and on the
// blocked RLE above said the same thing: the LDS is what the 25 waves of a CU queue on (70 - 83 %
// busy, half of it bank conflicts of the packer's gathers, then of the blocked scatter), so this
// path is built around LDS cycles:
//  * the array a layer works on stays in REGISTERS between layers where it can: 16 registers hold
//    a sub-chunk element-per-lane (v[k] = element 64 k + lane), loaded from HBM (first layer) or LDS;
//    a delta layer is register arithmetic (the right neighbour by one DPP move) and hands its
//    registers to the RLE that follows;
//  * RLE: per 64 elements one compare, one ballot, one rank (mbcnt) and one store group -- the run
//    ends of a step go to CONSECUTIVE ranks, so neither the value store nor the end-position store
//    has a bank conflict.  What is stored per run is where it ENDS; lengths are differences of those
//    and are taken by the packer in registers;
//  * the compacted arrays lie in LDS with one dword of padding behind every 32 elements, so that
//    the packer can take 16 consecutive elements per lane conflict-free (lanes 2 j and 2 j + 1 share a
//    block of 32: 33 dwords from pair to pair) -- a PAIR of lanes makes the whole number of output
//    words that 32 elements are whatever the bit width (pack16_store), by compile-time shifts (one
//    v_lshl_or_b32 per element for the bit widths met in practice), and minimum and maximum come out
//    of the same registers: no pass of its own, no gather.  (32 per lane, the first version: a sorted
//    column's arrays have 700 - 800 elements, and 40 lanes of 64 had nothing to pack);
//  * the chunk metadata image is a register (lane j = word j), not LDS: 4224 + 2176 bytes per wave,
//    25 waves per CU as before.
// ---------------------------------------------------------------------------
constexpr uint32_t kX4Bytes = (1024 + 32) * 4;     // 1024 elements, padded
constexpr uint32_t kE4Bytes = 1024 * 2 + 32 * 4;   // 1024 run ends (u16), padded
__device__ __forceinline__ uint32_t x4_addr(uint32_t i) { return (i + (i >> 5)) << 2; }         // byte offset of element i
__device__ __forceinline__ uint32_t e4_addr(uint32_t i) { return (i << 1) + ((i >> 5) << 2); }  // ... of run end i

__device__ __forceinline__ uint32_t lanes_below_me(uint64_t mask, uint32_t add)
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, add));
}

// the value of the lane above (lane 63: `last`): one DPP move (wave_shl:1; the lane without a source keeps `last`)
__device__ __forceinline__ uint32_t from_lane_above(uint32_t v, uint32_t last)
{
  return (uint32_t)__builtin_amdgcn_update_dpp((int)last, (int)v, 0x130, 0xF, 0xF, false);
}

// element 64 k + lane + 1 for every lane, out of step k's register `cur` and step k + 1's `next`: the lane
// above's `cur`, and for lane 63 lane 0's `next` -- which a rotation of `next` by one lane puts there (two
// DPP moves; through v_readlane it was a third instruction, a copy into a vector register and two waits
// for the scalar register in between)
__device__ __forceinline__ uint32_t next_element(uint32_t cur, uint32_t next)
{
  const int rolled = __builtin_amdgcn_mov_dpp((int)next, 0x134, 0xF, 0xF, false);               // wave_rol:1 (every lane has a source)
  return (uint32_t)__builtin_amdgcn_update_dpp(rolled, (int)cur, 0x130, 0xF, 0xF, false);       // wave_shl:1
}

// 64-lane reductions on the DPP crossbar: the result is lane 63's.  (A lane without a source takes the
// operation's identity: written that way the compiler folds the move into the minimum / maximum itself --
// one instruction per step; with the lane's own value as the fallback it was three: a copy, the move, the
// operation -- 72 of the encoder's vector instructions per sub-chunk.)
__device__ __forceinline__ int32_t wave_min_i32(int32_t v)
{
#define HC_STEP(CTRL, MASK)                                                                          \
  {                                                                                                  \
    const int32_t u = __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, MASK, 0xF, false);            \
    v = u < v ? u : v;                                                                               \
  }
  HC_STEP(0x111, 0xF) HC_STEP(0x112, 0xF) HC_STEP(0x114, 0xF) HC_STEP(0x118, 0xF) HC_STEP(0x142, 0xA) HC_STEP(0x143, 0xC)
#undef HC_STEP
  return (int32_t)read_lane((uint32_t)v, 63);
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
#define HC_STEP(CTRL, MASK)                                                                          \
  {                                                                                                  \
    const int32_t u = __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, MASK, 0xF, false);       \
    v = u > v ? u : v;                                                                               \
  }
  HC_STEP(0x111, 0xF) HC_STEP(0x112, 0xF) HC_STEP(0x114, 0xF) HC_STEP(0x118, 0xF) HC_STEP(0x142, 0xA) HC_STEP(0x143, 0xC)
#undef HC_STEP
  return (int32_t)read_lane((uint32_t)v, 63);
}

// the n elements of a sub-chunk element-per-lane: v[k] = element 64 k + lane (0 behind n)
__device__ __forceinline__ void load16_global(cgptr src, uint32_t n, uint32_t (&v)[16], int lane)
{
  // (one per-lane offset, the steps as immediate offsets of the loads; the lanes behind n in the last
  // step read element n - 1 again: in bounds, and not used by anything)
  // (the base is wave-uniform -- scalar registers -- and the lane's offset one 32-bit register for all 16
  // loads, the steps their immediate offsets: sixteen 64-bit addresses kept across the loop over the
  // sub-chunks were being spilled, and every reload of one waits for the stores of the sub-chunk before)
  const HC_GLOBAL uint32_t* base = reinterpret_cast<const HC_GLOBAL uint32_t*>(src);
  const uint32_t at = (uint32_t)lane;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    v[k] = 0;
    if (64u * k + 64u <= n) // (wave-uniform)
      v[k] = base[at + 64u * k];
    else if (64u * k < n)
      v[k] = base[min(64u * k + at, n - 1u)];
  }
}
// The same for a full sub-chunk (1024 elements, all but a partition's last one): ONE statement, the base in
// scalar registers, the lane's byte offset one 32-bit register, the steps the loads' immediate offsets.  (As
// written above the compiler made a 64-bit address per load -- three vector instructions each, 48 of the
// encoder's 926 per sub-chunk -- and a compare and two branches around every one of them.)  The wait is part
// of the statement: the compiler does not know that the registers are in flight.
__device__ __forceinline__ void load16_global_full(cgptr src, uint32_t (&v)[16], int lane)
{
  const uint32_t at = (uint32_t)lane << 2;
  asm volatile("s_nop 4\n\t" // (a scalar register written just before is not read as an address too early)
               "global_load_dword %0, %16, %17\n\t"
               "global_load_dword %1, %16, %17 offset:256\n\t"
               "global_load_dword %2, %16, %17 offset:512\n\t"
               "global_load_dword %3, %16, %17 offset:768\n\t"
               "global_load_dword %4, %16, %17 offset:1024\n\t"
               "global_load_dword %5, %16, %17 offset:1280\n\t"
               "global_load_dword %6, %16, %17 offset:1536\n\t"
               "global_load_dword %7, %16, %17 offset:1792\n\t"
               "global_load_dword %8, %16, %17 offset:2048\n\t"
               "global_load_dword %9, %16, %17 offset:2304\n\t"
               "global_load_dword %10, %16, %17 offset:2560\n\t"
               "global_load_dword %11, %16, %17 offset:2816\n\t"
               "global_load_dword %12, %16, %17 offset:3072\n\t"
               "global_load_dword %13, %16, %17 offset:3328\n\t"
               "global_load_dword %14, %16, %17 offset:3584\n\t"
               "global_load_dword %15, %16, %17 offset:3840\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                 "=&v"(v[8]), "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]),
                 "=&v"(v[15])
               : "v"(at), "s"(src)
               : "memory");
}
__device__ __forceinline__ void load16_lds(const uint8_t* X, uint32_t n, uint32_t (&v)[16], int lane)
{
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    // (all sixteen whatever n: behind n whatever lies there, in bounds, used by nothing -- four loads too many
    // on a sorted column's arrays, sixteen tests and branches fewer)
    v[k] = *reinterpret_cast<const uint32_t*>(X + x4_addr(64u * k + (uint32_t)lane));
  }
  // (all of them waited for here, once: what follows stores to LDS under conditions between its uses of
  // v[k], and the compiler, unable to count those stores, would wait for an empty LDS queue in front of
  // every step)
#pragma unroll
  for (int k = 0; k < 16; ++k)
    asm volatile("" : "+v"(v[k]));
}

// v[k] <- element i + 1 minus element i (i = 64 k + lane): the delta layer on n >= 1 elements in
// registers (reference block_delta_compress :317-328); what lands at and behind n - 1 is not used.
// (All sixteen steps whatever n, like load16_lds: no test at all.  A loop that is left at the first step behind
// n -- the form that pays in rle16 -- ran this kernel 14 % slower here and 4 % slower in load16_lds.)
__device__ __forceinline__ void delta16(uint32_t (&v)[16], uint32_t n)
{
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    v[k] = (k + 1 < 16 ? next_element(v[k], v[(k + 1) % 16]) : from_lane_above(v[k], 0u)) - v[k];
  }
}

// the array in registers -> LDS (padded), n elements, with the tail up to a multiple of 32 filled
// with copies of the last element (the packer takes whole blocks of 32: a copy does not move
// minimum or maximum, and the bits behind n are masked off)
__device__ __forceinline__ void store16_lds(uint8_t* X, uint32_t n, const uint32_t (&v)[16], int lane)
{
  uint32_t last = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (64u * k < n) {
      const uint32_t i = 64u * k + (uint32_t)lane;
      if (i < n)
        *reinterpret_cast<uint32_t*>(X + x4_addr(i)) = v[k];
      if (64u * k + 64u >= n)
        last = read_lane(v[k], (int)((n - 1u) & 63u));
    }
  }
  const uint32_t fill = (32u - (n & 31u)) & 31u;
  if ((uint32_t)lane < fill)
    *reinterpret_cast<uint32_t*>(X + x4_addr(n + (uint32_t)lane)) = last;
}

// RLE of the n elements in v (reference block_rle_compress :124-241): the values of the runs to
// X[0 .. m), the positions behind their last elements to Eb[0 .. m) (padded layouts), both filled
// up to a multiple of 32 entries for the packer (values: copies of the last one; ends: going on at
// the last run's length).  Returns m.
// FULL: n == 1024 (a full sub-chunk, all but a partition's last one: the steps' tests on n are gone).
template <bool FULL = false>
__device__ __forceinline__ uint32_t rle16(const uint32_t (&v)[16], uint32_t n, uint8_t* X, uint8_t* Eb, int lane)
{
  if (FULL)
    n = 1024u;
  uint32_t m = 0, last = 0;
  const uint32_t xbase = uniform((uint32_t)(uintptr_t)(const HC_LDS uint8_t*)X);
  // the run ends of a step: value and end position to consecutive ranks
  auto emit = [&](bool f, uint32_t cur, uint32_t end_pos) {
    const uint64_t ends = wave_ballot(f);
    if (f) {
      const uint32_t rank = lanes_below_me(ends, m);
      // X + 4 (rank / 32), the padding in front of the rank's block, as ONE instruction (the compiler makes
      // three of it: shift, mask, add)
      uint32_t padded;
      asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(padded) : "v"(rank >> 5), "s"(xbase));
      *reinterpret_cast<HC_LDS uint32_t*>((uintptr_t)(padded + (rank << 2))) = cur;
      *reinterpret_cast<HC_LDS uint16_t*>((uintptr_t)(padded + (kX4Bytes + (rank << 1)))) = (uint16_t)end_pos; // (Eb = X + kX4Bytes)
    }
    m += (uint32_t)__builtin_popcountll(ends);
  };
  if (!FULL && n == 0) { // (a delta layer left nothing: no run, nothing to fill)
    lds_lane_exchange_fence();
    return 0;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (FULL ? k < 15 : 64u * k + 64u < n) { // (wave-uniform) a full step: every element has a right neighbour
      const uint32_t cur = v[k];
      const uint32_t nx = next_element(cur, v[(k + 1) % 16]);
      emit(cur != nx, cur, 64u * k + 1u + (uint32_t)lane);
    } else { // the last step (64 k < n <= 64 k + 64): the last element ends a run whatever follows
      const uint32_t cur = v[k];
      const uint32_t nx = from_lane_above(cur, 0u);
      const uint32_t i1 = 64u * k + 1u + (uint32_t)lane;
      last = read_lane(cur, (int)((n - 1u) & 63u));
      emit((i1 < n && cur != nx) || i1 == n, cur, i1);
      break;
    }
  }
  const uint32_t fill = (32u - (m & 31u)) & 31u;
  if (fill != 0) { // (m >= 1 here: n >= 1)
    lds_lane_exchange_fence();
    uint32_t before = 0;
    if (m >= 2)
      before = uniform((uint32_t)*reinterpret_cast<const uint16_t*>(Eb + e4_addr(m - 2u)));
    lds_lane_exchange_fence();
    if ((uint32_t)lane < fill) {
      *reinterpret_cast<uint32_t*>(X + x4_addr(m + (uint32_t)lane)) = last;
      *reinterpret_cast<uint16_t*>(Eb + e4_addr(m + (uint32_t)lane)) = (uint16_t)(n + ((uint32_t)lane + 1u) * (n - before));
    }
  }
  lds_lane_exchange_fence();
  return m;
}

// Bit packing, 16 elements per lane (reference block_bitpack :523-552: element i at bit i BW, LSB first).
// A PAIR of lanes makes the BW words of 32 elements: the even lane packs elements [0, 16) from bit 0 on,
// the odd lane [16, 32) -- also from bit 0 on, and where BW is odd its 16 BW bits start in the middle of a
// word: the even lane then takes the odd lane's first half word into the upper half of its last one (one
// DPP move), and the odd lane's words are its own moved down by 16 bits (one v_alignbit each).  All shifts
// are known at compile time: one v_lshl_or_b32 per item, a second instruction where an item straddles two
// words.  PAIRS: the items are pairs of elements already joined to 2 BW bits (the run lengths, which
// come two to a register).  (Round 4, second pass: 32 elements per lane left 40 of the 64 lanes idle on
// arrays of 700 - 800 elements -- a sub-chunk of a sorted column -- at the full instruction count.)
template <int BW, bool PAIRS, class Get>
__device__ __forceinline__ void pack16_store(
    Get get, HC_GLOBAL uint32_t* data, uint32_t words, uint32_t last_mask, int lane)
{
  constexpr int EB = PAIRS ? 2 * BW : BW, NI = PAIRS ? 8 : 16;
  constexpr int H = (BW + 1) / 2; // words of 16 elements from bit 0 on = the words an even lane writes
  uint32_t w[H];
#pragma unroll
  for (int j = 0; j < H; ++j)
    w[j] = 0;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = i * EB, word = p >> 5, sh = p & 31;
    const uint32_t item = get(i);
    w[word] |= item << sh;
    if (sh + EB > 32)
      w[word + 1] |= item >> (32 - sh);
  }
  // (the lane number through an empty asm statement: what depends on it below -- for each of the 33 bit
  // widths -- is then not something to compute once in front of the loop over the sub-chunks, keep in 60
  // registers and spill; a reload from scratch waits for every store before it)
  asm volatile("" : "+v"(lane));
  const bool odd = (lane & 1) != 0;
  uint32_t mine = (uint32_t)H;
  if constexpr ((BW & 1) != 0) {
    // (quad_perm [1, 1, 3, 3]: every lane reads the odd lane of its pair)
    const uint32_t first_of_odd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[0], 0xF5, 0xF, 0xF, false);
    const uint32_t shared = w[H - 1] | (first_of_odd << 16);
#pragma unroll
    for (int j = 0; j + 1 < H; ++j) {
      const uint32_t down = __builtin_amdgcn_alignbit(w[j + 1], w[j], 16);
      w[j] = odd ? down : w[j];
    }
    w[H - 1] = shared;
    mine = odd ? (uint32_t)(H - 1) : (uint32_t)H;
  }
  // my words: the pair's are [pair BW, pair BW + BW), the odd lane's behind the even lane's H; the array's
  // last word holds no bit of the fill behind its last element
  const uint32_t w0 = (uint32_t)(lane >> 1) * (uint32_t)BW + (odd ? (uint32_t)H : 0u);
  const uint32_t inside = min(words - min(w0, words), mine); // how many of my words lie inside the array
  const uint32_t last_at = words - 1u - w0;                  // (which of mine is the array's last word, if any)
  HC_GLOBAL uint32_t* const q = data + w0;
#pragma unroll
  for (int j = 0; j < H; ++j)
    if ((uint32_t)j < inside)
      q[j] = (uint32_t)j == last_at ? w[j] & last_mask : w[j];
}

// 4-byte values of 17 .. 31 bits (columns that hardly compress): 32 elements per lane, the shifts in
// scalar registers, a loop that is not unrolled: the element goes from LDS into the accumulator and is gone.
__device__ __forceinline__ void write_wide4(
    HC_GLOBAL uint32_t* data, const HC_LDS uint8_t* X, uint32_t n, uint32_t bw, uint32_t fr, uint32_t words,
    uint32_t last_mask, int lane)
{
  const uint32_t blocks = (n + 31u) >> 5;
  if ((uint32_t)lane >= blocks)
    return;
  const HC_LDS uint8_t* p = X + (uint32_t)lane * 132u;
  uint64_t acc = 0;
  uint32_t sh = 0, wi = (uint32_t)lane * bw;
#pragma nounroll
  for (int k = 0; k < 32; ++k) {
    acc |= (uint64_t)(*reinterpret_cast<const HC_LDS uint32_t*>(p + 4 * k) - fr) << sh;
    sh += bw;
    if (sh >= 32u) { // (wave-uniform; once per element at these widths)
      if (wi < words)
        data[wi] = wi + 1u == words ? (uint32_t)acc & last_mask : (uint32_t)acc;
      ++wi;
      acc >>= 32;
      sh -= 32u;
    }
  }
}

// One array of the sub-chunk to HBM (reference block_write :646-680 / get_for_bitwidth :394-471 /
// block_bitpack) from the padded LDS layouts.  LENGTHS: the run lengths (16-bit elements), taken
// as differences of the run ends in Eb; else the 32-bit elements in X.  n elements (the arrays are
// filled to a multiple of 32, see rle16 / store16_lds).  Returns the byte length the format
// records, 0xFFFFFFFF when the array does not fit (reference :668-671).
template <bool LENGTHS>
__device__ __forceinline__ uint32_t write_array4(
    gptr out, uint32_t off, uint32_t limit, const uint8_t* X, const uint8_t* Eb, uint32_t n, int bp, int lane)
{
  constexpr uint32_t ES = LENGTHS ? 2 : 4;
  typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
  HC_GLOBAL uint32_t* dst = reinterpret_cast<HC_GLOBAL uint32_t*>(out + off);
  // my 16 elements: block min(lane, last block) -- the lanes behind the array look at its last
  // block again, which moves neither minimum nor maximum, and what they pack is stored nowhere
  const uint32_t blocks = (n + 15u) >> 4;
  const uint32_t tb = blocks == 0 ? 0u : min((uint32_t)lane, blocks - 1u);
  uint32_t y[16];
  uint32_t bw = 8 * ES; // (a raw array is its elements at their full width, no frame of reference)
  uint32_t fr = 0;
  if (LENGTHS) {
    // run lengths = differences of the run ends, two at a time (16-bit halves): pair q = {end 2q, end 2q + 1}
    // minus {end 2q - 1, end 2q}
    const uint8_t* p = Eb + (tb * 32u + (tb >> 1) * 4u); // e4_addr(16 tb)
    uint32_t e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      e[q] = *reinterpret_cast<const uint32_t*>(p + 4 * q);
    // the run end in front of my block: the entry below (block 0: 0) -- for an even block behind the
    // dword of padding
    uint32_t below = tb == 0 ? 0u : (uint32_t)*reinterpret_cast<const uint16_t*>(p - ((tb & 1u) ? 2 : 6)) << 16;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t shifted = __builtin_amdgcn_alignbit(e[q], below, 16); // {hi of the pair below, my lo}
      const u16x2 d = __builtin_bit_cast(u16x2, e[q]) - __builtin_bit_cast(u16x2, shifted);
      below = e[q];
      y[q] = __builtin_bit_cast(uint32_t, d);
    }
    if (bp) {
      // frame of reference = the smallest length, bit width from largest - smallest (reference
      // get_for_bitwidth :394-471; lengths are 1 .. 1024: the same signed or not)
      u16x2 mn2 = __builtin_bit_cast(u16x2, y[0]), mx2 = mn2;
#pragma unroll
      for (int q = 1; q < 8; ++q) {
        const u16x2 x = __builtin_bit_cast(u16x2, y[q]);
        mn2 = __builtin_elementwise_min(mn2, x);
        mx2 = __builtin_elementwise_max(mx2, x);
      }
      int32_t mn = 0, mx = 0;
      if (n > 0) {
        mn = wave_min_i32((int32_t)min((uint32_t)mn2.x, (uint32_t)mn2.y));
        mx = wave_max_i32((int32_t)max((uint32_t)mx2.x, (uint32_t)mx2.y));
      }
      const uint32_t range = (uint32_t)(mx - mn);
      bw = range ? 32u - (uint32_t)__builtin_clz(range) : 0u;
      fr = (uint32_t)mn;
      const uint32_t fr2 = fr | (fr << 16);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        y[q] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, y[q]) - __builtin_bit_cast(u16x2, fr2));
    }
  } else {
    const uint8_t* p = X + (tb * 16u + (tb >> 1)) * 4u; // x4_addr(16 tb)
#pragma unroll
    for (int k = 0; k < 16; ++k)
      y[k] = *reinterpret_cast<const uint32_t*>(p + 4 * k);
    if (bp) {
      // frame of reference = minimum under the SIGNED interpretation, bit width from max - min
      int32_t mn = 0, mx = 0;
      if (n > 0) {
        mn = (int32_t)y[0];
        mx = mn;
#pragma unroll
        for (int k = 1; k < 16; ++k) {
          const int32_t x = (int32_t)y[k];
          mn = x < mn ? x : mn;
          mx = x > mx ? x : mx;
        }
        mn = wave_min_i32(mn);
        mx = wave_max_i32(mx);
      }
      const uint32_t range = (uint32_t)mx - (uint32_t)mn;
      bw = range ? 32u - (uint32_t)__builtin_clz(range) : 0u;
      fr = (uint32_t)mn;
      if (bw <= 16u || bw == 32u) { // (the wide path reads the array again)
#pragma unroll
        for (int k = 0; k < 16; ++k)
          y[k] -= fr;
      }
    }
  }
  const uint32_t bits = n * bw;
  const uint32_t words = (bits + 31u) >> 5;
  const uint32_t ob = bp ? 8u + 4u * words : n * ES;
  if (off + (bp ? ob : 4u * words) > limit)
    return 0xFFFFFFFFu;
  if (bp && lane == 0) {
    // [FOR][pad to 4][bitwidth<<16 | n]; pads written as 0
    dst[0] = LENGTHS ? (fr & 0xFFFFu) : fr;
    dst[1] = (bw << 16) | n;
  }
  if (words == 0)
    return ob;
  HC_GLOBAL uint32_t* data = dst + (bp ? 2 : 0);
  const uint32_t tail = bits & 31u;
  const uint32_t last_mask = tail ? (1u << tail) - 1u : ~0u;
  if (LENGTHS) {
    // two lengths of a register -> one item of 2 bw bits: lo | hi << bw (both below 2^bw)
    if (bw < 16u) {
      const uint32_t low = (1u << bw) - 1u;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        y[q] = (y[q] & low) | ((y[q] >> (16u - bw)) & ~low);
    }
    auto get = [&](int i) -> uint32_t { return y[i % 8]; };
    switch (bw) {
#define HC_CASE(B) case B: pack16_store<B, true>(get, data, words, last_mask, lane); break;
      HC_CASE(1) HC_CASE(2) HC_CASE(3) HC_CASE(4) HC_CASE(5) HC_CASE(6) HC_CASE(7) HC_CASE(8)
      HC_CASE(9) HC_CASE(10) HC_CASE(11) HC_CASE(12) HC_CASE(13) HC_CASE(14) HC_CASE(15) HC_CASE(16)
#undef HC_CASE
    default: break; // (lengths are below 2^11)
    }
  } else {
    auto get = [&](int i) -> uint32_t { return y[i % 16]; };
    switch (bw) {
#define HC_CASE(B) case B: pack16_store<B, false>(get, data, words, last_mask, lane); break;
      HC_CASE(1) HC_CASE(2) HC_CASE(3) HC_CASE(4) HC_CASE(5) HC_CASE(6) HC_CASE(7) HC_CASE(8)
      HC_CASE(9) HC_CASE(10) HC_CASE(11) HC_CASE(12) HC_CASE(13) HC_CASE(14) HC_CASE(15) HC_CASE(16)
      HC_CASE(32)
#undef HC_CASE
    default:
      write_wide4(data, (const HC_LDS uint8_t*)X, n, bw, fr, words, last_mask, lane);
      break;
    }
  }
  return ob;
}

// One partition by the calling wave: in[0 .. in_bytes64) -> out, -> the compressed bytes (0: no input).
// `my`: the wave's wave_lds_bytes<S>() of LDS.
template <int S>
__device__ __forceinline__ uint32_t cascaded_encode_partition(
    cgptr in, const size_t in_bytes64, gptr out, uint8_t* my, const int type_tag, const int R, const int D, const int bp,
    const int lane)
{
  typedef typename UIntOf<S>::type UT;
  UT* bufA = reinterpret_cast<UT*>(my);
  uint16_t* cnts = reinterpret_cast<uint16_t*>(my + enc_buf_bytes());                       // (the generic path's layout)
  uint32_t* meta = reinterpret_cast<uint32_t*>(my + enc_buf_bytes() + (CB / S) * 2);
  (void)bufA;
  (void)cnts;
  (void)meta;
  if (in == nullptr || in_bytes64 == 0) // reference :856-860
    return 0;
  const uint32_t in_bytes = (uint32_t)in_bytes64;
  const uint32_t N = in_bytes / S;
  const uint32_t limit = 4u * (2u + (in_bytes + 3u) / 4u); // reference :852-854
  bool use = !(R == 0 && D == 0 && bp == 0);
  uint32_t cur = ru(kPartMeta, S);
  constexpr uint32_t CE = CB / S;
  const uint32_t nchunks = (N + CE - 1) / CE;
  const uint32_t msz = chunk_metadata_size<S>(R, D);
  const uint32_t dh_off = ru((uint32_t)(4 + 4 * (R + 1)), (uint32_t)S);
  const int layers = R > D ? R : D;

  if constexpr (S == 4) {
    // ---- the fast path (see rle16 / write_array4 above)
    static_assert(kWavesPerBlock == 1 || true, "");
    uint8_t* const X = my;
    uint8_t* const Eb = my + kX4Bytes;
    for (uint32_t c = 0; c < nchunks && use; ++c) {
      const uint32_t chunk_start = cur;
      cur += msz;
      uint32_t n = min(N - c * CE, CE);
      uint32_t img = 0; // the chunk metadata image: lane j holds its word j (reference :1004-1014)
      uint32_t v[16];
      if (n == CE)
        load16_global_full(in + (size_t)c * CB, v, lane);
      else
        load16_global(in + (size_t)c * CB, n, v, lane);
      // (measurement builds, scripts/pmc_cascaded_stages.sh: a sub-chunk is left behind stage HC_CASC_STOP_AFTER --
      // 1 RLE, 2 its lengths packed, 3 delta, 4 the second RLE, 5 its lengths packed -- and the counters of two such
      // builds differ by what the stage between them executes; the output is then not a stream)
#if HC_CASC_STOP_AFTER
      int stage = 0;
#define HC_STAGE_DONE()                       \
  if (++stage == HC_CASC_STOP_AFTER) {        \
    cut_short = true;                         \
    break;                                    \
  }
      bool cut_short = false;
#else
#define HC_STAGE_DONE()
#endif
      // (Touching the next sub-chunk's lines here, so that its loads hit the L2 -- a wave meets the HBM's
      // latency 16 times per partition -- made the kernel slower, 1958 -> 1825 GB/s: it is not what the
      // waves wait for.)
      bool in_regs = true; // the array the next layer works on is in v (else: in X)
      int rr = R, dr = D;
      for (int l = 0; l < layers && use; ++l) {
        if (rr > 0) { // reference :913-953
          if (!in_regs)
            load16_lds(X, n, v, lane);
          const uint32_t m = (in_regs && n == CE) ? rle16<true>(v, n, X, Eb, lane) : rle16(v, n, X, Eb, lane);
#pragma unroll
          for (int k = 0; k < 16; ++k)
            v[k] = 0; // (the registers are free from here on: what comes next loads its own)
          HC_STAGE_DONE()
          const uint32_t ob = write_array4<true>(out, cur, limit, X, Eb, m, bp, lane);
          if (ob == 0xFFFFFFFFu) {
            use = false;
            break;
          }
          cur += ru(ob, 4);
          img = lane == R - rr + 1 ? ob : img;
          n = m;
          --rr;
          in_regs = false;
          HC_STAGE_DONE()
        }
        if (dr > 0) { // reference :955-977
          if (n == 0) { // undefined in the reference (:323); raw fallback here
            use = false;
            break;
          }
          if (!in_regs)
            load16_lds(X, n, v, lane);
          const uint32_t head = read_lane(v[0], 0);
          img = lane == (int)(dh_off / 4) + (D - dr) ? head : img;
          delta16(v, n);
          n -= 1;
          --dr;
          in_regs = true;
          HC_STAGE_DONE()
        }
      }
#undef HC_STAGE_DONE
      if (!use)
        break;
#if HC_CASC_STOP_AFTER
      if (cut_short) {
        if (in_regs) { // (what the stage left in registers counts as used)
#pragma unroll
          for (int k = 0; k < 16; ++k)
            asm volatile("" ::"v"(v[k]));
        }
        continue;
      }
#endif
      if (in_regs)
        store16_lds(X, n, v, lane);
#pragma unroll
      for (int k = 0; k < 16; ++k)
        v[k] = 0;
      lds_lane_exchange_fence();
      const uint32_t ob = write_array4<false>(out, cur, limit, X, Eb, n, bp, lane); // (reference :983-984: 4-byte elements need no alignment gap)
      if (ob == 0xFFFFFFFFu) {
        use = false;
        break;
      }
      cur += ru(ob, 4); // reference :999-1001
      img = lane == 0 ? cur - chunk_start : img;
      img = lane == R + 1 ? ob : img;
      // flush the chunk metadata image (reference :1004-1014)
      if ((uint32_t)lane < msz / 4)
        reinterpret_cast<HC_GLOBAL uint32_t*>(out + chunk_start)[lane] = img;
    }
  } else
  for (uint32_t c = 0; c < nchunks && use; ++c) {
    const uint32_t chunk_start = cur;
    cur += msz;
    uint32_t n = min(N - c * CE, CE);
    // sub-chunk -> LDS (16 bytes per lane per step; inputs are 4-byte aligned)
    {
      cgptr src = in + (size_t)c * CB;
      const uint32_t nb = n * S;
      for (uint32_t o = (uint32_t)lane * 16u; o < nb; o += kWave * 16u) {
        if (o + 16 <= nb) {
          const u32x4 q = load_u128_any(src + o);
          *reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(bufA) + o) = q;
        } else {
          for (uint32_t k = o; k < nb; ++k)
            reinterpret_cast<uint8_t*>(bufA)[k] = src[k];
        }
      }
      if (lane < 16)
        meta[lane] = 0;
    }
    // Both layers work in place: a round reads its elements and one more and
    // then writes at or below them (RLE compacts, delta keeps the index), LDS
    // operations of a wave execute in order, and within a round every lane's
    // loads precede every lane's store.
    UT* const x = bufA;
    int rr = R, dr = D;
    for (int l = 0; l < layers && use; ++l) {
      if (rr > 0) { // reference :913-953
        const uint32_t m = wave_rle<UT>(x, n, cnts, lane);
        const uint32_t ob = wave_write_array<uint16_t>(out, cur, limit, cnts, m, bp, lane);
        if (ob == 0xFFFFFFFFu) {
          use = false;
          break;
        }
        cur += ru(ob, 4);
        if (lane == 0)
          meta[R - rr + 1] = ob;
        n = m;
        --rr;
      }
      if (dr > 0) { // reference :955-977
        if (n == 0) { // undefined in the reference (:323); raw fallback here
          use = false;
          break;
        }
        if (lane == 0)
          *reinterpret_cast<UT*>(reinterpret_cast<uint8_t*>(meta) + dh_off + (D - dr) * S) = x[0];
        lds_lane_exchange_fence();
        wave_delta<UT>(x, n, lane);
        n -= 1;
        --dr;
      }
    }
    if (!use)
      break;
    const uint32_t fin = ru(cur, S); // reference :983-984
    if (S > 4 && fin > cur && lane == 0)
      *reinterpret_cast<HC_GLOBAL uint32_t*>(out + cur) = 0; // alignment gap
    const uint32_t ob = wave_write_array<UT>(out, fin, limit, x, n, bp, lane);
    if (ob == 0xFFFFFFFFu) {
      use = false;
      break;
    }
    const uint32_t after = fin + ru(ob, 4);
    cur = ru(after, S); // reference :999-1001
    if (S > 4 && cur > after && lane == 0)
      *reinterpret_cast<HC_GLOBAL uint32_t*>(out + after) = 0;
    if (lane == 0) {
      meta[0] = cur - chunk_start;
      meta[R + 1] = ob;
    }
    // flush the chunk metadata image (reference :1004-1014)
    if ((uint32_t)lane < msz / 4)
      reinterpret_cast<HC_GLOBAL uint32_t*>(out + chunk_start)[lane] = meta[lane];
  }

  uint32_t total;
  if (use) {
    total = cur;
  } else { // raw fallback (reference :1019-1053)
    const uint32_t raw = ru(kPartMeta, S);
    const uint32_t nb = N * S;
    wave_copy(out + raw, in, nb, lane);
    if ((nb & 3u) && lane == 0)
      for (uint32_t k = nb; k < ru(nb, 4); ++k)
        out[raw + k] = 0;
    total = raw + ru(nb, 4);
  }
  if (lane == 0) {
    const uint32_t h = use ? ((uint32_t)R | ((uint32_t)D << 8) | ((uint32_t)bp << 16)) : 0u;
    reinterpret_cast<HC_GLOBAL uint32_t*>(out)[0] = h | ((uint32_t)type_tag << 24);
    reinterpret_cast<HC_GLOBAL uint32_t*>(out)[1] = N * S;
  }
  return total;
}

// (launch bound: HC_CASC_OCC waves per SIMD; LDS allows 25 one-wave workgroups per CU)
#ifndef HC_CASC_OCC
#define HC_CASC_OCC 6
#endif
template <int S>
__global__ __launch_bounds__(kWave * kWavesPerBlock, HC_CASC_OCC) void cascaded_compress_kernel(
    const uint8_t* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes_arr,
    uint8_t* const* __restrict__ out_ptrs,
    size_t* __restrict__ out_bytes_arr, const size_t batch, const int type_tag,
    const int R, const int D, const int bp)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; // kWavesPerBlock * wave_lds_bytes<S>()
  const int lane = lane_id();
  // everything that steers the layers is wave-uniform: say so (see uniform())
  const int wave = (int)uniform((uint32_t)(threadIdx.x >> 6));
  const size_t part = (size_t)blockIdx.x * kWavesPerBlock + wave;
  if (part >= batch)
    return;
  uint8_t* my = smem + wave * wave_lds_bytes<S>();
  cgptr in = to_global(uniform_ptr(in_ptrs[part]));
  const size_t in_bytes64 = uniform((uint64_t)in_bytes_arr[part]);
  gptr out = to_global(uniform_ptr(out_ptrs[part]));
  const uint32_t total = cascaded_encode_partition<S>(in, in_bytes64, out, my, type_tag, R, D, bp, lane);
  if (lane == 0)
    out_bytes_arr[part] = total;
}

// The same for the high-level manager (placement.hpp; as snappy_compress_placed_kernel): one wave per
// workgroup, as many workgroups as the device holds, partitions off a ticket counter, each compressed into the
// wave's slot and moved to its place in the container.
// (5 waves per SIMD instead of 6, no spills: 1 990 -> 1 915 GB/s)
template <int S>
__global__ __launch_bounds__(kWave, HC_CASC_OCC) void cascaded_compress_placed_kernel(
    const uint8_t* const* __restrict__ in_ptrs, const size_t* __restrict__ in_bytes_arr,
    size_t* __restrict__ out_bytes_arr, const uint32_t batch, const int type_tag, const int R, const int D, const int bp,
    uint32_t* __restrict__ ticket, const Placement place)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; // wave_lds_bytes<S>()
  const int lane = lane_id();
  gptr slot = to_global(place.slots + (size_t)blockIdx.x * place.slot_bytes);
  for (uint32_t part = next_chunk(ticket, lane); part < batch;) {
    const uint32_t asked = ask_next_chunk(ticket, lane);
    cgptr in = to_global(uniform_ptr(in_ptrs[part]));
    const size_t in_bytes64 = uniform((uint64_t)in_bytes_arr[part]);
    const uint32_t total = cascaded_encode_partition<S>(in, in_bytes64, slot, smem, type_tag, R, D, bp, lane);
    if (lane == 0)
      out_bytes_arr[part] = total;
    // (an empty partition takes no room: offset = the cursor as it stands.  Measured at 100 000 partitions: the
    // manager's compress 3.14 ms, without this copy 3.05, without its atomic add either 3.01; the batched call 2.78)
    place_chunk(place, part, slot, total, lane);
    part = chunk_asked_for(asked);
  }
}

// ---------------------------------------------------------------------------
// Decoder
// ---------------------------------------------------------------------------

// One array into an LDS element buffer (reference block_read :702-737 +
// block_bitunpack :563-618).  `src` are the array's 32-bit words, either in
// the staged LDS image or in HBM.  Returns the element count, or -1 when the
// array leaves the sub-chunk buffer.
template <typename ET, typename WordPtr>
__device__ __forceinline__ int unpack_array(
    WordPtr src, uint32_t nbytes, int bp, ET* dst, uint32_t max_elems, int lane)
{
  constexpr uint32_t ES = sizeof(ET);
  if (!bp) {
    const uint32_t n = nbytes / ES;
    if (n > max_elems)
      return -1;
    const uint32_t words = (n * ES + 3) / 4;
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    for (uint32_t w = (uint32_t)lane; w < words; w += kWave)
      d[w] = src[w];
    return (int)n;
  }
  constexpr uint32_t HDRW = ES > 4 ? 4 : 2; // header words
  if (nbytes < HDRW * 4)
    return -1;
  ET fr;
  if (ES > 4)
    fr = (ET)uniform((uint64_t)src[0] | ((uint64_t)src[1] << 32));
  else
    fr = (ET)uniform((uint32_t)src[0]);
  const uint32_t word = uniform((uint32_t)src[ES > 4 ? 2 : 1]);
  const uint32_t bw = word >> 16;
  const uint32_t n = word & 0xFFFFu;
  if (n == 0)
    return 0;
  if (n > max_elems || bw > 8 * ES)
    return -1;
  const uint32_t words = (n * bw + 31) / 32;
  if (HDRW * 4 + 4 * words > ru(nbytes, 4))
    return -1;
  WordPtr data = src + HDRW;
  for (uint32_t i = (uint32_t)lane; i < n; i += kWave) {
    ET x = 0;
    if (bw) {
      const uint32_t bit = i * bw;
      const uint32_t w0 = bit >> 5, sh = bit & 31u;
      const uint32_t last = words - 1;
      const uint64_t lo = (uint64_t)data[w0] | ((uint64_t)data[min(w0 + 1, last)] << 32);
      uint64_t v = lo >> sh;
      if (ES > 4 && sh + bw > 64)
        v |= (uint64_t)data[min(w0 + 2, last)] << (64 - sh);
      const uint64_t m = bw >= 64 ? ~0ull : ((1ull << bw) - 1ull);
      x = (ET)(v & m);
    }
    dst[i] = (ET)(x + fr);
  }
  return (int)n;
}

// The same array, element by element (for a pass that uses the values at once
// instead of parking them in LDS).  open() returns the element count or -1.
template <typename ET, typename WordPtr>
struct ArrayReader {
  static constexpr uint32_t ES = sizeof(ET);
  WordPtr data;
  uint32_t bw, last, fr;
  int bp;
  __device__ __forceinline__ int open(WordPtr src, uint32_t nbytes, int bp_, uint32_t max_elems)
  {
    static_assert(ES <= 4, "run lengths are 16-bit");
    bp = bp_;
    if (!bp) {
      const uint32_t n = nbytes / ES;
      data = src;
      return n > max_elems ? -1 : (int)n;
    }
    if (nbytes < 8)
      return -1;
    fr = uniform((uint32_t)src[0]);
    const uint32_t word = uniform((uint32_t)src[1]);
    bw = word >> 16;
    const uint32_t n = word & 0xFFFFu;
    if (n == 0)
      return 0;
    if (n > max_elems || bw > 8 * ES)
      return -1;
    const uint32_t words = (n * bw + 31) / 32;
    if (8 + 4 * words > ru(nbytes, 4))
      return -1;
    data = src + 2;
    last = words - 1;
    return (int)n;
  }
  // element i (< the count open() returned)
  __device__ __forceinline__ ET get(uint32_t i) const
  {
    if (!bp) {
      const uint32_t b = i * ES;
      return (ET)((uint32_t)data[b >> 2] >> (8u * (b & 3u)));
    }
    uint32_t x = 0;
    if (bw) {
      const uint32_t bit = i * bw;
      const uint32_t w0 = bit >> 5, sh = bit & 31u;
      const uint64_t lo = (uint64_t)data[w0] | ((uint64_t)data[min(w0 + 1, last)] << 32);
      x = (uint32_t)(lo >> sh) & (uint32_t)((1ull << bw) - 1ull);
    }
    return (ET)(x + fr);
  }
};

// Array at byte offset `rel` of the sub-chunk that starts at comp + pos:
// bounds as in the reference (:712-713), source = staged image when the array
// lies inside it.
template <typename ET, uint32_t STAGE_WORDS>
__device__ __forceinline__ int wave_read_array(
    cgptr comp, uint32_t end_words, uint32_t pos, uint32_t rel, uint32_t nbytes,
    int bp, const uint32_t* stage, ET* dst, uint32_t max_elems, int lane)
{
  const uint32_t off = pos + rel;
  if (!array_inside(pos, rel, nbytes, end_words))
    return -1;
  if (array_staged(rel, nbytes, STAGE_WORDS * 4))
    return unpack_array<ET>(stage + rel / 4, nbytes, bp, dst, max_elems, lane);
  return unpack_array<ET>(reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + off), nbytes, bp, dst,
                          max_elems, lane);
}

// ---- the blocked passes of the decoder --------------------------------------
// E values of a round per lane -> the running sum in front of each of them (an exclusive prefix
// over the whole array, `carry` = the sum in front of the round), stored one element on: y[pos0 + k + 1]
// = carry + v[0 .. k] summed, and y[first element of the round] = carry by lane 0.  In registers that
// is a shift by one element: what lane t stores at y[pos0 + k] is the sum up to its own element k - 1,
// or for k = 0 the lane below's last one (one DPP move), so every lane stores an aligned block.
// Returns the new carry.  `store`: the lanes whose block (elements pos0 .. pos0 + E - 1 of y) is wanted.
template <typename UT>
__device__ __forceinline__ UT prefix_store_shifted(
    UT (&v)[Blocked<sizeof(UT)>::E], UT carry, UT* y, uint32_t pos0, bool store, int lane)
{
  constexpr int S = sizeof(UT), E = Blocked<S>::E;
#pragma unroll
  for (int k = 1; k < E; ++k)
    v[k] = (UT)(v[k] + v[k - 1]); // inclusive, inside the lane
  UT incl;
  if (S > 4)
    incl = (UT)wave_scan_add_u64((uint64_t)v[E - 1]);
  else
    incl = (UT)wave_scan_add_u32((uint32_t)v[E - 1]);
  const UT base = (UT)(carry + incl - v[E - 1]); // the sum in front of my block
  typedef typename std::conditional<(S > 4), uint64_t, uint32_t>::type WT;
  UT s[E];
  s[0] = (UT)from_lane_below((WT)(UT)(base + v[E - 1]), (WT)carry, lane);
#pragma unroll
  for (int k = 1; k < E; ++k)
    s[k] = (UT)(base + v[k - 1]);
  if (store)
    store_block(y + pos0, s);
  UT total;
  if (S > 4)
    total = (UT)((uint64_t)read_lane((uint32_t)((uint64_t)incl), 63) | ((uint64_t)read_lane((uint32_t)((uint64_t)incl >> 32), 63) << 32));
  else
    total = (UT)read_lane((uint32_t)incl, 63);
  return (UT)(carry + total);
}

// ---------------------------------------------------------------------------
// 4-byte elements: the decoder's fast path (round 4), the mirror of the encoder's (rle16 /
// write_array4).  Element buffers in the padded layout (a dword behind every 32 elements: a lane's
// block of 32 or 16 consecutive elements is conflict-free dword by dword); a bit-packed array comes
// out of its words 32 elements per lane with compile-time shifts (one v_bfe / v_alignbit per
// element); the run lengths never touch LDS: their prefix sum is 32 additions inside the lane plus
// one wave scan, and every run drops its marker at its start position itself.
// What bounds this decoder is how many waves a CU holds (PMC: no unit half busy, the waves wait on
// their own LDS round trips; rounds 2-3 measured throughput proportional to the waves: 10 / 12 / 14 per
// CU = 713 / 844 / 960 GB/s), and LDS is what limits them.  So ONE element buffer: a sub-chunk of 4-byte
// elements is a single round of 64 lanes x 16 elements, all of whose reads come before its stores (LDS
// operations of a wave execute in order), so every pass -- an RLE expansion, a delta layer, the
// unpacking of an array whose words had to be copied in from HBM first -- works in place.
// LDS of a wave: [stage 1 KiB][element buffer, padded][markers 2 KiB + one entry] = 7 312 bytes, 6
// allocation granules: 21 waves per CU (rounds 1-3: two buffers, 14).  The one entry behind the
// markers: the runs of a lane behind the last run of the array (their bits read as 0) put their
// markers at `total`, behind the output, where nothing looks -- entry 1024 when the sub-chunk is full.
// ---------------------------------------------------------------------------
constexpr uint32_t kDec4Stage = 0, kDec4X = 1024, kDec4Marks = 1024 + kX4Bytes;
constexpr uint32_t kDec4Bytes = kDec4Marks + 2048 + 16;

// element k (0 .. 31) of BW bits out of the BW words w (LSB first, reference block_bitunpack :563-618)
template <int BW>
__device__ __forceinline__ uint32_t unpacked(const uint32_t (&w)[BW == 0 ? 1 : BW], int k)
{
  if (BW == 0)
    return 0u;
  const int p = k * BW, word = p >> 5, sh = p & 31;
  uint32_t x = w[word % (BW == 0 ? 1 : BW)] >> sh;
  if (sh + BW > 32)
    x |= w[(word + 1) % (BW == 0 ? 1 : BW)] << (32 - sh);
  return BW >= 32 ? x : x & ((1u << (BW & 31)) - 1u);
}

// What an array header says (reference block_read :702-737 + block_bitunpack :563-618, bounds as in
// unpack_array above): the element count (-1: the array leaves the sub-chunk buffer or is
// malformed), and for n > 0 the bit width, the frame of reference and where the words begin.
template <int ES, typename WordPtr>
__device__ __forceinline__ int open_array4(
    WordPtr src, uint32_t nbytes, int bp, uint32_t max_elems, uint32_t& bw, uint32_t& fr, WordPtr& data, uint32_t& words)
{
  if (!bp) {
    const uint32_t n = nbytes / ES;
    if (n > max_elems)
      return -1;
    bw = 8 * ES;
    fr = 0;
    data = src;
    words = (n * ES + 3) / 4;
    return (int)n;
  }
  if (nbytes < 8)
    return -1;
  fr = uniform((uint32_t)src[0]);
  const uint32_t word = uniform((uint32_t)src[1]);
  bw = word >> 16;
  const uint32_t n = word & 0xFFFFu;
  if (n == 0)
    return 0;
  if (n > max_elems || bw > 8 * ES)
    return -1;
  words = (n * bw + 31) / 32;
  if (8 + 4 * words > ru(nbytes, 4))
    return -1;
  data = src + 2;
  return (int)n;
}

// my BW words of an array of `words` words: [lane BW, lane BW + BW), the ones behind the array read
// as its last word (in bounds); then the bits behind element n are cleared
template <int BW, typename WordPtr>
__device__ __forceinline__ void load_words(
    WordPtr data, uint32_t words, uint32_t n, uint32_t (&w)[BW == 0 ? 1 : BW], int lane)
{
  if (BW == 0) {
    w[0] = 0;
    return;
  }
  const uint32_t w0 = (uint32_t)lane * BW;
#pragma unroll
  for (int j = 0; j < BW; ++j)
    w[j] = data[min(w0 + (uint32_t)j, words - 1u)];
  // bits of mine that belong to elements: n BW - 32 w0, clamped to [0, 32 BW]
  const int32_t left = (int32_t)(n * BW) - (int32_t)(32u * w0);
  if (left < 32 * BW) { // (the lane the array ends in, and the lanes behind it)
#pragma unroll
    for (int j = 0; j < BW; ++j) {
      const int32_t keep = left - 32 * j;
      w[j] = keep >= 32 ? w[j] : (keep <= 0 ? 0u : w[j] & ((1u << (keep & 31)) - 1u));
    }
  }
}

// The same 16 elements per lane (the encoder's pack16_store read backwards): the BW words of 32 elements
// belong to a PAIR of lanes; the even lane's 16 elements begin at bit 0 of them, the odd lane's at bit
// 16 BW -- word BW / 2 (rounded down), and where BW is odd 16 bits into that word: its words are then
// moved down by 16 bits (one v_alignbit each).  H = the words 16 elements span.
template <int BW>
__device__ __forceinline__ uint32_t unpacked16(const uint32_t (&w)[BW == 0 ? 1 : (BW + 1) / 2], int k)
{
  if (BW == 0)
    return 0u;
  constexpr int H = BW == 0 ? 1 : (BW + 1) / 2;
  const int p = k * BW, word = p >> 5, sh = p & 31;
  uint32_t x = w[word % H] >> sh;
  if (sh + BW > 32)
    x |= w[(word + 1) % H] << (32 - sh);
  return BW >= 32 ? x : x & ((1u << (BW & 31)) - 1u);
}
template <int BW, typename WordPtr>
__device__ __forceinline__ void load_words16(
    WordPtr data, uint32_t words, uint32_t n, uint32_t (&w)[BW == 0 ? 1 : (BW + 1) / 2], int lane)
{
  if (BW == 0) {
    w[0] = 0;
    return;
  }
  constexpr int H = BW == 0 ? 1 : (BW + 1) / 2;
  // (the lane number through an empty asm statement, as in pack16_store: what depends on it here, for each
  // bit width, is not to be computed once in front of the loop over the sub-chunks and kept)
  asm volatile("" : "+v"(lane));
  const bool odd = (lane & 1) != 0;
  const uint32_t w0 = (uint32_t)(lane >> 1) * BW + (odd ? (uint32_t)(BW / 2) : 0u);
#pragma unroll
  for (int j = 0; j < H; ++j)
    w[j] = data[min(w0 + (uint32_t)j, words - 1u)];
  if ((BW & 1) != 0) {
#pragma unroll
    for (int j = 0; j + 1 < H; ++j) {
      const uint32_t down = __builtin_amdgcn_alignbit(w[j + 1], w[j], 16);
      w[j] = odd ? down : w[j];
    }
    w[H - 1] = odd ? w[H - 1] >> 16 : w[H - 1];
  }
  // bits of mine that belong to elements: n BW minus where my 16 begin, clamped to [0, 16 BW]
  const int32_t left = (int32_t)(n * BW) - (int32_t)(16u * (uint32_t)lane * BW);
  if (left < 16 * BW) { // (the lane the array ends in, and the lanes behind it)
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const int32_t keep = left - 32 * j;
      w[j] = keep >= 32 ? w[j] : (keep <= 0 ? 0u : w[j] & ((1u << (keep & 31)) - 1u));
    }
  }
}

// The final array of a sub-chunk -> X (padded), 16 elements per lane.  Returns the count or -1.
// (LDS-typed pointers: through generic ones every access here would be a flat_* one.  Inlined -- for a
// while this and mark_run_starts4 were out of line, because with 32 elements per lane the 18 bit widths of
// each in the middle of the layer loop took the kernel to 300 registers; a call, though, waits for every
// vector memory operation in flight on both sides of it (the callee may use any register), i.e. for
// the words of the next sub-chunk asked for at the top of this one AND for the stores of the one before,
// three times per sub-chunk: with 16 per lane they fit inline, 2 217 -> 2 637 GB/s)
__device__ __forceinline__ int unpack_values4(const HC_LDS uint32_t* src, uint32_t nbytes, int bp, HC_LDS uint8_t* X, int lane)
{
  typedef const HC_LDS uint32_t* WordPtr;
  uint32_t bw, fr, words = 0;
  WordPtr data = src;
  const int n = open_array4<4>(src, nbytes, bp, 1024u, bw, fr, data, words);
  if (n <= 0)
    return n;
  const uint32_t blocks = ((uint32_t)n + 15u) >> 4;
  HC_LDS uint32_t* const mine = reinterpret_cast<HC_LDS uint32_t*>(X + ((uint32_t)lane * 16u + ((uint32_t)lane >> 1)) * 4u); // x4_addr(16 lane)
  auto go = [&](auto BWC) {
    constexpr int BW = decltype(BWC)::value;
    if ((uint32_t)lane < blocks) {
      uint32_t w[BW == 0 ? 1 : (BW + 1) / 2];
      load_words16<BW>(data, words, (uint32_t)n, w, lane);
#pragma unroll
      for (int k = 0; k < 16; ++k)
        mine[k] = unpacked16<BW>(w, k) + fr;
    }
  };
  switch (bw) {
#define HC_CASE(B) case B: go(std::integral_constant<int, B>{}); break;
    HC_CASE(0) HC_CASE(1) HC_CASE(2) HC_CASE(3) HC_CASE(4) HC_CASE(5) HC_CASE(6) HC_CASE(7) HC_CASE(8)
    HC_CASE(9) HC_CASE(10) HC_CASE(11) HC_CASE(12) HC_CASE(13) HC_CASE(14) HC_CASE(15) HC_CASE(16) HC_CASE(32)
#undef HC_CASE
  default: return -3; // (the bit widths 17 .. 31: unpack_values_wide4, from HBM)
  }
  return n;
}

// The final array of a sub-chunk at bit widths 17 .. 31 (a column that hardly compresses), an
// element per lane, straight from HBM (what rounds 1-3 did for every width).  Returns the count or -1.
__device__ __forceinline__ int unpack_values_wide4(
    const HC_GLOBAL uint32_t* src, uint32_t nbytes, int bp, uint8_t* X, int lane)
{
  uint32_t bw, fr, words = 0;
  const HC_GLOBAL uint32_t* data = src;
  const int n = open_array4<4>(src, nbytes, bp, 1024u, bw, fr, data, words);
  if (n <= 0)
    return n;
  const uint32_t last = words - 1;
  for (uint32_t i = (uint32_t)lane; i < (uint32_t)n; i += kWave) {
    const uint32_t bit = i * bw;
    const uint32_t w0 = bit >> 5, sh = bit & 31u;
    const uint64_t lo = (uint64_t)data[w0] | ((uint64_t)data[min(w0 + 1, last)] << 32);
    *reinterpret_cast<uint32_t*>(X + x4_addr(i)) = ((uint32_t)(lo >> sh) & (uint32_t)((1ull << (bw & 63)) - 1ull)) + fr;
  }
  return n;
}

// The run lengths of a layer -> markers: run i (counted from 1) drops i at its start position, the
// exclusive prefix sum of the lengths.  Returns the number of runs (-1: malformed), `total` = the
// sum of the lengths.  The markers must have been zeroed.
struct Runs4
{
  int n;          // number of runs; -1 / -2: malformed / too long; -3: take the element-per-lane code
  uint32_t total; // sum of the lengths
};
__device__ __forceinline__ Runs4 mark_run_starts4(
    const HC_LDS uint32_t* src, uint32_t nbytes, int bp, HC_LDS uint16_t* marks, int lane)
{
  uint32_t total = 0; // (a local, returned by value: through a reference every use is a flat load)
  typedef const HC_LDS uint32_t* WordPtr;
  uint32_t bw, fr, words = 0;
  WordPtr data = src;
  const int n = open_array4<2>(src, nbytes, bp, 1024u, bw, fr, data, words);
  if (n <= 0)
    return Runs4{n, 0u};
  fr &= 0xFFFFu;
  // (lengths are 16-bit and x + FOR wraps there in the reference and the oracle: a frame of reference that
  // could make one wrap is not an encoder's -- such a stream takes the element-per-lane code, -3)
  if (fr + (bw >= 16 ? 0xFFFFu : (1u << bw) - 1u) > 0xFFFFu)
    return Runs4{-3, 0u};
  const uint32_t blocks = ((uint32_t)n + 15u) >> 4;
  const bool active = (uint32_t)lane < blocks;
  const uint32_t nv = active ? min((uint32_t)n - 16u * (uint32_t)lane, 16u) : 0u; // my runs
  bool good = true;
  auto go = [&](auto BWC) {
    constexpr int BW = decltype(BWC)::value;
    // my 16 lengths, then their running sums (behind the array's last run the bits read as 0: length =
    // FOR there, taken off the lane's total again below)
    uint32_t run[16];
    if (active) {
      uint32_t w[BW == 0 ? 1 : (BW + 1) / 2];
      load_words16<BW>(data, words, (uint32_t)n, w, lane);
#pragma unroll
      for (int k = 0; k < 16; ++k)
        run[k] = unpacked16<BW>(w, k) + fr;
#pragma unroll
      for (int k = 1; k < 16; ++k)
        run[k] += run[k - 1];
    }
    const uint32_t lane_total = active ? run[15] - (16u - nv) * fr : 0u;
    const uint32_t incl = wave_scan_add_u32(lane_total);
    total = read_lane(incl, 63);
    if (total > 1024u) { // (runs longer than the sub-chunk: reference :1371-1380 would write past its buffer)
      good = false;
      return;
    }
    if (active) {
      const uint32_t base = incl - lane_total;
      // A marker is the PADDED index of the run's value in the element buffer, plus one (0: no run starts
      // here): 16 lane + lane / 2 + k + 1 for run 16 lane + k -- it grows with the run like the run's number
      // does, and the expansion needs no address arithmetic.  The runs of mine behind the array's last run
      // start at or behind `total`: their markers go to `total`, behind the output (see the layout note above).
      const uint32_t first = 16u * (uint32_t)lane + ((uint32_t)lane >> 1) + 1u;
      marks[base] = (uint16_t)first;
#pragma unroll
      for (int k = 1; k < 16; ++k)
        marks[min(base + run[k - 1], total)] = (uint16_t)(first + (uint32_t)k);
    }
  };
  switch (bw) {
#define HC_CASE(B) case B: go(std::integral_constant<int, B>{}); break;
    HC_CASE(0) HC_CASE(1) HC_CASE(2) HC_CASE(3) HC_CASE(4) HC_CASE(5) HC_CASE(6) HC_CASE(7) HC_CASE(8)
    HC_CASE(9) HC_CASE(10) HC_CASE(11) HC_CASE(12)
#undef HC_CASE
  default: return Runs4{-3, 0u}; // (run lengths of a 1024-element sub-chunk need 11 bits at most: wider ones, 13 .. 16, element by element)
  }
  return Runs4{good ? n : -2, total};
}

template <int S>
__device__ __forceinline__ void cascaded_decode_partition(
    const uint8_t* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes_arr,
    const size_t* __restrict__ out_caps, uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ actual_bytes,
    hipcompStatus_t* __restrict__ statuses, const size_t part, uint8_t* const smem, const int lane);

__device__ __forceinline__ void cascaded_decode_partition4(
    const uint8_t* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes_arr,
    const size_t* __restrict__ out_caps, uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ actual_bytes,
    hipcompStatus_t* __restrict__ statuses, const size_t part, uint8_t* const smem, const int lane);

// (launch bound: 4 waves per SIMD where LDS lets a CU hold 14 of these one-wave workgroups, 5 for the
// 4-byte fast path, which holds 21)
#ifndef HC_CASC_DEC_OCC
#define HC_CASC_DEC_OCC 5
#endif
template <int S>
__global__ __launch_bounds__(kWave, S == 4 ? HC_CASC_DEC_OCC : 4) void cascaded_decompress_kernel(
    const uint8_t* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes_arr,
    const size_t* __restrict__ out_caps, const size_t batch,
    uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ actual_bytes,
    hipcompStatus_t* __restrict__ statuses)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; // dec_lds_bytes<S>()
  const int lane = lane_id();
  // Each width has its own launch (the LDS a wave needs depends on it); a partition is
  // handled by the launch that matches ITS type byte (the reference dispatches on
  // partition 0 only).  Undecodable headers are reported by the 1-byte launch.
  // A launch is a fixed number of waves (as many as its LDS lets the chip
  // hold); wave w takes partitions w, w + waves, ...  The headers of the next
  // 64 of them are looked at together, one per lane, so that a launch with
  // nothing to do is over after batch / (64 x waves) steps.
  const size_t waves = gridDim.x;
  for (size_t first = blockIdx.x; first < batch; first += kWave * waves) {
    const size_t p = first + (size_t)lane * waves;
    bool mine = false;
    if (p < batch) {
      cgptr comp = to_global(comp_ptrs[p]);
      const bool bad_header = comp == nullptr || comp_bytes_arr[p] < kPartMeta;
      const uint32_t type = bad_header ? 0xFFu : (uint32_t)comp[3];
      mine = (bad_header || type > 7)
                 ? S == 1
                 : ((S == 1 && type <= 1) || (S == 2 && (type == 2 || type == 3))
                    || (S == 4 && (type == 4 || type == 5)) || (S == 8 && (type == 6 || type == 7)));
    }
    for (uint64_t todo = wave_ballot(mine); todo != 0; todo &= todo - 1) {
      const size_t part = first + (size_t)__builtin_ctzll(todo) * waves;
      if constexpr (S == 4)
        cascaded_decode_partition4(comp_ptrs, comp_bytes_arr, out_caps, out_ptrs, actual_bytes, statuses, part, smem, lane);
      else
        cascaded_decode_partition<S>(comp_ptrs, comp_bytes_arr, out_caps, out_ptrs, actual_bytes, statuses, part, smem, lane);
    }
  }
}

// One partition of 4-byte elements, by one wave: the fast path.
__device__ __forceinline__ void cascaded_decode_partition4(
    const uint8_t* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes_arr,
    const size_t* __restrict__ out_caps, uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ actual_bytes,
    hipcompStatus_t* __restrict__ statuses, const size_t part, uint8_t* const smem, const int lane)
{
  constexpr int S = 4;
  constexpr uint32_t CE = 1024;
  cgptr comp = to_global(uniform_ptr(comp_ptrs[part]));
  const size_t comp_bytes64 = uniform((uint64_t)comp_bytes_arr[part]);
  const bool bad_header = comp == nullptr || comp_bytes64 < kPartMeta;
  uint32_t type = 0xFFu, byte2 = 0;
  if (!bad_header) {
    type = uniform((uint32_t)comp[3]);
    byte2 = uniform((uint32_t)comp[2]);
  }
  const bool undecodable = bad_header || type > 7 || (byte2 >> 4) != 0;
  auto finish = [&](bool ok, uint32_t bytes) {
    if (lane == 0) {
      actual_bytes[part] = ok ? bytes : 0;
      statuses[part] = ok ? hipcompSuccess : hipcompErrorCannotDecompress;
    }
  };
  if (undecodable) {
    finish(false, 0);
    return;
  }
  const uint32_t comp_bytes = (uint32_t)comp_bytes64;
  const uint32_t hdr = uniform(*reinterpret_cast<const HC_GLOBAL uint32_t*>(comp));
  const int R = (int)(hdr & 0xFFu), D = (int)((hdr >> 8) & 0xFFu), bp = (int)((hdr >> 16) & 0x0Fu);
  const uint32_t ub = uniform(*reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + 4));
  const uint32_t N = ub / S;
  gptr out = to_global(uniform_ptr(out_ptrs[part]));
  if (uniform((uint64_t)out_caps[part]) < (size_t)N * S) { // reference :1214-1223
    finish(false, 0);
    return;
  }
  if (R == 0 && D == 0 && bp == 0) { // raw partition (reference :1225-1254)
    if (comp_bytes < kPartMeta + N * S) {
      finish(false, 0);
      return;
    }
    wave_copy(out, comp + kPartMeta, N * S, lane);
    finish(true, N * S);
    return;
  }
  const uint32_t msz = chunk_metadata_size<S>(R, D);
  if (R > 7 || msz > 64) {
    finish(false, 0);
    return;
  }
  uint32_t* const stage = reinterpret_cast<uint32_t*>(smem + kDec4Stage);
  uint8_t* const X = smem + kDec4X;
  uint16_t* const marks = reinterpret_cast<uint16_t*>(smem + kDec4Marks);

  const uint32_t end_w = comp_bytes / 4;
  const uint32_t dh_word = (uint32_t)(R + 2); // the delta heads follow the R + 2 size words (4-byte elements: no padding)
  const int layers = R > D ? R : D;
  uint32_t pos = kPartMeta, done = 0;
  bool ok = true;
  uint32_t pf[kStagePerLane];
  auto prefetch = [&](uint32_t p) {
    const HC_GLOBAL uint32_t* w = reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + p);
    const uint32_t avail = end_w - p / 4;
#pragma unroll
    for (int k = 0; k < (int)kStagePerLane; ++k) {
      const uint32_t idx = (uint32_t)lane + (uint32_t)k * kWave;
      pf[k] = w[min(idx, avail - 1)];
    }
  };
  if (pos / 4 < end_w)
    prefetch(pos);
  while (pos / 4 < end_w) { // reference :1268
    if ((pos + msz) / 4 > end_w) {
      ok = false;
      break;
    }
#pragma unroll
    for (int k = 0; k < (int)kStagePerLane; ++k)
      stage[lane + k * kWave] = pf[k];
    // the chunk metadata (the first 16 words of the staged image at most) in a register, lane j = word j:
    // the staged image may be given up for an array that lies outside it (below)
    const uint32_t img = pf[0];
    auto meta_at = [&](uint32_t j) -> uint32_t { return read_lane(img, (int)j); };
    const uint32_t csz = meta_at(0);
    if (csz < 4 || csz > comp_bytes - pos) { // (see the generic decoder)
      ok = false;
      break;
    }
    const uint32_t next_pos = pos + (csz / 4) * 4; // reference :1412-1413
    if (next_pos / 4 < end_w)
      prefetch(next_pos);
    // array offsets inside the chunk (reference :1291-1305)
    uint32_t offs_final = 0;
    for (int i = 0; i < R; ++i)
      offs_final = ru(offs_final + meta_at((uint32_t)i + 1u), 4u);
    bool staged = true; // the staged image still holds the head of the sub-chunk
    // bounds of an array at byte offset `rel` of the sub-chunk (reference block_read :712-713)
    auto inside = [&](uint32_t rel, uint32_t nbytes) { return array_inside(pos, rel, nbytes, end_w); };
    // `count` words from HBM to LDS
    auto fetch = [&](uint32_t* dst, uint32_t rel, uint32_t count) {
      const HC_GLOBAL uint32_t* g = reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + pos + rel);
      for (uint32_t i = (uint32_t)lane; i < count; i += kWave)
        dst[i] = g[i];
      lds_lane_exchange_fence();
    };
    // ---- the final array -> X
    int n = -1;
    {
      const uint32_t rel = msz + offs_final, nbytes = meta_at(1u + (uint32_t)R);
      if (inside(rel, nbytes)) {
        if (array_staged(rel, nbytes, kStageWords * 4)) {
          n = unpack_values4((const HC_LDS uint32_t*)(stage + rel / 4), nbytes, bp, (HC_LDS uint8_t*)X, lane);
        } else {
          // outside the staged image: its words (1026 at most are looked at) into X itself, unpacked in
          // place -- a lane's loads come before its stores, and so do the wave's
          fetch(reinterpret_cast<uint32_t*>(X), rel, min((nbytes + 3u) / 4u, 1026u));
          n = unpack_values4((const HC_LDS uint32_t*)X, nbytes, bp, (HC_LDS uint8_t*)X, lane);
        }
        if (n == -3) // a bit width of 17 .. 31
          n = unpack_values_wide4(reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + pos + rel), nbytes, bp, X, lane);
      }
    }
    if (n < 0) {
      ok = false;
      break;
    }
    lds_lane_exchange_fence();
    bool stored = false;
    int fused_delta = -1;
    for (int l = layers - 1; l >= 0 && ok; --l) {
      if (l < D && l != fused_delta) { // reference block_delta_decompress :343-377, standalone
        if ((uint32_t)n + 1 > CE) {
          ok = false;
          break;
        }
        // x[0] = head, x[i + 1] = x[i] + old x[i], in place: every lane has its 16 elements in registers
        // before any lane stores (the sums move one element on: the last one of the lane below comes by DPP)
        const uint32_t carry = meta_at(dh_word + (uint32_t)l);
        {
          const uint32_t pos0 = (uint32_t)lane * 16u;
          uint32_t v[16];
          uint32_t* px = reinterpret_cast<uint32_t*>(X + x4_addr(pos0));
#pragma unroll
          for (int k = 0; k < 16; ++k)
            v[k] = px[k];
          lds_lane_exchange_fence();
#pragma unroll
          for (int k = 1; k < 16; ++k)
            v[k] += v[k - 1];
          const uint32_t incl = wave_scan_add_u32(v[15]);
          const uint32_t base = carry + incl - v[15];
          const uint32_t first = from_lane_below(base + v[15], carry, lane);
          if (pos0 <= (uint32_t)n) {
            px[0] = first;
#pragma unroll
            for (int k = 1; k < 16; ++k)
              px[k] = base + v[k - 1];
          }
        }
        lds_lane_exchange_fence();
        ++n;
      }
      if (l < R) { // reference block_rle_decompress :255-305
        uint32_t o = 0;
        for (int i = 0; i < l; ++i)
          o = ru(o + meta_at((uint32_t)i + 1u), 4u);
        {
          u32x4 z = {0, 0, 0, 0};
          u32x4* mz = reinterpret_cast<u32x4*>(marks);
          mz[lane] = z;
          mz[lane + kWave] = z;
        }
        lds_lane_exchange_fence();
        uint32_t total = 0;
        {
          const uint32_t rel = msz + o, nbytes = meta_at((uint32_t)l + 1u);
          int m = -1;
          if (inside(rel, nbytes)) {
            // the array's words: in the staged image, or staged now in its place if they fit it
            const uint32_t need = ru(nbytes, 4);
            uint32_t at = ~0u; // word of `stage` the array begins at
            if (staged && array_staged(rel, nbytes, kStageWords * 4)) {
              at = rel / 4;
            } else if (need <= kStageWords * 4) {
              fetch(stage, rel, need / 4);
              staged = false;
              at = 0;
            }
            // the element-per-lane form (rounds 1-3), for what the blocked one does not take
            uint32_t carry = 0;
            bool too_long = false;
            auto run_starts = [&](auto words) {
              ArrayReader<uint16_t, decltype(words)> lengths;
              const int mm = lengths.open(words, nbytes, bp, CE);
              if (mm < 0)
                return mm;
              for (uint32_t b0 = 0; b0 < (uint32_t)mm; b0 += kWave) {
                const uint32_t i = b0 + (uint32_t)lane;
                const uint32_t cv = i < (uint32_t)mm ? (uint32_t)lengths.get(i) : 0u;
                const uint32_t incl = wave_scan_add_u32(cv);
                const uint32_t start = carry + incl - cv;
                carry += read_lane(incl, 63);
                if (i < (uint32_t)mm && start < CE)
                  marks[start] = (uint16_t)(i + (i >> 5) + 1u); // (the padded index of the run's value, plus one)
                too_long = too_long || carry > CE;
              }
              return mm;
            };
            if (at != ~0u) {
              const Runs4 runs = mark_run_starts4((const HC_LDS uint32_t*)(stage + at), nbytes, bp, (HC_LDS uint16_t*)marks, lane);
              m = runs.n;
              total = runs.total;
              if (m == -3) { // a frame of reference that could wrap a 16-bit length, a width of 13 .. 16 bits
                m = run_starts(static_cast<const uint32_t*>(stage + at));
                total = carry;
                if (too_long)
                  m = -2;
              }
            } else { // an array of run lengths larger than the staged image (1 KiB): from HBM
              m = run_starts(reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + pos + rel));
              total = carry;
              if (too_long)
                m = -2;
            }
          }
          if (m < 0 || m != n) {
            ok = false;
            break;
          }
        }
        lds_lane_exchange_fence();
        const bool with_delta = l >= 1 && (l - 1) < D; // next op: delta of layer l-1
        // (the last expansion of a sub-chunk could store straight to HBM, and did until the stores were
        // looked at: a lane's 16 consecutive elements are four 16-byte stores that each touch 64 different
        // lines, 256 partial-line writes per sub-chunk -- through the element buffer and out again
        // lane-per-element they are 4 fully coalesced ones)
        const bool to_hbm = false;
        if (with_delta && total + 1 > CE) {
          ok = false;
          break;
        }
        if (to_hbm && done + total > N) { // reference :1395-1402
          ok = false;
          break;
        }
        uint32_t head = 0;
        if (with_delta)
          head = meta_at(dh_word + (uint32_t)(l - 1));
        gptr gdst = out + (size_t)done * S;
        // the expansion: one round of 64 lanes x 16 consecutive output elements (see the generic decoder),
        // in place: the values of the runs are read (all lanes) before anything is stored
        {
          const uint32_t pos0 = (uint32_t)lane * 16u;
          uint32_t mk[16];
          {
            const u32x4* mp = reinterpret_cast<const u32x4*>(marks + pos0);
            const u32x4 t0 = mp[0], t1 = mp[1];
            const uint32_t w[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int k = 0; k < 16; ++k)
              mk[k] = (k & 1) ? w[k / 2] >> 16 : w[k / 2] & 0xFFFFu;
          }
          // (behind `total` there is nothing but the markers of runs behind the array's last one: whatever
          // they make of the elements behind `total` is not stored, and every marker is an index inside X)
          uint32_t r[16];
          r[0] = mk[0];
#pragma unroll
          for (int k = 1; k < 16; ++k)
            r[k] = max(r[k - 1], mk[k]);
          const uint32_t upto = wave_scan_max_u32(r[15]);
          const uint32_t below = max(1u, dpp_u32<0x138, 0xF>(upto)); // (>= 1: the lanes behind `total` read x[0])
          uint32_t v[16];
#pragma unroll
          for (int k = 0; k < 16; ++k)
            v[k] = *reinterpret_cast<const uint32_t*>(X + 4u * (max(r[k], below) - 1u));
          lds_lane_exchange_fence();
          if (with_delta) {
#pragma unroll
            for (int k = 1; k < 16; ++k)
              v[k] += v[k - 1];
            const uint32_t incl = wave_scan_add_u32(v[15]);
            const uint32_t base = head + incl - v[15];
            const uint32_t first = from_lane_below(base + v[15], head, lane);
            if (pos0 <= total) {
              uint32_t* py = reinterpret_cast<uint32_t*>(X + x4_addr(pos0));
              py[0] = first;
#pragma unroll
              for (int k = 1; k < 16; ++k)
                py[k] = base + v[k - 1];
            }
          } else if (to_hbm) {
            if (pos0 + 16u <= total) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                u32x4 t = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
                *reinterpret_cast<HC_GLOBAL u32x4_unaligned*>(gdst + (size_t)pos0 * S + 16u * i) = t;
              }
            } else if (pos0 < total) {
#pragma unroll
              for (int k = 0; k < 16; ++k)
                if (pos0 + (uint32_t)k < total)
                  reinterpret_cast<HC_GLOBAL uint32_t*>(gdst)[pos0 + k] = v[k];
            }
          } else {
            if (pos0 < total) {
              uint32_t* py = reinterpret_cast<uint32_t*>(X + x4_addr(pos0));
#pragma unroll
              for (int k = 0; k < 16; ++k)
                py[k] = v[k];
            }
          }
        }
        lds_lane_exchange_fence();
        if (with_delta) {
          fused_delta = l - 1;
          n = (int)total + 1;
        } else {
          n = (int)total;
        }
        stored = to_hbm;
      }
    }
    if (!ok)
      break;
    if (!stored) {
      if (done + (uint32_t)n > N) { // reference :1395-1402
        ok = false;
        break;
      }
      // sub-chunk -> output (outputs are 4-byte aligned: cascaded.h:178-193): four consecutive elements
      // per lane and step (they lie in one block of the padded layout), 1 KiB per store instruction
      HC_GLOBAL uint32_t* dst = reinterpret_cast<HC_GLOBAL uint32_t*>(out + (size_t)done * S);
#ifndef HC_CASC_DEC_LATE_WAIT
      // The words of the next sub-chunk, asked for at the top of this one, are here by now: have them
      // waited for HERE, in front of this sub-chunk's stores.  Vector memory operations are counted in
      // order, and a wait at the top of the next sub-chunk -- behind the stores -- is a wait for the
      // stores to reach memory as well.
#pragma unroll
      for (int k = 0; k < (int)kStagePerLane; ++k)
        asm volatile("" : "+v"(pf[k]));
#endif
      if (n == (int)CE) { // a full sub-chunk (all but a partition's last one): no tests, one address, immediate steps
        const uint8_t* px0 = X + x4_addr(4u * (uint32_t)lane);
        HC_GLOBAL uint32_t* d0 = dst + 4u * (uint32_t)lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) { // (x4_addr(e + 256 q) = x4_addr(e) + 4 (256 + 8) q for e < 256)
          const uint32_t* px = reinterpret_cast<const uint32_t*>(px0 + 1056 * q);
          u32x4 t = {px[0], px[1], px[2], px[3]};
          *reinterpret_cast<HC_GLOBAL u32x4_unaligned*>(d0 + 256 * q) = t;
        }
      } else
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t e = 4u * ((uint32_t)lane + 64u * q);
        if (256u * q < (uint32_t)n) { // (wave-uniform)
          const uint32_t* px = reinterpret_cast<const uint32_t*>(X + x4_addr(e));
          if (e + 4u <= (uint32_t)n) {
            u32x4 t = {px[0], px[1], px[2], px[3]};
            *reinterpret_cast<HC_GLOBAL u32x4_unaligned*>(dst + e) = t;
          } else {
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (e + (uint32_t)k < (uint32_t)n)
                dst[e + k] = px[k];
          }
        }
      }
    }
    done += (uint32_t)n;
    pos = next_pos;
  }
  if (done != N)
    ok = false;
  finish(ok, N * S);
}

// One partition, by one wave (the launch that owns its width).
template <int S>
__device__ __forceinline__ void cascaded_decode_partition(
    const uint8_t* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes_arr,
    const size_t* __restrict__ out_caps, uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ actual_bytes,
    hipcompStatus_t* __restrict__ statuses, const size_t part, uint8_t* const smem, const int lane)
{
  typedef typename UIntOf<S>::type UT;
  constexpr int E = Blocked<S>::E;
  cgptr comp = to_global(uniform_ptr(comp_ptrs[part]));
  const size_t comp_bytes64 = uniform((uint64_t)comp_bytes_arr[part]);
  const bool bad_header = comp == nullptr || comp_bytes64 < kPartMeta;
  uint32_t type = 0xFFu, byte2 = 0;
  if (!bad_header) {
    type = uniform((uint32_t)comp[3]);
    byte2 = uniform((uint32_t)comp[2]);
  }
  // (byte 2 is use_bp, 0 or 1; a high nibble marked the larger sub-chunks of rounds 2-3, an extension
  // that is gone: such a stream is not decodable)
  const bool undecodable = bad_header || type > 7 || (byte2 >> 4) != 0;
  auto finish = [&](bool ok, uint32_t bytes) {
    if (lane == 0) {
      actual_bytes[part] = ok ? bytes : 0;
      statuses[part] = ok ? hipcompSuccess : hipcompErrorCannotDecompress;
    }
  };
  if (undecodable) {
    finish(false, 0);
    return;
  }
  const uint32_t comp_bytes = (uint32_t)comp_bytes64;
  const uint32_t hdr = uniform(*reinterpret_cast<const HC_GLOBAL uint32_t*>(comp));
  const int R = (int)(hdr & 0xFFu), D = (int)((hdr >> 8) & 0xFFu), bp = (int)((hdr >> 16) & 0x0Fu);
  const uint32_t ub = uniform(*reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + 4));
  const uint32_t N = ub / S;
  gptr out = to_global(uniform_ptr(out_ptrs[part]));
  if (uniform((uint64_t)out_caps[part]) < (size_t)N * S) { // reference :1214-1223
    finish(false, 0);
    return;
  }
  if (R == 0 && D == 0 && bp == 0) { // raw partition (reference :1225-1254)
    if (comp_bytes < ru(kPartMeta, S) + N * S) {
      finish(false, 0);
      return;
    }
    wave_copy(out, comp + ru(kPartMeta, S), N * S, lane);
    finish(true, N * S);
    return;
  }
  const uint32_t msz = chunk_metadata_size<S>(R, D);
  if (R > 7 || msz > 64) {
    finish(false, 0);
    return;
  }
  uint8_t* my = smem;
  UT* bufA = reinterpret_cast<UT*>(my);
  UT* bufB = reinterpret_cast<UT*>(my + dec_buf_bytes());
  uint16_t* marks = reinterpret_cast<uint16_t*>(my + 2 * dec_buf_bytes());
  uint32_t* stage = reinterpret_cast<uint32_t*>(my + 2 * dec_buf_bytes() + (CB / S) * 2);
  const uint32_t* meta = stage; // the chunk metadata is the head of the staged image

  constexpr uint32_t CE = CB / S;
  const uint32_t end_w = comp_bytes / 4;
  const uint32_t dh_off = ru((uint32_t)(4 + 4 * (R + 1)), (uint32_t)S);
  const int layers = R > D ? R : D;
  uint32_t pos = ru(kPartMeta, S), done = 0;
  bool ok = true;
  // kStageWords words of the sub-chunk at `p`, kStagePerLane per lane, clipped to the
  // partition (words past the end read as 0); issued one sub-chunk ahead
  uint32_t pf[kStagePerLane];
  auto prefetch = [&](uint32_t p) {
    const HC_GLOBAL uint32_t* w = reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + p);
    const uint32_t avail = end_w - p / 4;
#pragma unroll
    for (int k = 0; k < (int)kStagePerLane; ++k) {
      const uint32_t idx = (uint32_t)lane + (uint32_t)k * kWave;
      pf[k] = w[min(idx, avail - 1)];
    }
  };
  if (pos / 4 < end_w)
    prefetch(pos);
  while (pos / 4 < end_w) { // reference :1268
    if ((pos + msz) / 4 > end_w) {
      ok = false;
      break;
    }
#pragma unroll
    for (int k = 0; k < (int)kStagePerLane; ++k)
      stage[lane + k * kWave] = pf[k];
    const uint32_t csz = uniform(meta[0]);
    // A sub-chunk has to lie inside the partition and to move the cursor on:
    // with csz in 1..3 the reference's cursor (:1412-1413) stands still, and a
    // sub-chunk that also decodes to zero elements would then be read forever;
    // a huge csz wraps the 32-bit cursor backwards.
    if (csz < 4 || csz > comp_bytes - pos) {
      ok = false;
      break;
    }
    const uint32_t next_pos = ru(pos + (csz / 4) * 4, S); // reference :1412-1413; <= comp_bytes + S: no wrap
    if (next_pos / 4 < end_w)
      prefetch(next_pos);
    // array offsets inside the chunk (reference :1291-1305)
    uint32_t offs_final = 0;
    {
      uint32_t o = 0;
      for (int i = 0; i < R; ++i) {
        const uint32_t sz = uniform(meta[i + 1]);
        o = ru(o + sz, (i == R - 1) ? (S > 4 ? (uint32_t)S : 4u) : 4u);
      }
      offs_final = o;
    }
    UT* x = bufA;
    UT* y = bufB;
    int n = wave_read_array<UT, kStageWords>(comp, end_w, pos, msz + offs_final, uniform(meta[1 + R]), bp, stage, x, CE, lane);
    if (n < 0) {
      ok = false;
      break;
    }
    // Layers undone in the exact reverse of the encoder (see the oracle for
    // the case num_deltas > num_RLEs where the reference's order is wrong).
    // Two fusions keep the passes over LDS short: an RLE expansion whose
    // output feeds a delta layer does the prefix sum in the same pass, and the
    // last expansion of a sub-chunk stores straight to HBM.
    bool stored = false;      // sub-chunk already written to the output
    int fused_delta = -1;     // delta layer already undone by a fused expansion
    for (int l = layers - 1; l >= 0 && ok; --l) {
      if (l < D && l != fused_delta) { // reference block_delta_decompress :343-377
        if ((uint32_t)n + 1 > CE) {
          ok = false;
          break;
        }
        // y[0] = head, y[i + 1] = y[i] + x[i]: n + 1 elements (blocked: prefix_store_shifted)
        UT carry = *reinterpret_cast<const UT*>(reinterpret_cast<const uint8_t*>(meta) + dh_off + l * S);
        for (uint32_t b0 = 0; b0 <= (uint32_t)n; b0 += Blocked<S>::ROUND) {
          const uint32_t pos0 = b0 + (uint32_t)lane * E;
          UT v[E];
          load_block(x + min(pos0, CE - (uint32_t)E), v); // (what lies at and behind n takes no part in what is used)
          carry = prefix_store_shifted<UT>(v, carry, y, pos0, pos0 <= (uint32_t)n, lane);
        }
        lds_lane_exchange_fence();
        UT* t = x; x = y; y = t;
        ++n;
      }
      if (l < R) { // reference block_rle_decompress :255-305
        uint32_t o = 0;
        for (int i = 0; i < l; ++i)
          o = ru(o + uniform(meta[i + 1]), 4u);
        // Each run drops its index+1 at its start position (exclusive prefix
        // of the lengths) in `marks`; a running max over the positions then
        // names the run of every output element.  The lengths are used as they
        // come out of the packed array (reference block_read :702-737 +
        // block_bitunpack :563-618; bounds as :712-713).
        {
          u32x4 z = {0, 0, 0, 0};
          u32x4* mz = reinterpret_cast<u32x4*>(marks);
          for (uint32_t k = (uint32_t)lane; k < CE * 2 / 16; k += kWave)
            mz[k] = z;
        }
        uint32_t carry = 0;
        bool too_long = false;
        auto run_starts = [&](auto words) {
          ArrayReader<uint16_t, decltype(words)> lengths;
          const int m = lengths.open(words, uniform(meta[l + 1]), bp, CE);
          if (m < 0 || m != n)
            return false;
          for (uint32_t b0 = 0; b0 < (uint32_t)n; b0 += kWave) {
            const uint32_t i = b0 + (uint32_t)lane;
            const uint32_t cv = i < (uint32_t)n ? (uint32_t)lengths.get(i) : 0u;
            const uint32_t incl = wave_scan_add_u32(cv);
            const uint32_t start = carry + incl - cv;
            carry += read_lane(incl, 63);
            if (i < (uint32_t)n && start < CE)
              marks[start] = (uint16_t)(i + 1);
            too_long = too_long || carry > CE;
          }
          return true;
        };
        {
          const uint32_t rel = msz + o, nbytes = uniform(meta[l + 1]);
          const uint32_t off = pos + rel;
          bool good = array_inside(pos, rel, nbytes, end_w);
          if (good) {
            if (array_staged(rel, nbytes, kStageWords * 4))
              good = run_starts(static_cast<const uint32_t*>(stage + rel / 4));
            else
              good = run_starts(reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + off));
          }
          if (!good) {
            ok = false;
            break;
          }
        }
        if (too_long) {
          ok = false;
          break;
        }
        const uint32_t total = carry;
        const bool with_delta = l >= 1 && (l - 1) < D; // next op: delta of layer l-1
        const bool to_hbm = l == 0 && S >= 4;          // last op of the sub-chunk
        if (with_delta && total + 1 > CE) {
          ok = false;
          break;
        }
        if (to_hbm && done + total > N) { // reference :1395-1402
          ok = false;
          break;
        }
        UT head = 0;
        if (with_delta)
          head = *reinterpret_cast<const UT*>(reinterpret_cast<const uint8_t*>(meta) + dh_off + (l - 1) * S);
        gptr gdst = out + (size_t)done * S;
        // The expansion, blocked: a lane takes E consecutive output elements -- their markers (one or
        // two 16-byte reads), the running maximum inside the lane on top of the maximum of the lanes
        // below (one wave scan), the values of those runs (E reads), and then either the prefix sum of
        // the delta layer that follows (prefix_store_shifted) or the block as it is, to LDS or to HBM.
        // (n >= 1 here whenever total > 0: run 1 starts at element 0, so the maximum is never 0 inside `total`.)
        uint32_t run_carry = 1; // (>= 1: the lanes behind `total` then read x[0], not x[-1])
        UT sum_carry = head;
        const uint32_t reach = with_delta ? total + 1 : total; // elements of y that are wanted
        for (uint32_t b0 = 0; b0 < reach; b0 += Blocked<S>::ROUND) {
          const uint32_t pos0 = b0 + (uint32_t)lane * E;
          uint32_t mk[E];
          {
            const u32x4* mp = reinterpret_cast<const u32x4*>(marks + min(pos0, CE - (uint32_t)E));
            uint32_t w[E / 2];
#pragma unroll
            for (int i = 0; i < E / 8; ++i) {
              const u32x4 t = mp[i];
              w[4 * i] = t.x;
              w[4 * i + 1] = t.y;
              w[4 * i + 2] = t.z;
              w[4 * i + 3] = t.w;
            }
#pragma unroll
            for (int k = 0; k < E; ++k)
              mk[k] = (k & 1) ? w[(k / 2) % (E / 2)] >> 16 : w[(k / 2) % (E / 2)] & 0xFFFFu;
          }
          // (a lane whose block begins at or behind `total` has nothing there: its markers do not count)
          const bool has = pos0 < total;
          uint32_t r[E];
          r[0] = has ? mk[0] : 0u;
#pragma unroll
          for (int k = 1; k < E; ++k)
            r[k] = max(r[k - 1], has ? mk[k] : 0u);
          const uint32_t upto = wave_scan_max_u32(r[E - 1]);
          const uint32_t below = max(run_carry, dpp_u32<0x138, 0xF>(upto)); // the run in force at my first element
          run_carry = max(run_carry, read_lane(upto, 63));
          UT v[E];
#pragma unroll
          for (int k = 0; k < E; ++k)
            v[k] = x[max(r[k], below) - 1u];
          if (with_delta) {
            sum_carry = prefix_store_shifted<UT>(v, sum_carry, y, pos0, pos0 <= total, lane);
          } else if (to_hbm) {
            if (pos0 + (uint32_t)E <= total) {
              uint32_t w[Blocked<S>::W];
#pragma unroll
              for (int i = 0; i < Blocked<S>::W; ++i)
                w[i] = S == 8 ? (uint32_t)((uint64_t)v[(i / 2) % E] >> (32 * (i & 1))) : (uint32_t)v[i % E];
#pragma unroll
              for (int i = 0; i < Blocked<S>::W / 4; ++i) {
                u32x4 t = {w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
                *reinterpret_cast<HC_GLOBAL u32x4_unaligned*>(gdst + (size_t)pos0 * S + 16u * i) = t;
              }
            } else if (pos0 < total) {
#pragma unroll
              for (int k = 0; k < E; ++k)
                if (pos0 + (uint32_t)k < total)
                  reinterpret_cast<HC_GLOBAL UT*>(gdst)[pos0 + k] = v[k];
            }
          } else {
            if (pos0 < total)
              store_block(y + pos0, v);
          }
        }
        lds_lane_exchange_fence();
        if (with_delta) {
          fused_delta = l - 1;
          n = (int)total + 1;
        } else {
          n = (int)total;
        }
        stored = to_hbm;
        UT* t = x; x = y; y = t;
      }
    }
    if (!ok)
      break;
    if (!stored) {
      if (done + (uint32_t)n > N) { // reference :1395-1402
        ok = false;
        break;
      }
      // sub-chunk -> output
      gptr dst = out + (size_t)done * S;
      const uint32_t nb = (uint32_t)n * S;
      const uint8_t* srcb = reinterpret_cast<const uint8_t*>(x);
      if ((reinterpret_cast<uintptr_t>(dst) & 3u) == 0) {
        const uint32_t words = nb / 4;
        for (uint32_t w = (uint32_t)lane; w < words; w += kWave)
          reinterpret_cast<HC_GLOBAL uint32_t*>(dst)[w] = reinterpret_cast<const uint32_t*>(srcb)[w];
        for (uint32_t k = words * 4 + (uint32_t)lane; k < nb; k += kWave)
          dst[k] = srcb[k];
      } else {
        for (uint32_t k = (uint32_t)lane; k < nb; k += kWave)
          dst[k] = srcb[k];
      }
    }
    done += (uint32_t)n;
    pos = next_pos;
  }
  if (done != N)
    ok = false;
  finish(ok, N * S);
}

__global__ __launch_bounds__(256) void cascaded_get_sizes_kernel(
    const uint8_t* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes, size_t* __restrict__ out_sizes,
    size_t batch)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch)
    return;
  size_t n = 0;
  if (comp_bytes[i] >= kPartMeta) // reference CascadedBatch.hip:262-281
    n = *reinterpret_cast<const HC_GLOBAL uint32_t*>(to_global(comp_ptrs[i]) + 4);
  out_sizes[i] = n;
}

} // namespace

namespace {

typedef void (*CompressKernel)(
    const uint8_t* const*, const size_t*, uint8_t* const*, size_t*, size_t, int, int, int, int);
typedef void (*DecompressKernel)(
    const uint8_t* const*, const size_t*, const size_t*, size_t, uint8_t* const*, size_t*, hipcompStatus_t*);

template <int S>
hipError_t launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes, const size_t* out_caps, size_t batch,
    uint8_t* const* out_ptrs, size_t* actual_bytes, hipcompStatus_t* statuses, hipStream_t stream)
{
  DecompressKernel k = cascaded_decompress_kernel<S>;
  constexpr uint32_t lds = dec_lds_bytes<S>();
  static_assert(lds <= 64 * 1024, "fits the default dynamic LDS limit: no attribute to raise");
  // As many one-wave workgroups as the chip holds at once -- LDS (handed out in 1280-byte granules) or
  // registers, whichever is short first: every wave takes the same number of partitions, so a
  // workgroup that has to wait for a slot makes the launch twice as long (met in round 4: 21 by LDS,
  // 20 by registers, 5376 workgroups launched, 12.8 waves per CU at work on average).
  static std::atomic<int> g_per_cu[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  int cached = dev >= 0 && dev < 16 ? g_per_cu[dev].load(std::memory_order_acquire) : 0;
  if (cached == 0) {
    int by_api = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&by_api, reinterpret_cast<const void*>(k), kWave, lds) != hipSuccess || by_api < 1)
      by_api = 1;
    const int by_lds = (int)((160u * 1024u) / ((lds + 1279u) / 1280u * 1280u));
    cached = by_api < by_lds ? by_api : by_lds;
    if (cached > 32)
      cached = 32;
    if (dev >= 0 && dev < 16)
      g_per_cu[dev].store(cached, std::memory_order_release);
  }
  const uint32_t per_cu = (uint32_t)cached;
  const size_t resident = (size_t)num_cus_of_current_device() * per_cu;
  k<<<dim3((unsigned)(batch < resident ? batch : resident)), dim3(kWave), lds, stream>>>(
      comp_ptrs, comp_bytes, out_caps, batch, out_ptrs, actual_bytes, statuses);
  return hipGetLastError();
}

} // namespace

void cascaded_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, size_t batch, int type_tag,
    int elem_size, int num_rles, int num_deltas, int use_bp, hipStream_t stream)
{
  const dim3 grid((unsigned)((batch + kWavesPerBlock - 1) / kWavesPerBlock));
  const dim3 block(kWave * kWavesPerBlock);
  CompressKernel k = elem_size == 1   ? cascaded_compress_kernel<1>
                     : elem_size == 2 ? cascaded_compress_kernel<2>
                     : elem_size == 4 ? cascaded_compress_kernel<4>
                                      : cascaded_compress_kernel<8>;
  const uint32_t lds = kWavesPerBlock
                       * (elem_size == 1   ? wave_lds_bytes<1>()
                          : elem_size == 2 ? wave_lds_bytes<2>()
                          : elem_size == 4 ? wave_lds_bytes<4>()
                                           : wave_lds_bytes<8>());
  k<<<grid, block, lds, stream>>>(in_ptrs, in_bytes, out_ptrs, out_bytes, batch, type_tag, num_rles, num_deltas, use_bp);
}

namespace {
typedef void (*PlacedKernel)(const uint8_t* const*, const size_t*, size_t*, uint32_t, int, int, int, int, uint32_t*, Placement);
struct PlacedShape { PlacedKernel k; uint32_t lds; int slot; };
PlacedShape placed_shape(int elem_size)
{
  switch (elem_size) {
  case 1: return {cascaded_compress_placed_kernel<1>, wave_lds_bytes<1>(), 0};
  case 2: return {cascaded_compress_placed_kernel<2>, wave_lds_bytes<2>(), 1};
  case 4: return {cascaded_compress_placed_kernel<4>, wave_lds_bytes<4>(), 2};
  default: return {cascaded_compress_placed_kernel<8>, wave_lds_bytes<8>(), 3};
  }
}
// workgroups of the placed kernel the device holds at once (0: could not be found out)
unsigned placed_resident(int elem_size)
{
  static std::atomic<unsigned> known[4]; // (the same on every device of the process)
  const PlacedShape sh = placed_shape(elem_size);
  unsigned r = known[sh.slot].load(std::memory_order_relaxed);
  if (r == 0) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sh.k, kWave, sh.lds) != hipSuccess || per_cu <= 0) {
      (void)hipGetLastError();
      return 0;
    }
    r = (unsigned)per_cu * (unsigned)num_cus_of_current_device();
    known[sh.slot].store(r, std::memory_order_relaxed);
  }
  return r;
}
} // namespace

size_t cascaded_placement_slots(int elem_size)
{
  return placed_resident(elem_size);
}

hipError_t cascaded_launch_compress_placed(
    const uint8_t* const* in_ptrs, const size_t* in_bytes, size_t* out_bytes, size_t batch, int type_tag,
    int elem_size, int num_rles, int num_deltas, int use_bp, uint32_t* ticket, const Placement& place,
    hipStream_t stream)
{
  const unsigned resident = placed_resident(elem_size);
  if (resident == 0 || batch == 0 || batch >= 0xFFFFFFFFull)
    return hipErrorInvalidValue;
  const PlacedShape sh = placed_shape(elem_size);
  sh.k<<<dim3(batch < resident ? (unsigned)batch : resident), dim3(kWave), sh.lds, stream>>>(
      in_ptrs, in_bytes, out_bytes, (uint32_t)batch, type_tag, num_rles, num_deltas, use_bp, ticket, place);
  return hipGetLastError();
}

// One launch per element width, as the reference (CascadedBatch.hip:387-429) -- but a launch
// leaves the partitions of the other widths alone instead of decoding everything with the
// width of partition 0.  The common width first.
hipError_t cascaded_launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    const size_t* out_caps, size_t batch, uint8_t* const* out_ptrs,
    size_t* actual_bytes, hipcompStatus_t* statuses, hipStream_t stream)
{
#define HC_DEC(S)                                                                                              \
  {                                                                                                            \
    const hipError_t e = launch_decompress<S>(comp_ptrs, comp_bytes, out_caps, batch, out_ptrs, actual_bytes, \
                                              statuses, stream);                                               \
    if (e != hipSuccess)                                                                                       \
      return e;                                                                                                \
  }
  HC_DEC(4) HC_DEC(8) HC_DEC(2) HC_DEC(1)
#undef HC_DEC
  return hipSuccess;
}

void cascaded_launch_get_sizes(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    size_t* out_sizes, size_t batch, hipStream_t stream)
{
  cascaded_get_sizes_kernel<<<dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, stream>>>(
      comp_ptrs, comp_bytes, out_sizes, batch);
}


// ---- hipcomp/cascaded_select.h: the sample of a batch, and the totals per candidate ----------
namespace {

__global__ __launch_bounds__(64) void cascaded_select_sample_kernel(
    const uint8_t* const* __restrict__ in_ptrs, const size_t* __restrict__ in_bytes, const size_t batch,
    const size_t parts, const size_t clip, const int elem_size, const uint8_t** __restrict__ s_in_ptrs,
    size_t* __restrict__ s_in_bytes, uint8_t** __restrict__ s_out_ptrs, uint8_t* __restrict__ slots, const size_t slot_bytes)
{
  const size_t j = threadIdx.x;
  if (j >= parts)
    return;
  const size_t p = j * batch / parts; // spread evenly over the batch
  size_t n = in_bytes[p];
  n = n < clip ? n : clip;
  n = n / (size_t)elem_size * (size_t)elem_size;
  s_in_ptrs[j] = in_ptrs[p];
  s_in_bytes[j] = in_ptrs[p] ? n : 0;
  s_out_ptrs[j] = slots + j * slot_bytes;
}

__global__ __launch_bounds__(64) void cascaded_select_totals_kernel(
    const size_t* __restrict__ s_in_bytes, const size_t* __restrict__ s_out_bytes, const size_t parts,
    const size_t candidates, const size_t stride, unsigned long long* __restrict__ totals)
{
  const size_t c = threadIdx.x;
  if (c > candidates)
    return;
  unsigned long long sum = 0;
  for (size_t j = 0; j < parts; ++j)
    sum += c < candidates ? s_out_bytes[c * stride + j] : s_in_bytes[j]; // (the last entry: the uncompressed bytes)
  totals[c] = sum;
}

} // namespace

void cascaded_launch_select_sample(
    const uint8_t* const* in_ptrs, const size_t* in_bytes, size_t batch, size_t parts, size_t clip, int elem_size,
    const uint8_t** s_in_ptrs, size_t* s_in_bytes, uint8_t** s_out_ptrs, uint8_t* slots, size_t slot_bytes,
    hipStream_t stream)
{
  cascaded_select_sample_kernel<<<dim3(1), dim3(64), 0, stream>>>(
      in_ptrs, in_bytes, batch, parts, clip, elem_size, s_in_ptrs, s_in_bytes, s_out_ptrs, slots, slot_bytes);
}

void cascaded_launch_select_totals(
    const size_t* s_in_bytes, const size_t* s_out_bytes, size_t parts, size_t candidates, size_t stride,
    unsigned long long* totals, hipStream_t stream)
{
  cascaded_select_totals_kernel<<<dim3(1), dim3(64), 0, stream>>>(s_in_bytes, s_out_bytes, parts, candidates, stride, totals);
}

} // namespace hcamd
