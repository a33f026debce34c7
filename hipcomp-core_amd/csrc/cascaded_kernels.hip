// cascaded_kernels.hip -- gfx950 kernels of the batched Cascaded codec
// (RLE -> Delta -> BitPack fused over 4096-byte sub-chunks, all in LDS).
//
// Wire format and layer semantics: reference src/CascadedKernels.hiph:101-1435
// (restated in oracle/cascaded_oracle.c).  Mechanism:
//
//   reference                                 here
//   ----------------------------------------  -------------------------------
//   one 128-thread block per partition,       one WAVE per partition, four
//   6+ __syncthreads per layer, hipCUB        partitions per 256-thread block:
//   BlockScan / BlockReduce                   no block barrier at all; run
//                                             ends ranked with one ballot +
//                                             popcount per 64 elements, min /
//                                             max by DPP-style shuffles
//   RLE: count pass, scan, scatter pass,      one pass: value, rank and run
//   adjacent-difference pass                  length from the same ballot
//   bit-packed arrays built in LDS, then      packed words go straight from
//   copied out word by word                   the LDS element array to HBM
//   decompress: 4 launches, type taken from   4 launches (LDS sized per
//   partition 0, RLE expand = 1 thread/run    width), type taken from EACH
//   (a 4096-long run = 4096 serial stores)    partition; RLE expand = run
//                                             markers + DPP running max;
//                                             delta = DPP prefix sum
//
// Bytes the reference leaves undefined (stale LDS / unwritten gaps,
// SURVEY.md App. C.4) are written as 0 here, so the output is deterministic.

#include "cascaded_launch.hpp"
#include "lz4_launch.hpp" // num_cus_of_current_device
#include "wave_utils.hpp"

#include <type_traits>

namespace hcamd {

namespace {

// Sub-chunk size CB (a template parameter of the kernels): 4096 is the reference's
// (CascadedKernels.hiph:88, hard-wired; its opts.chunk_size is "not currently used",
// cascaded.h:93-100); 8192 and 16384 are honoured here (the 160 KiB
// of LDS permit them).  Streams with the larger sub-chunks say so in the high nibble
// of header byte 2 (use_bp, 0 or 1 in the reference): 0 = 4096, 1 = 8192, 2 = 16384.
constexpr uint32_t kPartMeta = 8;
#ifndef HC_CASC_WAVES
#define HC_CASC_WAVES 1 // 10 KiB of LDS per wave: separate blocks pack 14 per CU, 4-wave blocks only 12
#endif
constexpr int kWavesPerBlock = HC_CASC_WAVES;

template <int S> struct UIntOf;
template <> struct UIntOf<1> { typedef uint8_t type; typedef int8_t stype; };
template <> struct UIntOf<2> { typedef uint16_t type; typedef int16_t stype; };
template <> struct UIntOf<4> { typedef uint32_t type; typedef int32_t stype; };
template <> struct UIntOf<8> { typedef uint64_t type; typedef int64_t stype; };

__device__ __forceinline__ uint32_t ru(uint32_t a, uint32_t b) { return (a + b - 1) / b * b; }

template <int CB>
__host__ __device__ constexpr uint32_t elem_buf_bytes() { return CB + 16; }

template <int S, int CB>
__host__ __device__ constexpr uint32_t wave_lds_bytes()
{
  // encoder: one element buffer (every layer works in place) + run-count
  // array + 64-byte metadata image
  return elem_buf_bytes<CB>() + (CB / S) * 2 + 64;
}

// decoder: the head of the compressed sub-chunk (metadata + arrays) is staged
// in LDS, stage_words<CB>() 32-bit words, stage_per_lane<CB>() per lane, loaded one
// sub-chunk ahead into registers.  Arrays that do not fit are read from HBM
// directly.  1 KiB holds the whole sub-chunk of a column that compresses 4x or
// better and lets a CU hold 14 waves of the 4-byte / 4096 launch (2 KiB: 12;
// measured 960 vs 843 GB/s at ratio 5.3, 770 vs 840 GB/s at ratio 2.2).
// (per 4096 bytes of sub-chunk; the larger sub-chunks stage 2 and 4 KiB, which costs
// their launches no resident wave)
template <int CB>
__host__ __device__ constexpr uint32_t stage_per_lane() { return 4u * (CB / 4096); }
template <int CB>
__host__ __device__ constexpr uint32_t stage_words() { return stage_per_lane<CB>() * kWave; }
template <int S, int CB>
__host__ __device__ constexpr uint32_t dec_lds_bytes()
{
  // two element buffers + run markers + staged sub-chunk
  return 2 * elem_buf_bytes<CB>() + (CB / S) * 2 + stage_words<CB>() * 4;
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
// (reference block_rle_compress :124-241)
template <typename UT>
__device__ __forceinline__ uint32_t wave_rle(
    const UT* in, uint32_t n, UT* vals, uint16_t* cnts, int lane)
{
  uint32_t m = 0, prev_end = 0;
  const uint64_t below_me = (1ull << lane) - 1;
  for (uint32_t base = 0; base < n; base += kWave) {
    const uint32_t i = base + (uint32_t)lane;
    const bool active = i < n;
    // (read whether or not the lane has an element: the buffer holds a multiple
    // of 64 elements and one more, and what lies behind n is not used)
    const UT v = in[i];
    const UT nx = in[i + 1];
    const bool is_end = active && (i + 1 == n || nx != v);
    const uint64_t mask = wave_ballot(is_end);
    const uint64_t below = mask & below_me;
    const uint32_t rank = m + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const uint32_t pe = below ? base + 64u - (uint32_t)__builtin_clzll(below) : prev_end;
    if (is_end) {
      vals[rank] = v;
      cnts[rank] = (uint16_t)(i + 1 - pe);
    }
    m += (uint32_t)__builtin_popcountll(mask);
    if (mask)
      prev_end = base + 64u - (uint32_t)__builtin_clzll(mask);
  }
  return m;
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

template <int S, int CB>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void cascaded_compress_kernel(
    const uint8_t* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes_arr,
    uint8_t* const* __restrict__ out_ptrs,
    size_t* __restrict__ out_bytes_arr, const size_t batch, const int type_tag,
    const int R, const int D, const int bp)
{
  typedef typename UIntOf<S>::type UT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; // kWavesPerBlock * wave_lds_bytes<S, CB>()
  const int lane = lane_id();
  // everything that steers the layers is wave-uniform: say so (see uniform())
  const int wave = (int)uniform((uint32_t)(threadIdx.x >> 6));
  const size_t part = (size_t)blockIdx.x * kWavesPerBlock + wave;
  if (part >= batch)
    return;
  uint8_t* my = smem + wave * wave_lds_bytes<S, CB>();
  UT* bufA = reinterpret_cast<UT*>(my);
  uint16_t* cnts = reinterpret_cast<uint16_t*>(my + elem_buf_bytes<CB>());
  uint32_t* meta = reinterpret_cast<uint32_t*>(my + elem_buf_bytes<CB>() + (CB / S) * 2);

  cgptr in = to_global(uniform_ptr(in_ptrs[part]));
  const size_t in_bytes64 = uniform((uint64_t)in_bytes_arr[part]);
  gptr out = to_global(uniform_ptr(out_ptrs[part]));
  if (in == nullptr || in_bytes64 == 0) { // reference :856-860
    if (lane == 0)
      out_bytes_arr[part] = 0;
    return;
  }
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

  for (uint32_t c = 0; c < nchunks && use; ++c) {
    const uint32_t chunk_start = cur;
    cur += msz;
    uint32_t n = min(N - c * CE, CE);
    // sub-chunk -> LDS (16 bytes per lane per step; inputs are 4-byte aligned)
    {
      cgptr src = in + (size_t)c * CB;
      const uint32_t nb = n * S;
      uint32_t* dstw = reinterpret_cast<uint32_t*>(bufA);
      for (uint32_t o = (uint32_t)lane * 16u; o < nb; o += kWave * 16u) {
        if (o + 16 <= nb) {
          const u32x4 q = load_u128_any(src + o);
          *reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(bufA) + o) = q;
        } else {
          for (uint32_t k = o; k < nb; ++k)
            reinterpret_cast<uint8_t*>(bufA)[k] = src[k];
        }
      }
      (void)dstw;
      if (lane < 16)
        meta[lane] = 0;
    }
    // Both layers work in place: a step reads elements [base, base + 64] and
    // then writes at or below base + 63 (RLE compacts, delta keeps the
    // index), LDS operations of a wave execute in order, and within a step
    // every lane's loads precede every lane's store.
    UT* const x = bufA;
    int rr = R, dr = D;
    for (int l = 0; l < layers && use; ++l) {
      if (rr > 0) { // reference :913-953
        const uint32_t m = wave_rle<UT>(x, n, x, cnts, lane);
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
        for (uint32_t i = (uint32_t)lane; i + 1 < n; i += kWave) {
          const UT hi = x[i + 1], lo = x[i];
          lds_lane_exchange_fence();
          x[i] = (UT)(hi - lo);
        }
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
    constexpr uint32_t kCode = CB == 8192 ? 1u : CB == 16384 ? 2u : 0u; // sub-chunk size (see the head of the file)
    const uint32_t h = use ? ((uint32_t)R | ((uint32_t)D << 8) | (((uint32_t)bp | (kCode << 4)) << 16)) : 0u;
    reinterpret_cast<HC_GLOBAL uint32_t*>(out)[0] = h | ((uint32_t)type_tag << 24);
    reinterpret_cast<HC_GLOBAL uint32_t*>(out)[1] = N * S;
    out_bytes_arr[part] = total;
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
  if ((off & 3u) || (off + ru(nbytes, 4)) / 4 > end_words)
    return -1;
  if (rel + ru(nbytes, 4) <= STAGE_WORDS * 4)
    return unpack_array<ET>(stage + rel / 4, nbytes, bp, dst, max_elems, lane);
  return unpack_array<ET>(reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + off), nbytes, bp, dst,
                          max_elems, lane);
}

template <int S, int CB>
__device__ __forceinline__ void cascaded_decode_partition(
    const uint8_t* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes_arr,
    const size_t* __restrict__ out_caps, uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ actual_bytes,
    hipcompStatus_t* __restrict__ statuses, const size_t part, uint8_t* const smem, const int lane);

template <int S, int CB>
__global__ __launch_bounds__(kWave) void cascaded_decompress_kernel(
    const uint8_t* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes_arr,
    const size_t* __restrict__ out_caps, const size_t batch,
    uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ actual_bytes,
    hipcompStatus_t* __restrict__ statuses)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; // dec_lds_bytes<S, CB>()
  const int lane = lane_id();
  // Each width and sub-chunk size has its own launch (the LDS a wave needs
  // depends on both); a partition is handled by the launch that matches ITS
  // type byte (the reference dispatches on partition 0 only) and ITS sub-chunk
  // code.  Undecodable headers are reported by the 1-byte 4096 launch.
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
      const uint32_t code = bad_header ? 0u : (uint32_t)comp[2] >> 4;
      constexpr uint32_t kCode = CB == 8192 ? 1u : CB == 16384 ? 2u : 0u;
      mine = (bad_header || type > 7 || code > 2)
                 ? (S == 1 && kCode == 0)
                 : (code == kCode
                    && ((S == 1 && type <= 1) || (S == 2 && (type == 2 || type == 3))
                        || (S == 4 && (type == 4 || type == 5)) || (S == 8 && (type == 6 || type == 7))));
    }
    for (uint64_t todo = wave_ballot(mine); todo != 0; todo &= todo - 1)
      cascaded_decode_partition<S, CB>(comp_ptrs, comp_bytes_arr, out_caps, out_ptrs, actual_bytes, statuses,
                                       first + (size_t)__builtin_ctzll(todo) * waves, smem, lane);
  }
}

// One partition, by one wave (the launch that owns its width and sub-chunk size).
template <int S, int CB>
__device__ __forceinline__ void cascaded_decode_partition(
    const uint8_t* const* __restrict__ comp_ptrs, const size_t* __restrict__ comp_bytes_arr,
    const size_t* __restrict__ out_caps, uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ actual_bytes,
    hipcompStatus_t* __restrict__ statuses, const size_t part, uint8_t* const smem, const int lane)
{
  typedef typename UIntOf<S>::type UT;
  cgptr comp = to_global(uniform_ptr(comp_ptrs[part]));
  const size_t comp_bytes64 = uniform((uint64_t)comp_bytes_arr[part]);
  const bool bad_header = comp == nullptr || comp_bytes64 < kPartMeta;
  uint32_t type = 0xFFu, code = 0;
  if (!bad_header) {
    type = uniform((uint32_t)comp[3]);
    code = uniform((uint32_t)comp[2]) >> 4;
  }
  const bool undecodable = bad_header || type > 7 || code > 2;
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
  UT* bufB = reinterpret_cast<UT*>(my + elem_buf_bytes<CB>());
  uint16_t* marks = reinterpret_cast<uint16_t*>(my + 2 * elem_buf_bytes<CB>());
  uint32_t* stage = reinterpret_cast<uint32_t*>(my + 2 * elem_buf_bytes<CB>() + (CB / S) * 2);
  const uint32_t* meta = stage; // the chunk metadata is the head of the staged image

  constexpr uint32_t CE = CB / S;
  const uint32_t end_w = comp_bytes / 4;
  const uint32_t dh_off = ru((uint32_t)(4 + 4 * (R + 1)), (uint32_t)S);
  const int layers = R > D ? R : D;
  uint32_t pos = ru(kPartMeta, S), done = 0;
  bool ok = true;
  // stage_words<CB>() words of the sub-chunk at `p`, stage_per_lane<CB>() per lane, clipped to the
  // partition (words past the end read as 0); issued one sub-chunk ahead
  uint32_t pf[stage_per_lane<CB>()];
  auto prefetch = [&](uint32_t p) {
    const HC_GLOBAL uint32_t* w = reinterpret_cast<const HC_GLOBAL uint32_t*>(comp + p);
    const uint32_t avail = end_w - p / 4;
#pragma unroll
    for (int k = 0; k < (int)stage_per_lane<CB>(); ++k) {
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
    for (int k = 0; k < (int)stage_per_lane<CB>(); ++k)
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
    int n = wave_read_array<UT, stage_words<CB>()>(comp, end_w, pos, msz + offs_final, uniform(meta[1 + R]), bp, stage, x, CE, lane);
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
        UT carry = *reinterpret_cast<const UT*>(reinterpret_cast<const uint8_t*>(meta) + dh_off + l * S);

        for (uint32_t b0 = 0; b0 < (uint32_t)n; b0 += kWave) {
          const uint32_t i = b0 + (uint32_t)lane;
          const UT own = i < (uint32_t)n ? x[i] : (UT)0;
          UT incl;
          if (S > 4)
            incl = (UT)wave_scan_add_u64((uint64_t)own);
          else
            incl = (UT)wave_scan_add_u32((uint32_t)own);
          if (i < (uint32_t)n)
            y[i] = (UT)(carry + incl - own);
          if (S > 4) {
            const uint64_t tot = (uint64_t)read_lane((uint32_t)((uint64_t)incl), 63)
                                 | ((uint64_t)read_lane((uint32_t)((uint64_t)incl >> 32), 63) << 32);
            carry = (UT)(carry + (UT)tot);
          } else {
            carry = (UT)(carry + (UT)read_lane((uint32_t)incl, 63));
          }
        }
        if (lane == 0)
          y[n] = carry;
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
          bool good = !((off & 3u) || (off + ru(nbytes, 4)) / 4 > end_w);
          if (good) {
            if (rel + ru(nbytes, 4) <= stage_words<CB>() * 4)
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
        HC_GLOBAL UT* gdst = reinterpret_cast<HC_GLOBAL UT*>(out + (size_t)done * S);
        uint32_t run_carry = 0;
        UT sum_carry = head;
        for (uint32_t b0 = 0; b0 < total; b0 += kWave) {
          const uint32_t j = b0 + (uint32_t)lane;
          const uint32_t mk = j < total ? marks[j] : 0u;
          uint32_t r = wave_scan_max_u32(mk);
          r = r > run_carry ? r : run_carry;
          run_carry = read_lane(r, 63);
          UT v = j < total ? x[r - 1] : (UT)0;
          if (with_delta) {
            UT incl;
            if (S > 4) {
              incl = (UT)wave_scan_add_u64((uint64_t)v);
              const uint64_t tot = (uint64_t)read_lane((uint32_t)((uint64_t)incl), 63)
                                   | ((uint64_t)read_lane((uint32_t)((uint64_t)incl >> 32), 63) << 32);
              v = (UT)(sum_carry + incl);
              sum_carry = (UT)(sum_carry + (UT)tot);
            } else {
              incl = (UT)wave_scan_add_u32((uint32_t)v);
              v = (UT)(sum_carry + incl);
              sum_carry = (UT)(sum_carry + (UT)read_lane((uint32_t)incl, 63));
            }
            if (j < total)
              y[j + 1] = v;
          } else if (to_hbm) {
            if (j < total)
              gdst[j] = v;
          } else {
            if (j < total)
              y[j] = v;
          }
        }
        if (with_delta) {
          if (lane == 0)
            y[0] = head;
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
CompressKernel compress_kernel_of(uint32_t cb, uint32_t& lds)
{
  switch (cb) {
  case 8192: lds = kWavesPerBlock * wave_lds_bytes<S, 8192>(); return cascaded_compress_kernel<S, 8192>;
  case 16384: lds = kWavesPerBlock * wave_lds_bytes<S, 16384>(); return cascaded_compress_kernel<S, 16384>;
  default: lds = kWavesPerBlock * wave_lds_bytes<S, 4096>(); return cascaded_compress_kernel<S, 4096>;
  }
}

template <int S, int CB>
hipError_t launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes, const size_t* out_caps, size_t batch,
    uint8_t* const* out_ptrs, size_t* actual_bytes, hipcompStatus_t* statuses, hipStream_t stream)
{
  DecompressKernel k = cascaded_decompress_kernel<S, CB>;
  constexpr uint32_t lds = dec_lds_bytes<S, CB>();
  if (lds > 64 * 1024) { // has to be asked for (idempotent, cheap)
    const hipError_t e
        = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
      return e;
  }
  // as many one-wave workgroups as the chip holds with this much LDS each (handed out in 1280-byte granules)
  uint32_t per_cu = (160u * 1024u) / ((lds + 1279u) / 1280u * 1280u);
  if (per_cu > 32)
    per_cu = 32;
  const size_t resident = (size_t)num_cus_of_current_device() * per_cu;
  k<<<dim3((unsigned)(batch < resident ? batch : resident)), dim3(kWave), lds, stream>>>(
      comp_ptrs, comp_bytes, out_caps, batch, out_ptrs, actual_bytes, statuses);
  return hipSuccess;
}

} // namespace

// chunk_bytes: 4096, 8192 or 16384 (the caller maps every other value to 4096)
void cascaded_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, size_t batch, int type_tag,
    int elem_size, int num_rles, int num_deltas, int use_bp, uint32_t chunk_bytes, hipStream_t stream)
{
  const dim3 grid((unsigned)((batch + kWavesPerBlock - 1) / kWavesPerBlock));
  const dim3 block(kWave * kWavesPerBlock);
  uint32_t lds = 0;
  CompressKernel k = elem_size == 1   ? compress_kernel_of<1>(chunk_bytes, lds)
                     : elem_size == 2 ? compress_kernel_of<2>(chunk_bytes, lds)
                     : elem_size == 4 ? compress_kernel_of<4>(chunk_bytes, lds)
                                      : compress_kernel_of<8>(chunk_bytes, lds);
  k<<<grid, block, lds, stream>>>(in_ptrs, in_bytes, out_ptrs, out_bytes, batch, type_tag, num_rles, num_deltas, use_bp);
}

// One launch per element width and sub-chunk size (the reference: one per
// width); a launch leaves the partitions of the others alone.  The common
// ones first.
hipError_t cascaded_launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    const size_t* out_caps, size_t batch, uint8_t* const* out_ptrs,
    size_t* actual_bytes, hipcompStatus_t* statuses, hipStream_t stream)
{
  // (every launch is set up before the first one runs anything the caller could see half-done:
  // the LDS limits are raised first, in the order of the launches)
#define HC_DEC(S, CB)                                                                                              \
  {                                                                                                                \
    const hipError_t e = launch_decompress<S, CB>(comp_ptrs, comp_bytes, out_caps, batch, out_ptrs, actual_bytes, \
                                                  statuses, stream);                                               \
    if (e != hipSuccess)                                                                                           \
      return e;                                                                                                    \
  }
  HC_DEC(4, 4096) HC_DEC(8, 4096) HC_DEC(2, 4096) HC_DEC(1, 4096)
  HC_DEC(4, 8192) HC_DEC(8, 8192) HC_DEC(2, 8192) HC_DEC(1, 8192)
  HC_DEC(4, 16384) HC_DEC(8, 16384) HC_DEC(2, 16384) HC_DEC(1, 16384)
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

} // namespace hcamd
