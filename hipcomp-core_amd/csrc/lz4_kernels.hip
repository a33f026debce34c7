// lz4_kernels.hip -- gfx950 kernels of the batched LZ4 block codec: the one
// translation unit.  The device code lies in the parts included below,
//   lz4_common.hiph  hash, sequence writers, the reference's insert rule, the
//                    one-window match search, match length, emission, tickets
//   lz4_mix.hiph     "mix" shape: hash (+ tag) tables in LDS, block-pipelined walk
//                    over match-less stretches -- data that does not compress
//   lz4_far.hiph     "far" shapes: tables in device memory (the caller's temp
//                    buffer) or LDS, several sequences per trip to memory --
//                    data that compresses; the sampling kernel that picks
//   lz4_decode.hiph  the decoder
// and the host launchers follow here.
//
// Compressed bytes are those of the reference's wave64 encoder
// (reference src/LZ4Kernels.hiph:793-969 compressStream<T>), produced by a
// different mechanism:
//
//   reference                               here
//   --------------------------------------  ---------------------------------
//   32 KiB hash table per chunk in HBM      data without matches: table in LDS
//   (temp space), global_store_short        (ds_read_u16 / ds_write_b16); data
//                                           that compresses: one table per
//                                           RESIDENT WAVE in the temp space
//                                           (32 waves per CU instead of 5)
//   every table candidate is verified by    a second LDS table holds 8 more
//   a 4-byte gather from the input (a       hash bits of the word each entry
//   64-line gather per window: the memory   was made from; a candidate whose
//   pipe's bound, scripts/probes/           tag differs cannot match and is
//   gather_rate.hip)                        not fetched
//   warpMatchAny = 64-step LDS loop, twice  in-window duplicates: found through
//   per window (:218-245)                   the table itself (one-window path:
//                                           lane ids posted in reversed lane
//                                           order; walk: a lane that does not
//                                           read back its own insert shares a
//                                           slot), exact compare only for
//                                           those lanes
//   second warpMatchAny for the insert      insert rule (incl. the wave64
//   (:722-741) + hardware arbitration of    `int` truncation)
//   same-address global_store_short         = ONE masked LDS store with the
//                                           lanes in priority order ("sigma
//                                           order", see sigma_of_lane)
//   one window, one sequence at a time      match-less stretches: blocks of
//   (:925-956)                              windows, all LDS traffic of a block
//                                           issued back to back, decisions one
//                                           block later (walk_*); compressible
//                                           data: one trip to the table and to
//                                           the candidates serves several
//                                           sequences (far_straight_several)
//   shuffleLiterals (:754-791)              one unaligned dword load per lane
//   1 byte/lane literal + match compare     16-byte/lane copies, 4-byte/lane
//                                           match-length compare
//
// One chunk per wavefront: the window loop is a serial dependency chain, the
// 64 lanes are the 64 window positions.
//
// Decoder: reference src/LZ4Kernels.hiph:971-1097 decompressStream.

#include "lz4_launch.hpp"
#include "placement.hiph"
#include "wave_utils.hpp"


#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>


namespace hcamd {

namespace {

#include "lz4_common.hiph"
#include "lz4_mix.hiph"
#include "lz4_far.hiph"
#include "lz4_decode.hiph"

} // namespace

// ---- launchers -----------------------------------------------------------

namespace {

// Per-device facts and one-time setup, looked up by the calling thread's
// current device (one process may drive several GPUs, one thread each, as the
// reference's callers do).  Both steps are idempotent, so a race between two
// first callers on one device is harmless.
constexpr int kMaxDevices = 64;
std::atomic<int> g_num_cus[kMaxDevices];
std::atomic<int> g_lds_raised[kMaxDevices]; // 0 = not yet, 1 = done, < 0 = -hipError

// -1: no current device, or one beyond the per-device state kept here
int current_device()
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
    return -1;
  return dev;
}

} // namespace

int num_cus_of_current_device()
{
  const int dev = current_device();
  if (dev < 0)
    return 256;
  int n = g_num_cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

namespace {

typedef void (*MixKernel)(
    const uint8_t* const*, const size_t*, uint8_t* const*, size_t*, uint32_t, uint32_t, uint32_t, uint32_t,
    uint32_t, uint32_t*, uint32_t, const uint32_t*, const uint32_t*, Lz4Placement, uint32_t);
typedef void (*FarKernel)(const uint8_t* const*, const size_t*, uint8_t* const*, size_t*, uint32_t, uint16_t*,
                          uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t, uint32_t, const uint32_t*,
                          const uint32_t*, Lz4Placement, uint32_t*, uint32_t*);

typedef void (*PairKernel)(
    const uint8_t* const*, const size_t*, uint8_t* const*, size_t*, uint32_t, uint32_t, uint32_t,
    uint32_t, uint32_t*, uint32_t, const uint32_t*, const uint32_t*, Lz4Placement, uint32_t);

PairKernel pair_kernel_for(int elem_size)
{
  return elem_size == 1 ? lz4_compress_kernel_pair<1> : elem_size == 2 ? lz4_compress_kernel_pair<2>
                                                                      : lz4_compress_kernel_pair<4>;
}

MixKernel mix_kernel_for(int elem_size)
{
  return elem_size == 1 ? lz4_compress_kernel_mix<1> : elem_size == 2 ? lz4_compress_kernel_mix<2>
                                                                      : lz4_compress_kernel_mix<4>;
}
template <int FORM>
FarKernel far_kernel_of_form(int elem_size)
{
  return elem_size == 1 ? lz4_compress_kernel_far<1, FORM> : elem_size == 2 ? lz4_compress_kernel_far<2, FORM>
                                                                             : lz4_compress_kernel_far<4, FORM>;
}
FarKernel far_kernel_for(int elem_size, uint32_t cls)
{
  return cls == kClassWide ? far_kernel_of_form<kFormWide>(elem_size)
         : cls == kClassDense ? far_kernel_of_form<kFormChains>(elem_size) : far_kernel_of_form<kFormLean>(elem_size);
}

// more than 64 KiB of dynamic LDS has to be asked for, once per kernel and device
hipError_t raise_dynamic_lds_limit()
{
  const int dev = current_device();
  if (dev < 0)
    return hipErrorInvalidDevice;
  const int state = g_lds_raised[dev].load(std::memory_order_acquire);
  if (state == 1)
    return hipSuccess;
  if (state < 0)
    return (hipError_t)(-state);
  hipError_t r = hipSuccess;
  for (int es = 1; es <= 4 && r == hipSuccess; es *= 2) {
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel_for(es)),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (uint32_t cls = kClassDense; cls <= kClassWide && r == hipSuccess; ++cls)
      r = hipFuncSetAttribute(reinterpret_cast<const void*>(far_kernel_for(es, cls)),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  g_lds_raised[dev].store(r == hipSuccess ? 1 : -(int)r, std::memory_order_release);
  return r;
}

constexpr uint32_t kLdsPerCu = 160u * 1024u;
constexpr uint32_t kLdsGranule = 1280u; // the CU hands LDS out in these (scripts/probes/lds_occupancy.hip)

uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

// persistent workgroups: enough to occupy every CU; late ones find the ticket
// counter exhausted and leave at once
void set_groups(Lz4CompressShape& sh, size_t batch)
{
  const uint32_t w = sh.waves();
  uint32_t per_cu = kLdsPerCu / round_up(sh.lds_bytes, kLdsGranule);
  if (per_cu > 8)
    per_cu = 8;
  if (per_cu * w > 32)
    per_cu = 32 / w;
  const size_t want = (batch + w - 1) / w;
  const size_t cap = (size_t)num_cus_of_current_device() * per_cu;
  sh.groups = (uint32_t)(want < cap ? want : cap);
}

} // namespace

Lz4CompressShape lz4_compress_shape_mix(uint32_t ht_size, size_t batch)
{
  Lz4CompressShape sh;
#ifdef HC_TAG_HALF
  sh.stride_tagged = round_up(ht_size * 2u + ht_size / 2u, 16u);
#else
  sh.stride_tagged = round_up(ht_size * 3u, 16u);
#endif
  sh.stride_plain = round_up(ht_size * 2u, 16u);
  // most waves per CU first (workgroups of g waves, as many as fit), then
  // most of them with tags
  uint32_t best_waves = 0, best_tagged = 0;
  sh.tagged = 1;
  sh.plain = 0;
  for (uint32_t g = kLz4MaxWavesPerGroup; g >= 1; --g) {
    if ((size_t)g > batch && g > 1)
      continue;
    for (uint32_t t = g;; --t) {
      const uint32_t lds = t * sh.stride_tagged + (g - t) * sh.stride_plain;
      if (lds <= kLdsPerCu) {
        uint32_t per_cu = kLdsPerCu / round_up(lds, kLdsGranule);
        if (per_cu > 8)
          per_cu = 8;
        const uint32_t waves = g * per_cu, tagged = t * per_cu;
        if (waves > best_waves || (waves == best_waves && tagged > best_tagged)) {
          best_waves = waves;
          best_tagged = tagged;
          sh.tagged = t;
          sh.plain = g - t;
        }
        break; // fewer tags in a group of this size cannot be better
      }
      if (t == 0)
        break;
    }
  }
  sh.lds_bytes = sh.tagged * sh.stride_tagged + sh.plain * sh.stride_plain;
  set_groups(sh, batch);
  return sh;
}

// The pair shape (lz4_mix.hiph, lz4_compress_kernel_pair): one chunk per workgroup of two waves.
// `tagged`: with a tag table (64 KiB chunks: three workgroups per CU instead of four).
struct Lz4PairShape
{
  uint32_t tagged, table_bytes, lds_bytes, groups;
};
Lz4PairShape lz4_compress_shape_pair(uint32_t ht_size, size_t batch, uint32_t tagged)
{
  Lz4PairShape sh;
  sh.tagged = tagged; // (0: no tags, 1: a tag table, 2: tags in the positions)
  sh.table_bytes = round_up(ht_size * (tagged == 1u ? 3u : 2u), 16u);
  sh.lds_bytes = sh.table_bytes + 64u; // (kPairSyncBytes)
#ifdef HC_MEASUREMENT_KNOBS
  // (HIPCOMP_LZ4_PAIR_LDS: more LDS than a pair needs, i.e. fewer pairs per CU -- what the pairs of a CU cost each other)
  if (const char* e = std::getenv("HIPCOMP_LZ4_PAIR_LDS"))
    if ((uint32_t)std::atoi(e) > sh.lds_bytes && (uint32_t)std::atoi(e) <= 64u * 1024u)
      sh.lds_bytes = (uint32_t)std::atoi(e);
#endif
  uint32_t per_cu = kLdsPerCu / round_up(sh.lds_bytes, kLdsGranule);
  if (per_cu > 8)
    per_cu = 8;
  const size_t cap = (size_t)num_cus_of_current_device() * per_cu;
  sh.groups = (uint32_t)(batch < cap ? batch : cap);
  return sh;
}

// 0: the mix kernel of rounds 1-4 (four lone waves per CU); 1: pairs with tag tables; 2: pairs without.
// Pairs where they are not slower than the lone waves (scripts/sweep_pair.py, profiles/r05_pair_sweep.txt):
// chunks of more than 32 KiB (the walk is what the second wave shares; the rest of a chunk's work is wave
// 0's alone: 32 KiB chunks 415 against 429 GB/s, 64 KiB 466 against 421) and of at most 64 KiB (longer chunks
// take the walk of one wave), and a batch of two rounds or more of the 3 pairs a CU holds (four lone waves
// hold a chunk more: 1000 x 64 KiB 268 against 338 GB/s, 1500: 360 against 338).
// With the tags in the positions (4-byte elements: four pairs per CU, every lone wave with a filter too) pairs win
// from 32 KiB chunks and a thousand chunks on (sweep_pair3.log: 32 KiB 1 121 against 1 052 GB/s, 16 KiB 891 / 880,
// 1000 x 64 KiB 677 / 684, 1500: 738 / 692).
int lz4_pair_mode(uint32_t ht_size, size_t max_chunk_bytes, size_t batch, bool inpos)
{
  const size_t cus = (size_t)num_cus_of_current_device();
  int mode = ht_size >= 8192 && max_chunk_bytes <= 65536
                     && (inpos ? max_chunk_bytes > 16384 && batch >= 4u * cus
                               : max_chunk_bytes > 32768 && batch >= 2u * 3u * cus)
                 ? 1 : 0;
#ifdef HC_MEASUREMENT_KNOBS
  if (const char* e = std::getenv("HIPCOMP_LZ4_PAIR"))
    mode = ht_size >= 8192 ? std::atoi(e) : 0;
#endif
  return mode;
}

// (measurement knob HIPCOMP_LZ4_INPOS=0, knobs build only: 4-byte elements with the tag tables of the other widths)
bool lz4_inpos_wanted()
{
#ifdef HC_MEASUREMENT_KNOBS
  if (const char* e = std::getenv("HIPCOMP_LZ4_INPOS"))
    return std::atoi(e) != 0;
#endif
  return true;
}

// The library that ships reads nothing from the environment: every chunk goes where the routing
// kernel sends it.  The knobs below exist in the measurement / test build only
// (`make VARIANT=knobs EXTRA=-DHC_MEASUREMENT_KNOBS` -> lib/libhipcomp_knobs.so: the same device code,
// tests/test_build_guards_cpu.py compares the code objects), where they are read at every call so that
// the tests can switch shapes inside one process.
Lz4Mode lz4_mode_from_environment()
{
#ifdef HC_MEASUREMENT_KNOBS
  const char* e = std::getenv("HIPCOMP_LZ4_SHAPE");
  if (e && std::strcmp(e, "mix") == 0)
    return Lz4Mode::Mix;
  if (e && std::strcmp(e, "far") == 0)
    return Lz4Mode::Far;
  if (e && std::strcmp(e, "fars") == 0)
    return Lz4Mode::FarSparse;
  if (e && std::strcmp(e, "farw") == 0)
    return Lz4Mode::FarWide;
#endif
  return Lz4Mode::Auto;
}

namespace {

// Launch geometry of the far kernel for one class of data (lz4_far.hiph, above
// lz4_compress_kernel_far): per workgroup `near` waves with their table in LDS
// and `far` waves with their table in the temp buffer, `slots` scratch slots per
// wave, `groups` workgroups.
struct FarGeometry
{
  uint32_t near, far, slots, groups, lds_bytes;
  uint32_t waves() const { return near + far; }
};

// Measurement knob HIPCOMP_LZ4_GEOMETRY="near,far,slots" (knobs build only): that geometry for every
// far-type launch, as many workgroups as fit a CU.
bool geometry_from_environment(uint32_t& near, uint32_t& far, uint32_t& slots)
{
#ifdef HC_MEASUREMENT_KNOBS
  const char* e = std::getenv("HIPCOMP_LZ4_GEOMETRY");
  unsigned a = 0, b = 0, c = 0;
  if (e && std::sscanf(e, "%u,%u,%u", &a, &b, &c) == 3 && a + b >= 1 && a + b <= (unsigned)kFarMaxWavesPerGroup
      && c >= 64 && c <= 4096 && (c & (c - 1)) == 0) {
    near = a;
    far = b;
    slots = c;
    return true;
  }
#else
  (void)near;
  (void)far;
  (void)slots;
#endif
  return false;
}

// lanes a trip of the device-table waves' lean form looks up (HIPCOMP_LZ4_SPAN: measurement knob, knobs build only)
uint32_t far_span(uint32_t cls)
{
#ifdef HC_MEASUREMENT_KNOBS
  if (const char* e = std::getenv("HIPCOMP_LZ4_SPAN")) {
    const int v = std::atoi(e);
    if (v >= 8 && v <= 64)
      return (uint32_t)v;
  }
#endif
  return cls == kClassDense ? (uint32_t)kFarSpanFull : (uint32_t)kFarSpan;
}

uint32_t groups_per_cu(uint32_t lds_bytes, uint32_t waves)
{
  uint32_t g = kLdsPerCu / round_up(lds_bytes ? lds_bytes : 1u, kLdsGranule);
  if (g > 8)
    g = 8;
  if (g * waves > 32)
    g = 32 / waves;
  return g;
}

FarGeometry far_geometry(uint32_t ht_size, uint32_t cls, size_t batch, size_t far_capacity)
{
  const uint32_t table = 2u * (ht_size < 8 ? 8u : ht_size);
  const uint32_t cus = (uint32_t)num_cus_of_current_device();
  FarGeometry g = {};
  uint32_t near = 0, far = 0, slots = 0;
  if (geometry_from_environment(near, far, slots)) {
    g.near = near;
    g.far = far;
    g.slots = slots;
    g.lds_bytes = near * (table + 2u * slots) + far * 2u * slots;
    uint32_t per_cu = g.lds_bytes <= kLdsPerCu ? groups_per_cu(g.lds_bytes, g.waves()) : 0;
    g.groups = per_cu * cus;
  } else {
    // few chunks: a wave with its table in LDS for each of them, as far as LDS goes
    const uint32_t lone_lds = table + 2u * kFarScratchSlots;
    const uint32_t lone_per_cu = lone_lds <= kLdsPerCu ? groups_per_cu(lone_lds, 1) : 0;
    if (lone_per_cu > 0 && batch <= (size_t)lone_per_cu * cus) {
      g.near = 1;
      g.far = 0;
      g.slots = kFarScratchSlots;
      g.lds_bytes = lone_lds;
      g.groups = (uint32_t)batch;
      return g;
    }
    // else: workgroups of 1, 2 or 4 LDS-table waves and as many device-table waves as fill the CU's
    // 32 wave slots (dense, wide) or two and a half per LDS-table wave (sparse), as many workgroups per CU as
    // LDS holds -- the split with the most LDS-table waves per CU, then the smallest workgroups
    // (64 KiB chunks: 4 x (1 + 7), sparse 2 x (2 + 5); 8 KiB chunks: 8 x (1 + 3); chunks of 2 KiB: 8 x (4 + 0))
    g.slots = 512;
    uint32_t best = 0, best_near = 0;
    for (uint32_t wn = 1; wn <= 4; wn *= 2)
      for (uint32_t per_cu = 8; per_cu >= 1; --per_cu) {
        if (per_cu * wn > 32)
          continue;
        const uint32_t nf = 32 / per_cu - wn;
        if (wn + nf > (uint32_t)kFarMaxWavesPerGroup)
          continue;
        const uint32_t lds = wn * table + (wn + nf) * 2u * g.slots;
        if (lds <= kLdsPerCu && kLdsPerCu / round_up(lds, kLdsGranule) >= per_cu) {
          // (sparse: pairs of LDS-table waves, so that two and a half device-table waves go with each)
          if (per_cu * wn > best_near || (cls == kClassSparse && per_cu * wn == best_near && wn == 2)) {
            best_near = per_cu * wn;
            best = per_cu;
            g.near = wn;
            g.far = nf;
          }
          break; // (fewer workgroups of this kind per CU hold no more LDS tables)
        }
      }
    if (best == 0) { // (tables beyond what LDS holds: device-table waves only)
      g.near = 0;
      g.far = 4;
      g.slots = kFarScratchSlots;
      best = 8;
    }
    // sparse data (text): the device-table waves beyond two and a half per LDS-table wave only queue
    // on the fabric (64 KiB chunks, 65 536 of them, LDS-table + device-table waves per CU: 4 + 8: 55.2
    // GB/s, 4 + 10: 60.7, 4 + 11: 60.4, 4 + 12: 58.2, 4 + 14: 54.8)
    if (cls == kClassSparse && g.near > 0) {
      const uint32_t most = g.near >= 2 ? 5 * g.near / 2 : 3;
      if (g.far > most)
        g.far = most;
    }
    g.groups = best * cus;
  }
  // no more device-table waves than the batch needs and the temp buffer has tables for
  if (g.groups > 0 && g.far > 0) {
    const size_t want = (batch + g.groups - 1) / g.groups; // waves per workgroup that have a chunk
    if (want < g.waves())
      g.far = (uint32_t)(want > g.near ? want - g.near : 0);
    if ((size_t)g.groups * g.far > far_capacity)
      g.far = (uint32_t)(far_capacity / g.groups);
    if (g.near == 0 && g.far == 0)
      g.groups = 0;
  }
  g.lds_bytes = g.near * (table + 2u * g.slots) + g.far * 2u * g.slots;
  if (g.groups > 0 && (size_t)g.groups * g.waves() > batch + g.waves() - 1)
    g.groups = (uint32_t)((batch + g.waves() - 1) / g.waves());
  return g;
}

} // namespace

#ifdef HC_TRIP_STATS
// (measurement build only; the name makes it pass the export map)
extern "C" int hipcompBatchedLZ4DebugTripStats(uint32_t* host16, int reset)
{
  uint32_t zeros[16] = {};
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_trip_stats), sizeof(zeros)) != hipSuccess)
    return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_trip_stats), zeros, sizeof(zeros)) != hipSuccess)
    return 2;
  return 0;
}
extern "C" int hipcompBatchedLZ4DebugTripLog(uint32_t* host_words, uint32_t* count, int reset)
{
  if (hipMemcpyFromSymbol(count, HIP_SYMBOL(g_trip_log_n), 4) != hipSuccess
      || hipMemcpyFromSymbol(host_words, HIP_SYMBOL(g_trip_log), 4u << 16) != hipSuccess)
    return 1;
  uint32_t zero = 0;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_trip_log_n), &zero, 4) != hipSuccess)
    return 2;
  return 0;
}
#endif

#ifdef HC_PAIR_DEBUG
// (diagnostic build only; the name makes it pass the export map)
extern "C" int hipcompBatchedLZ4DebugPair(uint32_t* host16, int reset)
{
  uint32_t zeros[32] = {};
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_pair_dbg), sizeof(zeros)) != hipSuccess)
    return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_pair_dbg), zeros, sizeof(zeros)) != hipSuccess)
    return 2;
  return 0;
}
#endif

#ifdef HC_MIX_STAMPS
// (diagnostic build only; the name makes it pass the export map)
extern "C" int hipcompBatchedLZ4DebugMixStamps(unsigned long long* host8, int reset)
{
  unsigned long long zeros[8] = {};
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_mix_stamps), sizeof(zeros)) != hipSuccess)
    return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_mix_stamps), zeros, sizeof(zeros)) != hipSuccess)
    return 2;
  return 0;
}
#endif

#ifdef HC_DEC_STAMPS
// (diagnostic build only; the name makes it pass the export map)
extern "C" int hipcompBatchedLZ4DebugDecodeStamps(unsigned long long* host8, int reset)
{
  unsigned long long zeros[8] = {};
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_dec_stamps), sizeof(zeros)) != hipSuccess)
    return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_dec_stamps), zeros, sizeof(zeros)) != hipSuccess)
    return 2;
  return 0;
}
#endif

// words[0 .. blockDim.x) = 0: the ticket counters of a call, on its stream
__global__ void lz4_zero_words_kernel(uint32_t* words)
{
  words[threadIdx.x] = 0;
}

size_t lz4_placement_slots()
{
  // (the far kernels: at most 32 waves per CU; the mix kernel 4)
  return (size_t)num_cus_of_current_device() * 32u;
}

size_t lz4_compress_temp_bytes_used(uint32_t ht_size, size_t batch)
{
  // header, the four class lists, alignment, one table per chunk but no more than the chip holds waves
  const size_t tables = batch < 8192 ? batch : 8192;
  return 4 + kHeaderWords * sizeof(uint32_t) + (kNumClasses + 1) * batch * sizeof(uint32_t) + 16
         + tables * (size_t)(ht_size < 8 ? 8 : ht_size) * sizeof(uint16_t);
}

hipError_t lz4_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, uint32_t ht_size,
    size_t batch, int elem_size, void* temp, size_t temp_bytes,
    size_t max_chunk_bytes, Lz4Mode mode, hipStream_t stream, const Lz4Placement* place_or_null)
{
  const Lz4Placement place = place_or_null ? *place_or_null : Lz4Placement();
  // ---- the temp buffer: header (ticket counters, list lengths, sample totals), the class
  // lists of the routing kernel, hash tables for the device-table waves of the far kernel --
  // as much of that as the (contract-sized) buffer holds
  uint32_t* header = nullptr;
  uint32_t* lists = nullptr;
  uint32_t* retry_list = nullptr;
  uint16_t* far_tables = nullptr;
  size_t far_capacity = 0;
  if (temp != nullptr) {
    const uintptr_t base = reinterpret_cast<uintptr_t>(temp), end = base + temp_bytes;
    const uintptr_t aligned = (base + 3u) & ~uintptr_t(3);
    if (aligned + kHeaderWords * sizeof(uint32_t) <= end) {
      header = reinterpret_cast<uint32_t*>(aligned);
      uintptr_t at = aligned + kHeaderWords * sizeof(uint32_t);
      if (at + kNumClasses * batch * sizeof(uint32_t) <= end) {
        lists = reinterpret_cast<uint32_t*>(at);
        at += kNumClasses * batch * sizeof(uint32_t);
        // (a fifth list: the chunks the far kernels give back to the LDS shape, lz4_common.hiph give_away)
        if (at + batch * sizeof(uint32_t) <= end) {
          retry_list = reinterpret_cast<uint32_t*>(at);
          at += batch * sizeof(uint32_t);
        }
      }
      const uintptr_t tables = (at + 15u) & ~uintptr_t(15);
      if (tables < end) {
        far_tables = reinterpret_cast<uint16_t*>(tables);
        far_capacity = (end - tables) / ((size_t)(ht_size < 8 ? 8 : ht_size) * sizeof(uint16_t));
      }
    }
  }
  const hipError_t raised = raise_dynamic_lds_limit();
  if (raised != hipSuccess)
    return raised;
  const Lz4CompressShape mix = lz4_compress_shape_mix(ht_size, batch);
  // 4-byte elements in chunks of at most 64 KiB: the tags live in the positions' two spare bits (lz4_common.hiph,
  // Tables INPOS) -- no tag table, four pairs per CU instead of three, four lone waves all with a filter
  const bool inpos = elem_size == 4 && max_chunk_bytes <= 65536 && lz4_inpos_wanted();
  const int pair_mode = lz4_pair_mode(ht_size, max_chunk_bytes, batch, inpos);
  const Lz4PairShape pair = lz4_compress_shape_pair(ht_size, batch, inpos ? 2u : pair_mode == 1 ? 1u : 0u);
  Lz4CompressShape mix_inpos = mix;
  if (inpos) { // (every wave's tables are the position table alone)
    mix_inpos.tagged = 0;
    mix_inpos.plain = kLz4MaxWavesPerGroup;
    while (mix_inpos.plain > 1 && (size_t)mix_inpos.plain > batch)
      --mix_inpos.plain;
    mix_inpos.lds_bytes = mix_inpos.plain * mix_inpos.stride_plain;
    set_groups(mix_inpos, batch);
  }
  const Lz4CompressShape& mixs = inpos ? mix_inpos : mix;
  // about 16 KiB of input per ticket, but at least 4 tickets per wave so
  // that the last ones even out the load
  auto chunks_per_ticket = [&](size_t all_waves) {
    uint32_t per_ticket = 1;
    while (per_ticket < 64 && (size_t)per_ticket * (max_chunk_bytes ? max_chunk_bytes : 1) < 16384
           && (size_t)per_ticket * 2 * 4 * all_waves <= batch)
      per_ticket *= 2;
    return per_ticket;
  };
  // (`ticket_word`: the header's word that is this launch's ticket counter; give: its waves may hand chunks
  // that open like data that compresses on to the sparse class, whose kernel runs behind them)
  auto launch_mix = [&](const uint32_t* count, const uint32_t* list, uint32_t ticket_word, bool give) {
    uint32_t* ticket = header ? header + ticket_word : nullptr;
    // ticket == nullptr: no persistent workgroups, one chunk per wave
    const dim3 grid(ticket ? mixs.groups : (unsigned)((batch + mixs.waves() - 1) / mixs.waves()));
    const size_t resident = pair_mode != 0 ? (size_t)pair.groups : (size_t)mixs.groups * mixs.waves(); // chunks in flight
    const uint32_t per_ticket = chunks_per_ticket(resident);
    if (pair_mode != 0)
      pair_kernel_for(elem_size)<<<dim3(ticket ? pair.groups : (unsigned)batch), dim3(2 * kWave), pair.lds_bytes, stream>>>(
          in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, pair.tagged, pair.table_bytes,
          (uint32_t)batch, ticket, per_ticket, count, list, place, give ? 1u : 0u);
    else
    mix_kernel_for(elem_size)<<<grid, dim3(mixs.waves() * kWave), mixs.lds_bytes, stream>>>(
        in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, mixs.tagged, mixs.stride_tagged, mixs.stride_plain,
        (uint32_t)batch, ticket, per_ticket, count, list, place, (give ? 1u : 0u) | (inpos ? 2u : 0u));
  };
  auto launch_far = [&](uint32_t cls, const uint32_t* counts, const uint32_t* all_lists) -> bool {
    const FarGeometry g = far_geometry(ht_size, cls, batch, far_tables ? far_capacity : 0);
    if (g.groups == 0 || g.lds_bytes > kLdsPerCu)
      return false;
    far_kernel_for(elem_size, cls)<<<dim3(g.groups), dim3(g.waves() * kWave), g.lds_bytes, stream>>>(
        in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, far_tables, g.near, g.slots,
        far_span(cls), (uint32_t)batch, header + cls,
        chunks_per_ticket((size_t)g.groups * g.waves()), cls, counts, all_lists, place,
        counts && retry_list ? header + kHeaderRetryCount : nullptr, counts ? retry_list : nullptr);
    return true;
  };
  if (!header) { // (a temp buffer too small for a ticket counter)
    if (place.slots) // (one slot per RESIDENT wave: needs the persistent grids)
      return hipErrorInvalidValue;
    launch_mix(nullptr, nullptr, kClassMix, false);
    return hipSuccess;
  }
  // (zeroed by a kernel, not hipMemsetAsync: see lz4_launch_decompress)
  lz4_zero_words_kernel<<<dim3(1), dim3(kHeaderWords), 0, stream>>>(header);
  {
    const hipError_t zeroed = hipGetLastError();
    if (zeroed != hipSuccess)
      return zeroed;
  }
  if (mode == Lz4Mode::Auto && lists) {
    // every chunk to the shape its data calls for
    // (chunks per workgroup: one per wave while that leaves the chip room, at most 64 -- one list
    // atomic per workgroup and class, and the atomics of a class all go to one address)
    uint32_t per_group = kRouteWaves;
    while (per_group < kRouteMostPerGroup && batch / per_group > 4096)
      per_group *= 2;
    lz4_route_kernel<<<dim3((unsigned)((batch + per_group - 1) / per_group)), dim3(kRouteWaves * kWave), 0, stream>>>(
        in_ptrs, in_bytes, (uint32_t)batch, per_group, header, lists);
    launch_mix(header + 4 + kClassMix, lists + kClassMix * batch, kClassMix, true);
    for (uint32_t cls = kClassDense; cls <= kClassWide; ++cls)
      if (!launch_far(cls, header + 4, lists))
        return hipErrorInvalidValue; // (cannot happen: the LDS-table waves need nothing but the header)
    // what the far kernels gave back (chunks that open without a match): once more the LDS shape, which keeps them
    if (retry_list)
      launch_mix(header + kHeaderRetryCount, retry_list, kHeaderRetryTicket, false);
    return hipSuccess;
  }
  const uint32_t forced = mode == Lz4Mode::Far ? kClassDense : mode == Lz4Mode::FarSparse ? kClassSparse
                          : mode == Lz4Mode::FarWide ? kClassWide : kClassMix;
  if (forced == kClassMix || !launch_far(forced, nullptr, nullptr))
    launch_mix(nullptr, nullptr, kClassMix, false);
  return hipSuccess;
}

hipError_t lz4_launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    const size_t* out_caps, size_t batch, uint8_t* const* out_ptrs,
    size_t* actual_bytes, hipcompStatus_t* statuses, bool write_out,
    hipStream_t stream, void* temp, size_t temp_bytes)
{
  // More chunks than the chip holds waves: a persistent grid that draws its chunks from a ticket counter
  // (lz4_decode.hiph), else a wave per chunk by position.  The counter is ONE word of the caller's temp buffer,
  // zeroed on the stream -- a different word for every call of the process (a running call number, modulo the
  // buffer's words: 6 per chunk by the size contract, i.e. at least 49 152 where tickets are used at all), so
  // that calls in flight at once on several streams may share one temp buffer, as they may with the reference,
  // which never touches it (src/lowlevel/LZ4CompressionKernels.hip:224-249; tests/test_lz4_gpu.py:
  // test_concurrent_decompress_calls_share_one_temp_buffer).
  static std::atomic<uint32_t> calls{0};
  uint32_t* ticket = nullptr;
  size_t groups = (batch + kDecompWavesPerBlock - 1) / kDecompWavesPerBlock;
  const size_t resident = (size_t)num_cus_of_current_device() * (32 / kDecompWavesPerBlock);
  if (groups > resident && temp != nullptr) {
    const uintptr_t at = (reinterpret_cast<uintptr_t>(temp) + 3u) & ~uintptr_t(3);
    const uintptr_t end = reinterpret_cast<uintptr_t>(temp) + temp_bytes;
    if (at + sizeof(uint32_t) <= end) {
      const size_t words = (end - at) / sizeof(uint32_t);
      ticket = reinterpret_cast<uint32_t*>(at) + calls.fetch_add(1, std::memory_order_relaxed) % words;
    }
  }
  if (ticket) {
    // (a kernel, not hipMemsetAsync: with the memset, a graph captured from compress + decompress replayed
    // with wrong bytes in round 4 -- tests/test_graph_capture_gpu.py; with the kernel it does not)
    lz4_zero_words_kernel<<<dim3(1), dim3(1), 0, stream>>>(ticket);
    const hipError_t zeroed = hipGetLastError();
    if (zeroed != hipSuccess)
      return zeroed;
    groups = resident;
  }
  const dim3 grid((unsigned)groups);
  const dim3 block(kWave * kDecompWavesPerBlock);
  if (write_out)
    lz4_decompress_kernel<true><<<grid, block, 0, stream>>>(
        comp_ptrs, comp_bytes, out_caps, batch, out_ptrs, actual_bytes, statuses, ticket);
  else
    lz4_decompress_kernel<false><<<grid, block, 0, stream>>>(
        comp_ptrs, comp_bytes, nullptr, batch, nullptr, actual_bytes, nullptr, ticket);
  return hipSuccess;
}

} // namespace hcamd
