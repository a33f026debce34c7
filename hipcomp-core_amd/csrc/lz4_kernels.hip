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
#include "wave_utils.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>


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
    uint32_t, uint32_t*, uint32_t, const uint32_t*);
typedef void (*FarKernel)(const uint8_t* const*, const size_t*, uint8_t* const*, size_t*, uint32_t, uint16_t*,
                          uint32_t, uint32_t*, uint32_t, const uint32_t*);

MixKernel mix_kernel_for(int elem_size)
{
  return elem_size == 1 ? lz4_compress_kernel_mix<1> : elem_size == 2 ? lz4_compress_kernel_mix<2>
                                                                      : lz4_compress_kernel_mix<4>;
}
FarKernel far_kernel_for(int elem_size, bool wide)
{
  if (wide)
    return elem_size == 1 ? lz4_compress_kernel_far<1, true> : elem_size == 2 ? lz4_compress_kernel_far<2, true>
                                                                               : lz4_compress_kernel_far<4, true>;
  return elem_size == 1 ? lz4_compress_kernel_far<1, false> : elem_size == 2 ? lz4_compress_kernel_far<2, false>
                                                                              : lz4_compress_kernel_far<4, false>;
}

typedef void (*BothKernel)(const uint8_t* const*, const size_t*, uint8_t* const*, size_t*, uint32_t, uint16_t*,
                           uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t, const uint32_t*);
BothKernel both_kernel_for(int elem_size, bool wide)
{
  if (wide)
    return elem_size == 1 ? lz4_compress_kernel_both<1, true> : elem_size == 2 ? lz4_compress_kernel_both<2, true>
                                                                                : lz4_compress_kernel_both<4, true>;
  return elem_size == 1 ? lz4_compress_kernel_both<1, false> : elem_size == 2 ? lz4_compress_kernel_both<2, false>
                                                                               : lz4_compress_kernel_both<4, false>;
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
    for (int wide = 0; wide < 2 && r == hipSuccess; ++wide)
      r = hipFuncSetAttribute(reinterpret_cast<const void*>(both_kernel_for(es, wide != 0)),
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
  sh.stride_tagged = round_up(ht_size * 3u, 16u);
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

Lz4Mode lz4_mode_from_environment()
{
  // read at every call (a getenv is nanoseconds beside a launch): the tests
  // switch shapes inside one process
  const char* e = std::getenv("HIPCOMP_LZ4_SHAPE");
  if (e && std::strcmp(e, "mix") == 0)
    return Lz4Mode::Mix;
  if (e && std::strcmp(e, "far") == 0)
    return Lz4Mode::Far;
  if (e && std::strcmp(e, "farw") == 0)
    return Lz4Mode::FarWide;
  return Lz4Mode::Auto;
}

namespace {
// Measurement knob HIPCOMP_LZ4_BOTH="near,far,slots": the far shapes run as the "both"
// kernel with that many LDS-table waves and device-table waves per workgroup and that
// many scratch slots per wave (a power of two).
struct BothGeometry
{
  uint32_t near = 0, far = 0, slots = 0;
  bool on() const { return near + far > 0; }
};
BothGeometry both_from_environment()
{
  BothGeometry g;
  const char* e = std::getenv("HIPCOMP_LZ4_BOTH");
  unsigned a = 0, b = 0, c = 0;
  if (e && std::sscanf(e, "%u,%u,%u", &a, &b, &c) == 3 && a + b >= 1 && a + b <= (unsigned)kBothMaxWavesPerGroup
      && c >= 64 && c <= 4096 && (c & (c - 1)) == 0) {
    g.near = a;
    g.far = b;
    g.slots = c;
  }
  return g;
}
} // namespace

hipError_t lz4_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, uint32_t ht_size,
    size_t batch, int elem_size, uint32_t* scratch, uint16_t* far_tables, size_t far_capacity,
    size_t max_chunk_bytes, Lz4Mode mode, hipStream_t stream)
{
  const Lz4CompressShape mix = lz4_compress_shape_mix(ht_size, batch);
  // far: as many workgroups as the chip holds and the caller's buffer has tables for
  Lz4CompressShape far = {};
  far.plain = kFarWavesPerGroup;
  far.lds_bytes = kFarWavesPerGroup * kFarScratchSlots * (uint32_t)sizeof(uint16_t);
  const uint32_t far_groups_per_cu = [] { // (measurement knob: fewer resident waves)
    const char* e = std::getenv("HIPCOMP_LZ4_FAR_GROUPS");
    const int v = e ? std::atoi(e) : 0;
    return (uint32_t)(v >= 1 && v <= kFarGroupsPerCu ? v : kFarGroupsPerCu);
  }();
  far.groups = (uint32_t)num_cus_of_current_device() * far_groups_per_cu;
  if ((size_t)far.groups * kFarWavesPerGroup > far_capacity)
    far.groups = (uint32_t)(far_capacity / kFarWavesPerGroup);
  // it needs the ticket counter, and pays once the batch is more than the mix
  // shape has in flight at once (whose waves are the faster ones)
  const bool far_possible = scratch != nullptr && far_tables != nullptr && far.groups > 0;
  if ((mode == Lz4Mode::Far || mode == Lz4Mode::FarWide) && !far_possible) // (forced by the environment)
    mode = Lz4Mode::Mix;
  if (mode == Lz4Mode::Auto
      && !(far_possible && far.groups * far.waves() > mix.groups * mix.waves() && batch > (size_t)mix.groups * mix.waves()))
    mode = Lz4Mode::Mix;
  const hipError_t raised = raise_dynamic_lds_limit();
  if (raised != hipSuccess)
    return raised;
  uint32_t* ticket = scratch;
  const uint32_t* chosen = nullptr; // the sampling kernel's counters, if it runs
  if (scratch) {
    const hipError_t e = hipMemsetAsync(scratch, 0, 4 * sizeof(uint32_t), stream);
    if (e != hipSuccess)
      return e;
    if (mode == Lz4Mode::Auto) {
      lz4_sample_kernel<<<kSampleChunks, kWave, 0, stream>>>(in_ptrs, in_bytes, (uint32_t)batch, scratch + 1);
      chosen = scratch + 1;
    }
  }
  // about 16 KiB of input per ticket, but at least 4 tickets per wave so
  // that the last ones even out the load
  auto chunks_per_ticket = [&](const Lz4CompressShape& sh) {
    uint32_t per_ticket = 1;
    const size_t all_waves = (size_t)sh.groups * sh.waves();
    while (per_ticket < 64 && (size_t)per_ticket * (max_chunk_bytes ? max_chunk_bytes : 1) < 16384
           && (size_t)per_ticket * 2 * 4 * all_waves <= batch)
      per_ticket *= 2;
    return per_ticket;
  };
  if (mode != Lz4Mode::Far && mode != Lz4Mode::FarWide) {
    // ticket == nullptr: no persistent workgroups, one chunk per wave
    const dim3 grid(ticket ? mix.groups : (unsigned)((batch + mix.waves() - 1) / mix.waves()));
    mix_kernel_for(elem_size)<<<grid, dim3(mix.waves() * kWave), mix.lds_bytes, stream>>>(
        in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, mix.tagged, mix.stride_tagged, mix.stride_plain,
        (uint32_t)batch, ticket, chunks_per_ticket(mix), chosen);
  }
  const BothGeometry both = both_from_environment();
  Lz4CompressShape bs = {};
  if (both.on()) {
    bs.tagged = both.near; // (waves with their table in LDS)
    bs.plain = both.far;
    bs.lds_bytes = both.near * (2u * (ht_size < 8 ? 8u : ht_size) + 2u * both.slots) + both.far * 2u * both.slots;
    if (bs.lds_bytes > kLdsPerCu)
      return hipErrorInvalidValue;
    set_groups(bs, batch);
    if (both.far > 0 && (size_t)bs.groups * both.far > far_capacity)
      bs.groups = (uint32_t)(far_capacity / both.far);
    if (bs.groups == 0)
      return hipErrorInvalidValue;
  }
  for (int wide = 0; wide < 2; ++wide)
    if (mode == Lz4Mode::Auto || mode == (wide ? Lz4Mode::FarWide : Lz4Mode::Far)) {
      if (both.on())
        both_kernel_for(elem_size, wide != 0)<<<dim3(bs.groups), dim3(bs.waves() * kWave), bs.lds_bytes, stream>>>(
            in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, far_tables, both.near, both.slots, (uint32_t)batch, ticket,
            chunks_per_ticket(bs), chosen);
      else
        far_kernel_for(elem_size, wide != 0)<<<dim3(far.groups), dim3(far.waves() * kWave), far.lds_bytes, stream>>>(
            in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, far_tables, (uint32_t)batch, ticket,
            chunks_per_ticket(far), chosen);
    }
  return hipSuccess;
}

void lz4_launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    const size_t* out_caps, size_t batch, uint8_t* const* out_ptrs,
    size_t* actual_bytes, hipcompStatus_t* statuses, bool write_out,
    hipStream_t stream)
{
  const dim3 grid((unsigned)((batch + kDecompWavesPerBlock - 1) / kDecompWavesPerBlock));
  const dim3 block(kWave * kDecompWavesPerBlock);
  if (write_out)
    lz4_decompress_kernel<true><<<grid, block, 0, stream>>>(
        comp_ptrs, comp_bytes, out_caps, batch, out_ptrs, actual_bytes, statuses);
  else
    lz4_decompress_kernel<false><<<grid, block, 0, stream>>>(
        comp_ptrs, comp_bytes, nullptr, batch, nullptr, actual_bytes, nullptr);
}

} // namespace hcamd
