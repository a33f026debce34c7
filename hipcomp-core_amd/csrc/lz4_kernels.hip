// lz4_kernels.hip -- gfx950 kernels of the batched LZ4 block codec.
//
// Compressed bytes are those of the reference's wave64 encoder
// (reference src/LZ4Kernels.hiph:793-969 compressStream<T>), produced by a
// different mechanism:
//
//   reference                               here
//   --------------------------------------  ---------------------------------
//   32 KiB hash table per chunk in HBM      table in LDS (ds_read_u16 /
//   (temp space), global_store_short        ds_write_b16); temp space unused
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
//   one window at a time                    match-less stretches: blocks of
//                                           windows, all LDS traffic of a block
//                                           issued back to back, decisions one
//                                           block later (walk_*)
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
#include <cstdlib>
#include <cstring>


namespace hcamd {

namespace {

constexpr uint32_t kNullOffset = 0xFFFFu;

// The walk (see walk_step): windows per block, and how many windows without a
// match in a row start it.
constexpr int kLz4WalkBlock = 4;
constexpr int kLz4WalkAfter = 2;

// An LDS address outside every workgroup's allocation (the whole LDS has 160 KiB).
constexpr uint32_t kLdsNowhere = 0x30000u;


__device__ __forceinline__ uint32_t hash_sum(uint32_t key)
{
  // reference hash() :557-561 before masking
  return __brev(key) + (key ^ 0xc375u);
}

// 8 bits of the hash that take no part in the slot number (tables have at
// most 2^14 slots): the tag of a table entry.
__device__ __forceinline__ uint32_t tag_of(uint32_t hsum)
{
  return (hsum >> 14) & 0xFFu;
}

// Write `n` in LZ4's linear small-integer code: n/255 bytes of 0xFF then
// n%255.  (reference writeLSIC :267-278)
__device__ __forceinline__ uint32_t write_lsic(gptr out, uint32_t number, int lane)
{
  const uint32_t num = number / 255u + 1u;
  const uint8_t last = (uint8_t)(number % 255u);
  // (one trip nearly always; unrolled eight times the compiler keeps eight
  // lane offsets in registers for the whole kernel)
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (uint32_t i = (uint32_t)lane; i < num; i += kWave)
    out[i] = (i + 1 < num) ? (uint8_t)0xFF : last;
  return num;
}

// One LZ4 sequence (reference writeSequenceData :665-715, token_type
// :280-351).  match_bytes == 0 marks the final, literal-only sequence whose
// token low nibble is 0xC in the reference (uint8_t(0 - 4) & 0x0f).
__device__ __forceinline__ uint32_t write_sequence(
    gptr comp, uint32_t c, cgptr lit_src, uint32_t lit_bytes,
    uint32_t match_bytes, uint32_t offset_bytes, int lane)
{
  if (lane == 0) {
    const uint32_t lh = lit_bytes >= 15 ? 15u : lit_bytes;
    const uint32_t mh = match_bytes >= 19 ? 15u : ((match_bytes - 4u) & 0x0fu);
    comp[c] = (uint8_t)((lh << 4) | mh);
  }
  ++c;
  if (lit_bytes >= 15)
    c += write_lsic(comp + c, lit_bytes - 15u, lane);
  wave_copy(comp + c, lit_src, lit_bytes, lane);
  c += lit_bytes;
  if (match_bytes > 0) {
    if (lane == 0) {
      comp[c] = (uint8_t)(offset_bytes & 0xffu);
      comp[c + 1] = (uint8_t)((offset_bytes >> 8) & 0xffu);
    }
    c += 2;
    if (match_bytes >= 19)
      c += write_lsic(comp + c, match_bytes - 19u, lane);
  }
  return c;
}

// ---------------------------------------------------------------------------
// The two LDS tables of one chunk.
//   pos[h]  element position & 0xFFFF of the entry, 0xFFFF = empty (the
//           reference's table, :157, :736)
//   tag[h]  tag_of() the word the entry was made from.  A candidate matches
//           only if its 4 bytes equal the window word, which implies equal
//           tags -- so an entry whose tag differs is rejected without
//           fetching the candidate's bytes.  Holds while a slot's position
//           names the element it was made from, i.e. for chunks of at most
//           65536 elements (`filter`); beyond that the 16-bit position may
//           alias an element 65536 further on and every candidate is fetched
//           as in the reference.
// Both tables are written by the same lanes under the same mask in the same
// lane order, so they stay entry for entry in step (ds_write_b8 and
// ds_write_b16 resolve same-address lanes alike: tests/test_hw_probes.py).
// ---------------------------------------------------------------------------
template <bool TAGS>
struct Tables
{
  static constexpr bool tags = TAGS; // a tag table exists (a launch-wide choice, lz4_launch_compress)
  uint16_t* pos;
  uint8_t* tag;     // valid only with TAGS
  uint32_t pos_lds; // LDS byte addresses of the two
  uint32_t tag_lds;
  bool filter;      // the tags may be used to reject candidates of this chunk
};

__device__ __forceinline__ uint32_t lds_addr_of(const void* p)
{
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)p;
}

// pos[hpos] = pval, tag[hpos] = tval for the lanes of `mask` (a scalar lane
// mask: no per-lane flag, no compare), as ONE instruction group.
template <class TT>
__device__ __forceinline__ void tables_store_masked(
    const TT& T, uint32_t hpos, uint32_t pval, uint32_t tval, uint64_t mask)
{
  const uint32_t pa = T.pos_lds + 2u * hpos;
  uint64_t saved;
  if (T.tags) {
    const uint32_t ta = T.tag_lds + hpos;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_and_b64 exec, %0, %1\n\t"
                 "ds_write_b16 %2, %3\n\t"
                 "ds_write_b8 %4, %5\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "s"(mask), "v"(pa), "v"(pval), "v"(ta), "v"(tval)
                 : "memory", "scc");
  } else {
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_and_b64 exec, %0, %1\n\t"
                 "ds_write_b16 %2, %3\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "s"(mask), "v"(pa), "v"(pval)
                 : "memory", "scc");
  }
}

// ---------------------------------------------------------------------------
// Hash-table insert for lanes [0, n) of the window at element position d,
// reproducing what the reference's insertHashTableWarp (:722-741) does when
// it runs 64 lanes wide:
//   n <= 31: per slot the highest lane's position is stored;
//   n >= 32: numValidThreadsToMask (:717-720) returns `int` and the 64-bit
//            match mask is kept in a `const int`, so
//            - a slot shared with lane 31 is left alone, except that lane 63
//              stores if it is in that slot;
//            - otherwise the highest lane among lanes 0..30 of the slot
//              stores;
//            - a slot that holds only lanes >= 32: they ALL execute the same
//              global_store_short and the hardware picks the survivor.
//              Measured on MI355X (tests/test_hw_probes.py): the lanes of a
//              wave are written in the order  for g in 0..3, for p in 3..0,
//              for q in 0..3: lane 16g+4q+p,  last write survives.
//
// Mechanism here: ds_write_b16 keeps the HIGHEST lane among lanes that hit
// one address (measured, same test), so the whole rule is ONE masked store
// once the window lanes sit in the physical lanes in priority order
// ("sigma order": physical lane p carries window lane sigma(p)):
//   physical lanes  0..31: window lanes >= 32 in the hardware's write order
//                          above (the reference's global_store_short);
//   physical lanes 32..63: window lanes 0..31 in natural order -- they
//                          override the first half wherever a slot also has
//                          a window lane below 32.
// In that order window lane 31 is physical lane 63 and window lane 63 is
// physical lane 19.  The walk loads its window words in sigma order straight
// from memory; the one-window path permutes them (one ds_bpermute).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sigma_of_lane(int lane)
{
  const uint32_t r = (uint32_t)lane;
  return lane < 32 ? 32u + (r & 16u) + 4u * (r & 3u) + (3u - ((r >> 2) & 3u)) : r - 32u;
}

constexpr uint64_t kSigmaLane31 = 1ull << 63; // physical lane of window lane 31
constexpr uint64_t kSigmaLane63 = 1ull << 19; // physical lane of window lane 63

// Lanes that store when the first n >= 32 window lanes are inserted; `below`
// = physical lanes whose window lane is < n, hpos in sigma order.  Also hands
// back the lanes that share window lane 31's slot.
__device__ __forceinline__ uint64_t sigma_store_mask(uint32_t hpos, uint64_t below, bool n_is_64, uint64_t& in31)
{
  const uint32_t h31 = read_lane(hpos, 63);
  in31 = wave_ballot(hpos == h31); // includes physical lane 63 itself
  return (below & ~in31) | (n_is_64 ? kSigmaLane63 : 0ull);
}

// Insert of the first n >= 32 window lanes of the window at d; word_sigma =
// the window words in sigma order.
template <class TT>
__device__ __forceinline__ void insert_sigma(
    const TT& T, uint32_t word_sigma, uint32_t d, int n, uint32_t sig, uint32_t hmask)
{
  const uint32_t hs = hash_sum(word_sigma);
  const uint32_t hp = hs & hmask;
  uint64_t in31;
  const uint64_t store = sigma_store_mask(hp, wave_ballot(sig < (uint32_t)n), n == 64, in31);
  tables_store_masked(T, hp, (d + sig) & 0xFFFFu, tag_of(hs), store);
}

// table lookups of the one-window path and the ds_bpermute that mirrors the
// slots issued back to back, ONE wait for all (left to itself the compiler
// waits for the reads first, then issues the permute)
template <class TT>
__device__ __forceinline__ void lds_lookup_with_bpermute(
    const TT& T, uint32_t hpos, int bp_addr4, uint32_t bp_data,
    uint32_t& slot_value, uint32_t& tag_value, uint32_t& bp_value)
{
  const uint32_t pa = T.pos_lds + 2u * hpos;
  if (T.tags) {
    const uint32_t ta = T.tag_lds + hpos;
    asm volatile("ds_read_u16 %0, %3\n\tds_read_u8 %1, %4\n\tds_bpermute_b32 %2, %5, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(slot_value), "=&v"(tag_value), "=&v"(bp_value)
                 : "v"(pa), "v"(ta), "v"(bp_addr4), "v"(bp_data)
                 : "memory");
  } else {
    tag_value = 0;
    asm volatile("ds_read_u16 %0, %2\n\tds_bpermute_b32 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(slot_value), "=&v"(bp_value)
                 : "v"(pa), "v"(bp_addr4), "v"(bp_data)
                 : "memory");
  }
}

// ---------------------------------------------------------------------------
// One window of the match search = 64 consecutive element positions, one per
// lane (reference :847-962), in natural lane order: the one-window path.
// ---------------------------------------------------------------------------
struct Window
{
  uint32_t d;         // first element (wave-uniform)
  int nv;             // lanes holding a position that may start a match (uniform)
  bool valid;         // lane < nv
  uint32_t word;      // the 4 bytes at element d + lane
  uint32_t hpos;      // my table slot
  uint32_t tag;       // my word's tag
  uint32_t h_old;     // what the slot held before this window
  uint32_t t_old;     // ... and its tag
  uint32_t cand;      // element the slot points to
  uint64_t probe;     // lanes whose candidate is usable: its word gets verified (uniform)
  uint32_t cand_word; // 4 bytes at cand (in flight until first use)
  uint32_t next_word; // 4 bytes at element d + nv + lane (in flight)
  uint32_t w_raw;     // marker read back from my slot: lowest lane in it
};

struct Decision
{
  bool match;
  int f;                   // first lane with a match
  uint32_t match_location; // element it matches
};

template <int S, int NVMAX>
__device__ __forceinline__ void window_begin(
    Window& W, uint32_t d, uint32_t word, uint32_t L, uint32_t hmask, int lane)
{
  constexpr uint32_t LVM = (12 + S - 1) / S;
  W.d = d;
  W.nv = min(NVMAX, (int)(L - d - LVM)); // >= 1
  W.valid = lane < W.nv;
  W.word = word;
  const uint32_t hs = hash_sum(word);
  W.hpos = hs & hmask;
  W.tag = tag_of(hs);
}

// (B) candidate from earlier windows (reference isValidHash :634-663,
// convertIdx :619-632), its 4-byte verify load, then the load of the next
// window's words: the latter is issued AFTER the verify so that waiting for the
// verify (in-order vmcnt) does not wait for it.  Both loads are unconditional
// with a clamped, always readable index so that the compiler can count them.
template <int S, class TT>
__device__ __forceinline__ void window_candidate(
    Window& W, const TT& T, cgptr in, uint32_t last_word, int lane, bool load_next)
{
  const uint32_t pos = W.d + (uint32_t)lane;
  // The slot holds the low 16 bits of an element before pos: the candidate
  // is the nearest such element, 1..65536 elements back.
  const uint32_t back = (pos - 1u - W.h_old) & 0xFFFFu; // distance - 1
  const uint32_t cand = pos - 1u - back;
  // The reference accepts any candidate within 65535 ELEMENTS and then
  // truncates the byte offset to 16 bits (:651, :954), which corrupts
  // typed-mode (S > 1) streams of chunks larger than 64 KiB.  Candidates
  // whose byte distance does not fit are rejected here; for chunks
  // <= 64 KiB this never triggers, so those stay bit-identical
  // (DESIGN.md "deliberate deviations").
  // (one ballot per compare, combined as scalars: a ballot of the combined
  // per-lane condition goes through a VGPR)
  W.probe = wave_ballot(W.h_old != kNullOffset) & wave_ballot(back < 65535u / S)
            & lanes_below<64>(W.nv);
  if (T.filter)
    W.probe &= wave_ballot(W.t_old == W.tag);
  W.cand = cand;
  const uint32_t own = min(pos, last_word);
  uint32_t at; // probe ? cand : own, straight from the scalar lane mask
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(at) : "v"(own), "v"(cand), "s"(W.probe));
  W.cand_word = load_u32_any(in + (size_t)at * S);
  // Next words: not after a window with a match (this one most likely has
  // one too, the words would be dropped, and a load in flight into a
  // register the match path wants to reuse makes that path wait for it).
  if (load_next)
    W.next_word = load_u32_any(in + (size_t)min(pos + (uint32_t)W.nv, last_word) * S);
}

// (A) in-window duplicates: lowest lane holding my word, found through the
// hash table itself (no scratch LDS).  Every valid lane posts its lane id into
// its own table slot with the lanes in REVERSED order (`pr` = slot and valid
// flag of the mirrored lane), so that ds_write_b16's "highest lane wins"
// leaves the LOWEST window lane of each slot; reading the slot back names that
// lane.  If it holds my word it is exactly min{u : word_u == word_t};
// otherwise two different words share the slot and the lane is settled by the
// exact fallback in window_decide.  The markers are taken off the table again
// by window_insert_first (the tag table is not touched by any of this).
template <class TT>
__device__ __forceinline__ void window_markers(
    Window& W, const TT& T, uint32_t pr, uint32_t rev_lane)
{
  lds_lane_exchange_fence();
  if (pr & 0x80000000u)
    T.pos[pr & 0x7FFFFFFFu] = (uint16_t)rev_lane;
  lds_lane_exchange_fence();
  W.w_raw = T.pos[W.hpos];
  lds_lane_exchange_fence();
}

__device__ __forceinline__ uint32_t window_winner(const Window& W, int lane)
{
  // an invalid lane names itself, so it is neither duplicate nor unresolved
  return W.valid ? W.w_raw : (uint32_t)lane;
}

// First lane with an equal lower lane (nv if none) and that lower lane.
// nw = word of window_winner's lane (one ds_bpermute, issued by the caller).
template <int NVMAX>
__device__ __forceinline__ void window_first_duplicate(
    const Window& W, uint32_t nw, int lane, int& f, uint32_t& mlane)
{
  const uint64_t vmask = lanes_below<NVMAX>(W.nv);
  const uint32_t w = window_winner(W, lane);
  // masks are combined as scalars: each ballot is one v_cmp
  const uint64_t eqmask = wave_ballot(nw == W.word);
  const uint64_t dupmask = eqmask & wave_ballot(w != (uint32_t)lane);
  const uint64_t unres = vmask & ~eqmask;

  f = dupmask ? __builtin_ctzll(dupmask) : W.nv;
  mlane = read_lane(w, f & 63);
  uint64_t U = unres & lanes_below<NVMAX>(f);
  if (__builtin_expect(U != 0, 0)) {
    do {
      const int u = __builtin_ctzll(U);
      U &= U - 1;
      const uint32_t v = read_lane(W.word, u);
      const uint64_t m = wave_ballot(W.word == v) & vmask;
      const int lo = __builtin_ctzll(m);
      if (lo != u) {
        f = u;
        mlane = (uint32_t)lo;
        break;
      }
    } while (U);
  }
}

// Lanes below f whose table candidate holds the lane's word.
template <int NVMAX>
__device__ __forceinline__ uint64_t window_table_matches(const Window& W, int f)
{
  return wave_ballot(W.cand_word == W.word) & W.probe & lanes_below<NVMAX>(f);
}

// The earliest lane with a verified table candidate wins over the first
// in-window duplicate at f (reference :896-923).
__device__ __forceinline__ Decision window_settle(const Window& W, int f, uint32_t mlane, uint64_t tmask)
{
  const bool in_window = f < W.nv;
  if (tmask)
    f = __builtin_ctzll(tmask);
  const uint32_t tcand = read_lane(W.cand, f & 63);
  Decision D;
  D.match = tmask != 0 || in_window;
  D.f = f;
  D.match_location = tmask ? tcand : W.d + mlane; // reference :925-956
  return D;
}

template <int NVMAX>
__device__ __forceinline__ Decision window_decide(const Window& W, uint32_t nw, int lane)
{
  int f;
  uint32_t mlane;
  window_first_duplicate<NVMAX>(W, nw, lane, f, mlane);
  return window_settle(W, f, mlane, window_table_matches<NVMAX>(W, f));
}

// Table state "only the first f lanes of W were inserted", from the state in
// which W's slots hold its markers.
template <int NVMAX, class TT>
__device__ __forceinline__ void window_insert_first(
    const Window& W, const TT& T, int f, int perm_addr4, uint32_t sig, uint32_t hmask, int lane)
{
  if (W.valid)
    T.pos[W.hpos] = (uint16_t)W.h_old;
  lds_lane_exchange_fence();
  if (f >= 32) {
    const uint32_t ws = (uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr4, (int)W.word);
    insert_sigma(T, ws, W.d, f, sig, hmask);
  } else if (lane < f) {
    // n <= 31: the highest lane of a slot stores -- natural lane order
    T.pos[W.hpos] = (uint16_t)(W.d + (uint32_t)lane);
    if (T.tags)
      T.tag[W.hpos] = (uint8_t)W.tag;
  }
  lds_lane_exchange_fence();
}

// First mismatching element between the strings at elements `prev` and `pos`
// (reference lengthOfMatch :592-617), compared 4 bytes per lane per step.
template <int S>
__device__ __forceinline__ uint32_t match_length(
    cgptr in, uint32_t prev, uint32_t pos, uint32_t limit, int lane)
{
  cgptr a = in + (size_t)prev * S;
  cgptr b = in + (size_t)pos * S;
  const uint32_t limit_bytes = limit * S;
  for (uint32_t j = 0; j < limit_bytes; j += 4 * kWave) {
    const uint32_t i = j + 4u * (uint32_t)lane;
    uint32_t diff_at = 4; // byte index of first difference inside my dword
    if (j + 4 * kWave <= limit_bytes) {
      // (wave-uniform, the usual case: every lane's dword lies inside the
      // limit -- no per-lane branches)
      const uint32_t x = load_u32_any(a + i) ^ load_u32_any(b + i);
      diff_at = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
    } else if (i + 4 <= limit_bytes) {
      const uint32_t x = load_u32_any(a + i) ^ load_u32_any(b + i);
      if (x)
        diff_at = (uint32_t)__builtin_ctz(x) >> 3;
    } else if (i < limit_bytes) {
      // tail shorter than a dword: byte loads, stop at the limit
      const uint32_t rem = limit_bytes - i;
      diff_at = rem; // "mismatch" at the limit ends the search
      for (uint32_t k = 0; k < rem; ++k)
        if (a[i + k] != b[i + k]) {
          diff_at = k;
          break;
        }
    } else {
      diff_at = 0; // past the limit
    }
    const uint64_t m = wave_ballot(diff_at < 4);
    if (m) {
      const int l = __builtin_ctzll(m);
      const uint32_t byte_idx = j + 4u * (uint32_t)l + read_lane(diff_at, l);
      const uint32_t mb = byte_idx < limit_bytes ? byte_idx : limit_bytes;
      return mb / S;
    }
  }
  return limit;
}

// ---------------------------------------------------------------------------
// The walk: a stretch of windows without a match (incompressible data) taken
// a BLOCK of G full windows at a time, in sigma order.  Nothing in a window's
// table traffic depends on what the table returns -- lookup, insert (on the
// guess that the window has no match) and the read-back of the insert depend
// on the window's words only -- so the 5 G LDS operations of a block are
// issued back to back (LDS operations of a wave execute in order) and cost
// their issue slots, not their latency.  What the lookups return is looked at
// in later steps (walk_step):
//   * a lane whose read-back is not its own position shares its slot with
//     another lane of the window: the only lanes that can be one half of an
//     in-window duplicate (equal words hash alike and only one lane per slot
//     reads itself back).  Nearly always there is none; else exact compare.
//   * a lane whose slot held an entry with its word's tag: its candidate is
//     fetched (about one lane in four windows on random data, instead of a
//     64-line gather per window) and compared.
// A window that does have a match ends the walk: the inserts of all younger
// windows and its own are taken back (newest first, from the kept old slot
// contents) and the one-window path redoes it.
//
// A lone wave on its SIMD pays ~4 cycles per instruction and ~30 for every
// trip of a value from the vector to the scalar unit and back (a ballot
// combined by s_and and consumed by v_cndmask, an exec mask made from a
// compare: scripts/probes/latency_table.hip), and the walk has no second wave
// to hide that behind.  So its per-lane decisions stay per lane (v_cmp into
// vcc, v_cndmask out of it; a lane that must not store writes to a scratch
// slot behind the tables instead of being masked off) and are gathered into
// ONE scalar test per block (walk_decide).  Lane constants do the rest: the
// lanes behind the last valid window lane (3 for bytes, 1 for shorts) walk as
// copies of window lane 31 -- the one lane that never stores -- and window
// lane 63 of a 64-lane window compares with a slot number no lane can have.
// ---------------------------------------------------------------------------
template <int G>
struct WalkBlock
{
  uint32_t word[G];   // sigma order
  uint32_t hpos[G];
  uint32_t tag[G];
  uint32_t pos[G];    // my element
  // What the tables held / hold.  Read by inline asm (walk_tables) so that the
  // compiler takes the registers as the full 32-bit values ds_read_u16 /
  // ds_read_u8 make of them (through its own loads it masks each of them again,
  // 8 to 12 instructions a step) -- and therefore waited for by hand:
  // lgkmcnt(0) at the start of every step, of the roll-back and of the drain.
  uint32_t h_old[G];  // my slot before the window
  uint32_t t_old[G];  // ... and its tag
  uint32_t rb[G];     // my slot after the window's insert
  uint32_t at[G];     // element whose 4 bytes were fetched as my candidate's; pos = none
  uint32_t sharer;    // != 0: I share window lane 31's slot in some window of the block
};

// per-lane constants of the walk
struct WalkLanes
{
  uint32_t sig;        // my window lane (sigma order; lanes behind the last valid one: 31)
  uint32_t never;      // OR-ed into my slot number for the "shares lane 31's slot" test: window
                       // lane 63 of a 64-lane window stores whatever that test says
  uint32_t counts;     // ~0: my read-back / sharer flag counts (valid lane other than window lane 31)
  uint32_t scratch_pos, scratch_tag; // LDS addresses a lane that does not store writes to
  uint64_t validc;     // the valid lanes as a mask (slow paths)
};

// The walk's loads are issued and awaited by hand, and they land in
// ACCUMULATION registers (AGPRs), which the compiler does not allocate:
//  * vector memory operations complete in issue order and `s_waitcnt
//    vmcnt(N)` waits for all but the N youngest; a step issues exactly G
//    candidate loads, then G word loads, so the counts are known.  The
//    compiler's own counting gives up on the walk's control flow (it waited
//    for the words loaded two blocks ahead at every step: a trip to HBM);
//  * a value in flight must not be touched before its wait.  In compiler-
//    allocated registers that cannot be promised (it may move a register
//    that an asm statement is about to wait for); an AGPR named in the asm
//    text is out of its reach.  After the wait the value is read into a
//    normal register (v_accvgpr_read_b32) and is the compiler's from there.
// Slots: words of block b in a[4 (b % 3) ..], its candidate words in
// a[12 + 4 (b % 3) ..].
#define HC_WALK_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", \
                      "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23"

template <int A>
__device__ __forceinline__ void agpr_load_u32(cgptr base, uint32_t byte_off)
{
  asm volatile("global_load_dword a[%2], %0, %1" : : "v"(byte_off), "s"(base), "n"(A) : HC_WALK_AGPRS);
}

// waits until at most N younger loads are in flight, then a[A0 .. A0+3] -> r
template <int A0, int N>
__device__ __forceinline__ void agpr_take4(uint32_t (&r)[4])
{
  asm volatile("s_waitcnt vmcnt(%4)\n\t"
               "v_accvgpr_read_b32 %0, a[%5]\n\t"
               "v_accvgpr_read_b32 %1, a[%6]\n\t"
               "v_accvgpr_read_b32 %2, a[%7]\n\t"
               "v_accvgpr_read_b32 %3, a[%8]"
               : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3])
               : "n"(N), "n"(A0), "n"(A0 + 1), "n"(A0 + 2), "n"(A0 + 3));
}

// issues the loads of the words of the G windows from element d0 on into
// slot R; CLAMP: the block may reach past the last full window (it is loaded
// ahead of knowing), keep the loads readable
template <int S, int G, bool CLAMP, int R>
__device__ __forceinline__ void walk_load(cgptr in, uint32_t d0, uint32_t sig, uint32_t last_word)
{
  static_assert(G == 4, "slot layout");
  constexpr int NVMAX = kWave - 3 / S;
  uint32_t e[G];
#pragma unroll
  for (int k = 0; k < G; ++k) {
    e[k] = d0 + (uint32_t)(k * NVMAX) + sig;
    if (CLAMP)
      e[k] = min(e[k], last_word);
  }
  agpr_load_u32<4 * R + 0>(in, e[0] * (uint32_t)S);
  agpr_load_u32<4 * R + 1>(in, e[1] * (uint32_t)S);
  agpr_load_u32<4 * R + 2>(in, e[2] * (uint32_t)S);
  agpr_load_u32<4 * R + 3>(in, e[3] * (uint32_t)S);
}

// table traffic of a block: lookup, insert, read-back per window, no waits
template <int S, int G, class TT>
__device__ __forceinline__ void walk_tables(
    WalkBlock<G>& B, const TT& T, uint32_t d0, const WalkLanes& W, uint32_t hmask)
{
  constexpr int NVMAX = kWave - 3 / S;
  B.sharer = 0;
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const uint32_t hs = hash_sum(B.word[k]);
    const uint32_t hp = hs & hmask;
    B.hpos[k] = hp;
    B.tag[k] = tag_of(hs);
    B.pos[k] = d0 + (uint32_t)(k * NVMAX) + W.sig;
    const uint32_t pa = T.pos_lds + 2u * hp, ta = T.tag_lds + hp;
    if (TT::tags)
      asm volatile("ds_read_u16 %0, %2\n\tds_read_u8 %1, %3"
                   : "=&v"(B.h_old[k]), "=&v"(B.t_old[k]) : "v"(pa), "v"(ta) : "memory");
    else
      asm volatile("ds_read_u16 %0, %1" : "=&v"(B.h_old[k]) : "v"(pa) : "memory");
    // the insert rule (insert_sigma) per lane: a lane in window lane 31's slot
    // stores to the scratch slot instead and notes that it shares (window
    // lane 63 of a 64-lane window stores in any case but is a sharer like any
    // other).  ds_write_b16 takes the low half of the position.
    const uint32_t h31 = read_lane(hp, 63);
    uint32_t pa_st, ta_st;
    uint32_t mine = hp;
    if (NVMAX == 64) {
      mine = hp | W.never;
      asm volatile("v_cmp_ne_u32_e32 vcc, %1, %2\n\t"
                   "v_cndmask_b32_e32 %0, 1, %0, vcc"
                   : "+v"(B.sharer) : "s"(h31), "v"(hp) : "vcc");
    }
    uint32_t unused = 0;
    uint32_t& sharer = NVMAX == 64 ? unused : B.sharer;
    if (TT::tags)
      asm volatile("v_cmp_ne_u32_e32 vcc, %3, %4\n\t"
                   "v_cndmask_b32_e32 %0, %5, %6, vcc\n\t"
                   "v_cndmask_b32_e32 %1, %7, %8, vcc\n\t"
                   "v_cndmask_b32_e32 %2, 1, %2, vcc\n\t"
                   "ds_write_b16 %0, %9\n\t"
                   "ds_write_b8 %1, %10"
                   : "=&v"(pa_st), "=&v"(ta_st), "+v"(sharer)
                   : "s"(h31), "v"(mine), "v"(W.scratch_pos), "v"(pa), "v"(W.scratch_tag), "v"(ta),
                     "v"(B.pos[k]), "v"(B.tag[k])
                   : "vcc", "memory");
    else
      asm volatile("v_cmp_ne_u32_e32 vcc, %2, %3\n\t"
                   "v_cndmask_b32_e32 %0, %4, %5, vcc\n\t"
                   "v_cndmask_b32_e32 %1, 1, %1, vcc\n\t"
                   "ds_write_b16 %0, %6"
                   : "=&v"(pa_st), "+v"(sharer)
                   : "s"(h31), "v"(mine), "v"(W.scratch_pos), "v"(pa), "v"(B.pos[k])
                   : "vcc", "memory");
    (void)ta_st;
    asm volatile("ds_read_u16 %0, %1" : "=&v"(B.rb[k]) : "v"(pa) : "memory");
  }
}

// What the lookups of a block returned: the candidates that cannot be ruled
// out are fetched.  ONE load per window whatever the data (lanes without a
// candidate re-read their own window word, a line that is in L1), so that the
// loads in flight can be counted.  The table entries are what walk_tables'
// own ds_read_u16 / ds_read_u8 left in the registers (zero-extended by the
// hardware, waited for at the start of the step).
// SMALL: the chunk has at most 65536 BYTES, so a slot's position is the
// candidate itself and its distance always fits.
template <int S, int G, int R, bool SMALL, class TT>
__device__ __forceinline__ void walk_probe(WalkBlock<G>& B, const TT& T, cgptr in)
{
  const uint32_t tag_bits = T.filter ? 0xFFu : 0u; // no filter: every tag "equal"
#pragma unroll
  for (int k = 0; k < G; ++k) {
    uint32_t at;
    if (SMALL) {
      if (TT::tags)
        asm volatile("v_xor_b32_e32 %0, %3, %4\n\t"
                     "v_and_b32_e32 %0, %5, %0\n\t"
                     "v_cmp_eq_u32_e32 vcc, 0, %0\n\t"
                     "v_cndmask_b32_e32 %0, %1, %2, vcc\n\t"
                     "v_cmp_ne_u32_e32 vcc, 0xffff, %2\n\t"
                     "v_cndmask_b32_e32 %0, %1, %0, vcc"
                     : "=&v"(at)
                     : "v"(B.pos[k]), "v"(B.h_old[k]), "v"(B.t_old[k]), "v"(B.tag[k]), "s"(tag_bits)
                     : "vcc");
      else
        asm volatile("v_cmp_ne_u32_e32 vcc, 0xffff, %2\n\t"
                     "v_cndmask_b32_e32 %0, %1, %2, vcc"
                     : "=&v"(at)
                     : "v"(B.pos[k]), "v"(B.h_old[k])
                     : "vcc");
    } else {
      // candidate = nearest element before mine with the slot's low 16 bits,
      // usable if its byte distance fits 16 bits (see window_candidate)
      uint32_t back;
      asm volatile("v_sub_u32_e32 %1, %2, %3\n\t"
                   "v_add_u32_e32 %1, -1, %1\n\t"
                   "v_and_b32_e32 %1, 0xffff, %1\n\t"
                   "v_sub_u32_e32 %0, %2, %1\n\t"
                   "v_add_u32_e32 %0, -1, %0\n\t"
                   "v_cmp_gt_u32_e32 vcc, %4, %1\n\t"
                   "v_cndmask_b32_e32 %0, %2, %0, vcc\n\t"
                   "v_cmp_ne_u32_e32 vcc, 0xffff, %3\n\t"
                   "v_cndmask_b32_e32 %0, %2, %0, vcc"
                   : "=&v"(at), "=&v"(back)
                   : "v"(B.pos[k]), "v"(B.h_old[k]), "s"(65535u / S)
                   : "vcc");
      if (TT::tags) {
        uint32_t tdiff;
        asm volatile("v_xor_b32_e32 %1, %3, %4\n\t"
                     "v_and_b32_e32 %1, %5, %1\n\t"
                     "v_cmp_eq_u32_e32 vcc, 0, %1\n\t"
                     "v_cndmask_b32_e32 %0, %2, %0, vcc"
                     : "+v"(at), "=&v"(tdiff)
                     : "v"(B.pos[k]), "v"(B.t_old[k]), "v"(B.tag[k]), "s"(tag_bits)
                     : "vcc");
      }
    }
    B.at[k] = at;
  }
  agpr_load_u32<12 + 4 * R + 0>(in, B.at[0] * (uint32_t)S);
  agpr_load_u32<12 + 4 * R + 1>(in, B.at[1] * (uint32_t)S);
  agpr_load_u32<12 + 4 * R + 2>(in, B.at[2] * (uint32_t)S);
  agpr_load_u32<12 + 4 * R + 3>(in, B.at[3] * (uint32_t)S);
}

// first window of the block that has a match, G if none
template <int S, int G>
__device__ __forceinline__ int walk_decide(
    const WalkBlock<G>& B, const uint32_t (&cand_word)[G], const WalkLanes& W)
{
  static_assert(G == 4, "operand lists below");
  // per lane: miss = 0 iff one of my candidates holds my word (a lane without
  // a candidate has read its own bytes: 1); odd != 0 iff one of my read-backs
  // is not my position or I share window lane 31's slot
  uint32_t miss, odd, t0, t1;
  asm volatile("v_xor_b32_e32 %0, %4, %5\n\t"
               "v_cmp_ne_u32_e32 vcc, %6, %7\n\t"
               "v_cndmask_b32_e32 %0, 1, %0, vcc\n\t"
               "v_xor_b32_e32 %2, %8, %9\n\t"
               "v_cmp_ne_u32_e32 vcc, %10, %11\n\t"
               "v_cndmask_b32_e32 %2, 1, %2, vcc\n\t"
               "v_xor_b32_e32 %3, %12, %13\n\t"
               "v_cmp_ne_u32_e32 vcc, %14, %15\n\t"
               "v_cndmask_b32_e32 %3, 1, %3, vcc\n\t"
               "v_min3_u32 %0, %0, %2, %3\n\t"
               "v_xor_b32_e32 %2, %16, %17\n\t"
               "v_cmp_ne_u32_e32 vcc, %18, %19\n\t"
               "v_cndmask_b32_e32 %2, 1, %2, vcc\n\t"
               "v_min_u32_e32 %0, %0, %2\n\t"
               "v_xor_b32_e32 %1, %20, %7\n\t"
               "v_xor_b32_e32 %2, %21, %11\n\t"
               "v_xor_b32_e32 %3, %22, %15\n\t"
               "v_or3_b32 %1, %1, %2, %3\n\t"
               "v_xor_b32_e32 %2, %23, %19\n\t"
               "v_or_b32_e32 %1, %1, %2\n\t"
               "v_and_b32_e32 %1, 0xffff, %1"
               : "=&v"(miss), "=&v"(odd), "=&v"(t0), "=&v"(t1)
               : "v"(cand_word[0]), "v"(B.word[0]), "v"(B.at[0]), "v"(B.pos[0]),
                 "v"(cand_word[1]), "v"(B.word[1]), "v"(B.at[1]), "v"(B.pos[1]),
                 "v"(cand_word[2]), "v"(B.word[2]), "v"(B.at[2]), "v"(B.pos[2]),
                 "v"(cand_word[3]), "v"(B.word[3]), "v"(B.at[3]), "v"(B.pos[3]),
                 "v"(B.rb[0]), "v"(B.rb[1]), "v"(B.rb[2]), "v"(B.rb[3])
               : "vcc");
  // the one trip to the scalar unit of the block
  const uint64_t hits = wave_ballot(miss == 0);
  const uint64_t odds = wave_ballot(((odd | B.sharer) & W.counts) != 0);
  if (__builtin_expect((hits | odds) == 0, 1))
    return G;
  // Something to look at -- in about a third of the blocks of incompressible
  // data, because two of a window's 61 lanes hash to one slot (11 % of the
  // windows), so this path is kept short: only what the flags call for.
  // (All the lane masks first -- independent vector compares, one trip to the
  // scalar side for the lot -- then the loops.)
  const uint64_t validc = W.validc;
  const bool with31 = wave_ballot((B.sharer & W.counts) != 0) != 0; // a lane shares window lane 31's slot (rare)
  // a lane that does not read back its own insert shares its slot; window lane
  // 31 never stores (for it the test says nothing), it takes part iff another
  // lane is in its slot
  uint64_t U[G];
#pragma unroll
  for (int k = 0; k < G; ++k)
    U[k] = wave_ballot(((B.rb[k] ^ B.pos[k]) & 0xFFFFu) != 0) & validc & ~kSigmaLane31;
  if (__builtin_expect(with31, 0)) {
#pragma unroll
    for (int k = 0; k < G; ++k)
      if (wave_ballot(B.hpos[k] == read_lane(B.hpos[k], 63)) & validc & ~kSigmaLane31)
        U[k] |= kSigmaLane31;
  }
  // exact: does a slot sharer hold the word of another lane?
#define HC_SHARERS_OF(k)                                             \
  for (uint64_t u_k = U[k]; u_k != 0; u_k &= u_k - 1) {              \
    const uint32_t v = read_lane(B.word[k], __builtin_ctzll(u_k));   \
    const uint64_t m = wave_ballot(B.word[k] == v) & validc;         \
    if (m & (m - 1))                                                 \
      return k;                                                      \
  }
  if (__builtin_expect(hits == 0, 1)) { // the usual case written out: no candidate holds its lane's word
    HC_SHARERS_OF(0)
    HC_SHARERS_OF(1)
    HC_SHARERS_OF(2)
    HC_SHARERS_OF(3)
    return G;
  }
#pragma unroll
  for (int k = 0; k < G; ++k) {
    if (wave_ballot(cand_word[k] == B.word[k] && B.at[k] != B.pos[k]) & validc)
      return k;
    HC_SHARERS_OF(k)
  }
#undef HC_SHARERS_OF
  return G;
}

// takes the inserts of windows [from, G) of the block off the tables again,
// newest first
template <int G, class TT>
__device__ __forceinline__ void walk_undo(const WalkBlock<G>& B, const TT& T, int from, uint64_t validc)
{
#pragma unroll
  for (int k = G - 1; k >= 0; --k)
    if (k >= from)
      tables_store_masked(T, B.hpos[k], B.h_old[k], B.t_old[k], validc);
}

// One step of the walk works on three blocks, each in another stage:
//   p1  (one block back)   its lookups -- a whole step old -- are read and its
//                          candidates asked for (walk_probe);
//   cur                    its words, asked for two steps ago, go through the
//                          tables (walk_tables);
//   p2  (two blocks back)  its candidate words, asked for a whole step ago,
//                          decide whether the walk goes on (walk_decide).
// Order in a step: candidates of p1 asked for, word loads for the block two
// ahead, tables of cur, decision about p2 -- so nothing is used before a whole
// step has passed since it was asked for, and (vector memory operations
// complete in issue order) the candidate words of p2 wait for no load younger
// than the words asked for TWO steps ago: a step issues G candidate loads,
// then G word loads, so behind the words of `cur` WORDS_YOUNGER = 4 G loads
// have been issued (3 G in the first step of a walk) and behind the candidate
// words of p2 3 G.
// Returns the window of p2 that has a match (the tables are then back in the
// state before that window), G if none (or no p2 yet).
template <int S, int G, int RC, int WORDS_YOUNGER, bool HAVE_P2, bool SMALL, class TT>
__device__ __forceinline__ int walk_step(
    WalkBlock<G>& cur, WalkBlock<G>& p1, WalkBlock<G>& p2, const TT& T, cgptr in, uint32_t d_cur,
    const WalkLanes& W, uint32_t hmask, uint32_t last_word)
{
  constexpr int NVMAX = kWave - 3 / S;
  constexpr int R1 = (RC + 2) % 3; // slot of p1, and of the block two ahead of cur
  constexpr int R2 = (RC + 1) % 3; // slot of p2
  // The LDS operations of the step before (long done) are taken off the
  // counter here, in one instruction: it holds 15, a step issues 12 to 20, and
  // the compiler otherwise keeps it in range with a wait in front of every LDS
  // operation of this step.
  __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
  walk_probe<S, G, R1, SMALL>(p1, T, in);
  walk_load<S, G, true, R1>(in, d_cur + (uint32_t)(2 * G * NVMAX), W.sig, last_word);
  agpr_take4<4 * RC, WORDS_YOUNGER>(cur.word);
  walk_tables<S, G>(cur, T, d_cur, W, hmask);
  if (!HAVE_P2)
    return G;
  uint32_t cand_word[G];
  agpr_take4<12 + 4 * R2, 3 * G>(cand_word);
  const int j = walk_decide<S, G>(p2, cand_word, W);
  if (__builtin_expect(j < G, 0)) {
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): what cur's lookups of this step read
    walk_undo<G>(cur, T, 0, W.validc);
    walk_undo<G>(p1, T, 0, W.validc);
    walk_undo<G>(p2, T, j, W.validc);
  }
  return j;
}

// End of a walk: x (at element dx, slot RX) has been through the tables, y --
// the block before it, if any -- has had its candidates asked for.  Decides
// both, oldest first.  Returns the element of the first window with a match
// (the tables are back in the state before it), or the element behind x.
template <int S, int G, int RX, bool SMALL, class TT>
__device__ __forceinline__ uint32_t walk_drain(
    WalkBlock<G>& x, WalkBlock<G>& y, bool have_y, const TT& T, cgptr in, uint32_t dx,
    const WalkLanes& W, bool& match)
{
  constexpr int NVMAX = kWave - 3 / S;
  constexpr int RY = (RX + 2) % 3;
  __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): x's lookups
  walk_probe<S, G, RX, SMALL>(x, T, in);
  uint32_t cw[G];
  if (have_y) {
    agpr_take4<12 + 4 * RY, 0>(cw);
    const int j = walk_decide<S, G>(y, cw, W);
    if (j < G) {
      walk_undo<G>(x, T, 0, W.validc);
      walk_undo<G>(y, T, j, W.validc);
      match = true;
      return dx - (uint32_t)(G * NVMAX) + (uint32_t)(j * NVMAX);
    }
  }
  agpr_take4<12 + 4 * RX, 0>(cw);
  const int j = walk_decide<S, G>(x, cw, W);
  if (j < G)
    walk_undo<G>(x, T, j, W.validc);
  match = j < G;
  return dx + (uint32_t)(j * NVMAX);
}

// The walk from element d on (two blocks of full windows lie ahead).  Three
// blocks of registers rotate through the stages of walk_step.  Returns the
// element where it ended: the window there has a match (`match`; the tables are
// in the state before it) or too few full windows are left for another block.
template <int S, int G, bool SMALL, class TT>
__device__ __forceinline__ uint32_t walk_run(
    const TT& T, cgptr in, uint32_t d, uint32_t L, const WalkLanes& WL, uint32_t hmask, uint32_t last_word,
    bool& match)
{
  constexpr uint32_t LVM = (12 + S - 1) / S;
  constexpr int NVMAX = kWave - 3 / S;
  WalkBlock<G> A, B, C;
  uint32_t da = d; // first element of the newest block that has been through the tables
  match = false;
  walk_load<S, G, true, 0>(in, da, WL.sig, last_word);
  walk_load<S, G, true, 1>(in, da + (uint32_t)(G * NVMAX), WL.sig, last_word);
  walk_load<S, G, true, 2>(in, da + (uint32_t)(2 * G * NVMAX), WL.sig, last_word);
  agpr_take4<0, 2 * G>(A.word);
  walk_tables<S, G>(A, T, da, WL, hmask);
  // CUR takes the block behind the newest one (it has to be made of full
  // windows), P1 is the newest one, P2 the one before it
#define HC_WALK_STEP(CUR, P1, P2, RC, YOUNGER, HAVE_P1, HAVE_P2)                            \
  {                                                                                         \
    const uint32_t dn = da + (uint32_t)(G * NVMAX);                                         \
    if ((int)(L - dn - LVM) < G * NVMAX)                                                    \
      return walk_drain<S, G, (RC + 2) % 3, SMALL>(P1, P2, HAVE_P1, T, in, da, WL, match);  \
    const int j = walk_step<S, G, RC, YOUNGER, HAVE_P2, SMALL>(CUR, P1, P2, T, in, dn, WL,  \
                                                               hmask, last_word);           \
    if (j < G) {                                                                            \
      match = true;                                                                         \
      return da - (uint32_t)(G * NVMAX) + (uint32_t)(j * NVMAX);                            \
    }                                                                                       \
    da = dn;                                                                                \
  }
  // (the first two steps have fewer loads behind them and nothing to decide yet)
  HC_WALK_STEP(B, A, C, 1, 3 * G, false, false)
  HC_WALK_STEP(C, B, A, 2, 4 * G, true, true)
#undef HC_WALK_STEP
  // The steady state.  It is left for the drain through ONE exit behind the
  // loop (`rot` = the step that did not run): with the drain inside the loop
  // the compiler joined its path with the loop's and paid for the join with 21
  // register copies per step on the loop's path.
#define HC_WALK_STEP(CUR, P1, P2, RC)                                                       \
  {                                                                                         \
    const uint32_t dn = da + (uint32_t)(G * NVMAX);                                         \
    if ((int)(L - dn - LVM) < G * NVMAX) {                                                  \
      rot = RC;                                                                             \
      break;                                                                                \
    }                                                                                       \
    jm = walk_step<S, G, RC, 4 * G, true, SMALL>(CUR, P1, P2, T, in, dn, WL, hmask,         \
                                                 last_word);                                \
    if (jm < G) {                                                                           \
      rot = 3;                                                                              \
      break;                                                                                \
    }                                                                                       \
    da = dn;                                                                                \
  }
  int rot, jm = G;
  for (;;) {
    HC_WALK_STEP(A, C, B, 0)
    HC_WALK_STEP(B, A, C, 1)
    HC_WALK_STEP(C, B, A, 2)
  }
#undef HC_WALK_STEP
  if (rot == 3) { // a window of the block two behind the newest one has a match
    match = true;
    return da - (uint32_t)(G * NVMAX) + (uint32_t)(jm * NVMAX);
  }
  if (rot == 0)
    return walk_drain<S, G, 2, SMALL>(C, B, true, T, in, da, WL, match);
  if (rot == 1)
    return walk_drain<S, G, 0, SMALL>(A, C, true, T, in, da, WL, match);
  return walk_drain<S, G, 1, SMALL>(B, A, true, T, in, da, WL, match);
}

// The sequence that ends with the match D found in the window at element wd
// (words `word`): literals from token_start, match, offset (reference
// writeSequenceData :665-715).  Returns the new output cursor in c and the
// element after the match in d_after.
template <int S>
__device__ __forceinline__ void emit_match(
    gptr out, uint32_t& c, cgptr in, uint32_t token_start, uint32_t wd, uint32_t word,
    const Decision& D, uint32_t L, int lane, uint32_t& d_after)
{
  constexpr uint32_t MEL = (5 + S - 1) / S; // min ending literals, elements
  const uint32_t mpos = wd + (uint32_t)D.f;
  const uint32_t off_elems = (mpos - D.match_location) & 0xFFFFu;
  const uint32_t lit = mpos - token_start;
  const uint32_t ml = match_length<S>(in, D.match_location, mpos, L - mpos - MEL, lane);
  const uint32_t lit_bytes = lit * S, match_bytes = ml * S;
  const uint32_t offset_bytes = (off_elems * S) & 0xFFFFu;
  if (token_start == wd && lit_bytes < 15 && match_bytes < 19) {
    // Fast path: the whole sequence started in this window and is short, so
    // its literal bytes are the low bytes of the lanes' window words: token,
    // literals and offset leave as byte stores straight from registers.
    // Byte i of the sequence: 0 = token, 1..lit_bytes = literals, then
    // offset lo, hi.
    const uint32_t i = (uint32_t)lane;
    const uint32_t li = i - 1; // literal byte index
    const uint32_t src = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((li / S) * 4u), (int)word);
    uint32_t bt = (src >> (8u * (li % S))) & 0xFFu;
    if (i == 0)
      bt = (lit_bytes << 4) | ((match_bytes - 4u) & 0x0Fu);
    else if (i == lit_bytes + 1)
      bt = offset_bytes & 0xFFu;
    else if (i == lit_bytes + 2)
      bt = offset_bytes >> 8;
    if (i < lit_bytes + 3)
      out[c + i] = (uint8_t)bt;
    c += lit_bytes + 3;
  } else {
    c = write_sequence(out, c, in + (size_t)token_start * S, lit_bytes, match_bytes,
                       offset_bytes, lane);
  }
  d_after = token_start + lit + ml;
}

// Next chunk number for this wave: one atomic by lane 0, result wave-uniform.
// Written as one asm statement on purpose.  In C++ an `if (lane == 0)
// atomicAdd` at the loop head sits back to back with the `if (lane == 0)`
// store that ends the previous chunk; the compiler threaded the two together
// and `v_readfirstlane` then ran with lane 0 split off (an endless loop).  An
// unconditional atomic with per-lane addends (1, 0, 0, ...) avoids that but
// becomes a 64-step serial scan in the compiler's atomic optimizer -- a third
// of the time of a 1 KiB chunk.  The statement narrows exec to lane 0 of the
// lanes it was entered with and puts it back (scratch SGPRs are the
// compiler's choice); it is only ever reached with all 64 lanes active, the
// small-chunk GPU tests (many tickets per wave) are its regression test.
__device__ __forceinline__ uint32_t take_ticket(uint32_t* ticket, uint32_t count)
{
  // (offset and addend are made inside the statement from scalars: as vector
  // operands they would each hold a register for the whole kernel)
  uint32_t t, addend, zero;
  uint64_t saved;
  asm volatile("s_mov_b64 %3, exec\n\t"
               "s_and_b64 exec, %3, 1\n\t"
               "v_mov_b32_e32 %1, 0\n\t"
               "v_mov_b32_e32 %2, %4\n\t"
               "global_atomic_add %0, %1, %2, %5 sc0\n\t"
               "s_waitcnt vmcnt(0)\n\t"
               "s_mov_b64 exec, %3"
               : "=&v"(t), "=&v"(zero), "=&v"(addend), "=&s"(saved)
               : "s"(count), "s"(ticket)
               : "memory", "scc");
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
}

// What one wave does: chunks from the ticket counter until the batch is
// exhausted, each through the tables at my_smem.  TAGS: a tag table lies behind
// the position table.  WALK: match-less stretches take the walk (needs up to
// 256 vector registers and the accumulation registers to itself, i.e. at most
// four waves per workgroup).
template <int S, bool TAGS, bool WALK>
__device__ __forceinline__ void compress_wave(
    const uint8_t* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    uint8_t* const* __restrict__ out_ptrs,
    size_t* __restrict__ out_bytes,
    const uint32_t ht_size,
    uint8_t* const my_smem,
    const uint32_t wave,
    const uint32_t batch,
    uint32_t* __restrict__ ticket,
    const uint32_t chunks_per_ticket)
{
  constexpr uint32_t LVM = (12 + S - 1) / S; // last valid match, elements
  constexpr int INV = 3 / S;                 // lanes without a full 4-byte word
  constexpr int NVMAX = kWave - INV;
  constexpr int G = kLz4WalkBlock;

  const int lane = lane_id();

  const uint32_t hmask = ht_size - 1;
  Tables<TAGS> T;
  T.pos = reinterpret_cast<uint16_t*>(my_smem);
  T.tag = my_smem + 2 * ht_size;
  T.pos_lds = uniform(lds_addr_of(my_smem));
  T.tag_lds = T.pos_lds + 2 * ht_size;
  T.filter = false;
  const uint32_t sig = sigma_of_lane(lane);
  const int perm_addr4 = (int)(sig * 4u);
  WalkLanes WL;
  WL.validc = wave_ballot(sig < (uint32_t)NVMAX); // valid lanes of a full window, sigma order
  WL.sig = sig < (uint32_t)NVMAX ? sig : 31u;
  WL.never = (NVMAX == 64 && sig == 63u) ? 0x10000u : 0u;
  WL.counts = (sig < (uint32_t)NVMAX && sig != 31u) ? ~0u : 0u;
  // (a bit mask to the compiler, not a condition: it would turn every `& counts`
  // into a scalar lane-mask AND and rebuild a vector value from it for the ballot)
  asm volatile("" : "+v"(WL.counts));
  // where the stores of lanes that must not store go: an LDS address beyond
  // all a workgroup can own -- the hardware drops them (tests/test_hw_probes.py)
  // and the tables can fill the 160 KiB to the last byte.  (As vector
  // registers the compiler cannot re-make from a constant at every use.)
  asm volatile("v_mov_b32_e32 %0, %2\n\tv_add_u32_e32 %1, 2, %0"
               : "=&v"(WL.scratch_pos), "=v"(WL.scratch_tag)
               : "s"(kLdsNowhere));
  const uint32_t rev_lane = 63u - (uint32_t)lane;
  const int rev_addr4 = (int)(rev_lane * 4u);

 for (;;) {
  // A ticket is good for chunks_per_ticket consecutive chunks (more than one
  // for small chunks: atomics on one address run at ~85 M/s chip-wide, which
  // would cap 1 KiB chunks at 87 GB/s).  Without a ticket counter (temp
  // buffer too small to hold one): one chunk per wave, numbered by position
  // in the grid.
  const uint32_t first = ticket ? take_ticket(ticket, chunks_per_ticket)
                                : (uint32_t)blockIdx.x * (uint32_t)(blockDim.x >> 6) + wave;
  if (first >= batch)
    break;
  const uint32_t stop = ticket ? min(first + chunks_per_ticket, batch) : first + 1u;
  for (uint32_t chunk = first; chunk < stop; ++chunk) {
  cgptr __restrict__ in = to_global(in_ptrs[chunk]);
  const uint32_t len = (uint32_t)in_bytes[chunk];
  gptr __restrict__ out = to_global(out_ptrs[chunk]);
  const uint32_t L = (len + S - 1) / S;
  T.filter = TAGS && L <= 65536u;

  // ---- LDS init (reference :815-818 fills the table with NULL_OFFSET); tags
  // of empty slots are never looked at
  {
    u32x4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    u32x4* p = reinterpret_cast<u32x4*>(my_smem);
    const uint32_t nvec = ((ht_size * 2 + 15) & ~15u) >> 4;
    for (uint32_t i = (uint32_t)lane; i < nvec; i += kWave)
      p[i] = ones;
  }

  uint32_t d = 0, c = 0;
  // highest element with 4 readable bytes; windows exist only while
  // d + LVM < L, so the clamped loads are only ever used with L > LVM
  const uint32_t last_word = L > LVM ? L - LVM - 1 : 0;
  // window word of lane t = the 4 bytes at element d+t (reference :848-854;
  // for every lane < nv none of them is masked), always loaded one window
  // ahead.
  uint32_t next = 0;
  if (L > LVM)
    next = load_u32_any(in + (size_t)min((uint32_t)lane, last_word) * S);
  // windows without a match in a row: the walk starts after kLz4WalkAfter
  int cold = 0;

  uint32_t token_start = 0; // first element not yet written out
  while (d < L) {
    if (WALK && cold >= kLz4WalkAfter && (int)(L - d - LVM) >= 2 * G * NVMAX) {
      // ---- the walk (here: two blocks of full windows lie ahead)
      bool match;
      d = (len <= 65536u) ? walk_run<S, G, true>(T, in, d, L, WL, hmask, last_word, match)
                          : walk_run<S, G, false>(T, in, d, L, WL, hmask, last_word, match);
      // the tables are in the state before the window at d, which has a match,
      // or (!match) the walk has run out of blocks of full windows at d
      next = load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
      cold = match ? 0 : 1; // (1: not back into the walk for the few windows left)
    }

    if (d + LVM >= L) {
      // literals to the end of the chunk (reference :832-845)
      c = write_sequence(out, c, in + (size_t)token_start * S,
                         len - token_start * S, 0, 0, lane);
      break;
    }
    // ---- one window at a time: three LDS round trips, then the decision
    Window P;
    uint32_t pr;
    window_begin<S, NVMAX>(P, d, next, L, hmask, lane);
    lds_lookup_with_bpermute(T, P.hpos, rev_addr4, P.hpos | (P.valid ? 0x80000000u : 0u),
                             P.h_old, P.t_old, pr);
    window_candidate<S>(P, T, in, last_word, lane, cold > 0);
    window_markers(P, T, pr, rev_lane);
    const uint32_t nw = (uint32_t)__builtin_amdgcn_ds_bpermute(
        (int)(window_winner(P, lane) * 4u), (int)P.word);
    const Decision D = window_decide<NVMAX>(P, nw, lane);
    if (D.match) {
      window_insert_first<NVMAX>(P, T, D.f, perm_addr4, sig, hmask, lane);
      emit_match<S>(out, c, in, token_start, P.d, P.word, D, L, lane, d);
      next = load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
      token_start = d;
      cold = 0;
    } else {
      // no match in this window (reference :958-962): all nv lanes go in
      window_insert_first<NVMAX>(P, T, P.nv, perm_addr4, sig, hmask, lane);
      d += (uint32_t)P.nv;
      next = cold > 0 ? P.next_word
                      : load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
      ++cold;
    }
  }
  if (lane == 0)
    out_bytes[chunk] = c;
  } // next chunk of this ticket
  if (!ticket)
    break;
 } // next ticket
}

// Workgroup shapes.  Each wave owns one set of tables and takes chunks from a
// global ticket counter until the batch is exhausted; the waves never
// synchronise with each other.
//   "mix"   tables in LDS.  They (ht_size x u16, + ht_size x u8 of tags; 32 /
//           48 KiB for 64 KiB chunks) are the only LDS user and LDS is what
//           limits residency; the CU hands LDS out in 1280-byte granules, so
//           ONE workgroup that owns all 160 KiB holds more tables than several
//           small ones.  Up to four waves, the first n_tagged of them with a
//           tag table (64 KiB chunks: two with, two without = 160 KiB to the
//           byte), all with the walk: the shape for data with match-less
//           stretches, and for batches small enough to be in flight at once.
//   "far"   tables in device memory, 32 waves per CU, no tags, no walk (64
//           vector registers): for data that has a match in nearly every
//           window (compress_wave_far below).
// `mode` (may be null) points at the three counters of the sampling kernel: a
// kernel whose shape is not the one they call for leaves at once.
constexpr uint32_t kModeMix = 1, kModeFar = 2, kModeFarWide = 3;

// {words that repeated, words looked at, words equal to one 1, 2, 4 or 8 bytes before} -> shape
__device__ __forceinline__ uint32_t sampled_mode(const uint32_t* counters)
{
  const uint32_t repeats = uniform(counters[0]), looked = uniform(counters[1]), near = uniform(counters[2]);
  if (looked == 0 || repeats * 4u <= looked)
    return kModeMix;
  // data that compresses: mostly by words repeating a few bytes on (runs) or not
  return near * 2u > looked ? kModeFarWide : kModeFar;
}

template <int S>
__global__ __launch_bounds__(kLz4MaxWavesPerGroup * kWave) void lz4_compress_kernel_mix(
    const uint8_t* const* __restrict__ in_ptrs, const size_t* __restrict__ in_bytes,
    uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ out_bytes,
    const uint32_t ht_size, const uint32_t n_tagged, const uint32_t stride_tagged, const uint32_t stride_plain,
    const uint32_t batch, uint32_t* __restrict__ ticket, const uint32_t chunks_per_ticket,
    const uint32_t* __restrict__ mode)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (mode && sampled_mode(mode) != kModeMix)
    return;
  const uint32_t wave = uniform((uint32_t)(threadIdx.x >> 6));
  if (wave < n_tagged)
    compress_wave<S, true, true>(in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, smem + wave * stride_tagged,
                                 wave, batch, ticket, chunks_per_ticket);
  else
    compress_wave<S, false, true>(in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size,
                                  smem + n_tagged * stride_tagged + (wave - n_tagged) * stride_plain, wave, batch,
                                  ticket, chunks_per_ticket);
}

// ---------------------------------------------------------------------------
// The "far" shape: the position table of a chunk lies in device memory (a
// slice of the caller's temp buffer, where the reference keeps it), so that
// LDS no longer limits how many chunks a CU works on -- sixteen waves instead
// of five.  For data with a match in nearly every window the encoder is a
// chain of dependent round trips per sequence and only more chains in flight
// make it faster; a table round trip that is several times longer is the
// lesser evil.  One window at a time, no tags, no walk.
//
// What LDS did for the tables is done as follows.
//   in-window duplicates: lane ids posted in reversed lane order as in
//     window_markers, but into a small per-wave LDS scratch indexed by the low
//     bits of the slot number.  Lanes of one table slot share a scratch slot,
//     so a lane whose scratch slot names a lane with ITS word has found the
//     lowest lane holding that word; any other lane is settled by the exact
//     fallback of window_decide.
//   insert rule: the same store masks as insert_sigma / the n <= 31 rule, but
//     of the lanes of a mask that share a table slot only the one LDS would
//     have kept stores (slot_tops) -- no same-address stores, so nothing
//     depends on how global memory would arbitrate them.
// A wave's loads from its table see its earlier stores: same wave, same
// address, program order.
// ---------------------------------------------------------------------------
constexpr int kFarWavesPerGroup = 4;
constexpr int kFarFirst = 8; // lanes whose table slots are looked up before the rest
// Elements a window must have ahead of it for the straight-line path: the
// window, the lanes of the match, the bytes looked at for the match length, the chunk's tail.
constexpr uint32_t kFarFastMargin = 400;
constexpr int kFarGroupsPerCu = 8; // 32 waves: 64 vector registers each
constexpr uint32_t kFarScratchSlots = 2048; // u16 each, per wave: 128 KiB per CU

// Of the lanes of `range`, those that are the highest lane of their table slot
// among the lanes of `range`.
__device__ __forceinline__ uint64_t slot_tops(
    uint32_t hpos, uint64_t range, uint16_t* scr, int lane)
{
  uint64_t tops = 0;
  if (__builtin_popcountll(range) <= 6) {
    uint64_t rem = range;
    while (rem) {
      const int u = 63 - __builtin_clzll(rem);
      tops |= 1ull << u;
      rem &= ~wave_ballot(hpos == read_lane(hpos, u));
    }
    return tops;
  }
  // lane ids into the scratch in natural lane order: the highest lane of a
  // scratch slot stays
  const uint32_t ks = hpos & (kFarScratchSlots - 1u);
  const bool in = (range >> lane) & 1ull;
  lds_lane_exchange_fence();
  if (in)
    scr[ks] = (uint16_t)lane;
  lds_lane_exchange_fence();
  const uint32_t top = in ? (uint32_t)scr[ks] : (uint32_t)lane;
  lds_lane_exchange_fence();
  const uint32_t htop = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(top * 4u), (int)hpos);
  // top == lane: highest lane of my scratch slot, hence of my table slot;
  // htop == hpos: a higher lane of my table slot;  else a lane of another
  // table slot hides mine: exact compare
  tops = wave_ballot(top == (uint32_t)lane) & range;
  uint64_t U = range & ~tops & ~wave_ballot(htop == hpos);
  while (__builtin_expect(U != 0, 0)) {
    const int u = __builtin_ctzll(U);
    U &= U - 1;
    const uint64_t m = wave_ballot(hpos == read_lane(hpos, u)) & range;
    if (63 - __builtin_clzll(m) == u)
      tops |= 1ull << u;
  }
  return tops;
}

// table[hpos] = value for the lanes of `mask` (distinct slots)
__device__ __forceinline__ void far_store_masked(
    HC_GLOBAL uint16_t* table, uint32_t hpos, uint32_t value, uint64_t mask, int lane)
{
  if ((mask >> lane) & 1ull)
    table[hpos] = (uint16_t)value;
}

// Table state "the first n lanes of W were inserted" (see insert_sigma for the rule).
template <int NVMAX>
__device__ __forceinline__ void far_insert_first(
    const Window& W, HC_GLOBAL uint16_t* table, uint16_t* scr, int n, int perm_addr4, uint32_t sig,
    uint32_t hmask, int lane)
{
  if (n >= 32) {
    const uint32_t ws = (uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr4, (int)W.word);
    const uint32_t hp = hash_sum(ws) & hmask;
    uint64_t in31;
    const uint64_t store = sigma_store_mask(hp, wave_ballot(sig < (uint32_t)n), n == 64, in31);
    far_store_masked(table, hp, (W.d + sig) & 0xFFFFu, slot_tops(hp, store, scr, lane), lane);
  } else if (n > 0) {
    far_store_masked(table, W.hpos, (W.d + (uint32_t)lane) & 0xFFFFu,
                     slot_tops(W.hpos, lanes_below<NVMAX>(n), scr, lane), lane);
  }
}

// ---------------------------------------------------------------------------
// The common window of data that compresses, as straight a line as it can be
// written (the far kernel is bound by its instruction count, the scalar one
// above all -- 8 waves share a SIMD's issue slots -- and by trips to memory):
// no literals pending, the match among the first kFarFirst lanes, a short
// sequence.  Two forms:
//   lean (WIDE = false)  a table match, no duplicate below it, match < 19 bytes;
//   wide                 a table match or a lane with an equal lower lane, at
//                        most one match length byte (match < 274 bytes).
// Nothing is changed before all of that is known; a window that is anything
// else is left to the caller.  Returns false if the window it stopped at had
// no match of its kind among the first lanes at all (the caller lets the form
// rest until its general code meets one).  only_one: take one window at most.
//
// The window's words as this path sees them: from memory (`next`) on entry,
// from then on the words of the window before, moved down by the lanes the
// sequence took (ds_bpermute: the lanes this path looks at are all there) --
// the load of the new window's words is then off the chain from one sequence
// to the next; the general code waits for it.
// ---------------------------------------------------------------------------
template <int S, bool WIDE>
__device__ __forceinline__ bool far_straight(
    cgptr __restrict__ in, gptr __restrict__ out, HC_GLOBAL uint16_t* const table, uint16_t* const scr,
    const uint32_t hmask, const uint32_t L, const uint32_t last_word, const int lane, uint32_t& d, uint32_t& c,
    uint32_t& token_start, int& cold, uint32_t& next, const int only_one)
{
  uint32_t wnd = next;
  // (wide: the 64 words behind the window's as well -- all 64 lanes of the next
  // window then hold its words whatever the sequence moved by, and the length of
  // a match against a lower lane can be read off the window)
  uint32_t next_hi = 0;
  if (WIDE)
    next_hi = load_u32_any(in + (size_t)min(d + 64u + (uint32_t)lane, last_word) * S);
  bool armed = true;
  uint32_t runs_rest = 0, runs_fails = 0; // (the runs trip below: tried again after twice as many windows when it took nothing)
  while (d + kFarFastMargin <= L) {
    // ---- Wide: runs of values never seen before (run-length data whose values
    // are of the element's size; `step` = 1 lane).  Every sequence is "the
    // literals up to the first lane that equals the lane `step` below, then a
    // match against that lane for as long as that goes on", and one trip to the
    // table settles all of them
    // inside kRunSpan lanes -- provided no lane there has a table candidate, and
    // every lane that shares its scratch slot (hence possibly its table slot, or
    // its word) with a higher lane does so only with the lanes `step`, 2 x
    // `step`, ... above it for which that equality goes on without a gap: then the
    // first lane of a window with an equal lower lane is the first one at least
    // `step` above its start that equals the lane `step` below, and that lane is
    // the lowest one with its word.  The sequences come off the mask of "equal
    // to the lane `step` below" with scalar bit operations and are written at
    // once as in far_straight_several.  (2- and 4-byte elements: with byte elements
    // the usual runs are of wider values, which this does not take, and trying costs.)
    if (WIDE && S > 1) {
      if (runs_rest != 0) {
        --runs_rest;
      } else {
        runs_rest = min((1u << runs_fails) - 1u, 15u);
        runs_fails = min(runs_fails + 1u, 5u);
        constexpr uint32_t kRunSpan = 48;
        const uint32_t word = wnd;
        const uint32_t hpos = hash_sum(word) & hmask;
        const uint32_t pos = d + (uint32_t)lane;
        // (equal to the lane right below: with any other distance a lane that shares a
        // slot with its partners could hide one that does not belong to them)
        constexpr uint32_t step = 1;
        uint64_t same_below;
        {
          const uint32_t lower = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)lane - step) & 63u) * 4u), (int)word);
          same_below = wave_ballot((uint32_t)lane >= step && lower == word);
        }
        // the last lane of the unbroken stretch of "equal to the lane `step` below" that
        // begins at the lane `step` above mine, or my own lane if that lane is not one
        // (127: the stretch reaches the window's edge)
        const uint32_t theirs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)lane - step) & 63u) * 4u), (int)word);
        const uint32_t differ = word ^ theirs; // (lanes below `step`: against a wrapped lane, not used)
        uint32_t stretch_end;
        {
          const uint32_t from = (uint32_t)lane + step;
          const uint64_t breaks_above = from < 64u ? ~same_below >> from : ~0ull;
          uint32_t lo, hi;
          asm("v_ffbl_b32 %0, %1" : "=v"(lo) : "v"((uint32_t)breaks_above));
          asm("v_ffbl_b32 %0, %1" : "=v"(hi) : "v"((uint32_t)(breaks_above >> 32)));
          const uint32_t r = min(lo, hi | 32u); // lanes from `from` on that are equal to the one `step` below
          stretch_end = from >= 64u ? (uint32_t)lane : (r == ~0u ? 127u : (r == 0u ? (uint32_t)lane : from + r - 1u));
        }
        bool trouble = false;
        if ((uint32_t)lane < kRunSpan) {
          const uint32_t h_old = table[hpos];
          const uint32_t ks = hpos & (kFarScratchSlots - 1u);
          lds_lane_exchange_fence();
          scr[ks] = (uint16_t)lane; // (the highest lane of a slot stays)
          lds_lane_exchange_fence();
          const uint32_t top = scr[ks];
          lds_lane_exchange_fence();
          const uint32_t back = (pos - 1u - h_old) & 0xFFFFu;
          // the highest lane with my word that the stretch explains: mine + a multiple of `step`
          const uint32_t reach = stretch_end == 127u ? 127u : (uint32_t)lane + (stretch_end - (uint32_t)lane) / step * step;
          trouble = ((h_old != kNullOffset) & (back < 65535u / S)) || top > reach;
        }
        if (wave_ballot(trouble) == 0) {
          uint32_t start = 0;
          uint64_t match_lanes = 0, start_lanes = 0;
          while (start + step < kRunSpan) {
            const uint64_t above = same_below & (~0ull << (start + step)) & lanes_below<64>(kRunSpan);
            if (above == 0)
              break;
            const uint32_t g = (uint32_t)__builtin_ctzll(above);
            const uint32_t n_same = (uint32_t)__builtin_ctzll(~(same_below >> g) | (1ull << 63));
            if ((g - start) * S >= 15u || g + n_same >= 62u)
              break; // (length bytes for the literals; a stretch up to the window's edge may go on)
            // (the partly equal word behind the stretch: at most 3 bytes, less than an element for S = 4)
            const uint32_t q = g + n_same;
            const uint32_t ml = (n_same * S + ((uint32_t)__builtin_ctz(read_lane(differ, (int)q) | 0x80000000u) >> 3)) / S;
            if (ml * S >= 19u + 255u)
              break;
            match_lanes |= 1ull << g;
            start_lanes |= 1ull << start;
            start = g + ml;
          }
          if (match_lanes != 0) {
            const uint64_t lits = match_lanes - start_lanes; // lanes start..match-1 of every sequence
            const bool is_match = ((match_lanes >> lane) & 1ull) != 0, is_lit = ((lits >> lane) & 1ull) != 0;
            // my match, if I am a match lane: as long as the stretch from me on, and the partly equal word behind it
            const uint32_t my_end = (uint32_t)lane + (uint32_t)__builtin_ctzll(~(same_below >> lane) | (1ull << 63));
            const uint32_t behind = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((my_end & 63u) * 4u), (int)differ);
            const uint32_t match_bytes = (((my_end - (uint32_t)lane) * S + ((uint32_t)__builtin_ctz(behind | 0x80000000u) >> 3)) / S) * S;
            const uint64_t long_lanes = wave_ballot(is_match && match_bytes >= 19u); // one length byte
            const bool is_long = ((long_lanes >> lane) & 1ull) != 0;
            const uint32_t lits_below = __builtin_amdgcn_mbcnt_hi((uint32_t)(lits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lits, 0u));
            const uint32_t seqs_below = __builtin_amdgcn_mbcnt_hi((uint32_t)(match_lanes >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)match_lanes, 0u));
            const uint32_t longs_below = __builtin_amdgcn_mbcnt_hi((uint32_t)(long_lanes >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)long_lanes, 0u));
            // (behind the token and the literals below mine of my sequence)
            const uint32_t at = c + lits_below * S + 3u * seqs_below + longs_below + 1u;
            if (is_lit) {
              if (S == 1)
                out[at] = (uint8_t)word;
              else if (S == 2)
                *reinterpret_cast<HC_GLOBAL uint16_t __attribute__((aligned(1)))*>(out + at) = (uint16_t)word;
              else
                *reinterpret_cast<HC_GLOBAL u32_unaligned*>(out + at) = word;
            }
            if (is_match) {
              const uint64_t upto = start_lanes & ((2ull << lane) - 1ull);
              const uint32_t lit_mine = (uint32_t)lane - (63u - (uint32_t)__builtin_clzll(upto | 1ull));
              out[at - 1u - lit_mine * S] = (uint8_t)(((lit_mine * S) << 4) | (is_long ? 15u : match_bytes - 4u));
              *reinterpret_cast<HC_GLOBAL uint16_t __attribute__((aligned(1)))*>(out + at) = (uint16_t)(step * S);
              if (is_long)
                out[at + 2u] = (uint8_t)(match_bytes - 19u);
            }
            far_store_masked(table, hpos, pos & 0xFFFFu, lits, lane);
            c += (uint32_t)__builtin_popcountll(lits) * S + 3u * (uint32_t)__builtin_popcountll(match_lanes)
                 + (uint32_t)__builtin_popcountll(long_lanes);
            const uint32_t moved = start;
            const int from4 = (int)((((uint32_t)lane + moved) & 63u) * 4u);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(from4, (int)next);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(from4, (int)next_hi);
            wnd = (uint32_t)lane + moved < 64u ? lo : hi;
            d += moved;
            token_start = d;
            cold = 0;
            next = load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
            next_hi = load_u32_any(in + (size_t)min(d + 64u + (uint32_t)lane, last_word) * S);
            runs_rest = 0;
            runs_fails = 0;
            if (only_one)
              break;
            continue;
          }
        }
      }
    }
    const uint32_t word = wnd;
    const uint32_t hpos = hash_sum(word) & hmask;
    uint32_t h_old = kNullOffset;
    if (lane < kFarFirst)
      h_old = table[hpos];
    const uint32_t pos = d + (uint32_t)lane;
    const uint32_t back = (pos - 1u - h_old) & 0xFFFFu; // (see window_candidate)
    const uint32_t cand = pos - 1u - back;
    const bool usable = (h_old != kNullOffset) & (back < 65535u / S);
    // (no lane with a candidate -- values never seen before, as in runs of new
    // values: no trip to memory for them)
    uint32_t cand_word = ~word;
    if (!WIDE || wave_ballot(usable) != 0)
      cand_word = load_u32_any(in + (size_t)(usable ? cand : pos) * S);
    const uint64_t tmask = wave_ballot(usable & (cand_word == word));
    int f;
    uint32_t mloc;
    uint32_t below_by = 0; // wide: the match is the lane this many lanes below (0: a table match)
    if (!WIDE) {
      if (tmask == 0) {
        armed = false;
        break;
      }
      f = __builtin_ctzll(tmask); // < kFarFirst
      // a duplicate among lanes 0..f-1 would come first
      bool duplicate = false;
      for (int u = 0; u + 1 < f; ++u)
        duplicate |= (wave_ballot(word == read_lane(word, u)) & lanes_below<64>(f) & ~lanes_below<64>(u + 1)) != 0;
      if (duplicate)
        break;
      mloc = read_lane(cand, f);
    } else {
      const int k = tmask ? __builtin_ctzll(tmask) : kFarFirst; // first lane with a table match
      // a lane below k with an equal lower lane comes first (reference :868-894).
      // Among the first 8 lanes (one DPP row): my word against the 7 lanes below.
      uint64_t dups = 0;
      if (k >= 2) {
        // row_shr:j -- lane t reads lane t - j; lanes without one keep ~word
#define HC_EQ_BELOW(j) \
  ((uint32_t)__builtin_amdgcn_update_dpp((int)~word, (int)word, 0x110 + (j), 0xF, 0xF, false) == word ? 1u : 0u)
        const uint32_t eq = HC_EQ_BELOW(1) | HC_EQ_BELOW(2) | HC_EQ_BELOW(3) | HC_EQ_BELOW(4) | HC_EQ_BELOW(5)
                            | HC_EQ_BELOW(6) | HC_EQ_BELOW(7);
#undef HC_EQ_BELOW
        dups = wave_ballot(eq != 0) & lanes_below<64>(k);
      }
      if (dups) {
        f = __builtin_ctzll(dups);
        const uint32_t lowest = (uint32_t)__builtin_ctzll(wave_ballot(word == read_lane(word, f))); // lowest lane holding the word
        mloc = d + lowest;
        below_by = (uint32_t)f - lowest;
      } else if (tmask) {
        f = k;
        mloc = read_lane(cand, k);
      } else {
        armed = false;
        break;
      }
    }
    const uint32_t mpos = d + (uint32_t)f;
    // match length: the first 32 bytes (8 lanes, one line each side); the wide
    // form goes on with the general search
    uint32_t ml = 0;
    bool have_length = false;
    if (WIDE && below_by != 0) {
      // against a lower lane of the window: the words are all here -- the first
      // lane from f on whose word differs from the one below_by lanes below, and
      // the equal low bytes of that word
      const uint32_t theirs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)lane - below_by) & 63u) * 4u), (int)word);
      const uint32_t differ = word ^ theirs;
      const uint64_t stops = wave_ballot(differ != 0) & ~lanes_below<64>(f);
      if (stops) {
        const int q = __builtin_ctzll(stops);
        ml = (((uint32_t)(q - f)) * S + ((uint32_t)__builtin_ctz(read_lane(differ, q)) >> 3)) / S;
        have_length = true;
      }
    }
    if (!have_length) {
      uint32_t x = 0;
      if (lane < 8)
        x = load_u32_any(in + (size_t)mloc * S + 4u * (uint32_t)lane)
            ^ load_u32_any(in + (size_t)mpos * S + 4u * (uint32_t)lane);
      const uint32_t diff_at = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
      const uint64_t stop = wave_ballot(diff_at < 4u);
      if (stop) {
        const int sl = __builtin_ctzll(stop);
        ml = (4u * (uint32_t)sl + read_lane(diff_at, sl)) / S;
      } else if (WIDE) {
        ml = match_length<S>(in, mloc, mpos, L - mpos - (5 + S - 1) / S, lane);
      } else {
        break;
      }
    }
    const uint32_t lit_bytes = (uint32_t)f * S, match_bytes = ml * S;
    if (lit_bytes >= 15u || match_bytes >= (WIDE ? 19u + 255u : 19u))
      break; // (more length bytes than this path writes)
    // ---- decided: insert the first f lanes, write the sequence in ONE store
    // (token, literals from the window registers, offset, wide: at most one
    // match length byte), move on
    if (f > 0)
      far_store_masked(table, hpos, pos & 0xFFFFu, slot_tops(hpos, lanes_below<64>(f), scr, lane), lane);
    {
      const uint32_t offset_bytes = (((mpos - mloc) & 0xFFFFu) * S) & 0xFFFFu;
      const uint32_t ext = (WIDE && match_bytes >= 19u) ? 1u : 0u;
      const uint32_t i = (uint32_t)lane, li = i - 1u;
      const uint32_t src = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((li / S) * 4u), (int)word);
      uint32_t bt = (src >> (8u * (li % S))) & 0xFFu;
      if (i == 0)
        bt = (lit_bytes << 4) | (ext ? 15u : match_bytes - 4u);
      else if (i == lit_bytes + 1)
        bt = offset_bytes & 0xFFu;
      else if (i == lit_bytes + 2)
        bt = offset_bytes >> 8;
      else if (WIDE && i == lit_bytes + 3)
        bt = match_bytes - 19u;
      if (i < lit_bytes + 3 + ext)
        out[c + i] = (uint8_t)bt;
      c += lit_bytes + 3 + ext;
    }
    // the next window's first lanes from this window's words (`next`: from
    // memory, long there) if the sequence left them inside it
    const uint32_t moved = (uint32_t)f + ml;
    wnd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)lane + moved) & 63u) * 4u), (int)next);
    if (WIDE) {
      const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)lane + moved) & 63u) * 4u), (int)next_hi);
      wnd = (uint32_t)lane + moved < 64u ? wnd : hi;
    }
    d = mpos + ml;
    token_start = d;
    cold = 0;
    next = load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
    if (WIDE)
      next_hi = load_u32_any(in + (size_t)min(d + 64u + (uint32_t)lane, last_word) * S);
    if (only_one)
      break;
    if (WIDE && __builtin_expect(moved > 64u, 0)) {
      // (a real branch: as a select it would make every trip wait for the load)
      asm volatile("" ::: "memory");
      wnd = next;
    }
  }
  return armed;
}

// ---------------------------------------------------------------------------
// The lean form, several sequences per trip to memory: the
// first kFarSpan lanes look their table slots up and fetch 16 bytes at their
// candidates -- they hold the match length of a match shorter than 16 bytes as
// well -- and the sequences are then taken off one after the other in registers
// (the next one's window starts where the match ended; its first table match is
// the next lane with one) for as long as they stay inside those lanes.  A
// sequence is taken if its literal lanes sit in table slots of their own (two
// lanes with one word, the duplicate of reference :868-894, share one) and no
// lane of its window up to the match sits in a slot that an earlier sequence of
// the trip has written (what it looked up was read before that); the table is
// written per sequence.  A window that is anything else is left to the caller.  Returns false if the window it
// stopped at had no table match among those lanes at all.
// ---------------------------------------------------------------------------
#ifndef HC_FAR_SPAN
#define HC_FAR_SPAN 40 // (measurement builds; bytes, 20 000 chunks: 32: harness 88 / text 38.5 GB/s, 40: 97 / 38.6, 48: 99.5 / 37.2)
#endif
constexpr int kFarSpan = HC_FAR_SPAN;
constexpr int kFarSpanMost = 52; // (the words 12 bytes on of its lanes are still in the window)

__device__ __forceinline__ int first_set_or_minus_one(uint64_t m) // (s_ff1_i32_b64 as it is)
{
  int r;
  asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m));
  return r;
}

template <int S>
__device__ __forceinline__ bool far_straight_several(
    cgptr __restrict__ in, gptr __restrict__ out, HC_GLOBAL uint16_t* const table, uint16_t* const scr,
    const uint32_t hmask, const uint32_t L, const uint32_t last_word, const int lane, uint32_t& d, uint32_t& c,
    uint32_t& token_start, int& cold, uint32_t& next, const uint32_t span)
{
  // lanes (= elements) the window's words can move down by (two registers of words)
  constexpr uint32_t kReach = 64;
  constexpr uint32_t kMostLiterals = 14 / S; // < 15 literal bytes: no length bytes
  uint32_t wnd = next;
  // (the 64 words behind the window's as well: the next window's words are then
  // there whatever the trip's sequences moved by)
  uint32_t next_hi = load_u32_any(in + (size_t)min(d + 64u + (uint32_t)lane, last_word) * S);
  bool armed = true;
  while (d + kFarFastMargin <= L) {
    const uint32_t word = wnd;
    const uint32_t hpos = hash_sum(word) & hmask;
    const uint32_t pos = d + (uint32_t)lane;
    // the words 4, 8 and 12 bytes on
    const uint32_t d1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane + 4u / S) & 63u) * 4, (int)word);
    const uint32_t d2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane + 8u / S) & 63u) * 4, (int)word);
    const uint32_t d3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane + 12u / S) & 63u) * 4, (int)word);
    // what a lane says about its table match, in one word: bit 30 set: there is
    // one, bit 31: of 16 bytes or more, 26-29 matching bytes - 4, 0-15 the offset in bytes
    uint32_t about = 0;
    bool shares = false; // a higher lane (of the span) sits in my scratch slot, hence possibly in my table slot
    if ((uint32_t)lane < span) {
      {
        const uint32_t ks = hpos & (kFarScratchSlots - 1u);
        lds_lane_exchange_fence();
        scr[ks] = (uint16_t)lane; // (the highest lane of a slot stays)
        lds_lane_exchange_fence();
        shares = (uint32_t)scr[ks] != (uint32_t)lane;
        lds_lane_exchange_fence();
      }
      const uint32_t h_old = table[hpos];
      const uint32_t back = (pos - 1u - h_old) & 0xFFFFu; // (see window_candidate)
      const uint32_t cand = pos - 1u - back;
      const bool usable = (h_old != kNullOffset) & (back < 65535u / S);
      const u32x4 cw = load_u128_any(in + (size_t)(usable ? cand : 0u) * S);
      // first differing byte among bytes 4..15 of the match (v_ffbl_b32: -1 for 0)
      uint32_t f1, f2, f3;
      asm("v_ffbl_b32 %0, %1" : "=v"(f1) : "v"(cw.y ^ d1));
      asm("v_ffbl_b32 %0, %1" : "=v"(f2) : "v"(cw.z ^ d2));
      asm("v_ffbl_b32 %0, %1" : "=v"(f3) : "v"(cw.w ^ d3));
      const uint32_t more = min(min(min(f1, f2 | 32u), f3 | 64u) >> 3, 12u); // 0..11, 12: all 12 bytes equal
      const uint32_t code = (usable && cw.x == word) ? (more < 12u ? 1u : 3u) : 0u;
      about = (((back + 1u) * S) & 0xFFFFu) | (more << 26) | (code << 30);
    }
    const uint64_t matches = wave_ballot(about >= (1u << 30));
    uint32_t start = 0; // lane at which the next sequence's window starts
    uint64_t stale = 0; // lanes whose table slot the trip's sequences have written: what they looked up is no longer there
    // the next sequence: its match lane, the lanes in the slots of its literal
    // lanes, and what the match lane says -- 0 if it is not one this path takes
    int f;
    uint64_t touched;
    auto pick = [&]() -> uint32_t {
      f = first_set_or_minus_one(matches & (~0ull << min(start, 63u))); // (start <= span - 1 + 15; bit 63 of matches is never set)
      const uint32_t lit = (uint32_t)f - start; // (no match: huge)
      if (lit > kMostLiterals)
        return 0u;
      const uint32_t a = read_lane(about, f);
      const uint64_t range = (1ull << f) - (1ull << start); // the literal lanes
      uint64_t clash = stale & (range | (1ull << f));
      touched = 0;
      // (a literal lane in the slot of the match lane as well: the match lane may
      // then be the duplicate of a lower lane, which the general code settles)
      for (uint32_t u = start; u < (uint32_t)f; ++u) {
        const uint64_t same_slot = wave_ballot(hpos == read_lane(hpos, (int)u));
        clash |= same_slot & (range | (1ull << f)) & (~1ull << u);
        touched |= same_slot;
      }
      return clash == 0 ? a : 0u;
    };
    // What follows a match lane's sequence, for all lanes at once: bits 0-6
    // where the next sequence's window starts, 7-13 its match lane (the first
    // table match at or above that; 127: none), bit 14: that next sequence is
    // one the loop can take without looking -- a short match, few enough
    // literals, and none of its lanes (start .. match) has a higher lane in its
    // scratch slot or sits in a slot a sequence of the trip has written: its
    // literal lanes then have slots of their own, and what it looked up is still
    // there.  While that holds a sequence costs the loop one v_readlane.
    const uint64_t sharing = wave_ballot(shares);
    const uint32_t after = (uint32_t)lane + (4u + ((about >> 26) & 15u)) / S;
    auto first_at_or_above = [&](uint64_t mask) -> uint32_t { // (per lane, from `after`; 127: none)
      const uint64_t m = mask >> (after & 63u);
      uint32_t lo, hi;
      asm("v_ffbl_b32 %0, %1" : "=v"(lo) : "v"((uint32_t)m));
      asm("v_ffbl_b32 %0, %1" : "=v"(hi) : "v"((uint32_t)(m >> 32)));
      const uint32_t r = min(lo, hi | 32u);
      return (after < 64u && r != ~0u) ? after + r : 127u;
    };
    const uint32_t next_match = first_at_or_above(matches);
    bool easy;
    {
      const uint32_t next_about = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((next_match & 63u) * 4u), (int)about);
      // (a sequence without literals inserts nothing and has only its match lane to go stale)
      easy = next_match < 64u && (next_about >> 30) == 1u && next_match - after <= kMostLiterals
             && (next_match == after || first_at_or_above(sharing) > next_match);
    }
    uint32_t follows = after | (next_match << 7) | (easy ? 1u << 14 : 0u);
    // the short sequences: which lanes they start and match at is all the loop notes
    uint64_t match_lanes = 0, start_lanes = 0;
    uint32_t a = pick();
    while ((a >> 30) == 1u) {
      asm("s_bitset1_b64 %0, %1" : "+s"(match_lanes) : "s"(f));
      asm("s_bitset1_b64 %0, %1" : "+s"(start_lanes) : "s"(start));
      if (touched != 0) {
        // lanes in the slots just written are no longer easy to pass
        stale |= touched;
        follows = after | (next_match << 7) | ((easy && first_at_or_above(stale) > next_match) ? 1u << 14 : 0u);
      }
      uint32_t then = read_lane(follows, f);
      start = then & 127u; // (= f + match length)
      while ((then & (1u << 14)) != 0) {
        f = (int)((then >> 7) & 127u);
        asm("s_bitset1_b64 %0, %1" : "+s"(match_lanes) : "s"(f));
        asm("s_bitset1_b64 %0, %1" : "+s"(start_lanes) : "s"(start));
        then = read_lane(follows, f);
        start = then & 127u;
      }
      a = pick();
    }
    // Their bytes, all at once: token, literals, offset per sequence, in lane
    // order.  Where a lane's bytes go is a count of the lanes below it (literal
    // lanes: S bytes each; sequences: 3 bytes each): every literal lane writes
    // its own element, the match lane its sequence's token in front of the
    // literals and the offset behind them.  The literal lanes go into the table
    // (slots of their own, across the trip).
    if (match_lanes != 0) {
      const uint64_t lits = (match_lanes - start_lanes) & ~match_lanes; // lanes start..match-1 of every sequence
      const bool is_match = ((match_lanes >> lane) & 1ull) != 0, is_lit = ((lits >> lane) & 1ull) != 0;
      const uint32_t lits_below = __builtin_amdgcn_mbcnt_hi((uint32_t)(lits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lits, 0u));
      const uint32_t seqs_below = __builtin_amdgcn_mbcnt_hi((uint32_t)(match_lanes >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)match_lanes, 0u));
      // (behind the token and the literals below mine of my sequence)
      const uint32_t at = c + lits_below * S + 3u * seqs_below + 1u;
      if (is_lit) {
        if (S == 1)
          out[at] = (uint8_t)word;
        else if (S == 2)
          *reinterpret_cast<HC_GLOBAL uint16_t __attribute__((aligned(1)))*>(out + at) = (uint16_t)word;
        else
          *reinterpret_cast<HC_GLOBAL u32_unaligned*>(out + at) = word;
      }
      if (is_match) {
        // my sequence starts at the highest start lane at or below me
        const uint64_t upto = start_lanes & ((2ull << lane) - 1ull);
        const uint32_t lit_mine = (uint32_t)lane - (63u - (uint32_t)__builtin_clzll(upto | 1ull));
        const uint32_t match_bytes = ((4u + ((about >> 26) & 15u)) / S) * S;
        out[at - 1u - lit_mine * S] = (uint8_t)(((lit_mine * S) << 4) | (match_bytes - 4u));
        *reinterpret_cast<HC_GLOBAL uint16_t __attribute__((aligned(1)))*>(out + at) = (uint16_t)about;
      }
      far_store_masked(table, hpos, pos & 0xFFFFu, lits, lane);
      c += (uint32_t)__builtin_popcountll(lits) * S + 3u * (uint32_t)__builtin_popcountll(match_lanes);
    }
    if ((a >> 30) == 3u) {
      // a match of 16 bytes or more ends the trip: its length from memory, the
      // sequence by the general writer
      const uint32_t lit = (uint32_t)f - start;
      const uint32_t mpos = d + (uint32_t)f, offset_bytes = a & 0xFFFFu;
      const uint32_t ml = match_length<S>(in, mpos - offset_bytes / S, mpos, L - mpos - (5 + S - 1) / S, lane);
      c = write_sequence(out, c, in + (size_t)(d + start) * S, lit * S, ml * S, offset_bytes, lane);
      far_store_masked(table, hpos, pos & 0xFFFFu, (1ull << f) - (1ull << start), lane);
      start = (uint32_t)f + ml;
    }
    if (start == 0) {
      armed = matches != 0;
      break;
    }
    const uint32_t moved = start;
    {
      const int from4 = (int)((((uint32_t)lane + moved) & 63u) * 4u);
      const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(from4, (int)next);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(from4, (int)next_hi);
      wnd = (uint32_t)lane + moved < 64u ? lo : hi;
    }
    d += moved;
    token_start = d;
    cold = 0;
    next = load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
    next_hi = load_u32_any(in + (size_t)min(d + 64u + (uint32_t)lane, last_word) * S);
    if (__builtin_expect(moved > kReach, 0)) {
      // (a real branch: as a select it would make every trip wait for the load)
      asm volatile("" ::: "memory");
      wnd = next;
    }
  }
  return armed;
}

template <int S, bool WIDE>
__device__ __forceinline__ void compress_wave_far(
    const uint8_t* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    uint8_t* const* __restrict__ out_ptrs,
    size_t* __restrict__ out_bytes,
    const uint32_t ht_size,
    HC_GLOBAL uint16_t* const table,
    uint16_t* const scr,
    const uint32_t batch,
    uint32_t* __restrict__ ticket,
    const uint32_t chunks_per_ticket,
    const uint32_t span)
{
  constexpr uint32_t LVM = (12 + S - 1) / S;
  constexpr int NVMAX = kWave - 3 / S;
  const int lane = lane_id();
  const uint32_t hmask = ht_size - 1;
  const uint32_t sig = sigma_of_lane(lane);
  const int perm_addr4 = (int)(sig * 4u);
  const uint32_t rev_lane = 63u - (uint32_t)lane;
  const int rev_addr4 = (int)(rev_lane * 4u);
  Tables<false> no_tags; // (window_candidate asks it whether tags filter: never)
  no_tags.filter = false;

  for (;;) {
    const uint32_t first = take_ticket(ticket, chunks_per_ticket);
    if (first >= batch)
      break;
    const uint32_t stop = min(first + chunks_per_ticket, batch);
    for (uint32_t chunk = first; chunk < stop; ++chunk) {
      cgptr __restrict__ in = to_global(in_ptrs[chunk]);
      const uint32_t len = (uint32_t)in_bytes[chunk];
      gptr __restrict__ out = to_global(out_ptrs[chunk]);
      const uint32_t L = (len + S - 1) / S;
      {
        u32x4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        HC_GLOBAL u32x4* p = reinterpret_cast<HC_GLOBAL u32x4*>(table);
        const uint32_t nvec = ((ht_size * 2 + 15) & ~15u) >> 4;
        // (the lane's first address is made here, per chunk: as a loop invariant of
        // the whole kernel it costs two registers that the lean forms do not have)
        uint32_t i0 = (uint32_t)lane;
        asm volatile("" : "+v"(i0));
        for (uint32_t i = i0; i < nvec; i += kWave)
          p[i] = ones;
      }
      uint32_t d = 0, c = 0;
      const uint32_t last_word = L > LVM ? L - LVM - 1 : 0;
      uint32_t next = 0;
      if (L > LVM)
        next = load_u32_any(in + (size_t)min((uint32_t)lane, last_word) * S);
      int cold = 0;
      uint32_t token_start = 0;
      // the straight-line path (far_straight) is tried while it keeps finding its kind of window
      bool straight = true;
      while (d < L) {
        // ---- the common windows of data that compresses (far_straight, the
        // form of this kernel)
        if (straight && token_start == d) {
          if (!WIDE)
            straight = far_straight_several<S>(in, out, table, scr, hmask, L, last_word, lane, d, c, token_start, cold, next, span);
          else
            straight = far_straight<S, WIDE>(in, out, table, scr, hmask, L, last_word, lane, d, c, token_start, cold, next, 0);
        }
        if (d + LVM >= L) {
          c = write_sequence(out, c, in + (size_t)token_start * S, len - token_start * S, 0, 0, lane);
          break;
        }
        Window P;
        window_begin<S, NVMAX>(P, d, next, L, hmask, lane);
        // Table slots of the first kFarFirst lanes only: that is where the match
        // of a window of compressible data is, and a slot costs a memory
        // transaction.
        P.h_old = kNullOffset;
        if (lane < kFarFirst)
          P.h_old = table[P.hpos];
        P.t_old = 0;
        window_candidate<S>(P, no_tags, in, last_word, lane, cold > 0);
        uint64_t tmask = window_table_matches<NVMAX>(P, P.nv);
        int f = 0;
        uint32_t mlane = 0;
        // A table match of lane 0 is the decision: no lane is earlier.  Else
        // the duplicates inside the window are looked for.
        if (!(tmask & 1ull)) {
          const uint32_t ks = P.hpos & (kFarScratchSlots - 1u);
          const uint32_t pr = (uint32_t)__builtin_amdgcn_ds_bpermute(
              rev_addr4, (int)(ks | (P.valid ? 0x80000000u : 0u)));
          lds_lane_exchange_fence();
          if (pr & 0x80000000u)
            scr[pr & 0x7FFFFFFFu] = (uint16_t)rev_lane;
          lds_lane_exchange_fence();
          P.w_raw = scr[ks];
          lds_lane_exchange_fence();
          const uint32_t nw = (uint32_t)__builtin_amdgcn_ds_bpermute(
              (int)(window_winner(P, lane) * 4u), (int)P.word);
          window_first_duplicate<NVMAX>(P, nw, lane, f, mlane);
          tmask &= lanes_below<NVMAX>(f);
          if (tmask == 0 && f > kFarFirst) {
            // no match among the first lanes: the slots of the lanes up to f
            P.h_old = kNullOffset;
            if (lane >= kFarFirst && lane < f)
              P.h_old = table[P.hpos];
            window_candidate<S>(P, no_tags, in, last_word, lane, false);
            tmask = window_table_matches<NVMAX>(P, f);
          }
        }
        const Decision D = window_settle(P, f, mlane, tmask);
        straight |= (WIDE ? D.match : tmask != 0) && D.f < kFarFirst;
        if (D.match) {
          far_insert_first<NVMAX>(P, table, scr, D.f, perm_addr4, sig, hmask, lane);
          emit_match<S>(out, c, in, token_start, P.d, P.word, D, L, lane, d);
          next = load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
          token_start = d;
          cold = 0;
        } else {
          far_insert_first<NVMAX>(P, table, scr, P.nv, perm_addr4, sig, hmask, lane);
          d += (uint32_t)P.nv;
          next = cold > 0 ? P.next_word
                          : load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
          ++cold;
        }
      }
      if (lane == 0)
        out_bytes[chunk] = c;
    }
  }
}

// tables = far_waves x max(ht_size, 8) x u16 in device memory, 16-byte aligned
// WIDE: the wide form of the straight-line path (far_straight) instead of the
// lean one -- one kernel with both forms has either run slower (64 registers)
template <int S, bool WIDE>
__global__ __launch_bounds__(kFarWavesPerGroup * kWave, kFarGroupsPerCu) void lz4_compress_kernel_far(
    const uint8_t* const* __restrict__ in_ptrs, const size_t* __restrict__ in_bytes,
    uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ out_bytes,
    const uint32_t ht_size, uint16_t* __restrict__ tables,
    const uint32_t batch, uint32_t* __restrict__ ticket, const uint32_t chunks_per_ticket,
    const uint32_t* __restrict__ mode)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; // kFarScratchSlots x u16 per wave
  if (mode && sampled_mode(mode) != (WIDE ? kModeFarWide : kModeFar))
    return;
  // lanes a trip of the lean form looks up: more when nearly every sampled word
  // repeated (short sequences, issue slots the limit: harness 126 -> 146 GB/s),
  // fewer otherwise (text is bound by the lines a trip pulls in: 38.5 vs 37.3)
  uint32_t span = kFarSpan;
  if (mode && uniform(mode[0]) * 8u > uniform(mode[1]) * 7u)
    span = kFarSpanMost;
  const uint32_t wave = uniform((uint32_t)(threadIdx.x >> 6));
  const size_t gw = (size_t)blockIdx.x * kFarWavesPerGroup + wave;
  compress_wave_far<S, WIDE>(in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size,
                       (HC_GLOBAL uint16_t*)(tables + gw * max(ht_size, 8u)), // (16 bytes at least: filled 16 at a time)
                       reinterpret_cast<uint16_t*>(smem) + wave * kFarScratchSlots, batch, ticket,
                       chunks_per_ticket, span);
}

// Which shape suits the data: kSampleChunks chunks spread over the batch (one
// wave each) have kSampleBytes from their middle looked at, and the 4-byte
// words there (one per byte position) counted that hash to a slot an earlier
// word of the same sample has hashed to: about n / (2 x 16384) of them for
// data without repeats, most of them for data LZ4 compresses.  Only speed
// depends on the answer, never the compressed bytes.
constexpr int kSampleChunks = 64;
constexpr uint32_t kSampleBytes = 2048;

__global__ __launch_bounds__(kWave) void lz4_sample_kernel(
    const uint8_t* const* __restrict__ in_ptrs, const size_t* __restrict__ in_bytes, const uint32_t batch,
    uint32_t* __restrict__ counters)
{
  __shared__ uint32_t seen[16384 / 32];
  const int lane = lane_id();
  const uint32_t chunk = (uint32_t)(((uint64_t)batch * blockIdx.x) / gridDim.x);
  cgptr in = to_global(uniform_ptr(in_ptrs[chunk]));
  const uint32_t len = uniform((uint32_t)in_bytes[chunk]);
  if (len < 64)
    return;
  const uint32_t n = min(kSampleBytes, len - 4u);
  const uint32_t from = (len - 4u - n) / 2;
  for (int i = lane; i < 16384 / 32; i += kWave)
    seen[i] = 0;
  __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): one wave, LDS operations in order
  uint32_t repeats = 0, near = 0;
  for (uint32_t p = (uint32_t)lane; p < n; p += kWave) {
    const uint32_t w = load_u32_any(in + from + p);
    const uint32_t h = hash_sum(w) & 16383u;
    const uint32_t old = atomicOr(&seen[h >> 5], 1u << (h & 31u));
    repeats += (old >> (h & 31u)) & 1u;
    // the word 1, 2, 4 or 8 bytes before (runs of elements of those sizes): from >= 8 here
    if (from >= 8u)
      near += (w == load_u32_any(in + from + p - 1u)) | (w == load_u32_any(in + from + p - 2u))
              | (w == load_u32_any(in + from + p - 4u)) | (w == load_u32_any(in + from + p - 8u));
  }
  for (int o = 32; o > 0; o >>= 1) {
    repeats += __shfl_xor(repeats, o);
    near += __shfl_xor(near, o);
  }
  if (lane == 0) {
    atomicAdd(&counters[0], repeats);
    atomicAdd(&counters[1], n);
    atomicAdd(&counters[2], near);
  }
}

// --------------------------------------------------------------------------
// Decoder.  One chunk per wavefront, kDecompWavesPerBlock chunks per
// workgroup.  All lanes parse the (wave-uniform) token stream; literal runs
// are 16-byte/lane copies, matches are copied with the reference's
// `src[i % offset]` rule (coopCopyOverlap :530-555).
//
// Deliberate tightening versus the reference (DESIGN.md "LZ4 decoder"):
// reads of the compressed stream are bounded by comp_len and offset == 0 is
// rejected; both are undefined behaviour in the reference.
// --------------------------------------------------------------------------
constexpr int kDecompWavesPerBlock = 4;

// Linear small-integer code: bytes are added up to and including the first one
// that is not 0xFF (reference readLSIC).  64 bytes per step, one per lane: a
// 64 KiB literal run has 257 of them, and one byte per step is one memory
// round trip per byte.  False: the stream ends inside the code.
__device__ __forceinline__ bool read_lsic(
    cgptr comp, uint32_t& c, uint32_t end, uint32_t& num, int lane)
{
  for (;;) {
    if (c >= end)
      return false;
    const uint32_t at = c + (uint32_t)lane;
    const uint32_t b = at < end ? (uint32_t)comp[at] : 0u; // past the end: acts as a terminator
    const uint64_t stop = wave_ballot(b != 0xFFu);
    if (stop == 0) { // 64 x 0xFF, all inside the stream
      num += 255u * (uint32_t)kWave;
      c += (uint32_t)kWave;
      continue;
    }
    const int k = __builtin_ctzll(stop);
    if (c + (uint32_t)k >= end)
      return false; // only 0xFF up to the end of the stream
    num += 255u * (uint32_t)k + read_lane(b, k);
    c += (uint32_t)k + 1u;
    return true;
  }
}

// Bytes the decoder's fast path may touch from the token on: token, up to 14
// literals, 2 offset bytes, one match length byte.
constexpr uint32_t kFastSeqBytes = 18;
// The several-sequences step looks at tokens up to 63 bytes on and at what they
// need behind them.
constexpr uint32_t kBatchReach = 64 + kFastSeqBytes;
constexpr uint32_t kBatchRest = 16;

template <bool WRITE_OUT>
__global__ __launch_bounds__(kWave * kDecompWavesPerBlock) void lz4_decompress_kernel(
    const uint8_t* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* __restrict__ out_caps,
    const size_t batch,
    uint8_t* const* __restrict__ out_ptrs,
    size_t* __restrict__ actual_bytes,
    hipcompStatus_t* __restrict__ statuses)
{
  __shared__ uint8_t rank_to_lane[kDecompWavesPerBlock][kWave]; // (the several-sequences step)
  const int lane = lane_id();
  // everything that steers the parse is wave-uniform: say so (see uniform())
  const size_t chunk
      = (size_t)blockIdx.x * kDecompWavesPerBlock + uniform((uint32_t)(threadIdx.x >> 6));
  if (chunk >= batch)
    return;
  cgptr comp = to_global(uniform_ptr(comp_ptrs[chunk]));
  const uint32_t end = uniform((uint32_t)comp_bytes[chunk]);
  const uint32_t cap = WRITE_OUT ? uniform((uint32_t)out_caps[chunk]) : 0xFFFFFFFFu;
  gptr out = WRITE_OUT ? to_global(uniform_ptr(out_ptrs[chunk])) : nullptr;

  // Issue slots, not latency, bound this kernel on data that compresses: a CU
  // runs 32 of these waves and each SIMD issues one scalar and one vector
  // instruction per four cycles, so a sequence costs max(scalar, vector
  // instructions) x 32 cycles on its CU.  Left alone the compiler computes
  // everything wave-uniform -- the whole parse -- on the scalar unit (58 scalar
  // against 21 vector instructions per sequence, profiles/r02_lz4_pmc_per_
  // sequence_decompress.txt).  Hence: the output position d lives in a vector
  // register (vd, same value in all lanes, see in_vector_register) and so does
  // everything computed from it, and the conditions of the fast paths are
  // folded into one sign bit each instead of a mask per compare.
  uint32_t c = 0, vd = 0;
  // (positions stay below 2^31, so that differences can be tested by sign)
  const uint32_t capc = min(cap, 0x7FFFFFFFu);
  bool corrupt = false;
  // Short sequences are parsed from a register window of the stream
  // (StreamWindow), i.e. without a memory round trip in the chain that leads
  // from one token to the next.
  StreamWindow sw;
  // sequences to go before the several-sequences step is tried again: twice as
  // many after every try that took nothing (data of long matches never has two
  // short sequences in a row), none after one that did
  uint32_t batch_rest = 0, batch_fails = 0;
  for (;;) {
    uint32_t tok = 0;
    // ---- fast paths: token, up to 14 literals, offset and at most one match
    // length byte inside the stream; literals + match at most 64 bytes (one
    // byte per lane).  The loop is left at the first sequence that is anything
    // else: it takes the general path below.
    while (c + kFastSeqBytes <= end) {
      sw.ensure(comp, c, end, kBatchReach, lane);
      const uint32_t idx = c - sw.base;
      // ---- several sequences as one step: lane i looks at the stream byte i
      // bytes on as if a token stood there; the sequences that do start are
      // followed from the first one (one v_readlane each) for as long as they
      // are short ones (lengths in the token, all of it inside the stream as
      // the fast paths below ask) and their output fits 64 bytes and the
      // buffer; then every output byte finds its sequence (the highest output
      // start at or below it; its stream lane through a small table in LDS) and
      // its source: the stream window for literals, out[] for matches -- whose
      // source has to lie in front of the step's output (which also says
      // offset != 0 and that it does not overlap), or the step ends in front
      // of that sequence.
      if (batch_rest != 0) {
        --batch_rest;
      } else {
        batch_rest = min((1u << batch_fails) - 1u, kBatchRest); // (undone below if the step takes something)
        batch_fails = min(batch_fails + 1u, 5u);
        const uint32_t i = (uint32_t)lane;
        auto window_bytes = [&](uint32_t at) -> uint32_t { // 4 bytes at byte index `at` (per lane) of the window
          const uint32_t q4 = at & ~3u;
          const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)q4, (int)sw.words);
          const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(q4 + 4u), (int)sw.words);
          return __builtin_amdgcn_alignbyte(hi, lo, at & 3u);
        };
        const uint32_t w_here = window_bytes(idx + i);
        const uint32_t lit_here = (w_here >> 4) & 15u, mlc_here = w_here & 15u;
        const uint32_t off_here = window_bytes(idx + i + 1u + lit_here) & 0xFFFFu;
        // stream bytes | output bytes << 8 | a short one << 16
        const uint32_t says = (3u + lit_here) | ((lit_here + mlc_here + 4u) << 8)
                              | ((lit_here < 15u && mlc_here < 15u) ? 1u << 16 : 0u);
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)vd);
        const uint32_t room = min(capc - min(d0, capc), (uint32_t)kWave);
        const uint32_t last_start = end - c - kFastSeqBytes; // (tokens up to here are the fast paths')
        uint32_t at = 0, total = 0, count = 0;
        uint64_t stream_starts = 0, out_starts = 0;
        uint32_t p = read_lane(says, 0);
        while ((p >> 16) != 0 && total + ((p >> 8) & 0xFFu) <= room && at <= last_start) {
          asm("s_bitset1_b64 %0, %1" : "+s"(stream_starts) : "s"(at));
          asm("s_bitset1_b64 %0, %1" : "+s"(out_starts) : "s"(total));
          total += (p >> 8) & 0xFFu;
          at += p & 0xFFu;
          ++count;
          p = at < (uint32_t)kWave ? read_lane(says, (int)(at & 63u)) : 0u;
        }
        if (count >= 2) {
          const int wave_in_block = (int)(threadIdx.x >> 6);
          if ((stream_starts >> i) & 1ull)
            rank_to_lane[wave_in_block][__builtin_amdgcn_mbcnt_hi((uint32_t)(stream_starts >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)stream_starts, 0u))]
                = (uint8_t)i;
          lds_lane_exchange_fence();
          const uint64_t upto = out_starts & ((2ull << i) - 1ull);
          const uint32_t o_mine = 63u - (uint32_t)__builtin_clzll(upto | 1ull); // where my sequence's output starts
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(out_starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)out_starts, 0u))
                                + (uint32_t)((out_starts >> i) & 1ull) - 1u;
          const uint32_t t_mine = rank_to_lane[wave_in_block][rank & 63u]; // its stream lane
          lds_lane_exchange_fence();
          const uint32_t w_mine = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(t_mine * 4u), (int)w_here);
          const uint32_t off_mine = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(t_mine * 4u), (int)off_here);
          const uint32_t lit_mine = (w_mine >> 4) & 15u;
          // the match: 0 < offset <= what exists in front of it
          const uint32_t bad_mine = (off_mine - 1u) | (d0 + o_mine + lit_mine - off_mine);
          const uint64_t bad_lanes = wave_ballot(i < total && (int32_t)bad_mine < 0);
          if (bad_lanes != 0) { // the step ends in front of the first such sequence
            const int b = __builtin_ctzll(bad_lanes);
            total = read_lane(o_mine, b);
            at = read_lane(t_mine, b);
          }
          if (total != 0) {
            if (WRITE_OUT) {
              const uint32_t k = i - o_mine;
              const uint32_t widx = idx + t_mine + 1u + k;
              const uint32_t wword = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(widx & ~3u), (int)sw.words);
              // A byte comes from the stream window (literal), from out[] in front of
              // the step (a match that reaches back that far), or from a lower lane of
              // this very step -- which may have its byte from a lower lane again
              // (matches of matches, matches that overlap themselves).  Every lane
              // keeps the lane its byte comes from; six rounds of "take the source's
              // source" (chains halve each round) bring all of them to a lane of
              // the first two kinds.
              const bool literal = k < lit_mine;
              const int32_t from_rel = (int32_t)(i - off_mine); // (match lanes) < 0: in front of the step
              const bool outside = i >= total || literal || from_rel < 0;
              uint32_t val = wword >> ((widx & 3u) * 8u);
              if (i < total && !literal && from_rel < 0)
                val = static_cast<cgptr>(out)[d0 + (uint32_t)from_rel];
              uint32_t from = outside ? (i | 0x80u) : (uint32_t)from_rel; // bit 7: a lane that has its byte
              while (wave_ballot((from & 0x80u) == 0u) != 0) {
                const uint32_t theirs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((from & 63u) * 4u), (int)from);
                from = (from & 0x80u) ? from : theirs;
              }
              val = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((from & 63u) * 4u), (int)val);
              if (i < total)
                out[d0 + i] = (uint8_t)val;
            }
            c += at;
            vd += total;
            batch_rest = 0;
            batch_fails = 0;
            continue;
          }
        }
      }
      const uint32_t tw = read_lane(sw.words, (int)(idx >> 2)) >> ((idx & 3u) * 8u); // token = low byte
      tok = tw & 0xFFu;
      const uint32_t litf = (tw >> 4) & 15u;
      if (litf == 15u)
        break;
      { // literal length in the token itself
        // offset (2 bytes) and the byte behind it: a length byte if the
        // token's match field is 15
        const uint32_t vt2 = in_vector_register(sw.bytes_at(idx + 1u + litf));
        const uint32_t vmlc = in_vector_register(tw) & 15u;
        const uint32_t voff = vt2 & 0xFFFFu;
        const uint32_t vml1 = vmlc + 4u;
        const uint32_t vdl = vd + litf;
        // One byte per lane for literals AND match: lane i < lit carries
        // literal i, lane lit + j match byte j.  `a` is the source as an index
        // relative to d: >= 0 means a literal of this very sequence, i.e. a
        // byte of the stream window (for match bytes too: no round trip
        // through the output); < 0 means earlier output.
        const uint32_t i = (uint32_t)lane;
        // (1) the common one: match length in the token (mlc < 15), output
        // fits (d + lit + ml <= cap), offset inside what exists (0 < off <=
        // d + lit), source and destination do not overlap (off >= ml)
        // (off >= ml >= 4 says off != 0)
        const uint32_t bad1 = (14u - vmlc) | (capc - (vdl + vml1)) | (vdl - voff) | (voff - vml1);
        if (wave_ballot((int32_t)bad1 < 0) == 0) {
          if (WRITE_OUT) {
            const int32_t a = (int32_t)(i - (i < litf ? 0u : voff));
            const uint32_t sidx = idx + 1u + (uint32_t)max(a, 0);
            const uint32_t sword = (uint32_t)__builtin_amdgcn_ds_bpermute(
                (int)((sidx >> 2) * 4u), (int)sw.words);
            const uint32_t byte = sword >> ((sidx & 3u) * 8u);
            // Earlier output: loaded whether or not the lane needs it (cheaper
            // than a branch); d + a lies inside the buffer for every lane that
            // stores (0 <= d + lit - off, d + a < d + lit + ml <= cap).  Earlier
            // stores of this wave to out[] are ordered before this load (one
            // wave, in-order vector memory, one L1).
            if (i < litf + vml1) {
              const uint32_t gb = static_cast<cgptr>(out)[vd + (uint32_t)a];
              out[vd + i] = (uint8_t)(a >= 0 ? byte : gb);
            }
          }
          c += 3u + litf;
          vd = vdl + vml1;
          continue;
        }
        // (2) one length byte (not 255: a second one would follow) and / or
        // a match that overlaps itself: its source repeats with period `off`.
        // n2 <= 64, d + n2 <= cap, 0 < off <= d + lit
        const uint32_t vext = (vt2 >> 16) & 0xFFu;
        const uint32_t vml2 = vml1 + (vmlc == 15u ? vext : 0u);
        const uint32_t vn2 = vml2 + litf;
        const uint32_t bad2 = ((vmlc == 15u) & (vext == 255u) ? ~0u : 0u) | ((uint32_t)kWave - vn2)
                              | (capc - (vd + vn2)) | (voff - 1u) | (vdl - voff);
        if (wave_ballot((int32_t)bad2 < 0) == 0) {
          if (WRITE_OUT) {
            uint32_t j = i - litf; // match byte index (lanes >= lit)
            if (wave_ballot(voff < vml2) != 0) { // (lanes below lit: unused)
              // runs of 1-, 2-, 4-, 8-byte elements: the period is a power of two
              if (wave_ballot((voff & (voff - 1u)) != 0) == 0)
                j &= voff - 1u;
              else
                j = small_mod(j & 63u, voff);
            }
            const int32_t a = i < litf ? (int32_t)i : (int32_t)(litf + j - voff);
            const uint32_t sidx = idx + 1u + (uint32_t)max(a, 0);
            const uint32_t sword = (uint32_t)__builtin_amdgcn_ds_bpermute(
                (int)((sidx >> 2) * 4u), (int)sw.words);
            const uint32_t byte = sword >> ((sidx & 3u) * 8u);
            if (i < vn2) {
              const uint32_t gb = static_cast<cgptr>(out)[vd + (uint32_t)a];
              out[vd + i] = (uint8_t)(a >= 0 ? byte : gb);
            }
          }
          c += 3u + litf + ((tw & 15u) == 15u ? 1u : 0u);
          vd += vn2;
          continue;
        }
      }
      break;
    }
    if (c >= end)
      break;
    // ---- general path: everything scalar again
    uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane((int)vd);
    // (the fast loop was left either at a token it had read or short of the end)
    if (c + kFastSeqBytes > end)
      tok = uniform((uint32_t)comp[c]);
    ++c;
    uint32_t lit = tok >> 4;
    if (lit == 15 && !read_lsic(comp, c, end, lit, lane)) {
      corrupt = true;
      break;
    }
    if (d + lit > cap || lit > end - c) { // reference :1008
      corrupt = true;
      break;
    }
    if (WRITE_OUT && lit) {
      // short runs (the common case on compressible data): one byte per lane
      if (lit <= kWave) {
        if ((uint32_t)lane < lit)
          out[d + lane] = comp[c + lane];
      } else {
        wave_copy(out + d, comp + c, lit, lane);
      }
    }
    c += lit;
    d += lit;
    if (c < end) { // reference :1035
      if (end - c < 2) {
        corrupt = true;
        break;
      }
      const uint32_t lit_end = c; // the literal run ends where the offset field starts
      const uint32_t offset = uniform((uint32_t)comp[c] | ((uint32_t)comp[c + 1] << 8));
      c += 2;
      uint32_t ml = 4 + (tok & 0x0fu);
      if ((tok & 0x0fu) == 15 && !read_lsic(comp, c, end, ml, lane)) {
        corrupt = true;
        break;
      }
      if (d < offset || d + ml > cap || offset == 0) { // reference :1054
        corrupt = true;
        break;
      }
      if (WRITE_OUT) {
        // Earlier stores of this wave to out[] are ordered before these
        // loads (one wave, in-order vector memory, one L1).
        // Source of the match: normally the already written output.  When the
        // match reaches back only into the literal run of this same sequence
        // (offset <= lit) the very same bytes sit in the compressed stream
        // just before the offset field -- reading them there avoids a
        // store -> load round trip through memory on out[] (reference
        // :1062-1070 does the same from its LDS staging buffer).
        cgptr src = offset <= lit ? comp + (lit_end - offset) : static_cast<cgptr>(out + d - offset);
        gptr dst = out + d;
        if (offset >= ml) {
          if (ml <= kWave) {
            if ((uint32_t)lane < ml)
              dst[lane] = src[lane];
          } else {
            // long match, source and destination do not overlap (offset >= ml)
            wave_copy(dst, src, ml, lane);
          }
        } else if ((offset & (offset - 1u)) == 0) {
          // the period is a power of two (runs of 1-, 2-, 4-, 8-byte elements)
          for (uint32_t i = (uint32_t)lane; i < ml; i += kWave)
            dst[i] = src[i & (offset - 1u)];
        } else {
          for (uint32_t i = (uint32_t)lane; i < ml; i += kWave)
            dst[i] = src[i % offset];
        }
      }
      d += ml;
    }
    vd = d;
  }
  if (lane == 0) {
    if (actual_bytes)
      actual_bytes[chunk] = corrupt ? 0 : vd; // reference :1088-1096
    if (WRITE_OUT && statuses)
      statuses[chunk] = corrupt ? hipcompErrorCannotDecompress : hipcompSuccess;
  }
}

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
  static const Lz4Mode mode = [] {
    const char* e = std::getenv("HIPCOMP_LZ4_SHAPE");
    if (e && std::strcmp(e, "mix") == 0)
      return Lz4Mode::Mix;
    if (e && std::strcmp(e, "far") == 0)
      return Lz4Mode::Far;
    if (e && std::strcmp(e, "farw") == 0)
      return Lz4Mode::FarWide;
    return Lz4Mode::Auto;
  }();
  return mode;
}

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
  static const uint32_t far_groups_per_cu = [] { // (measurement knob: fewer resident waves)
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
  for (int wide = 0; wide < 2; ++wide)
    if (mode == Lz4Mode::Auto || mode == (wide ? Lz4Mode::FarWide : Lz4Mode::Far))
      far_kernel_for(elem_size, wide != 0)<<<dim3(far.groups), dim3(far.waves() * kWave), far.lds_bytes, stream>>>(
          in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, far_tables, (uint32_t)batch, ticket,
          chunks_per_ticket(far), chosen);
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
