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
//   warpMatchAny = 64-step LDS loop, twice  in-window duplicate search through
//   per window (:218-245)                   the hash table itself: lane ids
//                                           posted in reversed lane order, one
//                                           read-back + one ds_bpermute, exact
//                                           fallback only for colliding lanes
//   second warpMatchAny for the insert      insert rule (incl. the wave64
//   (:722-741) + hardware arbitration of    `int` truncation)
//   same-address global_store_short         expressed as ONE lane-permuted masked
//                                           LDS store (see insert_image)
//   shuffleLiterals (:754-791)              one unaligned dword load per lane
//   1 byte/lane literal + match compare     16-byte/lane copies, 4-byte/lane
//                                           match-length compare
//
// One chunk per wavefront (64-thread workgroup): the window loop is a serial
// dependency chain, the 64 lanes are the 64 window positions.
//
// Decoder: reference src/LZ4Kernels.hiph:971-1097 decompressStream.

#include "lz4_launch.hpp"
#include "wave_utils.hpp"

#include <atomic>


namespace hcamd {

namespace {

constexpr uint32_t kNullOffset = 0xFFFFu;


__device__ __forceinline__ uint32_t hash_sum(uint32_t key)
{
  // reference hash() :557-561 before masking
  return __brev(key) + (key ^ 0xc375u);
}

// Write `n` in LZ4's linear small-integer code: n/255 bytes of 0xFF then
// n%255.  (reference writeLSIC :267-278)
__device__ __forceinline__ uint32_t write_lsic(gptr out, uint32_t number, int lane)
{
  const uint32_t num = number / 255u + 1u;
  const uint8_t last = (uint8_t)(number % 255u);
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
#ifdef HC_ABL_NO_STORES
  {
    uint32_t cc = c + 1;
    if (lit_bytes >= 15) cc += (lit_bytes - 15u) / 255u + 1u;
    cc += lit_bytes;
    if (match_bytes > 0) { cc += 2; if (match_bytes >= 19) cc += (match_bytes - 19u) / 255u + 1u; }
    return cc;
  }
#endif
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
// one address (measured, same test), so the whole rule is ONE store whose
// lanes are permuted (ds_bpermute) into priority order:
//   physical lanes  0..31: window lanes >= 32 in the hardware's write order
//                          above (the reference's global_store_short);
//   physical lanes 32..63: window lanes 0..31 in natural order -- they
//                          override the first half wherever a slot also has
//                          a window lane below 32.
// What travels through the permute is the "insert image" of a window lane:
// bits 0..15 the value, bits 16..29 the slot, bit 31 "this lane stores".
// ---------------------------------------------------------------------------
constexpr uint32_t kImageStore = 1u << 31;

__device__ __forceinline__ int make_insert_perm_addr4(int lane)
{
  uint32_t src;
  if (lane < 32) {
    const uint32_t r = (uint32_t)lane;
    src = 32u + (r & 16u) + 4u * (r & 3u) + (3u - ((r >> 2) & 3u));
  } else {
    src = (uint32_t)lane - 32u;
  }
  return (int)(src * 4u);
}

// Insert image of this window lane for an insert of the first n >= 32 window
// lanes.  restore31: lane 31's slot keeps its old content under the rule, so
// lane 31 can carry that old content (slot_old) instead -- used to undo the
// duplicate-search marker of that slot in the same store.
template <int NVMAX>
__device__ __forceinline__ uint32_t insert_image(
    uint32_t hpos, uint32_t pos, uint32_t slot_old, int n, int lane, bool restore31)
{
  // The rule as lane masks (scalar): lanes 0..30 and 32..n-1 store unless
  // they share lane 31's slot; lane 63 of a full 64-lane window stores in
  // any case; lane 31 stores (the old content) only to undo a marker, and
  // not if lane 63 overwrites that slot anyway.
  const uint32_t h31 = read_lane(hpos, 31);
  const uint64_t in31 = wave_ballot(hpos == h31);
  const uint64_t unless31 = lanes_below<NVMAX>(n) & ~(1ull << 31);
  uint64_t always = 0;
  if (NVMAX == 64 && n == 64) {
    always = 1ull << 63;
    if (in31 >> 63)
      restore31 = false;
  }
  if (restore31)
    always |= 1ull << 31;
  const uint64_t store = always | (unless31 & ~in31);
  uint32_t flag;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(flag) : "v"(kImageStore), "s"(store));
  const uint32_t value = lane == 31 ? slot_old : pos; // 16 bits each
  return (hpos << 16) | flag | value;
}

__device__ __forceinline__ void store_insert_image(uint16_t* table, uint32_t image)
{
  if ((int32_t)image < 0)
    table[(image >> 16) & 0x3FFFu] = (uint16_t)image;
}

// n <= 31: plain store of the first n lanes.
__device__ __forceinline__ void insert_short_window(
    uint16_t* table, uint32_t hpos, uint32_t pos, int n, int lane)
{
  if (lane < n)
    table[hpos] = (uint16_t)pos;
}

// table[slot] and ds_bpermute issued back to back, ONE wait for both (left to
// itself the compiler waits for the read, then issues the permute).
__device__ __forceinline__ void lds_read_u16_with_bpermute(
    const uint16_t* slot, int bp_addr4, uint32_t bp_data, uint32_t& slot_value, uint32_t& bp_value)
{
  const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint16_t*)slot;
  asm volatile("ds_read_u16 %0, %2\n\tds_bpermute_b32 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(slot_value), "=&v"(bp_value)
               : "v"(a), "v"(bp_addr4), "v"(bp_data)
               : "memory");
}

// ---------------------------------------------------------------------------
// One window of the match search = 64 consecutive element positions, one per
// lane (reference :847-962).  Its work comes in two halves:
//   table half    hash, table lookup, candidate verification load, search
//                 markers, table insert;
//   decision half first lane with a match (in-window duplicate or verified
//                 table candidate).
// ---------------------------------------------------------------------------
struct Window
{
  uint32_t d;         // first element (wave-uniform)
  int nv;             // lanes holding a position that may start a match (uniform)
  bool valid;         // lane < nv
  uint32_t word;      // the 4 bytes at element d + lane
  uint32_t hpos;      // my table slot
  uint32_t h_old;     // what the slot held before this window
  uint32_t cand;      // element the slot points to
  uint64_t probe;     // lanes whose candidate is usable: its word gets verified (uniform)
  uint32_t cand_word; // 4 bytes at cand (in flight until first use)
  uint32_t next_word; // 4 bytes at element d + nv + lane, in the pipelined walk
                      // at d + 2 nv + lane (in flight)
  uint32_t w_raw;     // marker read back from my slot: lowest lane in it
  uint32_t pimage;    // lane-permuted insert image of the whole window
};

struct Decision
{
  bool match;
  int f;                   // first lane with a match
  uint32_t match_location; // element it matches
};

// FULL: the caller knows that the window has all NVMAX lanes.
template <int S, int NVMAX, bool FULL = false>
__device__ __forceinline__ void window_begin(
    Window& W, uint32_t d, uint32_t word, uint32_t L, uint32_t hmask, int lane)
{
  constexpr uint32_t LVM = (12 + S - 1) / S;
  W.d = d;
  W.nv = FULL ? NVMAX : min(NVMAX, (int)(L - d - LVM)); // >= 1
  W.valid = lane < W.nv;
  W.word = word;
  W.hpos = hash_sum(word) & hmask;
}

// (B) candidate from earlier windows (reference isValidHash :634-663,
// convertIdx :619-632), its 4-byte verify load, then the load of the next
// window's words: the latter is issued AFTER the verify so that waiting for the
// verify (in-order vmcnt) does not wait for it.  Both loads are unconditional
// with a clamped, always readable index so that the compiler can count them.
// FULL: a full window follows this one, so every lane's own position is a
// readable word and needs no clamp.
template <int S, bool FULL = false>
__device__ __forceinline__ void window_candidate(
    Window& W, cgptr in, uint32_t last_word, int lane, bool load_next = true, int ahead = 1)
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
#ifdef HC_ABL_NO_VERIFY
  W.probe = 0;
#else
  // (one ballot per compare, combined as scalars: a ballot of the combined
  // per-lane condition goes through a VGPR)
  W.probe = wave_ballot(W.h_old != kNullOffset) & wave_ballot(back < 65535u / S)
            & lanes_below<64>(W.nv);
#endif
  W.cand = cand;
  const uint32_t own = FULL ? pos : min(pos, last_word);
  uint32_t at; // probe ? cand : own, straight from the scalar lane mask
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(at) : "v"(own), "v"(cand), "s"(W.probe));
  W.cand_word = load_u32_any(in + (size_t)at * S);
  // Next words: not after a window with a match (this one most likely has
  // one too, the words would be dropped, and a load in flight into a
  // register the match path wants to reuse makes that path wait for it).
  // The pipelined walk loads two windows ahead: one step of it is shorter
  // than a trip to HBM.
  if (load_next)
    W.next_word = load_u32_any(in + (size_t)min(pos + (uint32_t)(ahead * W.nv), last_word) * S);
}

// (A) in-window duplicates: lowest lane holding my word, found through the
// hash table itself (no scratch LDS).  Every valid lane posts its lane id into
// its own table slot with the lanes in REVERSED order (`pr` = slot and valid
// flag of the mirrored lane), so that ds_write_b16's "highest lane wins"
// leaves the LOWEST window lane of each slot; reading the slot back names that
// lane.  If it holds my word it is exactly min{u : word_u == word_t};
// otherwise two different words share the slot and the lane is settled by the
// exact fallback in window_decide.
// The same LDS round trip carries the lane permute of the insert image of a
// window WITHOUT a match (all nv lanes, reference :958-962); storing that image
// overwrites every marker but the one of lane 31's slot, whose old content the
// image carries (insert_image).  Nothing is waited for here.
// FULL: all NVMAX lanes are valid and `pr` is the plain slot of the mirrored
// lane -- except that the 64 - NVMAX lanes mirroring the invalid window lanes
// were handed window lane 0's slot: they post into it unconditionally and
// always lose to window lane 0 itself (the highest physical lane), so the
// marker store needs no exec mask.
template <int NVMAX, bool WITH_IMAGE = true, bool FULL = false>
__device__ __forceinline__ void window_markers(
    Window& W, uint16_t* table, uint32_t pr, uint32_t rev_lane, int perm_addr4, int lane)
{
  uint32_t image = 0;
  if (WITH_IMAGE)
    image = insert_image<NVMAX>(
        W.hpos, (W.d + (uint32_t)lane) & 0xFFFFu, W.h_old, W.nv, lane, true);
  lds_lane_exchange_fence();
  if (FULL)
    table[pr] = (uint16_t)rev_lane;
  else if (pr & 0x80000000u)
    table[pr & 0x7FFFFFFFu] = (uint16_t)rev_lane;
  lds_lane_exchange_fence();
  W.w_raw = table[W.hpos];
  if (WITH_IMAGE)
    W.pimage = (uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr4, (int)image);
  lds_lane_exchange_fence();
}

__device__ __forceinline__ uint32_t window_winner(const Window& W, int lane)
{
  // an invalid lane names itself, so it is neither duplicate nor unresolved
  return W.valid ? W.w_raw : (uint32_t)lane;
}

// nw = word of window_winner's lane (one ds_bpermute, issued by the caller).
template <int NVMAX>
__device__ __forceinline__ Decision window_decide(const Window& W, uint32_t nw, int lane)
{
  const uint64_t vmask = lanes_below<NVMAX>(W.nv);
  const uint32_t w = window_winner(W, lane);
  // masks are combined as scalars: each ballot is one v_cmp
  const uint64_t eqmask = wave_ballot(nw == W.word);
  const uint64_t dupmask = eqmask & wave_ballot(w != (uint32_t)lane);
  const uint64_t unres = vmask & ~eqmask;

  // first lane with an equal lower lane (nv if none) and that lower lane
  int f = dupmask ? __builtin_ctzll(dupmask) : W.nv;
  uint32_t mlane = read_lane(w, f & 63);
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
  // earliest lane (< f) with a verified table candidate wins
  // (reference :896-923)
  const uint64_t tmask
      = wave_ballot(W.cand_word == W.word) & W.probe & lanes_below<NVMAX>(f);
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

// False only if window_decide would find no match: every valid lane is alone
// in its slot (then it has no duplicate and nothing is unresolved) and no
// table candidate holds its lane's word.
__device__ __forceinline__ uint64_t window_suspect_lanes(const Window& W, int lane)
{
  return wave_ballot(window_winner(W, lane) != (uint32_t)lane)
         | (wave_ballot(W.cand_word == W.word) & W.probe);
}

// Table state "only the first f lanes of W were inserted", from any state in
// which W's slots hold markers or W's full insert.
template <int NVMAX>
__device__ __forceinline__ void window_insert_first(
    const Window& W, uint16_t* table, int f, int perm_addr4, int lane)
{
  const uint32_t pos16 = (W.d + (uint32_t)lane) & 0xFFFFu;
  if (W.valid)
    table[W.hpos] = (uint16_t)W.h_old;
  lds_lane_exchange_fence();
  if (f >= 32) {
    const uint32_t im = insert_image<NVMAX>(W.hpos, pos16, 0, f, lane, false);
    store_insert_image(table, (uint32_t)__builtin_amdgcn_ds_bpermute(perm_addr4, (int)im));
  } else {
    insert_short_window(table, W.hpos, pos16, f, lane);
  }
}

// The same group without the wait, and the wait as a separate statement that
// hands the two results on: whatever is written between the two runs in the
// shadow of the LDS round trip.  `done_first` is a scalar the caller wants
// computed BEFORE the wait (it pins that computation above it; without the
// operand the compiler may sink it below).
__device__ __forceinline__ void lds_read_u16_with_bpermute_issue(
    const uint16_t* slot, int bp_addr4, uint32_t bp_data, uint32_t& slot_value, uint32_t& bp_value)
{
  const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint16_t*)slot;
  asm volatile("ds_read_u16 %0, %2\n\tds_bpermute_b32 %1, %3, %4"
               : "=&v"(slot_value), "=&v"(bp_value)
               : "v"(a), "v"(bp_addr4), "v"(bp_data)
               : "memory");
}

__device__ __forceinline__ void lds_wait_for(uint32_t& slot_value, uint32_t& bp_value, uint64_t done_first)
{
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(slot_value), "+v"(bp_value) : "s"(done_first) : "memory");
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

// One step of the pipelined walk.  Three windows are in flight: P and Q have
// had their table half and their insert (on the guess that they have no
// match), N is the window behind Q.  N's table half runs while P's
// verification load -- issued two steps ago -- is in its last stretch; P's
// decision is then, nearly always, one cheap test.
//   kWalkOn     P had no match; Q, N and the next window move up
//   kWalkMatch  P has the match D; N's markers and Q's insert are off the
//               table again, P's full insert still has to be cut back
//               (window_insert_first)
//   kWalkDrain  P had no match and no full window follows N: N is inserted,
//               Q and N are still to be decided, in this order
constexpr int kWalkOn = 0, kWalkMatch = 1, kWalkDrain = 2;

// Puts back what W's slots held before W (its markers or its insert).
__device__ __forceinline__ void window_undo(const Window& W, uint16_t* table)
{
  if (W.valid)
    table[W.hpos] = (uint16_t)W.h_old;
  lds_lane_exchange_fence();
}

template <int S, int NVMAX>
__device__ __forceinline__ int walk_step(
    const Window& P, const Window& Q, Window& N, Decision& D, uint16_t* table, cgptr in,
    uint32_t L, uint32_t last_word, uint32_t hmask, uint32_t rev_lane, int rev_addr4_full,
    int perm_addr4, int lane)
{
  constexpr uint32_t LVM = (12 + S - 1) / S;
  uint32_t prN;
  // every window loads the words of the window two behind it: N's are P's
  window_begin<S, NVMAX, true>(N, Q.d + (uint32_t)NVMAX, P.next_word, L, hmask, lane);
  lds_read_u16_with_bpermute_issue(table + N.hpos, rev_addr4_full, N.hpos, N.h_old, prN);
  // in the shadow of that LDS round trip: P's test (below)
  const uint64_t suspect = window_suspect_lanes(P, lane);
  lds_wait_for(N.h_old, prN, suspect);
  window_candidate<S, true>(N, in, last_word, lane, true, 2);
  window_markers<NVMAX, true, true>(N, table, prN, rev_lane, perm_addr4, lane);
  // Nearly always P has no slot shared by two lanes (so neither a duplicate
  // nor anything for the exact fallback) and no verified candidate: one test
  // for all of that (and for "N is the last full window") instead of the
  // full decision.  A taken branch costs a lone wave ~24 cycles, an untaken
  // one ~10 (scripts/probes/branch_cost.hip).
  const bool last = (int)(L - N.d - LVM) < 2 * NVMAX;
  if (__builtin_expect((suspect != 0) | last, 0)) {
    const uint32_t nwP = (uint32_t)__builtin_amdgcn_ds_bpermute(
        (int)(window_winner(P, lane) * 4u), (int)P.word);
    D = window_decide<NVMAX>(P, nwP, lane);
    if (D.match) {
      window_undo(N, table); // newest first: N read its slots after Q's insert
      window_undo(Q, table);
      return kWalkMatch;
    }
    if (last) {
      store_insert_image(table, N.pimage);
      return kWalkDrain;
    }
  }
  store_insert_image(table, N.pimage);
  return kWalkOn;
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
// of the time of a 1 KiB chunk.
__device__ __forceinline__ uint32_t take_ticket(uint32_t* ticket, uint32_t count)
{
  uint32_t t;
  const uint32_t zero = 0, one = count;
  asm volatile("s_mov_b64 s[20:21], exec\n\t"
               "s_mov_b64 exec, 1\n\t"
               "global_atomic_add %0, %1, %2, %3 sc0\n\t"
               "s_waitcnt vmcnt(0)\n\t"
               "s_mov_b64 exec, s[20:21]"
               : "=&v"(t)
               : "v"(zero), "v"(one), "s"(ticket)
               : "s20", "s21", "memory");
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
}

// Workgroup shape: the hash table (ht_size x u16, 32 KiB for 64 KiB chunks)
// is the only LDS user and LDS is what limits residency.  The CU allocates
// LDS in 1280-byte granules, so five separate 32 KiB workgroups do not fit
// into its 160 KiB (5 x 26 granules) but ONE workgroup of five waves with
// 5 x 32 KiB does.  Each wave of the workgroup owns one table and takes
// chunks from a global ticket counter until the batch is exhausted; the waves
// never synchronise with each other.
template <int S>
__global__ __launch_bounds__(kLz4MaxWavesPerGroup * kWave) void lz4_compress_kernel(
    const uint8_t* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    uint8_t* const* __restrict__ out_ptrs,
    size_t* __restrict__ out_bytes,
    const uint32_t ht_size,
    const uint32_t table_stride,
    const uint32_t batch,
    uint32_t* __restrict__ ticket,
    const uint32_t chunks_per_ticket)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  constexpr uint32_t LVM = (12 + S - 1) / S; // last valid match, elements
  constexpr int INV = 3 / S;                 // lanes without a full 4-byte word
  constexpr int NVMAX = kWave - INV;

  const int lane = lane_id();
  const uint32_t wave = uniform((uint32_t)(threadIdx.x >> 6));
  uint8_t* const my_smem = smem + wave * table_stride;
  uint16_t* const table = reinterpret_cast<uint16_t*>(my_smem);
  const uint32_t hmask = ht_size - 1;
  const int perm_addr4 = make_insert_perm_addr4(lane);
  const uint32_t rev_lane = 63u - (uint32_t)lane;
  const int rev_addr4 = (int)(rev_lane * 4u);
  // for full windows: the lanes that mirror the invalid window lanes take
  // window lane 0's slot instead (window_markers)
  const int rev_addr4_full = lane < INV ? 0 : rev_addr4;

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

  // ---- LDS init (reference :815-818 fills the table with NULL_OFFSET)
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
  // the previous window had no match: walk the following ones pipelined
  bool cold = false;

  uint32_t token_start = 0; // first element not yet written out
  while (d < L) {
    if (cold && (int)(L - d - LVM) >= 3 * NVMAX) {
      // ---- pipelined walk over match-less full windows (here: three full
      // windows lie ahead).  A window's insert is done at once, on the guess
      // that it has no match; its decision follows two windows later, when
      // its verification load has had two steps to arrive.  A match rolls the
      // table back and drops the newer windows.
      Window A, B, C, W; // W: the window the walk ended on
      Decision D;
      {
        uint32_t pr;
        window_begin<S, NVMAX, true>(A, d, next, L, hmask, lane);
        lds_read_u16_with_bpermute(table + A.hpos, rev_addr4_full, A.hpos, A.h_old, pr);
        window_candidate<S, true>(A, in, last_word, lane, true, 2);
        window_markers<NVMAX, true, true>(A, table, pr, rev_lane, perm_addr4, lane);
        store_insert_image(table, A.pimage);
        const uint32_t wordsB = load_u32_any(
            in + (size_t)min(d + (uint32_t)(NVMAX + lane), last_word) * S);
        window_begin<S, NVMAX, true>(B, d + (uint32_t)NVMAX, wordsB, L, hmask, lane);
        lds_read_u16_with_bpermute(table + B.hpos, rev_addr4_full, B.hpos, B.h_old, pr);
        window_candidate<S, true>(B, in, last_word, lane, true, 2);
        window_markers<NVMAX, true, true>(B, table, pr, rev_lane, perm_addr4, lane);
        store_insert_image(table, B.pimage);
      }
      // the three windows in flight rotate through the roles (no copies)
      Window Qd, Nd; // on kWalkDrain: the two undecided windows
      int r;
      for (;;) {
        r = walk_step<S, NVMAX>(A, B, C, D, table, in, L, last_word, hmask, rev_lane,
                                rev_addr4_full, perm_addr4, lane);
        if (r != kWalkOn) {
          W = A; Qd = B; Nd = C;
          break;
        }
        r = walk_step<S, NVMAX>(B, C, A, D, table, in, L, last_word, hmask, rev_lane,
                                rev_addr4_full, perm_addr4, lane);
        if (r != kWalkOn) {
          W = B; Qd = C; Nd = A;
          break;
        }
        r = walk_step<S, NVMAX>(C, A, B, D, table, in, L, last_word, hmask, rev_lane,
                                rev_addr4_full, perm_addr4, lane);
        if (r != kWalkOn) {
          W = C; Qd = A; Nd = B;
          break;
        }
      }
      if (r == kWalkDrain) {
        // the oldest window had no match; the two behind it, oldest first
        const uint32_t nwQ = (uint32_t)__builtin_amdgcn_ds_bpermute(
            (int)(window_winner(Qd, lane) * 4u), (int)Qd.word);
        D = window_decide<NVMAX>(Qd, nwQ, lane);
        if (D.match) {
          window_undo(Nd, table);
          W = Qd;
        } else {
          const uint32_t nwN = (uint32_t)__builtin_amdgcn_ds_bpermute(
              (int)(window_winner(Nd, lane) * 4u), (int)Nd.word);
          D = window_decide<NVMAX>(Nd, nwN, lane);
          W = Nd;
        }
      }
      if (D.match) {
        window_insert_first<NVMAX>(W, table, D.f, perm_addr4, lane);
        emit_match<S>(out, c, in, token_start, W.d, W.word, D, L, lane, d);
        next = load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
        token_start = d;
        cold = false;
        continue;
      }
      // (only after a drain: W is the last window, its words-two-behind load
      // belongs to the window after the next one, so the next one's words
      // are those its predecessor loaded)
      d = W.d + (uint32_t)NVMAX;
      next = Qd.next_word;
      // (at least one more window follows, not three full ones)
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
    lds_read_u16_with_bpermute(
        table + P.hpos, rev_addr4, P.hpos | (P.valid ? 0x80000000u : 0u), P.h_old, pr);
    window_candidate<S>(P, in, last_word, lane, cold);
    // (no insert image yet: after a window with a match this one most likely
    // has one too and would not use it)
    window_markers<NVMAX, false>(P, table, pr, rev_lane, perm_addr4, lane);
    const uint32_t nw = (uint32_t)__builtin_amdgcn_ds_bpermute(
        (int)(window_winner(P, lane) * 4u), (int)P.word);
    const Decision D = window_decide<NVMAX>(P, nw, lane);
    if (D.match) {
      window_insert_first<NVMAX>(P, table, D.f, perm_addr4, lane);
      emit_match<S>(out, c, in, token_start, P.d, P.word, D, L, lane, d);
      next = load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
      token_start = d;
      cold = false;
    } else {
      // no match in this window (reference :958-962): all nv lanes go in
      window_insert_first<NVMAX>(P, table, P.nv, perm_addr4, lane);
      d += (uint32_t)P.nv;
      next = cold ? P.next_word
                  : load_u32_any(in + (size_t)min(d + (uint32_t)lane, last_word) * S);
      cold = true;
    }
  }
  if (lane == 0)
    out_bytes[chunk] = c;
  } // next chunk of this ticket
  if (!ticket)
    break;
 } // next ticket
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

  uint32_t c = 0, d = 0;
  bool corrupt = false;
  // Short sequences are parsed from a register window of the stream
  // (StreamWindow), i.e. without a memory round trip in the chain that leads
  // from one token to the next.
  StreamWindow sw;
  while (c < end) {
    uint32_t tok = 0;
    bool tok_known = false;
    // ---- fast paths: token, up to 14 literals, offset and at most one match
    // length byte inside the stream; literals + match at most 64 bytes (one
    // byte per lane).  Anything else takes the general path below.
    if (c + kFastSeqBytes <= end) {
      sw.ensure(comp, c, end, kFastSeqBytes, lane);
      const uint32_t idx = c - sw.base;
      tok = sw.bytes_at(idx) & 0xFFu;
      tok_known = true;
      const uint32_t litf = tok >> 4, mlc = tok & 15u;
      if (litf < 15u) { // literal length in the token itself
        // offset (2 bytes) and the byte behind it: a length byte if the
        // token's match field is 15
        const uint32_t t2 = sw.bytes_at(idx + 1u + litf);
        const uint32_t off = t2 & 0xFFFFu;
        // One byte per lane for literals AND match: lane i < lit carries
        // literal i, lane lit + j match byte j.  `a` is the source as an index
        // relative to d: >= 0 means a literal of this very sequence, i.e. a
        // byte of the stream window (for match bytes too: no round trip
        // through the output); < 0 means earlier output.
        const uint32_t i = (uint32_t)lane;
        // (1) the common one: match length in the token, output fits, offset
        // inside what exists, source and destination do not overlap
        const uint32_t ml1 = mlc + 4u;
        if ((mlc < 15u) & (d + litf + ml1 <= cap) & (off != 0u) & (off <= d + litf) & (off >= ml1)) {
          if (WRITE_OUT) {
            const int32_t a = (int32_t)(i - (i < litf ? 0u : off));
            const uint32_t sidx = idx + 1u + (uint32_t)max(a, 0);
            const uint32_t sword = (uint32_t)__builtin_amdgcn_ds_bpermute(
                (int)((sidx >> 2) * 4u), (int)sw.words);
            uint32_t byte = sword >> ((sidx & 3u) * 8u);
            if (off > litf) { // some match bytes come from earlier output
              // Earlier stores of this wave to out[] are ordered before this
              // load (one wave, in-order vector memory, one L1).
              const uint32_t gb = *(static_cast<cgptr>(out + d) + min(a, -1));
              byte = a >= 0 ? byte : gb;
            }
            if (i < litf + ml1)
              out[d + i] = (uint8_t)byte;
          }
          c += 1u + litf + 2u;
          d += litf + ml1;
          continue;
        }
        // (2) one length byte (not 255: a second one would follow) and / or
        // a match that overlaps itself: its source repeats with period `off`
        const bool has_ext = mlc == 15u;
        const uint32_t ext = (t2 >> 16) & 0xFFu;
        const uint32_t ml2 = ml1 + (has_ext ? ext : 0u);
        const uint32_t n2 = litf + ml2;
        if (!(has_ext & (ext == 255u)) & (n2 <= (uint32_t)kWave) & (d + n2 <= cap) & (off != 0u)
            & (off <= d + litf)) {
          if (WRITE_OUT) {
            uint32_t j = i - litf; // match byte index (lanes >= lit)
            if (off < ml2)
              j = small_mod(j & 63u, off); // (lanes below lit: unused)
            const int32_t a = i < litf ? (int32_t)i : (int32_t)(litf + j - off);
            const uint32_t sidx = idx + 1u + (uint32_t)max(a, 0);
            const uint32_t sword = (uint32_t)__builtin_amdgcn_ds_bpermute(
                (int)((sidx >> 2) * 4u), (int)sw.words);
            uint32_t byte = sword >> ((sidx & 3u) * 8u);
            if (off > litf) {
              const uint32_t gb = *(static_cast<cgptr>(out + d) + min(a, -1));
              byte = a >= 0 ? byte : gb;
            }
            if (i < n2)
              out[d + i] = (uint8_t)byte;
          }
          c += 1u + litf + 2u + (has_ext ? 1u : 0u);
          d += n2;
          continue;
        }
      }
    }
    if (!tok_known)
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
        } else {
          for (uint32_t i = (uint32_t)lane; i < ml; i += kWave)
            dst[i] = src[i % offset];
        }
      }
      d += ml;
    }
  }
  if (lane == 0) {
    if (actual_bytes)
      actual_bytes[chunk] = corrupt ? 0 : d; // reference :1088-1096
    if (WRITE_OUT && statuses)
      statuses[chunk] = corrupt ? hipcompErrorCannotDecompress : hipcompSuccess;
  }
}

} // namespace

// ---- launchers -----------------------------------------------------------

size_t lz4_compress_lds_bytes(uint32_t ht_size)
{
  return (ht_size * 2 + 15) & ~15u;
}

namespace {

// Per-device facts and one-time setup, looked up by the calling thread's
// current device (one process may drive several GPUs, one thread each, as the
// reference's callers do).  Both steps are idempotent, so a race between two
// first callers on one device is harmless.
constexpr int kMaxDevices = 64;
std::atomic<int> g_num_cus[kMaxDevices];
std::atomic<int> g_lds_raised[kMaxDevices]; // 0 = not yet, 1 = done, < 0 = -hipError

int current_device()
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
    dev = 0;
  return dev;
}

int num_cus_of_current_device()
{
  const int dev = current_device();
  int n = g_num_cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// more than 64 KiB of dynamic LDS has to be asked for, once per kernel and device
hipError_t raise_dynamic_lds_limit()
{
  const int dev = current_device();
  const int state = g_lds_raised[dev].load(std::memory_order_acquire);
  if (state == 1)
    return hipSuccess;
  if (state < 0)
    return (hipError_t)(-state);
  hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_compress_kernel<1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (r == hipSuccess)
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_compress_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (r == hipSuccess)
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(lz4_compress_kernel<4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  g_lds_raised[dev].store(r == hipSuccess ? 1 : -(int)r, std::memory_order_release);
  return r;
}

} // namespace

Lz4CompressShape lz4_compress_shape(uint32_t ht_size, size_t batch)
{
  const int num_cus = num_cus_of_current_device();
  constexpr uint32_t kLdsPerCu = 160u * 1024u;
  Lz4CompressShape sh;
  sh.table_stride = (uint32_t)lz4_compress_lds_bytes(ht_size);
  uint32_t w = kLdsPerCu / sh.table_stride;
  if (w > (uint32_t)kLz4MaxWavesPerGroup)
    w = kLz4MaxWavesPerGroup;
  if ((size_t)w > batch)
    w = (uint32_t)batch;
  sh.waves = w;
  sh.lds_bytes = w * sh.table_stride;
  // enough workgroups to occupy every CU; late ones find the ticket counter
  // exhausted and leave at once
  const size_t per_cu = kLdsPerCu / sh.lds_bytes;
  const size_t want = (batch + w - 1) / w;
  const size_t cap = (size_t)num_cus * (per_cu > 4 ? 4 : per_cu);
  sh.groups = (uint32_t)(want < cap ? want : cap);
  return sh;
}

hipError_t lz4_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, uint32_t ht_size,
    size_t batch, int elem_size, uint32_t* ticket, size_t max_chunk_bytes, hipStream_t stream)
{
  const Lz4CompressShape sh = lz4_compress_shape(ht_size, batch);
  // about 16 KiB of input per ticket, but at least 4 tickets per wave so
  // that the last ones even out the load
  uint32_t per_ticket = 1;
  const size_t all_waves = (size_t)sh.groups * sh.waves;
  while (per_ticket < 64 && (size_t)per_ticket * (max_chunk_bytes ? max_chunk_bytes : 1) < 16384
         && (size_t)per_ticket * 2 * 4 * all_waves <= batch)
    per_ticket *= 2;
  // ticket == nullptr: no persistent workgroups, one chunk per wave
  const dim3 grid(ticket ? sh.groups : (unsigned)((batch + sh.waves - 1) / sh.waves)), block(sh.waves * kWave);
  if (ticket) {
    const hipError_t e = hipMemsetAsync(ticket, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess)
      return e;
  }
  const hipError_t raised = raise_dynamic_lds_limit();
  if (raised != hipSuccess)
    return raised;
  switch (elem_size) {
  case 1:
    lz4_compress_kernel<1><<<grid, block, sh.lds_bytes, stream>>>(
        in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, sh.table_stride, (uint32_t)batch, ticket, per_ticket);
    break;
  case 2:
    lz4_compress_kernel<2><<<grid, block, sh.lds_bytes, stream>>>(
        in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, sh.table_stride, (uint32_t)batch, ticket, per_ticket);
    break;
  default:
    lz4_compress_kernel<4><<<grid, block, sh.lds_bytes, stream>>>(
        in_ptrs, in_bytes, out_ptrs, out_bytes, ht_size, sh.table_stride, (uint32_t)batch, ticket, per_ticket);
    break;
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
