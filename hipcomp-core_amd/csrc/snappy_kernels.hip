// snappy_kernels.hip -- gfx950 kernels of the batched raw-Snappy codec.
//
// Encoder: compressed bytes are those of the reference's wave64 encoder
// (reference src/snappy/compression.hiph:190-385), by a different mechanism:
//
//   reference                                 here
//   ----------------------------------------  -------------------------------
//   128-thread block per chunk: wave 1        one wave per chunk does both
//   searches, wave 0 emits the previous       (the two were strictly
//   result, two __syncthreads per step        alternating anyway); 8 KiB LDS
//                                             per chunk -> 20 chunks per CU
//   HashMatchAny: 12 ballots + per-lane       lanes post their id into their
//   64-bit xor/or per window (:157-172)       own hash-map slot (slot == hash,
//                                             so no false sharing), read back
//                                             the highest lane of the slot;
//                                             only slots with >= 2 lanes get
//                                             an exact ballot mask
//   unaligned_load32: two aligned loads +     one unaligned global_load_dword
//   funnel shift
//   byte-per-lane literal copy                16-byte/lane copies
//
// The hash-map update mask `(2ULL << literal_cnt) - 1` with literal_cnt == 64
// is a shift by the type width in the reference (:240); on gfx9 it evaluates
// to 1 (v_lshlrev_b64 uses shift & 63), i.e. only lane 0 updates the map for a
// window without a match.  Reproduced explicitly.
//
// Decoder: the Snappy format as the reference decodes it (reference
// src/snappy/decompression.hiph:106-211, decompression_decode.hiph:72-150),
// one wave per chunk; see DESIGN.md for the three places where undefined
// behaviour of the reference is replaced by an error status.

#include "snappy_launch.hpp"
#include "lz4_launch.hpp" // num_cus_of_current_device
#include "placement.hiph"
#include "wave_utils.hpp"

#include <atomic>

namespace hcamd {

namespace {

constexpr uint32_t kHashBits = 12;
constexpr uint32_t kHashEntries = 1u << kHashBits;
constexpr uint32_t kMaxLiteral = 256;
constexpr uint32_t kMaxCopyDistance = 32768;

__device__ __forceinline__ uint32_t snap_hash(uint32_t v)
{
  return (v * ((1u << 20) + 0x2a00u + 0x6au + 1u)) >> (32 - kHashBits);
}

// lowest set bit, -1 for 0 (s_ff1_i32_b64 as it is; __builtin_ctzll(0) is undefined)
__device__ __forceinline__ int first_set_or_minus_one(uint64_t m)
{
  int r;
  asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m));
  return r;
}

// number of lanes below mine that are set in m
__device__ __forceinline__ uint32_t lanes_set_below(uint64_t m)
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// lanes of a window whose table candidates the encoder fetches before the others'
constexpr int kFirstLanes = 8;
// bytes a window must have ahead of it for the straight path (the window's words
// of all lanes, the general Match60 behind it)
constexpr uint32_t kStraightMargin = 64 + 8 + 64 + 8;
// lanes whose candidates one trip of the straight path fetches; it looks at the
// words of lanes 0..kSpan+11, which come from the window before if the elements
// moved on by no more than kStraightReach bytes
#ifndef HC_SNAPPY_SPAN
#define HC_SNAPPY_SPAN 52 // (measurement builds: 16..52; 32: 71.2, 40: 77.4, 48: 82.7, 52: 84.0 GB/s on text)
#endif
constexpr int kSpan = HC_SNAPPY_SPAN;
// pick() below takes the first event at or above `start` with `start` clamped to lane 63: with an event AT lane
// 63 and an element that ends beyond lane 63 it would take that same event again and again (an endless loop,
// met in round 3 with a 64-lane span).  Lanes from kSpan on never report an event, so the span has to stop short
// of lane 63.
static_assert(kSpan >= 8 && kSpan < 64, "the straight path's pick() relies on bit 63 of `events` never being set");
constexpr uint32_t kStraightReach = 64; // (two registers of words: all 64 lanes of the next window are there)

// One chunk by the calling wave: src[0 .. len) -> dst, -> the compressed bytes.  hash_map: kHashEntries
// uint16 in LDS.
__device__ __forceinline__ uint32_t snappy_encode_chunk(
    cgptr __restrict__ src, const uint32_t len, gptr __restrict__ dst, uint16_t* hash_map, const int lane)
{
  // varint of the uncompressed length (reference :316-322)
  uint32_t c = 0;
  {
    uint32_t v = len;
    while (v > 0x7f) {
      if (lane == 0)
        dst[c] = (uint8_t)(v | 0x80);
      ++c;
      v >>= 7;
    }
    if (lane == 0)
      dst[c] = (uint8_t)v;
    ++c;
  }
  // hash map = 0 (reference :330-334)
  {
    u32x4 z = {0, 0, 0, 0};
    u32x4* p = reinterpret_cast<u32x4*>(hash_map);
    for (uint32_t i = (uint32_t)lane; i < kHashEntries * 2 / 16; i += kWave)
      p[i] = z;
  }

  const uint64_t lane_bit = 1ull << lane;
  const uint64_t below_me = lane_bit - 1;
  const uint32_t last_word = len >= 4 ? len - 4 : 0; // highest readable dword start

  uint32_t pos = 0;
  if (len < 4) {
    // no 4-byte word exists: the reference's search finds nothing and the
    // chunk leaves as one literal (and no lane may load a dword from it)
    if (len > 0) {
      if (lane == 0)
        dst[c] = (uint8_t)((len - 1) << 2);
      if ((uint32_t)lane < len)
        dst[c + 1 + lane] = src[lane];
      c += 1 + len;
    }
    pos = len;
  }
  // Words of the window at `pos` from memory, asked for as soon as `pos` is
  // known (`next`), and the words the straight path works on (`wnd`): those of
  // `next` the first time, from then on the words of the window before moved
  // down by the bytes the element took (ds_bpermute; the lanes that path looks
  // at are all there), so that the load is off the chain from element to element.
  uint32_t next = len >= 4 ? load_u32_any(src + min((uint32_t)lane, last_word)) : 0;
  // (and the 64 words behind them: the elements of one trip may move on by more
  // than the lanes the straight path looks at leave of a single window)
  uint32_t next_hi = len >= 4 ? load_u32_any(src + min(64u + (uint32_t)lane, last_word)) : 0;
  uint32_t wnd = next;
  while (pos < len) {
    const uint32_t pos0 = pos;
    uint32_t copy_len = 0, distance = 0, lit = 0;
    bool straight = false;
    // ---- The common stretch of data that compresses, in a straight line
    // (reference FindFourByteMatch :190-246, Match60 :251-269, StoreLiterals /
    // StoreCopy :73-151).  ONE trip to memory serves several elements: the first
    // kSpan lanes look their candidates up and fetch 16 bytes there, which hold
    // the match length as well if the match is shorter than 16 bytes; then the
    // elements are taken off one after the other (the next one's window starts
    // where the match ended, its first hit is the next lane with a hit) for as
    // long as they stay inside those lanes.  Literals come from the window's
    // registers.  The hash-map update is posted for all kSpan lanes at once --
    // the read-back is also the test for equal hashes: the highest lane of a
    // slot survives, so a lane that reads another lane's position has a higher
    // lane with its hash, and the elements stop in front of such a lane (its
    // candidate, and what the lanes above it would have to see, are not what
    // this path assumes) -- and afterwards set right: only the lanes up to each
    // hit stay.  Whatever else a window holds is left to the general code below
    // with the hash map as it was.
    if (pos0 + kStraightMargin <= len) {
      const uint32_t my = pos0 + (uint32_t)lane;
      const uint32_t data32 = wnd;
      const uint32_t hash = snap_hash(data32);
      // the words 4, 8 and 12 bytes on
      const uint32_t d1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane + 4u) & 63u) * 4, (int)data32);
      const uint32_t d2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane + 8u) & 63u) * 4, (int)data32);
      const uint32_t d3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane + 12u) & 63u) * 4, (int)data32);
      uint32_t h_old = 0;
      // what a lane says about itself, in one word: bits 30-31 code (1: a match
      // shorter than 16 bytes, 3: a longer one, 2: a higher lane has my hash),
      // 26-29 copy length - 4, 24-25 bytes of the copy element, 0-23 those bytes
      uint32_t word_of_lane = 0;
      uint32_t dist = 0;
      bool shares = false; // a higher lane (of the kSpan) has my hash
      if (lane < kSpan) {
        h_old = hash_map[hash];
        uint32_t toff = (pos0 & ~0xffffu) | h_old;
        if (toff >= pos0)
          toff = (toff >= 0x10000u) ? toff - 0x10000u : pos0;
        const bool tprobe = toff < pos0 && toff + kMaxCopyDistance >= my;
        // (a lane without a candidate reads the head of the chunk: no branch around the load)
        u32x4 cand = load_u128_any(src + (tprobe ? toff : 0u));
        lds_lane_exchange_fence();
        hash_map[hash] = (uint16_t)my;
        lds_lane_exchange_fence();
        uint32_t posted = hash_map[hash];
        lds_lane_exchange_fence();
        // (the candidates are looked at only now: left to itself the compiler waits
        // for them before it posts)
        asm volatile("" : "+v"(cand.x), "+v"(cand.y), "+v"(cand.z), "+v"(cand.w), "+v"(posted));
        dist = my - toff;
        // first differing byte among bytes 4..15 of the match (v_ffbl_b32: -1 for 0)
        uint32_t f1, f2, f3;
        asm("v_ffbl_b32 %0, %1" : "=v"(f1) : "v"(cand.y ^ d1));
        asm("v_ffbl_b32 %0, %1" : "=v"(f2) : "v"(cand.z ^ d2));
        asm("v_ffbl_b32 %0, %1" : "=v"(f3) : "v"(cand.w ^ d3));
        const uint32_t extra = min(min(min(f1, f2 | 32u), f3 | 64u) >> 3, 12u); // 0..11, 12: all 12 bytes equal
        const uint32_t code = (tprobe && cand.x == data32) ? (extra < 12u ? 1u : 3u) : 0u;
        shares = posted != (my & 0xFFFFu);
        // the copy element (reference StoreCopy :118-151)
        const bool two = (int32_t)((extra - 8u) & (dist - 2048u)) < 0; // extra < 8 and dist < 2048
        const uint32_t tag2 = (((dist & 0x700u) >> 3) | (extra << 2) | 0x01u) | ((dist & 0xFFu) << 8) | (2u << 24);
        const uint32_t tag3 = (((extra + 3u) << 2) | 0x2u) | ((dist & 0xFFFFu) << 8) | (3u << 24);
        word_of_lane = (two ? tag2 : tag3) | (extra << 26) | (code << 30);
      }
      const uint64_t events = wave_ballot(word_of_lane >= (1u << 30));
      const uint64_t sharing = wave_ballot(shares);
      // The elements, one after the other: which lanes they start and hit at is
      // all the loop notes.  An element is taken if no two of its lanes (start
      // .. hit) have one hash -- the later one might hit the earlier one -- and
      // none of them has the hash of a lane an earlier element of the trip has
      // put into the hash map (what it looked up was read before that); only a
      // lane that has a higher lane with its hash can be either, and for those
      // the lanes with that hash are found with a ballot.  A match of 16 bytes
      // or more ends the trip and is written on its own below.
      uint32_t start = 0;                        // lane at which the next element's window starts
      uint64_t hit_lanes = 0, start_lanes = 0;   // (as bit sets) of the elements taken
      uint64_t stale = 0;                        // lanes with the hash of a lane taken so far
      int t;
      uint64_t touched;
      auto pick = [&]() -> uint32_t {
        t = first_set_or_minus_one(events & (~0ull << min(start, 63u))); // (bit 63 of events is never set: kSpan < 64, static_assert above)
        if (t < 0)
          return 0u;
        const uint64_t range = (2ull << t) - (1ull << start); // lanes start..t
        uint64_t clash = stale & range;
        touched = 0;
        for (uint64_t todo = sharing & range; todo != 0; todo &= todo - 1) {
          const int u = __builtin_ctzll(todo);
          const uint64_t same_hash = wave_ballot(hash == read_lane(hash, u));
          clash |= same_hash & range & (~1ull << u);
          touched |= same_hash;
        }
        return clash == 0 ? read_lane(word_of_lane, t) : 0u;
      };
      // What follows a hit lane's element, for all lanes at once: bits 0-6 where
      // the next element's window starts, 8-14 its hit lane (the first event at
      // or above that; 127: none), bit 31: that next element is one the loop can
      // take without looking -- a short match, and no lane from its start to its
      // hit has a higher lane with its hash (so none of its lanes can clash with
      // anything).  While that holds an element costs the loop one v_readlane.
      const uint32_t after = (uint32_t)lane + 4u + ((word_of_lane >> 26) & 15u);
      auto first_at_or_above = [&](uint64_t mask) -> uint32_t { // (per lane, from `after`; 127: none)
        const uint64_t m = mask >> (after & 63u);
        uint32_t lo, hi;
        asm("v_ffbl_b32 %0, %1" : "=v"(lo) : "v"((uint32_t)m));
        asm("v_ffbl_b32 %0, %1" : "=v"(hi) : "v"((uint32_t)(m >> 32)));
        const uint32_t r = min(lo, hi | 32u);
        return (after < 64u && r != ~0u) ? after + r : 127u;
      };
      const uint32_t next_hit = first_at_or_above(events);
      bool easy;
      {
        const uint32_t next_sharing = first_at_or_above(sharing);
        const uint32_t next_word = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((next_hit & 63u) * 4u), (int)word_of_lane);
        easy = next_hit < 64u && (next_word >> 30) == 1u && next_sharing > next_hit;
      }
      uint32_t follows = after | (next_hit << 8) | (easy ? 1u << 31 : 0u);
      uint32_t about = pick();
      while ((about >> 30) == 1u) {
        if (touched != 0) {
          // lanes with the hash of one just taken are no longer easy to pass
          stale |= touched;
          follows = after | (next_hit << 8) | ((easy && first_at_or_above(stale) > next_hit) ? 1u << 31 : 0u);
        }
        // The elements that can be taken without a look, one after the other: four scalar
        // instructions each (as in the LZ4 encoder, lz4_far.hiph).  `then` = what follows the hit
        // lane t; while its sign says "easy": the next element starts at then[5:0] and hits at the
        // lane in bits 8-13 -- s_bitset1_b64 and v_readlane look at the low six bits of their index only.
        {
          uint32_t then, tx = (uint32_t)t;
          // (the element pick() found is noted here, in front of the first v_readlane: `follows` may have
          // been made by the instruction in front of this statement -- check_asm_hazards.py H9)
          asm volatile("s_bitset1_b64 %[hits], %[t]\n\t"
                       "s_bitset1_b64 %[starts], %[start]\n\t"
                       "v_readlane_b32 %[then], %[follows], %[t]\n\t"
                       "s_cmp_lt_i32 %[then], 0\n\t"
                       "s_cbranch_scc0 2f\n"
                       "1:\n\t"
                       "s_lshr_b32 %[t], %[then], 8\n\t"
                       "s_bitset1_b64 %[starts], %[then]\n\t"
                       "s_bitset1_b64 %[hits], %[t]\n\t"
                       "v_readlane_b32 %[then], %[follows], %[t]\n\t"
                       "s_cmp_lt_i32 %[then], 0\n\t"
                       "s_cbranch_scc1 1b\n"
                       "2:"
                       : [then] "=&s"(then), [t] "+s"(tx), [starts] "+s"(start_lanes), [hits] "+s"(hit_lanes)
                       : [follows] "v"(follows), [start] "s"(start)
                       : "scc");
          t = (int)(tx & 63u);
          start = then & 127u; // (= t + 4 + copy length - 4)
        }
        about = pick();
      }
      // Their bytes, all at once: [literal tag, literals] copy element per
      // element, in lane order.  Where a lane's byte goes is a count of the
      // lanes below it -- literal lanes, literal tags (one per element that has
      // literals), copy elements (two bytes each, three for some) -- so every
      // literal lane writes its own byte, the first one of an element its tag in
      // front of that, the hit lane its copy element.
      if (hit_lanes != 0) {
        const uint64_t lits = ((hit_lanes << 1) - start_lanes) & ~hit_lanes; // lanes start..hit-1 of every element
        const uint64_t firsts = start_lanes & ~hit_lanes;                    // (an element without literals starts at its hit)
        const uint64_t threes = wave_ballot(((word_of_lane >> 24) & 3u) == 3u) & hit_lanes;
        const bool is_hit = ((hit_lanes >> lane) & 1ull) != 0, is_lit = ((lits >> lane) & 1ull) != 0;
        const bool is_first = ((firsts >> lane) & 1ull) != 0, is_three = ((threes >> lane) & 1ull) != 0;
        const uint32_t at = c + lanes_set_below(lits) + lanes_set_below(firsts) + (is_first ? 1u : 0u)
                            + 2u * lanes_set_below(hit_lanes) + lanes_set_below(threes);
        if (is_hit)
          *reinterpret_cast<HC_GLOBAL uint16_t __attribute__((aligned(1)))*>(dst + at) = (uint16_t)word_of_lane;
        if (is_three)
          dst[at + 2u] = (uint8_t)(word_of_lane >> 16);
        if (is_lit)
          dst[at] = (uint8_t)data32;
        if (is_first) // its literals run up to the element's hit lane
          dst[at - 1u] = (uint8_t)(((uint32_t)__builtin_ctzll((hit_lanes >> lane) | (1ull << 63)) - 1u) << 2);
        c += (uint32_t)__builtin_popcountll(lits) + (uint32_t)__builtin_popcountll(firsts)
             + 2u * (uint32_t)__builtin_popcountll(hit_lanes) + (uint32_t)__builtin_popcountll(threes);
      }
      if (__builtin_expect(about >= (3u << 30), 0)) {
        // Match60 (reference :251-269; 60 bytes are there), then the element --
        // [literal tag, literals] copy element -- in one store: the literal lanes
        // write their own byte, the lane of the hit the literal tag, the lanes
        // behind it the copy element
        lit = (uint32_t)t - start;
        distance = read_lane(dist, t);
        const uint32_t match_pos = pos0 + (uint32_t)t + 4;
        bool mis = true;
        if (lane < 60)
          mis = src[match_pos + lane] != src[match_pos - distance + lane];
        const uint32_t xt = (uint32_t)__builtin_ctzll(wave_ballot(mis));
        const uint32_t copy_tag = (((xt + 3u) << 2) | 0x2u) | (distance << 8);
        {
          const uint32_t rel = (uint32_t)lane - start;
          const uint32_t k = rel - lit - 1u;
          const bool is_lit = rel < lit;
          const bool is_tag = lit > 0 && rel == lit;
          const bool is_copy = k < 3u;
          uint32_t val = is_lit ? data32 : copy_tag >> (8u * k);
          val = is_tag ? (lit - 1u) << 2 : val;
          uint32_t off = rel + (is_lit ? 1u : (lit > 0 ? 0u : ~0u));
          off = is_tag ? 0u : off;
          if (is_lit || is_tag || is_copy)
            dst[c + off] = (uint8_t)val;
        }
        c += lit + (lit > 0 ? 1u : 0u) + 3u;
        asm("s_bitset1_b64 %0, %1" : "+s"(hit_lanes) : "s"(t));
        asm("s_bitset1_b64 %0, %1" : "+s"(start_lanes) : "s"(start));
        start = (uint32_t)t + 4u + xt;
      }
      // lanes start..hit of every element: their hash-map update stays
      const uint64_t stay = (hit_lanes << 1) - start_lanes;
      // the hash map as the elements leave it: every other lane takes its update
      // back, then the lanes that stay post theirs again (a lane inside a match
      // may have shared its slot with one of them)
      const bool stays = ((stay >> lane) & 1ull) != 0;
      if (lane < kSpan && !stays)
        hash_map[hash] = (uint16_t)h_old;
      lds_lane_exchange_fence();
      if (stays)
        hash_map[hash] = (uint16_t)my;
      lds_lane_exchange_fence();
      if (start != 0) {
        const uint32_t moved = start;
        pos = pos0 + moved;
        const int from4 = (int)((((uint32_t)lane + moved) & 63u) * 4u);
        const uint32_t moved_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(from4, (int)next);
        const uint32_t moved_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(from4, (int)next_hi);
        next = load_u32_any(src + min(pos + (uint32_t)lane, last_word));
        next_hi = load_u32_any(src + min(pos + 64u + (uint32_t)lane, last_word));
        wnd = (uint32_t)lane + moved < 64u ? moved_lo : moved_hi;
        if (__builtin_expect(moved > kStraightReach, 0)) {
          // (a real branch: as a select it would make every trip wait for the load)
          asm volatile("" ::: "memory");
          wnd = next;
        }
        straight = true;
      }
    }
    if (straight)
      continue;
    // ---- FindFourByteMatch (reference :190-246)
    const uint32_t maxpos = pos0 + kMaxLiteral - (kWave - 1);
    uint32_t p = pos0;
    uint32_t literal_cnt;
    do {
      const uint32_t my = p + (uint32_t)lane;
      const bool valid4 = my + 4 <= len;
      const uint32_t raw = p == pos0 ? next : load_u32_any(src + min(my, last_word));
      const uint32_t data32 = valid4 ? raw : 0;
      const uint32_t hash = valid4 ? snap_hash(data32) : 0;

      // table candidate first (its verify load is the long latency)
      const uint32_t h_old = hash_map[hash];
      uint32_t toff = (p & ~0xffffu) | h_old;
      if (toff >= p)
        toff = (toff >= 0x10000u) ? toff - 0x10000u : p;
      const bool tprobe = valid4 && toff < p && toff + kMaxCopyDistance >= my;
      // Candidates are fetched for the first kFirstLanes lanes only: that is
      // where the match of a window of data that compresses is, and a 64-lane
      // gather is 64 memory transactions.  The other lanes re-read their own
      // word (one line); their candidates follow below if no early lane hits.
      const bool tprobe1 = tprobe && lane < kFirstLanes;
      const uint32_t tword = load_u32_any(src + (tprobe1 ? toff : min(my, last_word)));

      // HashMatchAny (reference :157-172): all 64 lanes take part, lanes past
      // the end with hash 0.  Post lane ids (highest lane survives), read the
      // slot back, restore it.
      lds_lane_exchange_fence();
      hash_map[hash] = (uint16_t)lane;
      lds_lane_exchange_fence();
      const uint32_t top = hash_map[hash];
      lds_lane_exchange_fence();
      hash_map[hash] = (uint16_t)h_old;
      lds_lane_exchange_fence();
      uint64_t local_match = lane_bit;
      uint64_t pending = wave_ballot(top != (uint32_t)lane); // lanes in slots with >= 2 lanes
      while (pending) {
        const int u = __builtin_ctzll(pending);
        const uint32_t hu = read_lane(hash, u);
        const uint64_t gm = wave_ballot(hash == hu);
        if (hash == hu)
          local_match = gm;
        pending &= ~gm;
      }
      if (!valid4)
        local_match = 0;

      // nearest lower lane with my hash
      const uint64_t below = local_match & below_me;
      const uint32_t lml = below ? (uint32_t)(63 - __builtin_clzll(below)) : 0xFFFFFFFFu;
      const uint32_t lmd = (uint32_t)__builtin_amdgcn_ds_bpermute(
          (int)(min(lml, (uint32_t)lane) * 4u), (int)data32);
      const bool local_hit = valid4 && lml < (uint32_t)lane && lmd == data32;
      bool table_hit = !local_hit && tprobe1 && tword == data32;
      const uint32_t offset = local_hit ? p + lml : toff;

      uint64_t match_mask = wave_ballot(local_hit || table_hit);
      if ((match_mask & ((1ull << kFirstLanes) - 1ull)) == 0) {
        // no hit among the first lanes: the first hit of all lanes decides
        const bool tprobe2 = tprobe && lane >= kFirstLanes;
        if (wave_ballot(tprobe2) != 0) {
          const uint32_t tword2 = load_u32_any(src + (tprobe2 ? toff : min(my, last_word)));
          table_hit = !local_hit && tprobe2 && tword2 == data32;
          match_mask |= wave_ballot(table_hit);
        }
      }
      if (match_mask) {
        literal_cnt = (uint32_t)__builtin_ctzll(match_mask);
        distance = read_lane(my - offset, (int)literal_cnt);
        copy_len = 4;
      } else {
        literal_cnt = kWave;
      }
      // hash-map update (reference :240-242); literal_cnt == 64 -> mask 1
      const uint64_t upd = (2ull << (literal_cnt & 63u)) - 1ull;
      const uint64_t m = local_match & upd;
      if ((uint32_t)lane <= literal_cnt && m != 0 && lane == 63 - __builtin_clzll(m))
        hash_map[hash] = (uint16_t)my;
      p += literal_cnt;
    } while (literal_cnt == kWave && p < maxpos);
    lit = min(p, len) - pos0;

    // ---- Match60 (reference :251-269)
    if (copy_len) {
      const uint32_t match_pos = pos0 + lit + 4;
      const uint32_t n = min(len - match_pos, 60u);
      bool mis = true;
      if ((uint32_t)lane < n)
        mis = src[match_pos + lane] != src[match_pos - distance + lane];
      copy_len += (uint32_t)__builtin_ctzll(wave_ballot(mis));
    }

    // ---- StoreLiterals / StoreCopy (reference :73-151).  The output buffer
    // holds 32 + n + n/6 bytes by contract, which this encoder cannot exceed
    // (DESIGN.md), so the reference's per-byte bounds checks are not needed.
    // The copy element as up to three bytes (wave-uniform)
    uint32_t copy_tag = 0, copy_bytes = 0;
    if (copy_len > 0) {
      if (copy_len < 12 && distance < 2048) {
        copy_tag = (((distance & 0x700u) >> 3) | ((copy_len - 4) << 2) | 0x01u) | ((distance & 0xFFu) << 8);
        copy_bytes = 2;
      } else {
        copy_tag = (((copy_len - 1) << 2) | 0x2u) | (distance << 8);
        copy_bytes = 3;
      }
    }
    if (lit <= 60) {
      // Common case: literal tag (one byte), literals and copy element are at
      // most 64 bytes -- ONE byte-per-lane store instead of three lane-0
      // stores and a general copy.
      const uint32_t hdr = lit > 0 ? 1u : 0u;
      const uint32_t cb = hdr + lit; // where the copy element starts
      const uint32_t i = (uint32_t)lane;
      uint32_t b = (lit - 1u) << 2; // lane 0 of a literal element: its tag
      if (i >= hdr && i < cb)
        b = src[pos0 + i - hdr];
      if (i >= cb)
        b = copy_tag >> (8u * (i - cb));
      if (i < cb + copy_bytes)
        dst[c + i] = (uint8_t)b;
      c += cb + copy_bytes;
    } else {
      const uint32_t lm1 = lit - 1; // 60 <= lm1 <= 255 -> one length byte
      if (lane == 0) {
        dst[c] = (uint8_t)(60 << 2);
        dst[c + 1] = (uint8_t)lm1;
      }
      c += 2;
      wave_copy(dst + c, src + pos0, lit, lane);
      c += lit;
      if ((uint32_t)lane < copy_bytes)
        dst[c + lane] = (uint8_t)(copy_tag >> (8u * (uint32_t)lane));
      c += copy_bytes;
    }
    pos = pos0 + lit + copy_len;
    next = load_u32_any(src + min(pos + (uint32_t)lane, last_word));
    next_hi = load_u32_any(src + min(pos + 64u + (uint32_t)lane, last_word));
    wnd = next;
  }
  return c;
}

__global__ __launch_bounds__(kWave) void snappy_compress_kernel(
    const uint8_t* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    uint8_t* const* __restrict__ out_ptrs,
    size_t* __restrict__ out_bytes,
    const size_t* __restrict__ out_available, uint32_t* __restrict__ statuses)
{
  __shared__ __attribute__((aligned(16))) uint16_t hash_map[kHashEntries];

  const int lane = (int)threadIdx.x;
  const size_t chunk = blockIdx.x;
  cgptr __restrict__ src = to_global(uniform_ptr(in_ptrs[chunk]));
  const uint32_t len = uniform((uint32_t)in_bytes[chunk]);
  gptr __restrict__ dst = to_global(uniform_ptr(out_ptrs[chunk]));
  // (only through hipcomp::gpu_snap, the reference's internal entry point -- the batched API has neither:
  // reference compression.hiph:307-311 and :383 let the encoder run and report an output that did not
  // fit; here a buffer smaller than the worst case for this input is not written to at all)
  if (out_available != nullptr) {
    const size_t room = uniform((uint64_t)out_available[chunk]);
    if (room != 0 && room < (size_t)32 + len + len / 6) { // reference get_max_compressed_length
      if (lane == 0) {
        out_bytes[chunk] = 0;
        if (statuses)
          statuses[chunk] = 1;
      }
      return;
    }
  }
  if (statuses != nullptr && lane == 0)
    statuses[chunk] = 0;
  const uint32_t c = snappy_encode_chunk(src, len, dst, hash_map, lane);
  if (lane == 0)
    out_bytes[chunk] = c;
}

// The same for the high-level manager (placement.hpp): a grid as large as the device holds workgroups, each
// takes chunks off a ticket counter, compresses into its slot and moves the chunk to its place in the
// container.  The copy reads what this wave has just written (its own stores, in order: the lines are in
// the L2), so the compressed bytes cross the HBM once.
__global__ __launch_bounds__(kWave) void snappy_compress_placed_kernel(
    const uint8_t* const* __restrict__ in_ptrs,
    const size_t* __restrict__ in_bytes,
    size_t* __restrict__ out_bytes, const uint32_t batch, uint32_t* __restrict__ ticket, const Placement place)
{
  __shared__ __attribute__((aligned(16))) uint16_t hash_map[kHashEntries];
  const int lane = (int)threadIdx.x;
  gptr __restrict__ slot = to_global(place.slots + (size_t)blockIdx.x * place.slot_bytes);
  for (uint32_t chunk = next_chunk(ticket, lane); chunk < batch;) {
    const uint32_t asked = ask_next_chunk(ticket, lane);
    cgptr __restrict__ src = to_global(uniform_ptr(in_ptrs[chunk]));
    const uint32_t len = uniform((uint32_t)in_bytes[chunk]);
    const uint32_t c = snappy_encode_chunk(src, len, slot, hash_map, lane);
    if (lane == 0)
      out_bytes[chunk] = c;
    place_chunk(place, chunk, slot, c, lane);
    // (every load of the copy has come back when it returns -- its stores needed them -- so the next chunk may
    // write the slot)
    chunk = chunk_asked_for(asked);
  }
}

// Varint preamble (reference get_uncompressed_sizes_kernel,
// SnappyBatchKernels.hip:84-134, and decode_uncompressed_size,
// decompression.hiph:70-104).  Returns false on the ">= 2^31" error.
__device__ __forceinline__ bool read_preamble(
    cgptr comp, uint32_t end, uint32_t& cur, uint32_t& n)
{
  n = comp[cur++];
  if (n > 0x7f) {
    uint32_t c = (cur < end) ? comp[cur++] : 0;
    n = (n & 0x7f) | (c << 7);
    if (n >= (0x80u << 7)) {
      c = (cur < end) ? comp[cur++] : 0;
      n = (n & 0x3fff) | (c << 14);
      if (n >= (0x80u << 14)) {
        c = (cur < end) ? comp[cur++] : 0;
        n = (n & 0x1fffff) | (c << 21);
        if (n >= (0x80u << 21)) {
          c = (cur < end) ? comp[cur++] : 0;
          if (c < 0x8)
            n = (n & 0xfffffff) | (c << 28);
          else
            return false;
        }
      }
    }
  }
  return true;
}

__global__ __launch_bounds__(256) void snappy_get_sizes_kernel(
    const uint8_t* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes, size_t* __restrict__ out_sizes,
    size_t batch)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch)
    return;
  cgptr comp = to_global(comp_ptrs[i]);
  const uint32_t end = (uint32_t)comp_bytes[i];
  uint32_t n = 0;
  if (end > 0) {
    uint32_t cur = 0;
    if (!read_preamble(comp, end, cur, n))
      n = 0;
  }
  out_sizes[i] = n;
}

constexpr int kDecompWavesPerBlock = 4;

// How far past an element's tag the decoder looks in the register window
// before it asks for the window again: in the several-elements step elements
// start within 64 bytes of the first one's tag and reach up to 61 bytes further.
constexpr uint32_t kSnappyBatchReach = 128;
// Least bytes of stream left for the window to be used (tag + 3).
constexpr uint32_t kSnappyWindowMin = 4;

__global__ __launch_bounds__(kWave * kDecompWavesPerBlock) void snappy_decompress_kernel(
    const uint8_t* const* __restrict__ comp_ptrs,
    const size_t* __restrict__ comp_bytes,
    const size_t* __restrict__ out_caps, const size_t batch,
    uint8_t* const* __restrict__ out_ptrs, size_t* __restrict__ actual_bytes,
    hipcompStatus_t* __restrict__ statuses)
{
  // What a tag byte says, looked up instead of computed (the parse below is
  // bound by its vector instruction count): output bytes | stream bytes << 8 |
  // literal << 16 | 1-byte-offset copy << 17 | not for the fast path << 31
  // (4-byte offset, literal with a length field).
  __shared__ uint32_t tag_lut[256];
  __shared__ uint16_t start_of_output[kDecompWavesPerBlock][kWave]; // (the several-elements step below)
  static_assert(kWave * kDecompWavesPerBlock == 256, "one tag per thread");
  {
    const uint32_t b = threadIdx.x, kind = b & 3u, n6 = b >> 2;
    const bool lit = kind == 0u, c1 = kind == 1u;
    const uint32_t blen = c1 ? (n6 & 7u) + 4u : n6 + 1u;
    const uint32_t need = lit ? 1u + blen : kind + 1u;
    const bool slow = kind == 3u || (lit && n6 >= 60u);
    tag_lut[b] = blen | (need << 8) | (lit ? 1u << 16 : 0u) | (c1 ? 1u << 17 : 0u) | (slow ? 1u << 31 : 0u);
  }
  __syncthreads();
  const int lane = lane_id();
  // everything that steers the parse is wave-uniform: say so (see uniform())
  const size_t chunk = (size_t)blockIdx.x * kDecompWavesPerBlock + uniform((uint32_t)(threadIdx.x >> 6));
  if (chunk >= batch)
    return;
  cgptr comp = to_global(uniform_ptr(comp_ptrs[chunk]));
  const uint32_t end = uniform((uint32_t)comp_bytes[chunk]);
  gptr out = to_global(uniform_ptr(out_ptrs[chunk]));

  uint32_t usize = 0, bytes_left = 0;
  bool error = false;
  if (end == 0) {
    error = true;
  } else {
    uint32_t cur = 0;
    if (!read_preamble(comp, end, cur, usize))
      error = true;
    usize = uniform(usize);
    cur = uniform(cur);
    size_t cap = out_caps ? uniform((uint64_t)out_caps[chunk]) : 0; // (null: hipcomp::gpu_unsnap, "all have room")
    if (cap == 0)
      cap = usize; // reference decompression.hiph:148-149
    if ((cur >= end && usize != 0) || usize > cap)
      error = true;
    if (error) {
      usize = 0; // the reference reports size - bytes_left = 0 here
    } else {
      // Issue slots bound this kernel (32 waves per CU; a SIMD issues one
      // scalar and one vector instruction per ~4 cycles) and left alone the
      // compiler runs the whole wave-uniform parse on the scalar unit (59
      // scalar against 6 vector instructions per element in round 1).  So the
      // three positions live in vector registers -- same value in all lanes,
      // see in_vector_register -- the stream window is read with ds_bpermute,
      // the element kinds are told apart by selects instead of branches and
      // all that can go wrong is folded into one sign test.
      uint32_t vcur = in_vector_register(cur), vdst = in_vector_register(0u);
      uint32_t vleft = in_vector_register(usize);
      const uint32_t i = (uint32_t)lane;
      StreamWindowV sw;
      uint32_t batch_rest = 0, batch_fails = 0; // (the several-elements step below)
      for (;;) {
        // ---- elements with a one-byte length: literals of up to 60 bytes,
        // copies with 1- and 2-byte offsets, at least 4 bytes of stream left
        for (;;) {
          const uint32_t avail = end - vcur; // (cur <= end always)
          if (wave_ballot((int32_t)((vleft - 1u) | (avail - kSnappyWindowMin)) < 0) != 0)
            break;
          // ---- several elements as one step: lane i looks at the stream byte i
          // bytes on as if an element started there (what it says: the table);
          // the elements that do start are followed from the first one (one
          // v_readlane each) for as long as they are of this path's kinds and
          // their output fits 64 bytes; then every output byte finds its element
          // (the highest output start at or below it; its stream lane through a
          // small table in LDS) and its source: the stream window for literals,
          // out[] for copies -- whose source has to lie in front of the step's
          // output, or the step ends in front of that copy.
          sw.ensure(comp, vcur, end, kSnappyBatchReach, lane);
          // (tried again after twice as many elements every time it took nothing,
          // at once after it did: long literals and copies never come two to a step)
          if (batch_rest != 0) {
            --batch_rest;
          } else {
            batch_rest = min((1u << batch_fails) - 1u, 15u);
            batch_fails = min(batch_fails + 1u, 5u);
            const uint32_t ib = vcur - sw.base;
            const uint32_t w_here = sw.bytes_at(ib + i);
            const uint32_t e_here = tag_lut[w_here & 0xFFu];
            const uint32_t avail_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)avail);
            const uint32_t room = min((uint32_t)__builtin_amdgcn_readfirstlane((int)vleft), (uint32_t)kWave);
            // What a lane says about the element that would start at its stream byte: stream bytes |
            // output bytes << 8 if it is one this step takes -- of this path's kinds, all of it inside
            // the stream -- else a word that ends the walk.
            const uint32_t need_here = (e_here >> 8) & 0xFFu, ob_here = e_here & 0xFFu;
            const uint32_t says = ((e_here >> 31) == 0u && i + need_here <= avail_s) ? (need_here | (ob_here << 8)) : 0x10000u;
            // The walk from element to element is all the scalar unit does for an element (as in the
            // LZ4 decoder, lz4_decode.hiph: the step was bound by its scalar instructions): stream
            // position (low byte) and output bytes (above it) move on with ONE add, the word that ends
            // the walk and an output that no longer fits fail the same compare, the position past lane
            // 63 is one masked test; written out, unrolled once (the compiler's loop: eleven scalar
            // instructions and five branches per element).
            uint64_t stream_starts;
            uint32_t acc;
            {
              const uint32_t limit = uniform((room << 8) | 0xFFu);
              uint32_t nxt, p, tmp;
              asm volatile("s_mov_b64 %[starts], 0\n\t"
                           "s_mov_b32 %[acc], 0\n\t"
                           "v_readlane_b32 %[p], %[says], 0\n"
                           "1:\n\t"
                           "s_add_u32 %[nxt], %[acc], %[p]\n\t"
                           "s_cmp_gt_u32 %[nxt], %[limit]\n\t"
                           "s_cbranch_scc1 2f\n\t"            // the element at acc is not taken: acc stands
                           "s_bitset1_b64 %[starts], %[acc]\n\t" // (bit acc[5:0] = its stream position)
                           "s_and_b32 %[tmp], %[nxt], 0xc0\n\t"
                           "s_cbranch_scc1 3f\n\t"            // the next element lies beyond the 64 lanes
                           "v_readlane_b32 %[p], %[says], %[nxt]\n\t"
                           "s_add_u32 %[acc], %[nxt], %[p]\n\t"
                           "s_cmp_gt_u32 %[acc], %[limit]\n\t"
                           "s_cbranch_scc1 3f\n\t"            // the element at nxt is not taken
                           "s_bitset1_b64 %[starts], %[nxt]\n\t"
                           "s_and_b32 %[tmp], %[acc], 0xc0\n\t"
                           "s_cbranch_scc1 2f\n\t"
                           "v_readlane_b32 %[p], %[says], %[acc]\n\t"
                           "s_branch 1b\n"
                           "3:\n\t"
                           "s_mov_b32 %[acc], %[nxt]\n"
                           "2:"
                           : [starts] "=&s"(stream_starts), [acc] "=&s"(acc), [nxt] "=&s"(nxt), [p] "=&s"(p), [tmp] "=&s"(tmp)
                           : [says] "v"(says), [limit] "s"(limit)
                           : "scc");
            }
            uint32_t at = acc & 0xFFu, total = acc >> 8;
            if (__builtin_popcountll(stream_starts) >= 2) {
              // Every output byte finds its element on the vector side: the lanes where the taken
              // elements start know where their output starts (a prefix sum of the output bytes over
              // those lanes) and post that and their own number at the output lane in question (a
              // 64-entry table in LDS, cleared first: a one-byte literal may start at any lane); a
              // running maximum hands every output lane the latest start at or below it.
              const int wave_in_block = (int)(threadIdx.x >> 6);
              const bool on = ((stream_starts >> i) & 1ull) != 0;
              const uint32_t ob = on ? ob_here : 0u;
              const uint32_t o_start = wave_scan_add_u32(ob) - ob;
              lds_lane_exchange_fence();
              start_of_output[wave_in_block][i] = 0;
              lds_lane_exchange_fence();
              if (on)
                start_of_output[wave_in_block][o_start & 63u] = (uint16_t)(((o_start << 6) | i) + 1u);
              lds_lane_exchange_fence();
              const uint32_t got = start_of_output[wave_in_block][i];
              lds_lane_exchange_fence();
              const uint32_t latest = wave_scan_max_u32(got) - 1u;
              const uint32_t o_mine = (latest >> 6) & 63u; // where my element's output starts
              const uint32_t t_mine = latest & 63u;        // its stream lane
              const uint32_t w_mine = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(t_mine * 4u), (int)w_here);
              const uint32_t e_mine = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(t_mine * 4u), (int)e_here);
              const bool lit_mine = (e_mine & (1u << 16)) != 0u;
              const uint32_t o16 = (w_mine >> 8) & 0xFFFFu;
              const uint32_t offset_mine = (e_mine & (1u << 17)) ? ((w_mine & 0xe0u) << 3) | (o16 & 0xFFu) : o16;
              // a copy: 0 < offset <= dst
              const uint32_t bad_mine = lit_mine ? 0u : (offset_mine - 1u) | (vdst + o_mine - offset_mine);
              const uint64_t bad_lanes = wave_ballot(i < total && (int32_t)bad_mine < 0);
              if (bad_lanes != 0) { // the step ends in front of the first such copy
                const int b = __builtin_ctzll(bad_lanes);
                total = read_lane(o_mine, b);
                at = read_lane(t_mine, b);
              }
              if (total != 0) {
                const uint32_t widx = ib + t_mine + 1u + (i - o_mine);
                const uint32_t wword = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(widx & ~3u), (int)sw.words);
                // A byte comes from the stream window (literal), from out[] in front
                // of the step (a copy that reaches back that far), or from a lower
                // lane of this very step -- which may have its byte from a lower lane
                // again (copies of copies, copies that overlap themselves).  Every
                // lane keeps the lane its byte comes from; rounds of "take the
                // source's source" (chains halve each round) bring all of them to a
                // lane of the first two kinds.
                const int32_t from_rel = (int32_t)(i - offset_mine); // (copy lanes) < 0: in front of the step
                const bool outside = i >= total || lit_mine || from_rel < 0;
                uint32_t val = wword >> ((widx & 3u) * 8u);
                if (i < total && !lit_mine && from_rel < 0)
                  val = static_cast<cgptr>(out)[vdst + (uint32_t)from_rel];
                uint32_t from = outside ? (i | 0x80u) : (uint32_t)from_rel; // bit 7: a lane that has its byte
                while (wave_ballot((from & 0x80u) == 0u) != 0) {
                  const uint32_t theirs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((from & 63u) * 4u), (int)from);
                  from = (from & 0x80u) ? from : theirs;
                }
                val = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((from & 63u) * 4u), (int)val);
                if (i < total)
                  out[vdst + i] = (uint8_t)val;
                vcur += at;
                vdst += total;
                vleft -= total;
                batch_rest = 0;
                batch_fails = 0;
                continue;
              }
            }
          }
          const uint32_t t = sw.bytes_at(vcur - sw.base);
          const uint32_t e = tag_lut[t & 0xFFu];
          const uint32_t blen = e & 0xFFu, need = (e >> 8) & 0xFFu;
          const bool lit = (e & (1u << 16)) != 0u;
          // ---- a literal of up to 32 bytes and the copy behind it as one step
          // (what text is made of): lanes below the literal's length carry its
          // bytes, the lanes behind them the copy's, whose source may be those
          // very literal bytes (then they come from the stream window too, not
          // from out[]).  Anything else about the pair leaves it to the code
          // below, one element at a time.
          if (wave_ballot(lit) != 0) {
            const uint32_t t2 = sw.bytes_at(vcur - sw.base + need); // (need <= 61: inside the window)
            const uint32_t e2 = tag_lut[t2 & 0xFFu];
            const uint32_t blen2 = e2 & 0xFFu, need2 = (e2 >> 8) & 0xFFu;
            const uint32_t off16b = (t2 >> 8) & 0xFFFFu;
            const uint32_t offset2 = (e2 & (1u << 17)) ? ((t2 & 0xe0u) << 3) | (off16b & 0xFFu) : off16b;
            const uint32_t n = blen + blen2;
            // both tags for this path, the second a copy; <= 64 bytes that fit what
            // is left of the output and of the stream; 0 < offset <= dst + literal
            const uint32_t bad2 = e | e2 | ((e2 & (1u << 16)) ? ~0u : 0u) | (32u - blen) | ((uint32_t)kWave - n) | (vleft - n)
                                  | (avail - need - need2) | (offset2 - 1u) | (vdst + blen - offset2);
            if (wave_ballot((int32_t)bad2 < 0) == 0) {
              uint32_t k = i - blen; // copy byte index (lanes behind the literal)
              if (wave_ballot(offset2 < blen2) != 0)
                k = small_mod(k & 63u, offset2);
              const int32_t s = (int32_t)(blen + k - offset2); // its source relative to dst; >= 0: a byte of the literal
              const bool from_window = i < blen || s >= 0;
              const uint32_t widx = vcur - sw.base + 1u + (i < blen ? i : (uint32_t)max(s, 0));
              const uint32_t wword = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(widx & ~3u), (int)sw.words);
              if (i < n) {
                const uint32_t ob = static_cast<cgptr>(out)[vdst + (from_window ? i : (uint32_t)s)];
                out[vdst + i] = (uint8_t)(from_window ? wword >> ((widx & 3u) * 8u) : ob);
              }
              vcur += need + need2;
              vdst += n;
              vleft -= n;
              continue;
            }
          }
          const uint32_t off16 = (t >> 8) & 0xFFFFu;
          const uint32_t offset = (e & (1u << 17)) ? ((t & 0xe0u) << 3) | (off16 & 0xFFu) : off16;
          // tag for this path; fits what is left of the output and of the
          // stream; copy: 0 < offset <= dst
          const uint32_t bad = e | (vleft - blen) | (avail - need) | (lit ? 0u : (offset - 1u) | (vdst - offset));
          if (wave_ballot((int32_t)bad < 0) != 0)
            break;
          // one byte per lane.  Copies: lane % offset -- for offset >= blen (no
          // overlap) that is the lane.  Earlier stores of this wave to out[]
          // are ordered before these loads (one wave, in-order vector memory,
          // one L1).
          uint32_t k = i;
          if (wave_ballot(!lit && offset < blen) != 0)
            k = small_mod(i, offset);
          // literal bytes come from the register window (it reaches 64 bytes
          // past the tag), copy bytes from out[]; a literal's lanes load their
          // own destination byte instead (inside the buffer, value unused)
          const uint32_t widx = vcur - sw.base + 1u + i;
          const uint32_t wword = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(widx & ~3u), (int)sw.words);
          if (i < blen) {
            const uint32_t ob = static_cast<cgptr>(out)[lit ? vdst + i : vdst - offset + k];
            out[vdst + i] = (uint8_t)(lit ? wword >> ((widx & 3u) * 8u) : ob);
          }
          vcur += need;
          vdst += blen;
          vleft -= blen;
        }
        // ---- anything else, one element: scalar again
        uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)vcur);
        uint32_t dst_pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)vdst);
        bytes_left = (uint32_t)__builtin_amdgcn_readfirstlane((int)vleft);
        if (bytes_left == 0 || cur >= end)
          break;
        // tag byte and the three bytes behind it; bytes past the end read as
        // 0, every use is guarded by a length check
        uint32_t t = uniform((uint32_t)comp[cur]);
        for (uint32_t j = 1; j < 4 && cur + j < end; ++j)
          t |= uniform((uint32_t)comp[cur + j]) << (8 * j);
        const uint32_t b0 = t & 0xFFu;
        uint32_t blen, offset;
        if (b0 & 3u) {
          if (!(b0 & 2u)) { // xxxxxx01.oooooooo
            if (end - cur < 2)
              break;
            offset = ((b0 & 0xe0u) << 3) | ((t >> 8) & 0xFFu);
            blen = ((b0 >> 2) & 7u) + 4u;
            cur += 2;
          } else if (b0 & 1u) { // 4-byte offset
            if (end - cur < 5)
              break;
            offset = uniform(load_u32_any(comp + cur + 1));
            blen = (b0 >> 2) + 1u;
            cur += 5;
          } else { // 2-byte offset
            if (end - cur < 3)
              break;
            offset = (t >> 8) & 0xFFFFu;
            blen = (b0 >> 2) + 1u;
            cur += 3;
          }
          if (offset == 0 || offset > dst_pos || bytes_left < blen)
            break;
          if ((uint32_t)lane < blen) {
            const uint32_t k = small_mod((uint32_t)lane, offset);
            out[dst_pos + lane] = out[dst_pos - offset + k];
          }
        } else {
          blen = b0 >> 2;
          cur += 1;
          if (blen >= 60) {
            const uint32_t nb = blen - 59;
            if (end - cur < nb)
              break;
            blen = 0;
            for (uint32_t j = 0; j < nb; ++j)
              blen |= uniform((uint32_t)comp[cur + j]) << (8 * j);
            cur += nb;
          }
          blen += 1;
          if (blen == 0 || bytes_left < blen || end - cur < blen)
            break;
          if (blen <= kWave) {
            if ((uint32_t)lane < blen)
              out[dst_pos + lane] = comp[cur + lane];
          } else {
            wave_copy(out + dst_pos, comp + cur, blen, lane);
          }
          cur += blen;
        }
        vcur = cur;
        vdst = dst_pos + blen;
        vleft = bytes_left - blen;
      }
      bytes_left = (uint32_t)__builtin_amdgcn_readfirstlane((int)vleft);
      if (bytes_left != 0)
        error = true;
    }
  }
  if (lane == 0) {
    if (actual_bytes)
      actual_bytes[chunk] = usize - bytes_left; // reference decompression.hiph:197-198
    if (statuses)
      statuses[chunk] = error ? hipcompErrorCannotDecompress : hipcompSuccess;
  }
}

} // namespace

void snappy_launch_compress(
    const uint8_t* const* in_ptrs, const size_t* in_bytes,
    uint8_t* const* out_ptrs, size_t* out_bytes, size_t batch,
    hipStream_t stream, const size_t* out_available, uint32_t* statuses)
{
  snappy_compress_kernel<<<dim3((unsigned)batch), dim3(kWave), 0, stream>>>(
      in_ptrs, in_bytes, out_ptrs, out_bytes, out_available, statuses);
}

namespace {
unsigned placed_grid(size_t batch)
{
  static std::atomic<unsigned> known{0}; // (the same on every device of the process)
  unsigned resident = known.load(std::memory_order_relaxed);
  if (resident == 0) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, snappy_compress_placed_kernel, kWave, 0) != hipSuccess
        || per_cu <= 0) {
      (void)hipGetLastError();
      return 0;
    }
    resident = (unsigned)per_cu * (unsigned)num_cus_of_current_device();
    known.store(resident, std::memory_order_relaxed);
  }
  return batch < resident ? (unsigned)batch : resident;
}
} // namespace

size_t snappy_placement_slots()
{
  return placed_grid(~size_t(0));
}

hipError_t snappy_launch_compress_placed(
    const uint8_t* const* in_ptrs, const size_t* in_bytes, size_t* out_bytes, size_t batch,
    uint32_t* ticket, const Placement& place, hipStream_t stream)
{
  const unsigned grid = placed_grid(batch);
  if (grid == 0 || batch >= 0xFFFFFFFFull)
    return hipErrorInvalidValue;
  snappy_compress_placed_kernel<<<dim3(grid), dim3(kWave), 0, stream>>>(in_ptrs, in_bytes, out_bytes, (uint32_t)batch,
                                                                       ticket, place);
  return hipGetLastError();
}

void snappy_launch_decompress(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    const size_t* out_caps, size_t batch, uint8_t* const* out_ptrs,
    size_t* actual_bytes, hipcompStatus_t* statuses, hipStream_t stream)
{
  const unsigned grid = (unsigned)((batch + kDecompWavesPerBlock - 1) / kDecompWavesPerBlock);
  snappy_decompress_kernel<<<dim3(grid), dim3(kWave * kDecompWavesPerBlock), 0, stream>>>(
      comp_ptrs, comp_bytes, out_caps, batch, out_ptrs, actual_bytes, statuses);
}

void snappy_launch_get_sizes(
    const uint8_t* const* comp_ptrs, const size_t* comp_bytes,
    size_t* out_sizes, size_t batch, hipStream_t stream)
{
  const unsigned grid = (unsigned)((batch + 255) / 256);
  snappy_get_sizes_kernel<<<dim3(grid), dim3(256), 0, stream>>>(
      comp_ptrs, comp_bytes, out_sizes, batch);
}

} // namespace hcamd
