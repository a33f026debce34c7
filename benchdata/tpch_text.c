/*
 * tpch_text.c -- deterministic TPC-H `lineitem`-like text for bench.py
 * (BASELINE.json configs[3]: "hipcompBatchedSnappy on TPC-H lineitem CSV
 * bytes").  MEASUREMENT INPUT ONLY, not product code.
 *
 * There is no dbgen and no network in the image, so the rows are made here
 * from the column grammar of the TPC-H specification (clause 4.2.3, table
 * LINEITEM) in dbgen's .tbl layout: 16 fields, every field followed by the
 * separator character 0x7C, one row per line:
 *   orderkey partkey suppkey linenumber quantity extendedprice discount tax
 *   returnflag linestatus shipdate commitdate receiptdate shipinstruct
 *   shipmode comment
 * The output is cut into independent 4 MiB blocks (block b is seeded with
 * seed + b and starts on a row boundary), so the bytes do not depend on the
 * number of threads that fill them.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_BYTES ((size_t)4 << 20)

static const char* const INSTRUCT[4] = {"DELIVER IN PERSON", "COLLECT COD", "NONE", "TAKE BACK RETURN"};
static const char* const MODES[7] = {"REG AIR", "AIR", "RAIL", "SHIP", "TRUCK", "MAIL", "FOB"};
static const char* const WORDS[] = {
  "furiously", "sly", "carefully", "blithely", "quickly", "fluffily", "slyly", "quietly", "ruthlessly", "thinly",
  "closely", "doggedly", "daringly", "bravely", "stealthily", "permanently", "enticingly", "idly", "busily",
  "regular", "final", "ironic", "even", "bold", "silent", "special", "pending", "unusual", "express", "packages",
  "requests", "accounts", "deposits", "foxes", "ideas", "theodolites", "pinto", "beans", "instructions",
  "dependencies", "excuses", "platelets", "asymptotes", "courts", "dolphins", "multipliers", "sauternes",
  "warthogs", "frets", "dinos", "attainments", "somas", "Tiresias", "patterns", "forges", "braids", "hockey",
  "players", "frays", "warhorses", "dugouts", "notornis", "epitaphs", "pearls", "tithes", "waters", "orbits",
  "gifts", "sheaves", "depths", "sentiments", "decoys", "realms", "pains", "grouches", "escapades", "sleep",
  "wake", "are", "cajole", "haggle", "nag", "use", "boost", "affix", "detect", "integrate", "maintain", "nod",
  "was", "lose", "sublate", "solve", "thrash", "promise", "engage", "hinder", "print", "x-ray", "breach", "eat",
  "grow", "impress", "mold", "poach", "serve", "run", "dazzle", "snooze", "doze", "unwind", "kindle", "play",
  "hang", "believe", "doubt", "about", "above", "according", "to", "across", "after", "against", "along",
  "alongside", "of", "among", "around", "at", "atop", "before", "behind", "beneath", "beside", "besides",
  "between", "beyond", "by", "despite", "during", "except", "for", "from", "in", "place", "inside", "instead",
  "into", "near", "on", "outside", "over", "past", "since", "through", "throughout", "toward", "under", "until",
  "up", "upon", "without", "with", "within"};
#define NWORDS (sizeof(WORDS) / sizeof(WORDS[0]))

typedef struct { uint64_t s; } rng_t;
static inline uint64_t next64(rng_t* r)
{
  uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint32_t below(rng_t* r, uint32_t n) { return (uint32_t)((next64(r) >> 32) * (uint64_t)n >> 32); }

static inline char* put_uint(char* p, uint64_t v)
{
  char t[24];
  int n = 0;
  do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = t[--n];
  return p;
}
static inline char* put_str(char* p, const char* s) { while (*s) *p++ = *s++; return p; }
static inline char* put_2(char* p, unsigned v) { *p++ = (char)('0' + v / 10); *p++ = (char)('0' + v % 10); return p; }

/* days since 1992-01-01 -> "YYYY-MM-DD" */
static char* put_date(char* p, int days)
{
  static const int mdays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  int y = 1992;
  for (;;) {
    const int leap = (y % 4 == 0 && (y % 100 != 0 || y % 400 == 0));
    const int len = 365 + leap;
    if (days < len) {
      int m = 0;
      for (;; ++m) {
        const int ml = mdays[m] + (m == 1 && leap);
        if (days < ml) break;
        days -= ml;
      }
      p = put_uint(p, (uint64_t)y); *p++ = '-'; p = put_2(p, (unsigned)(m + 1)); *p++ = '-';
      return put_2(p, (unsigned)(days + 1));
    }
    days -= len;
    ++y;
  }
}

static void fill_block(uint8_t* out, size_t n, uint64_t seed, size_t block)
{
  rng_t r = {seed + 0x632BE59BD9B4E019ull * (uint64_t)(block + 1)};
  char row[512];
  uint64_t okey = (uint64_t)block * 40000u + 1;
  unsigned line = 0, lines_in_order = 0;
  size_t at = 0;
  while (at < n) {
    if (line >= lines_in_order) { okey += 1 + below(&r, 4) * (below(&r, 8) == 0 ? 8 : 1); line = 0; lines_in_order = 1 + below(&r, 7); }
    ++line;
    const unsigned part = 1 + below(&r, 200000), supp = 1 + below(&r, 10000), qty = 1 + below(&r, 50);
    const uint64_t price = (uint64_t)qty * (90000u + (part / 10) % 20001u + 100u * (part % 1000u)); /* cents */
    const unsigned disc = below(&r, 11), tax = below(&r, 9);
    const int ship = 1 + (int)below(&r, 2526);            /* 1992-01-02 .. 1998-12-01 */
    const int commit = ship + (int)below(&r, 61);
    const int receipt = ship + 1 + (int)below(&r, 30);
    const int open = ship > 1263;                         /* after the "current date" 1995-06-17 */
    char* p = row;
    p = put_uint(p, okey); *p++ = '|';
    p = put_uint(p, part); *p++ = '|';
    p = put_uint(p, supp); *p++ = '|';
    p = put_uint(p, line); *p++ = '|';
    p = put_uint(p, qty); *p++ = '|';
    p = put_uint(p, price / 100); *p++ = '.'; p = put_2(p, (unsigned)(price % 100)); *p++ = '|';
    *p++ = '0'; *p++ = '.'; p = put_2(p, disc); *p++ = '|';
    *p++ = '0'; *p++ = '.'; p = put_2(p, tax); *p++ = '|';
    *p++ = open ? 'N' : (below(&r, 2) ? 'R' : 'A'); *p++ = '|';
    *p++ = open ? 'O' : 'F'; *p++ = '|';
    p = put_date(p, ship); *p++ = '|';
    p = put_date(p, commit); *p++ = '|';
    p = put_date(p, receipt); *p++ = '|';
    p = put_str(p, INSTRUCT[below(&r, 4)]); *p++ = '|';
    p = put_str(p, MODES[below(&r, 7)]); *p++ = '|';
    { /* comment: 10..43 characters of words */
      const unsigned want = 10 + below(&r, 34);
      char* c0 = p;
      while ((unsigned)(p - c0) < want) {
        if (p != c0) *p++ = ' ';
        p = put_str(p, WORDS[below(&r, (uint32_t)NWORDS)]);
      }
      p = c0 + want;
    }
    *p++ = '|';
    *p++ = '\n';
    size_t len = (size_t)(p - row);
    if (len > n - at) len = n - at;
    memcpy(out + at, row, len);
    at += len;
  }
}

typedef struct { uint8_t* out; size_t n; uint64_t seed; size_t b0, b1; } job_t;
static void* worker(void* v)
{
  job_t* j = (job_t*)v;
  for (size_t b = j->b0; b < j->b1; ++b) {
    const size_t lo = b * BLOCK_BYTES, hi = lo + BLOCK_BYTES < j->n ? lo + BLOCK_BYTES : j->n;
    fill_block(j->out + lo, hi - lo, j->seed, b);
  }
  return NULL;
}

/* Fills out[0, n) with rows; `threads` only changes how long it takes. */
void benchdata_tpch_lineitem_text(uint8_t* out, size_t n, uint64_t seed, int threads)
{
  const size_t blocks = (n + BLOCK_BYTES - 1) / BLOCK_BYTES;
  if (threads < 1) threads = 1;
  if ((size_t)threads > blocks) threads = (int)(blocks ? blocks : 1);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * (size_t)threads);
  const size_t per = (blocks + (size_t)threads - 1) / (size_t)threads;
  for (int k = 0; k < threads; ++k) {
    size_t b0 = (size_t)k * per, b1 = b0 + per;
    if (b0 > blocks) b0 = blocks;
    if (b1 > blocks) b1 = blocks;
    job_t j = {out, n, seed, b0, b1};
    jobs[k] = j;
    pthread_create(&th[k], NULL, worker, &jobs[k]);
  }
  for (int k = 0; k < threads; ++k)
    pthread_join(th[k], NULL);
  free(th);
  free(jobs);
}
