// Probe: what the memory side of ONE "trip" of the far-type encoders costs a CU on gfx950, with no
// computation beside it -- the ceiling of the text rows (lz4/text, snappy/text):
//   probe   `span` lanes read 2 bytes at a random slot of the wave's own 32 KiB table
//   fetch   `span` lanes read 16 unaligned bytes at a random place of the wave's own 64 KiB chunk,
//           the place taken from what the probe returned (a dependent round trip, as in the encoder)
//   insert  6 lanes write 2 bytes to their slot; 24 lanes write one output byte each (contiguous)
// per wave: table 32 KiB + chunk 64 KiB + output, all its own, so the footprint grows with the waves.
// Build: hipcc -O3 --offload-arch=gfx950 trip_rate.hip -o bin/trip_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((address_space(1))) uint8_t* gp;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: probes (else the table is "in LDS": no global probe), bit 1: fetch, bit 2: inserts + output bytes
// POL: 0 plain, 1 probes sc1 (L1 bypass), 2 probes nt, 3 output stores nt, 4 probes sc1 + output nt
template <int MODE, int POL>
__global__ __launch_bounds__(256) void k_trip(uint8_t* buf, unsigned long long* out, int trips, uint32_t span, uint32_t waves_total)
{
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= waves_total)
    return;
  gp table = (gp)buf + (size_t)wave * (160u << 10);
  gp chunk = table + (32u << 10);
  gp outp = chunk + (64u << 10);
  uint32_t acc = lane * 2654435761u;
  uint32_t c = 0;
  for (int t = 0; t < trips; ++t) {
    uint32_t hh = (acc ^ (uint32_t)t * 0x9E3779B9u) * 2654435761u;
    hh ^= hh >> 15;
    hh *= 0x2c1b3c6du;
    hh ^= hh >> 12;
    const uint32_t slot = hh & 16383u;
    uint32_t h_old = hh >> 16;
    if ((MODE & 1) && lane < span) {
      if (POL == 1 || POL == 4)
        asm volatile("global_load_ushort %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(h_old) : "v"(table + 2 * slot) : "memory");
      else if (POL == 2)
        asm volatile("global_load_ushort %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(h_old) : "v"(table + 2 * slot) : "memory");
      else
        h_old = *(const __attribute__((address_space(1))) uint16_t*)(table + 2 * slot);
    }
    uint32_t x = h_old;
    if ((MODE & 2) && lane < span) {
      // the candidate: a place of the chunk made from what the slot held (dependent on the probe)
      const uint32_t at = ((h_old * 40503u) ^ hh) & 65519u;
      typedef u32x4 __attribute__((aligned(1))) u128u;
      const u32x4 cw = *(const __attribute__((address_space(1))) u128u*)(chunk + at);
      x = cw.x ^ cw.y ^ cw.z ^ cw.w;
    }
    acc += x;
    if (MODE & 4) {
      if (lane < 6)
        *(__attribute__((address_space(1))) uint16_t*)(table + 2 * slot) = (uint16_t)acc;
      if (lane < 24) {
        if (POL == 3 || POL == 4)
          __builtin_nontemporal_store((uint8_t)acc, (__attribute__((address_space(1))) uint8_t*)(outp + ((c + lane) & 32767u)));
        else
          outp[(c + lane) & 32767u] = (uint8_t)acc;
      }
      c += 24;
    }
  }
  if (acc == 0xdeadbeef)
    out[0] = acc;
}

template <int MODE, int POL> float run(uint8_t* buf, unsigned long long* d, int waves_per_cu, uint32_t span, int trips)
{
  const uint32_t waves = 256u * waves_per_cu;
  const int blocks = (int)((waves + 3) / 4);
  k_trip<MODE, POL><<<blocks, 256>>>(buf, d, 50, span, waves);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  k_trip<MODE, POL><<<blocks, 256>>>(buf, d, trips, span, waves);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main()
{
  setvbuf(stdout, NULL, _IONBF, 0);
  const size_t bytes = (size_t)256 * 32 * (160u << 10) + 4096;
  uint8_t* buf;
  if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  unsigned long long* d;
  hipMalloc(&d, 64);
  const int trips = 2000;
  printf("# per CU: cycles between two trips (2.4 GHz), one trip = `span` lanes; text moves ~40 bytes of input per trip\n");
  printf("%-34s %5s %5s %10s %12s %14s\n", "what", "waves", "span", "ms", "cyc/trip/CU", "GB/s at 40 B");
#define ROW(name, MODE, POL, W, SPAN)                                                                      \
  {                                                                                                        \
    const float ms = run<MODE, POL>(buf, d, W, SPAN, trips);                                               \
    const double per_cu = (double)ms * 1e-3 / ((double)trips * W) * 2.4e9;                                 \
    printf("%-34s %5d %5d %10.3f %12.1f %14.1f\n", name, W, SPAN, ms, per_cu, 256.0 * 40.0 / (per_cu / 2.4e9) / 1e9); \
  }
  for (int w : {8, 14, 20, 32}) {
    for (uint32_t span : {40u, 64u}) {
      ROW("probe only", 1, 0, w, span)
      ROW("fetch only (table in LDS)", 2, 0, w, span)
      ROW("probe -> fetch", 3, 0, w, span)
      ROW("probe -> fetch + stores", 7, 0, w, span)
      ROW("fetch + stores (table in LDS)", 6, 0, w, span)
    }
    ROW("probe -> fetch + stores, sc1 probes", 7, 1, w, 40)
    ROW("probe -> fetch + stores, nt probes", 7, 2, w, 40)
    ROW("probe -> fetch + stores, nt output", 7, 3, w, 40)
    ROW("probe -> fetch + stores, sc1 + nt out", 7, 4, w, 40)
    ROW("fetch + stores, nt output", 6, 3, w, 40)
  }
  return 0;
}
