// Experiment: what the partition-at-a-time probe kernel waits for.  Emulates radix_probe_kernel<INNER, MODE 0> on
// synthetic data: 1024 partitions of 2^19 16-byte probe rows, a 3 MiB table region per partition, <= 3 dependent
// 16-byte slot reads per row (all rows read one slot, half a second, a tenth a third, as a 3-way cuckoo table does),
// every row emitted to two 8-byte output columns.  Variants differ in how a tile gets its output position:
//   POS 0: pos = input position (no counter at all: the floor)
//   POS 1: one global atomicAdd per tile, waited for between two barriers (what the library ships)
//   POS 2: the same atomic, but the next tile's rows are requested before its result is waited for
//   POS 3: one global atomicAdd per WORKGROUP for all its tiles (upper bound; exact when every row is emitted)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint64_t fmix64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(v, d); if ((int)lane_id() >= d) v += o; }
  return v;
}

constexpr uint32_t kSlotsPerPart = 3u * 1024 * 1024 / 16;

template <int TB, int RPT, int POS>
__global__ __launch_bounds__(TB) void probe_emul(const ull2 *__restrict__ prow, const ulonglong2 *__restrict__ table, uint32_t rows_per_part,
                                                 uint32_t W, unsigned long long *__restrict__ out_count,
                                                 unsigned long long *__restrict__ o0, unsigned long long *__restrict__ o1) {
  constexpr int kWaves = TB / 64, kTile = TB * RPT;
  __shared__ uint32_t s_cnt[RPT * kWaves], s_pre[RPT * kWaves];
  __shared__ unsigned long long s_base;
  const int tid = threadIdx.x, wave = tid >> 6;
  const uint32_t b = blockIdx.x, x = b & 7u, i = b >> 3;
  const uint32_t part = (i / W) * 8u + x, slice = i % W;
  const ulonglong2 *tb = table + (size_t)part * kSlotsPerPart;
  const int64_t pbeg = (int64_t)part * rows_per_part;
  const int64_t lo = pbeg + (int64_t)slice * kTile, hi = pbeg + rows_per_part, step = (int64_t)W * kTile;
  unsigned long long wg_base = 0;
  if (POS == 3) {
    if (tid == 0) {
      unsigned long long mine = 0;
      for (int64_t t = lo; t < hi; t += step) mine += (hi - t) < kTile ? (hi - t) : kTile;
      s_base = atomicAdd(out_count, mine);
    }
    __syncthreads();
    wg_base = s_base;
    __syncthreads();
  }
  ull2 nrow[RPT];
  if (POS == 2) {
#pragma unroll
    for (int j = 0; j < RPT; j++) { const int64_t r = lo + (int64_t)j * TB + tid; nrow[j] = r < hi ? __builtin_nontemporal_load(prow + r) : ull2{0, 0}; }
  }
  for (int64_t tbase = lo; tbase < hi; tbase += step) {
    uint64_t k[RPT], pay[RPT], val[RPT];
    bool look[RPT], emit[RPT];
    ulonglong2 rw[RPT];
    uint32_t s1[RPT], s2[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const int64_t r = tbase + (int64_t)j * TB + tid;
      look[j] = r < hi;
      k[j] = 0, pay[j] = 0, val[j] = 0;
      if (look[j]) {
        const ull2 row = POS == 2 ? nrow[j] : __builtin_nontemporal_load(prow + r);
        k[j] = row.x, pay[j] = row.y;
      }
      emit[j] = look[j];
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      s1[j] = s2[j] = 0;
      if (look[j]) {
        const uint64_t h = fmix64(k[j]), g = fmix64(h + 0x9E3779B97F4A7C15ULL);
        s1[j] = (uint32_t)(((g & 0xFFFFFFFFull) * kSlotsPerPart) >> 32);
        s2[j] = (uint32_t)(((g >> 32) * kSlotsPerPart) >> 32);
        rw[j] = tb[(uint32_t)(((h & 0xFFFFFFFFull) * kSlotsPerPart) >> 32)];
      }
    }
#pragma unroll
    for (int c = 1; c <= 3; c++) {
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (look[j]) {
          // the table holds zeros: "found at stage c" is decided by key bits so that 50 % stop at 1, 40 % at 2, 10 % at 3
          const uint32_t sel = (uint32_t)(k[j] >> 40) % 10u;
          const bool hit = rw[j].x == 0xDEADBEEFull || (c == 1 ? sel < 5 : (c == 2 ? sel < 9 : true));
          if (hit) { val[j] = rw[j].y + k[j]; look[j] = false; }
          else rw[j] = tb[c == 1 ? s1[j] : s2[j]];
        }
      }
    }
    int64_t pos[RPT];
    if (POS == 0 || POS == 3) {
#pragma unroll
      for (int j = 0; j < RPT; j++) pos[j] = POS == 0 ? tbase + (int64_t)j * TB + tid : (int64_t)wg_base + (int64_t)j * TB + tid;
      if (POS == 3) wg_base += (hi - tbase) < kTile ? (hi - tbase) : kTile;
    } else {
      uint32_t rank[RPT];
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        const uint64_t m = __ballot(emit[j]);
        rank[j] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane_id() == 0) s_cnt[j * kWaves + wave] = __popcll(m);
      }
      __syncthreads();
      unsigned long long old = 0;
      if (wave == 0) {
        const int q = lane_id();
        const uint32_t e = q < RPT * kWaves ? s_cnt[q] : 0;
        const uint32_t incl = wave_incl_scan_u32(e);
        if (q < RPT * kWaves) s_pre[q] = incl - e;
        if (lane_id() == 63) old = incl ? atomicAdd(out_count, (unsigned long long)incl) : 0ull;
      }
      if (POS == 2) {
#pragma unroll
        for (int j = 0; j < RPT; j++) { const int64_t r = tbase + step + (int64_t)j * TB + tid; nrow[j] = r < hi ? __builtin_nontemporal_load(prow + r) : ull2{0, 0}; }
      }
      if (wave == 0 && lane_id() == 63) s_base = old;
      __syncthreads();
      const unsigned long long obase = s_base;
#pragma unroll
      for (int j = 0; j < RPT; j++) pos[j] = (int64_t)(obase + s_pre[j * kWaves + wave] + rank[j]);
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) if (emit[j]) __builtin_nontemporal_store(pay[j], o0 + pos[j]);
#pragma unroll
    for (int j = 0; j < RPT; j++) if (emit[j]) __builtin_nontemporal_store(val[j], o1 + pos[j]);
    if (POS == 1 || POS == 2) __syncthreads();
  }
}


// PAIR: 2-slot buckets (32 bytes); a row is handled by a PAIR of adjacent lanes, lane h reads slot h of the bucket, so one
// wave-load serves 32 rows and touches 32 distinct 32-byte segments.  90 % of the rows are decided by their first
// bucket, the rest by the second (2-choice bucketized cuckoo).  TWOLOADS: one lane per row issues both 16-byte loads of
// the bucket back to back instead.
template <int TB, int HPT, bool TWOLOADS>
__global__ __launch_bounds__(TB) void probe_pair(const ull2 *__restrict__ prow, const ulonglong2 *__restrict__ table, uint32_t rows_per_part,
                                                 uint32_t W, unsigned long long *__restrict__ out_count,
                                                 unsigned long long *__restrict__ o0, unsigned long long *__restrict__ o1) {
  constexpr int kRowsPerThread = TWOLOADS ? HPT : HPT / 2;           // HPT half-rows per thread in the pair form
  constexpr int kTile = TB * kRowsPerThread;
  constexpr int kWaves = TB / 64;
  __shared__ uint32_t s_cnt[HPT * kWaves], s_pre[HPT * kWaves];
  __shared__ unsigned long long s_base;
  const int tid = threadIdx.x, wave = tid >> 6;
  const uint32_t b = blockIdx.x, x = b & 7u, i = b >> 3;
  const uint32_t part = (i / W) * 8u + x, slice = i % W;
  const ulonglong2 *tb = table + (size_t)part * kSlotsPerPart;
  const int64_t pbeg = (int64_t)part * rows_per_part;
  const int64_t lo = pbeg + (int64_t)slice * kTile, hi = pbeg + rows_per_part, step = (int64_t)W * kTile;
  const uint32_t half = TWOLOADS ? 0u : (tid & 1u);
  for (int64_t tbase = lo; tbase < hi; tbase += step) {
    uint64_t k[HPT], pay[HPT], val[HPT];
    bool look[HPT], emit[HPT];
    uint32_t b2[HPT];
    ulonglong2 rw[HPT], rw2[HPT];
#pragma unroll
    for (int j = 0; j < HPT; j++) {
      // pair form: wave-load j covers 32 consecutive rows, lanes 2q and 2q+1 share row q
      const int64_t r = TWOLOADS ? tbase + (int64_t)j * TB + tid : tbase + (int64_t)j * (TB / 2) + (tid >> 1);
      look[j] = r < hi && (TWOLOADS || j < HPT);
      k[j] = 0, pay[j] = 0, val[j] = 0, b2[j] = 0;
      if (look[j]) {
        const ull2 row = __builtin_nontemporal_load(prow + r);
        k[j] = row.x, pay[j] = row.y;
        const uint64_t h = fmix64(k[j]), g = fmix64(h + 0x9E3779B97F4A7C15ULL);
        const uint32_t b1 = (uint32_t)(((h & 0xFFFFFFFFull) * (kSlotsPerPart / 2)) >> 32);
        b2[j] = (uint32_t)(((g & 0xFFFFFFFFull) * (kSlotsPerPart / 2)) >> 32);
        rw[j] = tb[b1 * 2 + half];
        if (TWOLOADS) rw2[j] = tb[b1 * 2 + 1];
      }
      emit[j] = look[j] && (TWOLOADS || half == 0);
    }
#pragma unroll
    for (int j = 0; j < HPT; j++) {
      if (!look[j]) continue;
      const uint32_t sel = (uint32_t)(k[j] >> 40) % 10u;
      uint64_t got = rw[j].y + (TWOLOADS ? rw2[j].y : 0);
      bool hit = rw[j].x == 0xDEADBEEFull || sel < 9;   // 90 % decided by the first bucket
      if (!hit) {
        rw[j] = tb[b2[j] * 2 + half];
        if (TWOLOADS) rw2[j] = tb[b2[j] * 2 + 1];
        got = rw[j].y + (TWOLOADS ? rw2[j].y : 0);
      }
      if (!TWOLOADS) got += __shfl_xor(got, 1, 64);       // the pair's other half
      val[j] = got + k[j];
    }
    uint32_t rank[HPT];
#pragma unroll
    for (int j = 0; j < HPT; j++) {
      const uint64_t m = __ballot(emit[j]);
      rank[j] = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (lane_id() == 0) s_cnt[j * kWaves + wave] = __popcll(m);
    }
    __syncthreads();
    if (wave == 0) {
      const int q = lane_id();
      const uint32_t e = q < HPT * kWaves ? s_cnt[q] : 0;
      const uint32_t incl = wave_incl_scan_u32(e);
      if (q < HPT * kWaves) s_pre[q] = incl - e;
      if (lane_id() == 63) s_base = incl ? atomicAdd(out_count, (unsigned long long)incl) : 0ull;
    }
    __syncthreads();
    const unsigned long long obase = s_base;
#pragma unroll
    for (int j = 0; j < HPT; j++) if (emit[j]) __builtin_nontemporal_store(pay[j], o0 + obase + s_pre[j * kWaves + wave] + rank[j]);
#pragma unroll
    for (int j = 0; j < HPT; j++) if (emit[j]) __builtin_nontemporal_store(val[j], o1 + obase + s_pre[j * kWaves + wave] + rank[j]);
    __syncthreads();
  }
}

template <int TB, int HPT, bool TWOLOADS>
void run_pair(const char *name, const ull2 *prow, const ulonglong2 *table, unsigned long long *cnt, unsigned long long *o0, unsigned long long *o1, uint32_t W) {
  const uint32_t P = 1024, rows_per_part = (1u << 29) / P;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 3; it++) {
    CK(hipMemset(cnt, 0, 8));
    CK(hipEventRecord(e0));
    probe_pair<TB, HPT, TWOLOADS><<<P * W, TB>>>(prow, table, rows_per_part, W, cnt, o0, o1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (it > 0 && t < best) best = t;
  }
  unsigned long long c; CK(hipMemcpy(&c, cnt, 8, hipMemcpyDeviceToHost));
  printf("  %-64s W=%-3u %.3f ms (count %llu)\n", name, W, best, c);
  fflush(stdout);
}

template <int TB, int RPT, int POS>
void run(const char *name, const ull2 *prow, const ulonglong2 *table, unsigned long long *cnt, unsigned long long *o0, unsigned long long *o1, uint32_t W) {
  const uint32_t P = 1024, rows_per_part = (1u << 29) / P;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 3; it++) {
    CK(hipMemset(cnt, 0, 8));
    CK(hipEventRecord(e0));
    probe_emul<TB, RPT, POS><<<P * W, TB>>>(prow, table, rows_per_part, W, cnt, o0, o1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (it > 0 && t < best) best = t;
  }
  unsigned long long c; CK(hipMemcpy(&c, cnt, 8, hipMemcpyDeviceToHost));
  printf("  %-64s W=%-3u %.3f ms (count %llu)\n", name, W, best, c);
  fflush(stdout);
}

__global__ void fill_rows(ull2 *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = ull2{fmix64(i + 1), i};
}

int main() {
  unsigned long long *cnt, *o0, *o1; ull2 *prow; ulonglong2 *table;
  const size_t n = (size_t)1 << 29;
  CK(hipMalloc(&cnt, 8)); CK(hipMalloc(&o0, n * 8)); CK(hipMalloc(&o1, n * 8)); CK(hipMalloc(&prow, n * 16));
  CK(hipMalloc(&table, (size_t)1024 * kSlotsPerPart * 16)); CK(hipMemset(table, 0, (size_t)1024 * kSlotsPerPart * 16));
  fill_rows<<<4096, 256>>>(prow, n); CK(hipDeviceSynchronize());
  printf("probe emulation: 2^29 rows, 1024 partitions, 3 MiB table region each, 1.6 slot reads per row, 2 x 8-B columns out\n");
  run_pair<512, 8, false>("2-slot buckets, lane pairs, TB 512 x 8 half-rows (1.1 bucket reads)", prow, table, cnt, o0, o1, 64);
  run_pair<512, 4, true>("2-slot buckets, one lane issues both 16-B loads, TB 512 x 4 rows", prow, table, cnt, o0, o1, 64);
  run_pair<1024, 4, false>("2-slot buckets, lane pairs, TB 1024 x 4 half-rows", prow, table, cnt, o0, o1, 64);
  run_pair<512, 4, false>("2-slot buckets, lane pairs, TB 512 x 4 half-rows", prow, table, cnt, o0, o1, 128);
  run<512, 4, 1>("TB 512 x 4 rows, atomic per tile (shipped form, 1.6 slot reads)", prow, table, cnt, o0, o1, 64);
  return 0;
  run<512, 4, 0>("TB 512 x 4 rows, pos = input position (no counter)", prow, table, cnt, o0, o1, 64);
  run<512, 4, 1>("TB 512 x 4 rows, atomic per tile (shipped form)", prow, table, cnt, o0, o1, 64);
  run<512, 4, 2>("TB 512 x 4 rows, atomic per tile, next rows requested first", prow, table, cnt, o0, o1, 64);
  run<512, 4, 3>("TB 512 x 4 rows, one atomic per workgroup", prow, table, cnt, o0, o1, 64);
  run<256, 4, 0>("TB 256 x 4 rows, no counter", prow, table, cnt, o0, o1, 64);
  run<256, 4, 1>("TB 256 x 4 rows, atomic per tile", prow, table, cnt, o0, o1, 64);
  run<256, 4, 2>("TB 256 x 4 rows, atomic per tile, next rows first", prow, table, cnt, o0, o1, 64);
  run<256, 8, 1>("TB 256 x 8 rows, atomic per tile", prow, table, cnt, o0, o1, 64);
  run<256, 8, 2>("TB 256 x 8 rows, atomic per tile, next rows first", prow, table, cnt, o0, o1, 64);
  run<1024, 4, 1>("TB 1024 x 4 rows, atomic per tile", prow, table, cnt, o0, o1, 64);
  run<1024, 4, 2>("TB 1024 x 4 rows, atomic per tile, next rows first", prow, table, cnt, o0, o1, 64);
  run<1024, 2, 1>("TB 1024 x 2 rows, atomic per tile", prow, table, cnt, o0, o1, 64);
  run<512, 8, 1>("TB 512 x 8 rows, atomic per tile", prow, table, cnt, o0, o1, 64);
  run<512, 8, 2>("TB 512 x 8 rows, atomic per tile, next rows first", prow, table, cnt, o0, o1, 64);
  run<512, 4, 1>("TB 512 x 4 rows, atomic per tile", prow, table, cnt, o0, o1, 32);
  run<512, 4, 2>("TB 512 x 4 rows, atomic per tile, next rows first", prow, table, cnt, o0, o1, 32);
  run<512, 4, 2>("TB 512 x 4 rows, atomic per tile, next rows first", prow, table, cnt, o0, o1, 16);
  run<256, 4, 2>("TB 256 x 4 rows, atomic per tile, next rows first", prow, table, cnt, o0, o1, 128);
  return 0;
}
