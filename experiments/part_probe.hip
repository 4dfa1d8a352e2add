// Experiment: partition-at-a-time probing.  1024 partitions, each with its own table region;
// workgroup -> (partition, slice) either XCD-aware (b%8 = XCD) or naive; every row does
// NLOADS dependent 16-byte loads in one random 128-byte line of its partition's region.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t fmix64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }

template <int NLOADS, int RPT, bool XCD, bool UNROLL>
__global__ __launch_bounds__(512) void probe_emul(const ulonglong2 *__restrict__ table, uint32_t lines_per_part, uint32_t rows_per_wg,
                                                  uint32_t W, unsigned long long *__restrict__ sink) {
  const uint32_t b = blockIdx.x;
  uint32_t part, slice;
  if (XCD) { const uint32_t x = b & 7u, i = b >> 3; part = (i / W) * 8u + x; slice = i % W; }
  else { part = b / W; slice = b % W; }
  const ulonglong2 *tb = table + (size_t)part * lines_per_part * 8;
  unsigned long long acc = 0;
  const uint64_t seed = ((uint64_t)part << 32) ^ (slice * 7919u);
  for (uint32_t t = 0; t < rows_per_wg; t += 512 * RPT) {
    if (UNROLL) {
      uint32_t line[RPT]; ulonglong2 r[RPT];
#pragma unroll
      for (int j = 0; j < RPT; j++) { line[j] = (uint32_t)((fmix64(seed + t + j * 512 + threadIdx.x) & 0xFFFFFFFFull) * lines_per_part >> 32); r[j] = tb[line[j] * 8]; }
#pragma unroll
      for (int l = 1; l < NLOADS; l++) {
#pragma unroll
        for (int j = 0; j < RPT; j++) { uint32_t pos = (l + (uint32_t)(r[j].x & 0)) & 7; r[j] = tb[line[j] * 8 + pos]; }
      }
#pragma unroll
      for (int j = 0; j < RPT; j++) acc += r[j].x;
    } else {
#pragma unroll 1
      for (int j = 0; j < RPT; j++) {
        const uint32_t line = (uint32_t)((fmix64(seed + t + j * 512 + threadIdx.x) & 0xFFFFFFFFull) * lines_per_part >> 32);
        ulonglong2 r = tb[line * 8];
#pragma unroll
        for (int l = 1; l < NLOADS; l++) { uint32_t pos = (l + (uint32_t)(r.x & 0)) & 7; r = tb[line * 8 + pos]; }
        acc += r.x;
      }
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

template <int NLOADS, int RPT, bool XCD, bool UNROLL>
void run(const char *name, const ulonglong2 *table, double mib, unsigned long long *sink) {
  const uint32_t W = 64, P = 1024;
  const uint32_t lines = (uint32_t)(mib * 1024 * 1024 / 128);
  const uint32_t rows_per_wg = (1u << 29) / (P * W);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; it++) {
    CK(hipEventRecord(e0));
    probe_emul<NLOADS, RPT, XCD, UNROLL><<<P * W, 512>>>(table, lines, rows_per_wg, W, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (it == 1) printf("  %-52s %.3f ms\n", name, t);
  }
}

int main() {
  unsigned long long *sink; CK(hipMalloc(&sink, 8));
  ulonglong2 *table; const size_t bytes = (size_t)1024 * 4 * 1024 * 1024; CK(hipMalloc(&table, bytes)); CK(hipMemset(table, 0x5A, bytes));
  for (double mib : {1.0, 2.0, 3.0, 4.0}) {
    printf("table region per partition %.1f MiB, 2^29 rows, 1024 partitions x 64 workgroups\n", mib);
    run<1, 8, true, true>("xcd-aware, 1 load, 8 rows unrolled", table, mib, sink);
    run<1, 8, false, true>("naive map, 1 load, 8 rows unrolled", table, mib, sink);
    run<2, 8, true, true>("xcd-aware, 2 loads same line, 8 rows unrolled", table, mib, sink);
    run<2, 8, true, false>("xcd-aware, 2 loads same line, one row at a time", table, mib, sink);
    run<4, 8, true, true>("xcd-aware, 4 loads same line, 8 rows unrolled", table, mib, sink);
    run<4, 8, true, false>("xcd-aware, 4 loads same line, one row at a time", table, mib, sink);
  }
  return 0;
}
