// Experiment: cost of random 16-byte lookups served by L2 / Infinity Cache, by access shape.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t fmix64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }

// VARIANT 0: one random 16-B load per row
//         1: two dependent random loads per row
//         2: four independent loads covering one 64-B line
//         3: one random load + dependent load of the next slot (same line mostly)
//         4: two independent random loads per row (cuckoo shape)
//         5: three dependent random loads
template <int VARIANT, int RPT>
__global__ __launch_bounds__(512) void lookup_kernel(const ulonglong2 *__restrict__ table, uint32_t nslots, int64_t rows_per_thread,
                                                     unsigned long long *__restrict__ sink) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  unsigned long long acc = 0;
  for (int64_t it = 0; it < rows_per_thread; it += RPT) {
    uint32_t idx[RPT];
    ulonglong2 r[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      idx[j] = (uint32_t)((fmix64(tid * 1315423911ull + it + j) & 0xFFFFFFFFull) * nslots >> 32);
      if (VARIANT == 2) idx[j] &= ~3u;
      r[j] = table[idx[j]];
    }
    if (VARIANT == 1 || VARIANT == 5) {
#pragma unroll
      for (int j = 0; j < RPT; j++) { idx[j] = (uint32_t)(((r[j].x ^ idx[j]) & 0xFFFFFFFFull) * nslots >> 32); }
#pragma unroll
      for (int j = 0; j < RPT; j++) r[j] = table[idx[j]];
    }
    if (VARIANT == 5) {
#pragma unroll
      for (int j = 0; j < RPT; j++) { idx[j] = (uint32_t)(((r[j].x ^ idx[j]) & 0xFFFFFFFFull) * nslots >> 32); }
#pragma unroll
      for (int j = 0; j < RPT; j++) r[j] = table[idx[j]];
    }
    if (VARIANT == 2) {
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        ulonglong2 b = table[idx[j] + 1], c = table[idx[j] + 2], d = table[idx[j] + 3];
        r[j].x ^= b.x ^ c.x ^ d.x;
      }
    }
    if (VARIANT == 3) {
#pragma unroll
      for (int j = 0; j < RPT; j++) { uint32_t nx = idx[j] + 1 + (uint32_t)(r[j].x & 0); r[j] = table[nx < nslots ? nx : 0]; }
    }
    if (VARIANT == 4) {
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        uint32_t i2 = (uint32_t)((fmix64(tid * 2654435761ull + it + j) & 0xFFFFFFFFull) * nslots >> 32);
        ulonglong2 b = table[i2];
        r[j].x ^= b.x;
      }
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) acc += r[j].x;
  }
  if (acc == 0x1234567) sink[0] = acc;
}

template <int V>
void run(const char *name, const ulonglong2 *table, uint32_t nslots, unsigned long long *sink, int64_t total_rows) {
  const int grid = 256 * 2, tb = 512;
  int64_t rpt = total_rows / ((int64_t)grid * tb);
  rpt = rpt / 8 * 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; it++) {
    CK(hipEventRecord(e0));
    lookup_kernel<V, 8><<<grid, tb>>>(table, nslots, rpt, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (it == 1) printf("  %-40s %.3f ms  %.1f G rows/s\n", name, t, (double)rpt * grid * tb / t / 1e6);
  }
}

int main(int argc, char **argv) {
  const int64_t rows = int64_t(1) << 29;
  unsigned long long *sink; CK(hipMalloc(&sink, 8));
  for (double mib : {0.75, 1.5, 3.0, 6.0, 24.0, 96.0, 1024.0}) {
    uint32_t nslots = (uint32_t)(mib * 1024 * 1024 / 16);
    ulonglong2 *table; CK(hipMalloc(&table, (size_t)nslots * 16));
    CK(hipMemset(table, 0x5A, (size_t)nslots * 16));
    printf("table %.2f MiB (every workgroup reads all of it)\n", mib);
    run<0>("1 random load", table, nslots, sink, rows);
    run<1>("2 dependent random loads", table, nslots, sink, rows);
    run<5>("3 dependent random loads", table, nslots, sink, rows);
    run<4>("2 independent random loads", table, nslots, sink, rows);
    run<3>("1 random + dependent next slot", table, nslots, sink, rows);
    run<2>("4 loads of one 64-B line", table, nslots, sink, rows);
    CK(hipFree(table));
  }
  return 0;
}
