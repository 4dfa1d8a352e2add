// Experiment: throughput of random 8-byte atomics by table size, scope, type and locality (HBM / L2 / LDS).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t fmix64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }

// VARIANT 0: u64 add, agent scope          1: u64 add, workgroup scope (rate only; not coherent across XCDs)
//         2: f64 hardware add (unsafe)     3: f64 atomicAdd as compiled by default
//         4: i64 min                       5: four u64 adds to one 64-B line (accumulators of one group side by side)
//         6: four u64 adds to four separate arrays (accumulators column-wise)
template <int VARIANT, int RPT>
__global__ __launch_bounds__(256) void atomic_kernel(unsigned long long *__restrict__ table, uint32_t nslots, int64_t rows_per_thread) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  for (int64_t it = 0; it < rows_per_thread; it += RPT) {
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const uint64_t r = fmix64(tid * 1315423911ull + it + j);
      uint32_t idx = (uint32_t)((r & 0xFFFFFFFFull) * nslots >> 32);
      if (VARIANT == 0) atomicAdd(&table[idx], 1ull);
      if (VARIANT == 1) __hip_atomic_fetch_add(&table[idx], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (VARIANT == 2) unsafeAtomicAdd(reinterpret_cast<double *>(&table[idx]), 1.0);
      if (VARIANT == 3) atomicAdd(reinterpret_cast<double *>(&table[idx]), 1.0);
      if (VARIANT == 4) atomicMin(reinterpret_cast<long long *>(&table[idx]), (long long)(r >> 32));
      if (VARIANT == 5) {
        idx &= ~7u;
        atomicAdd(&table[idx], 1ull); atomicAdd(&table[idx + 1], 1ull); atomicAdd(&table[idx + 2], 1ull); atomicAdd(&table[idx + 3], 1ull);
      }
      if (VARIANT == 6) {
        const uint32_t q = nslots / 4;
        idx = idx % q;
        atomicAdd(&table[idx], 1ull); atomicAdd(&table[q + idx], 1ull); atomicAdd(&table[2 * q + idx], 1ull); atomicAdd(&table[3 * q + idx], 1ull);
      }
    }
  }
}

// LDS-privatised: G groups x R replicas of one f64 + one u64 accumulator per WG; flushed with global atomics.
template <int RPT>
__global__ __launch_bounds__(512) void lds_kernel(unsigned long long *__restrict__ table, uint32_t G, uint32_t R, int64_t rows_per_thread) {
  extern __shared__ unsigned long long lds[];
  double *fsum = reinterpret_cast<double *>(lds);
  unsigned long long *cnt = lds + (size_t)G * R;
  for (uint32_t i = threadIdx.x; i < 2 * G * R; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint32_t rep = threadIdx.x & (R - 1);
  for (int64_t it = 0; it < rows_per_thread; it += RPT) {
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const uint64_t r = fmix64(tid * 1315423911ull + it + j);
      const uint32_t g = (uint32_t)((r & 0xFFFFFFFFull) * G >> 32);
      unsafeAtomicAdd(&fsum[g * R + rep], 1.0);
      atomicAdd(&cnt[g * R + rep], 1ull);
    }
  }
  __syncthreads();
  for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) {
    double s = 0; unsigned long long c = 0;
    for (uint32_t r = 0; r < R; r++) { s += fsum[g * R + r]; c += cnt[g * R + r]; }
    if (c) { unsafeAtomicAdd(reinterpret_cast<double *>(&table[g]), s); atomicAdd(&table[G + g], c); }
  }
}

template <int V>
void run(const char *name, unsigned long long *table, uint32_t nslots, int64_t total_rows) {
  const int grid = 256 * 8, tb = 256;
  int64_t rpt = total_rows / ((int64_t)grid * tb);
  rpt = rpt / 4 * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; it++) {
    CK(hipEventRecord(e0));
    atomic_kernel<V, 4><<<grid, tb>>>(table, nslots, rpt);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    const int per = (V == 5 || V == 6) ? 4 : 1;
    if (it == 1) printf("  %-44s %8.3f ms  %7.2f G atomics/s\n", name, t, (double)rpt * grid * tb * per / t / 1e6);
  }
}

int main(int argc, char **argv) {
  const int64_t rows = int64_t(1) << 27;
  for (int lg : {6, 10, 13, 16, 19, 22, 27}) {
    const uint32_t nslots = 1u << lg;
    unsigned long long *table; CK(hipMalloc(&table, (size_t)nslots * 8));
    CK(hipMemset(table, 0, (size_t)nslots * 8));
    printf("table 2^%d x 8 B = %.3f MiB\n", lg, nslots * 8.0 / 1048576);
    run<0>("u64 add (agent scope)", table, nslots, rows);
    run<1>("u64 add (workgroup scope)", table, nslots, rows);
    run<2>("f64 add hardware (unsafeAtomicAdd)", table, nslots, rows);
    if (lg >= 13) run<3>("f64 add default atomicAdd", table, nslots, rows);
    run<4>("i64 min", table, nslots, rows);
    if (lg >= 10) run<5>("4 x u64 add in one 64-B line", table, nslots, rows);
    if (lg >= 10) run<6>("4 x u64 add in 4 arrays", table, nslots, rows);
    CK(hipFree(table));
  }
  unsigned long long *table; CK(hipMalloc(&table, 1 << 20)); CK(hipMemset(table, 0, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&lds_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("LDS-privatised f64 add + u64 add per row, 2^29 rows\n");
  const uint32_t cfg[][2] = {{1, 1}, {1, 32}, {6, 1}, {6, 8}, {6, 32}, {64, 1}, {64, 16}, {1024, 1}, {1024, 4}, {4096, 1}};
  for (auto &c : cfg) {
    const uint32_t G = c[0], R = c[1];
    const int grid = 256 * 2, tb = 512;
    const int64_t rpt = (int64_t(1) << 29) / ((int64_t)grid * tb);
    for (int it = 0; it < 2; it++) {
      CK(hipEventRecord(e0));
      lds_kernel<4><<<grid, tb, (size_t)G * R * 16>>>(table, G, R, rpt);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (it == 1) printf("  G=%-5u R=%-3u %8.3f ms  %7.2f G rows/s\n", G, R, t, (double)rpt * grid * tb / t / 1e6);
    }
  }
  return 0;
}
