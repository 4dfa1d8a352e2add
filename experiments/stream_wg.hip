// Experiment: 2^29 16-byte rows read by 65536 short-lived workgroups (8192 rows each, 512 threads,
// 8 rows per thread per tile) vs 1024 long-lived ones; with and without a per-tile
// ballot + sync + global atomic + 2x8-byte compacted stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int MODE>  // 0: read only; 1: + sync/atomic per tile; 2: + stores
__global__ __launch_bounds__(512) void stream_kernel(const ulonglong2 *__restrict__ rows, int64_t rows_per_wg, uint64_t *__restrict__ o1,
                                                     uint64_t *__restrict__ o2, unsigned long long *__restrict__ counter,
                                                     unsigned long long *__restrict__ sink) {
  __shared__ unsigned long long s_base;
  const int64_t lo = blockIdx.x * rows_per_wg, hi = lo + rows_per_wg;
  unsigned long long acc = 0;
  for (int64_t t = lo; t < hi; t += 4096) {
    ulonglong2 r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = rows[t + j * 512 + threadIdx.x];
    if (MODE >= 1) {
      __syncthreads();
      if (threadIdx.x == 63) s_base = atomicAdd(counter, 4096ull);
      __syncthreads();
    }
    if (MODE == 2) {
      const unsigned long long base = s_base;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        __builtin_nontemporal_store((uint64_t)r[j].x, o1 + base + j * 512 + threadIdx.x);
        __builtin_nontemporal_store((uint64_t)r[j].y, o2 + base + j * 512 + threadIdx.x);
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) acc += r[j].x ^ r[j].y;
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

int main() {
  const int64_t n = int64_t(1) << 29;
  ulonglong2 *rows; uint64_t *o1, *o2; unsigned long long *counter, *sink;
  CK(hipMalloc(&rows, n * 16)); CK(hipMalloc(&o1, n * 8)); CK(hipMalloc(&o2, n * 8)); CK(hipMalloc(&counter, 8)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(rows, 1, n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wgs : {65536, 8192, 1024}) {
    for (int mode = 0; mode < 3; mode++) {
      for (int it = 0; it < 2; it++) {
        CK(hipMemset(counter, 0, 8));
        CK(hipEventRecord(e0));
        if (mode == 0) stream_kernel<0><<<wgs, 512>>>(rows, n / wgs, o1, o2, counter, sink);
        if (mode == 1) stream_kernel<1><<<wgs, 512>>>(rows, n / wgs, o1, o2, counter, sink);
        if (mode == 2) stream_kernel<2><<<wgs, 512>>>(rows, n / wgs, o1, o2, counter, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it == 1) printf("wgs %6d mode %d: %.3f ms\n", wgs, mode, t);
      }
    }
  }
  return 0;
}
