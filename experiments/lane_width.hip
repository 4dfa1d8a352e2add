// Experiment (round 3): does the cost of a divergent wave-load depend on the WIDTH of the per-lane access?
// lane_cost.hip measured ~2.1 cycles per active lane for 16-byte loads from an L2-resident table.  Here: 4-, 8- and 16-byte
// loads from tables of 1 MiB and 3 MiB, 64 active lanes, independent addresses (throughput, as in the probe kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t fmix64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }

template <typename T> __device__ __forceinline__ unsigned long long low(const T &v);
template <> __device__ __forceinline__ unsigned long long low<uint32_t>(const uint32_t &v) { return v; }
template <> __device__ __forceinline__ unsigned long long low<unsigned long long>(const unsigned long long &v) { return v; }
template <> __device__ __forceinline__ unsigned long long low<ulonglong2>(const ulonglong2 &v) { return v.y; }

template <typename T>
__global__ __launch_bounds__(512) void width_kernel(const T *__restrict__ table, uint32_t nslots, int iters, unsigned long long *__restrict__ sink) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  unsigned long long acc = 0;
  uint32_t idx = (uint32_t)((fmix64(tid) & 0xFFFFFFFFull) * nslots >> 32);
  for (int it = 0; it < iters; it++) {
    acc += low<T>(table[idx]);
    idx = (uint32_t)((fmix64(tid + it * 977) & 0xFFFFFFFFull) * nslots >> 32);
  }
  if (acc == 0x1234567) sink[0] = acc;
}

template <typename T>
static void run(const char *name, void *table, size_t bytes, unsigned long long *sink) {
  const uint32_t nslots = (uint32_t)(bytes / sizeof(T));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 3, iters = 1024;
  for (int it = 0; it < 2; it++) {
    CK(hipEventRecord(e0));
    width_kernel<T><<<grid, 512>>>((const T *)table, nslots, iters, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (it == 1) {
      double waveloads = (double)grid * 8 * iters;
      printf("%-8s table %4zu KiB: %.3f ms -> %.0f cycles per wave-load per CU, %.1f G lane-loads/s\n", name, bytes >> 10, t,
             t * 1e6 / (waveloads / 256) * 2.1, waveloads * 64 / t / 1e6);
    }
  }
}

int main() {
  const size_t maxb = 8u << 20;
  void *table; unsigned long long *sink;
  CK(hipMalloc(&table, maxb)); CK(hipMalloc(&sink, 8));
  unsigned long long *h = (unsigned long long *)malloc(maxb);
  unsigned long long s = 88172645463325252ull;
  for (size_t i = 0; i < maxb / 8; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s; }
  CK(hipMemcpy(table, h, maxb, hipMemcpyHostToDevice));
  for (size_t bytes : {(size_t)64 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)3 << 20, (size_t)6 << 20}) {
    run<uint32_t>("4-byte", table, bytes, sink);
    run<unsigned long long>("8-byte", table, bytes, sink);
    run<ulonglong2>("16-byte", table, bytes, sink);
  }
  return 0;
}
