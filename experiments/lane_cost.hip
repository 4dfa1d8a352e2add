// Experiment: does a divergent 16-byte wave-load cost per ACTIVE LANE or per INSTRUCTION?
// Every wave issues the same number of load instructions; only `active` lanes of each wave take part.
// Table 3 MiB (L2-resident), random lines.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t fmix64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }

template <bool DEP>
__global__ __launch_bounds__(512) void lane_kernel(const ulonglong2 *__restrict__ table, uint32_t nslots, int iters, int active,
                                                   unsigned long long *__restrict__ sink) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  unsigned long long acc = 0;
  if ((int)(threadIdx.x & 63) < active) {
    uint32_t idx = (uint32_t)((fmix64(tid) & 0xFFFFFFFFull) * nslots >> 32);
    for (int it = 0; it < iters; it++) {
      ulonglong2 r = table[idx];
      acc += r.y;
      if (DEP) idx = (uint32_t)((fmix64(r.x + idx + it) & 0xFFFFFFFFull) * nslots >> 32);   // true dependency on the loaded value
      else idx = (uint32_t)((fmix64(tid + it * 977) & 0xFFFFFFFFull) * nslots >> 32);
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

int main() {
  const uint32_t nslots = 3 * 1024 * 1024 / 16;
  ulonglong2 *table; unsigned long long *sink;
  CK(hipMalloc(&table, (size_t)nslots * 16)); CK(hipMalloc(&sink, 8));
  // fill with pseudo-random content so dependent chains wander over the whole table
  unsigned long long *h = (unsigned long long *)malloc((size_t)nslots * 16);
  unsigned long long s = 88172645463325252ull;
  for (size_t i = 0; i < (size_t)nslots * 2; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = s; }
  CK(hipMemcpy(table, h, (size_t)nslots * 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 3, iters = 1024;  // 24 waves per CU, like the probe kernel
  for (int dep = 0; dep < 2; dep++)
    for (int active : {64, 32, 16, 4, 1}) {
      for (int it = 0; it < 2; it++) {
        CK(hipEventRecord(e0));
        if (dep) lane_kernel<true><<<grid, 512>>>(table, nslots, iters, active, sink);
        else lane_kernel<false><<<grid, 512>>>(table, nslots, iters, active, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it == 1) {
          double waveloads = (double)grid * 8 * iters;
          printf("%s active lanes %2d: %.3f ms  -> %.1f ns per wave-load per CU (%.0f cycles @2.1GHz), %.1f G lane-loads/s\n", dep ? "dependent  " : "independent",
                 active, t, t * 1e6 / (waveloads / 256), t * 1e6 / (waveloads / 256) * 2.1, waveloads * active / t / 1e6);
        }
      }
    }
  return 0;
}
