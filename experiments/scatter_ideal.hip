// Experiment: upper bound for a scatter of 16-B rows when every store is a full aligned
// G-byte segment (G = 64, 128, 256, 1024) at a pseudo-random destination; inputs stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t fmix64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }

template <int LANES>  // lanes per segment (16 B each)
__global__ __launch_bounds__(1024) void scatter_ideal(const ulonglong2 *k2, const ulonglong2 *w2, ulonglong2 *out, int64_t n, int lg) {
  int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n / 2; i += stride) {
    ulonglong2 a = k2[i], b = w2[i];
    // rows 2i, 2i+1 -> consecutive 16-B slots of segment (2i / LANES)
    int64_t seg = (2 * i) / LANES;
    int64_t dseg = fmix64(seg) & ((n / LANES) - 1);
    int64_t d = dseg * LANES + ((2 * i) % LANES);
    out[d] = make_ulonglong2(a.x, b.x);
    out[d + 1] = make_ulonglong2(a.y, b.y);
  }
}
int main(int argc, char **argv) {
  int lg = argc > 1 ? atoi(argv[1]) : 29;
  int64_t n = int64_t(1) << lg;
  uint64_t *k, *w; ulonglong2 *out;
  CK(hipMalloc(&k, n * 8)); CK(hipMalloc(&w, n * 8)); CK(hipMalloc(&out, n * 16));
  CK(hipMemset(k, 1, n * 8)); CK(hipMemset(w, 2, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int v = 0; v < 5; v++) {
    for (int it = 0; it < 2; it++) {
      CK(hipEventRecord(e0));
      const int grid = 2048;
      switch (v) {
        case 0: scatter_ideal<2><<<grid, 1024>>>((ulonglong2 *)k, (ulonglong2 *)w, out, n, lg); break;
        case 1: scatter_ideal<4><<<grid, 1024>>>((ulonglong2 *)k, (ulonglong2 *)w, out, n, lg); break;
        case 2: scatter_ideal<8><<<grid, 1024>>>((ulonglong2 *)k, (ulonglong2 *)w, out, n, lg); break;
        case 3: scatter_ideal<16><<<grid, 1024>>>((ulonglong2 *)k, (ulonglong2 *)w, out, n, lg); break;
        case 4: scatter_ideal<64><<<grid, 1024>>>((ulonglong2 *)k, (ulonglong2 *)w, out, n, lg); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      printf("segment %4d B: %.3f ms (%.0f GB/s r+w)\n", (v == 4 ? 64 : (2 << v)) * 16, t, n * 32 / t / 1e6);
    }
  }
  return 0;
}
