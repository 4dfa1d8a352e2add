// Experiment: one lane per ~184-byte record, records contiguous.  How fast can lanes pull their own record through
// registers with dependent 8-byte (or 16-byte) unaligned global loads, compared with staging tiles in LDS?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int W>  // bytes per load: 8 or 16
__global__ __launch_bounds__(256) void stream_kernel(const uint8_t *__restrict__ data, const int64_t *__restrict__ rec_off, int64_t n,
                                                     int rec_len, unsigned long long *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += stride) {
    const uint8_t *p = data + rec_off[row];
    unsigned long long acc = 0;
    int pos = 0;
    while (pos < rec_len) {  // each step depends on the bytes just read (as a parser's would)
      if (W == 8) {
        unsigned long long v;
        __builtin_memcpy(&v, p + pos, 8);
        acc += v;
        pos += 8 + (int)(v & 0);
      } else {
        unsigned long long v[2];
        __builtin_memcpy(v, p + pos, 16);
        acc += v[0] ^ v[1];
        pos += 16 + (int)(v[0] & 0);
      }
    }
    out[row] = acc;
  }
}

int main() {
  const int64_t n = 1 << 22;
  const int rec_len = 181;
  const size_t bytes = (size_t)n * rec_len + 64;
  uint8_t *data; int64_t *off; unsigned long long *out;
  CK(hipMalloc(&data, bytes)); CK(hipMemset(data, 0x11, bytes));
  CK(hipMalloc(&off, n * 8)); CK(hipMalloc(&out, n * 8));
  int64_t *h = (int64_t *)malloc(n * 8);
  for (int64_t i = 0; i < n; i++) h[i] = i * rec_len;
  CK(hipMemcpy(off, h, n * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++)
    for (int grid : {1024, 2048, 8192, 16384})
      for (int it = 0; it < 2; it++) {
        CK(hipEventRecord(e0));
        if (w == 0) stream_kernel<8><<<grid, 256>>>(data, off, n, rec_len, out);
        else stream_kernel<16><<<grid, 256>>>(data, off, n, rec_len, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it == 1) printf("%2d-byte loads, grid %5d: %.3f ms  (%.1f GB/s of input, %.2f G rows/s)\n", w ? 16 : 8, grid, t, n * (double)rec_len / t / 1e6, n / t / 1e6);
      }
  return 0;
}
