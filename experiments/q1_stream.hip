// Experiment: what bounds the fused Q1 pass (46 B/row: shipdate 8, two 1-byte BYTES keys + 4 B of shared offsets,
// four FLOAT64 inputs; WHERE, two projections, 6 LDS atomics per row)?  Forms compared on 2^27 rows:
//   MODE 0  read everything, xor into a register (the streaming ceiling for this access pattern)
//   MODE 1  + key bytes through the offsets, LDS key-table lookup, WHERE, 6 LDS atomics per row (R replicas)
// each as "strided" (a thread's ROWS rows are grid-strides apart, 8-byte loads) and "pair" (a thread owns two
// neighbouring rows: 16-byte loads), at several block sizes / workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

struct Cols {
  const long long *ship;
  const uint8_t *rf, *ls;
  const uint32_t *offs;
  const double *qty, *price, *disc, *tax;
  int64_t n;
  long long cutoff;
  double *out;  // [6 planes][8 groups]
};

static constexpr int G = 6, R = 64, PLANES = 6;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef long long l2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t gid_of(const uint32_t *t_gid, uint8_t a, uint8_t b) {
  const uint32_t h = (a * 0x9E3779B1u + b * 0x85EBCA77u) >> 26;  // 64-slot LDS table
  return t_gid[h];
}

template <int ROWS, int MODE, int BLK, bool NT>
__global__ __launch_bounds__(BLK) void q1_strided(Cols c) {
  __shared__ double acc[PLANES][(G + 1) * R];
  __shared__ uint32_t t_gid[64];
  for (int i = threadIdx.x; i < PLANES * (G + 1) * R; i += BLK) (&acc[0][0])[i] = 0;
  if (threadIdx.x < 64) t_gid[threadIdx.x] = G;
  __syncthreads();
  if (threadIdx.x < 6) {
    const uint8_t a = 65 + threadIdx.x / 2, b = 70 + threadIdx.x % 2;
    t_gid[(a * 0x9E3779B1u + b * 0x85EBCA77u) >> 26] = threadIdx.x;  // (collisions are harmless for the timing)
  }
  __syncthreads();
  const uint32_t rep = threadIdx.x & (R - 1);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * BLK;
  unsigned long long sink = 0;
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * BLK + threadIdx.x; i0 < c.n; i0 += stride * ROWS) {
    int64_t row[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) row[k] = i0 + k * stride < c.n ? i0 + k * stride : 0;
    long long ship[ROWS];
    double q[ROWS], p[ROWS], d[ROWS], t[ROWS];
    uint32_t o0[ROWS], o1[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
      if (NT) {
        ship[k] = __builtin_nontemporal_load(c.ship + row[k]);
        q[k] = __builtin_nontemporal_load(c.qty + row[k]);
        p[k] = __builtin_nontemporal_load(c.price + row[k]);
        d[k] = __builtin_nontemporal_load(c.disc + row[k]);
        t[k] = __builtin_nontemporal_load(c.tax + row[k]);
      } else {
        ship[k] = c.ship[row[k]]; q[k] = c.qty[row[k]]; p[k] = c.price[row[k]]; d[k] = c.disc[row[k]]; t[k] = c.tax[row[k]];
      }
      o0[k] = c.offs[row[k]];
      o1[k] = c.offs[row[k] + 1];
    }
    uint8_t a[ROWS], b[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
      a[k] = o1[k] > o0[k] ? c.rf[o0[k]] : 0;
      b[k] = o1[k] > o0[k] ? c.ls[o0[k]] : 0;
    }
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < ROWS; k++)
        sink += ship[k] ^ __double_as_longlong(q[k]) ^ __double_as_longlong(p[k]) ^ __double_as_longlong(d[k]) ^ __double_as_longlong(t[k]) ^ a[k] ^ (b[k] << 8);
    } else {
#pragma unroll
      for (int k = 0; k < ROWS; k++) {
        const bool ok = i0 + k * stride < c.n && ship[k] <= c.cutoff;
        const uint32_t g = gid_of(t_gid, a[k], b[k]);
        const uint32_t at = (ok ? g : G) * R + rep;
        const double dp = p[k] * (1.0 - d[k]), ch = dp * (1.0 + t[k]);
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[0][at]), 1ull);
        unsafeAtomicAdd(&acc[1][at], q[k]);
        unsafeAtomicAdd(&acc[2][at], p[k]);
        unsafeAtomicAdd(&acc[3][at], dp);
        unsafeAtomicAdd(&acc[4][at], ch);
        unsafeAtomicAdd(&acc[5][at], d[k]);
      }
    }
  }
  __syncthreads();
  if (MODE == 0) {
    if (sink == 0x1234567) c.out[0] = 1;
  } else {
    for (int t = threadIdx.x; t < PLANES * G; t += BLK) {
      const int pl = t / G, g = t % G;
      double s = 0;
#pragma unroll 2
      for (int r = 0; r < R; r++) s += pl == 0 ? static_cast<double>(reinterpret_cast<unsigned long long *>(&acc[0][0])[g * R + r]) : acc[pl][g * R + r];
      unsafeAtomicAdd(&c.out[pl * 8 + g], s);
    }
  }
}

// a thread owns rows 2j, 2j+1: 16-byte loads of the 8-byte columns, one 8-byte load of two offsets + the next one
template <int PAIRS, int MODE, int BLK, bool NT>
__global__ __launch_bounds__(BLK) void q1_pair(Cols c) {
  __shared__ double acc[PLANES][(G + 1) * R];
  __shared__ uint32_t t_gid[64];
  for (int i = threadIdx.x; i < PLANES * (G + 1) * R; i += BLK) (&acc[0][0])[i] = 0;
  if (threadIdx.x < 64) t_gid[threadIdx.x] = G;
  __syncthreads();
  if (threadIdx.x < 6) {
    const uint8_t a = 65 + threadIdx.x / 2, b = 70 + threadIdx.x % 2;
    t_gid[(a * 0x9E3779B1u + b * 0x85EBCA77u) >> 26] = threadIdx.x;
  }
  __syncthreads();
  const uint32_t rep = threadIdx.x & (R - 1);
  const int64_t npairs = c.n / 2;  // (n even in this experiment)
  const int64_t stride = static_cast<int64_t>(gridDim.x) * BLK;
  unsigned long long sink = 0;
  for (int64_t j0 = static_cast<int64_t>(blockIdx.x) * BLK + threadIdx.x; j0 < npairs; j0 += stride * PAIRS) {
    int64_t pr[PAIRS];
#pragma unroll
    for (int k = 0; k < PAIRS; k++) pr[k] = j0 + k * stride < npairs ? j0 + k * stride : 0;
    l2 ship[PAIRS];
    d2 q[PAIRS], p[PAIRS], d[PAIRS], t[PAIRS];
    u2 o[PAIRS];
    uint32_t o2[PAIRS];
#pragma unroll
    for (int k = 0; k < PAIRS; k++) {
      if (NT) {
        ship[k] = __builtin_nontemporal_load(reinterpret_cast<const l2 *>(c.ship) + pr[k]);
        q[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(c.qty) + pr[k]);
        p[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(c.price) + pr[k]);
        d[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(c.disc) + pr[k]);
        t[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(c.tax) + pr[k]);
      } else {
        ship[k] = reinterpret_cast<const l2 *>(c.ship)[pr[k]];
        q[k] = reinterpret_cast<const d2 *>(c.qty)[pr[k]];
        p[k] = reinterpret_cast<const d2 *>(c.price)[pr[k]];
        d[k] = reinterpret_cast<const d2 *>(c.disc)[pr[k]];
        t[k] = reinterpret_cast<const d2 *>(c.tax)[pr[k]];
      }
      o[k] = reinterpret_cast<const u2 *>(c.offs)[pr[k]];
      o2[k] = c.offs[2 * pr[k] + 2];
    }
    uint8_t a[PAIRS][2], b[PAIRS][2];
#pragma unroll
    for (int k = 0; k < PAIRS; k++) {
      a[k][0] = o[k].y > o[k].x ? c.rf[o[k].x] : 0;
      b[k][0] = o[k].y > o[k].x ? c.ls[o[k].x] : 0;
      a[k][1] = o2[k] > o[k].y ? c.rf[o[k].y] : 0;
      b[k][1] = o2[k] > o[k].y ? c.ls[o[k].y] : 0;
    }
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < PAIRS; k++)
        for (int h = 0; h < 2; h++)
          sink += ship[k][h] ^ __double_as_longlong(q[k][h]) ^ __double_as_longlong(p[k][h]) ^ __double_as_longlong(d[k][h]) ^
                  __double_as_longlong(t[k][h]) ^ a[k][h] ^ (b[k][h] << 8);
    } else {
#pragma unroll
      for (int k = 0; k < PAIRS; k++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const bool ok = j0 + k * stride < npairs && ship[k][h] <= c.cutoff;
          const uint32_t g = gid_of(t_gid, a[k][h], b[k][h]);
          const uint32_t at = (ok ? g : G) * R + rep;
          const double dp = p[k][h] * (1.0 - d[k][h]), ch = dp * (1.0 + t[k][h]);
          atomicAdd(reinterpret_cast<unsigned long long *>(&acc[0][at]), 1ull);
          unsafeAtomicAdd(&acc[1][at], q[k][h]);
          unsafeAtomicAdd(&acc[2][at], p[k][h]);
          unsafeAtomicAdd(&acc[3][at], dp);
          unsafeAtomicAdd(&acc[4][at], ch);
          unsafeAtomicAdd(&acc[5][at], d[k][h]);
        }
      }
    }
  }
  __syncthreads();
  if (MODE == 0) {
    if (sink == 0x1234567) c.out[0] = 1;
  } else {
    for (int t = threadIdx.x; t < PLANES * G; t += BLK) {
      const int pl = t / G, g = t % G;
      double s = 0;
#pragma unroll 2
      for (int r = 0; r < R; r++) s += pl == 0 ? static_cast<double>(reinterpret_cast<unsigned long long *>(&acc[0][0])[g * R + r]) : acc[pl][g * R + r];
      unsafeAtomicAdd(&c.out[pl * 8 + g], s);
    }
  }
}

template <typename K>
static void timeit(const char *name, K kern, int blk, int wgs, Cols c) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int it = 0; it < 4; it++) {
    CK(hipMemset(c.out, 0, 48 * 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(blk), 0, 0, c);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (it && t < best) best = t;
  }
  double h[48];
  CK(hipMemcpy(h, c.out, sizeof(h), hipMemcpyDeviceToHost));
  double cnt = 0;
  for (int g = 0; g < 6; g++) cnt += h[g];
  printf("%-34s blk %4d wgs %5d: %.3f ms  %.2f TB/s (46 B/row)  count %.0f\n", name, blk, wgs, best, c.n * 46.0 / best / 1e9, cnt);
  fflush(stdout);
}

int main() {
  const int64_t n = int64_t(1) << 27;
  std::vector<long long> ship(n);
  std::vector<uint8_t> rf(n), ls(n);
  std::vector<uint32_t> offs(n + 1);
  std::vector<double> f(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  Cols c;
  void *d;
  for (int64_t i = 0; i < n; i++) { ship[i] = 8036 + rnd() % 2526; rf[i] = 65 + rnd() % 3; ls[i] = 70 + rnd() % 2; offs[i] = i; }
  offs[n] = n;
  CK(hipMalloc(&d, n * 8)); CK(hipMemcpy(d, ship.data(), n * 8, hipMemcpyHostToDevice)); c.ship = (const long long *)d;
  CK(hipMalloc(&d, n)); CK(hipMemcpy(d, rf.data(), n, hipMemcpyHostToDevice)); c.rf = (const uint8_t *)d;
  CK(hipMalloc(&d, n)); CK(hipMemcpy(d, ls.data(), n, hipMemcpyHostToDevice)); c.ls = (const uint8_t *)d;
  CK(hipMalloc(&d, (n + 1) * 4)); CK(hipMemcpy(d, offs.data(), (n + 1) * 4, hipMemcpyHostToDevice)); c.offs = (const uint32_t *)d;
  const double **dst[4] = {&c.qty, &c.price, &c.disc, &c.tax};
  for (int k = 0; k < 4; k++) {
    for (int64_t i = 0; i < n; i++) f[i] = (rnd() % 1000) / 100.0;
    CK(hipMalloc(&d, n * 8)); CK(hipMemcpy(d, f.data(), n * 8, hipMemcpyHostToDevice)); *dst[k] = (const double *)d;
  }
  CK(hipMalloc(&d, 48 * 8)); c.out = (double *)d;
  c.n = n;
  c.cutoff = 10562 - 36;
  int cus = 256;
  for (int per_cu : {2, 4, 8}) {
    timeit("strided rows1 read-only", q1_strided<1, 0, 256, false>, 256, cus * per_cu, c);
    timeit("strided rows2 read-only", q1_strided<2, 0, 256, false>, 256, cus * per_cu, c);
    timeit("strided rows4 read-only", q1_strided<4, 0, 256, false>, 256, cus * per_cu, c);
    timeit("strided rows4 read-only nt", q1_strided<4, 0, 256, true>, 256, cus * per_cu, c);
    timeit("pair x1 read-only", q1_pair<1, 0, 256, false>, 256, cus * per_cu, c);
    timeit("pair x2 read-only", q1_pair<2, 0, 256, false>, 256, cus * per_cu, c);
    timeit("pair x2 read-only nt", q1_pair<2, 0, 256, true>, 256, cus * per_cu, c);
    timeit("strided rows1 full", q1_strided<1, 1, 256, false>, 256, cus * per_cu, c);
    timeit("strided rows2 full", q1_strided<2, 1, 256, false>, 256, cus * per_cu, c);
    timeit("strided rows4 full", q1_strided<4, 1, 256, false>, 256, cus * per_cu, c);
    timeit("strided rows4 full nt", q1_strided<4, 1, 256, true>, 256, cus * per_cu, c);
    timeit("pair x1 full", q1_pair<1, 1, 256, false>, 256, cus * per_cu, c);
    timeit("pair x2 full", q1_pair<2, 1, 256, false>, 256, cus * per_cu, c);
    timeit("pair x2 full nt", q1_pair<2, 1, 256, true>, 256, cus * per_cu, c);
  }
  for (int per_cu : {1, 2, 4}) {
    timeit("strided rows2 full b512", q1_strided<2, 1, 512, false>, 512, cus * per_cu, c);
    timeit("strided rows4 full b512", q1_strided<4, 1, 512, false>, 512, cus * per_cu, c);
    timeit("pair x2 full b512", q1_pair<2, 1, 512, false>, 512, cus * per_cu, c);
    timeit("pair x2 full nt b512", q1_pair<2, 1, 512, true>, 512, cus * per_cu, c);
  }
  return 0;
}
