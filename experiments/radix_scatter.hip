// Experiment: one-pass radix partition of (k, w) int64 rows into 2^B partitions of
// 16-byte AoS rows, with sector-aligned flushes (per-partition 64-B carry buffers in
// LDS) against a plain direct scatter.  Standalone: hipcc -O3 --offload-arch=gfx950.
//   ./radix_scatter [log2_rows=29] [bits=11] [iters=3]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);    \
      exit(1);                                                                                  \
    }                                                                                           \
  } while (0)

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

__global__ void gen_kernel(uint64_t *k, uint64_t *w, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) {
    k[i] = fmix64(uint64_t(i) * 0x9E3779B97F4A7C15ULL + 12345) >> 3;  // distinct-ish random keys
    w[i] = uint64_t(i);
  }
}

__global__ void copy_kernel(const ulonglong2 *a, const ulonglong2 *b, ulonglong2 *out, int64_t n2) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n2; i += stride) {
    ulonglong2 x = a[i], y = b[i];
    out[2 * i] = make_ulonglong2(x.x, y.x);
    out[2 * i + 1] = make_ulonglong2(x.y, y.y);
  }
}

template <int BITS>
__device__ __forceinline__ uint32_t part_of(uint64_t key) {
  return uint32_t(fmix64(key) >> (64 - BITS));
}

// ---- histogram: one workgroup per chunk
template <int BITS, int THREADS>
__global__ __launch_bounds__(THREADS) void hist_kernel(const uint64_t *__restrict__ k, int64_t n, int64_t chunk_rows,
                                                       uint32_t *__restrict__ hist) {
  constexpr int P = 1 << BITS;
  __shared__ uint32_t h[P];
  for (int i = threadIdx.x; i < P; i += THREADS) h[i] = 0;
  __syncthreads();
  const int64_t lo = blockIdx.x * chunk_rows;
  int64_t hi = lo + chunk_rows;
  if (hi > n) hi = n;
  const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(k);
  for (int64_t i = lo / 2 + threadIdx.x; i < hi / 2; i += THREADS) {
    ulonglong2 x = k2[i];
    atomicAdd(&h[part_of<BITS>(x.x)], 1u);
    atomicAdd(&h[part_of<BITS>(x.y)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P; i += THREADS) hist[(int64_t)blockIdx.x * P + i] = h[i];
}

// ---- scan: offs[c][p] = start[p] + sum_{c'<c} hist[c'][p]; start[p] aligned to 8 rows
// One block of 1024 threads per 64 partitions: thread (g = tid/64, px = tid%64) owns chunk group g.
template <int BITS>
__global__ __launch_bounds__(1024) void colsum_kernel(const uint32_t *__restrict__ hist, int nchunks,
                                                      uint32_t *__restrict__ totals) {
  constexpr int P = 1 << BITS;
  const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + px;
  __shared__ uint32_t s[16][64];
  const int per = (nchunks + 15) / 16;
  uint32_t sum = 0;
  for (int c = g * per; c < (g + 1) * per && c < nchunks; c++) sum += hist[(int64_t)c * P + p];
  s[g][px] = sum;
  __syncthreads();
  if (g == 0) {
    uint32_t t = 0;
    for (int i = 0; i < 16; i++) t += s[i][px];
    totals[p] = t;
  }
}

template <int BITS>
__global__ __launch_bounds__(1024) void starts_kernel(const uint32_t *__restrict__ totals, uint32_t *__restrict__ starts) {
  constexpr int P = 1 << BITS;
  __shared__ uint32_t s[P];
  for (int i = threadIdx.x; i < P; i += 1024) s[i] = (totals[i] + 7u) & ~7u;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) {
      uint32_t t = s[i];
      s[i] = acc;
      acc += t;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P; i += 1024) starts[i] = s[i];
}

template <int BITS>
__global__ __launch_bounds__(1024) void offs_kernel(uint32_t *__restrict__ hist, int nchunks,
                                                    const uint32_t *__restrict__ starts) {
  constexpr int P = 1 << BITS;
  const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + px;
  __shared__ uint32_t s[16][64];
  const int per = (nchunks + 15) / 16;
  uint32_t sum = 0;
  for (int c = g * per; c < (g + 1) * per && c < nchunks; c++) sum += hist[(int64_t)c * P + p];
  s[g][px] = sum;
  __syncthreads();
  uint32_t base = starts[p];
  for (int i = 0; i < g; i++) base += s[i][px];
  for (int c = g * per; c < (g + 1) * per && c < nchunks; c++) {
    uint32_t v = hist[(int64_t)c * P + p];
    hist[(int64_t)c * P + p] = base;
    base += v;
  }
}

// ---- scatter with sector-aligned flushes
template <int BITS, int THREADS, int ROWS>
__global__ __launch_bounds__(THREADS) void scatter_carry_kernel(const uint64_t *__restrict__ k,
                                                                const uint64_t *__restrict__ w, int64_t n,
                                                                int64_t chunk_rows, const uint32_t *__restrict__ offs,
                                                                ulonglong2 *__restrict__ out) {
  constexpr int P = 1 << BITS;
  constexpr int OWN = (P + THREADS - 1) / THREADS;
  static_assert(ROWS % 2 == 0, "rows per thread must be even");
  __shared__ ulonglong2 carry[P * 4];
  __shared__ uint32_t cnt[P];
  __shared__ uint32_t gbase[P];
  uint32_t olo[OWN], ocnt[OWN];
#pragma unroll
  for (int o = 0; o < OWN; o++) {
    const int p = threadIdx.x + o * THREADS;
    if (p < P) {
      const uint32_t g0 = offs[(int64_t)blockIdx.x * P + p];
      const uint32_t a = g0 & 3u;
      gbase[p] = g0 - a;
      cnt[p] = a;
      olo[o] = a;
      ocnt[o] = a;
    }
  }
  __syncthreads();
  const int64_t lo = blockIdx.x * chunk_rows;
  int64_t hi = lo + chunk_rows;
  if (hi > n) hi = n;
  const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(k);
  const ulonglong2 *w2 = reinterpret_cast<const ulonglong2 *>(w);
  constexpr int TILE = THREADS * ROWS;
  ulonglong2 nk[ROWS / 2], nw[ROWS / 2];
#pragma unroll
  for (int j = 0; j < ROWS / 2; j++) {
    const int64_t r = lo + (int64_t)(j * THREADS + threadIdx.x) * 2;
    if (r < hi) nk[j] = k2[r >> 1], nw[j] = w2[r >> 1];
  }
  for (int64_t t0 = lo; t0 < hi; t0 += TILE) {
    uint64_t kk[ROWS], ww[ROWS];
    uint32_t pp[ROWS], rk[ROWS], tot[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS / 2; j++) {
      const int64_t r = t0 + (int64_t)(j * THREADS + threadIdx.x) * 2;
      if (r < hi) {  // n and chunk_rows are even: both rows valid together
        const ulonglong2 a = nk[j], b = nw[j];
        kk[2 * j] = a.x, kk[2 * j + 1] = a.y, ww[2 * j] = b.x, ww[2 * j + 1] = b.y;
        pp[2 * j] = part_of<BITS>(a.x), pp[2 * j + 1] = part_of<BITS>(a.y);
      } else {
        pp[2 * j] = pp[2 * j + 1] = 0xFFFFFFFFu;
      }
    }
#pragma unroll
    for (int j = 0; j < ROWS / 2; j++) {  // prefetch the next tile while this one is ranked and flushed
      const int64_t r = t0 + TILE + (int64_t)(j * THREADS + threadIdx.x) * 2;
      if (r < hi) nk[j] = k2[r >> 1], nw[j] = w2[r >> 1];
    }
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      if (pp[j] != 0xFFFFFFFFu) rk[j] = atomicAdd(&cnt[pp[j]], 1u);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      if (pp[j] != 0xFFFFFFFFu) {
        tot[j] = cnt[pp[j]];
        const uint32_t nfl = tot[j] & ~3u;
        if (rk[j] < nfl) out[gbase[pp[j]] + rk[j]] = make_ulonglong2(kk[j], ww[j]);
      }
    }
#pragma unroll
    for (int o = 0; o < OWN; o++) {
      const int p = threadIdx.x + o * THREADS;
      if (p < P && cnt[p] >= 4) {
        const uint32_t gb = gbase[p];
        for (uint32_t s = olo[o]; s < ocnt[o]; s++) out[gb + s] = carry[p * 4 + s];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      if (pp[j] != 0xFFFFFFFFu) {
        const uint32_t nfl = tot[j] & ~3u;
        if (rk[j] >= nfl) carry[pp[j] * 4 + (rk[j] - nfl)] = make_ulonglong2(kk[j], ww[j]);
      }
    }
#pragma unroll
    for (int o = 0; o < OWN; o++) {
      const int p = threadIdx.x + o * THREADS;
      if (p < P) {
        const uint32_t t = cnt[p], nfl = t & ~3u;
        if (nfl) {
          gbase[p] += nfl;
          olo[o] = 0;
        }
        cnt[p] = t - nfl;
        ocnt[o] = t - nfl;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int o = 0; o < OWN; o++) {
    const int p = threadIdx.x + o * THREADS;
    if (p < P) {
      const uint32_t gb = gbase[p];
      for (uint32_t s = olo[o]; s < ocnt[o]; s++) out[gb + s] = carry[p * 4 + s];
    }
  }
}


// ---- block-wide exclusive scan of one value per thread (THREADS multiple of 64)
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum /*[THREADS/64+1]*/, uint32_t *total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  if (wv == 0) {
    uint32_t t = lane < THREADS / 64 ? wsum[lane] : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t y = __shfl_up(t, d, 64);
      if (lane >= d) t += y;
    }
    if (lane < THREADS / 64) wsum[lane] = t;  // inclusive wave totals
  }
  __syncthreads();
  const uint32_t base = wv ? wsum[wv - 1] : 0;
  *total = wsum[THREADS / 64 - 1];
  return base + x - v;
}

// ---- tile-local counting sort in LDS, coalesced flush of whole runs (no carry)
template <int BITS, int THREADS, int ROWS>
__global__ __launch_bounds__(THREADS) void scatter_sorted_kernel(const uint64_t *__restrict__ k,
                                                                 const uint64_t *__restrict__ w, int64_t n,
                                                                 int64_t chunk_rows, const uint32_t *__restrict__ offs,
                                                                 ulonglong2 *__restrict__ out) {
  constexpr int P = 1 << BITS;
  static_assert(P <= THREADS, "one owner thread per partition");
  constexpr int TILE = THREADS * ROWS;
  __shared__ ulonglong2 staging[TILE];
  __shared__ uint32_t dest[TILE];
  __shared__ uint32_t cnt[P], start[P], cur[P];
  __shared__ uint32_t wsum[THREADS / 64 + 1];
  if (threadIdx.x < P) {
    cur[threadIdx.x] = offs[(int64_t)blockIdx.x * P + threadIdx.x];
    cnt[threadIdx.x] = 0;
  }
  __syncthreads();
  const int64_t lo = blockIdx.x * chunk_rows;
  int64_t hi = lo + chunk_rows;
  if (hi > n) hi = n;
  const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(k);
  const ulonglong2 *w2 = reinterpret_cast<const ulonglong2 *>(w);
  ulonglong2 nk[ROWS / 2], nw[ROWS / 2];
#pragma unroll
  for (int j = 0; j < ROWS / 2; j++) {
    const int64_t r = lo + (int64_t)(j * THREADS + threadIdx.x) * 2;
    if (r < hi) nk[j] = k2[r >> 1], nw[j] = w2[r >> 1];
  }
  for (int64_t t0 = lo; t0 < hi; t0 += TILE) {
    uint64_t kk[ROWS], ww[ROWS];
    uint32_t pp[ROWS], rk[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS / 2; j++) {
      const int64_t r = t0 + (int64_t)(j * THREADS + threadIdx.x) * 2;
      if (r < hi) {
        const ulonglong2 a = nk[j], b = nw[j];
        kk[2 * j] = a.x, kk[2 * j + 1] = a.y, ww[2 * j] = b.x, ww[2 * j + 1] = b.y;
        pp[2 * j] = part_of<BITS>(a.x), pp[2 * j + 1] = part_of<BITS>(a.y);
      } else {
        pp[2 * j] = pp[2 * j + 1] = 0xFFFFFFFFu;
      }
    }
#pragma unroll
    for (int j = 0; j < ROWS / 2; j++) {
      const int64_t r = t0 + TILE + (int64_t)(j * THREADS + threadIdx.x) * 2;
      if (r < hi) nk[j] = k2[r >> 1], nw[j] = w2[r >> 1];
    }
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      if (pp[j] != 0xFFFFFFFFu) rk[j] = atomicAdd(&cnt[pp[j]], 1u);
    __syncthreads();
    uint32_t total;
    const uint32_t mine = threadIdx.x < P ? cnt[threadIdx.x] : 0;
    const uint32_t ex = block_excl_scan<THREADS>(mine, wsum, &total);
    if (threadIdx.x < P) start[threadIdx.x] = ex;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      if (pp[j] != 0xFFFFFFFFu) {
        const uint32_t pos = start[pp[j]] + rk[j];
        staging[pos] = make_ulonglong2(kk[j], ww[j]);
        dest[pos] = cur[pp[j]] + rk[j];
      }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < total; i += THREADS) out[dest[i]] = staging[i];
    if (threadIdx.x < P) {
      cur[threadIdx.x] += mine;
      cnt[threadIdx.x] = 0;
    }
    __syncthreads();
  }
}


// ---- per-partition 128-B line buffers in LDS: every global store is a full, lane-coalesced line
template <int BITS, int THREADS, int ROWS>
__global__ __launch_bounds__(THREADS) void scatter_bins_kernel(const uint64_t *__restrict__ k,
                                                               const uint64_t *__restrict__ w, int64_t n,
                                                               int64_t chunk_rows, const uint32_t *__restrict__ offs,
                                                               ulonglong2 *__restrict__ out) {
  constexpr int P = 1 << BITS;
  constexpr int OWN = (P + THREADS - 1) / THREADS;
  __shared__ ulonglong2 bins[P * 8];
  __shared__ uint32_t cnt[P];    // rows in the bin (incl. phantom slots before the first flush) + this tile's arrivals
  __shared__ uint32_t gbase[P];  // line-aligned absolute row position of bin slot 0
  __shared__ uint8_t lo[P];      // first real slot (chunk-start misalignment), 0 after the first flush
#pragma unroll
  for (int o = 0; o < OWN; o++) {
    const int p = threadIdx.x + o * THREADS;
    if (p < P) {
      const uint32_t g0 = offs[(int64_t)blockIdx.x * P + p];
      const uint32_t a = g0 & 7u;
      gbase[p] = g0 - a;
      cnt[p] = a;
      lo[p] = (uint8_t)a;
    }
  }
  __syncthreads();
  const int64_t lo_row = blockIdx.x * chunk_rows;
  int64_t hi = lo_row + chunk_rows;
  if (hi > n) hi = n;
  const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(k);
  const ulonglong2 *w2 = reinterpret_cast<const ulonglong2 *>(w);
  constexpr int TILE = THREADS * ROWS;
  ulonglong2 nk[ROWS / 2], nw[ROWS / 2];
#pragma unroll
  for (int j = 0; j < ROWS / 2; j++) {
    const int64_t r = lo_row + (int64_t)(j * THREADS + threadIdx.x) * 2;
    if (r < hi) nk[j] = k2[r >> 1], nw[j] = w2[r >> 1];
  }
  for (int64_t t0 = lo_row; t0 < hi; t0 += TILE) {
    uint64_t kk[ROWS], ww[ROWS];
    uint32_t pp[ROWS], rk[ROWS], nfl[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS / 2; j++) {
      const int64_t r = t0 + (int64_t)(j * THREADS + threadIdx.x) * 2;
      if (r < hi) {
        const ulonglong2 a = nk[j], b = nw[j];
        kk[2 * j] = a.x, kk[2 * j + 1] = a.y, ww[2 * j] = b.x, ww[2 * j + 1] = b.y;
        pp[2 * j] = part_of<BITS>(a.x), pp[2 * j + 1] = part_of<BITS>(a.y);
      } else {
        pp[2 * j] = pp[2 * j + 1] = 0xFFFFFFFFu;
      }
    }
#pragma unroll
    for (int j = 0; j < ROWS / 2; j++) {
      const int64_t r = t0 + TILE + (int64_t)(j * THREADS + threadIdx.x) * 2;
      if (r < hi) nk[j] = k2[r >> 1], nw[j] = w2[r >> 1];
    }
    // A: rank inside the partition's pending rows
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      if (pp[j] != 0xFFFFFFFFu) rk[j] = atomicAdd(&cnt[pp[j]], 1u);
    __syncthreads();
    // B: rows of the first line go to the bin, rows of further complete lines straight to HBM
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      if (pp[j] != 0xFFFFFFFFu) {
        nfl[j] = cnt[pp[j]] & ~7u;
        if (rk[j] < 8u) {
          if (nfl[j] != 0 || true) bins[pp[j] * 8 + rk[j]] = make_ulonglong2(kk[j], ww[j]);
        } else if (rk[j] < nfl[j]) {
          out[gbase[pp[j]] + rk[j]] = make_ulonglong2(kk[j], ww[j]);
        }
      }
    }
    __syncthreads();
    // C: flush every full first line, 8 lanes per line
    for (int i = threadIdx.x; i < P * 8; i += THREADS) {
      const int p = i >> 3, sl = i & 7;
      if (cnt[p] >= 8u && sl >= (int)lo[p]) out[gbase[p] + sl] = bins[i];
    }
    __syncthreads();
    // D: rows beyond the last complete line become the new bin content
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      if (pp[j] != 0xFFFFFFFFu && nfl[j] != 0 && rk[j] >= nfl[j] && rk[j] >= 8u)
        bins[pp[j] * 8 + (rk[j] - nfl[j])] = make_ulonglong2(kk[j], ww[j]);
#pragma unroll
    for (int o = 0; o < OWN; o++) {
      const int p = threadIdx.x + o * THREADS;
      if (p < P) {
        const uint32_t t = cnt[p], f = t & ~7u;
        if (f) {
          gbase[p] += f;
          cnt[p] = t - f;
          lo[p] = 0;
        }
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < P * 8; i += THREADS) {
    const int p = i >> 3, sl = i & 7;
    if (sl >= (int)lo[p] && sl < (int)cnt[p]) out[gbase[p] + sl] = bins[i];
  }
}

// ---- plain direct scatter: rank from an LDS counter, row stored straight from registers
template <int BITS, int THREADS, int ROWS>
__global__ __launch_bounds__(THREADS) void scatter_direct_kernel(const uint64_t *__restrict__ k,
                                                                 const uint64_t *__restrict__ w, int64_t n,
                                                                 int64_t chunk_rows, const uint32_t *__restrict__ offs,
                                                                 ulonglong2 *__restrict__ out) {
  constexpr int P = 1 << BITS;
  __shared__ uint32_t cur[P];
  for (int p = threadIdx.x; p < P; p += THREADS) cur[p] = offs[(int64_t)blockIdx.x * P + p];
  __syncthreads();
  const int64_t lo = blockIdx.x * chunk_rows;
  int64_t hi = lo + chunk_rows;
  if (hi > n) hi = n;
  const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(k);
  const ulonglong2 *w2 = reinterpret_cast<const ulonglong2 *>(w);
  constexpr int TILE = THREADS * ROWS;
  for (int64_t t0 = lo; t0 < hi; t0 += TILE) {
#pragma unroll
    for (int j = 0; j < ROWS / 2; j++) {
      const int64_t r = t0 + (int64_t)(j * THREADS + threadIdx.x) * 2;
      if (r < hi) {
        const ulonglong2 a = k2[r >> 1], b = w2[r >> 1];
        const uint32_t p0 = part_of<BITS>(a.x), p1 = part_of<BITS>(a.y);
        const uint32_t d0 = atomicAdd(&cur[p0], 1u);
        const uint32_t d1 = atomicAdd(&cur[p1], 1u);
        out[d0] = make_ulonglong2(a.x, b.x);
        out[d1] = make_ulonglong2(a.y, b.y);
      }
    }
  }
}

// ---- verification: every row of partition p hashes to p; checksums preserved
template <int BITS>
__global__ void verify_kernel(const ulonglong2 *out, const uint32_t *starts, const uint32_t *totals,
                              unsigned long long *sums, unsigned int *bad) {
  const int p = blockIdx.x;
  const uint32_t s = starts[p], c = totals[p];
  unsigned long long sk = 0, sw = 0;
  for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) {
    ulonglong2 r = out[s + i];
    if (part_of<BITS>(r.x) != (uint32_t)p) atomicAdd(bad, 1u);
    sk += r.x;
    sw += r.y * 0x9E3779B97F4A7C15ULL ^ r.x;
  }
  atomicAdd(&sums[0], sk);
  atomicAdd(&sums[1], sw);
}

__global__ void checksum_kernel(const uint64_t *k, const uint64_t *w, int64_t n, unsigned long long *sums) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long sk = 0, sw = 0;
  for (; i < n; i += stride) {
    sk += k[i];
    sw += w[i] * 0x9E3779B97F4A7C15ULL ^ k[i];
  }
  atomicAdd(&sums[0], sk);
  atomicAdd(&sums[1], sw);
}

template <int BITS>
int run(int lg, int iters) {
  constexpr int P = 1 << BITS;
  const int64_t n = int64_t(1) << lg;
  uint64_t *k, *w;
  ulonglong2 *out;
  CK(hipMalloc(&k, n * 8));
  CK(hipMalloc(&w, n * 8));
  const int64_t out_rows = n + 8 * P;
  CK(hipMalloc(&out, out_rows * 16));
  gen_kernel<<<(n + 255) / 256, 256>>>(k, w, n);
  CK(hipDeviceSynchronize());
  int nchunks = 512;
  int64_t chunk_rows = (n + nchunks - 1) / nchunks;
  chunk_rows = (chunk_rows + 8191) / 8192 * 8192;
  nchunks = int((n + chunk_rows - 1) / chunk_rows);
  uint32_t *hist, *totals, *starts;
  CK(hipMalloc(&hist, (int64_t)nchunks * P * 4));
  CK(hipMalloc(&totals, P * 4));
  CK(hipMalloc(&starts, P * 4));
  unsigned long long *sums;
  unsigned int *bad;
  CK(hipMalloc(&sums, 32));
  CK(hipMalloc(&bad, 4));
  hipEvent_t e[8];
  for (auto &x : e) CK(hipEventCreate(&x));
  printf("rows 2^%d, partitions %d, chunks %d x %lld rows\n", lg, P, nchunks, (long long)chunk_rows);
  for (int variant = 0; variant < 4; variant++) {
    for (int it = 0; it < iters; it++) {
      CK(hipMemsetAsync(out, 0xFF, out_rows * 16));
      CK(hipEventRecord(e[0]));
      hist_kernel<BITS, 1024><<<nchunks, 1024>>>(k, n, chunk_rows, hist);
      CK(hipEventRecord(e[1]));
      colsum_kernel<BITS><<<P / 64, 1024>>>(hist, nchunks, totals);
      starts_kernel<BITS><<<1, 1024>>>(totals, starts);
      offs_kernel<BITS><<<P / 64, 1024>>>(hist, nchunks, starts);
      CK(hipEventRecord(e[2]));
      if (variant == 0)
        scatter_carry_kernel<BITS, 1024, 4><<<nchunks, 1024>>>(k, w, n, chunk_rows, hist, out);
      else if (variant == 2) {
        if constexpr (BITS <= 10) scatter_bins_kernel<BITS, 1024, 4><<<nchunks, 1024>>>(k, w, n, chunk_rows, hist, out);
      }
      else if (variant == 3) {
        if constexpr (BITS <= 10) scatter_sorted_kernel<BITS, 1024, 6><<<nchunks, 1024>>>(k, w, n, chunk_rows, hist, out);
      }
      else
        scatter_direct_kernel<BITS, 1024, 4><<<nchunks, 1024>>>(k, w, n, chunk_rows, hist, out);
      CK(hipEventRecord(e[3]));
      CK(hipEventSynchronize(e[3]));
      float th, ts, tc;
      CK(hipEventElapsedTime(&th, e[0], e[1]));
      CK(hipEventElapsedTime(&ts, e[1], e[2]));
      CK(hipEventElapsedTime(&tc, e[2], e[3]));
      printf("  %-7s hist %.3f ms (%.0f GB/s)  scan %.3f ms  scatter %.3f ms (%.0f GB/s r+w)\n",
             variant == 0 ? "carry4" : variant == 1 ? "direct" : variant == 2 ? "bins4" : "sorted6", th, n * 8 / th / 1e6, ts, tc, n * 32 / tc / 1e6);
    }
    CK(hipMemset(sums, 0, 32));
    CK(hipMemset(bad, 0, 4));
    checksum_kernel<<<1024, 256>>>(k, w, n, sums);
    verify_kernel<BITS><<<P, 256>>>(out, starts, totals, sums + 2, bad);
    unsigned long long hs[4];
    unsigned int hb;
    CK(hipMemcpy(hs, sums, 32, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    const bool ok = hb == 0 && hs[0] == hs[2] && hs[1] == hs[3];
    printf("  verify %s: misplaced %u, checksums %s\n", ok ? "OK" : "FAILED", hb,
           (hs[0] == hs[2] && hs[1] == hs[3]) ? "match" : "DIFFER");
    if (!ok) return 1;
  }
  // reference: interleaving copy (same bytes, no scatter)
  for (int it = 0; it < 2; it++) {
    CK(hipEventRecord(e[0]));
    copy_kernel<<<4096, 256>>>((const ulonglong2 *)k, (const ulonglong2 *)w, out, n / 2);
    CK(hipEventRecord(e[1]));
    CK(hipEventSynchronize(e[1]));
    float t;
    CK(hipEventElapsedTime(&t, e[0], e[1]));
    printf("  interleaving copy %.3f ms (%.0f GB/s r+w)\n", t, n * 32 / t / 1e6);
  }
  return 0;
}

int main(int argc, char **argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 29;
  const int bits = argc > 2 ? atoi(argv[2]) : 11;
  const int iters = argc > 3 ? atoi(argv[3]) : 3;
  switch (bits) {
    case 9: return run<9>(lg, iters);
    case 10: return run<10>(lg, iters);
    case 11: return run<11>(lg, iters);
    default: fprintf(stderr, "bits must be 9..11\n"); return 2;
  }
}
