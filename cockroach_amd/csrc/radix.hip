// radix.hip -- radix-partitioned hash join (see radix.h for the design).
//
// Replaces, for large unique-key build sides, the same reference code as join.hip's
// packed table: HashTable.FullBuild / DistinctCheck / distinctCollect*
// (colexechash/hashtable.go:477,921; colexecjoin/hashjoiner_tmpl.go:16-337).
//
// Kernels and their bound (all HBM streaming; measured on MI355X, 2^29 rows of 16 B):
//   radix_hist_kernel      reads the key column once                       ~5.5 TB/s
//   radix_scatter_kernel   streams (key, payload) in, full 128-B lines out ~4.7 TB/s (copy: 5.0)
//   radix_split_kernel     per partition: refine into <=64 sub-partitions
//   radix_table_kernel     per sub-partition: open-addressing table built in LDS, written out coalesced
//   radix_probe_kernel     partition-at-a-time probe out of the XCD's L2
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "device.h"
#include "radix.h"

namespace crgpu {

typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t rpart(uint64_t h) { return static_cast<uint32_t>(h >> 54); }
__device__ __forceinline__ uint32_t rsub(uint64_t h, uint32_t smask) { return static_cast<uint32_t>(h >> 47) & smask; }
// Sub-partition among ANY number of them (not only a power of two), from 32 hash bits below the partition's: the
// aggregation's regions, which nothing probes by hash afterwards, so their number follows the row count
__device__ __forceinline__ uint32_t rsub_any(uint64_t h, uint32_t nsub) { return __umulhi(static_cast<uint32_t>(h >> 16), nsub); }
// The three candidate slots of a key inside its sub-partition's table (3-way cuckoo hashing):
// 32-bit hash fields scaled to [0, kRadixSlots).
__device__ __forceinline__ uint32_t rscale(uint32_t x) {
  return static_cast<uint32_t>((static_cast<uint64_t>(x) * kRadixSlots) >> 32);
}
__device__ __forceinline__ void rslots(uint64_t h, uint32_t *s) {
  const uint64_t g = fmix64(h + 0x9E3779B97F4A7C15ULL);
  s[0] = rscale(static_cast<uint32_t>(h));
  s[1] = rscale(static_cast<uint32_t>(g));
  s[2] = rscale(static_cast<uint32_t>(g >> 32));
}

// Key word -> the 64-bit value whose bit fields address the tables.  Hashed tables: fmix64.  Direct-addressed tables
// (radix.h, RadixDirect): the key word x = key - lo itself, laid out so that rpart / rsub read the same fields --
// [63:54] = x >> (kRadixDirectBits + sub_bits) (< 1024 by the span limit), [47 + sub_bits - 1 : 47] = the next
// sub_bits bits of x, [kRadixDirectBits - 1 : 0] = the slot.
template <bool DIRECT>
__device__ __forceinline__ uint64_t rhash(uint64_t k, uint32_t sub_bits) {
  if (!DIRECT) return fmix64(k);
  const uint64_t hi = k >> kRadixDirectBits;
  return ((hi >> sub_bits) << 54) | ((hi & ((1ull << sub_bits) - 1ull)) << 47) | (k & ((1ull << kRadixDirectBits) - 1ull));
}

__device__ __forceinline__ uint64_t rload(const uint8_t *src, int64_t i, int width) {
  switch (width) {
    case 1: return src[i];
    case 2: return reinterpret_cast<const uint16_t *>(src)[i];
    case 4: return reinterpret_cast<const uint32_t *>(src)[i];
    default: return reinterpret_cast<const uint64_t *>(src)[i];
  }
}
// output columns are written once and never re-read by the kernel: non-temporal, so that
// they do not displace the partition's table from L2
__device__ __forceinline__ void rstore(uint8_t *dst, int64_t pos, int width, uint64_t v) {
  switch (width) {
    case 1: __builtin_nontemporal_store(static_cast<uint8_t>(v), dst + pos); break;
    case 2: __builtin_nontemporal_store(static_cast<uint16_t>(v), reinterpret_cast<uint16_t *>(dst) + pos); break;
    case 4: __builtin_nontemporal_store(static_cast<uint32_t>(v), reinterpret_cast<uint32_t *>(dst) + pos); break;
    default: __builtin_nontemporal_store(v, reinterpret_cast<uint64_t *>(dst) + pos); break;
  }
}

// rows r0, r0+1 (r0 even) of one join side as {packed key, payload}
// FAST: 0 generic loader; 1 vector loads of key and payload; 2 vector load of the key, the payload is the row id (a
// compile-time form of its own: a run-time test of s.pay inside form 1 cost config J's probe-side scatter 0.2 ms)
// DIRECT: the key word becomes x = key - lo; a key outside the build side's range has no key (ok = false)
template <int FAST, bool DIRECT = false>
__device__ __forceinline__ void radix_load2(const RadixSrc &s, int64_t r0, int64_t hi, uint64_t *k, uint64_t *v,
                                            bool *ok) {
  if (FAST != 0 && r0 + 1 < hi) {
    const ulonglong2 a = reinterpret_cast<const ulonglong2 *>(s.kp.c[0].data)[r0 >> 1];
    k[0] = a.x, k[1] = a.y;
    if (FAST == 1) {
      const ulonglong2 b = reinterpret_cast<const ulonglong2 *>(s.pay)[r0 >> 1];
      v[0] = b.x, v[1] = b.y;
    } else {
      v[0] = static_cast<uint64_t>(r0), v[1] = static_cast<uint64_t>(r0 + 1);
    }
    ok[0] = ok[1] = true;
  } else {
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int64_t r = r0 + t;
      k[t] = 0, v[t] = 0;
      ok[t] = r < hi && key_pack(s.kp, r, k[t]);
      if (ok[t]) v[t] = s.pay ? rload(s.pay, r, s.pay_width) : static_cast<uint64_t>(r);
    }
  }
  if (DIRECT) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
      k[t] -= s.direct.lo;  // (wraps far above the span when the key is below lo)
      ok[t] = ok[t] && k[t] < s.direct.span;
    }
  }
}
// one row's key word (the probe kernel's unpartitioned modes)
template <bool DIRECT>
__device__ __forceinline__ bool radix_key(const RadixSrc &s, int64_t r, uint64_t &k) {
  bool ok = key_pack(s.kp, r, k);
  if (DIRECT) {
    k -= s.direct.lo;
    ok = ok && k < s.direct.span;
  }
  return ok;
}

// ------------------------------------------------------------------ level 1: histogram per chunk
template <bool FAST, bool DIRECT = false>
__global__ __launch_bounds__(1024) void radix_hist_kernel(RadixSrc src, int64_t n, int64_t chunk_rows,
                                                          uint32_t *__restrict__ hist) {
  __shared__ uint32_t h[kRadixP1];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t lo = blockIdx.x * chunk_rows;
  int64_t hi = lo + chunk_rows;
  if (hi > n) hi = n;
  for (int64_t r0 = lo + 2 * static_cast<int64_t>(threadIdx.x); r0 < hi; r0 += 2048) {
    uint64_t kk[2], vv[2];
    bool ok[2];
    radix_load2<FAST ? 2 : 0, DIRECT>(src, r0, hi, kk, vv, ok);  // (form 2: the payload is not loaded)
#pragma unroll
    for (int t = 0; t < 2; t++)
      if (ok[t]) atomicAdd(&h[rpart(rhash<DIRECT>(kk[t], src.direct.sub_bits))], 1u);
  }
  __syncthreads();
  hist[static_cast<int64_t>(blockIdx.x) * kRadixP1 + threadIdx.x] = h[threadIdx.x];
}

// ------------------------------------------------------------------ level 1: offsets
// totals[p] = rows of partition p; starts[p] = first row of partition p (8-row aligned);
// hist[c][p] becomes the first row chunk c writes for partition p.
// One block of 1024 threads per 64 partitions: thread (g = tid / 64, px = tid % 64) owns chunk group g.
__global__ __launch_bounds__(1024) void radix_colsum_kernel(const uint32_t *__restrict__ hist, int nchunks,
                                                            uint32_t *__restrict__ totals) {
  const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + px;
  __shared__ uint32_t s[16][64];
  const int per = (nchunks + 15) / 16;
  uint32_t sum = 0;
  for (int c = g * per; c < (g + 1) * per && c < nchunks; c++) sum += hist[static_cast<int64_t>(c) * kRadixP1 + p];
  s[g][px] = sum;
  __syncthreads();
  if (g == 0) {
    uint32_t t = 0;
    for (int i = 0; i < 16; i++) t += s[i][px];
    totals[p] = t;
  }
}

__global__ __launch_bounds__(1024) void radix_starts_kernel(const uint32_t *__restrict__ totals,
                                                            uint32_t *__restrict__ starts) {
  __shared__ uint32_t wsum[16];
  const uint32_t padded = (totals[threadIdx.x] + 7u) & ~7u;
  const uint32_t incl = wave_incl_scan_u32(padded);
  if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < static_cast<int>(threadIdx.x >> 6); w++) base += wsum[w];
  starts[threadIdx.x] = base + incl - padded;
}

__global__ __launch_bounds__(1024) void radix_offs_kernel(uint32_t *__restrict__ hist, int nchunks,
                                                          const uint32_t *__restrict__ starts) {
  const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + px;
  __shared__ uint32_t s[16][64];
  const int per = (nchunks + 15) / 16;
  uint32_t sum = 0;
  for (int c = g * per; c < (g + 1) * per && c < nchunks; c++) sum += hist[static_cast<int64_t>(c) * kRadixP1 + p];
  s[g][px] = sum;
  __syncthreads();
  uint32_t base = starts[p];
  for (int i = 0; i < g; i++) base += s[i][px];
  for (int c = g * per; c < (g + 1) * per && c < nchunks; c++) {
    const uint32_t v = hist[static_cast<int64_t>(c) * kRadixP1 + p];
    hist[static_cast<int64_t>(c) * kRadixP1 + p] = base;
    base += v;
  }
}

// ------------------------------------------------------------------ level 1: scatter
// Per-partition 128-byte line buffers in LDS (1024 x 128 B): rows are ranked with LDS
// atomics; a partition's first pending line is completed in its buffer and written by 8
// adjacent lanes, further complete lines of the same tile go out straight from registers,
// the remainder waits in the buffer.  Every HBM write is therefore a whole line (random
// 64-byte writes run at 3.4 TB/s, 16-byte ones at 2.0, whole lines at 4.7).
// CELLS: no histogram pass ran.  Chunk c writes its rows of partition p into the fixed-capacity cell (p, c) of
// `cell_cap` rows (a multiple of 8: every cell starts on a line) and reports how many it held in cell_cnt[p][c]; rows
// beyond the capacity are dropped and the flag is raised (skewed keys: the caller repeats the pass with offsets from a
// histogram).  For hashed keys a cell holds chunk_rows / 1024 rows on average with a standard deviation of its square
// root, so a capacity eight deviations above the mean practically never overflows.
template <int FAST, bool CELLS, bool DIRECT = false>
__global__ __launch_bounds__(1024) void radix_scatter_kernel(RadixSrc src, int64_t n, int64_t chunk_rows,
                                                             const uint32_t *__restrict__ offs,
                                                             ulonglong2 *__restrict__ out, uint32_t cell_cap,
                                                             uint32_t *__restrict__ cell_cnt, uint32_t *__restrict__ flag) {
  constexpr int P = kRadixP1, THREADS = 1024, ROWS = 4, TILE = THREADS * ROWS;
  __shared__ ulonglong2 bins[P * 8];
  __shared__ uint32_t cnt[P];    // rows pending in the buffer (incl. phantom slots before the first flush)
  __shared__ uint32_t gbase[P];  // line-aligned absolute row position of buffer slot 0
  __shared__ uint8_t first[P];   // first real slot: chunk-start misalignment, 0 after the first flush
  // CELLS: first row position past the cell of partition p
#define RADIX_CELL_END(p) ((static_cast<uint32_t>(p) * gridDim.x + blockIdx.x + 1u) * cell_cap)
  if (CELLS) {
    gbase[threadIdx.x] = (threadIdx.x * gridDim.x + blockIdx.x) * cell_cap;
    cnt[threadIdx.x] = 0;
    first[threadIdx.x] = 0;
  } else {
    const uint32_t g0 = offs[static_cast<int64_t>(blockIdx.x) * P + threadIdx.x];
    const uint32_t a = g0 & 7u;
    gbase[threadIdx.x] = g0 - a;
    cnt[threadIdx.x] = a;
    first[threadIdx.x] = static_cast<uint8_t>(a);
  }
  __syncthreads();
  const int64_t lo = blockIdx.x * chunk_rows;
  int64_t hi = lo + chunk_rows;
  if (hi > n) hi = n;
  uint64_t nk[ROWS], nv[ROWS];
  bool nok[ROWS];
#pragma unroll
  for (int j = 0; j < ROWS / 2; j++)
    radix_load2<FAST, DIRECT>(src, lo + static_cast<int64_t>(j * THREADS + threadIdx.x) * 2, hi, nk + 2 * j, nv + 2 * j, nok + 2 * j);
  for (int64_t t0 = lo; t0 < hi; t0 += TILE) {
    uint64_t kk[ROWS], vv[ROWS];
    uint32_t pp[ROWS], rk[ROWS], nfl[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      kk[j] = nk[j], vv[j] = nv[j];
      pp[j] = nok[j] ? rpart(rhash<DIRECT>(kk[j], src.direct.sub_bits)) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int j = 0; j < ROWS / 2; j++)  // prefetch the next tile while this one is ranked and flushed
      radix_load2<FAST, DIRECT>(src, t0 + TILE + static_cast<int64_t>(j * THREADS + threadIdx.x) * 2, hi, nk + 2 * j, nv + 2 * j,
                        nok + 2 * j);
    // A: rank among the partition's pending rows
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      if (pp[j] != 0xFFFFFFFFu) rk[j] = atomicAdd(&cnt[pp[j]], 1u);
    __syncthreads();
    // B: rows of the first line join the buffer; rows of further complete lines go straight to HBM
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      if (pp[j] != 0xFFFFFFFFu) {
        nfl[j] = cnt[pp[j]] & ~7u;
        if (rk[j] < 8u) bins[pp[j] * 8 + rk[j]] = make_ulonglong2(kk[j], vv[j]);
        else if (rk[j] < nfl[j] && (!CELLS || gbase[pp[j]] + rk[j] < RADIX_CELL_END(pp[j])))
          out[gbase[pp[j]] + rk[j]] = make_ulonglong2(kk[j], vv[j]);
      }
    }
    __syncthreads();
    // C: flush every complete first line, 8 lanes per line
    for (int i = threadIdx.x; i < P * 8; i += THREADS) {
      const int p = i >> 3, sl = i & 7;
      if (cnt[p] >= 8u && sl >= static_cast<int>(first[p]) && (!CELLS || gbase[p] < RADIX_CELL_END(p))) out[gbase[p] + sl] = bins[i];
    }
    __syncthreads();
    // D: rows past the last complete line are the new buffer content
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      if (pp[j] != 0xFFFFFFFFu && rk[j] >= 8u && rk[j] >= nfl[j])
        bins[pp[j] * 8 + (rk[j] - nfl[j])] = make_ulonglong2(kk[j], vv[j]);
    {
      const uint32_t t = cnt[threadIdx.x], f = t & ~7u;
      if (f) {
        gbase[threadIdx.x] += f;
        cnt[threadIdx.x] = t - f;
        first[threadIdx.x] = 0;
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < P * 8; i += THREADS) {
    const int p = i >> 3, sl = i & 7;
    if (sl >= static_cast<int>(first[p]) && sl < static_cast<int>(cnt[p]) && (!CELLS || gbase[p] < RADIX_CELL_END(p)))
      out[gbase[p] + sl] = bins[i];
  }
  if (CELLS) {
    const uint32_t held = gbase[threadIdx.x] - (threadIdx.x * gridDim.x + blockIdx.x) * cell_cap + cnt[threadIdx.x];
    cell_cnt[threadIdx.x * gridDim.x + blockIdx.x] = held < cell_cap ? held : cell_cap;
    if (held > cell_cap) atomicOr(flag, 1u);
  }
#undef RADIX_CELL_END
}

// Before the cells form is chosen: the partition histogram of kSkewSample rows taken at a fixed stride.  A partition
// that draws more than kSkewLimit of them (64 expected, standard deviation 8) holds well over its share of the rows
// -- a hot key -- and would overflow its cells; the probe then partitions with the full histogram instead.
static constexpr int kSkewSample = 65536, kSkewLimit = 100;
template <bool FAST, bool DIRECT = false>
__global__ __launch_bounds__(1024) void radix_skew_sample_kernel(RadixSrc src, int64_t n, int64_t stride,
                                                                 uint32_t *__restrict__ hist) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * 1024 + threadIdx.x) * stride;
  if (i >= n) return;
  uint64_t kk = 0;
  bool ok = true;
  if (FAST) kk = reinterpret_cast<const uint64_t *>(src.kp.c[0].data)[i];
  else ok = key_pack(src.kp, i, kk);
  if (DIRECT) {
    kk -= src.direct.lo;
    ok = ok && kk < src.direct.span;
  }
  if (ok) atomicAdd(&hist[rpart(rhash<DIRECT>(kk, src.direct.sub_bits))], 1u);
}
__global__ __launch_bounds__(1024) void radix_skew_check_kernel(const uint32_t *__restrict__ hist, uint32_t *__restrict__ flag) {
  if (hist[threadIdx.x] > static_cast<uint32_t>(kSkewLimit)) atomicOr(flag, 1u);
}

// cell counts -> per partition: pref[p][c] = rows of the cells before c (pref[p][nchunks] = rows of the partition)
__global__ __launch_bounds__(512) void radix_cell_prefix_kernel(const uint32_t *__restrict__ cell_cnt, uint32_t nchunks,
                                                                uint32_t *__restrict__ pref, uint32_t *__restrict__ totals) {
  __shared__ uint32_t wsum[8];
  const uint32_t p = blockIdx.x, c = threadIdx.x;
  const uint32_t v = c < nchunks ? cell_cnt[p * nchunks + c] : 0;
  const uint32_t incl = wave_incl_scan_u32(v);
  if (lane_id() == 63) wsum[c >> 6] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (uint32_t w = 0; w < 8; w++) {
    if (w < (c >> 6)) base += wsum[w];
    total += wsum[w];
  }
  if (c < nchunks) pref[p * (nchunks + 1) + c] = base + incl - v;
  if (c == 0) {
    pref[p * (nchunks + 1) + nchunks] = total;
    totals[p] = total;
  }
}

// ------------------------------------------------------------------ level 2 (build side): refine one partition
// One workgroup per level-1 partition, ONE pass: sub-partition (p, s) owns a fixed region of
// kRadixSubCap rows, so no counting pass is needed; each 4096-row tile is counting-sorted by
// sub-partition in LDS and written out as runs (~64 rows = 1 KiB per run, coalesced).
// sub_count[p][s] = rows written; a region that would overflow raises the flag (the caller
// falls back to the classic table) and the excess rows are dropped.
template <bool DIRECT, bool ANY = false>
__global__ __launch_bounds__(1024) void radix_split_kernel(const ulonglong2 *__restrict__ in,
                                                           const uint32_t *__restrict__ starts,
                                                           const uint32_t *__restrict__ totals, uint32_t smask,
                                                           ulonglong2 *__restrict__ out, uint32_t *__restrict__ sub_count,
                                                           uint32_t *__restrict__ overflow, uint32_t subcap) {
  constexpr int THREADS = 1024, ROWS = 4, TILE = THREADS * ROWS;
  const int p = blockIdx.x;
  const uint32_t nsub = smask + 1;
  __shared__ ulonglong2 stage[TILE];
  __shared__ uint32_t tcnt[128], tstart[128], cur[128];  // up to 128 sub-partitions (the join uses <= 64)
  __shared__ uint32_t s_half;
  if (threadIdx.x < 128) tcnt[threadIdx.x] = 0, cur[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t s0 = starts[p], n = totals[p];
  ulonglong2 *obase = out + static_cast<int64_t>(p) * nsub * subcap;
  for (uint32_t t0 = 0; t0 < n; t0 += TILE) {
    ulonglong2 row[ROWS];
    uint32_t sp[ROWS], rk[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      const uint32_t i = t0 + j * THREADS + threadIdx.x;
      sp[j] = 0xFFFFFFFFu;
      if (i < n) {
        row[j] = in[s0 + i];
        sp[j] = DIRECT ? static_cast<uint32_t>(row[j].x >> kRadixDirectBits) & smask
                       : (ANY ? rsub_any(fmix64(row[j].x), nsub) : rsub(fmix64(row[j].x), smask));
        rk[j] = atomicAdd(&tcnt[sp[j]], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 128) {  // two waves scan 64 counters each; the second half is shifted by the first half's total
      const uint32_t c = tcnt[threadIdx.x];
      const uint32_t incl = wave_incl_scan_u32(c);
      tstart[threadIdx.x] = incl - c;
      if (threadIdx.x == 63) s_half = incl;
    }
    __syncthreads();
    if (threadIdx.x >= 64 && threadIdx.x < 128) tstart[threadIdx.x] += s_half;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      if (sp[j] != 0xFFFFFFFFu) stage[tstart[sp[j]] + rk[j]] = row[j];
    __syncthreads();
    const uint32_t tile_rows = n - t0 < static_cast<uint32_t>(TILE) ? n - t0 : TILE;
    for (uint32_t i = threadIdx.x; i < tile_rows; i += THREADS) {
      const ulonglong2 r = stage[i];
      const uint32_t s = DIRECT ? static_cast<uint32_t>(r.x >> kRadixDirectBits) & smask
                                : (ANY ? rsub_any(fmix64(r.x), nsub) : rsub(fmix64(r.x), smask));
      const uint32_t pos = cur[s] + (i - tstart[s]);
      if (pos < subcap) obase[static_cast<int64_t>(s) * subcap + pos] = r;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      cur[threadIdx.x] += tcnt[threadIdx.x];
      tcnt[threadIdx.x] = 0;
    }
    __syncthreads();
  }
  if (threadIdx.x < nsub) {
    const uint32_t c = cur[threadIdx.x];
    sub_count[p * nsub + threadIdx.x] = c;
    if (c > subcap) atomicOr(overflow, 1u);
  }
}

// ------------------------------------------------------------------ wide rows: <= 16-byte BYTES keys (hash aggregation)
// The same two partition levels for 32-byte rows {k0, k1, value, length}: the key's bytes little-endian in two words,
// zero-padded, so two keys are equal exactly when (k0, k1, length) are.  A 128-byte line holds four rows.  Keys longer
// than 16 bytes raise bit 1 of the flag (the caller abandons the attempt); they are skipped by both passes alike.
__device__ __forceinline__ bool wide_load(const WideSrc &s, int64_t r, ulonglong4 *row) {
  if (s.val2 != nullptr) {  // one integer key column, two value columns
    *row = make_ulonglong4(static_cast<uint64_t>(col_get_int(s.key, r)), 0ull, s.val != nullptr ? s.val[r] : 0ull, s.val2[r]);
    return true;
  }
  if (s.key.type != CRGPU_BYTES) {  // two integer key columns
    *row = make_ulonglong4(static_cast<uint64_t>(col_get_int(s.key, r)), static_cast<uint64_t>(col_get_int(s.key2, r)),
                           s.val != nullptr ? s.val[r] : 0ull, 16ull);
    return true;
  }
  const uint32_t a0 = s.key.offsets[r], len = s.key.offsets[r + 1] - a0;
  const uint8_t *p = static_cast<const uint8_t *>(s.key.data) + a0;
  uint64_t k0 = 0, k1 = 0;
  if (len > 16u) return false;
  if (len >= 8u) {  // unaligned 8-byte loads inside the value
    __builtin_memcpy(&k0, p, 8);
    if (len > 8u) {
      uint64_t t;
      __builtin_memcpy(&t, p + len - 8, 8);
      k1 = t >> (8u * (16u - len));
    }
  } else {
    for (uint32_t j = 0; j < len; j++) k0 |= static_cast<uint64_t>(p[j]) << (8u * j);
  }
  *row = make_ulonglong4(k0, k1, s.val != nullptr ? s.val[r] : 0ull, len);
  return true;
}

__global__ __launch_bounds__(1024) void wide_hist_kernel(WideSrc src, int64_t n, int64_t chunk_rows,
                                                         uint32_t *__restrict__ hist, uint32_t *__restrict__ flag) {
  __shared__ uint32_t h[kRadixP1];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t lo = blockIdx.x * chunk_rows;
  int64_t hi = lo + chunk_rows;
  if (hi > n) hi = n;
  bool bad = false;
  const uint64_t wmask = src.val2 != nullptr ? 0ull : ~0ull;
  for (int64_t r0 = lo + threadIdx.x; r0 < hi; r0 += 4096) {
    ulonglong4 row[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t r = r0 + j * 1024;
      ok[j] = false;
      if (r < hi) {
        ok[j] = wide_load(src, r, &row[j]);
        bad |= !ok[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (ok[j]) atomicAdd(&h[rpart(wide_key_hash(row[j].x, row[j].y, row[j].w & wmask))], 1u);
  }
  if (bad) atomicOr(flag, 2u);
  __syncthreads();
  hist[static_cast<int64_t>(blockIdx.x) * kRadixP1 + threadIdx.x] = h[threadIdx.x];
}

// radix_scatter_kernel for 32-byte rows: four rows per line buffer, otherwise the same four phases.
__global__ __launch_bounds__(1024) void wide_scatter_kernel(WideSrc src, int64_t n, int64_t chunk_rows,
                                                            const uint32_t *__restrict__ offs, ulonglong4 *__restrict__ out) {
  constexpr int P = kRadixP1, THREADS = 1024, ROWS = 4, TILE = THREADS * ROWS, RPL = 4;
  __shared__ ulonglong4 bins[P * RPL];
  __shared__ uint32_t cnt[P];
  __shared__ uint32_t gbase[P];
  __shared__ uint8_t first[P];
  {
    const uint32_t g0 = offs[static_cast<int64_t>(blockIdx.x) * P + threadIdx.x];
    const uint32_t a = g0 & (RPL - 1u);
    gbase[threadIdx.x] = g0 - a;
    cnt[threadIdx.x] = a;
    first[threadIdx.x] = static_cast<uint8_t>(a);
  }
  __syncthreads();
  const int64_t lo = blockIdx.x * chunk_rows;
  int64_t hi = lo + chunk_rows;
  if (hi > n) hi = n;
  ulonglong4 nrow[ROWS];
  bool nok[ROWS];
  const uint64_t wmask = src.val2 != nullptr ? 0ull : ~0ull;
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int64_t r = lo + j * THREADS + threadIdx.x;
    nok[j] = r < hi && wide_load(src, r, &nrow[j]);
  }
  for (int64_t t0 = lo; t0 < hi; t0 += TILE) {
    ulonglong4 row[ROWS];
    uint32_t pp[ROWS], rk[ROWS], nfl[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      row[j] = nrow[j];
      pp[j] = nok[j] ? rpart(wide_key_hash(row[j].x, row[j].y, row[j].w & wmask)) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int j = 0; j < ROWS; j++) {  // the next tile is requested while this one is ranked and flushed
      const int64_t r = t0 + TILE + j * THREADS + threadIdx.x;
      nok[j] = r < hi && wide_load(src, r, &nrow[j]);
    }
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      if (pp[j] != 0xFFFFFFFFu) rk[j] = atomicAdd(&cnt[pp[j]], 1u);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      if (pp[j] != 0xFFFFFFFFu) {
        nfl[j] = cnt[pp[j]] & ~(RPL - 1u);
        if (rk[j] < RPL) bins[pp[j] * RPL + rk[j]] = row[j];
        else if (rk[j] < nfl[j]) out[gbase[pp[j]] + rk[j]] = row[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P * RPL; i += THREADS) {
      const int p = i / RPL, sl = i % RPL;
      if (cnt[p] >= RPL && sl >= static_cast<int>(first[p])) out[gbase[p] + sl] = bins[i];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      if (pp[j] != 0xFFFFFFFFu && rk[j] >= RPL && rk[j] >= nfl[j]) bins[pp[j] * RPL + (rk[j] - nfl[j])] = row[j];
    {
      const uint32_t t = cnt[threadIdx.x], f = t & ~(RPL - 1u);
      if (f) {
        gbase[threadIdx.x] += f;
        cnt[threadIdx.x] = t - f;
        first[threadIdx.x] = 0;
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < P * RPL; i += THREADS) {
    const int p = i / RPL, sl = i % RPL;
    if (sl >= static_cast<int>(first[p]) && sl < static_cast<int>(cnt[p])) out[gbase[p] + sl] = bins[i];
  }
}

// radix_split_kernel for 32-byte rows (tiles of 2048 rows: the same 64 KiB of staging)
template <bool ANY>
__global__ __launch_bounds__(1024) void wide_split_kernel(const ulonglong4 *__restrict__ in, const uint32_t *__restrict__ starts,
                                                          const uint32_t *__restrict__ totals, uint32_t smask,
                                                          ulonglong4 *__restrict__ out, uint32_t *__restrict__ sub_count,
                                                          uint32_t *__restrict__ overflow, uint32_t subcap, uint64_t wmask) {
  constexpr int THREADS = 1024, ROWS = 2, TILE = THREADS * ROWS;
  const int p = blockIdx.x;
  const uint32_t nsub = smask + 1;
  __shared__ ulonglong4 stage[TILE];
  __shared__ uint32_t tcnt[128], tstart[128], cur[128];
  __shared__ uint32_t s_half;
  if (threadIdx.x < 128) tcnt[threadIdx.x] = 0, cur[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t s0 = starts[p], n = totals[p];
  ulonglong4 *obase = out + static_cast<int64_t>(p) * nsub * subcap;
  for (uint32_t t0 = 0; t0 < n; t0 += TILE) {
    ulonglong4 row[ROWS];
    uint32_t sp[ROWS], rk[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      const uint32_t i = t0 + j * THREADS + threadIdx.x;
      sp[j] = 0xFFFFFFFFu;
      if (i < n) {
        row[j] = in[s0 + i];
        sp[j] = ANY ? rsub_any(wide_key_hash(row[j].x, row[j].y, row[j].w & wmask), nsub)
                    : rsub(wide_key_hash(row[j].x, row[j].y, row[j].w & wmask), smask);
        rk[j] = atomicAdd(&tcnt[sp[j]], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const uint32_t c = tcnt[threadIdx.x];
      const uint32_t incl = wave_incl_scan_u32(c);
      tstart[threadIdx.x] = incl - c;
      if (threadIdx.x == 63) s_half = incl;
    }
    __syncthreads();
    if (threadIdx.x >= 64 && threadIdx.x < 128) tstart[threadIdx.x] += s_half;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ROWS; j++)
      if (sp[j] != 0xFFFFFFFFu) stage[tstart[sp[j]] + rk[j]] = row[j];
    __syncthreads();
    const uint32_t tile_rows = n - t0 < static_cast<uint32_t>(TILE) ? n - t0 : TILE;
    for (uint32_t i = threadIdx.x; i < tile_rows; i += THREADS) {
      const ulonglong4 r = stage[i];
      const uint32_t s = ANY ? rsub_any(wide_key_hash(r.x, r.y, r.w & wmask), nsub) : rsub(wide_key_hash(r.x, r.y, r.w & wmask), smask);
      const uint32_t pos = cur[s] + (i - tstart[s]);
      if (pos < subcap) obase[static_cast<int64_t>(s) * subcap + pos] = r;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      cur[threadIdx.x] += tcnt[threadIdx.x];
      tcnt[threadIdx.x] = 0;
    }
    __syncthreads();
  }
  if (threadIdx.x < nsub) {
    const uint32_t c = cur[threadIdx.x];
    sub_count[p * nsub + threadIdx.x] = c;
    if (c > subcap) atomicOr(overflow, 1u);
  }
}

// ------------------------------------------------------------------ level 3 (build side): one table per sub-partition
// A divergent 16-byte load costs ~2 cycles per ACTIVE lane on a CU (measured: 270 G lane-loads/s
// chip-wide, L1 or L2 hit alike), and a wave keeps looping until its slowest lane is done, so a
// table is judged by loads per lookup and by the LENGTH OF THE TAIL.  Linear probing (mean 2,
// tail 16+) and line-local buckets (mean 3.2) both measured 3-5x slower than this layout:
// 3-way cuckoo hashing, one 16-byte slot per candidate, built in LDS by one workgroup per
// sub-partition (<= 2560 rows into 3072 slots; load threshold of 3-way cuckoo is 0.91).
// A lookup reads at most three slots (1.6 on average for a hit; an EMPTY slot ends a miss early
// because occupied slots never become empty again).  The 48 KiB image is written out coalesced.
// CHECK_DUP: after the build every row looks its own key up; a row that finds ANOTHER row of its key first (a dropped
// or a second entry of a duplicate key) raises bit 8 of `failed` -- the fused group-join needs unique build keys and
// must not count a probe row once where the join emits it twice.
template <bool CHECK_DUP>
__global__ __launch_bounds__(512) void radix_table_kernel(const ulonglong2 *__restrict__ rows,
                                                          const uint32_t *__restrict__ sub_count,
                                                          ulonglong2 *__restrict__ table,
                                                          RadixSpecial *__restrict__ special, uint32_t *__restrict__ failed) {
  constexpr int THREADS = 512;
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  __shared__ ulonglong2 srow[kRadixSubCap];
  __shared__ uint32_t tab[kRadixSlots];
  const int q = blockIdx.x;
  const int64_t base = static_cast<int64_t>(q) * kRadixSubCap;
  uint32_t n = sub_count[q];
  if (n > static_cast<uint32_t>(kRadixSubCap)) n = kRadixSubCap;  // overflow is reported by the split kernel
  for (uint32_t i = threadIdx.x; i < n; i += THREADS) srow[i] = rows[base + i];
  for (int i = threadIdx.x; i < kRadixSlots; i += THREADS) tab[i] = kNone;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
    if (srow[i].x == kRadixEmpty) {  // the one key equal to the empty marker lives beside the table
      special->val = srow[i].y;
      if (CHECK_DUP) {
        if (atomicExch(&special->present, 1u) != 0u) atomicOr(failed, 8u);
      } else {
        special->present = 1;
      }
      continue;
    }
    uint32_t cur = i, from = kNone, rnd = (i + 1u) * 2654435761u + static_cast<uint32_t>(q);
    int iter = 0;
    for (; iter < 4096; iter++) {
      const unsigned long long kcur = srow[cur].x;
      uint32_t s[3];
      rslots(fmix64(kcur), s);
      bool placed = false;
#pragma unroll
      for (int c = 0; c < 3 && !placed; c++) {
        const uint32_t old = atomicCAS(&tab[s[c]], kNone, cur);
        // free slot, or a duplicate key: first writer wins.  (An eviction racing with this test can still leave two
        // entries of one key in the table: harmless, because every lookup stops at the FIRST slot holding its key and
        // the unique-key join paths need only one match.)
        if (old == kNone || srow[old].x == kcur) placed = true;
      }
      if (placed) break;
      rnd = rnd * 1664525u + 1013904223u;
      uint32_t c = (rnd >> 16) % 3u;  // evict one of the three occupants (random walk)
      if (s[c] == from) c = (c + 1u) % 3u;
      const uint32_t old = atomicExch(&tab[s[c]], cur);
      if (old == kNone) break;
      from = s[c];
      cur = old;
    }
    if (iter == 4096) atomicOr(failed, 2u);
  }
  __syncthreads();
  if (CHECK_DUP) {
    bool dup = false;
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
      const unsigned long long k = srow[i].x;
      if (k == kRadixEmpty) continue;
      uint32_t s[3];
      rslots(fmix64(k), s);
      bool found = false;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const uint32_t id = tab[s[c]];
        if (!found && id != kNone && srow[id].x == k) {
          found = true;
          dup = dup || id != i;
        }
      }
    }
    if (dup) atomicOr(failed, 8u);
  }
  ulonglong2 *dst = table + static_cast<int64_t>(q) * kRadixSlots;
  for (int i = threadIdx.x; i < kRadixSlots; i += THREADS) {
    const uint32_t id = tab[i];
    dst[i] = id == kNone ? make_ulonglong2(kRadixEmpty, 0) : srow[id];
  }
}

// ------------------------------------------------------------------ level 3, direct-addressed form
// The key words of a sub-partition are x = key - lo with the same high bits: slot = the low kRadixDirectBits bits, and
// unique keys cannot collide.  The image is assembled in LDS by plain stores and written out coalesced; untouched slots
// hold the EMPTY marker, which no key word of a direct table equals (x < 2^28).  Only the first 2^kRadixDirectBits
// slots of the table are ever read.
template <bool CHECK_DUP>
__global__ __launch_bounds__(512) void radix_direct_table_kernel(const ulonglong2 *__restrict__ rows,
                                                                 const uint32_t *__restrict__ sub_count,
                                                                 ulonglong2 *__restrict__ table, uint32_t *__restrict__ failed) {
  constexpr int THREADS = 512, SLOTS = 1 << kRadixDirectBits;
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  __shared__ ulonglong2 img[SLOTS];                 // 32 KiB: the table image
  __shared__ uint32_t claim[CHECK_DUP ? SLOTS : 1];  // CHECK_DUP: which row took the slot
  const int q = blockIdx.x;
  const int64_t base = static_cast<int64_t>(q) * kRadixSubCap;
  uint32_t n = sub_count[q];
  if (n > static_cast<uint32_t>(kRadixSubCap)) n = kRadixSubCap;  // overflow is reported by the split kernel
  for (int i = threadIdx.x; i < SLOTS; i += THREADS) {
    img[i] = make_ulonglong2(kRadixEmpty, 0);
    if (CHECK_DUP) claim[i] = kNone;
  }
  __syncthreads();
  bool dup = false;
  for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
    const ulonglong2 r = rows[base + i];
    const uint32_t slot = static_cast<uint32_t>(r.x) & (SLOTS - 1);
    if (CHECK_DUP) dup = dup || atomicCAS(&claim[slot], kNone, i) != kNone;
    // two rows of one key may both store here: the slot ends up with that key and the payload of one of them
    img[slot] = r;
  }
  if (CHECK_DUP && dup) atomicOr(failed, 8u);
  __syncthreads();
  ulonglong2 *dst = table + static_cast<int64_t>(q) * kRadixSlots;
  for (int i = threadIdx.x; i < SLOTS; i += THREADS) dst[i] = img[i];
}

// ------------------------------------------------------------------ level 3, perfect-hash form
// The probe kernel is bound by the number of divergent table reads it issues, and a cuckoo lookup costs
// 1.5 of them on average.  A sub-partition's keys are known when its table is built, so the table can be a PERFECT
// hash table (compress-hash-displace): the keys fall into kRadixBuckets buckets (~2 keys each); the buckets are
// placed largest first, each one searching the displacement d in [0, 255] that sends all its keys to free slots,
// slot = scale(a + d * b) with (a, b) two hash words of the key.  A lookup reads its bucket's displacement byte -- from
// LDS in the partition-at-a-time probe, which stages its partition's displacement bytes (<= 64 KiB) once -- and then
// exactly ONE table slot, hit or miss.  A bucket that finds no displacement raises `failed` and the build falls back
// to the cuckoo kernel (needs a >= 256-try streak of collisions; not observed).
__device__ __forceinline__ void chd_words(uint64_t h, uint32_t &bucket, uint32_t &a, uint32_t &b) {
  const uint64_t g = fmix64(h + 0x9E3779B97F4A7C15ULL);
  bucket = static_cast<uint32_t>(g) & (kRadixBuckets - 1);
  a = static_cast<uint32_t>(h);
  b = static_cast<uint32_t>(g >> 32) | 1u;
}
// The candidate slots of a key must look random in d: an arithmetic progression a + d * step (any step) visits only a
// handful of slots for the few percent of steps close to a small fraction of 2^32, so the progression goes through a
// multiply-xorshift mixer keyed by the key's second word.
__device__ __forceinline__ uint32_t chd_slot(uint32_t a, uint32_t b, uint32_t d) {
  uint32_t x = a + d * 0x9E3779B1u;
  x ^= x >> 16;
  x *= b;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return rscale(x);
}

// Displacement search of one bucket of S keys (S small and known at compile time: the keys' hash words stay in
// registers and the S claims of a try are issued together, so a try costs one LDS round trip, not S dependent ones).
template <int S>
__device__ __forceinline__ uint32_t chd_place(uint32_t *tab, const unsigned long long *skey, const uint16_t *order, uint32_t st) {
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  constexpr int K = S == 1 ? 8 : (S == 2 ? 4 : 1);  // displacements looked at per LDS round trip before claiming one
  uint32_t id[S], a[S], b[S];
  unsigned long long key[S];
  bool act[S];  // false: a duplicate of an earlier key of the bucket (first writer wins: it claims nothing)
#pragma unroll
  for (int t = 0; t < S; t++) {
    id[t] = order[st + t];
    key[t] = skey[id[t]];
    uint32_t bb;
    chd_words(fmix64(key[t]), bb, a[t], b[t]);
    act[t] = true;
#pragma unroll
    for (int u = 0; u < t; u++) act[t] = act[t] && key[u] != key[t];
  }
#pragma nounroll
  for (uint32_t d0 = 0; d0 < 256; d0 += K) {
    uint32_t sl[K][S];
    bool free_[K];
#pragma unroll
    for (int c = 0; c < K; c++)
#pragma unroll
      for (int t = 0; t < S; t++) sl[c][t] = chd_slot(a[t], b[t], d0 + c);
    if (K > 1) {  // plain reads of all K x S candidate slots go out together; the claim below decides
      uint32_t occ[K][S];
#pragma unroll
      for (int c = 0; c < K; c++)
#pragma unroll
        for (int t = 0; t < S; t++) occ[c][t] = tab[sl[c][t]];
#pragma unroll
      for (int c = 0; c < K; c++) {
        free_[c] = true;
#pragma unroll
        for (int t = 0; t < S; t++) free_[c] = free_[c] && (!act[t] || occ[c][t] == kNone);
      }
    } else {
      free_[0] = true;
    }
#pragma unroll
    for (int c = 0; c < K; c++) {
      if (!free_[c]) continue;
      uint32_t old[S];
#pragma unroll
      for (int t = 0; t < S; t++) old[t] = act[t] ? atomicCAS(&tab[sl[c][t]], kNone, id[t]) : kNone;
      bool ok = true;
#pragma unroll
      for (int t = 0; t < S; t++) ok = ok && old[t] == kNone;
      if (ok) return d0 + c;
      if (S > 1) {
#pragma unroll
        for (int t = 0; t < S; t++)
          if (act[t] && old[t] == kNone) atomicExch(&tab[sl[c][t]], kNone);  // release, try the next displacement
      }
    }
  }
  return 256;
}

static constexpr int kChdMaxBucket = 31;  // keys one bucket may hold (Poisson(2.5) over 2^26 buckets stays far below)
template <bool CHECK_DUP>
__global__ __launch_bounds__(512) void radix_chd_table_kernel(const ulonglong2 *__restrict__ rows,
                                                              const uint32_t *__restrict__ sub_count,
                                                              ulonglong2 *__restrict__ table, uint8_t *__restrict__ disp,
                                                              RadixSpecial *__restrict__ special, uint32_t *__restrict__ failed) {
  constexpr int THREADS = 512;
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  __shared__ unsigned long long skey[kRadixSubCap];  // 20 KiB
  __shared__ uint32_t tab[kRadixSlots];              // 12 KiB: row id per slot
  __shared__ uint32_t bcnt[kRadixBuckets];           // rows per bucket, then (start << 8 | rows)
  __shared__ uint16_t order[kRadixSubCap];           // row ids bucket by bucket
  __shared__ uint16_t rnk[kRadixSubCap];
  __shared__ uint16_t bsorted[kRadixBuckets];        // bucket ids, largest bucket first
  __shared__ uint8_t sdisp[kRadixBuckets];
  __shared__ uint32_t s_wsum[THREADS / 64];
  __shared__ uint32_t hsz[kChdMaxBucket + 2];        // buckets per size, then the fill cursor of a size in bsorted
  __shared__ uint32_t hfirst[kChdMaxBucket + 2];     // first position of a size in bsorted; [kChdMaxBucket + 1] unused
  const int q = blockIdx.x;
  const int64_t base = static_cast<int64_t>(q) * kRadixSubCap;
  uint32_t n = sub_count[q];
  if (n > static_cast<uint32_t>(kRadixSubCap)) n = kRadixSubCap;  // overflow is reported by the split kernel
  for (uint32_t i = threadIdx.x; i < n; i += THREADS) skey[i] = rows[base + i].x;
  for (int i = threadIdx.x; i < kRadixSlots; i += THREADS) tab[i] = kNone;
  for (int i = threadIdx.x; i < kRadixBuckets; i += THREADS) bcnt[i] = 0, sdisp[i] = 0;
  if (threadIdx.x < kChdMaxBucket + 2) hsz[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
    const unsigned long long k = skey[i];
    if (k == kRadixEmpty) {  // the one key equal to the empty marker lives beside the table
      special->val = rows[base + i].y;
      if (CHECK_DUP) {
        if (atomicExch(&special->present, 1u) != 0u) atomicOr(failed, 8u);
      } else {
        special->present = 1;
      }
      rnk[i] = 0xFFFF;
      continue;
    }
    uint32_t bk, a, b;
    chd_words(fmix64(k), bk, a, b);
    rnk[i] = static_cast<uint16_t>(atomicAdd(&bcnt[bk], 1u));
  }
  __syncthreads();
  {  // exclusive scan of the bucket sizes (2 per thread), kept as (start << 8 | size); histogram of the sizes
    constexpr int PER = kRadixBuckets / THREADS;
    uint32_t c[PER], sum = 0;
#pragma unroll
    for (int t = 0; t < PER; t++) {
      c[t] = bcnt[threadIdx.x * PER + t];
      sum += c[t];
      atomicAdd(&hsz[c[t] > static_cast<uint32_t>(kChdMaxBucket) ? kChdMaxBucket + 1 : c[t]], 1u);
    }
    const uint32_t incl = wave_incl_scan_u32(sum);
    if (lane_id() == 63) s_wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t ex = incl - sum;
    for (int w = 0; w < static_cast<int>(threadIdx.x >> 6); w++) ex += s_wsum[w];
#pragma unroll
    for (int t = 0; t < PER; t++) {
      bcnt[threadIdx.x * PER + t] = (ex << 8) | (c[t] > 255u ? 255u : c[t]);
      ex += c[t];
    }
  }
  __syncthreads();
  if (hsz[kChdMaxBucket + 1] != 0) {  // a bucket beyond what one thread places (uniform branch: LDS value after a barrier)
    if (threadIdx.x == 0) atomicOr(failed, 4u);
    return;
  }
  if (threadIdx.x == 0) {  // first position of every size in the largest-first order (32 values: serial is fine)
    uint32_t pos = 0;
    for (int sz = kChdMaxBucket; sz >= 0; sz--) {
      const uint32_t c = hsz[sz];
      hsz[sz] = pos;
      hfirst[sz] = pos;
      pos += c;
    }
  }
  for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
    if (rnk[i] == 0xFFFF) continue;
    uint32_t bk, a, b;
    chd_words(fmix64(skey[i]), bk, a, b);
    order[(bcnt[bk] >> 8) + rnk[i]] = static_cast<uint16_t>(i);
  }
  __syncthreads();
  for (int bk = threadIdx.x; bk < kRadixBuckets; bk += THREADS) bsorted[atomicAdd(&hsz[bcnt[bk] & 255u], 1u)] = static_cast<uint16_t>(bk);
  __syncthreads();
  // Largest buckets first, one size class at a time (a large bucket needs a nearly empty table: the smaller ones must
  // not get ahead of it); the buckets of one size search concurrently: claims are atomic and a failed attempt releases
  // what it claimed, so a concurrent search only ever sees a slot as taken that may be free a moment later.
  // A duplicate build key keeps its first claimant (first writer wins, as in the cuckoo table): equal keys share a
  // bucket and every slot, so the later one finds its own key in the slot and lets it stand.
  for (int size_i = kChdMaxBucket; size_i >= 1; size_i--) {
    const uint32_t size = static_cast<uint32_t>(size_i);
    const uint32_t first = hfirst[size], last = hfirst[size - 1];  // (uniform: written before the barriers above)
    if (first == last) continue;
    for (uint32_t pos = first + threadIdx.x; pos < last; pos += THREADS) {
      const uint32_t bk = bsorted[pos];
      const uint32_t st = bcnt[bk] >> 8;
      {
      uint32_t d = 0;
      if (size == 1) d = chd_place<1>(tab, skey, order, st);
      else if (size == 2) d = chd_place<2>(tab, skey, order, st);
      else if (size == 3) d = chd_place<3>(tab, skey, order, st);
      else if (size == 4) d = chd_place<4>(tab, skey, order, st);
      else if (size == 5) d = chd_place<5>(tab, skey, order, st);
      else if (size == 6) d = chd_place<6>(tab, skey, order, st);
      else if (size == 7) d = chd_place<7>(tab, skey, order, st);
      else if (size == 8) d = chd_place<8>(tab, skey, order, st);
      else
#pragma nounroll
      for (; d < 256; d++) {
        uint32_t claimed = 0;
        bool okk = true;
#pragma nounroll
        for (uint32_t t = 0; t < size && okk; t++) {
          const uint32_t id = order[st + t];
          const unsigned long long k = skey[id];
          uint32_t bb, a, b;
          chd_words(fmix64(k), bb, a, b);
          const uint32_t old = atomicCAS(&tab[chd_slot(a, b, d)], kNone, id);
          if (old == kNone) claimed |= 1u << t;
          else if (skey[old] != k) okk = false;
        }
        if (okk) break;
#pragma nounroll
        for (uint32_t t = 0; t < size; t++) {
          if (!((claimed >> t) & 1u)) continue;
          uint32_t bb, a, b;
          chd_words(fmix64(skey[order[st + t]]), bb, a, b);
          atomicExch(&tab[chd_slot(a, b, d)], kNone);  // release, try the next displacement
        }
      }
      if (d == 256) {
        if (atomicOr(failed, 2u) == 0u) {
          uint32_t occ = 0;
          for (int i = 0; i < kRadixSlots; i++) occ += tab[i] != kNone;
          printf("[chd] q=%d bk=%u size=%u st=%u n=%u occupied=%u class=[%u,%u) keys %llx %llx\n", q, bk, size, st, n, occ, first, last,
                 skey[order[st]], skey[order[st + size - 1]]);
        }
      } else sdisp[bk] = static_cast<uint8_t>(d);
      }
    }
    __syncthreads();
  }
  if (CHECK_DUP) {  // every row must own the slot its key leads to
    bool dup = false;
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
      if (rnk[i] == 0xFFFF) continue;
      uint32_t bk, a, b;
      chd_words(fmix64(skey[i]), bk, a, b);
      dup = dup || tab[chd_slot(a, b, sdisp[bk])] != i;
    }
    if (dup) atomicOr(failed, 8u);
  }
  ulonglong2 *dst = table + static_cast<int64_t>(q) * kRadixSlots;
  for (int i = threadIdx.x; i < kRadixSlots; i += THREADS) {
    const uint32_t id = tab[i];
    dst[i] = id == kNone ? make_ulonglong2(kRadixEmpty, 0) : rows[base + id];
  }
  for (int i = threadIdx.x; i < kRadixBuckets; i += THREADS) disp[static_cast<int64_t>(q) * kRadixBuckets + i] = sdisp[i];
}

// ------------------------------------------------------------------ probe
struct RadixProbeArgs {
  const ulonglong2 *prow;  // partitioned probe rows (MODE 0)
  const uint32_t *pstart;
  const uint32_t *pcount;
  RadixSrc src;  // probe columns (MODE 1, 2)
  int64_t n;
  const ulonglong2 *table;
  const RadixSpecial *special;
  const uint8_t *disp;  // perfect-hash tables: one displacement byte per (sub-partition, bucket)
  uint32_t smask;
  int32_t sub_bits;
  int32_t wg_per_part;
  int32_t _pad;
  // CELLS (MODE 0): the partitioned rows live in fixed-capacity cells (radix_scatter_kernel<.., true>)
  const uint32_t *cell_pref;  // [partition][nchunks + 1]
  uint32_t nchunks, cell_cap;
  uint64_t key_add;  // direct-addressed tables: lo, added to a key word that is emitted as a key column
  RadixEmit emit;
};

// MODE 0: partition-at-a-time (workgroups b, b+8, b+16, ... share an XCD, so the W
//         workgroups of a partition are given consecutive ids on one XCD and its CSR is
//         read from HBM once and then served by that XCD's L2);
// MODE 1: direct lookups of unpartitioned probe rows (small probes);
// MODE 2: only the rows whose key is NULL, as unmatched (LEFT OUTER / LEFT ANTI).
// TBL: 0 three-way cuckoo tables, 1 perfect-hash (CHD) tables, 2 direct-addressed tables (one slot read per row, nothing hashed).
template <int JT, int MODE, int TB, int RPT, int TBL, bool CELLS = false>
__global__ __launch_bounds__(TB) void radix_probe_kernel(RadixProbeArgs a) {
  constexpr bool CHD = TBL == 1;
  constexpr bool DIRECT = TBL == 2;
  constexpr int kWaves = TB / 64;
  constexpr int kTile = TB * RPT;
  __shared__ uint32_t s_cnt[RPT * kWaves];
  __shared__ uint32_t s_pre[RPT * kWaves];
  __shared__ unsigned long long s_base;
  __shared__ uint32_t s_pref[CELLS ? 513 : 1];  // CELLS: rows of the partition's cells before each cell
  float cell_guess = 0.f;
  extern __shared__ __align__(16) uint8_t s_disp[];  // CHD, MODE 0: the partition's displacement bytes [nsub][kRadixBuckets]
  const int tid = threadIdx.x, wave = tid >> 6;
  const uint32_t nsub = a.smask + 1;

  int64_t lo, hi, step;
  uint32_t part = 0;
  if (MODE == 0) {
    const uint32_t b = blockIdx.x, x = b & 7u, i = b >> 3;
    part = (i / a.wg_per_part) * 8u + x;
    const uint32_t slice = i % a.wg_per_part;
    // the workgroup takes tiles slice, slice + W, ... of its partition: no undersized slices
    const int64_t pbeg = CELLS ? 0 : a.pstart[part];  // CELLS: positions are virtual (the partition's rows, cell after cell)
    lo = pbeg + static_cast<int64_t>(slice) * kTile;
    hi = pbeg + a.pcount[part];
    step = static_cast<int64_t>(a.wg_per_part) * kTile;
    if (CELLS) {
      for (uint32_t i = tid; i <= a.nchunks; i += TB) s_pref[i] = a.cell_pref[part * (a.nchunks + 1) + i];
      cell_guess = hi > 0 ? static_cast<float>(a.nchunks) / static_cast<float>(hi) : 0.f;
      __syncthreads();
    }
    if (CHD && lo < hi) {  // stage the partition's displacement bytes: every lookup below then costs ONE table read
      const uint4 *src = reinterpret_cast<const uint4 *>(a.disp + static_cast<int64_t>(part) * nsub * kRadixBuckets);
      uint4 *dst = reinterpret_cast<uint4 *>(s_disp);
      for (uint32_t i = tid; i < nsub * (kRadixBuckets / 16); i += TB) dst[i] = src[i];
      __syncthreads();
    }
  } else {
    lo = static_cast<int64_t>(blockIdx.x) * kTile;
    hi = a.n;
    step = static_cast<int64_t>(gridDim.x) * kTile;
  }

  for (int64_t tbase = lo; tbase < hi; tbase += step) {
    uint64_t k[RPT], pay[RPT];
    unsigned long long val[RPT];
    bool look[RPT], inb[RPT], found[RPT], emit[RPT], knull[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const int64_t r = tbase + static_cast<int64_t>(j) * TB + tid;
      inb[j] = r < hi;
      look[j] = false, knull[j] = false, found[j] = false;
      k[j] = 0, pay[j] = 0, val[j] = 0;
      if (MODE == 0) {
        if (inb[j]) {  // (prefetching the next tile here was measured: no gain, the kernel is L2-request bound)
          int64_t phys = r;
          if (CELLS) {  // the cell holding virtual row r: cells are nearly equal, so a proportional guess is off by <= 1-2
            const uint32_t v = static_cast<uint32_t>(r);
            uint32_t c = static_cast<uint32_t>(static_cast<float>(v) * cell_guess);
            if (c >= a.nchunks) c = a.nchunks - 1;
            while (v < s_pref[c]) c--;
            while (v >= s_pref[c + 1]) c++;
            phys = static_cast<int64_t>(part * a.nchunks + c) * a.cell_cap + (v - s_pref[c]);
          }
          const ull2 row = __builtin_nontemporal_load(reinterpret_cast<const ull2 *>(a.prow) + phys);
          k[j] = row.x, pay[j] = row.y;
          look[j] = true;
        }
      } else if (inb[j]) {
        knull[j] = !radix_key<DIRECT>(a.src, r, k[j]);  // NULL, or outside the build side's key range: matches nothing
        if (MODE == 2) {
          inb[j] = knull[j];  // this pass only sees the rows without a key
        } else {
          look[j] = !knull[j];
          if (knull[j]) inb[j] = false;  // rows without a key are emitted by the MODE 2 pass
        }
        pay[j] = a.src.pay ? rload(a.src.pay, r, a.src.pay_width) : static_cast<uint64_t>(r);
      }
    }
    if (MODE != 2 && CHD) {
      // perfect-hash tables: the bucket's displacement byte (LDS in MODE 0), then exactly one slot read per row
      ulonglong2 rw[RPT];
      uint32_t db[RPT], sa[RPT], sb[RPT];
      const ulonglong2 *tb[RPT];
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (look[j] && k[j] == kRadixEmpty) {
          found[j] = a.special->present != 0;
          val[j] = a.special->val;
          look[j] = false;
        }
        db[j] = 0, sa[j] = 0, sb[j] = 1;
        tb[j] = a.table;
        if (look[j]) {
          const uint64_t h = fmix64(k[j]);
          const uint32_t p = MODE == 0 ? part : rpart(h);
          const uint32_t sub = rsub(h, a.smask);
          uint32_t bk;
          chd_words(h, bk, sa[j], sb[j]);
          tb[j] = a.table + static_cast<int64_t>(p * nsub + sub) * kRadixSlots;
          db[j] = MODE == 0 ? s_disp[sub * kRadixBuckets + bk]
                            : a.disp[static_cast<int64_t>(p * nsub + sub) * kRadixBuckets + bk];
        }
      }
#pragma unroll
      for (int j = 0; j < RPT; j++)
        if (look[j]) rw[j] = tb[j][chd_slot(sa[j], sb[j], db[j])];
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (look[j] && rw[j].x == k[j]) {
          found[j] = true;
          val[j] = rw[j].y;
        }
      }
    } else if (MODE != 2 && DIRECT) {
      // direct-addressed tables: the key word IS the address -- partition, sub-partition and slot are bit fields of x
      ulonglong2 rw[RPT];
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (look[j]) {
          const uint64_t q = k[j] >> kRadixDirectBits;  // (partition << sub_bits) | sub-partition
          rw[j] = a.table[static_cast<int64_t>(q) * kRadixSlots + (k[j] & ((1ull << kRadixDirectBits) - 1ull))];
        }
      }
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (look[j] && rw[j].x == k[j]) {
          found[j] = true;
          val[j] = rw[j].y;
        }
      }
    } else if (MODE != 2) {
      // At most three dependent slot reads per row (3-way cuckoo).  Each stage is issued for all
      // rows of the lane before the next one resolves; only lanes that still need a slot take
      // part in a load (the cost of a divergent load is ~2 cycles per ACTIVE lane).  Everything
      // is unrolled: indexing k[]/val[] with a loop variable would put them in scratch memory.
      const ulonglong2 *tb[RPT];
      uint32_t s1[RPT], s2[RPT];
      ulonglong2 rw[RPT];
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (look[j] && k[j] == kRadixEmpty) {
          found[j] = a.special->present != 0;
          val[j] = a.special->val;
          look[j] = false;
        }
        tb[j] = a.table;
        s1[j] = s2[j] = 0;
        if (look[j]) {
          const uint64_t h = fmix64(k[j]);
          const uint32_t p = MODE == 0 ? part : rpart(h);
          tb[j] = a.table + static_cast<int64_t>(p * nsub + rsub(h, a.smask)) * kRadixSlots;
          uint32_t s[3];
          rslots(h, s);
          s1[j] = s[1], s2[j] = s[2];
          rw[j] = tb[j][s[0]];
        }
      }
#pragma unroll
      for (int c = 1; c <= 3; c++) {
#pragma unroll
        for (int j = 0; j < RPT; j++) {
          if (look[j]) {
            if (rw[j].x == k[j]) {
              found[j] = true;
              val[j] = rw[j].y;
              look[j] = false;
            } else if (rw[j].x == kRadixEmpty || c == 3) {
              look[j] = false;
            } else {
              rw[j] = tb[j][c == 1 ? s1[j] : s2[j]];
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      if (JT == CRGPU_INNER || JT == CRGPU_LEFT_SEMI) emit[j] = found[j];
      else if (JT == CRGPU_LEFT_ANTI) emit[j] = inb[j] && !found[j];
      else emit[j] = inb[j];  // LEFT OUTER keeps every probe row
    }
    uint32_t rank[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const uint64_t m = __ballot(emit[j]);
      rank[j] = rank_in_mask(m);
      if (lane_id() == 0) s_cnt[j * kWaves + wave] = popc64(m);
    }
    __syncthreads();
    if (wave == 0) {
      constexpr int kPer = (RPT * kWaves + 63) / 64;
      uint32_t e[kPer], sum = 0;
#pragma unroll
      for (int t = 0; t < kPer; t++) {
        const int q = lane_id() * kPer + t;
        e[t] = q < RPT * kWaves ? s_cnt[q] : 0;
        sum += e[t];
      }
      const uint32_t incl = wave_incl_scan_u32(sum);
      uint32_t ex = incl - sum;
#pragma unroll
      for (int t = 0; t < kPer; t++) {
        const int q = lane_id() * kPer + t;
        if (q < RPT * kWaves) s_pre[q] = ex;
        ex += e[t];
      }
      if (lane_id() == 63) s_base = incl ? atomicAdd(a.emit.out_count, static_cast<unsigned long long>(incl)) : 0ull;
    }
    __syncthreads();
    const unsigned long long obase = s_base;
    int64_t pos[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) pos[j] = static_cast<int64_t>(obase + s_pre[j * kWaves + wave] + rank[j]);
    for (int c = 0; c < a.emit.ncopy; c++) {
      const RadixCopy cc = a.emit.copy[c];
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (!emit[j]) continue;
        uint64_t v;
        uint32_t present = 1;
        if (cc.side == R_KEY) {
          v = DIRECT ? k[j] + a.key_add : k[j];
          present = !knull[j];
        } else if (cc.side == R_PROBE_PAY) {
          v = pay[j];
        } else {  // build side: NULL when a LEFT OUTER row has no match (hashjoiner.go:671-682)
          v = cc.side == R_BUILD_KEY ? (DIRECT ? k[j] + a.key_add : k[j]) : val[j];
          present = found[j];
        }
        rstore(cc.dst, pos[j], cc.width, v);
        if (cc.dst_valid) cc.dst_valid[pos[j]] = static_cast<uint8_t>(present);
      }
    }
    if (a.emit.pair_p != nullptr) {
#pragma unroll
      for (int j = 0; j < RPT; j++)
        if (emit[j]) a.emit.pair_p[pos[j]] = static_cast<int32_t>(pay[j]);
    }
    if (a.emit.pair_b != nullptr) {
#pragma unroll
      for (int j = 0; j < RPT; j++)
        if (emit[j]) a.emit.pair_b[pos[j]] = found[j] ? static_cast<int32_t>(val[j]) : -1;
    }
    __syncthreads();  // s_cnt / s_pre / s_base are reused by the next tile
  }
}

// ================================================================== duplicate build keys
// One workgroup per sub-partition (<= kRadixSubCap rows).  The rows are grouped by key in LDS: a linear-probing table of
// row ids (CAS; a row either becomes the head of a new group or joins the group whose head has its key) gives every row
// its group and its rank inside it; an exclusive scan over the groups' counts gives every group its first position;
// the payloads are written grouped to pay_sorted; the group HEADS (distinct keys, so no two equal keys ever race) are
// inserted into the 3-way cuckoo table, whose image holds {key, (first grouped row << 16) | rows}.
static constexpr int kGroupSlots = 4096;  // linear-probing group table (load <= 0.625)
static constexpr size_t kGroupLds = static_cast<size_t>(kRadixSubCap) * 8 + kGroupSlots * 4 * 2 + kRadixSlots * 4 + kRadixSubCap * 2 * 2;
__global__ __launch_bounds__(512) void radix_group_kernel(const ulonglong2 *__restrict__ rows,
                                                          const uint32_t *__restrict__ sub_count,
                                                          ulonglong2 *__restrict__ table,
                                                          unsigned long long *__restrict__ pay_sorted,
                                                          RadixSpecial *__restrict__ special, uint32_t *__restrict__ failed) {
  constexpr int THREADS = 512;
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  extern __shared__ __align__(16) unsigned char g_lds[];
  unsigned long long *skey = reinterpret_cast<unsigned long long *>(g_lds);                       // [kRadixSubCap]
  uint32_t *lp = reinterpret_cast<uint32_t *>(skey + kRadixSubCap);                                // [kGroupSlots] head row of the group
  uint32_t *gcnt = lp + kGroupSlots;                                                                // [kGroupSlots] rows, then (start << 16 | rows)
  uint32_t *tab = gcnt + kGroupSlots;                                                               // [kRadixSlots] cuckoo: head row ids
  uint16_t *gof = reinterpret_cast<uint16_t *>(tab + kRadixSlots);                                  // [kRadixSubCap] group slot of the row
  uint16_t *rnk = gof + kRadixSubCap;                                                               // [kRadixSubCap] rank inside the group
  __shared__ uint32_t s_wsum[THREADS / 64];
  const int q = blockIdx.x;
  const int64_t base = static_cast<int64_t>(q) * kRadixSubCap;
  uint32_t n = sub_count[q];
  if (n > static_cast<uint32_t>(kRadixSubCap)) n = kRadixSubCap;  // overflow is reported by the split kernel
  for (uint32_t i = threadIdx.x; i < n; i += THREADS) skey[i] = rows[base + i].x;
  for (int i = threadIdx.x; i < kGroupSlots; i += THREADS) lp[i] = kNone, gcnt[i] = 0;
  for (int i = threadIdx.x; i < kRadixSlots; i += THREADS) tab[i] = kNone;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
    const unsigned long long k = skey[i];
    uint32_t s = static_cast<uint32_t>(fmix64(k) >> 20) & (kGroupSlots - 1);
    while (true) {
      const uint32_t old = atomicCAS(&lp[s], kNone, i);
      if (old == kNone || skey[old] == k) break;
      s = (s + 1) & (kGroupSlots - 1);
    }
    gof[i] = static_cast<uint16_t>(s);
    rnk[i] = static_cast<uint16_t>(atomicAdd(&gcnt[s], 1u));
  }
  __syncthreads();
  {  // exclusive scan of the 4096 group counts, 8 per thread, written back as (start << 16 | rows)
    constexpr int PER = kGroupSlots / THREADS;
    uint32_t c[PER], sum = 0;
#pragma unroll
    for (int t = 0; t < PER; t++) {
      c[t] = gcnt[threadIdx.x * PER + t];
      sum += c[t];
    }
    const uint32_t incl = wave_incl_scan_u32(sum);
    if (lane_id() == 63) s_wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t ex = incl - sum;
    for (int w = 0; w < static_cast<int>(threadIdx.x >> 6); w++) ex += s_wsum[w];
#pragma unroll
    for (int t = 0; t < PER; t++) {
      gcnt[threadIdx.x * PER + t] = (ex << 16) | c[t];
      ex += c[t];
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
    const uint32_t g = gof[i];
    pay_sorted[base + (gcnt[g] >> 16) + rnk[i]] = rows[base + i].y;
    if (lp[g] != i) continue;  // only the head of a group goes into the table
    const unsigned long long kh = skey[i];
    if (kh == kRadixEmpty) {  // the one key equal to the empty marker lives beside the table
      special->val = (static_cast<unsigned long long>(base + (gcnt[g] >> 16)) << 16) | (gcnt[g] & 0xFFFFu);
      special->present = 1;
      continue;
    }
    uint32_t cur = i, from = kNone, rnd = (i + 1u) * 2654435761u + static_cast<uint32_t>(q);
    int iter = 0;
    for (; iter < 4096; iter++) {
      uint32_t s[3];
      rslots(fmix64(skey[cur]), s);
      bool placed = false;
#pragma unroll
      for (int c = 0; c < 3 && !placed; c++) placed = atomicCAS(&tab[s[c]], kNone, cur) == kNone;
      if (placed) break;
      rnd = rnd * 1664525u + 1013904223u;
      uint32_t c = (rnd >> 16) % 3u;  // evict one of the three occupants (random walk)
      if (s[c] == from) c = (c + 1u) % 3u;
      const uint32_t old = atomicExch(&tab[s[c]], cur);
      if (old == kNone) break;
      from = s[c];
      cur = old;
    }
    if (iter == 4096) atomicOr(failed, 2u);
  }
  __syncthreads();
  ulonglong2 *dst = table + static_cast<int64_t>(q) * kRadixSlots;
  for (int i = threadIdx.x; i < kRadixSlots; i += THREADS) {
    const uint32_t id = tab[i];
    if (id == kNone) {
      dst[i] = make_ulonglong2(kRadixEmpty, 0);
    } else {
      const uint32_t g = gcnt[gof[id]];
      dst[i] = make_ulonglong2(skey[id], (static_cast<unsigned long long>(base + (g >> 16)) << 16) | (g & 0xFFFFu));
    }
  }
}

struct RadixDupArgs {
  const ulonglong2 *prow;  // partitioned probe rows (MODE 0)
  const uint32_t *pstart;
  const uint32_t *pcount;
  RadixSrc src;  // probe columns (MODE 1)
  int64_t n;
  const ulonglong2 *table;
  const RadixSpecial *special;
  const unsigned long long *pay_sorted;
  unsigned long long *vals;      // slot value of every probe row (0 = no match), indexed like prow (MODE 0) / the rows (MODE 1)
  unsigned long long *counters;  // [0] output rows, [1] probe rows seen
  uint32_t smask;
  int32_t wg_per_part;
  int32_t left_outer;
  int32_t _pad;
  RadixEmit emit;
};

// The tile loop bounds of a workgroup: MODE 0 partition-at-a-time with the XCD mapping of radix_probe_kernel, MODE 1 the
// unpartitioned rows.
template <int MODE, int kTile>
__device__ __forceinline__ void dup_tiles(const RadixDupArgs &a, int64_t &lo, int64_t &hi, int64_t &step, uint32_t &part) {
  part = 0;
  if (MODE == 0) {
    const uint32_t b = blockIdx.x, x = b & 7u, i = b >> 3;
    part = (i / a.wg_per_part) * 8u + x;
    const uint32_t slice = i % a.wg_per_part;
    const int64_t pbeg = a.pstart[part];
    lo = pbeg + static_cast<int64_t>(slice) * kTile;
    hi = pbeg + a.pcount[part];
    step = static_cast<int64_t>(a.wg_per_part) * kTile;
  } else {
    lo = static_cast<int64_t>(blockIdx.x) * kTile;
    hi = a.n;
    step = static_cast<int64_t>(gridDim.x) * kTile;
  }
}

// pass 1: look every probe row up (as radix_probe_kernel does), keep its slot value, sum the output rows
template <int MODE, int TB, int RPT>
__global__ __launch_bounds__(TB) void radix_dup_count_kernel(RadixDupArgs a) {
  constexpr int kTile = TB * RPT;
  __shared__ unsigned long long s_sum[TB / 64], s_rows[TB / 64];
  const int tid = threadIdx.x;
  const uint32_t nsub = a.smask + 1;
  int64_t lo, hi, step;
  uint32_t part;
  dup_tiles<MODE, kTile>(a, lo, hi, step, part);
  unsigned long long out_rows = 0, seen = 0;
  for (int64_t tbase = lo; tbase < hi; tbase += step) {
    uint64_t k[RPT];
    unsigned long long val[RPT];
    bool look[RPT], inb[RPT];
    const ulonglong2 *tb[RPT];
    uint32_t s1[RPT], s2[RPT];
    ulonglong2 rw[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const int64_t r = tbase + static_cast<int64_t>(j) * TB + tid;
      inb[j] = r < hi;
      look[j] = false;
      k[j] = 0, val[j] = 0;
      if (inb[j]) {
        if (MODE == 0) {
          k[j] = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(a.prow + r));
          look[j] = true;
        } else {
          look[j] = key_pack(a.src.kp, r, k[j]);
        }
      }
      if (look[j] && k[j] == kRadixEmpty) {
        if (a.special->present != 0) val[j] = a.special->val;
        look[j] = false;
      }
      tb[j] = a.table;
      s1[j] = s2[j] = 0;
      if (look[j]) {
        const uint64_t h = fmix64(k[j]);
        const uint32_t p = MODE == 0 ? part : rpart(h);
        tb[j] = a.table + static_cast<int64_t>(p * nsub + rsub(h, a.smask)) * kRadixSlots;
        uint32_t s[3];
        rslots(h, s);
        s1[j] = s[1], s2[j] = s[2];
        rw[j] = tb[j][s[0]];
      }
    }
#pragma unroll
    for (int c = 1; c <= 3; c++) {
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (look[j]) {
          if (rw[j].x == k[j]) {
            val[j] = rw[j].y;
            look[j] = false;
          } else if (rw[j].x == kRadixEmpty || c == 3) {
            look[j] = false;
          } else {
            rw[j] = tb[j][c == 1 ? s1[j] : s2[j]];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      if (!inb[j]) continue;
      const int64_t r = tbase + static_cast<int64_t>(j) * TB + tid;
      a.vals[r] = val[j];
      const unsigned long long cnt = val[j] & 0xFFFFull;
      out_rows += (a.left_outer && cnt == 0) ? 1ull : cnt;
      seen++;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    out_rows += __shfl_xor(out_rows, d, 64);
    seen += __shfl_xor(seen, d, 64);
  }
  if (lane_id() == 0) s_sum[tid >> 6] = out_rows, s_rows[tid >> 6] = seen;
  __syncthreads();
  if (tid == 0) {
    unsigned long long t0 = 0, t1 = 0;
    for (int w = 0; w < TB / 64; w++) t0 += s_sum[w], t1 += s_rows[w];
    if (t0) atomicAdd(&a.counters[0], t0);
    if (t1) atomicAdd(&a.counters[1], t1);
  }
}

// pass 2: every tile scans its rows' output counts, claims its output range with one atomicAdd and expands the rows
// cooperatively -- output row o of the tile finds its probe row by binary search over the scanned counts in LDS, so the
// output columns are written coalesced whatever the match counts are; build payloads come from the key's group in
// pay_sorted (consecutive, and L2-resident under the partition-at-a-time mapping).
template <int JT, int MODE, int TB, int RPT>
__global__ __launch_bounds__(TB) void radix_dup_emit_kernel(RadixDupArgs a) {
  constexpr int kTile = TB * RPT;
  __shared__ uint32_t s_off[kTile + 1];
  __shared__ unsigned long long s_key[kTile], s_pay[kTile], s_val[kTile];
  __shared__ uint32_t s_wsum[TB / 64];
  __shared__ unsigned long long s_base;
  const int tid = threadIdx.x;
  int64_t lo, hi, step;
  uint32_t part;
  dup_tiles<MODE, kTile>(a, lo, hi, step, part);
  for (int64_t tbase = lo; tbase < hi; tbase += step) {
    uint32_t e[RPT], sum = 0;
#pragma unroll
    for (int j = 0; j < RPT; j++) {  // blocked: thread t owns rows t * RPT .. t * RPT + RPT - 1 of the tile
      const int idx = tid * RPT + j;
      const int64_t r = tbase + idx;
      uint64_t k = 0, pay = 0;
      unsigned long long v = 0;
      e[j] = 0;
      if (r < hi) {
        bool keyed = true;
        if (MODE == 0) {
          const ull2 row = __builtin_nontemporal_load(reinterpret_cast<const ull2 *>(a.prow) + r);
          k = row.x, pay = row.y;
        } else {
          keyed = key_pack(a.src.kp, r, k);
          pay = a.src.pay ? rload(a.src.pay, r, a.src.pay_width) : static_cast<uint64_t>(r);
        }
        v = __builtin_nontemporal_load(a.vals + r);
        const uint32_t cnt = static_cast<uint32_t>(v & 0xFFFFull);
        e[j] = (JT == CRGPU_LEFT_OUTER && cnt == 0) ? 1u : cnt;
        if (!keyed) v |= 1ull << 63;  // a row without a key (NULL / out of range): its key column is NULL in the output
      }
      s_key[idx] = k, s_pay[idx] = pay, s_val[idx] = v;
      sum += e[j];
    }
    const uint32_t incl = wave_incl_scan_u32(sum);
    if (lane_id() == 63) s_wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t ex = incl - sum, total = 0;
    for (int w = 0; w < TB / 64; w++) {
      if (w < (tid >> 6)) ex += s_wsum[w];
      total += s_wsum[w];
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      s_off[tid * RPT + j] = ex;
      ex += e[j];
    }
    if (tid == 0) {
      s_off[kTile] = total;
      s_base = total ? atomicAdd(a.emit.out_count, static_cast<unsigned long long>(total)) : 0ull;
    }
    __syncthreads();
    const unsigned long long obase = s_base;
    for (uint32_t o = tid; o < total; o += TB) {
      uint32_t idx = 0;
#pragma unroll
      for (uint32_t stp = kTile / 2; stp != 0; stp >>= 1)
        if (s_off[idx + stp] <= o) idx += stp;  // the last row whose first output row is <= o owns o
      const unsigned long long v = s_val[idx];
      const bool found = (v & 0xFFFFull) != 0;
      const unsigned long long bval = found ? a.pay_sorted[((v >> 16) & 0x7FFFFFFFFFFFull) + (o - s_off[idx])] : 0ull;
      const int64_t pos = static_cast<int64_t>(obase + o);
      for (int c = 0; c < a.emit.ncopy; c++) {
        const RadixCopy cc = a.emit.copy[c];
        uint64_t x;
        uint32_t present = 1;
        if (cc.side == R_KEY) {
          x = s_key[idx];
          present = (v >> 63) == 0;
        } else if (cc.side == R_PROBE_PAY) {
          x = s_pay[idx];
        } else {  // build side: NULL when a LEFT OUTER row has no match (hashjoiner.go:671-682)
          x = cc.side == R_BUILD_KEY ? s_key[idx] : bval;
          present = found;
        }
        rstore(cc.dst, pos, cc.width, x);
        if (cc.dst_valid) cc.dst_valid[pos] = static_cast<uint8_t>(present);
      }
      if (a.emit.pair_p != nullptr) a.emit.pair_p[pos] = static_cast<int32_t>(s_pay[idx]);
      if (a.emit.pair_b != nullptr) a.emit.pair_b[pos] = found ? static_cast<int32_t>(bval) : -1;
    }
    __syncthreads();  // the LDS arrays are reused by the next tile
  }
}

// ------------------------------------------------------------------ host side
static bool env_flag_default_on(const char *name) {
  const char *v = std::getenv(name);
  return v == nullptr || v[0] != '0';
}
static void chunking(int64_t n, int *nchunks, int64_t *chunk_rows) {
  int64_t want = (n + 65535) / 65536;  // >= 64 rows per partition and chunk on average
  if (want < 1) want = 1;
  if (want > 512) want = 512;
  int64_t rows = (n + want - 1) / want;
  rows = (rows + 4095) / 4096 * 4096;
  if (rows < 4096) rows = 4096;
  *chunk_rows = rows;
  *nchunks = static_cast<int>((n + rows - 1) / rows);
  if (*nchunks < 1) *nchunks = 1;
}

// hist -> offsets -> scatter of one side into `rows` (AoS {key, payload}); totals/starts per partition.
static crgpu_status radix_partition(const RadixSrc &src, int64_t n, DevBuf &hist, DevBuf &totals, DevBuf &starts,
                                    DevBuf &rows, int64_t extra_rows, const char *prof_hist, const char *prof_scatter) {
  hipStream_t st = stream();
  int nchunks;
  int64_t chunk_rows;
  chunking(n, &nchunks, &chunk_rows);
  CRGPU_TRY(hist.ensure(static_cast<size_t>(nchunks) * kRadixP1 * 4));
  CRGPU_TRY(totals.ensure(kRadixP1 * 4));
  CRGPU_TRY(starts.ensure(kRadixP1 * 4));
  CRGPU_TRY(rows.ensure(static_cast<size_t>(n + 8 * kRadixP1 + extra_rows) * 16));
  {
    ProfScope prof(prof_hist);
    const bool direct = src.direct.span != 0;
    if (src.fast && direct)
      hipLaunchKernelGGL((radix_hist_kernel<true, true>), dim3(nchunks), dim3(1024), 0, st, src, n, chunk_rows, hist.as<uint32_t>());
    else if (src.fast)
      hipLaunchKernelGGL((radix_hist_kernel<true, false>), dim3(nchunks), dim3(1024), 0, st, src, n, chunk_rows, hist.as<uint32_t>());
    else if (direct)
      hipLaunchKernelGGL((radix_hist_kernel<false, true>), dim3(nchunks), dim3(1024), 0, st, src, n, chunk_rows, hist.as<uint32_t>());
    else
      hipLaunchKernelGGL((radix_hist_kernel<false, false>), dim3(nchunks), dim3(1024), 0, st, src, n, chunk_rows, hist.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
    hipLaunchKernelGGL(radix_colsum_kernel, dim3(kRadixP1 / 64), dim3(1024), 0, st, hist.as<uint32_t>(), nchunks,
                       totals.as<uint32_t>());
    hipLaunchKernelGGL(radix_starts_kernel, dim3(1), dim3(1024), 0, st, totals.as<uint32_t>(), starts.as<uint32_t>());
    hipLaunchKernelGGL(radix_offs_kernel, dim3(kRadixP1 / 64), dim3(1024), 0, st, hist.as<uint32_t>(), nchunks,
                       starts.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
  }
  {
    ProfScope prof(prof_scatter);
    const int form = src.fast ? (src.pay != nullptr ? 1 : 2) : 0;
#define CRGPU_SCATTER(F, D)                                                                                            \
  hipLaunchKernelGGL((radix_scatter_kernel<F, false, D>), dim3(nchunks), dim3(1024), 0, st, src, n, chunk_rows, \
                     hist.as<uint32_t>(), rows.as<ulonglong2>(), 0u, nullptr, nullptr)
    if (src.direct.span != 0) {
      if (form == 1) CRGPU_SCATTER(1, true);
      else if (form == 2) CRGPU_SCATTER(2, true);
      else CRGPU_SCATTER(0, true);
    } else {
      if (form == 1) CRGPU_SCATTER(1, false);
      else if (form == 2) CRGPU_SCATTER(2, false);
      else CRGPU_SCATTER(0, false);
    }
#undef CRGPU_SCATTER
    CRGPU_HIP(hipGetLastError());
  }
  return CRGPU_OK;
}

// The probe side without the histogram pass: one scatter into fixed-capacity (partition, chunk) cells, then the cells'
// prefix sums per partition.  *ok = false when a cell overflowed (keys whose hash bits are skewed): the caller then
// partitions with the histogram.  Costs one 4-byte flag read-back instead of a pass over the key column.
static crgpu_status radix_partition_cells(RadixTable &t, const RadixSrc &src, int64_t n, bool *ok) {
  hipStream_t st = stream();
  *ok = false;
  int nchunks;
  int64_t chunk_rows;
  chunking(n, &nchunks, &chunk_rows);
  const double mean = static_cast<double>(chunk_rows) / kRadixP1;
  uint32_t cap = static_cast<uint32_t>(mean + 8.0 * std::sqrt(mean) + 16.0);
  cap = (cap + 7u) & ~7u;
  if (static_cast<uint64_t>(kRadixP1) * nchunks * cap > 0xFFFFFFF0ull) return CRGPU_OK;  // 32-bit row positions
  CRGPU_TRY(t.p_rows.ensure(static_cast<size_t>(kRadixP1) * nchunks * cap * 16 + 256));
  CRGPU_TRY(t.p_cell_cnt.ensure(static_cast<size_t>(kRadixP1) * nchunks * 4));
  CRGPU_TRY(t.p_cell_pref.ensure(static_cast<size_t>(kRadixP1) * (nchunks + 1) * 4));
  CRGPU_TRY(t.p_totals.ensure(kRadixP1 * 4));
  CRGPU_TRY(t.flag.ensure(4));
  CRGPU_HIP(hipMemsetAsync(t.flag.p, 0, 4, st));
  {  // hot keys?  (a 4-byte read-back: ~25 us against the 4 ms a failed cells pass would cost)
    CRGPU_TRY(t.p_hist.ensure(kRadixP1 * 4));
    CRGPU_HIP(hipMemsetAsync(t.p_hist.p, 0, kRadixP1 * 4, st));
    const int64_t stride = n / kSkewSample > 0 ? n / kSkewSample : 1;
    const bool direct = src.direct.span != 0;
    if (src.fast && direct)
      hipLaunchKernelGGL((radix_skew_sample_kernel<true, true>), dim3(kSkewSample / 1024), dim3(1024), 0, st, src, n, stride, t.p_hist.as<uint32_t>());
    else if (src.fast)
      hipLaunchKernelGGL((radix_skew_sample_kernel<true, false>), dim3(kSkewSample / 1024), dim3(1024), 0, st, src, n, stride, t.p_hist.as<uint32_t>());
    else if (direct)
      hipLaunchKernelGGL((radix_skew_sample_kernel<false, true>), dim3(kSkewSample / 1024), dim3(1024), 0, st, src, n, stride, t.p_hist.as<uint32_t>());
    else
      hipLaunchKernelGGL((radix_skew_sample_kernel<false, false>), dim3(kSkewSample / 1024), dim3(1024), 0, st, src, n, stride, t.p_hist.as<uint32_t>());
    hipLaunchKernelGGL(radix_skew_check_kernel, dim3(1), dim3(1024), 0, st, t.p_hist.as<uint32_t>(), t.flag.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
    uint32_t skewed = 0;
    CRGPU_HIP(hipMemcpyAsync(&skewed, t.flag.p, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (skewed) return CRGPU_OK;  // *ok stays false
  }
  {
    ProfScope prof("radix_probe_scatter");
    const int form = src.fast ? (src.pay != nullptr ? 1 : 2) : 0;
#define CRGPU_SCATTER(F, D)                                                                                                  \
  hipLaunchKernelGGL((radix_scatter_kernel<F, true, D>), dim3(nchunks), dim3(1024), 0, st, src, n, chunk_rows, nullptr, \
                     t.p_rows.as<ulonglong2>(), cap, t.p_cell_cnt.as<uint32_t>(), t.flag.as<uint32_t>())
    if (src.direct.span != 0) {
      if (form == 1) CRGPU_SCATTER(1, true);
      else if (form == 2) CRGPU_SCATTER(2, true);
      else CRGPU_SCATTER(0, true);
    } else {
      if (form == 1) CRGPU_SCATTER(1, false);
      else if (form == 2) CRGPU_SCATTER(2, false);
      else CRGPU_SCATTER(0, false);
    }
#undef CRGPU_SCATTER
    CRGPU_HIP(hipGetLastError());
    hipLaunchKernelGGL(radix_cell_prefix_kernel, dim3(kRadixP1), dim3(512), 0, st, t.p_cell_cnt.as<uint32_t>(),
                       static_cast<uint32_t>(nchunks), t.p_cell_pref.as<uint32_t>(), t.p_totals.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
  }
  uint32_t overflow = 0;
  CRGPU_HIP(hipMemcpyAsync(&overflow, t.flag.p, 4, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  t.p_nchunks = static_cast<uint32_t>(nchunks);
  t.p_cell_cap = cap;
  *ok = overflow == 0;
  return CRGPU_OK;
}

// Level 1 only: t.rows_a holds partition p's rows in [starts[p], starts[p] + totals[p]).
crgpu_status radix_partition_level1(RadixTable &t, const RadixSrc &src, int64_t n) {
  t.ready = false;
  t.n = n;
  return radix_partition(src, n, t.hist, t.totals, t.starts, t.rows_a, 0, "radix_build_hist", "radix_build_scatter");
}

// Levels 1 and 2: t.rows_b holds the rows of sub-partition q in [q * subcap, q * subcap + sub_count[q]).
// *ok = false when a sub-partition overflowed its region (too many rows share hash bits: duplicate-heavy input).
crgpu_status radix_partition_split(RadixTable &t, const RadixSrc &src, int64_t n, bool *ok, uint32_t subcap, int log2_rows_per_sub,
                                   int max_sub_bits, uint32_t any_nsub) {
  hipStream_t st = stream();
  *ok = false;
  t.ready = false;
  t.n = n;
  int lg = 0;
  while ((int64_t(1) << lg) < n) lg++;
  int sb = lg - 10 - log2_rows_per_sub;  // 1024 level-1 partitions, then ~2^log2_rows_per_sub rows per sub-partition
  if (sb < 0) sb = 0;
  if (sb > max_sub_bits) sb = max_sub_bits;
  const bool direct = src.direct.span != 0;
  if (direct) sb = static_cast<int>(src.direct.sub_bits);  // the key range decides, not the row count
  if (any_nsub != 0 && (direct || any_nsub > 128u)) return fail(CRGPU_ERR_INVALID, "radix_partition_split: sub-partition count");
  t.sub_bits = any_nsub != 0 ? 0 : sb;  // (any_nsub: nothing may derive the sub-partition from hash bits afterwards)
  const uint32_t nsub = any_nsub != 0 ? any_nsub : 1u << sb;
  t.nsub = nsub;
  CRGPU_TRY(radix_partition(src, n, t.hist, t.totals, t.starts, t.rows_a, 0, "radix_build_hist", "radix_build_scatter"));
  CRGPU_TRY(t.rows_b.ensure(static_cast<size_t>(kRadixP1) * nsub * subcap * 16));
  CRGPU_TRY(t.sub_count.ensure(static_cast<size_t>(kRadixP1) * nsub * 4));
  CRGPU_TRY(t.flag.ensure(4));
  CRGPU_HIP(hipMemsetAsync(t.flag.p, 0, 4, st));
  {
    ProfScope prof("radix_build_split");
    if (direct)
      hipLaunchKernelGGL(radix_split_kernel<true>, dim3(kRadixP1), dim3(1024), 0, st, t.rows_a.as<ulonglong2>(),
                         t.starts.as<uint32_t>(), t.totals.as<uint32_t>(), nsub - 1, t.rows_b.as<ulonglong2>(),
                         t.sub_count.as<uint32_t>(), t.flag.as<uint32_t>(), subcap);
    else if (any_nsub != 0)
      hipLaunchKernelGGL((radix_split_kernel<false, true>), dim3(kRadixP1), dim3(1024), 0, st, t.rows_a.as<ulonglong2>(),
                         t.starts.as<uint32_t>(), t.totals.as<uint32_t>(), nsub - 1, t.rows_b.as<ulonglong2>(),
                         t.sub_count.as<uint32_t>(), t.flag.as<uint32_t>(), subcap);
    else
      hipLaunchKernelGGL(radix_split_kernel<false>, dim3(kRadixP1), dim3(1024), 0, st, t.rows_a.as<ulonglong2>(),
                         t.starts.as<uint32_t>(), t.totals.as<uint32_t>(), nsub - 1, t.rows_b.as<ulonglong2>(),
                         t.sub_count.as<uint32_t>(), t.flag.as<uint32_t>(), subcap);
    CRGPU_HIP(hipGetLastError());
  }
  uint32_t overflow = 0;
  CRGPU_HIP(hipMemcpyAsync(&overflow, t.flag.p, 4, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  *ok = overflow == 0;
  return CRGPU_OK;
}

// Levels 1 and 2 for 32-byte rows {k0, k1, value, length} (<= 16-byte BYTES keys): t.rows_b holds the rows of
// sub-partition q in [q * subcap, q * subcap + sub_count[q]).  *ok = false when a sub-partition overflowed its region or
// a key is longer than 16 bytes.
crgpu_status radix_partition_split_wide(RadixTable &t, const WideSrc &src, int64_t n, bool *ok, uint32_t subcap,
                                        int log2_rows_per_sub, int max_sub_bits, uint32_t any_nsub) {
  hipStream_t st = stream();
  *ok = false;
  t.ready = false;
  t.n = n;
  int lg = 0;
  while ((int64_t(1) << lg) < n) lg++;
  int sb = lg - 10 - log2_rows_per_sub;
  if (sb < 0) sb = 0;
  if (sb > max_sub_bits) sb = max_sub_bits;
  if (any_nsub > 128u) return fail(CRGPU_ERR_INVALID, "radix_partition_split_wide: sub-partition count");
  t.sub_bits = any_nsub != 0 ? 0 : sb;
  const uint32_t nsub = any_nsub != 0 ? any_nsub : 1u << sb;
  t.nsub = nsub;
  int nchunks;
  int64_t chunk_rows;
  chunking(n, &nchunks, &chunk_rows);
  CRGPU_TRY(t.hist.ensure(static_cast<size_t>(nchunks) * kRadixP1 * 4));
  CRGPU_TRY(t.totals.ensure(kRadixP1 * 4));
  CRGPU_TRY(t.starts.ensure(kRadixP1 * 4));
  CRGPU_TRY(t.rows_a.ensure(static_cast<size_t>(n + 8 * kRadixP1) * 32));
  CRGPU_TRY(t.rows_b.ensure(static_cast<size_t>(kRadixP1) * nsub * subcap * 32));
  CRGPU_TRY(t.sub_count.ensure(static_cast<size_t>(kRadixP1) * nsub * 4));
  CRGPU_TRY(t.flag.ensure(4));
  CRGPU_HIP(hipMemsetAsync(t.flag.p, 0, 4, st));
  {
    ProfScope prof("radix_build_hist");
    hipLaunchKernelGGL(wide_hist_kernel, dim3(nchunks), dim3(1024), 0, st, src, n, chunk_rows, t.hist.as<uint32_t>(),
                       t.flag.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
    hipLaunchKernelGGL(radix_colsum_kernel, dim3(kRadixP1 / 64), dim3(1024), 0, st, t.hist.as<uint32_t>(), nchunks,
                       t.totals.as<uint32_t>());
    hipLaunchKernelGGL(radix_starts_kernel, dim3(1), dim3(1024), 0, st, t.totals.as<uint32_t>(), t.starts.as<uint32_t>());
    hipLaunchKernelGGL(radix_offs_kernel, dim3(kRadixP1 / 64), dim3(1024), 0, st, t.hist.as<uint32_t>(), nchunks,
                       t.starts.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
  }
  {
    ProfScope prof("radix_build_scatter");
    hipLaunchKernelGGL(wide_scatter_kernel, dim3(nchunks), dim3(1024), 0, st, src, n, chunk_rows, t.hist.as<uint32_t>(),
                       t.rows_a.as<ulonglong4>());
    CRGPU_HIP(hipGetLastError());
  }
  {
    ProfScope prof("radix_build_split");
    if (any_nsub != 0)
      hipLaunchKernelGGL(wide_split_kernel<true>, dim3(kRadixP1), dim3(1024), 0, st, t.rows_a.as<ulonglong4>(), t.starts.as<uint32_t>(),
                         t.totals.as<uint32_t>(), nsub - 1, t.rows_b.as<ulonglong4>(), t.sub_count.as<uint32_t>(),
                         t.flag.as<uint32_t>(), subcap, src.val2 != nullptr ? 0ull : ~0ull);
    else
      hipLaunchKernelGGL(wide_split_kernel<false>, dim3(kRadixP1), dim3(1024), 0, st, t.rows_a.as<ulonglong4>(), t.starts.as<uint32_t>(),
                         t.totals.as<uint32_t>(), nsub - 1, t.rows_b.as<ulonglong4>(), t.sub_count.as<uint32_t>(),
                         t.flag.as<uint32_t>(), subcap, src.val2 != nullptr ? 0ull : ~0ull);
    CRGPU_HIP(hipGetLastError());
  }
  uint32_t flag = 0;
  CRGPU_HIP(hipMemcpyAsync(&flag, t.flag.p, 4, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  *ok = flag == 0;
  return CRGPU_OK;
}

int radix_direct_sub_bits(uint64_t span, int64_t n_build) {
  if (span == 0 || n_build <= 0) return -1;
  int sb = 0;
  while (sb <= kRadixMaxSubBits && (uint64_t(1) << (10 + sb + kRadixDirectBits)) < span) sb++;
  if (sb > kRadixMaxSubBits) return -1;                     // wider than every slot of the largest table set
  if (static_cast<uint64_t>(n_build) < span / 4) return -1;  // sparse keys: hashed tables
  return sb;
}

crgpu_status radix_build(RadixTable &t, const RadixSrc &src, int64_t n, bool *ok, bool check_dup) {
  hipStream_t st = stream();
  *ok = false;
  t.has_dup_keys = false;
  t.direct = RadixDirect{0, 0, 0, 0};
  bool split_ok = false;
  CRGPU_TRY(radix_partition_split(t, src, n, &split_ok, kRadixSubCap, 11, kRadixMaxSubBits));
  if (!split_ok) return CRGPU_OK;  // duplicate-heavy build side: the caller uses the classic table
  const uint32_t nsub = 1u << t.sub_bits;
  CRGPU_TRY(t.table.ensure(static_cast<size_t>(kRadixP1) * nsub * kRadixSlots * 16));
  CRGPU_TRY(t.special.ensure(sizeof(RadixSpecial)));
  CRGPU_HIP(hipMemsetAsync(t.special.p, 0, sizeof(RadixSpecial), st));
  uint32_t overflow = 0;
  t.chd = false;
  t.dup = false;
  t.cells_failed = false;
  if (src.direct.span != 0) {  // direct-addressed tables: plain placement, no hashing anywhere
    {
      ProfScope prof("radix_build_table");
      if (check_dup)
        hipLaunchKernelGGL(radix_direct_table_kernel<true>, dim3(kRadixP1 * nsub), dim3(512), 0, st, t.rows_b.as<ulonglong2>(),
                           t.sub_count.as<uint32_t>(), t.table.as<ulonglong2>(), t.flag.as<uint32_t>());
      else
        hipLaunchKernelGGL(radix_direct_table_kernel<false>, dim3(kRadixP1 * nsub), dim3(512), 0, st, t.rows_b.as<ulonglong2>(),
                           t.sub_count.as<uint32_t>(), t.table.as<ulonglong2>(), t.flag.as<uint32_t>());
      CRGPU_HIP(hipGetLastError());
    }
    if (check_dup) {
      CRGPU_HIP(hipMemcpyAsync(&overflow, t.flag.p, 4, hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipStreamSynchronize(st));
      t.has_dup_keys = (overflow & 8u) != 0;
    }
    t.direct = src.direct;
    t.ready = true;
    *ok = true;
    return CRGPU_OK;
  }
  // The perfect-hash form takes 1.06 ms off the probe of config J (6.78 -> 5.72 ms: one slot read per row instead of 1.5)
  // but its build costs 3.8 ms against the cuckoo kernel's 1.7 (size-ordered displacement searches are latency chains
  // of LDS atomics), so it only pays when the probe side is >~ 14 x the build side: opt-in, CRGPU_RADIX_TABLE=chd.
  const char *kind = std::getenv("CRGPU_RADIX_TABLE");
  if (kind != nullptr && kind[0] == 'c' && kind[1] == 'h') {
    CRGPU_TRY(t.disp.ensure(static_cast<size_t>(kRadixP1) * nsub * kRadixBuckets));
    {
      ProfScope prof("radix_build_table");
      if (check_dup)
        hipLaunchKernelGGL(radix_chd_table_kernel<true>, dim3(kRadixP1 * nsub), dim3(512), 0, st, t.rows_b.as<ulonglong2>(),
                           t.sub_count.as<uint32_t>(), t.table.as<ulonglong2>(), t.disp.as<uint8_t>(),
                           t.special.as<RadixSpecial>(), t.flag.as<uint32_t>());
      else
        hipLaunchKernelGGL(radix_chd_table_kernel<false>, dim3(kRadixP1 * nsub), dim3(512), 0, st, t.rows_b.as<ulonglong2>(),
                           t.sub_count.as<uint32_t>(), t.table.as<ulonglong2>(), t.disp.as<uint8_t>(),
                           t.special.as<RadixSpecial>(), t.flag.as<uint32_t>());
      CRGPU_HIP(hipGetLastError());
    }
    CRGPU_HIP(hipMemcpyAsync(&overflow, t.flag.p, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if ((overflow & 8u) && !(overflow & ~8u)) t.has_dup_keys = true;
    overflow &= ~8u;
    if (!overflow) {
      t.chd = true;
      t.ready = true;
      *ok = true;
      return CRGPU_OK;
    }
    // a bucket found no displacement (or held too many keys): the cuckoo form below takes over
    CRGPU_HIP(hipMemsetAsync(t.flag.p, 0, 4, st));
    CRGPU_HIP(hipMemsetAsync(t.special.p, 0, sizeof(RadixSpecial), st));
    overflow = 0;
  }
  {
    ProfScope prof("radix_build_table");
    if (check_dup)
      hipLaunchKernelGGL(radix_table_kernel<true>, dim3(kRadixP1 * nsub), dim3(512), 0, st, t.rows_b.as<ulonglong2>(),
                         t.sub_count.as<uint32_t>(), t.table.as<ulonglong2>(),
                         t.special.as<RadixSpecial>(), t.flag.as<uint32_t>());
    else
      hipLaunchKernelGGL(radix_table_kernel<false>, dim3(kRadixP1 * nsub), dim3(512), 0, st, t.rows_b.as<ulonglong2>(),
                         t.sub_count.as<uint32_t>(), t.table.as<ulonglong2>(),
                         t.special.as<RadixSpecial>(), t.flag.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
  }
  CRGPU_HIP(hipMemcpyAsync(&overflow, t.flag.p, 4, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  t.has_dup_keys = (overflow & 8u) != 0;
  overflow &= ~8u;
  if (overflow) return CRGPU_OK;  // a cuckoo insertion did not terminate (an unlucky hash): classic table
  t.ready = true;
  *ok = true;
  return CRGPU_OK;
}

template <int JT, int MODE, int TB, int RPT>
static crgpu_status launch_probe_jt(const RadixProbeArgs &a, int grid) {
  hipStream_t st = stream();
  if (a.src.direct.span != 0) {  // direct-addressed tables
    if (MODE == 0 && a.cell_pref != nullptr)
      hipLaunchKernelGGL((radix_probe_kernel<JT, MODE == 0 ? 0 : MODE, TB, RPT, 2, MODE == 0>), dim3(grid), dim3(TB), 0, st, a);
    else
      hipLaunchKernelGGL((radix_probe_kernel<JT, MODE, TB, RPT, 2>), dim3(grid), dim3(TB), 0, st, a);
    CRGPU_HIP(hipGetLastError());
    return CRGPU_OK;
  }
  if (MODE == 0 && a.cell_pref != nullptr && a.disp == nullptr) {  // rows in cells (cuckoo tables)
    hipLaunchKernelGGL((radix_probe_kernel<JT, MODE == 0 ? 0 : MODE, TB, RPT, 0, MODE == 0>), dim3(grid), dim3(TB), 0, st, a);
    CRGPU_HIP(hipGetLastError());
    return CRGPU_OK;
  }
  if (a.disp != nullptr) {  // perfect-hash tables; MODE 0 stages the partition's displacement bytes in dynamic LDS
    const size_t lds = MODE == 0 ? static_cast<size_t>(a.smask + 1) * kRadixBuckets : 0;
    if (lds > 48 * 1024) {
      static DeviceOnce attr_set;  // per instantiation
      if (attr_set.need()) {
        CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_probe_kernel<JT, MODE, TB, RPT, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * kRadixBuckets));
        attr_set.mark();
      }
    }
    hipLaunchKernelGGL((radix_probe_kernel<JT, MODE, TB, RPT, 1>), dim3(grid), dim3(TB), lds, st, a);
  } else {
    hipLaunchKernelGGL((radix_probe_kernel<JT, MODE, TB, RPT, 0>), dim3(grid), dim3(TB), 0, st, a);
  }
  CRGPU_HIP(hipGetLastError());
  return CRGPU_OK;
}
template <int MODE, int TB, int RPT>
static crgpu_status launch_probe_cfg(int jt, const RadixProbeArgs &a, int grid) {
  switch (jt) {
    case CRGPU_INNER: return launch_probe_jt<CRGPU_INNER, MODE, TB, RPT>(a, grid);
    case CRGPU_LEFT_OUTER: return launch_probe_jt<CRGPU_LEFT_OUTER, MODE, TB, RPT>(a, grid);
    case CRGPU_LEFT_SEMI: return launch_probe_jt<CRGPU_LEFT_SEMI, MODE, TB, RPT>(a, grid);
    case CRGPU_LEFT_ANTI: return launch_probe_jt<CRGPU_LEFT_ANTI, MODE, TB, RPT>(a, grid);
    default: return fail(CRGPU_ERR_INVALID, "radix probe: join type not handled");
  }
}
// 512 threads x 4 rows per tile, 64 workgroups per partition: the best of the shapes measured
// ((256,4) 7.5 ms, (1024,2) 7.3, (512,2) 6.9, (256,8) 8.3, (512,8) 8.6 against 6.6 on config J).
static constexpr int kProbeTB = 512, kProbeRPT = 4, kProbeTile = kProbeTB * kProbeRPT, kProbeWgPerPart = 64, kProbeWgPerPartDirect = 96;
template <int MODE>
static crgpu_status launch_probe(int jt, const RadixProbeArgs &a, int grid) {
  return launch_probe_cfg<MODE, kProbeTB, kProbeRPT>(jt, a, grid);
}

crgpu_status radix_probe(RadixTable &t, int jt, const RadixSrc &src_in, int64_t n, const RadixEmit &emit, bool partitioned) {
  if (!t.ready) return fail(CRGPU_ERR_STATE, "radix probe: table not built");
  if (n <= 0) return CRGPU_OK;
  constexpr int kTile = kProbeTile;
  RadixSrc src = src_in;
  src.direct = t.direct;  // the probe rows are addressed the way the tables were built
  RadixProbeArgs a;
  std::memset(&a, 0, sizeof(a));
  a.src = src;
  a.key_add = t.direct.lo;
  a.n = n;
  a.table = t.table.as<ulonglong2>();
  a.special = t.special.as<RadixSpecial>();
  a.disp = t.chd ? t.disp.as<uint8_t>() : nullptr;
  a.smask = (1u << t.sub_bits) - 1;
  a.sub_bits = t.sub_bits;
  a.emit = emit;
  const bool keeps_unmatched = jt == CRGPU_LEFT_OUTER || jt == CRGPU_LEFT_ANTI;
  if (partitioned) {
    // large probes skip the histogram pass (fixed-capacity cells); skewed keys and perfect-hash tables take the classic form
    const bool cells_on = env_flag_default_on("CRGPU_RADIX_CELLS");  // (read per probe: tests switch it)
    bool cells = false;
    if (cells_on && !t.chd && !t.cells_failed && n >= (int64_t(1) << 22) && n <= (int64_t(1) << 30)) {
      CRGPU_TRY(radix_partition_cells(t, src, n, &cells));
      if (!cells) t.cells_failed = true;  // the following probe chunks of this join go straight to the histogram form
    }
    if (cells) {
      a.cell_pref = t.p_cell_pref.as<uint32_t>();
      a.nchunks = t.p_nchunks;
      a.cell_cap = t.p_cell_cap;
    } else {
      CRGPU_TRY(radix_partition(src, n, t.p_hist, t.p_totals, t.p_starts, t.p_rows, 0, "radix_probe_hist", "radix_probe_scatter"));
    }
    a.prow = t.p_rows.as<ulonglong2>();
    a.pstart = t.p_starts.as<uint32_t>();
    a.pcount = t.p_totals.as<uint32_t>();
    int64_t w = (n / kRadixP1 + kTile - 1) / kTile;
    if (w < 1) w = 1;
    // = the workgroups resident on one XCD: 32 CUs x 2 with the cuckoo / perfect-hash lookups (86 VGPRs), x 3 with the
    // direct-addressed ones (72 VGPRs; measured on config J: 64 per partition 5.79 ms, 96 5.21, 128 5.18-5.46, 256 5.54)
    const int64_t wmax = t.direct.span != 0 ? kProbeWgPerPartDirect : kProbeWgPerPart;
    if (w > wmax) w = wmax;
    a.wg_per_part = static_cast<int32_t>(w);
    ProfScope prof("join_probe");
    CRGPU_TRY(launch_probe<0>(jt, a, kRadixP1 * static_cast<int>(w)));
  } else {
    int64_t tiles = (n + kTile - 1) / kTile;
    int64_t cap = static_cast<int64_t>(num_cus()) * 4;
    ProfScope prof("join_probe");
    CRGPU_TRY(launch_probe<1>(jt, a, static_cast<int>(tiles < cap ? tiles : cap)));
  }
  bool keyless = src.kp.n > 1 || t.direct.span != 0;  // probe values may lie outside the build side's range
  for (int c = 0; c < src.kp.n; c++) keyless = keyless || src.kp.c[c].valid != nullptr;
  if (keeps_unmatched && keyless) {  // rows without a key (NULL / out of range) never match but LEFT OUTER / ANTI keep them
    int64_t tiles = (n + kTile - 1) / kTile;
    int64_t cap = static_cast<int64_t>(num_cus()) * 4;
    CRGPU_TRY(launch_probe<2>(jt, a, static_cast<int>(tiles < cap ? tiles : cap)));
  }
  return CRGPU_OK;
}

crgpu_status radix_build_dup(RadixTable &t, const RadixSrc &src, int64_t n, bool *ok) {
  hipStream_t st = stream();
  *ok = false;
  t.dup = false;
  t.direct = RadixDirect{0, 0, 0, 0};
  bool split_ok = false;
  CRGPU_TRY(radix_partition_split(t, src, n, &split_ok, kRadixSubCap, 11, kRadixMaxSubBits));
  if (!split_ok) return CRGPU_OK;  // heavily skewed keys: a sub-partition overflowed, the caller uses the chained table
  const uint32_t nsub = 1u << t.sub_bits;
  CRGPU_TRY(t.table.ensure(static_cast<size_t>(kRadixP1) * nsub * kRadixSlots * 16));
  CRGPU_TRY(t.pay_sorted.ensure(static_cast<size_t>(kRadixP1) * nsub * kRadixSubCap * 8));
  CRGPU_TRY(t.special.ensure(sizeof(RadixSpecial)));
  CRGPU_HIP(hipMemsetAsync(t.special.p, 0, sizeof(RadixSpecial), st));
  static DeviceOnce attr_set;
  if (attr_set.need()) {
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&radix_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(kGroupLds)));
    attr_set.mark();
  }
  uint32_t overflow = 0;
  {
    ProfScope prof("radix_build_table");
    hipLaunchKernelGGL(radix_group_kernel, dim3(kRadixP1 * nsub), dim3(512), kGroupLds, st, t.rows_b.as<ulonglong2>(),
                       t.sub_count.as<uint32_t>(), t.table.as<ulonglong2>(), t.pay_sorted.as<unsigned long long>(),
                       t.special.as<RadixSpecial>(), t.flag.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
  }
  CRGPU_HIP(hipMemcpyAsync(&overflow, t.flag.p, 4, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  if (overflow) return CRGPU_OK;  // a cuckoo insertion did not terminate (an unlucky hash): chained table
  t.ready = true;
  t.dup = true;
  t.chd = false;
  *ok = true;
  return CRGPU_OK;
}

static constexpr int kDupTB = 512, kDupRPT = 4, kDupTile = kDupTB * kDupRPT;
static void dup_args(RadixTable &t, const RadixSrc &src, int64_t n, int jt, bool partitioned, RadixDupArgs &a, int *grid) {
  std::memset(&a, 0, sizeof(a));
  a.src = src;
  a.n = n;
  a.table = t.table.as<ulonglong2>();
  a.special = t.special.as<RadixSpecial>();
  a.pay_sorted = t.pay_sorted.as<unsigned long long>();
  a.vals = t.vals.as<unsigned long long>();
  a.counters = t.counters.as<unsigned long long>();
  a.smask = (1u << t.sub_bits) - 1;
  a.left_outer = jt == CRGPU_LEFT_OUTER;
  if (partitioned) {
    a.prow = t.p_rows.as<ulonglong2>();
    a.pstart = t.p_starts.as<uint32_t>();
    a.pcount = t.p_totals.as<uint32_t>();
    int64_t w = (n / kRadixP1 + kDupTile - 1) / kDupTile;
    if (w < 1) w = 1;
    if (w > kProbeWgPerPart) w = kProbeWgPerPart;
    a.wg_per_part = static_cast<int32_t>(w);
    *grid = kRadixP1 * static_cast<int>(w);
  } else {
    const int64_t tiles = (n + kDupTile - 1) / kDupTile;
    const int64_t cap = static_cast<int64_t>(num_cus()) * 4;
    *grid = static_cast<int>(tiles < cap ? tiles : cap);
  }
}

crgpu_status radix_probe_dup_count(RadixTable &t, int jt, const RadixSrc &src, int64_t n, bool partitioned, uint64_t *out_rows) {
  if (!t.ready || !t.dup) return fail(CRGPU_ERR_STATE, "radix probe: no duplicate-key table built");
  *out_rows = 0;
  if (n <= 0) return CRGPU_OK;
  hipStream_t st = stream();
  if (partitioned)
    CRGPU_TRY(radix_partition(src, n, t.p_hist, t.p_totals, t.p_starts, t.p_rows, 0, "radix_probe_hist", "radix_probe_scatter"));
  CRGPU_TRY(t.vals.ensure(static_cast<size_t>(n + 8 * kRadixP1) * 8));
  CRGPU_TRY(t.counters.ensure(16));
  CRGPU_HIP(hipMemsetAsync(t.counters.p, 0, 16, st));
  RadixDupArgs a;
  int grid;
  dup_args(t, src, n, jt, partitioned, a, &grid);
  {
    ProfScope prof("join_probe");
    if (partitioned) hipLaunchKernelGGL((radix_dup_count_kernel<0, kDupTB, kDupRPT>), dim3(grid), dim3(kDupTB), 0, st, a);
    else hipLaunchKernelGGL((radix_dup_count_kernel<1, kDupTB, kDupRPT>), dim3(grid), dim3(kDupTB), 0, st, a);
    CRGPU_HIP(hipGetLastError());
  }
  unsigned long long c[2] = {0, 0};
  CRGPU_HIP(hipMemcpyAsync(c, t.counters.p, 16, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  *out_rows = c[0];
  // the partition passes left the rows without a key (NULL / outside the build side's range) out: LEFT OUTER keeps them
  if (partitioned && jt == CRGPU_LEFT_OUTER) *out_rows += static_cast<uint64_t>(n) - c[1];
  return CRGPU_OK;
}

crgpu_status radix_probe_dup_emit(RadixTable &t, int jt, const RadixSrc &src, int64_t n, const RadixEmit &emit, bool partitioned) {
  if (!t.ready || !t.dup) return fail(CRGPU_ERR_STATE, "radix probe: no duplicate-key table built");
  if (n <= 0) return CRGPU_OK;
  if (jt != CRGPU_INNER && jt != CRGPU_LEFT_OUTER) return fail(CRGPU_ERR_INVALID, "radix probe: the duplicate-key table serves INNER and LEFT OUTER joins");
  hipStream_t st = stream();
  RadixDupArgs a;
  int grid;
  dup_args(t, src, n, jt, partitioned, a, &grid);
  a.emit = emit;
  {
    ProfScope prof("join_probe");
    if (partitioned && jt == CRGPU_INNER) hipLaunchKernelGGL((radix_dup_emit_kernel<CRGPU_INNER, 0, kDupTB, kDupRPT>), dim3(grid), dim3(kDupTB), 0, st, a);
    else if (partitioned) hipLaunchKernelGGL((radix_dup_emit_kernel<CRGPU_LEFT_OUTER, 0, kDupTB, kDupRPT>), dim3(grid), dim3(kDupTB), 0, st, a);
    else if (jt == CRGPU_INNER) hipLaunchKernelGGL((radix_dup_emit_kernel<CRGPU_INNER, 1, kDupTB, kDupRPT>), dim3(grid), dim3(kDupTB), 0, st, a);
    else hipLaunchKernelGGL((radix_dup_emit_kernel<CRGPU_LEFT_OUTER, 1, kDupTB, kDupRPT>), dim3(grid), dim3(kDupTB), 0, st, a);
    CRGPU_HIP(hipGetLastError());
  }
  if (partitioned && jt == CRGPU_LEFT_OUTER) {  // the rows without a key, as unmatched (radix_probe_kernel's MODE 2 pass)
    bool keyless = src.kp.n > 1;
    for (int c = 0; c < src.kp.n; c++) keyless = keyless || src.kp.c[c].valid != nullptr;
    if (keyless) {
      RadixProbeArgs pa;
      std::memset(&pa, 0, sizeof(pa));
      pa.src = src;
      pa.n = n;
      pa.table = t.table.as<ulonglong2>();
      pa.special = t.special.as<RadixSpecial>();
      pa.smask = (1u << t.sub_bits) - 1;
      pa.sub_bits = t.sub_bits;
      pa.emit = emit;
      const int64_t tiles = (n + kProbeTile - 1) / kProbeTile;
      const int64_t cap = static_cast<int64_t>(num_cus()) * 4;
      CRGPU_TRY(launch_probe<2>(jt, pa, static_cast<int>(tiles < cap ? tiles : cap)));
    }
  }
  return CRGPU_OK;
}

}  // namespace crgpu
