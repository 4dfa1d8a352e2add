// join.hip -- device-resident hash join.
//
// Replaces colexecjoin.hashJoiner (pkg/sql/colexec/colexecjoin/hashjoiner.go:166-712,
// hashjoiner_tmpl.go) and the colexechash.HashTable it drives (hashtable.go,
// hashtable_tmpl.go).  Results are the same multisets; the internals are not
// the reference's First/Next/Same/Visited arrays:
//
//  * PACKED table (single int / float / bool equality column, first match
//    suffices): open addressing over 16-byte {key64, val64} slots, linear
//    probing, 64-bit atomicCAS on the key word at build, one 16-byte load per
//    probe step.  val = build row id, or -- when the build side is {key, one
//    8-byte payload} ("covering") -- the payload itself, so a probe touches
//    exactly one random 16-byte slot and nothing else of the build side.
//    The probe kernel is fused with output materialisation: 1024 lanes x 8
//    rows per workgroup tile, ballot-compacted, one returning atomicAdd per
//    8192-row tile to claim output space.
//  * CHAINED table (multi-column / BYTES keys, duplicate build keys, RIGHT
//    SEMI/ANTI, INTERSECT/EXCEPT ALL): where the reference links rows through
//    First/Next, the build rows are counting-sorted by hash bucket: cstart[nb+1]
//    bucket boundaries, grp[n] row ids bucket by bucket and -- one integer key
//    column -- gkey[n] 16-byte {key value, row id} entries, so a bucket walk reads
//    consecutive memory; for other keys rep8[n] holds, per entry, the offset in its
//    bucket of the first entry with an equal key (its representative: a probe row
//    verifies one entry per distinct key).  The handle's `first` / `next` buffers
//    hold the per-bucket counters and every row's bucket number while the table is
//    built.  Probing is count -> exclusive scan -> fill (deterministic output
//    offsets, no per-row atomics), then columns are materialised through
//    (probe row, build row) pairs.
//
// Roofline: HBM-bound (random 16-byte slot reads + streaming columns).
#include <cstring>

#include <cstdlib>

#include "device.h"
#include "radix.h"

namespace crgpu {

static constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
struct alignas(16) Slot {
  unsigned long long key;
  unsigned long long val;
};
struct Special {  // the one build key whose packed value equals kEmptyKey
  unsigned int present;
  unsigned int _pad;
  unsigned long long val;
};

// Key bitmap in front of a packed table that does not fit the L2 (tables of 2 MiB ... 128 MiB: build sides below the
// radix threshold): one bit per hash value, 16 bits per build key, at most 2 MiB -- L2-resident.  A probe row whose bit
// is clear has no match and never touches the table; a selective join (TPC-H Q9: 5 % of the lineitems find their
// part) then reads ~one random HBM / Infinity-Cache line per SURVIVING row instead of one per row.  The bit index uses
// hash bits the slot index does not.
__device__ __forceinline__ uint32_t bloom_bit(uint64_t h, uint32_t bmask) { return static_cast<uint32_t>(h >> 36) & bmask; }

// ------------------------------------------------------------------ PACKED build
__global__ __launch_bounds__(256) void packed_build_kernel(KeyPack kp, int64_t n, Slot *__restrict__ slots,
                                                           uint64_t mask, const uint64_t *__restrict__ payload,
                                                           Special *__restrict__ special, uint32_t *__restrict__ bloom,
                                                           uint32_t bmask) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < n; row += stride) {
    uint64_t k;
    if (!key_pack(kp, row, k)) continue;  // NULL never matches (allowNullEquality=false, hashjoiner.go:264)
    const unsigned long long v = payload ? payload[row] : static_cast<unsigned long long>(row);
    if (k == kEmptyKey) {
      if (atomicCAS(&special->present, 0u, 1u) == 0u) special->val = v;
      continue;
    }
    const uint64_t hk = fmix64(k);
    if (bloom != nullptr) {
      const uint32_t bit = bloom_bit(hk, bmask);
      atomicOr(&bloom[bit >> 5], 1u << (bit & 31));
    }
    uint64_t idx = hk & mask;
    while (true) {
      unsigned long long old = atomicCAS(&slots[idx].key, kEmptyKey, static_cast<unsigned long long>(k));
      if (old == kEmptyKey) {
        slots[idx].val = v;
        break;
      }
      if (old == k) break;  // duplicate key: first writer wins (distinct build semantics)
      idx = (idx + 1) & mask;
    }
  }
}

// ------------------------------------------------------------------ PACKED probe, fused with materialisation
static constexpr int kMaxFused = 16;
enum CopySide : int32_t { SIDE_PROBE = 0, SIDE_BUILD = 1, SIDE_COVER = 2, SIDE_KEYCAST = 3 };
struct CopyCol {
  const uint8_t *src;
  uint8_t *dst;
  const uint8_t *src_valid;  // validity bitmap of the source (or null)
  uint8_t *dst_valid;        // one byte per output row (or null when the output cannot hold NULLs)
  int32_t width;
  int32_t side;
};
struct ProbeArgs {
  KeyPack kp;
  int64_t n;
  const Slot *slots;
  uint64_t mask;
  const Special *special;
  int32_t ncopy;
  int32_t _pad;
  CopyCol copy[kMaxFused];
  int32_t *pair_p;  // optional (probe row, build row) pairs for columns not fused
  int32_t *pair_b;
  uint8_t *matched;  // optional build-row matched flags (RIGHT/FULL OUTER)
  unsigned long long *out_count;
  const uint32_t *bloom;  // optional key bitmap (bloom_bit): a clear bit = no match, the table is not read
  uint32_t bmask, _pad2;
  const int32_t *list;    // optional: the probe rows to look at (bloom_filter_kernel's survivors); n counts list entries
  const uint8_t *keep;    // optional selection (a BOOL column): rows whose byte is 0 are not part of the probe side
};

__device__ __forceinline__ void store_width(uint8_t *dst, int64_t pos, int width, uint64_t v) {
  switch (width) {
    case 1: dst[pos] = static_cast<uint8_t>(v); break;
    case 2: reinterpret_cast<uint16_t *>(dst)[pos] = static_cast<uint16_t>(v); break;
    case 4: reinterpret_cast<uint32_t *>(dst)[pos] = static_cast<uint32_t>(v); break;
    default: reinterpret_cast<uint64_t *>(dst)[pos] = v; break;
  }
}
__device__ __forceinline__ uint64_t load_width(const uint8_t *src, int64_t i, int width) {
  switch (width) {
    case 1: return src[i];
    case 2: return reinterpret_cast<const uint16_t *>(src)[i];
    case 4: return reinterpret_cast<const uint32_t *>(src)[i];
    default: return reinterpret_cast<const uint64_t *>(src)[i];
  }
}

template <int JT, int TB, int RPT>
__global__ __launch_bounds__(TB) void packed_probe_kernel(ProbeArgs a) {
  constexpr int kWaves = TB / 64;
  constexpr int kTile = TB * RPT;
  __shared__ uint32_t s_cnt[RPT * kWaves];
  __shared__ uint32_t s_pre[RPT * kWaves];
  __shared__ unsigned long long s_base;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int64_t ntiles = (a.n + kTile - 1) / kTile;

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t tbase = tile * kTile;
    uint64_t k[RPT];
    unsigned long long val[RPT];
    bool found[RPT], inb[RPT], emit[RPT];
    uint64_t idx[RPT], hk[RPT];
    int32_t rowl[RPT];  // with a row list: the probe row of list entry tbase + j * TB + tid
    Slot s[RPT];
    // issue every first-step slot load of the tile before resolving any (memory-level parallelism)
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const int64_t at = tbase + static_cast<int64_t>(j) * TB + tid;
      inb[j] = at < a.n && (a.keep == nullptr || a.list != nullptr || a.keep[at] != 0);  // (a list holds selected rows only)
      rowl[j] = a.list != nullptr && inb[j] ? a.list[at] : 0;
      const int64_t row = a.list != nullptr ? static_cast<int64_t>(rowl[j]) : at;
      found[j] = false;
      val[j] = 0;
      k[j] = kEmptyKey;
      uint64_t kk = kEmptyKey;
      const bool probe = inb[j] && key_pack(a.kp, row, kk);
      if (probe) {
        k[j] = kk;
        if (k[j] == kEmptyKey) {
          found[j] = a.special->present != 0;
          val[j] = a.special->val;
        }
      }
      hk[j] = fmix64(k[j]);
      idx[j] = hk[j] & a.mask;
    }
    if (a.bloom != nullptr) {  // all bitmap words of the tile's rows first (L2 hits), then the table for the survivors
      uint32_t w[RPT];
#pragma unroll
      for (int j = 0; j < RPT; j++) w[j] = k[j] != kEmptyKey ? a.bloom[bloom_bit(hk[j], a.bmask) >> 5] : 0u;
#pragma unroll
      for (int j = 0; j < RPT; j++)
        if (k[j] != kEmptyKey && !((w[j] >> (bloom_bit(hk[j], a.bmask) & 31)) & 1u)) k[j] = kEmptyKey;  // (found stays false)
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      if (k[j] != kEmptyKey) {
        const ulonglong2 raw = *reinterpret_cast<const ulonglong2 *>(&a.slots[idx[j]]);
        s[j].key = raw.x;
        s[j].val = raw.y;
      } else {
        s[j].key = kEmptyKey;
        s[j].val = 0;
      }
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      if (k[j] != kEmptyKey) {
        while (true) {
          if (s[j].key == k[j]) {
            found[j] = true;
            val[j] = s[j].val;
            break;
          }
          if (s[j].key == kEmptyKey) break;
          idx[j] = (idx[j] + 1) & a.mask;
          const ulonglong2 raw = *reinterpret_cast<const ulonglong2 *>(&a.slots[idx[j]]);
          s[j].key = raw.x;
          s[j].val = raw.y;
        }
      }
      if (JT == CRGPU_INNER || JT == CRGPU_LEFT_SEMI || JT == CRGPU_RIGHT_OUTER) emit[j] = found[j];
      else if (JT == CRGPU_LEFT_ANTI) emit[j] = inb[j] && !found[j];
      else emit[j] = inb[j];  // LEFT / FULL OUTER keep every probe row
      if (a.matched != nullptr && found[j]) a.matched[val[j]] = 1;
    }
    // output positions: row order within the tile is preserved
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
        int q = lane_id() * kPer + t;
        e[t] = q < RPT * kWaves ? s_cnt[q] : 0;
        sum += e[t];
      }
      uint32_t incl = wave_incl_scan_u32(sum);
      uint32_t ex = incl - sum;
#pragma unroll
      for (int t = 0; t < kPer; t++) {
        int q = lane_id() * kPer + t;
        if (q < RPT * kWaves) s_pre[q] = ex;
        ex += e[t];
      }
      if (lane_id() == 63) s_base = incl ? atomicAdd(a.out_count, static_cast<unsigned long long>(incl)) : 0ull;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    int64_t pos[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) pos[j] = static_cast<int64_t>(base + s_pre[j * kWaves + wave] + rank[j]);

    for (int c = 0; c < a.ncopy; c++) {
      const CopyCol cc = a.copy[c];
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (!emit[j]) continue;
        const int64_t row = a.list != nullptr ? static_cast<int64_t>(rowl[j]) : tbase + static_cast<int64_t>(j) * TB + tid;
        uint64_t v = 0;
        uint32_t present = 1;
        if (cc.side == SIDE_PROBE) {
          v = load_width(cc.src, row, cc.width);
          if (cc.src_valid) present = (cc.src_valid[row >> 3] >> (row & 7)) & 1;
        } else if (!found[j]) {
          present = 0;  // outer join: NULL on the build side (hashjoiner.go:671-682)
        } else if (cc.side == SIDE_BUILD) {
          v = load_width(cc.src, static_cast<int64_t>(val[j]), cc.width);
          if (cc.src_valid) present = (cc.src_valid[val[j] >> 3] >> (val[j] & 7)) & 1;
        } else if (cc.side == SIDE_COVER) {
          v = val[j];
        } else {  // SIDE_KEYCAST: the build key equals the probe key numerically
          v = k[j];
        }
        store_width(cc.dst, pos[j], cc.width, v);
        if (cc.dst_valid) cc.dst_valid[pos[j]] = static_cast<uint8_t>(present);
      }
    }
    if (a.pair_p != nullptr) {
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (!emit[j]) continue;
        a.pair_p[pos[j]] = a.list != nullptr ? rowl[j] : static_cast<int32_t>(tbase + static_cast<int64_t>(j) * TB + tid);
        a.pair_b[pos[j]] = found[j] ? static_cast<int32_t>(val[j]) : -1;
      }
    }
    __syncthreads();  // s_cnt / s_pre / s_base are reused by the next tile
  }
}

// A probe side of which only a few per cent can match (TPC-H Q9: 60M lineitems against the 5 % of parts that are green)
// is mostly rows the key bitmap rejects, and the probe kernel is a heavy place to reject them: ~80 VGPRs, two barriers
// and an output-position atomic per tile whatever the rows do.  This pass only evaluates the bitmap -- key columns in,
// the surviving row numbers out -- and the probe kernel then runs over the survivors alone (ProbeArgs::list).
__global__ __launch_bounds__(1024) void bloom_filter_kernel(KeyPack kp, int64_t n, const uint8_t *__restrict__ sel,
                                                            const uint32_t *__restrict__ bloom, uint32_t bmask,
                                                            int32_t *__restrict__ list, unsigned long long *__restrict__ count) {
  constexpr int TB = 1024, RPT = 8, kWaves = TB / 64, kTile = TB * RPT;  // 8192-row tiles: one reservation atomic each
  __shared__ uint32_t s_cnt[RPT * kWaves];
  __shared__ unsigned long long s_base;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int64_t ntiles = (n + kTile - 1) / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t tbase = tile * kTile;
    uint32_t bit[RPT], w[RPT];
    bool keep[RPT], look[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const int64_t row = tbase + static_cast<int64_t>(j) * TB + tid;
      uint64_t kk = kEmptyKey;
      const bool has = row < n && (sel == nullptr || sel[row] != 0) && key_pack(kp, row, kk);
      keep[j] = has && kk == kEmptyKey;  // the key equal to the empty marker lives beside the table: let the probe decide
      look[j] = has && kk != kEmptyKey;
      bit[j] = bloom_bit(fmix64(kk), bmask);
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) w[j] = look[j] ? bloom[bit[j] >> 5] : 0u;
    uint32_t rank[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      keep[j] = keep[j] || (look[j] && ((w[j] >> (bit[j] & 31)) & 1u));
      const uint64_t m = __ballot(keep[j]);
      rank[j] = rank_in_mask(m);
      if (lane_id() == 0) s_cnt[j * kWaves + wave] = popc64(m);
    }
    __syncthreads();
    if (tid < 64) {  // RPT * kWaves = 128 counts: exclusive scan by the first wave (two each), one reservation per tile
      const uint32_t c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
      const uint32_t incl = wave_incl_scan_u32(c0 + c1);
      const uint32_t total = __shfl(incl, 63, 64);
      unsigned long long base = 0;
      if (tid == 0 && total) base = atomicAdd(count, static_cast<unsigned long long>(total));
      if (tid == 0) s_base = base;
      s_cnt[2 * tid] = incl - c0 - c1;
      s_cnt[2 * tid + 1] = incl - c1;
    }
    __syncthreads();
    const unsigned long long base = s_base;
#pragma unroll
    for (int j = 0; j < RPT; j++)
      if (keep[j]) list[base + s_cnt[j * kWaves + wave] + rank[j]] = static_cast<int32_t>(tbase + static_cast<int64_t>(j) * TB + tid);
    __syncthreads();  // s_cnt / s_base are reused by the next tile
  }
}

// The radix path's counterpart: a bitmap over the FIRST key column's value range (bit d = some build row has
// value lo + d).  TPC-H's keys are dense integers, so the bitmap of Q9's 20 M part keys is 2.5 MB and stays in the L2s,
// and a pass over ONE probe column rejects the 95 % of the lineitems whose part is not green before anything is
// scattered: the radix probe then partitions and probes the survivors only.  (A superset test: the other key columns
// are compared by the join itself.)
__global__ __launch_bounds__(256) void key0_bitmap_build_kernel(DCol c0, long long lo, unsigned long long range, int64_t n,
                                                                uint32_t *__restrict__ bitmap) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (col_is_null(c0, i)) continue;
    const uint64_t d = static_cast<uint64_t>(col_get_ival(c0, i)) - static_cast<uint64_t>(lo);
    // (the range was taken over the rows that HAVE a key: a row with a NULL in another key column may lie outside it,
    // and matches nothing)
    if (d <= range) atomicOr(&bitmap[d >> 5], 1u << (d & 31));
  }
}
// FAST: the column is INT64 without NULLs and 16-byte aligned: two rows per vector load.  The survivors' values go out
// beside their row numbers (`vals`: the compacted first key column, so that only the other key columns are gathered).
template <bool FAST>
__global__ __launch_bounds__(1024) void key0_filter_kernel(DCol c0, long long lo, unsigned long long range, int64_t n,
                                                           const uint32_t *__restrict__ bitmap, int32_t *__restrict__ list,
                                                           long long *__restrict__ vals, unsigned long long *__restrict__ count) {
  constexpr int TB = 1024, RPT = 8, kWaves = TB / 64, kTile = TB * RPT;  // 8192-row tiles: one reservation atomic each
  __shared__ uint32_t s_cnt[RPT * kWaves];
  __shared__ unsigned long long s_base;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int64_t ntiles = (n + kTile - 1) / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t tbase = tile * kTile;
    uint64_t d[RPT];
    uint32_t w[RPT];
    bool look[RPT];
    int64_t row[RPT];
    if (FAST) {
#pragma unroll
      for (int j = 0; j < RPT; j += 2) {  // rows 2q, 2q + 1 of the tile, q = (j / 2) * TB + tid
        row[j] = tbase + 2 * (static_cast<int64_t>(j / 2) * TB + tid);
        row[j + 1] = row[j] + 1;
        long long a = 0, b = 0;
        if (row[j + 1] < n) {
          typedef long long ll2v __attribute__((ext_vector_type(2)));
          const ll2v v = __builtin_nontemporal_load(reinterpret_cast<const ll2v *>(c0.data) + (row[j] >> 1));
          a = v.x, b = v.y;
        } else if (row[j] < n) {
          a = static_cast<const long long *>(c0.data)[row[j]];
        }
        d[j] = static_cast<uint64_t>(a) - static_cast<uint64_t>(lo);
        d[j + 1] = static_cast<uint64_t>(b) - static_cast<uint64_t>(lo);
        look[j] = row[j] < n && d[j] <= range;
        look[j + 1] = row[j + 1] < n && d[j + 1] <= range;
      }
    } else {
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        row[j] = tbase + static_cast<int64_t>(j) * TB + tid;
        look[j] = row[j] < n && !col_is_null(c0, row[j]);
        d[j] = look[j] ? static_cast<uint64_t>(col_get_ival(c0, row[j])) - static_cast<uint64_t>(lo) : 0;
        look[j] = look[j] && d[j] <= range;
      }
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) w[j] = look[j] ? bitmap[d[j] >> 5] : 0u;
    uint32_t rank[RPT];
    bool keep[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      keep[j] = look[j] && ((w[j] >> (d[j] & 31)) & 1u);
      const uint64_t m = __ballot(keep[j]);
      rank[j] = rank_in_mask(m);
      if (lane_id() == 0) s_cnt[j * kWaves + wave] = popc64(m);
    }
    __syncthreads();
    if (tid < 64) {
      const uint32_t c0n = s_cnt[2 * tid], c1n = s_cnt[2 * tid + 1];
      const uint32_t incl = wave_incl_scan_u32(c0n + c1n);
      const uint32_t total = __shfl(incl, 63, 64);
      unsigned long long base = 0;
      if (tid == 0 && total) base = atomicAdd(count, static_cast<unsigned long long>(total));
      if (tid == 0) s_base = base;
      s_cnt[2 * tid] = incl - c0n - c1n;
      s_cnt[2 * tid + 1] = incl - c1n;
    }
    __syncthreads();
    const unsigned long long base = s_base;
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      if (keep[j]) {
        const unsigned long long at = base + s_cnt[j * kWaves + wave] + rank[j];
        list[at] = static_cast<int32_t>(row[j]);
        vals[at] = static_cast<long long>(d[j] + static_cast<uint64_t>(lo));
      }
    }
    __syncthreads();
  }
}
// How many of `samples` evenly spaced probe rows pass the bitmap: the full pass is only worth it if most do not.
__global__ __launch_bounds__(256) void key0_sample_kernel(DCol c0, long long lo, unsigned long long range, int64_t n, int64_t samples,
                                                          const uint32_t *__restrict__ bitmap, unsigned long long *__restrict__ hits) {
  const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  bool hit = false;
  if (k < samples) {
    const int64_t i = k * (n / samples);
    if (!col_is_null(c0, i)) {
      const uint64_t d = static_cast<uint64_t>(col_get_ival(c0, i)) - static_cast<uint64_t>(lo);
      hit = d <= range && ((bitmap[d >> 5] >> (d & 31)) & 1u);
    }
  }
  const uint64_t m = __ballot(hit);
  if (lane_id() == 0 && m != 0) atomicAdd(hits, static_cast<unsigned long long>(popc64(m)));
}
__global__ __launch_bounds__(256) void widen_i32_i64_kernel(const int32_t *__restrict__ in, int64_t n, long long *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

// ------------------------------------------------------------------ CHAINED build / probe
// The chained table is laid out bucket by bucket (a counting sort of the build rows by hash bucket): start[b] .. start[b+1]
// delimits bucket b in grp[] (build row ids) and, for a single integer key column, gkey[] ({key value, row id} pairs).
// A probe row then reads one start pair and a few CONSECUTIVE gkey entries: ~2 random cache lines, where a linked chain
// costs two (next[] and the key column) per candidate.
__global__ __launch_bounds__(256) void chain_hist_kernel(KeyCols keys, int64_t n, bool null_eq, uint32_t nb_mask,
                                                         uint32_t *__restrict__ cnt, uint32_t *__restrict__ hsave) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < n; row += stride) {
    if (!null_eq && any_key_null(keys, row)) {
      hsave[row] = 0xFFFFFFFFu;  // never matches: not in the table
      continue;
    }
    const uint32_t h = hash_row32(keys, row, 1u) & nb_mask;
    hsave[row] = h;
    atomicAdd(&cnt[h], 1u);
  }
}
__global__ __launch_bounds__(256) void chain_narrow_kernel(const uint64_t *__restrict__ in, int64_t n,
                                                          uint32_t *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = static_cast<uint32_t>(in[i]);
}
__global__ __launch_bounds__(256) void chain_scatter_kernel(KeyPack kp, int64_t n, const uint32_t *__restrict__ hsave,
                                                            const uint32_t *__restrict__ start,
                                                            uint32_t *__restrict__ cnt, uint32_t *__restrict__ grp,
                                                            ulonglong2 *__restrict__ gkey) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < n; row += stride) {
    const uint32_t h = hsave[row];
    if (h == 0xFFFFFFFFu) continue;
    const uint32_t pos = start[h] + (atomicSub(&cnt[h], 1u) - 1u);  // cnt counts back down to zero
    grp[pos] = static_cast<uint32_t>(row);
    if (gkey != nullptr) {
      uint64_t k = 0;
      (void)key_pack(kp, row, k);  // rows without a key were left out by chain_hist_kernel (hsave = ~0)
      gkey[pos] = make_ulonglong2(k, static_cast<uint64_t>(row));
    }
  }
}

static constexpr int kChainCache = 8;  // matches per probe row that pass 1 hands to pass 3
static constexpr int kChainChunk = 4;  // bucket entries requested per round trip

// Generic keys (several columns, BYTES, floats): equality is decided on the build columns, a random read per column and
// candidate - and most candidates of a bucket are duplicates of one key.  Once per build, every bucket entry among the
// first kRepSpan of its bucket learns the offset of the FIRST entry with an equal key (its representative; itself if
// there is none); a probe row then verifies representatives only and lets the others follow them.  kRepSelf = verify
// this entry itself (entries beyond the span).
static constexpr uint32_t kRepSpan = 64;
static constexpr uint8_t kRepSelf = 255;
__global__ __launch_bounds__(256) void chain_mark_kernel(KeyCols keys, int64_t nb, bool null_eq,
                                                         const uint32_t *__restrict__ start,
                                                         const uint32_t *__restrict__ grp, uint8_t *__restrict__ rep8) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b < nb; b += stride) {
    const uint32_t s = start[b], e = start[b + 1];
    uint64_t reps = 0;  // offsets (< kRepSpan) that are representatives
    for (uint32_t j = s; j < e; j++) {
      const uint32_t off = j - s;
      if (off >= kRepSpan) {
        rep8[j] = kRepSelf;
        continue;
      }
      uint32_t rep = off;
      const int64_t rj = grp[j];
      for (uint64_t m = reps; m != 0; m &= m - 1) {
        const uint32_t q = static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(m))) - 1u;
        if (keys_equal(keys, rj, keys, static_cast<int64_t>(grp[s + q]), null_eq)) {
          rep = q;
          break;
        }
      }
      if (rep == off) reps |= 1ull << off;
      rep8[j] = static_cast<uint8_t>(rep);
    }
  }
}

struct ChainArgs {
  KeyCols probe, build;
  int64_t n;
  const uint32_t *start;  // [nb + 1] bucket boundaries in grp / gkey (row ids are 32-bit, so are these)
  const uint32_t *grp;    // build row ids, bucket by bucket
  const uint8_t *rep8;    // generic keys: offset in the bucket of each entry's representative (chain_mark_kernel), or nullptr
  const ulonglong2 *gkey; // {packed key word, row id} in the same order (keys that pack into 64 bits, KeyPack), else nullptr
  KeyPack kp;             // probe-side key word (when gkey is in use)
  uint32_t nb_mask;
  int32_t null_eq;
  uint8_t *matched;       // build-row matched flags (RIGHT/FULL OUTER, RIGHT SEMI/ANTI)
  unsigned int *claimed;  // "deleted" marks for INTERSECT/EXCEPT ALL (HashTableDeletingProbeMode, hashtable.go:46-58)
  uint32_t *counts;       // pass 1 output
  const uint64_t *offsets;  // pass 2 input
  int32_t *pair_p, *pair_b;
  uint4 *mcache;          // [kChainCache / 4][n] or nullptr: the first matches of every probe row, left by pass 1 for pass 3
};

// Iteration over the build rows whose key equals probe row i's: `ChainIter it = chain_begin(a, i); while (chain_next(a, i,
// it, r)) ...`.
struct ChainIter {
  uint32_t p, e;
  uint64_t k;
};
__device__ __forceinline__ ChainIter chain_begin(const ChainArgs &a, int64_t i) {
  const uint32_t h = hash_row32(a.probe, i, 1u) & a.nb_mask;
  ChainIter it;
  it.p = a.start[h];
  it.e = a.start[h + 1];
  it.k = 0;
  if (a.gkey != nullptr && !key_pack(a.kp, i, it.k)) it.e = it.p;  // outside the build side's key range: no match
  return it;
}
__device__ __forceinline__ bool chain_next(const ChainArgs &a, int64_t i, ChainIter &it, int64_t &r) {
  while (it.p < it.e) {
    const uint32_t p = it.p++;
    if (a.gkey != nullptr) {
      const ulonglong2 kr = a.gkey[p];  // key and row id in one 16-byte load
      if (kr.x != it.k) continue;
      r = static_cast<int64_t>(kr.y);
      return true;
    }
    r = static_cast<int64_t>(a.grp[p]);
    if (keys_equal(a.probe, i, a.build, r, a.null_eq != 0)) return true;
  }
  return false;
}

// One chunk of a generic-key bucket walk: loads kChainChunk row ids (rr) and their representative offsets, verifies the
// entries that stand for themselves against probe row i (keys_equal_chunk) and lets the others take their
// representative's verdict from `okmask`; leaves eq[t] = "entry p + t matches".  Expects a, i, rr, eq, okmask in scope.
#define CHAIN_GENERIC_CHUNK(P, S0, E)                                                                              \
  {                                                                                                                \
    uint32_t rp_[kChainChunk];                                                                                     \
    bool self_[kChainChunk];                                                                                       \
    _Pragma("unroll") for (int t = 0; t < kChainChunk; t++) {                                                      \
      const uint32_t q_ = (P) + t < (E) ? (P) + t : (E)-1;                                                         \
      rr[t] = a.grp[q_];                                                                                           \
      rp_[t] = a.rep8 != nullptr ? a.rep8[q_] : kRepSelf;                                                          \
    }                                                                                                              \
    _Pragma("unroll") for (int t = 0; t < kChainChunk; t++) {                                                      \
      self_[t] = rp_[t] == kRepSelf || rp_[t] == (P) + t - (S0);                                                   \
      eq[t] = (P) + t < (E) && self_[t];                                                                           \
    }                                                                                                              \
    keys_equal_chunk<kChainChunk>(a.probe, i, a.build, rr, a.null_eq != 0, eq);                                    \
    _Pragma("unroll") for (int t = 0; t < kChainChunk; t++) {                                                      \
      if ((P) + t < (E)) {                                                                                         \
        if (!self_[t]) eq[t] = (okmask >> rp_[t]) & 1ull;                                                          \
        else if (eq[t] && rp_[t] != kRepSelf) okmask |= 1ull << rp_[t];                                            \
      }                                                                                                            \
    }                                                                                                              \
  }

// pass 1: number of output rows each probe row produces (and all side effects:
// matched marks, claims).  `it` is the row's started bucket walk (empty for a row whose NULL key matches nothing).
template <int JT>
__device__ __forceinline__ void chain_count_row(const ChainArgs &a, int64_t i, ChainIter it) {
  uint32_t cnt = 0;
  bool found = false;
  // The first kChainCache matches are kept for pass 3 (coalesced 16-byte stores here, loads there), so only rows
  // with more matches walk their bucket twice.  0xFFFFFFFF in the first field = no match (the NULL build side of a
  // LEFT/FULL OUTER row).
  uint32_t m[kChainCache];
#pragma unroll
  for (int t = 0; t < kChainCache; t++) m[t] = t == 0 ? 0xFFFFFFFFu : 0u;
  // one match; false = this row needs no more
  auto on_match = [&](int64_t r) -> bool {
    if (JT == CRGPU_INTERSECT_ALL || JT == CRGPU_EXCEPT_ALL) {
      if (atomicCAS(&a.claimed[r], 0u, 1u) == 0u) {
        found = true;
        return false;
      }
      return true;
    }
    found = true;
#pragma unroll
    for (int t = 0; t < kChainCache; t++)
      if (cnt == static_cast<uint32_t>(t)) m[t] = static_cast<uint32_t>(r);
    cnt++;
    if (a.matched != nullptr) a.matched[r] = 1;
    return !(JT == CRGPU_LEFT_SEMI || JT == CRGPU_LEFT_ANTI);
  };
  if (a.gkey != nullptr) {
    // The entries of a bucket sit in one or two cache lines, but a load per entry, each waited for, still costs a
    // cache round trip apiece: kChainChunk entries are requested at once.
    bool go = true;
    for (uint32_t p = it.p; go && p < it.e; p += kChainChunk) {
      ulonglong2 e[kChainChunk];
#pragma unroll
      for (int t = 0; t < kChainChunk; t++) e[t] = a.gkey[p + t < it.e ? p + t : it.e - 1];
#pragma unroll
      for (int t = 0; t < kChainChunk; t++)
        if (go && p + t < it.e && e[t].x == it.k) go = on_match(static_cast<int64_t>(e[t].y));
    }
  } else {
    bool go = true;
    uint64_t okmask = 0;  // representatives (by offset in the bucket) that equal this row's key
    for (uint32_t p = it.p; go && p < it.e; p += kChainChunk) {
      uint32_t rr[kChainChunk];
      bool eq[kChainChunk];
      CHAIN_GENERIC_CHUNK(p, it.p, it.e)
#pragma unroll
      for (int t = 0; t < kChainChunk; t++)
        if (go && eq[t]) go = on_match(static_cast<int64_t>(rr[t]));
    }
  }
  uint32_t out;
  switch (JT) {
    case CRGPU_INNER:
    case CRGPU_RIGHT_OUTER: out = cnt; break;
    case CRGPU_LEFT_OUTER:
    case CRGPU_FULL_OUTER: out = cnt ? cnt : 1; break;
    case CRGPU_LEFT_SEMI:
    case CRGPU_INTERSECT_ALL: out = found; break;
    case CRGPU_LEFT_ANTI:
    case CRGPU_EXCEPT_ALL: out = !found; break;
    default: out = 0; break;  // RIGHT SEMI / RIGHT ANTI emit nothing while probing
  }
  a.counts[i] = out;
  if (a.mcache != nullptr) {
#pragma unroll
    for (int t = 0; t < kChainCache / 4; t++)  // plane t holds matches 4t .. 4t+3 of every row
      if (t == 0 || cnt > 4u * t) a.mcache[static_cast<int64_t>(t) * a.n + i] = make_uint4(m[4 * t], m[4 * t + 1], m[4 * t + 2], m[4 * t + 3]);
  }
}

// A bucket walk is a chain of dependent random reads (key -> start pair -> entries), so the kernel is bound by how many
// of them are in flight: every lane starts the walks of kChainRows rows before it follows any of them.
static constexpr int kChainRows = 2;
template <int JT>
__global__ __launch_bounds__(256) void chain_count_kernel(ChainArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i0 < a.n; i0 += stride * kChainRows) {
    ChainIter it[kChainRows];
#pragma unroll
    for (int t = 0; t < kChainRows; t++) {
      const int64_t i = i0 + t * stride;
      it[t].p = it[t].e = 0;
      it[t].k = 0;
      if (i < a.n && (a.null_eq || !any_key_null(a.probe, i))) it[t] = chain_begin(a, i);
    }
#pragma unroll
    for (int t = 0; t < kChainRows; t++) {
      const int64_t i = i0 + t * stride;
      if (i < a.n) chain_count_row<JT>(a, i, it[t]);
    }
  }
}

// pass 3: write the (probe row, build row) pairs at the scanned offsets.
template <int JT>
__global__ __launch_bounds__(256) void chain_fill_kernel(ChainArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    const uint32_t cnt = a.counts[i];
    if (cnt == 0) continue;
    uint64_t o = a.offsets[i];
    constexpr bool kPairs = JT == CRGPU_INNER || JT == CRGPU_LEFT_OUTER || JT == CRGPU_RIGHT_OUTER || JT == CRGPU_FULL_OUTER;
    if (!kPairs) {
      a.pair_p[o] = static_cast<int32_t>(i);
      continue;
    }
    uint32_t written = 0;
    if (a.null_eq || !any_key_null(a.probe, i)) {
      ChainIter it = chain_begin(a, i);
      int64_t r;
      while (written < cnt && chain_next(a, i, it, r)) {
        a.pair_p[o + written] = static_cast<int32_t>(i);
        a.pair_b[o + written] = static_cast<int32_t>(r);
        written++;
      }
    }
    if (written == 0) {  // unmatched row of a LEFT/FULL OUTER join
      a.pair_p[o] = static_cast<int32_t>(i);
      a.pair_b[o] = -1;
    }
  }
}

// Orders a wave's LDS writes before its later LDS reads by other lanes of the SAME wave (a wave's LDS instructions execute
// in order; the fences keep the compiler from moving them across this point).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// pass 3 of the pair-producing join types when pass 1 left its match cache: a wave expands its 64 probe rows into their
// CONSECUTIVE output rows cooperatively - lane q of each round finds the probe row that owns output row q by a binary
// search over the wave's exclusive counts in LDS - so pair_p / pair_b are written 256 bytes per store instruction
// whatever the match counts are (one lane writing its own row's 1..4 pairs strides the stores 4..16 bytes apart and
// instruction).  Rows with more than kChainCache matches walk their bucket again.
__global__ __launch_bounds__(256) void chain_fill_cached_kernel(ChainArgs a) {
  __shared__ uint32_t s_off[4][65];
  __shared__ __align__(16) uint32_t s_m[4][64 * kChainCache];
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t bbase = static_cast<int64_t>(blockIdx.x) * 256; bbase < a.n; bbase += stride) {
    const int64_t wbase = bbase + wave * 64;
    const int64_t i = wbase + lane;
    const bool in = i < a.n;
    const uint32_t cnt = in ? a.counts[i] : 0u;
    const uint64_t o = in ? a.offsets[i] : 0ull;
    const uint32_t incl = wave_incl_scan_u32(cnt);
    const uint32_t total = __shfl(incl, 63, 64);
    const uint64_t o0 = (static_cast<uint64_t>(__shfl(static_cast<uint32_t>(o >> 32), 0, 64)) << 32) |
                        __shfl(static_cast<uint32_t>(o), 0, 64);
    s_off[wave][lane] = incl - cnt;
    if (lane == 63) s_off[wave][64] = total;
    if (in) {  // not conditional on cnt: these loads go out together with the two above (a plane the row did not write is never used)
#pragma unroll
      for (int t = 0; t < kChainCache / 4; t++)
        *reinterpret_cast<uint4 *>(&s_m[wave][lane * kChainCache + 4 * t]) = a.mcache[static_cast<int64_t>(t) * a.n + i];
    }
    wave_lds_sync();  // s_off / s_m rows are private to the wave: no block barrier, a wave with a long walk holds up nobody
    for (uint32_t q = lane; q < total; q += 64) {
      uint32_t src = 0;
#pragma unroll
      for (uint32_t step = 32; step != 0; step >>= 1)
        if (s_off[wave][src + step] <= q) src += step;  // the last row whose first output row is <= q owns q
      const uint32_t first = s_off[wave][src];
      if (s_off[wave][src + 1] - first <= kChainCache) {
        a.pair_p[o0 + q] = static_cast<int32_t>(wbase + src);
        a.pair_b[o0 + q] = static_cast<int32_t>(s_m[wave][src * kChainCache + (q - first)]);  // 0xFFFFFFFF -> -1: no match
      }
    }
    if (cnt > kChainCache) {
      uint32_t written = 0;
      ChainIter it = chain_begin(a, i);
      if (a.gkey != nullptr) {
        for (uint32_t p = it.p; written < cnt && p < it.e; p += kChainChunk) {
          ulonglong2 e[kChainChunk];
#pragma unroll
          for (int t = 0; t < kChainChunk; t++) e[t] = a.gkey[p + t < it.e ? p + t : it.e - 1];
#pragma unroll
          for (int t = 0; t < kChainChunk; t++) {
            if (written < cnt && p + t < it.e && e[t].x == it.k) {
              a.pair_p[o + written] = static_cast<int32_t>(i);
              a.pair_b[o + written] = static_cast<int32_t>(e[t].y);
              written++;
            }
          }
        }
      } else {
        uint64_t okmask = 0;
        for (uint32_t p = it.p; written < cnt && p < it.e; p += kChainChunk) {
          uint32_t rr[kChainChunk];
          bool eq[kChainChunk];
          CHAIN_GENERIC_CHUNK(p, it.p, it.e)
#pragma unroll
          for (int t = 0; t < kChainChunk; t++) {
            if (written < cnt && eq[t]) {
              a.pair_p[o + written] = static_cast<int32_t>(i);
              a.pair_b[o + written] = static_cast<int32_t>(rr[t]);
              written++;
            }
          }
        }
      }
    }
    wave_lds_sync();  // the next round overwrites s_off / s_m
  }
}

// Build-side [min, max] of every key column over the rows that have a key (no NULL key column): mm[2c] = min, mm[2c+1] = max.
__global__ __launch_bounds__(256) void key_minmax_kernel(KeyPack kp, int64_t n, long long *__restrict__ mm) {
  long long lo[kMaxPackCols], hi[kMaxPackCols];
#pragma unroll
  for (int c = 0; c < kMaxPackCols; c++) lo[c] = 0x7FFFFFFFFFFFFFFFll, hi[c] = static_cast<long long>(0x8000000000000000ull);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < n; row += stride) {
    bool null = false;
#pragma unroll
    for (int c = 0; c < kMaxPackCols; c++)
      if (c < kp.n) null = null || col_is_null(kp.c[c], row);
    if (null) continue;
#pragma unroll
    for (int c = 0; c < kMaxPackCols; c++) {
      if (c < kp.n) {
        const long long v = col_get_ival(kp.c[c], row);
        lo[c] = v < lo[c] ? v : lo[c];
        hi[c] = v > hi[c] ? v : hi[c];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < kMaxPackCols; c++) {
    if (c < kp.n) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const long long l2 = __shfl_xor(lo[c], d, 64), h2 = __shfl_xor(hi[c], d, 64);
        lo[c] = l2 < lo[c] ? l2 : lo[c];
        hi[c] = h2 > hi[c] ? h2 : hi[c];
      }
      if (lane_id() == 0 && lo[c] <= hi[c]) {
        atomicMin(&mm[2 * c], lo[c]);
        atomicMax(&mm[2 * c + 1], hi[c]);
      }
    }
  }
}

// [min, max] of an INT64 column without NULLs (16-byte aligned): vector loads, one atomic pair per wave
typedef long long ll2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void i64_minmax_kernel(const ll2v *__restrict__ col, int64_t n, long long *__restrict__ mm) {
  long long lo = 0x7FFFFFFFFFFFFFFFll, hi = static_cast<long long>(0x8000000000000000ull);
  const int64_t pairs = n >> 1, stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < pairs; i += 4 * stride) {
    ll2v v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = __builtin_nontemporal_load(col + i + u * stride);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      lo = v[u].x < lo ? v[u].x : lo, lo = v[u].y < lo ? v[u].y : lo;
      hi = v[u].x > hi ? v[u].x : hi, hi = v[u].y > hi ? v[u].y : hi;
    }
  }
  for (; i < pairs; i += stride) {
    const ll2v v = col[i];
    lo = v.x < lo ? v.x : lo, lo = v.y < lo ? v.y : lo;
    hi = v.x > hi ? v.x : hi, hi = v.y > hi ? v.y : hi;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const long long v = reinterpret_cast<const long long *>(col)[n - 1];
    lo = v < lo ? v : lo, hi = v > hi ? v : hi;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const long long l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  // one atomic pair per WORKGROUP: same-address atomics serialise at ~10 ns each
  __shared__ long long s_lo[8], s_hi[8];
  if (lane_id() == 0) s_lo[threadIdx.x >> 6] = lo, s_hi[threadIdx.x >> 6] = hi;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 8; w++) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
    }
    if (lo <= hi) {
      atomicMin(&mm[0], lo);
      atomicMax(&mm[1], hi);
    }
  }
}

__global__ __launch_bounds__(256) void flag_eq_kernel(const uint8_t *__restrict__ in, int64_t n, uint8_t want,
                                                      uint8_t *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = in[i] == want;
}

static inline int grid_for(int64_t n, int block) {
  int64_t want = (n + block - 1) / block;
  int64_t cap = static_cast<int64_t>(num_cus()) * 8;
  if (want < 1) want = 1;
  return static_cast<int>(want < cap ? want : cap);
}
static inline uint64_t next_pow2(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

}  // namespace crgpu

using namespace crgpu;

// ====================================================================== handle
struct crgpu_join : crgpu::Handle {
  int join_type = 0;
  bool right_distinct = false;      // after the MakeHashJoinerSpec overrides
  bool track_build_matches = false;
  bool null_eq = false;             // INTERSECT/EXCEPT ALL treat NULLs as equal (hashjoiner.go:264-268)
  std::vector<int> left_types, right_types, left_eq, right_eq;
  bool packed = false;          // decided at build: the keys pack into one word and the first match decides the result
  bool pack_candidate = false;  // the key columns could pack (fixed-width families; several columns: integers / BOOL only)
  bool first_match = false, jt_packed_ok = false;
  KeyPack kp;                   // build-side key word (columns + the measured ranges); probe sides reuse lo / range / shift
  DevBuf minmax;
  // Output schema after the pushed-down projection: one entry per emitted column.
  struct OutSrc {
    bool is_left;
    int src;  // ordinal within its side
  };
  std::vector<OutSrc> out_src;

  std::vector<DCol> build_cols;
  int64_t n_build = 0;
  bool built = false;

  // packed table
  DevBuf slots, special;
  DevBuf bloom;             // key bitmap in front of a packed table larger than the L2 (bloom_bit)
  DevBuf cand;              // bloom_filter_kernel's survivors (probe row numbers) of the current probe chunk
  DevBuf sel_idx;           // crgpu_dev_join_probe_selected on the partitioned / chained tables: the selected rows, copied out
  std::vector<OutCol> sel_cols;
  uint32_t bloom_mask = 0;
  bool bloom_on = false;    // built with the table; switched off after a probe chunk in which most rows matched
  // radix path: bitmap over the first key column's value range (key0_bitmap_build_kernel), the survivors' compacted key /
  // payload / row-id columns of the current probe chunk
  DevBuf key0_bits;
  long long key0_lo = 0;
  unsigned long long key0_range = 0;
  bool key0_on = false;
  bool key0_pending = false;  // eligible, not built yet: the first probe chunk large enough to be filtered builds it
  OutCol pf_key[kMaxPackCols], pf_pay;
  DevBuf pf_rowid, pf_key0;
  DevBuf mcache;  // chained probe: first matches of every probe row (pass 1 -> pass 3)
  uint64_t cap = 0;
  bool covering = false;
  int cover_col = -1;
  // radix-partitioned table (large unique-key build sides, radix.h)
  RadixTable radix;
  bool use_radix = false;
  bool use_radix_dup = false;  // INNER / LEFT OUTER over a non-distinct build side: the grouped radix table (radix_build_dup)
  // chained table
  DevBuf first, next;          // chained table: per-bucket counters (zero again once built), per-row bucket numbers
  DevBuf cstart, grp, gkey, rep8;    // bucket boundaries, row ids and (single integer key) key values bucket by bucket
  bool chain_exact = false;    // gkey is in use: equal key word <=> equal key
  uint32_t nb = 0;
  // marks
  DevBuf matched, claimed;
  // scratch + outputs
  DevBuf scratch, counts, offsets, pair_p, pair_b, counter, flags, null_data;
  std::vector<OutCol> out;
  std::vector<DevBuf> valid_bytes;  // per output column: one byte per row, packed after the fused kernel
  std::vector<crgpu_dcol> views;
};

static bool left_in_output(int jt) { return !(jt == CRGPU_RIGHT_SEMI || jt == CRGPU_RIGHT_ANTI); }
static bool right_in_output(int jt) {
  return !(jt == CRGPU_LEFT_SEMI || jt == CRGPU_LEFT_ANTI || jt == CRGPU_INTERSECT_ALL || jt == CRGPU_EXCEPT_ALL);
}
static bool left_or_full_outer(int jt) { return jt == CRGPU_LEFT_OUTER || jt == CRGPU_FULL_OUTER; }

static int key_family(int t) {
  if (is_int_type(t)) return 1;
  return 10 + t;
}

extern "C" crgpu_status crgpu_dev_join_create(const crgpu_join_spec *spec, crgpu_join **out) {
  CRGPU_TRY(ensure_init());
  if (spec == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_create: null argument");
  if (spec->join_type < CRGPU_INNER || spec->join_type > CRGPU_RIGHT_ANTI)
    return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_create: unknown join type");
  if (spec->n_eq < 1 || spec->n_eq > kMaxKeyCols)
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_join_create: between 1 and 8 equality columns are supported");
  if (spec->n_left_cols < 0 || spec->n_right_cols < 1 || spec->n_left_cols > kMaxCols || spec->n_right_cols > kMaxCols)
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_join_create: column count out of range");
  auto j = new crgpu_join();
  {
    crgpu_status _os = j->open();
    if (_os != CRGPU_OK) { delete j; return _os; }
  }
  CRGPU_ENTER(j);
  j->join_type = spec->join_type;
  for (int i = 0; i < spec->n_left_cols; i++) j->left_types.push_back(spec->left_types[i]);
  for (int i = 0; i < spec->n_right_cols; i++) j->right_types.push_back(spec->right_types[i]);
  for (int i = 0; i < spec->n_eq; i++) {
    j->left_eq.push_back(spec->left_eq[i]);
    j->right_eq.push_back(spec->right_eq[i]);
  }
  for (int t : j->left_types)
    if (!valid_type(t)) { delete j; return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_join_create: unsupported left column type"); }
  for (int t : j->right_types)
    if (!valid_type(t)) { delete j; return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_join_create: unsupported right column type"); }
  for (int i = 0; i < spec->n_eq; i++) {
    int l = j->left_eq[i], r = j->right_eq[i];
    if (l < 0 || l >= spec->n_left_cols || r < 0 || r >= spec->n_right_cols) {
      delete j;
      return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_create: equality column ordinal out of range");
    }
    // same family, or integers of mixed widths (hashtable_tmpl.go:275-281)
    if (key_family(j->left_types[l]) != key_family(j->right_types[r])) {
      delete j;
      return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_join_create: equality columns of different type families");
    }
  }
  // MakeHashJoinerSpec (hashjoiner.go:827-883)
  int jt = j->join_type;
  bool rd = spec->right_eq_cols_are_key != 0;
  if (jt == CRGPU_LEFT_SEMI) rd = true;
  else if (jt == CRGPU_LEFT_ANTI || jt == CRGPU_RIGHT_ANTI || jt == CRGPU_RIGHT_SEMI || jt == CRGPU_INTERSECT_ALL ||
           jt == CRGPU_EXCEPT_ALL)
    rd = false;
  j->right_distinct = rd;
  j->track_build_matches = jt == CRGPU_RIGHT_OUTER || jt == CRGPU_FULL_OUTER || jt == CRGPU_RIGHT_SEMI || jt == CRGPU_RIGHT_ANTI;
  j->null_eq = jt == CRGPU_INTERSECT_ALL || jt == CRGPU_EXCEPT_ALL;
  {  // emitted columns: JoinType.MakeOutputTypes (join_type.go:101-140), then the optional projection
    std::vector<crgpu_join::OutSrc> all;
    if (left_in_output(jt))
      for (int c = 0; c < spec->n_left_cols; c++) all.push_back({true, c});
    if (right_in_output(jt))
      for (int c = 0; c < spec->n_right_cols; c++) all.push_back({false, c});
    if (spec->n_out_cols < 0) {
      j->out_src = all;
    } else {
      for (int k = 0; k < spec->n_out_cols; k++) {
        int q = spec->out_cols[k];
        if (q < 0 || q >= static_cast<int>(all.size())) {
          delete j;
          return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_create: projected column ordinal out of range");
        }
        j->out_src.push_back(all[q]);
      }
    }
  }
  // The packed table keeps one row per key: usable whenever the first match
  // decides the result (distinct build side, or pure existence tests).
  j->first_match = rd || jt == CRGPU_LEFT_ANTI;
  j->jt_packed_ok = jt == CRGPU_INNER || jt == CRGPU_LEFT_OUTER || jt == CRGPU_RIGHT_OUTER || jt == CRGPU_FULL_OUTER ||
                    jt == CRGPU_LEFT_SEMI || jt == CRGPU_LEFT_ANTI;
  // One fixed-width key column always packs into a 64-bit word; several do when they are integers / BOOLs whose
  // build-side ranges add up to <= 64 bits, which is only known at build time (KeyPack, device.h).
  j->pack_candidate = spec->n_eq <= kMaxPackCols;
  for (int i = 0; i < spec->n_eq; i++) {
    const int lt = j->left_types[j->left_eq[i]], rt = j->right_types[j->right_eq[i]];
    if (lt == CRGPU_BYTES || rt == CRGPU_BYTES) j->pack_candidate = false;
    if (spec->n_eq > 1 && (lt == CRGPU_FLOAT64 || rt == CRGPU_FLOAT64)) j->pack_candidate = false;
  }
  j->packed = false;
  *out = j;
  return CRGPU_OK;
}

extern "C" void crgpu_dev_join_destroy(crgpu_join *j) {
  if (j == nullptr) return;
  if (j->st != nullptr) (void)hipStreamSynchronize(j->st);  // its buffers return to the pool: no kernel may still use them
  delete j;
}

extern "C" crgpu_status crgpu_dev_join_reset(crgpu_join *j) {
  if (j == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_reset: null handle");
  CRGPU_ENTER(j);
  j->cancelled = 0;
  j->built = false;
  j->n_build = 0;
  j->build_cols.clear();
  return CRGPU_OK;
}


// ---- radix path (radix.h): eligibility and row sources
static int64_t env_i64(const char *name, int64_t def) {
  const char *v = std::getenv(name);
  return (v != nullptr && *v != 0) ? std::atoll(v) : def;
}
// INNER / LEFT SEMI joins through the radix tables: a bitmap over the first key column's values when their range is
// small enough to be worth a pass over the probe side (<= 2^27 bits = 16 MiB; dense surrogate keys are far below).
// Planned when the table is built (the range is known then), built by the first probe chunk that would use it: a build
// side of 15 M orders costs 0.7 ms of atomics that a probe side of 3 M rows never earns back.
static void plan_key0_bitmap(crgpu_join *j, int64_t n) {
  j->key0_on = false;
  j->key0_pending = false;
  const char *env = std::getenv("CRGPU_JOIN_KEY0_BITMAP");  // "0": no first-key bitmap (tests, measurements)
  if ((env != nullptr && env[0] == '0') || !(j->join_type == CRGPU_INNER || j->join_type == CRGPU_LEFT_SEMI)) return;
  unsigned long long range;
  const DCol &c0 = j->kp.c[0];
  if (j->kp.n > 1) range = j->kp.range[0];
  else if (j->radix.direct.span != 0 && c0.type != CRGPU_FLOAT64) range = j->radix.direct.span - 1;
  else return;
  if (range >= (1ull << 27) || static_cast<unsigned long long>(n) * 2 > range) return;
  j->key0_pending = true;
}
static crgpu_status build_key0_bitmap(crgpu_join *j, int64_t n) {
  j->key0_pending = false;
  long long lo;
  unsigned long long range;
  const DCol &c0 = j->kp.c[0];
  if (j->kp.n > 1) {
    lo = j->kp.lo[0], range = j->kp.range[0];
  } else if (j->radix.direct.span != 0 && c0.type != CRGPU_FLOAT64) {
    lo = static_cast<long long>(j->radix.direct.lo), range = j->radix.direct.span - 1;
  } else {
    return CRGPU_OK;
  }
  if (range >= (1ull << 27)) return CRGPU_OK;
  // a build side that fills most of its own value range (a join on the build side's dense primary key, config J) leaves
  // the bitmap nothing to reject but out-of-range keys: not worth the 2^27 atomics of building it
  if (static_cast<unsigned long long>(n) * 2 > range) return CRGPU_OK;
  hipStream_t st = stream();
  const size_t bytes = static_cast<size_t>((range >> 5) + 1) * 4;
  CRGPU_TRY(j->key0_bits.ensure(bytes));
  CRGPU_HIP(hipMemsetAsync(j->key0_bits.p, 0, bytes, st));
  ProfScope prof("join_build");
  hipLaunchKernelGGL(key0_bitmap_build_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, c0, lo, range, n, j->key0_bits.as<uint32_t>());
  CRGPU_HIP(hipGetLastError());
  j->key0_lo = lo;
  j->key0_range = range;
  j->key0_on = true;
  return CRGPU_OK;
}
static bool radix_eligible(const crgpu_join *j, int64_t n) {
  const int jt = j->join_type;
  if (!(jt == CRGPU_INNER || jt == CRGPU_LEFT_OUTER || jt == CRGPU_LEFT_SEMI || jt == CRGPU_LEFT_ANTI)) return false;
  if (j->track_build_matches || j->out_src.size() > static_cast<size_t>(kRadixMaxFused)) return false;
  return n > 0 && n >= env_i64("CRGPU_RADIX_MIN_BUILD", int64_t(1) << 22);
}
static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static RadixSrc radix_src(const KeyPack &kp, const DCol *pay) {
  RadixSrc s;
  std::memset(&s, 0, sizeof(s));
  s.kp = kp;
  s.pay = pay ? static_cast<const uint8_t *>(pay->data) : nullptr;
  s.pay_width = pay ? type_width(pay->type) : 0;
  const DCol &key = kp.c[0];
  s.fast = kp.n == 1 && key.type == CRGPU_INT64 && key.valid == nullptr && aligned16(key.data) &&
           (pay == nullptr || (s.pay_width == 8 && aligned16(pay->data)));
  return s;
}

// covering: every emitted build column is the int/bool key itself or ONE 8-byte non-null payload column, which then
// rides in the table's value word (no row-id indirection at probe time).
static void decide_covering(crgpu_join *j) {
  const int kc = j->right_eq[0];
  j->covering = false;
  j->cover_col = -1;
  // a build key column can be emitted from the probe row's key word when the key is that one non-float column
  const bool key_castable = j->kp.n == 1 && j->right_types[kc] != CRGPU_FLOAT64 && j->build_cols[kc].valid == nullptr;
  if (j->track_build_matches || j->out_src.size() > static_cast<size_t>(kMaxFused)) return;
  bool ok = true;
  int pc = -1;
  for (const auto &o : j->out_src) {
    if (o.is_left) continue;
    if (o.src == kc && key_castable) continue;
    if (pc == -1 || pc == o.src) {
      pc = o.src;
      ok = ok && type_width(j->right_types[pc]) == 8 && j->build_cols[pc].valid == nullptr;
    } else {
      ok = false;
    }
  }
  if (ok) {
    j->covering = true;
    j->cover_col = pc;
  }
}

// The build side's key word: one column packs as it is; several integer / BOOL columns are range-compressed, which
// needs their [min, max] over the build rows (one reduction pass + a 64-byte read-back).  *ok = false: the ranges do
// not fit 64 bits (or the key holds a BYTES / several FLOAT columns) and the join takes the chained table.
static crgpu_status make_build_keypack(crgpu_join *j, int64_t n, bool *ok) {
  *ok = false;
  KeyPack &kp = j->kp;
  std::memset(&kp, 0, sizeof(kp));
  if (!j->pack_candidate) return CRGPU_OK;
  kp.n = static_cast<int>(j->right_eq.size());
  for (int c = 0; c < kp.n; c++) {
    kp.c[c] = j->build_cols[j->right_eq[c]];
    kp.range[c] = ~0ull;
  }
  if (kp.n == 1) {
    *ok = true;
    return CRGPU_OK;
  }
  long long mm[2 * kMaxPackCols];
  for (int c = 0; c < kMaxPackCols; c++) mm[2 * c] = 0x7FFFFFFFFFFFFFFFll, mm[2 * c + 1] = static_cast<long long>(0x8000000000000000ull);
  hipStream_t st = stream();
  if (n > 0) {
    CRGPU_TRY(j->minmax.ensure(sizeof(mm)));
    CRGPU_HIP(hipMemcpyAsync(j->minmax.p, mm, sizeof(mm), hipMemcpyHostToDevice, st));
    ProfScope prof("join_build");
    hipLaunchKernelGGL(key_minmax_kernel, dim3(grid_for(n, 256 * 8)), dim3(256), 0, st, kp, n, j->minmax.as<long long>());
    CRGPU_HIP(hipGetLastError());
    CRGPU_HIP(hipMemcpyAsync(mm, j->minmax.p, sizeof(mm), hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
  }
  uint32_t shift = 0;
  for (int c = 0; c < kp.n; c++) {
    long long lo = mm[2 * c], hi = mm[2 * c + 1];
    if (lo > hi) lo = hi = 0;  // no build row has a key: nothing can match
    const uint64_t range = static_cast<uint64_t>(hi) - static_cast<uint64_t>(lo);
    uint32_t bits = 0;
    while (bits < 64 && (range >> bits) != 0) bits++;
    if (shift + bits > 64) return CRGPU_OK;  // does not fit: chained table
    kp.lo[c] = lo;
    kp.range[c] = range;
    kp.shift[c] = shift;
    shift += bits;
  }
  if (shift >= 64 && kp.n > 1) {
    // a full 64-bit word could collide with the tables' EMPTY marker only through the side slot, which handles it
  }
  *ok = true;
  return CRGPU_OK;
}
// Direct-addressed radix tables (radix.h, RadixDirect) for a build side whose ONE integer key column is dense: its
// [min, max] is measured (the first 2^16 rows decide whether the full pass over the key column is worth running) and,
// if the range fits the tables' slots and is at least a quarter full, the join addresses its tables by key - min:
// nothing is hashed, the table kernel places instead of walking cuckoo chains, a lookup reads one slot.  What CockroachDB's
// surrogate keys (TPC-H's too: custkey, partkey, suppkey) look like.  CRGPU_JOIN_DIRECT=0 keeps the hashed tables.
static crgpu_status maybe_direct(crgpu_join *j, int64_t n, RadixSrc *src) {
  const char *env = std::getenv("CRGPU_JOIN_DIRECT");
  if (env != nullptr && env[0] == '0') return CRGPU_OK;
  if (j->kp.n != 1 || !is_int_type(j->kp.c[0].type) || n <= 0) return CRGPU_OK;
  hipStream_t st = stream();
  const uint64_t widest = uint64_t(1) << (10 + kRadixMaxSubBits + kRadixDirectBits);
  long long mm[2 * kMaxPackCols];
  CRGPU_TRY(j->minmax.ensure(sizeof(mm)));
  const int64_t sample = n < 65536 ? n : 65536;
  for (int pass = 0; pass < 2; pass++) {
    const int64_t rows = pass == 0 ? sample : n;
    if (pass == 1 && rows == sample) break;
    for (int c = 0; c < kMaxPackCols; c++) mm[2 * c] = 0x7FFFFFFFFFFFFFFFll, mm[2 * c + 1] = static_cast<long long>(0x8000000000000000ull);
    CRGPU_HIP(hipMemcpyAsync(j->minmax.p, mm, sizeof(mm), hipMemcpyHostToDevice, st));
    {
      ProfScope prof("join_build");
      const DCol &kc = j->kp.c[0];
      if (kc.type == CRGPU_INT64 && kc.valid == nullptr && aligned16(kc.data)) {
        int64_t blocks = (rows / 2 + 512 * 8 - 1) / (512 * 8);
        const int64_t cap = static_cast<int64_t>(num_cus()) * 4;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(i64_minmax_kernel, dim3(static_cast<int>(blocks < cap ? blocks : cap)), dim3(512), 0, st,
                           static_cast<const ll2v *>(kc.data), rows, j->minmax.as<long long>());
      } else {
        hipLaunchKernelGGL(key_minmax_kernel, dim3(grid_for(rows, 256 * 8)), dim3(256), 0, st, j->kp, rows, j->minmax.as<long long>());
      }
      CRGPU_HIP(hipGetLastError());
    }
    CRGPU_HIP(hipMemcpyAsync(mm, j->minmax.p, sizeof(mm), hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (mm[0] > mm[1]) return CRGPU_OK;  // no row has a key
    if (static_cast<uint64_t>(mm[1]) - static_cast<uint64_t>(mm[0]) >= widest) return CRGPU_OK;  // too wide already
  }
  const uint64_t span = static_cast<uint64_t>(mm[1]) - static_cast<uint64_t>(mm[0]) + 1;
  const int sb = radix_direct_sub_bits(span, n);
  if (sb < 0) return CRGPU_OK;
  src->direct.lo = static_cast<uint64_t>(mm[0]);
  src->direct.span = span;
  src->direct.sub_bits = static_cast<uint32_t>(sb);
  return CRGPU_OK;
}

static KeyPack probe_keypack(const crgpu_join *j, const std::vector<DCol> &left) {
  KeyPack kp = j->kp;
  for (int c = 0; c < kp.n; c++) kp.c[c] = left[j->left_eq[c]];
  return kp;
}

static crgpu_status crgpu_dev_join_build_impl(crgpu_join *j, const crgpu_dcol *right_cols, int64_t n) {
  if (j == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_build: null handle");
  CRGPU_ENTER(j);
  CRGPU_CHECK_CANCEL(j);
  if (n < 0 || n > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_join_build: build side limited to 2^31 rows per table");
  if (n > 0 && right_cols == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_build: null columns");
  const int nr = static_cast<int>(j->right_types.size());
  j->build_cols.clear();
  for (int c = 0; c < nr; c++) {
    DCol d{j->right_types[c], 0, nullptr, nullptr, nullptr};
    if (n > 0) {
      if (right_cols[c].type != j->right_types[c]) return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_build: column type differs from the spec");
      if (right_cols[c].type == CRGPU_BYTES && right_cols[c].offsets == nullptr)
        return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_build: BYTES column without offsets");
      if (right_cols[c].type != CRGPU_BYTES && right_cols[c].data == nullptr)
        return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_build: null data pointer");
      d = to_dcol(right_cols[c]);
    }
    j->build_cols.push_back(d);
  }
  j->n_build = n;
  hipStream_t st = stream();

  if (j->track_build_matches) {
    CRGPU_TRY(j->matched.ensure(static_cast<size_t>(n > 0 ? n : 1)));
    CRGPU_HIP(hipMemsetAsync(j->matched.p, 0, static_cast<size_t>(n > 0 ? n : 1), st));
  }
  if (j->null_eq) {  // set-operation joins
    CRGPU_TRY(j->claimed.ensure(static_cast<size_t>(n > 0 ? n : 1) * 4));
    CRGPU_HIP(hipMemsetAsync(j->claimed.p, 0, static_cast<size_t>(n > 0 ? n : 1) * 4, st));
  }

  bool packable = false;
  CRGPU_TRY(make_build_keypack(j, n, &packable));
  j->packed = packable && j->first_match && j->jt_packed_ok;
  j->chain_exact = false;
  if (j->packed) {
    j->cap = next_pow2(static_cast<uint64_t>(n) * 2 < 1024 ? 1024 : static_cast<uint64_t>(n) * 2);
    decide_covering(j);
    j->use_radix = false;
    if (radix_eligible(j, n)) {
      RadixSrc src = radix_src(j->kp, (j->covering && j->cover_col >= 0) ? &j->build_cols[j->cover_col] : nullptr);
      CRGPU_TRY(maybe_direct(j, n, &src));
      bool ok = false;
      CRGPU_TRY(radix_build(j->radix, src, n, &ok));
      if (!ok && src.direct.span != 0) {  // (a sub-partition overflowed: duplicate keys after all) the hashed tables may still do
        src.direct = RadixDirect{0, 0, 0, 0};
        CRGPU_TRY(radix_build(j->radix, src, n, &ok));
      }
      j->use_radix = ok;
      j->key0_on = j->key0_pending = false;
      if (ok && n > 0) plan_key0_bitmap(j, n);
    }
    if (!j->use_radix) {
      CRGPU_TRY(j->slots.ensure(j->cap * sizeof(Slot)));
      CRGPU_TRY(j->special.ensure(sizeof(Special)));
      CRGPU_HIP(hipMemsetAsync(j->slots.p, 0xFF, j->cap * sizeof(Slot), st));
      CRGPU_HIP(hipMemsetAsync(j->special.p, 0, sizeof(Special), st));
      j->bloom_on = false;
      const char *bloom_env = std::getenv("CRGPU_JOIN_BLOOM");  // "0": no key bitmap (tests, measurements)
      const bool bloom_enabled = bloom_env == nullptr || bloom_env[0] != '0';
      if (bloom_enabled && n > 0 && j->cap * sizeof(Slot) > (size_t(2) << 20)) {  // the table will not stay in an XCD's L2
        uint64_t bits = 1u << 16;
        while (bits < static_cast<uint64_t>(n) * 16 && bits < (uint64_t(1) << 24)) bits <<= 1;
        CRGPU_TRY(j->bloom.ensure(bits / 8));
        CRGPU_HIP(hipMemsetAsync(j->bloom.p, 0, bits / 8, st));
        j->bloom_mask = static_cast<uint32_t>(bits - 1);
        j->bloom_on = true;
      }
    }
    if (n > 0 && !j->use_radix) {
      const uint64_t *payload = (j->covering && j->cover_col >= 0)
                                    ? static_cast<const uint64_t *>(j->build_cols[j->cover_col].data)
                                    : nullptr;
      ProfScope prof("join_build");
      hipLaunchKernelGGL(packed_build_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, j->kp, n,
                         j->slots.as<Slot>(), j->cap - 1, payload, j->special.as<Special>(),
                         j->bloom_on ? j->bloom.as<uint32_t>() : nullptr, j->bloom_mask);
      CRGPU_HIP(hipGetLastError());
    }
  }
  j->use_radix_dup = false;
  if (!j->packed && packable && !j->null_eq && (j->join_type == CRGPU_INNER || j->join_type == CRGPU_LEFT_OUTER) &&
      radix_eligible(j, n)) {
    // duplicate build keys: group them per sub-partition of the radix table (falls back to the chained table when the
    // keys are so skewed that a sub-partition overflows)
    decide_covering(j);
    RadixSrc src = radix_src(j->kp, (j->covering && j->cover_col >= 0) ? &j->build_cols[j->cover_col] : nullptr);
    bool ok = false;
    CRGPU_TRY(radix_build_dup(j->radix, src, n, &ok));
    j->use_radix_dup = ok;
  }
  if (!j->packed && !j->use_radix_dup) {
    uint64_t nb = next_pow2(static_cast<uint64_t>(n) < 256 ? 256 : static_cast<uint64_t>(n));
    if (nb > 0x80000000ull) nb = 0x80000000ull;
    j->nb = static_cast<uint32_t>(nb);
    CRGPU_TRY(j->first.ensure(static_cast<size_t>(nb) * 4));
    CRGPU_TRY(j->next.ensure(static_cast<size_t>(n > 0 ? n : 1) * 4));
    CRGPU_HIP(hipMemsetAsync(j->first.p, 0, static_cast<size_t>(nb) * 4, st));
    if (n > 0) {
      KeyCols keys;
      keys.n = static_cast<int>(j->right_eq.size());
      keys._pad = 0;
      for (int k = 0; k < keys.n; k++) keys.c[k] = j->build_cols[j->right_eq[k]];
      // Keys that pack into one word: the bucket-ordered copy of the key words decides equality by itself.  Not with
      // NULL = NULL semantics (set operations): a NULL row is in the table then, and its stored word is not its key.
      j->chain_exact = packable && !j->null_eq;
      CRGPU_TRY(j->cstart.ensure((static_cast<size_t>(nb) + 1) * 4));
      CRGPU_TRY(j->offsets.ensure((static_cast<size_t>(nb) + 1) * 8));  // scan output before narrowing; the probe reuses it
      CRGPU_TRY(j->grp.ensure(static_cast<size_t>(n) * 4));
      if (j->chain_exact) CRGPU_TRY(j->gkey.ensure(static_cast<size_t>(n) * 16));
      ProfScope prof("join_build");
      hipLaunchKernelGGL(chain_hist_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, keys, n, j->null_eq, j->nb - 1,
                         j->first.as<uint32_t>(), j->next.as<uint32_t>());
      CRGPU_HIP(hipGetLastError());
      CRGPU_TRY(exclusive_scan_u32(j->first.as<uint32_t>(), static_cast<int64_t>(nb), j->offsets.as<uint64_t>(),
                                   j->scratch, nullptr));
      // half the footprint of the array every probe row reads at random
      hipLaunchKernelGGL(chain_narrow_kernel, dim3(grid_for(static_cast<int64_t>(nb) + 1, 256)), dim3(256), 0, st,
                         j->offsets.as<uint64_t>(), static_cast<int64_t>(nb) + 1, j->cstart.as<uint32_t>());
      hipLaunchKernelGGL(chain_scatter_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, j->kp, n,
                         j->next.as<uint32_t>(), j->cstart.as<uint32_t>(), j->first.as<uint32_t>(),
                         j->grp.as<uint32_t>(), j->chain_exact ? j->gkey.as<ulonglong2>() : nullptr);
      CRGPU_HIP(hipGetLastError());
      if (!j->chain_exact) {
        CRGPU_TRY(j->rep8.ensure(static_cast<size_t>(n)));
        hipLaunchKernelGGL(chain_mark_kernel, dim3(grid_for(static_cast<int64_t>(nb), 256)), dim3(256), 0, st, keys,
                           static_cast<int64_t>(nb), j->null_eq, j->cstart.as<uint32_t>(), j->grp.as<uint32_t>(),
                           j->rep8.as<uint8_t>());
        CRGPU_HIP(hipGetLastError());
      }
    } else {
      CRGPU_TRY(j->cstart.ensure((static_cast<size_t>(nb) + 1) * 4));
      CRGPU_HIP(hipMemsetAsync(j->cstart.p, 0, (static_cast<size_t>(nb) + 1) * 4, st));
      CRGPU_TRY(j->grp.ensure(4));
      j->chain_exact = false;
    }
  }
  CRGPU_HIP(hipStreamSynchronize(st));
  j->built = true;
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_join_build(crgpu_join *j, const crgpu_dcol *right_cols, int64_t n) {
  const crgpu_status s_ = crgpu_dev_join_build_impl(j, right_cols, n);
  if (s_ == CRGPU_OK && j != nullptr) note_rows(j, n, 0);
  return s_;
}

// Output column bookkeeping ------------------------------------------------
static void join_setup_outputs(crgpu_join *j) {
  size_t nout = j->out_src.size();
  if (j->out.size() != nout) {
    j->out = std::vector<OutCol>(nout);
    j->valid_bytes = std::vector<DevBuf>(nout);
    j->views.resize(nout);
  }
}

// All-NULL column of `n` rows (emitRight's probe side, hashjoiner.go:436-443).
static crgpu_status make_null_col(int type, int64_t n, OutCol &o) {
  o.type = type;
  o.nullable = true;
  size_t w = type == CRGPU_BYTES ? 1 : type_width(type);
  CRGPU_TRY(o.data.ensure(static_cast<size_t>(n > 0 ? n : 1) * w));
  CRGPU_HIP(hipMemsetAsync(o.data.p, 0, static_cast<size_t>(n > 0 ? n : 1) * w, stream()));
  CRGPU_TRY(o.valid.ensure(static_cast<size_t>((n + 7) / 8 + 1)));
  CRGPU_HIP(hipMemsetAsync(o.valid.p, 0, static_cast<size_t>((n + 7) / 8 + 1), stream()));
  if (type == CRGPU_BYTES) {
    CRGPU_TRY(o.offsets.ensure(static_cast<size_t>(n + 1) * 4));
    CRGPU_HIP(hipMemsetAsync(o.offsets.p, 0, static_cast<size_t>(n + 1) * 4, stream()));
  }
  return CRGPU_OK;
}

static crgpu_status finish_result(crgpu_join *j, int64_t n, crgpu_dev_result *res) {
  for (size_t c = 0; c < j->out.size(); c++) j->views[c] = j->out[c].view();
  res->n = n;
  res->ncols = static_cast<int32_t>(j->out.size());
  res->_pad = 0;
  res->cols = j->views.data();
  return CRGPU_OK;
}

template <int TB, int RPT>
static crgpu_status launch_packed_probe(int jt, const ProbeArgs &a, int grid) {
  hipStream_t st = stream();
  switch (jt) {
    case CRGPU_INNER: hipLaunchKernelGGL((packed_probe_kernel<CRGPU_INNER, TB, RPT>), dim3(grid), dim3(TB), 0, st, a); break;
    case CRGPU_LEFT_OUTER: hipLaunchKernelGGL((packed_probe_kernel<CRGPU_LEFT_OUTER, TB, RPT>), dim3(grid), dim3(TB), 0, st, a); break;
    case CRGPU_RIGHT_OUTER: hipLaunchKernelGGL((packed_probe_kernel<CRGPU_RIGHT_OUTER, TB, RPT>), dim3(grid), dim3(TB), 0, st, a); break;
    case CRGPU_FULL_OUTER: hipLaunchKernelGGL((packed_probe_kernel<CRGPU_FULL_OUTER, TB, RPT>), dim3(grid), dim3(TB), 0, st, a); break;
    case CRGPU_LEFT_SEMI: hipLaunchKernelGGL((packed_probe_kernel<CRGPU_LEFT_SEMI, TB, RPT>), dim3(grid), dim3(TB), 0, st, a); break;
    case CRGPU_LEFT_ANTI: hipLaunchKernelGGL((packed_probe_kernel<CRGPU_LEFT_ANTI, TB, RPT>), dim3(grid), dim3(TB), 0, st, a); break;
    default: return fail(CRGPU_ERR_INVALID, "packed probe: join type not handled by the packed table");
  }
  CRGPU_HIP(hipGetLastError());
  return CRGPU_OK;
}

static crgpu_status probe_packed(crgpu_join *j, const std::vector<DCol> &left, int64_t n, const uint8_t *keep,
                                 crgpu_dev_result *res) {
  const int jt = j->join_type;
  const bool outer_nulls = left_or_full_outer(jt);
  hipStream_t st = stream();
  int64_t cap_rows = n > 0 ? n : 1;  // first-match joins emit at most one row per probe row

  ProbeArgs a;
  std::memset(&a, 0, sizeof(a));
  a.kp = probe_keypack(j, left);
  a.n = n;
  a.slots = j->slots.as<Slot>();
  a.mask = j->cap - 1;
  a.special = j->special.as<Special>();
  a.matched = j->track_build_matches ? j->matched.as<uint8_t>() : nullptr;
  a.keep = keep;
  CRGPU_TRY(j->counter.ensure(8));
  CRGPU_HIP(hipMemsetAsync(j->counter.p, 0, 8, st));
  a.out_count = j->counter.as<unsigned long long>();
  // Large probe chunks of joins that drop unmatched rows: the bitmap is evaluated by a pass of its own and the probe kernel
  // sees the survivors only (bloom_filter_kernel); outputs are then sized for the survivors, not for the chunk.
  const char *pf_env = std::getenv("CRGPU_JOIN_PREFILTER_MIN");  // rows per probe chunk from which the pass pays (tests: 0)
  const int64_t prefilter_min = pf_env != nullptr ? std::atoll(pf_env) : int64_t(1) << 21;
  if (j->bloom_on && (jt == CRGPU_INNER || jt == CRGPU_LEFT_SEMI) && n >= prefilter_min && n <= 0x7FFFFFF0ll) {
    CRGPU_TRY(j->cand.ensure(static_cast<size_t>(n) * 4));
    {
      ProfScope prof("join_probe");
      hipLaunchKernelGGL(bloom_filter_kernel, dim3(grid_for(n, 1024 * 8)), dim3(1024), 0, st, a.kp, n, keep, j->bloom.as<uint32_t>(),
                         j->bloom_mask, j->cand.as<int32_t>(), j->counter.as<unsigned long long>());
      CRGPU_HIP(hipGetLastError());
    }
    uint64_t ncand = 0;
    CRGPU_HIP(hipMemcpyAsync(&ncand, j->counter.p, 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipMemsetAsync(j->counter.p, 0, 8, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (ncand * 2 <= static_cast<uint64_t>(n)) {
      a.list = j->cand.as<int32_t>();
      a.n = static_cast<int64_t>(ncand);
      cap_rows = ncand > 0 ? static_cast<int64_t>(ncand) : 1;
    } else {
      j->bloom_on = false;  // most rows pass: the bitmap only adds a read per row, here and in the chunks that follow
    }
  }
  if (a.list == nullptr && j->bloom_on) {
    a.bloom = j->bloom.as<uint32_t>();
    a.bmask = j->bloom_mask;
  }

  // Decide per output column: fused into the probe kernel, or materialised afterwards through pairs.
  struct Later { int out_idx; bool is_left; int src; };
  std::vector<Later> later;
  std::vector<int> fused_out;
  int oc = 0;
  auto add_fused = [&](const uint8_t *src, const uint8_t *src_valid, int type, int side, bool nullable) -> crgpu_status {
    OutCol &o = j->out[oc];
    o.type = type;
    o.nullable = nullable;
    CRGPU_TRY(o.data.ensure(static_cast<size_t>(cap_rows) * type_width(type)));
    CopyCol &cc = a.copy[a.ncopy++];
    fused_out.push_back(oc);
    cc.src = src;
    cc.dst = o.data.as<uint8_t>();
    cc.src_valid = src_valid;
    cc.width = type_width(type);
    cc.side = side;
    cc.dst_valid = nullptr;
    if (nullable) {
      CRGPU_TRY(j->valid_bytes[oc].ensure(static_cast<size_t>(cap_rows)));
      CRGPU_TRY(o.valid.ensure(static_cast<size_t>((cap_rows + 7) / 8 + 1)));
      cc.dst_valid = j->valid_bytes[oc].as<uint8_t>();
    }
    return CRGPU_OK;
  };
  for (const auto &os : j->out_src) {
    const int c = os.src;
    if (os.is_left) {
      if (left[c].type != CRGPU_BYTES && a.ncopy < kMaxFused)
        CRGPU_TRY(add_fused(static_cast<const uint8_t *>(left[c].data), left[c].valid, left[c].type, SIDE_PROBE,
                            left[c].valid != nullptr));
      else
        later.push_back(Later{oc, true, c});
    } else {
      const DCol &b = j->build_cols[c];
      if (j->covering && a.ncopy < kMaxFused) {  // covering => c is the key column or the cover column
        CRGPU_TRY(add_fused(nullptr, nullptr, b.type, c == j->cover_col ? SIDE_COVER : SIDE_KEYCAST, outer_nulls));
      } else if (!j->covering && b.type != CRGPU_BYTES && a.ncopy < kMaxFused) {
        CRGPU_TRY(add_fused(static_cast<const uint8_t *>(b.data), b.valid, b.type, SIDE_BUILD,
                            outer_nulls || b.valid != nullptr));
      } else {
        later.push_back(Later{oc, false, c});
      }
    }
    oc++;
  }
  if (!later.empty()) {
    CRGPU_TRY(j->pair_p.ensure(static_cast<size_t>(cap_rows) * 4));
    CRGPU_TRY(j->pair_b.ensure(static_cast<size_t>(cap_rows) * 4));
    a.pair_p = j->pair_p.as<int32_t>();
    a.pair_b = j->pair_b.as<int32_t>();
  }

  uint64_t nout = 0;
  if (a.n > 0) {
    // 512 threads x 4 rows: 78 VGPRs, three workgroups per CU.  Measured against 1024 x 8 (128 VGPRs, one workgroup per
    // CU: every barrier and the tile's output-position atomic stall the whole CU): 5 % hit probe of 60M rows 0.95 -> 0.86 ms,
    // TPC-H Q9's probes 2.27 -> 2.13 ms; 512 x 8 and 256 x 8 (3 waves per SIMD) were 1.3-1.5 x slower.
    constexpr int TB = 512, RPT = 4;
    int64_t ntiles = (a.n + TB * RPT - 1) / (TB * RPT);
    int64_t cap_grid = static_cast<int64_t>(num_cus()) * 8;
    int grid = static_cast<int>(ntiles < cap_grid ? ntiles : cap_grid);
    {
      ProfScope prof("join_probe");
      CRGPU_TRY((launch_packed_probe<TB, RPT>(jt, a, grid)));
    }
    CRGPU_HIP(hipMemcpyAsync(&nout, j->counter.p, 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    // where most probe rows match the bitmap only adds a read per row: the following chunks go straight to the table
    if (j->bloom_on && (jt == CRGPU_INNER || jt == CRGPU_LEFT_SEMI) && n >= 4096 && nout * 10 > static_cast<uint64_t>(n) * 7) j->bloom_on = false;
  }
  for (int c = 0; c < a.ncopy; c++)  // pack the validity bytes of fused nullable columns into bitmaps
    if (a.copy[c].dst_valid != nullptr)
      CRGPU_TRY(pack_bitmap(a.copy[c].dst_valid, static_cast<int64_t>(nout), j->out[fused_out[c]].valid.as<uint8_t>()));
  for (const Later &l : later) {
    if (l.is_left)
      CRGPU_TRY(materialize(left[l.src], j->pair_p.as<int32_t>(), static_cast<int64_t>(nout), false, j->out[l.out_idx], j->scratch));
    else
      CRGPU_TRY(materialize(j->build_cols[l.src], j->pair_b.as<int32_t>(), static_cast<int64_t>(nout), outer_nulls,
                            j->out[l.out_idx], j->scratch));
  }
  CRGPU_HIP(hipStreamSynchronize(st));
  return finish_result(j, static_cast<int64_t>(nout), res);
}


// Probe through the radix-partitioned table.  Probe-side columns ride along as the row's
// 64-bit payload when one fixed-width non-null column (plus, optionally, the key itself)
// is all that is emitted; otherwise the payload is the probe row id and the columns are
// gathered afterwards, like the build columns of a non-covering table.
static crgpu_status probe_radix(crgpu_join *j, const std::vector<DCol> &left, int64_t n, crgpu_dev_result *res) {
  const int jt = j->join_type;
  const bool outer_nulls = left_or_full_outer(jt);
  const bool keeps_unmatched = jt == CRGPU_LEFT_OUTER || jt == CRGPU_LEFT_ANTI;
  hipStream_t st = stream();
  int64_t cap_rows = n > 0 ? n : 1;
  const int lkc = j->left_eq[0], rkc = j->right_eq[0];
  const DCol &lkey = left[lkc];
  // the probe key column can be emitted from the key word -- unless the tables are direct-addressed and unmatched rows
  // are kept: a probe key outside the build side's range has no key word then, yet its value must come out
  const bool key_castable = j->kp.n == 1 && lkey.type != CRGPU_FLOAT64 && !(keeps_unmatched && j->radix.direct.span != 0 && !j->use_radix_dup);

  int pay_col = -1;
  bool need_pairs_p = false;
  for (const auto &os : j->out_src) {
    if (!os.is_left) continue;
    if (os.src == lkc && key_castable) continue;
    const DCol &c = left[os.src];
    if (c.type == CRGPU_BYTES || c.valid != nullptr || (pay_col != -1 && pay_col != os.src)) need_pairs_p = true;
    else pay_col = os.src;
  }
  if (need_pairs_p) pay_col = -1;
  KeyPack pkp = probe_keypack(j, left);
  DCol pf_pay_col;
  const DCol *pay_ptr = pay_col >= 0 ? &left[pay_col] : nullptr;
  // A probe side much larger than the build side whose first key column mostly misses the build side's values: reject
  // those rows in a pass over that one column and hand the radix probe the survivors -- compacted key columns, the
  // payload column (or the survivors' ORIGINAL row numbers as the payload: the pairs then name rows of `left`).
  bool nullable_key = false;
  for (int c = 0; c < pkp.n; c++) nullable_key = nullable_key || pkp.c[c].valid != nullptr;
  const bool prefilter_size = !j->use_radix_dup && !nullable_key && n >= env_i64("CRGPU_JOIN_PREFILTER_MIN", int64_t(1) << 21) &&
                              n >= 4 * j->n_build && n <= 0x7FFFFFF0ll;
  if (j->key0_pending && prefilter_size) CRGPU_TRY(build_key0_bitmap(j, j->n_build));
  if (j->key0_on && prefilter_size) {
    // 2^14 evenly spaced rows first (a probe side whose keys all exist in the build side -- a foreign key into a sparse
    // table -- would pay a pass over the column and two written lists for nothing)
    CRGPU_TRY(j->counter.ensure(8));
    CRGPU_HIP(hipMemsetAsync(j->counter.p, 0, 8, st));
    const int64_t samples = n < (int64_t(1) << 14) ? n : int64_t(1) << 14;
    hipLaunchKernelGGL(key0_sample_kernel, dim3(grid_for(samples, 256)), dim3(256), 0, st, pkp.c[0], j->key0_lo, j->key0_range, n, samples,
                       j->key0_bits.as<uint32_t>(), j->counter.as<unsigned long long>());
    CRGPU_HIP(hipGetLastError());
    unsigned long long sample_hits = 0;
    CRGPU_HIP(hipMemcpyAsync(&sample_hits, j->counter.p, 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (sample_hits * 2 > static_cast<unsigned long long>(samples)) j->key0_on = false;
  }
  if (j->key0_on && prefilter_size) {
    CRGPU_TRY(j->cand.ensure(static_cast<size_t>(n) * 4));
    CRGPU_HIP(hipMemsetAsync(j->counter.p, 0, 8, st));
    {
      ProfScope prof("join_probe");
      // (the survivors' first-column values come out of the pass itself, sized for the worst case it is still worth it: n / 2)
      CRGPU_TRY(j->pf_key0.ensure(static_cast<size_t>(n) * 8));
      const bool fast = pkp.c[0].type == CRGPU_INT64 && aligned16(pkp.c[0].data);
      if (fast)
        hipLaunchKernelGGL(key0_filter_kernel<true>, dim3(grid_for(n, 1024 * 8)), dim3(1024), 0, st, pkp.c[0], j->key0_lo, j->key0_range, n,
                           j->key0_bits.as<uint32_t>(), j->cand.as<int32_t>(), j->pf_key0.as<long long>(), j->counter.as<unsigned long long>());
      else
        hipLaunchKernelGGL(key0_filter_kernel<false>, dim3(grid_for(n, 1024 * 8)), dim3(1024), 0, st, pkp.c[0], j->key0_lo, j->key0_range, n,
                           j->key0_bits.as<uint32_t>(), j->cand.as<int32_t>(), j->pf_key0.as<long long>(), j->counter.as<unsigned long long>());
      CRGPU_HIP(hipGetLastError());
    }
    unsigned long long m = 0;
    CRGPU_HIP(hipMemcpyAsync(&m, j->counter.p, 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (m * 2 > static_cast<unsigned long long>(n)) {
      j->key0_on = false;  // most rows pass: the bitmap only adds a pass, here and in the chunks that follow
    } else {
      const int64_t ms = static_cast<int64_t>(m);
      const int32_t *list = j->cand.as<int32_t>();
      pkp.c[0] = DCol{CRGPU_INT64, 0, j->pf_key0.p, nullptr, nullptr};  // (widened to INT64 by the pass: the value, whatever the column's type)
      for (int c = 1; c < pkp.n; c++) {
        CRGPU_TRY(materialize(pkp.c[c], list, ms, false, j->pf_key[c], j->scratch));
        pkp.c[c] = to_dcol(j->pf_key[c].view());
        pkp.c[c].valid = nullptr;
      }
      if (pay_col >= 0) {
        CRGPU_TRY(materialize(left[pay_col], list, ms, false, j->pf_pay, j->scratch));
        pf_pay_col = to_dcol(j->pf_pay.view());
        pf_pay_col.valid = nullptr;
      } else {
        CRGPU_TRY(j->pf_rowid.ensure(static_cast<size_t>(ms > 0 ? ms : 1) * 8));
        if (ms > 0) hipLaunchKernelGGL(widen_i32_i64_kernel, dim3(grid_for(ms, 256)), dim3(256), 0, st, list, ms, j->pf_rowid.as<long long>());
        CRGPU_HIP(hipGetLastError());
        pf_pay_col = DCol{CRGPU_INT64, 0, j->pf_rowid.p, nullptr, nullptr};
      }
      pay_ptr = &pf_pay_col;
      n = ms;
      cap_rows = n > 0 ? n : 1;
    }
  }
  RadixSrc src = radix_src(pkp, pay_ptr);
  int64_t min_probe = j->n_build >> 3;
  if (min_probe < (int64_t(1) << 20)) min_probe = int64_t(1) << 20;
  const bool partitioned = n >= env_i64("CRGPU_RADIX_MIN_PROBE", min_probe);
  if (j->use_radix_dup && n > 0) {  // duplicate build keys: the count pass sizes the outputs
    uint64_t total = 0;
    CRGPU_TRY(radix_probe_dup_count(j->radix, jt, src, n, partitioned, &total));
    if (total > 0x7FFFFFF0ull)
      return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_join_probe: more than 2^31 output rows for one probe chunk; probe in smaller chunks");
    cap_rows = total > 0 ? static_cast<int64_t>(total) : 1;
  }

  RadixEmit e;
  std::memset(&e, 0, sizeof(e));
  CRGPU_TRY(j->counter.ensure(8));
  CRGPU_HIP(hipMemsetAsync(j->counter.p, 0, 8, st));
  e.out_count = j->counter.as<unsigned long long>();
  struct Later { int out_idx; bool is_left; int src; };
  std::vector<Later> later;
  std::vector<int> fused_out;
  int oc = 0;
  auto add_fused = [&](int type, int side, bool nullable) -> crgpu_status {
    OutCol &o = j->out[oc];
    o.type = type;
    o.nullable = nullable;
    CRGPU_TRY(o.data.ensure(static_cast<size_t>(cap_rows) * type_width(type)));
    RadixCopy &cc = e.copy[e.ncopy++];
    fused_out.push_back(oc);
    cc.dst = o.data.as<uint8_t>();
    cc.width = type_width(type);
    cc.side = side;
    cc.dst_valid = nullptr;
    if (nullable) {
      CRGPU_TRY(j->valid_bytes[oc].ensure(static_cast<size_t>(cap_rows)));
      CRGPU_TRY(o.valid.ensure(static_cast<size_t>((cap_rows + 7) / 8 + 1)));
      cc.dst_valid = j->valid_bytes[oc].as<uint8_t>();
    }
    return CRGPU_OK;
  };
  bool need_pairs_b = false;
  for (const auto &os : j->out_src) {
    const int c = os.src;
    if (os.is_left) {
      if (c == lkc && key_castable) CRGPU_TRY(add_fused(lkey.type, R_KEY, keeps_unmatched && lkey.valid != nullptr));
      else if (c == pay_col) CRGPU_TRY(add_fused(left[c].type, R_PROBE_PAY, false));
      else later.push_back(Later{oc, true, c});
    } else if (j->covering) {  // c is the key column or the cover column
      CRGPU_TRY(add_fused(j->build_cols[c].type, c == j->cover_col ? R_BUILD_VAL : R_BUILD_KEY, outer_nulls));
    } else {
      later.push_back(Later{oc, false, c});
      need_pairs_b = true;
    }
    oc++;
  }
  (void)rkc;
  if (need_pairs_p) {
    CRGPU_TRY(j->pair_p.ensure(static_cast<size_t>(cap_rows) * 4));
    e.pair_p = j->pair_p.as<int32_t>();
  }
  if (need_pairs_b) {
    CRGPU_TRY(j->pair_b.ensure(static_cast<size_t>(cap_rows) * 4));
    e.pair_b = j->pair_b.as<int32_t>();
  }
  uint64_t nout = 0;
  if (n > 0) {
    if (j->use_radix_dup) CRGPU_TRY(radix_probe_dup_emit(j->radix, jt, src, n, e, partitioned));
    else CRGPU_TRY(radix_probe(j->radix, jt, src, n, e, partitioned));
    CRGPU_HIP(hipMemcpyAsync(&nout, j->counter.p, 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
  }
  for (int c = 0; c < e.ncopy; c++)
    if (e.copy[c].dst_valid != nullptr)
      CRGPU_TRY(pack_bitmap(e.copy[c].dst_valid, static_cast<int64_t>(nout), j->out[fused_out[c]].valid.as<uint8_t>()));
  for (const Later &l : later) {
    if (l.is_left)
      CRGPU_TRY(materialize(left[l.src], j->pair_p.as<int32_t>(), static_cast<int64_t>(nout), false, j->out[l.out_idx], j->scratch));
    else
      CRGPU_TRY(materialize(j->build_cols[l.src], j->pair_b.as<int32_t>(), static_cast<int64_t>(nout), outer_nulls,
                            j->out[l.out_idx], j->scratch));
  }
  CRGPU_HIP(hipStreamSynchronize(st));
  return finish_result(j, static_cast<int64_t>(nout), res);
}

template <int JT>
static void launch_chain(const ChainArgs &a, bool fill) {
  int grid = grid_for(a.n, 256);
  constexpr bool kPairs = JT == CRGPU_INNER || JT == CRGPU_LEFT_OUTER || JT == CRGPU_RIGHT_OUTER || JT == CRGPU_FULL_OUTER;
  if (fill && kPairs && a.mcache != nullptr)
    hipLaunchKernelGGL(chain_fill_cached_kernel, dim3(grid), dim3(256), 0, stream(), a);
  else if (fill) hipLaunchKernelGGL(chain_fill_kernel<JT>, dim3(grid), dim3(256), 0, stream(), a);
  else hipLaunchKernelGGL(chain_count_kernel<JT>, dim3(grid), dim3(256), 0, stream(), a);
}
static crgpu_status launch_chain_jt(int jt, const ChainArgs &a, bool fill) {
  switch (jt) {
    case CRGPU_INNER: launch_chain<CRGPU_INNER>(a, fill); break;
    case CRGPU_LEFT_OUTER: launch_chain<CRGPU_LEFT_OUTER>(a, fill); break;
    case CRGPU_RIGHT_OUTER: launch_chain<CRGPU_RIGHT_OUTER>(a, fill); break;
    case CRGPU_FULL_OUTER: launch_chain<CRGPU_FULL_OUTER>(a, fill); break;
    case CRGPU_LEFT_SEMI: launch_chain<CRGPU_LEFT_SEMI>(a, fill); break;
    case CRGPU_LEFT_ANTI: launch_chain<CRGPU_LEFT_ANTI>(a, fill); break;
    case CRGPU_INTERSECT_ALL: launch_chain<CRGPU_INTERSECT_ALL>(a, fill); break;
    case CRGPU_EXCEPT_ALL: launch_chain<CRGPU_EXCEPT_ALL>(a, fill); break;
    case CRGPU_RIGHT_SEMI: launch_chain<CRGPU_RIGHT_SEMI>(a, fill); break;
    case CRGPU_RIGHT_ANTI: launch_chain<CRGPU_RIGHT_ANTI>(a, fill); break;
    default: return fail(CRGPU_ERR_INVALID, "chained probe: unknown join type");
  }
  CRGPU_HIP(hipGetLastError());
  return CRGPU_OK;
}

static crgpu_status probe_chained(crgpu_join *j, const std::vector<DCol> &left, int64_t n, crgpu_dev_result *res) {
  const int jt = j->join_type;
  hipStream_t st = stream();
  uint64_t total = 0;
  if (n > 0) {
    ChainArgs a;
    std::memset(&a, 0, sizeof(a));
    a.probe.n = a.build.n = static_cast<int>(j->left_eq.size());
    for (int k = 0; k < a.probe.n; k++) {
      a.probe.c[k] = left[j->left_eq[k]];
      a.build.c[k] = j->build_cols[j->right_eq[k]];
    }
    a.n = n;
    a.start = j->cstart.as<uint32_t>();
    a.grp = j->grp.as<uint32_t>();
    a.gkey = nullptr;
    a.rep8 = j->chain_exact ? nullptr : j->rep8.as<uint8_t>();
    if (j->chain_exact) {
      a.gkey = j->gkey.as<ulonglong2>();
      a.kp = probe_keypack(j, left);
    }
    a.nb_mask = j->nb - 1;
    a.null_eq = j->null_eq;
    a.matched = j->track_build_matches ? j->matched.as<uint8_t>() : nullptr;
    a.claimed = j->null_eq ? j->claimed.as<unsigned int>() : nullptr;
    CRGPU_TRY(j->counts.ensure(static_cast<size_t>(n) * 4));
    CRGPU_TRY(j->offsets.ensure(static_cast<size_t>(n + 1) * 8));
    a.counts = j->counts.as<uint32_t>();
    a.offsets = j->offsets.as<uint64_t>();
    const bool pairs = jt == CRGPU_INNER || jt == CRGPU_LEFT_OUTER || jt == CRGPU_RIGHT_OUTER || jt == CRGPU_FULL_OUTER;
    const char *mc_env = getenv("CRGPU_JOIN_MATCH_CACHE");  // "0": both passes walk the buckets (what chunks beyond 2^28 rows do)
    if (pairs && n <= (int64_t(1) << 28) && !(mc_env != nullptr && mc_env[0] == '0')) {  // 32 bytes per probe row, at most 8 GiB
      CRGPU_TRY(j->mcache.ensure(static_cast<size_t>(n) * 4 * kChainCache));
      a.mcache = j->mcache.as<uint4>();
    }
    {
      ProfScope prof("join_probe");
      CRGPU_TRY(launch_chain_jt(jt, a, false));
    }
    if (left_in_output(jt)) {
      CRGPU_TRY(exclusive_scan_u32(a.counts, n, j->offsets.as<uint64_t>(), j->scratch, &total));
      if (total > 0x7FFFFFF0ull)
        return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_join_probe: more than 2^31 output rows for one probe chunk; probe in smaller chunks");
      CRGPU_TRY(j->pair_p.ensure(static_cast<size_t>(total ? total : 1) * 4));
      CRGPU_TRY(j->pair_b.ensure(static_cast<size_t>(total ? total : 1) * 4));
      a.pair_p = j->pair_p.as<int32_t>();
      a.pair_b = j->pair_b.as<int32_t>();
      if (total > 0) {
        ProfScope prof("join_probe");
        CRGPU_TRY(launch_chain_jt(jt, a, true));
      }
    }
  }
  int oc = 0;
  if (!left_in_output(jt)) total = 0;  // RIGHT SEMI / RIGHT ANTI: nothing while probing
  for (const auto &os : j->out_src) {
    if (!left_in_output(jt))
      CRGPU_TRY(make_null_col(j->right_types[os.src], 0, j->out[oc]));  // well-formed empty column
    else if (os.is_left)
      CRGPU_TRY(materialize(left[os.src], j->pair_p.as<int32_t>(), static_cast<int64_t>(total), false, j->out[oc],
                            j->scratch));
    else
      CRGPU_TRY(materialize(j->build_cols[os.src], j->pair_b.as<int32_t>(), static_cast<int64_t>(total),
                            left_or_full_outer(jt), j->out[oc], j->scratch));
    oc++;
  }
  CRGPU_HIP(hipStreamSynchronize(st));
  return finish_result(j, static_cast<int64_t>(total), res);
}

static crgpu_status join_probe_impl(crgpu_join *j, const crgpu_dcol *left_cols, int64_t n, const uint8_t *keep,
                                    crgpu_dev_result *res) {
  if (j == nullptr || res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_probe: null argument");
  CRGPU_ENTER(j);
  CRGPU_CHECK_CANCEL(j);
  if (!j->built) return fail(CRGPU_ERR_STATE, "crgpu_dev_join_probe: build side not built");
  if (n < 0 || n > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_join_probe: at most 2^31 rows per probe chunk");
  if (n > 0 && left_cols == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_probe: null columns");
  const int nl = static_cast<int>(j->left_types.size());
  std::vector<DCol> left;
  for (int c = 0; c < nl; c++) {
    DCol d{j->left_types[c], 0, nullptr, nullptr, nullptr};
    if (n > 0) {
      if (left_cols[c].type != j->left_types[c]) return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_probe: column type differs from the spec");
      if (left_cols[c].type == CRGPU_BYTES && left_cols[c].offsets == nullptr)
        return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_probe: BYTES column without offsets");
      if (left_cols[c].type != CRGPU_BYTES && left_cols[c].data == nullptr)
        return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_probe: null data pointer");
      d = to_dcol(left_cols[c]);
    }
    left.push_back(d);
  }
  join_setup_outputs(j);
  if (n == 0) keep = nullptr;
  // Empty build side short-circuits (hashjoiner.go:300-306, join_type.go:79-88): the
  // kernels below already produce the right answer on an empty table, so no special case.
  if (j->packed && !j->use_radix) return probe_packed(j, left, n, keep, res);
  if (keep != nullptr) {
    // the partitioned and the chained tables take dense probe rows: the selected rows are copied out first (what
    // crgpu_dev_deselect would do), then probed
    int64_t m = 0;
    CRGPU_TRY(j->sel_idx.ensure(static_cast<size_t>(n) * 4));
    CRGPU_TRY(select_flagged(keep, n, j->sel_idx.as<int32_t>(), j->scratch, &m));
    if (j->sel_cols.size() != static_cast<size_t>(nl)) j->sel_cols = std::vector<OutCol>(nl);
    for (int c = 0; c < nl; c++) {
      if (m == 0) {
        left[c] = DCol{j->left_types[c], 0, nullptr, nullptr, nullptr};
        continue;
      }
      CRGPU_TRY(materialize(left[c], j->sel_idx.as<int32_t>(), m, false, j->sel_cols[c], j->scratch));
      left[c] = to_dcol(j->sel_cols[c].view());
    }
    n = m;
  }
  if ((j->packed && j->use_radix) || j->use_radix_dup) return probe_radix(j, left, n, res);
  return probe_chained(j, left, n, res);
}

extern "C" crgpu_status crgpu_dev_join_probe(crgpu_join *j, const crgpu_dcol *left_cols, int64_t n,
                                             crgpu_dev_result *res) {
  const crgpu_status s_ = join_probe_impl(j, left_cols, n, nullptr, res);
  if (s_ == CRGPU_OK && j != nullptr) note_rows(j, n, res->n);
  return s_;
}

extern "C" crgpu_status crgpu_dev_join_probe_selected(crgpu_join *j, const crgpu_dcol *left_cols, int64_t n,
                                                      const uint8_t *keep_bool, crgpu_dev_result *res) {
  const crgpu_status s_ = join_probe_impl(j, left_cols, n, keep_bool, res);
  if (s_ == CRGPU_OK && j != nullptr) note_rows(j, n, res->n);
  return s_;
}

extern "C" crgpu_status crgpu_dev_join_emit_right(crgpu_join *j, crgpu_dev_result *res) {
  if (j == nullptr || res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_join_emit_right: null argument");
  CRGPU_ENTER(j);
  CRGPU_CHECK_CANCEL(j);
  if (!j->built) return fail(CRGPU_ERR_STATE, "crgpu_dev_join_emit_right: build side not built");
  join_setup_outputs(j);
  const int jt = j->join_type;
  int64_t count = 0;
  if (j->track_build_matches && j->n_build > 0) {
    const int64_t n = j->n_build;
    CRGPU_TRY(j->flags.ensure(static_cast<size_t>(n)));
    CRGPU_TRY(j->pair_b.ensure(static_cast<size_t>(n) * 4));
    const uint8_t want = jt == CRGPU_RIGHT_SEMI ? 1 : 0;  // emitRight(matched) hashjoiner.go:327,405
    hipLaunchKernelGGL(flag_eq_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream(), j->matched.as<uint8_t>(), n, want,
                       j->flags.as<uint8_t>());
    CRGPU_HIP(hipGetLastError());
    CRGPU_TRY(select_flagged(j->flags.as<uint8_t>(), n, j->pair_b.as<int32_t>(), j->scratch, &count));
  }
  int oc = 0;
  for (const auto &os : j->out_src) {
    if (os.is_left)
      CRGPU_TRY(make_null_col(j->left_types[os.src], count, j->out[oc]));
    else
      CRGPU_TRY(materialize(j->build_cols[os.src], j->pair_b.as<int32_t>(), count, false, j->out[oc], j->scratch));
    oc++;
  }
  CRGPU_HIP(hipStreamSynchronize(stream()));
  return finish_result(j, count, res);
}
