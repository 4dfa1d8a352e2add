// distinct.hip -- device-resident unordered DISTINCT.
//
// Replaces colexec.UnorderedDistinct.Next (pkg/sql/colexec/unordered_distinct.go:94-121) and
// colexechash.HashTable.DistinctBuild (pkg/sql/colexec/colexechash/hashtable.go:628-649): of every input batch
// only the tuples whose distinct columns were not seen before -- in earlier batches or earlier in the same batch --
// are emitted, all columns of the FIRST such row, in input order.
//
// Design (not the reference's chained table over 1024-row batches):
//  * Seen keys are numbered densely in emission order ("gid"); the distinct columns of emitted rows are appended to
//    a key store (the reference's ht.Vals in distinct build mode keeps the key columns only, hashtable.go:262-266).
//    Since a batch's new keys are exactly its output rows, the store is appended by copying the key columns of the
//    output (device-to-device, coalesced).
//  * The hash index is an open-addressing array of 64-bit entries {tag:23 | null mask:8 | gid:32}, at most half full
//    (capacity is ensured before a batch, so linear probing always ends).  While a batch is resolved an entry may hold
//    {tag | mask | store rows before the batch + row position}: rows of the same key compare against the claimant's ROW
//    (inputs are immutable during the call) and lower the entry to the SMALLEST position with a 64-bit atomicMin, which
//    is what makes the emitted row the first occurrence without any ordering between workgroups.
//  * An entry of the current batch is recognised by its position (>= the store's row count before the batch), so when
//    most rows of a batch are new the batch's key columns are appended to the store WHOLE and the entries are already
//    final: no pass rewrites them (the random 8-byte index write per new key was a third of the all-distinct case).
//    Rows of such a batch that were duplicates occupy store rows nothing points to; batches with few new keys append
//    only those (compacted) and a small publish pass renumbers their entries.
//  * nullsAreDistinct (UPSERT plans): a row with a NULL among its distinct columns never matches anything; it is
//    emitted and not inserted (hashtable.go allowNullEquality=false).
//
// Roofline: HBM-bound; one random index access per row (the 24 G/s atomics wall for new keys, L2-resident lookups
// for repeated ones) + the output gather.
#include <cmath>
#include <cstring>
#include <memory>

#include "device.h"
#include "radix.h"

namespace crgpu {

static constexpr unsigned long long kDEmpty = ~0ull;
static constexpr uint32_t kDNone = 0xFFFFFFFFu;
static constexpr uint32_t kDLost = 0xFFFFFFFEu;  // slot_of: the row saw an earlier row of its key in this batch
static constexpr uint32_t kDMaxCap = 1u << 30;

struct DHash {
  uint32_t idx;    // index position before masking (30 bits)
  uint32_t upper;  // tag << 8 | null mask
};
__device__ __forceinline__ DHash distinct_hash(const KeyCols &k, int64_t row) {
  uint32_t nm = 0;
  for (int c = 0; c < k.n; c++) nm |= (col_is_null(k.c[c], row) ? 1u : 0u) << c;
  // the reference's rehash (NULL leaves the running hash unchanged; the mask is compared separately), then a finaliser
  const uint64_t h = fmix64(hash_row64(k, row, 1) ^ (static_cast<uint64_t>(nm) << 56));
  DHash r;
  r.idx = static_cast<uint32_t>(h) & (kDMaxCap - 1);
  r.upper = (static_cast<uint32_t>(h >> 30) & 0x7FFFFFu) << 8 | nm;
  return r;
}

// row of the batch vs stored key gid; columns NULL on both sides (equal masks) are skipped
__device__ inline bool row_equals_store(const KeyCols &k, int64_t row, const KeyCols &st, uint32_t gid, uint32_t nm) {
  for (int c = 0; c < k.n; c++) {
    if ((nm >> c) & 1u) continue;
    const DCol &x = k.c[c];
    const DCol &s = st.c[c];
    switch (x.type) {
      case CRGPU_BOOL:
        if (col_get_bool(x, row) != col_get_bool(s, gid)) return false;
        break;
      case CRGPU_INT16:
      case CRGPU_INT32:
      case CRGPU_INT64:
        if (col_get_int(x, row) != col_get_int(s, gid)) return false;
        break;
      case CRGPU_FLOAT64: {
        const double p = col_get_f64(x, row), q = col_get_f64(s, gid);
        if (!(p == q || (p != p && q != q))) return false;
        break;
      }
      case CRGPU_BYTES: {
        const uint32_t xa = x.offsets[row], xl = x.offsets[row + 1] - xa;
        const uint32_t sa = s.offsets[gid], sl = s.offsets[gid + 1] - sa;
        if (xl != sl) return false;
        const uint8_t *px = reinterpret_cast<const uint8_t *>(x.data) + xa;
        const uint8_t *ps = reinterpret_cast<const uint8_t *>(s.data) + sa;
        for (uint32_t t = 0; t < xl; t++)
          if (px[t] != ps[t]) return false;
        break;
      }
      default: return false;
    }
  }
  return true;
}

struct DResolveArgs {
  KeyCols keys;   // distinct columns of the batch
  KeyCols store;  // distinct columns of the keys seen so far (no validity: the entry's mask says which are NULL)
  int64_t n;
  unsigned long long *meta;
  uint32_t mask;
  int32_t nulls_are_distinct;
  int32_t write_words;  // the batch may join the store whole: every row leaves its hash word (or kDEmpty) at gword[base + i]
  uint32_t base;      // store rows before this batch: an entry >= base names row (entry - base) of the batch
  uint32_t *slot_of;  // [n] index slot the row ended at (kDNone: repeats a stored key, or bypasses the table)
  uint8_t *flags;     // [n] preset for rows that bypass the table
  unsigned long long *gword;  // [base + i] hash word of every row (kDEmpty: never indexed); see the mark kernel
};
// one row through the index: see distinct_resolve_kernel
__device__ __forceinline__ void distinct_resolve_row(const DResolveArgs &a, int64_t i, uint32_t *slot_out, uint8_t *flag_out,
                                                     unsigned long long *word_out) {
  const DHash h = distinct_hash(a.keys, i);
  uint32_t slot = kDNone;
  uint8_t flag = 0;
  unsigned long long word = (static_cast<unsigned long long>(h.upper) << 32) | h.idx;
  if (a.nulls_are_distinct && (h.upper & 0xFFu) != 0) {
    flag = 1;  // never equal to anything, never looked up
    word = kDEmpty;
  } else {
    const uint32_t me = a.base + static_cast<uint32_t>(i);
    const unsigned long long claim = (static_cast<unsigned long long>(h.upper) << 32) | me;
    uint32_t idx = h.idx & a.mask;
    for (;;) {  // ends: the index is at most half full (ensured by the host before the launch)
      unsigned long long e = __hip_atomic_load(&a.meta[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e == kDEmpty) e = atomicCAS(&a.meta[idx], kDEmpty, claim);
      if (e == kDEmpty) {
        slot = idx;
        break;
      }
      const uint32_t up = static_cast<uint32_t>(e >> 32), lo = static_cast<uint32_t>(e);
      if (up == h.upper) {
        if (lo >= a.base) {  // claimed in this batch
          if (keys_equal(a.keys, i, a.keys, static_cast<int64_t>(lo - a.base), true)) {
            // same upper word: the smallest position wins.  An entry only ever decreases, so a row behind the
            // position it just read has nothing to do (with few distinct keys nearly every row: without this test
            // 2^27 rows over 1024 keys took 26.7 ms in contended atomics)
            if (me < lo) {
              atomicMin(&a.meta[idx], claim);
              slot = idx;
            } else {
              slot = kDLost;
              word = kDEmpty;
            }
            break;
          }
        } else if (row_equals_store(a.keys, i, a.store, lo, h.upper & 0xFFu)) {
          word = kDEmpty;
          break;  // seen in an earlier batch
        }
      }
      idx = (idx + 1) & a.mask;
    }
  }
  *slot_out = slot;
  *flag_out = flag;
  *word_out = word;
}
// LDSF (one INT64 distinct column without NULLs): the workgroup remembers in LDS the keys it has sent through the index.
// It walks its tiles in ascending order, so a key it remembers from an EARLIER tile belongs to a smaller row: a row that
// finds its key there cannot be a first occurrence and is dropped without touching the index (two dependent divergent
// global loads and a compare).  With few distinct keys that is nearly every row and the pass becomes a stream over the
// key column.  The table (2048 sets of two keys; a set's victim alternates with the tile) is read by the whole tile
// before any of the tile's rows writes it: two barriers per 2048 rows.
template <bool LDSF>
__global__ __launch_bounds__(256) void distinct_resolve_kernel(DResolveArgs a) {
  constexpr int RPT = LDSF ? 8 : 1, kTile = 256 * RPT, kSets = 2048;
  constexpr unsigned long long kNoKey = 0x8000000000000001ull;  // (a row holding this very value is never remembered)
  __shared__ ulonglong2 s_keys[LDSF ? kSets : 1];
  if (LDSF) {
    for (int j = threadIdx.x; j < kSets; j += 256) s_keys[j] = make_ulonglong2(kNoKey, kNoKey);
    __syncthreads();
  }
  const int64_t ntiles = (a.n + kTile - 1) / kTile;
  for (int64_t tile = blockIdx.x, round = 0; tile < ntiles; tile += gridDim.x, round++) {
    unsigned long long key[RPT];
    bool drop[RPT];
    if (LDSF) {
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const int64_t i = tile * kTile + k * 256 + threadIdx.x;
        key[k] = i < a.n ? static_cast<const unsigned long long *>(a.keys.c[0].data)[i] : kNoKey;
      }
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const ulonglong2 e = s_keys[static_cast<uint32_t>(fmix64(key[k])) & (kSets - 1)];
        drop[k] = key[k] != kNoKey && (e.x == key[k] || e.y == key[k]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int64_t i = tile * kTile + k * 256 + threadIdx.x;
      if (i >= a.n) continue;
      uint32_t slot = kDLost;
      uint8_t flag = 0;
      unsigned long long word = kDEmpty;
      if (!(LDSF && drop[k])) {
        distinct_resolve_row(a, i, &slot, &flag, &word);
        if (LDSF && key[k] != kNoKey) {
          // (two rows of the tile may pick the same place: one of them is forgotten, which only costs a later lookup)
          unsigned long long *set = reinterpret_cast<unsigned long long *>(&s_keys[static_cast<uint32_t>(fmix64(key[k])) & (kSets - 1)]);
          const unsigned long long e0 = set[0], e1 = set[1];
          const int way = (e0 == kNoKey || e0 == key[k]) ? 0 : (e1 == kNoKey || e1 == key[k]) ? 1 : static_cast<int>(round & 1);
          set[way] = key[k];
        }
      }
      if (a.write_words) a.gword[a.base + i] = word;
      a.slot_of[i] = slot;
      a.flags[i] = flag;
    }
    if (LDSF) __syncthreads();
  }
}

// a row is emitted iff its entry ended up holding its own position; the hash words of the rows that are not keep
// them out of any later index rebuild (this matters when the batch is appended to the store whole)
__global__ __launch_bounds__(256) void distinct_mark_kernel(const unsigned long long *__restrict__ meta,
                                                            const uint32_t *__restrict__ slot_of, int64_t n, uint32_t base,
                                                            uint8_t *__restrict__ flags, unsigned long long *__restrict__ gword) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t s = slot_of[i];
    if (s >= kDLost) continue;  // decided in the resolve kernel (flag and hash word already final)
    const bool mine = static_cast<uint32_t>(meta[s]) == base + static_cast<uint32_t>(i);
    flags[i] = mine ? 1 : 0;
    if (!mine && gword != nullptr) gword[base + i] = kDEmpty;  // lost the entry to an earlier row after claiming it
  }
}

// emitted row j becomes key g0 + j: its entry now names the stored key; hash word kept for index rebuilds
__global__ __launch_bounds__(256) void distinct_publish_kernel(const int32_t *__restrict__ idx, int64_t m,
                                                               const uint32_t *__restrict__ slot_of, uint32_t g0,
                                                               KeyCols keys, unsigned long long *__restrict__ meta,
                                                               unsigned long long *__restrict__ gword) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < m; j += stride) {
    const int64_t row = idx[j];
    const uint32_t s = slot_of[row];
    const DHash h = distinct_hash(keys, row);
    gword[g0 + j] = s == kDNone ? kDEmpty : (static_cast<unsigned long long>(h.upper) << 32) | h.idx;
    if (s != kDNone) meta[s] = (static_cast<unsigned long long>(h.upper) << 32) | (g0 + static_cast<uint32_t>(j));
  }
}

__global__ __launch_bounds__(256) void distinct_reindex_kernel(const unsigned long long *__restrict__ gword, uint32_t ngroups,
                                                               unsigned long long *__restrict__ meta, uint32_t mask) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x; gid < ngroups; gid += stride) {
    const unsigned long long w = gword[gid];
    if (w == kDEmpty) continue;  // a nullsAreDistinct row: stored, never indexed
    const unsigned long long e = (w & 0xFFFFFFFF00000000ull) | gid;
    uint32_t idx = static_cast<uint32_t>(w) & mask;
    while (atomicCAS(&meta[idx], kDEmpty, e) != kDEmpty) idx = (idx + 1) & mask;
  }
}

// BYTES key store append (the output's key column, or the batch's when it is appended whole): dst[j] = tail + (src[j] - src[0])
__global__ __launch_bounds__(256) void distinct_rebase_offsets_kernel(const uint32_t *__restrict__ src, int64_t n, uint32_t first,
                                                                      uint32_t tail, uint32_t *__restrict__ dst) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j <= n; j += stride) dst[j] = tail + (src[j] - first);
}

// ------------------------------------------------------------------ bulk path: a fresh table fed one large, mostly distinct batch
// Find-or-insert against the HBM index costs one random CAS with its returned value per new key (~10 G/s: 11.6 of the
// 14.3 ms of 2^27 all-distinct rows).  When the distinct column is one INT64 without NULLs the rows {key, position} are
// instead radix-partitioned with the join's build-side kernels (radix.hip: 1024 partitions, then sub-partitions of ~1024
// rows in fixed regions of kDBulkRows) and one workgroup per sub-partition finds, entirely in LDS, the smallest position
// of every key: rows that are not the first of their key clear their flag (the flags start out set, so an all-distinct
// region writes nothing).  The batch then joins the key store whole (or, when fewer than half of its rows are emitted,
// only those), and the hash words and the index are only written if more input follows.
static constexpr uint32_t kDBulkRows = 2048, kDBulkSlots = 4096;
static constexpr int64_t kDBulkMinRows = int64_t(1) << 23, kDBulkMaxRows = int64_t(1) << 27;
__global__ __launch_bounds__(1024) void distinct_bulk_kernel(const ulonglong2 *__restrict__ rows_all, const uint32_t *__restrict__ counts,
                                                             uint8_t *__restrict__ flags) {
  // slot -> the slot's KEY, claimed by a 64-bit compare-and-swap against the empty marker (comparing against the row that
  // claimed the slot was a global load in every lookup that met a taken slot, and nearly every wave has such a lane);
  // the one key equal to the marker has the slot behind the others
  __shared__ unsigned long long kkey[kDBulkSlots];
  __shared__ uint32_t minpos[kDBulkSlots + 1];  // slot -> smallest input position of the key
  constexpr unsigned long long kNoKey = ~0ull;
  const uint32_t n = counts[blockIdx.x] < kDBulkRows ? counts[blockIdx.x] : kDBulkRows;  // (overflow is reported by the split)
  if (n == 0) return;
  const ulonglong2 *rows = rows_all + static_cast<int64_t>(blockIdx.x) * kDBulkRows;
  constexpr int kRows = kDBulkRows / 1024;
  ulonglong2 rr[kRows];
#pragma unroll
  for (int k = 0; k < kRows; k++) {  // (requested before the index is cleared and before the first lookup)
    const uint32_t i = k * 1024 + threadIdx.x;
    if (i < n) rr[k] = rows[i];
  }
  for (uint32_t i = threadIdx.x; i < kDBulkSlots; i += blockDim.x) kkey[i] = kNoKey, minpos[i] = 0xFFFFFFFFu;
  if (threadIdx.x == 0) minpos[kDBulkSlots] = 0xFFFFFFFFu;
  __syncthreads();
  uint32_t slot[kRows], pos[kRows];
#pragma unroll
  for (int k = 0; k < kRows; k++) {
    const uint32_t i = k * 1024 + threadIdx.x;
    slot[k] = 0xFFFFFFFFu;
    if (i < n) {
      const ulonglong2 r = rr[k];
      pos[k] = static_cast<uint32_t>(r.y);
      uint32_t sl = static_cast<uint32_t>(fmix64(r.x ^ 0x5851F42D4C957F2Dull) >> 20) & (kDBulkSlots - 1);
      if (r.x == kNoKey) {
        sl = kDBulkSlots;
      } else {
        for (;;) {  // ends: at most kDBulkRows keys in twice as many slots
          unsigned long long old = kkey[sl];
          if (old == kNoKey) {
            old = atomicCAS(&kkey[sl], kNoKey, r.x);
            if (old == kNoKey) break;
          }
          if (old == r.x) break;
          sl = (sl + 1) & (kDBulkSlots - 1);
        }
      }
      atomicMin(&minpos[sl], pos[k]);
      slot[k] = sl;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kRows; k++)
    if (slot[k] != 0xFFFFFFFFu && minpos[slot[k]] != pos[k]) flags[pos[k]] = 0;
}

// Part form: tens to a few thousand keys per level-1 partition.  No level-2 split: one workgroup walks its partition in
// tiles and the partition's keys stay in LDS across the tiles; only at the end does the first row of every key set its
// flag (the flags start out clear).  More keys than the LDS index takes raise the overflow flag: attempt abandoned.
static constexpr uint32_t kDPartSlots = 8192, kDPartMaxKeys = 6144;
__global__ __launch_bounds__(1024) void distinct_bulk_part_kernel(const ulonglong2 *__restrict__ rows_all, const uint32_t *__restrict__ starts,
                                                                  const uint32_t *__restrict__ totals, uint8_t *__restrict__ flags,
                                                                  uint32_t *__restrict__ overflow) {
  // The index holds the KEYS (a slot is claimed by a 64-bit compare-and-swap against the empty marker; the one key equal
  // to the marker has a slot of its own), a tile's rows are requested before the first is looked up and the next
  // tile's while this one is processed -- as in agg_bulk_kernel's part form: comparing against the row
  // that claimed a slot tiles ago put a global load into every row's chain.
  __shared__ unsigned long long kkey[kDPartSlots];
  __shared__ uint32_t minpos[kDPartSlots + 1];  // [kDPartSlots]: the key equal to the empty marker
  __shared__ uint32_t s_claims, s_special;
  constexpr unsigned long long kNoKey = ~0ull;
  const uint32_t n = totals[blockIdx.x];
  if (n == 0) return;
  const ulonglong2 *rows = rows_all + starts[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < kDPartSlots; i += blockDim.x) kkey[i] = kNoKey, minpos[i] = 0xFFFFFFFFu;
  if (threadIdx.x == 0) s_claims = 0, s_special = 0, minpos[kDPartSlots] = 0xFFFFFFFFu;
  ulonglong2 nxt[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t i = k * 1024 + threadIdx.x;
    if (i < n) nxt[k] = rows[i];
  }
  __syncthreads();
  for (uint32_t t0 = 0; t0 < n; t0 += 2048) {
    ulonglong2 rr[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      rr[k] = nxt[k];
      const uint32_t j = t0 + 2048 + k * 1024 + threadIdx.x;
      if (j < n) nxt[k] = rows[j];
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint32_t i = t0 + k * 1024 + threadIdx.x;
      if (i < n) {
        const ulonglong2 r = rr[k];
        uint32_t sl = static_cast<uint32_t>(fmix64(r.x ^ 0x5851F42D4C957F2Dull) >> 20) & (kDPartSlots - 1);
        if (r.x == kNoKey) {
          sl = kDPartSlots;
          if (atomicCAS(&s_special, 0u, 1u) == 0u) atomicAdd(&s_claims, 1u);
        } else {
          for (uint32_t probes = 0;; probes++) {
            if (probes == kDPartSlots) {  // full of other keys
              atomicAdd(&s_claims, kDPartMaxKeys + 1);
              sl = 0xFFFFFFFFu;
              break;
            }
            unsigned long long old = kkey[sl];
            if (old == kNoKey) {
              old = atomicCAS(&kkey[sl], kNoKey, r.x);
              if (old == kNoKey) {
                atomicAdd(&s_claims, 1u);
                break;
              }
            }
            if (old == r.x) break;
            sl = (sl + 1) & (kDPartSlots - 1);
          }
        }
        if (sl != 0xFFFFFFFFu) atomicMin(&minpos[sl], static_cast<uint32_t>(r.y));
      }
    }
    __syncthreads();
    if (s_claims > kDPartMaxKeys) {
      if (threadIdx.x == 0) atomicOr(overflow, 1u);
      return;
    }
  }
  __syncthreads();
  if (s_claims > kDPartMaxKeys) {  // a wave that was ahead overflowed after the others had passed the check above
    if (threadIdx.x == 0) atomicOr(overflow, 1u);
    return;
  }
  for (uint32_t i = threadIdx.x; i < kDPartSlots + 1; i += blockDim.x)
    if (minpos[i] != 0xFFFFFFFFu) flags[minpos[i]] = 1;
}

// hash words of a batch that joins the store whole: emitted rows leave theirs, the others stay out of index rebuilds
__global__ __launch_bounds__(256) void distinct_bulk_words_kernel(KeyCols keys, int64_t n, const uint8_t *__restrict__ flags,
                                                                  unsigned long long *__restrict__ gword) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const DHash h = distinct_hash(keys, i);
    gword[i] = (flags == nullptr || flags[i]) ? (static_cast<unsigned long long>(h.upper) << 32) | h.idx : kDEmpty;
  }
}

// sets *flag when one of the first n bits of a validity bitmap is clear
__global__ __launch_bounds__(256) void distinct_bitmap_has_null_kernel(const uint8_t *__restrict__ valid, int64_t n, uint32_t *__restrict__ flag) {
  const int64_t nbytes = (n + 7) / 8;
  bool bad = false;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < nbytes; i += static_cast<int64_t>(gridDim.x) * 256) {
    const uint32_t want = (i == nbytes - 1 && (n & 7)) ? ((1u << (n & 7)) - 1u) : 0xFFu;
    bad |= (valid[i] & want) != want;
  }
  if (bad) *flag = 1u;
}

__global__ void distinct_sample_kernel(const long long *__restrict__ keys, int64_t stride, int nsample, long long *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nsample) out[i] = keys[static_cast<int64_t>(i) * stride];
}

static inline int dgrid(int64_t n) {
  int64_t want = (n + 255) / 256;
  const int64_t cap = static_cast<int64_t>(num_cus()) * 8;
  if (want < 1) want = 1;
  return static_cast<int>(want < cap ? want : cap);
}

}  // namespace crgpu

using namespace crgpu;

struct crgpu_distinct : crgpu::Handle {
  std::vector<int> types, dcols;
  bool nulls_are_distinct = false, error_on_dup = false;
  bool expect_new = true;  // the last batch was mostly new keys (the first one is assumed to be)
  uint32_t cap = 0;      // index entries
  uint32_t ngroups = 0;  // rows of the key store (the keys seen so far, plus the duplicates inside batches appended whole)
  size_t gcap = 0;       // key store capacity (keys)
  DevBuf meta, gword, slot_of, flags, idx, scratch;
  RadixTable radix;          // partition buffers of the bulk path
  bool index_stale = false;  // the bulk path appended keys whose index entries do not exist yet (rebuilt from gword)
  int64_t words_pending = 0;  // ... and the hash words of its first so many stored keys are not written either: only an
                              // index rebuild reads them, and an input that came in one batch never causes one
  struct KS { DevBuf data, offsets; uint64_t tail = 0; };
  std::vector<std::unique_ptr<KS>> ks;
  std::vector<OutCol> out;
  std::vector<crgpu_dcol> views;
};

static crgpu_status distinct_ensure(crgpu_distinct *d, size_t need_groups, bool with_index = true) {
  hipStream_t st = stream();
  if (need_groups > d->gcap) {
    size_t g = d->gcap ? d->gcap : 1024;
    while (g < need_groups) g *= 2;
    for (size_t k = 0; k < d->dcols.size(); k++) {
      const int t = d->types[d->dcols[k]];
      if (t == CRGPU_BYTES) {
        const bool fresh = d->ks[k]->offsets.p == nullptr;
        CRGPU_TRY(d->ks[k]->offsets.ensure((g + 1) * 4, true));
        if (fresh) CRGPU_HIP(hipMemsetAsync(d->ks[k]->offsets.p, 0, 4, st));
      } else {
        CRGPU_TRY(d->ks[k]->data.ensure(g * type_width(t), true));
      }
    }
    CRGPU_TRY(d->gword.ensure(g * 8, true));
    d->gcap = g;
  }
  if (!with_index) return CRGPU_OK;
  uint64_t want = 1024;
  while (want < 2 * static_cast<uint64_t>(need_groups)) want *= 2;
  // (the bound counts STORED rows: a mostly-new batch joins the key store whole, duplicates included, and the index is
  // sized for the store plus the incoming batch before the batch has been looked at)
  if (want > kDMaxCap)
    return fail(CRGPU_ERR_OOM, "crgpu_dev_distinct_consume: the index bound of 2^29 stored rows + incoming batch is reached on this GPU (spill: ExportBuffered)");
  if (want > d->cap || d->index_stale) {
    if (want < d->cap) want = d->cap;
    CRGPU_TRY(d->meta.ensure(static_cast<size_t>(want) * 8));
    CRGPU_HIP(hipMemsetAsync(d->meta.p, 0xFF, static_cast<size_t>(want) * 8, st));
    d->cap = static_cast<uint32_t>(want);
    d->index_stale = false;
    if (d->words_pending > 0) {  // (the bulk path: one INT64 distinct column without NULLs, every stored row an emitted one)
      KeyCols keys;
      std::memset(&keys, 0, sizeof(keys));
      keys.n = 1;
      keys.c[0] = DCol{CRGPU_INT64, 0, d->ks[0]->data.p, nullptr, nullptr};
      hipLaunchKernelGGL(distinct_bulk_words_kernel, dim3(dgrid(d->words_pending)), dim3(256), 0, st, keys, d->words_pending,
                         static_cast<const uint8_t *>(nullptr), d->gword.as<unsigned long long>());
      CRGPU_HIP(hipGetLastError());
      d->words_pending = 0;
    }
    if (d->ngroups > 0) {
      ProfScope prof("distinct_reindex");
      hipLaunchKernelGGL(distinct_reindex_kernel, dim3(dgrid(d->ngroups)), dim3(256), 0, st, d->gword.as<unsigned long long>(),
                         d->ngroups, d->meta.as<unsigned long long>(), d->cap - 1);
      CRGPU_HIP(hipGetLastError());
    }
  }
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_distinct_create(const int32_t *types, int32_t n_cols, const int32_t *distinct_cols,
                                                  int32_t n_distinct, int32_t nulls_are_distinct, int32_t error_on_dup,
                                                  crgpu_distinct **out) {
  CRGPU_TRY(ensure_init());
  if (types == nullptr || distinct_cols == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_distinct_create: null argument");
  if (n_cols < 1 || n_cols > kMaxCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_distinct_create: column count out of range");
  if (n_distinct < 1 || n_distinct > kMaxKeyCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_distinct_create: between 1 and 8 distinct columns");
  std::unique_ptr<crgpu_distinct> d(new crgpu_distinct());
  {
    crgpu_status _os = d->open();
    if (_os != CRGPU_OK) {  return _os; }
  }
  CRGPU_ENTER(d.get());
  for (int i = 0; i < n_cols; i++) {
    if (!valid_type(types[i])) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_distinct_create: unsupported column type");
    d->types.push_back(types[i]);
  }
  for (int i = 0; i < n_distinct; i++) {
    if (distinct_cols[i] < 0 || distinct_cols[i] >= n_cols) return fail(CRGPU_ERR_INVALID, "crgpu_dev_distinct_create: distinct column ordinal out of range");
    d->dcols.push_back(distinct_cols[i]);
    d->ks.emplace_back(new crgpu_distinct::KS());
  }
  d->nulls_are_distinct = nulls_are_distinct != 0;
  d->error_on_dup = error_on_dup != 0;
  d->out = std::vector<OutCol>(n_cols);
  d->views.resize(n_cols);
  *out = d.release();
  return CRGPU_OK;
}

extern "C" void crgpu_dev_distinct_destroy(crgpu_distinct *d) {
  if (d == nullptr) return;
  if (d->st != nullptr) (void)hipStreamSynchronize(d->st);  // its buffers return to the pool: no kernel may still use them
  delete d;
}

extern "C" crgpu_status crgpu_dev_distinct_reset(crgpu_distinct *d) {
  if (d == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_distinct_reset: null handle");
  CRGPU_ENTER(d);
  d->cancelled = 0;
  hipStream_t st = stream();
  if (d->cap) CRGPU_HIP(hipMemsetAsync(d->meta.p, 0xFF, static_cast<size_t>(d->cap) * 8, st));
  d->ngroups = 0;
  d->expect_new = true;
  d->index_stale = false;
  d->words_pending = 0;
  for (auto &k : d->ks) k->tail = 0;
  CRGPU_HIP(hipStreamSynchronize(st));
  return CRGPU_OK;
}

// Bulk path (see distinct_bulk_kernel).  *done = true: d->flags holds the emit flag of every row; the caller compacts,
// gathers, appends the emitted rows (or the whole batch) to the store and writes or defers their hash words.
static crgpu_status distinct_try_bulk(crgpu_distinct *d, const std::vector<DCol> &in, int64_t n, bool *done) {
  *done = false;
  static const bool on = getenv("CRGPU_DISTINCT_BULK") == nullptr || getenv("CRGPU_DISTINCT_BULK")[0] != '0';
  if (!on || d->ngroups != 0 || n < kDBulkMinRows || n > kDBulkMaxRows || d->dcols.size() != 1) return CRGPU_OK;
  DCol key = in[d->dcols[0]];
  if (key.type != CRGPU_INT64) return CRGPU_OK;
  hipStream_t st = stream();
  if (key.valid != nullptr) {  // a bitmap without a NULL in it (the output of a join or of a local aggregation) is as good as none
    CRGPU_TRY(d->radix.flag.ensure(4));
    CRGPU_HIP(hipMemsetAsync(d->radix.flag.p, 0, 4, st));
    hipLaunchKernelGGL(distinct_bitmap_has_null_kernel, dim3(dgrid((n + 7) / 8)), dim3(256), 0, st, key.valid, n, d->radix.flag.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
    uint32_t has_null = 0;
    CRGPU_HIP(hipMemcpyAsync(&has_null, d->radix.flag.p, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (has_null) return CRGPU_OK;
    key.valid = nullptr;
  }
  // mostly distinct?  8192 keys at a fixed stride; up to ~32 rows per key the regions hold (as the aggregator's sub form)
  constexpr int kSample = 8192;
  std::vector<int64_t> sample(kSample);
  CRGPU_TRY(d->scratch.ensure(static_cast<size_t>(kSample) * 8));
  hipLaunchKernelGGL(distinct_sample_kernel, dim3(kSample / 256), dim3(256), 0, st, static_cast<const long long *>(key.data), n / kSample,
                     kSample, d->scratch.as<long long>());
  CRGPU_HIP(hipGetLastError());
  CRGPU_HIP(hipMemcpyAsync(sample.data(), d->scratch.p, static_cast<size_t>(kSample) * 8, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  const size_t tab = static_cast<size_t>(kSample) * 4;
  std::vector<int64_t> tkey(tab);
  std::vector<int32_t> tused(tab, 0);
  int64_t distinct = 0, maxrun = 0;
  for (int i = 0; i < kSample; i++) {
    size_t at = static_cast<size_t>((static_cast<uint64_t>(sample[i]) * 0x9E3779B97F4A7C15ull) >> 40) & (tab - 1);
    while (tused[at] && tkey[at] != sample[i]) at = (at + 1) & (tab - 1);
    if (!tused[at]) {
      tkey[at] = sample[i];
      distinct++;
    }
    if (++tused[at] > maxrun) maxrun = tused[at];
  }
  int64_t dmax = 16 * (int64_t(kSample) * kSample) / n;
  if (dmax < 8) dmax = 8;
  if (dmax > kSample / 4) dmax = kSample / 4;
  const bool sub_form = kSample - distinct <= dmax;
  if (!sub_form) {
    // part form: G from G (1 - exp(-s / G)) = distinct; it takes ~3 ... ~4500 keys per level-1 partition and no key far
    // more frequent than the others (its rows would serialise on one LDS word)
    double lo = 1.0, hi = 1e12;
    for (int it = 0; it < 60; it++) {
      const double mid = std::sqrt(lo * hi);
      if (mid * (1.0 - std::exp(-static_cast<double>(kSample) / mid)) < static_cast<double>(distinct)) lo = mid;
      else hi = mid;
    }
    const double per_part = hi / kRadixP1, mu = static_cast<double>(kSample) / hi;
    if (per_part < 3.0 || per_part * 1.3 + 4.0 * std::sqrt(per_part) + 16.0 > static_cast<double>(kDPartMaxKeys)) return CRGPU_OK;
    if (static_cast<double>(maxrun) > std::max(8.0, 3.0 * mu + 6.0)) return CRGPU_OK;
  }
  RadixSrc src;
  std::memset(&src, 0, sizeof(src));
  src.kp.n = 1;
  src.kp.c[0] = key;
  src.kp.range[0] = ~0ull;
  src.pay = nullptr;  // the payload is the row's position
  src.fast = (reinterpret_cast<uintptr_t>(key.data) & 15u) == 0 ? 1 : 0;
  if (sub_form) {
    bool ok = false;
    // one region per ~1024 rows whatever the row count (the kernel's time follows the number of regions; see try_bulk in agg.hip)
    int64_t want = (n + (int64_t(kRadixP1) << 10) - 1) / (int64_t(kRadixP1) << 10);
    if (want > 128) want = 128;
    if ((want & (want - 1)) == 0) want = 0;  // a power of two: the split by hash bits gives the same count
    CRGPU_TRY(radix_partition_split(d->radix, src, n, &ok, kDBulkRows, 10, 7, static_cast<uint32_t>(want)));
    if (!ok) return CRGPU_OK;  // a region overflowed: many rows per key after all
    CRGPU_HIP(hipMemsetAsync(d->flags.p, 1, static_cast<size_t>(n), st));
    ProfScope prof("distinct_bulk");
    hipLaunchKernelGGL(distinct_bulk_kernel, dim3(static_cast<unsigned>(kRadixP1) * d->radix.nsub), dim3(1024), 0, st,
                       d->radix.rows_b.as<ulonglong2>(), d->radix.sub_count.as<uint32_t>(), d->flags.as<uint8_t>());
    CRGPU_HIP(hipGetLastError());
  } else {
    CRGPU_TRY(radix_partition_level1(d->radix, src, n));
    CRGPU_TRY(d->radix.flag.ensure(4));
    CRGPU_HIP(hipMemsetAsync(d->radix.flag.p, 0, 4, st));
    CRGPU_HIP(hipMemsetAsync(d->flags.p, 0, static_cast<size_t>(n), st));
    {
      ProfScope prof("distinct_bulk_part");
      hipLaunchKernelGGL(distinct_bulk_part_kernel, dim3(kRadixP1), dim3(1024), 0, st, d->radix.rows_a.as<ulonglong2>(),
                         d->radix.starts.as<uint32_t>(), d->radix.totals.as<uint32_t>(), d->flags.as<uint8_t>(),
                         d->radix.flag.as<uint32_t>());
      CRGPU_HIP(hipGetLastError());
    }
    uint32_t overflow = 0;
    CRGPU_HIP(hipMemcpyAsync(&overflow, d->radix.flag.p, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (overflow) return CRGPU_OK;  // more keys per partition than the sample suggested
  }
  *done = true;
  return CRGPU_OK;
}

static crgpu_status crgpu_dev_distinct_consume_impl(crgpu_distinct *d, const crgpu_dcol *cols, int64_t n, crgpu_dev_result *res) {
  if (d == nullptr || res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_distinct_consume: null argument");
  CRGPU_ENTER(d);
  CRGPU_CHECK_CANCEL(d);
  // one batch may not use up the whole index bound by itself (the index is sized for stored rows + n): 2^28 rows per call
  if (n < 0 || n > (int64_t(1) << 28)) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_distinct_consume: at most 2^28 rows per call; feed the input in smaller batches");
  if (n > 0 && cols == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_distinct_consume: null columns");
  const int nc = static_cast<int>(d->types.size());
  hipStream_t st = stream();
  std::vector<DCol> in;
  for (int c = 0; c < nc; c++) {
    DCol x{d->types[c], 0, nullptr, nullptr, nullptr};
    if (n > 0) {
      if (cols[c].type != d->types[c]) return fail(CRGPU_ERR_INVALID, "crgpu_dev_distinct_consume: column type differs from the spec");
      if (cols[c].type == CRGPU_BYTES && cols[c].offsets == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_distinct_consume: BYTES column without offsets");
      if (cols[c].type != CRGPU_BYTES && cols[c].data == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_distinct_consume: null data pointer");
      x = to_dcol(cols[c]);
    }
    in.push_back(x);
  }
  int64_t m = 0, m_store = 0;
  bool whole = false;
  bool bulk = false;
  if (n > 0) {
    CRGPU_TRY(d->flags.ensure(static_cast<size_t>(n)));
    CRGPU_TRY(d->idx.ensure(static_cast<size_t>(n) * 4));
    if (d->ngroups == 0 && n >= kDBulkMinRows) {
      CRGPU_TRY(distinct_ensure(d, static_cast<size_t>(n), false));  // key store and hash words only
      CRGPU_TRY(distinct_try_bulk(d, in, n, &bulk));
    }
  }
  // (every column fixed-width and without NULLs: a batch whose rows are all emitted is copied, not gathered)
  bool plain_cols = true;
  for (int c = 0; c < nc; c++) plain_cols = plain_cols && d->types[c] != CRGPU_BYTES && in[c].valid == nullptr;
  if (bulk) {
    CRGPU_TRY(select_flagged(d->flags.as<uint8_t>(), n, d->idx.as<int32_t>(), d->scratch, &m, plain_cols));
    whole = m * 2 > n;
    d->expect_new = whole;
    d->index_stale = true;
  } else if (n > 0) {
    CRGPU_TRY(distinct_ensure(d, static_cast<size_t>(d->ngroups) + static_cast<size_t>(n)));
    CRGPU_TRY(d->slot_of.ensure(static_cast<size_t>(n) * 4));
    DResolveArgs a;
    std::memset(&a, 0, sizeof(a));
    a.keys.n = a.store.n = static_cast<int>(d->dcols.size());
    for (size_t k = 0; k < d->dcols.size(); k++) {
      a.keys.c[k] = in[d->dcols[k]];
      a.store.c[k] = DCol{d->types[d->dcols[k]], 0, d->ks[k]->data.p, nullptr, d->ks[k]->offsets.as<uint32_t>()};
    }
    a.n = n;
    a.meta = d->meta.as<unsigned long long>();
    a.mask = d->cap - 1;
    a.nulls_are_distinct = d->nulls_are_distinct ? 1 : 0;
    a.base = d->ngroups;
    a.gword = d->gword.as<unsigned long long>();
    a.write_words = d->expect_new ? 1 : 0;
    a.slot_of = d->slot_of.as<uint32_t>();
    a.flags = d->flags.as<uint8_t>();
    {
      ProfScope prof("distinct_resolve");
      static const bool ldsf_on = getenv("CRGPU_DISTINCT_LDS") == nullptr || getenv("CRGPU_DISTINCT_LDS")[0] != '0';
      const bool ldsf = ldsf_on && a.keys.n == 1 && a.keys.c[0].type == CRGPU_INT64 && a.keys.c[0].valid == nullptr && n >= (int64_t(1) << 16);
      if (ldsf) hipLaunchKernelGGL(distinct_resolve_kernel<true>, dim3(dgrid(n)), dim3(256), 0, st, a);
      else hipLaunchKernelGGL(distinct_resolve_kernel<false>, dim3(dgrid(n)), dim3(256), 0, st, a);
      CRGPU_HIP(hipGetLastError());
      hipLaunchKernelGGL(distinct_mark_kernel, dim3(dgrid(n)), dim3(256), 0, st, d->meta.as<unsigned long long>(),
                         d->slot_of.as<uint32_t>(), n, d->ngroups, d->flags.as<uint8_t>(),
                         d->expect_new ? d->gword.as<unsigned long long>() : nullptr);
      CRGPU_HIP(hipGetLastError());
    }
    CRGPU_TRY(select_flagged(d->flags.as<uint8_t>(), n, d->idx.as<int32_t>(), d->scratch, &m));
    // most rows are new: the batch's key columns join the store as they are and the entries are final -- possible only
    // if the rows left their hash words, which they do when the previous batch (or nothing yet) suggested it
    whole = d->expect_new && m * 2 > n;
    d->expect_new = m * 2 > n;
    if (m > 0 && !whole) {
      ProfScope prof("distinct_publish");
      hipLaunchKernelGGL(distinct_publish_kernel, dim3(dgrid(m)), dim3(256), 0, st, d->idx.as<int32_t>(), m,
                         d->slot_of.as<uint32_t>(), d->ngroups, a.keys, d->meta.as<unsigned long long>(),
                         d->gword.as<unsigned long long>());
      CRGPU_HIP(hipGetLastError());
    }
  }
  // the output: every column of the emitted rows, in input order.  A fixed-width distinct column without NULLs is not
  // gathered into a buffer of its own when the rows the key store takes are exactly the emitted ones (every batch but a
  // mostly-new one with repeats inside): its values are written where the store wants them and handed out from there
  // (valid until the next call on the handle, like every result: only a later call can move the store)
  std::vector<void *> from_store(static_cast<size_t>(nc), nullptr);
  {
    ProfScope prof("distinct_gather");
    for (int c = 0; c < nc; c++) {
      int kc = -1, times = 0;
      for (size_t k = 0; k < d->dcols.size(); k++)
        if (d->dcols[k] == c) kc = static_cast<int>(k), times++;
      if (n > 0 && m > 0 && times == 1 && (!whole || m == n) && d->types[c] != CRGPU_BYTES && in[c].valid == nullptr) {
        const size_t w = static_cast<size_t>(type_width(d->types[c]));
        void *dst = d->ks[static_cast<size_t>(kc)]->data.as<uint8_t>() + static_cast<size_t>(d->ngroups) * w;
        if (m == n) CRGPU_HIP(hipMemcpyAsync(dst, in[c].data, static_cast<size_t>(n) * w, hipMemcpyDeviceToDevice, st));
        else CRGPU_TRY(gather_fixed(in[c], d->idx.as<int32_t>(), m, dst));
        d->out[c].type = d->types[c];
        d->out[c].nullable = false;
        from_store[static_cast<size_t>(c)] = dst;
        continue;
      }
      if (n == 0 || m == 0) {
        OutCol &o = d->out[c];
        o.type = d->types[c];
        o.nullable = false;
        CRGPU_TRY(o.data.ensure(16));
        if (o.type == CRGPU_BYTES) {
          CRGPU_TRY(o.offsets.ensure(4));
          CRGPU_HIP(hipMemsetAsync(o.offsets.p, 0, 4, st));
        }
      } else if (m == n && d->types[c] != CRGPU_BYTES && in[c].valid == nullptr) {  // every row is emitted: a copy
        OutCol &o = d->out[c];
        o.type = d->types[c];
        o.nullable = false;
        const size_t bytes = static_cast<size_t>(n) * type_width(o.type);
        CRGPU_TRY(o.data.ensure(bytes));
        CRGPU_HIP(hipMemcpyAsync(o.data.p, in[c].data, bytes, hipMemcpyDeviceToDevice, st));
      } else {
        CRGPU_TRY(materialize(in[c], d->idx.as<int32_t>(), m, false, d->out[c], d->scratch));
      }
    }
  }
  // hash words of the rows that join the store: the batch's (emitted rows only) or the output's.  When every stored row
  // is an emitted one they are left for the index rebuild to write, from the store's keys -- if there ever is one
  if (bulk && (!whole || m == n)) {
    d->words_pending = whole ? n : m;
  } else if (bulk) {
    KeyCols keys;
    std::memset(&keys, 0, sizeof(keys));
    keys.n = 1;
    keys.c[0] = whole ? in[d->dcols[0]] : DCol{CRGPU_INT64, 0, d->out[d->dcols[0]].data.p, nullptr, nullptr};
    const int64_t rows = whole ? n : m;
    hipLaunchKernelGGL(distinct_bulk_words_kernel, dim3(dgrid(rows)), dim3(256), 0, st, keys, rows,
                       whole ? d->flags.as<uint8_t>() : nullptr, d->gword.as<unsigned long long>());
    CRGPU_HIP(hipGetLastError());
  }
  // the key store grows by the distinct columns of the output -- or of the whole batch
  for (size_t k = 0; m > 0 && k < d->dcols.size(); k++) {
    const OutCol &o = d->out[d->dcols[k]];
    const DCol &x = in[d->dcols[k]];
    crgpu_distinct::KS &s = *d->ks[k];
    const int64_t rows = whole ? n : m;
    if (o.type == CRGPU_BYTES) {
      const uint32_t *offs = whole ? x.offsets : o.offsets.as<uint32_t>();
      const uint8_t *bytes = whole ? static_cast<const uint8_t *>(x.data) : o.data.as<uint8_t>();
      uint32_t ends[2] = {0, 0};
      CRGPU_HIP(hipMemcpyAsync(&ends[0], offs, 4, hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipMemcpyAsync(&ends[1], offs + rows, 4, hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipStreamSynchronize(st));
      const uint32_t total = ends[1] - ends[0];
      if (s.tail + total > 0xFFFFFFFFull) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_distinct_consume: more than 4 GiB of BYTES keys");
      CRGPU_TRY(s.data.ensure(static_cast<size_t>(s.tail + total + 16), true));
      if (total) CRGPU_HIP(hipMemcpyAsync(s.data.as<uint8_t>() + s.tail, bytes + ends[0], total, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(distinct_rebase_offsets_kernel, dim3(dgrid(rows + 1)), dim3(256), 0, st, offs, rows, ends[0],
                         static_cast<uint32_t>(s.tail), s.offsets.as<uint32_t>() + d->ngroups);
      CRGPU_HIP(hipGetLastError());
      s.tail += total;
    } else if (from_store[static_cast<size_t>(d->dcols[k])] == nullptr) {
      const size_t w = static_cast<size_t>(type_width(o.type));
      CRGPU_HIP(hipMemcpyAsync(s.data.as<uint8_t>() + static_cast<size_t>(d->ngroups) * w, whole ? x.data : o.data.p,
                               static_cast<size_t>(rows) * w, hipMemcpyDeviceToDevice, st));
    }
  }
  if (whole) m_store = n;
  else m_store = m;
  d->ngroups += static_cast<uint32_t>(m_store);
  CRGPU_HIP(hipStreamSynchronize(st));
  for (int c = 0; c < nc; c++) {
    d->views[c] = d->out[c].view();
    if (from_store[static_cast<size_t>(c)] != nullptr) d->views[c].data = from_store[static_cast<size_t>(c)];
  }
  res->n = m;
  res->ncols = nc;
  res->_pad = 0;
  res->cols = d->views.data();
  if (d->error_on_dup && m < n) return fail(CRGPU_ERR_DUPLICATE, "crgpu_dev_distinct_consume: duplicate tuple (errorOnDup)");
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_distinct_consume(crgpu_distinct *d, const crgpu_dcol *cols, int64_t n, crgpu_dev_result *res) {
  const crgpu_status s_ = crgpu_dev_distinct_consume_impl(d, cols, n, res);
  if (s_ == CRGPU_OK && d != nullptr) note_rows(d, n, res->n);
  return s_;
}
