// agg.hip -- device-resident online hash aggregation.
//
// Replaces colexec.hashAggregator.onlineAgg (pkg/sql/colexec/hash_aggregator.go:354-453)
// and the per-(group, function) accumulator objects of colexecagg
// (pkg/sql/colexec/colexecagg/hash_{sum,sum_int,avg,count,min_max,any_not_null,bool_and_or}_agg.eg.go).
//
// Design (not the reference's 4096-tuple buffer + equality chains + virtual Compute calls):
//  * ONE open-addressing table keyed by the group columns.  A slot holds either
//    EMPTY, OCCUPIED (key lives in the per-slot key store) or -- while a chunk is
//    being consumed -- the chunk row id of the row that claimed it.  Input columns
//    are immutable during the kernel, so "compare against the claimant's row" needs
//    no cross-workgroup publication protocol: the only mutable words are the slots,
//    and those are touched with atomics only.
//  * find-or-insert and aggregation are fused in one pass over the chunk; every
//    accumulator is a per-slot 64-bit word updated with HBM atomics
//    (global_atomic_add_f64 / add_u64 / smin / smax / umin / umax / and / or).
//    While few groups exist the lanes of a wave that hit the same slot are first
//    combined in registers (one atomic per wave and group instead of 64).
//  * after each chunk the newly claimed slots get their keys copied into the key
//    store; when the table passes 50 % load it is rebuilt at twice the size.
//    Rows that cannot be placed (probe limit) are deferred and replayed after growth.
//  * SUM_INT accumulates in 128 bits (two words) so the overflow test does not
//    depend on summation order: CRGPU_ERR_INT_OUT_OF_RANGE iff the exact group total
//    leaves int64 (the reference errors on transient overflow in arrival order,
//    hash_sum_int_agg.eg.go:295-298; see DESIGN.md "Divergences").
//  * float SUM/AVG use fp64 atomics: order-dependent in the last bits (stated
//    tolerance rel 1e-9).
//
// Roofline: HBM-bound (streaming input + random 8-byte atomics).
#include <cstring>
#include <memory>

#include "device.h"

namespace crgpu {

static constexpr uint32_t kSlotEmpty = 0xFFFFFFFFu;
static constexpr uint32_t kSlotOccupied = 0xFFFFFFFEu;
static constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
static constexpr int kMaxAggs = 64;
static constexpr int kMaxProbe = 512;

enum AccKind : int32_t {
  ACC_COUNT_ROWS = 0, ACC_COUNT, ACC_SUM_F64, ACC_AVG_F64, ACC_SUM_I128, ACC_MIN_I64, ACC_MAX_I64, ACC_MIN_F64,
  ACC_MAX_F64, ACC_BOOL_AND, ACC_BOOL_OR, ACC_ANY_KEY, ACC_ANY_ROW
};

struct AggOpDev {
  int32_t kind;
  int32_t has_filter;
  int32_t key_idx;  // ACC_ANY_KEY: which group column
  int32_t _pad;
  DCol col;
  DCol filter;
  unsigned long long *w0;
  unsigned long long *w1;
  uint8_t *seen;
};

struct KeyStoreCol {
  int32_t type;
  int32_t _pad;
  uint8_t *data;       // fixed-width: value per slot
  uint8_t *present;    // one byte per slot (0 => NULL key)
  uint64_t *b_off;     // BYTES: offset into b_buf per slot
  uint32_t *b_len;     // BYTES: length per slot
  uint8_t *b_buf;      // BYTES: append-only payload buffer
};
struct KeyStore {
  int32_t n;
  int32_t _pad;
  KeyStoreCol c[kMaxKeyCols];
};

// total-order map for float MIN/MAX: NaN is the smallest value
// (hash_min_max_agg.eg.go:966-981) -> 0; everything else order-preserving and > 0.
__device__ __forceinline__ unsigned long long f64_to_ordered(double f) {
  if (f != f) return 0ull;
  unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(f));
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ordered_to_f64(unsigned long long m) {
  if (m == 0ull) return __longlong_as_double(0x7FF8000000000000ll);
  unsigned long long b = (m >> 63) ? (m & 0x7FFFFFFFFFFFFFFFull) : ~m;
  return __longlong_as_double(static_cast<long long>(b));
}

// chunk row vs stored key of an OCCUPIED slot (NULLs are equal: GROUP BY semantics,
// hash_aggregator.go:269 allowNullEquality=true)
__device__ inline bool key_equals_store(const KeyCols &k, int64_t row, const KeyStore &st, uint32_t slot) {
  for (int c = 0; c < k.n; c++) {
    const DCol &x = k.c[c];
    const KeyStoreCol &s = st.c[c];
    const bool xn = col_is_null(x, row), sn = s.present[slot] == 0;
    if (xn || sn) {
      if (xn && sn) continue;
      return false;
    }
    switch (x.type) {
      case CRGPU_BOOL:
        if (col_get_bool(x, row) != (s.data[slot] != 0)) return false;
        break;
      case CRGPU_INT16:
        if (reinterpret_cast<const int16_t *>(x.data)[row] != reinterpret_cast<const int16_t *>(s.data)[slot]) return false;
        break;
      case CRGPU_INT32:
        if (reinterpret_cast<const int32_t *>(x.data)[row] != reinterpret_cast<const int32_t *>(s.data)[slot]) return false;
        break;
      case CRGPU_INT64:
        if (reinterpret_cast<const int64_t *>(x.data)[row] != reinterpret_cast<const int64_t *>(s.data)[slot]) return false;
        break;
      case CRGPU_FLOAT64: {
        double p = col_get_f64(x, row), q = reinterpret_cast<const double *>(s.data)[slot];
        if (!(p == q || (p != p && q != q))) return false;
        break;
      }
      case CRGPU_BYTES: {
        uint32_t xa = x.offsets[row], xl = x.offsets[row + 1] - xa;
        if (xl != s.b_len[slot]) return false;
        const uint8_t *px = reinterpret_cast<const uint8_t *>(x.data) + xa;
        const uint8_t *ps = s.b_buf + s.b_off[slot];
        for (uint32_t t = 0; t < xl; t++)
          if (px[t] != ps[t]) return false;
        break;
      }
      default: return false;
    }
  }
  return true;
}

struct AggArgs {
  KeyCols keys;
  const KeyStore *store;  // device copy
  const AggOpDev *ops;    // device copy
  int32_t naggs;
  int32_t combine;  // 1: combine lanes of a wave that share a slot before touching memory
  int64_t n;
  unsigned long long seq_base;  // global arrival number of row 0 of this chunk
  uint32_t *slots;
  uint32_t mask;
  int32_t only_deferred;
  uint32_t *hstore;
  uint8_t *deferred;
  unsigned int *counters;  // [0] new representatives, [1] deferred rows
  uint32_t *rep_rows, *rep_slots;
  uint32_t *slot_of_row;  // optional (ANY_NOT_NULL over non-group columns)
};

// ------------------------------------------------------------------ per-row accumulation (one atomic per lane)
__device__ __forceinline__ void add_i128(unsigned long long *lo, unsigned long long *hi, unsigned long long vlo,
                                         unsigned long long vhi) {
  const unsigned long long old = atomicAdd(lo, vlo);
  const unsigned long long carry = (old + vlo) < old ? 1ull : 0ull;
  const unsigned long long d = vhi + carry;
  if (d != 0ull) atomicAdd(hi, d);
}

__device__ inline void accumulate_direct(const AggOpDev &op, int64_t row, uint32_t slot, unsigned long long seq) {
  if (op.has_filter && (col_is_null(op.filter, row) || !col_get_bool(op.filter, row))) return;  // aggregators_util.go:216-234
  if (op.kind == ACC_COUNT_ROWS) {
    atomicAdd(&op.w0[slot], 1ull);
    return;
  }
  if (op.kind == ACC_ANY_KEY) return;
  if (col_is_null(op.col, row)) return;  // aggregates skip NULL inputs
  switch (op.kind) {
    case ACC_COUNT: atomicAdd(&op.w0[slot], 1ull); break;
    case ACC_SUM_F64:
      atomicAdd(reinterpret_cast<double *>(&op.w0[slot]), col_get_f64(op.col, row));
      op.seen[slot] = 1;
      break;
    case ACC_AVG_F64:
      atomicAdd(reinterpret_cast<double *>(&op.w0[slot]), col_get_f64(op.col, row));
      atomicAdd(&op.w1[slot], 1ull);
      break;
    case ACC_SUM_I128: {
      const long long v = col_get_int(op.col, row);
      add_i128(&op.w0[slot], &op.w1[slot], static_cast<unsigned long long>(v), v < 0 ? ~0ull : 0ull);
      op.seen[slot] = 1;
      break;
    }
    case ACC_MIN_I64:
    case ACC_MAX_I64: {
      const long long v = op.col.type == CRGPU_BOOL ? static_cast<long long>(col_get_bool(op.col, row)) : col_get_int(op.col, row);
      if (op.kind == ACC_MIN_I64) atomicMin(reinterpret_cast<long long *>(&op.w0[slot]), v);
      else atomicMax(reinterpret_cast<long long *>(&op.w0[slot]), v);
      op.seen[slot] = 1;
      break;
    }
    case ACC_MIN_F64: atomicMin(&op.w0[slot], f64_to_ordered(col_get_f64(op.col, row))); op.seen[slot] = 1; break;
    case ACC_MAX_F64: atomicMax(&op.w0[slot], f64_to_ordered(col_get_f64(op.col, row))); op.seen[slot] = 1; break;
    case ACC_BOOL_AND: if (!col_get_bool(op.col, row)) atomicAnd(&op.w0[slot], 0ull); op.seen[slot] = 1; break;
    case ACC_BOOL_OR: if (col_get_bool(op.col, row)) atomicOr(&op.w0[slot], 1ull); op.seen[slot] = 1; break;
    case ACC_ANY_ROW: atomicMin(&op.w0[slot], seq); break;  // first non-NULL in arrival order
    default: break;
  }
}

// ------------------------------------------------------------------ wave-combined accumulation
template <typename T, typename F>
__device__ __forceinline__ T wave_reduce(T v, F f) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = f(v, __shfl_xor(v, d, 64));
  return v;
}

// All lanes of the wave call this (converged); `mine` = this lane belongs to the
// group being flushed; the leader lane issues the memory update.
__device__ inline void accumulate_combined(const AggOpDev &op, int64_t row, uint32_t slot, unsigned long long seq,
                                           bool mine, bool leader) {
  bool take = mine;
  if (take && op.has_filter && (col_is_null(op.filter, row) || !col_get_bool(op.filter, row))) take = false;
  if (op.kind == ACC_ANY_KEY) return;
  if (op.kind == ACC_COUNT_ROWS) {
    const unsigned long long c = popc64(__ballot(take));
    if (leader && c) atomicAdd(&op.w0[slot], c);
    return;
  }
  if (take && col_is_null(op.col, row)) take = false;
  const unsigned long long cnt = popc64(__ballot(take));
  if (cnt == 0) return;  // wave-uniform
  switch (op.kind) {
    case ACC_COUNT:
      if (leader) atomicAdd(&op.w0[slot], cnt);
      break;
    case ACC_SUM_F64:
    case ACC_AVG_F64: {
      double v = take ? col_get_f64(op.col, row) : 0.0;
      v = wave_reduce(v, [](double a, double b) { return a + b; });
      if (leader) {
        atomicAdd(reinterpret_cast<double *>(&op.w0[slot]), v);
        if (op.kind == ACC_AVG_F64) atomicAdd(&op.w1[slot], cnt);
        else op.seen[slot] = 1;
      }
      break;
    }
    case ACC_SUM_I128: {
      const long long x = take ? col_get_int(op.col, row) : 0;
      unsigned long long lo = static_cast<unsigned long long>(x), hi = x < 0 ? ~0ull : 0ull;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long olo = __shfl_xor(lo, d, 64), ohi = __shfl_xor(hi, d, 64);
        const unsigned long long s = lo + olo;
        hi = hi + ohi + (s < lo ? 1ull : 0ull);
        lo = s;
      }
      if (leader) {
        add_i128(&op.w0[slot], &op.w1[slot], lo, hi);
        op.seen[slot] = 1;
      }
      break;
    }
    case ACC_MIN_I64:
    case ACC_MAX_I64: {
      const bool mn = op.kind == ACC_MIN_I64;
      long long v = mn ? 0x7FFFFFFFFFFFFFFFll : static_cast<long long>(0x8000000000000000ull);
      if (take) v = op.col.type == CRGPU_BOOL ? static_cast<long long>(col_get_bool(op.col, row)) : col_get_int(op.col, row);
      v = mn ? wave_reduce(v, [](long long a, long long b) { return a < b ? a : b; })
             : wave_reduce(v, [](long long a, long long b) { return a > b ? a : b; });
      if (leader) {
        if (mn) atomicMin(reinterpret_cast<long long *>(&op.w0[slot]), v);
        else atomicMax(reinterpret_cast<long long *>(&op.w0[slot]), v);
        op.seen[slot] = 1;
      }
      break;
    }
    case ACC_MIN_F64:
    case ACC_MAX_F64: {
      const bool mn = op.kind == ACC_MIN_F64;
      unsigned long long v = mn ? ~0ull : 0ull;
      if (take) v = f64_to_ordered(col_get_f64(op.col, row));
      v = mn ? wave_reduce(v, [](unsigned long long a, unsigned long long b) { return a < b ? a : b; })
             : wave_reduce(v, [](unsigned long long a, unsigned long long b) { return a > b ? a : b; });
      if (leader) {
        if (mn) atomicMin(&op.w0[slot], v);
        else atomicMax(&op.w0[slot], v);
        op.seen[slot] = 1;
      }
      break;
    }
    case ACC_BOOL_AND: {
      const bool any_false = __ballot(take && !col_get_bool(op.col, row)) != 0;
      if (leader) {
        if (any_false) atomicAnd(&op.w0[slot], 0ull);
        op.seen[slot] = 1;
      }
      break;
    }
    case ACC_BOOL_OR: {
      const bool any_true = __ballot(take && col_get_bool(op.col, row)) != 0;
      if (leader) {
        if (any_true) atomicOr(&op.w0[slot], 1ull);
        op.seen[slot] = 1;
      }
      break;
    }
    case ACC_ANY_ROW: {
      unsigned long long v = take ? seq : ~0ull;
      v = wave_reduce(v, [](unsigned long long a, unsigned long long b) { return a < b ? a : b; });
      if (leader) atomicMin(&op.w0[slot], v);
      break;
    }
    default: break;
  }
}

// ------------------------------------------------------------------ fused find-or-insert + aggregate
__global__ __launch_bounds__(256) void agg_consume_kernel(AggArgs a) {
  const KeyStore &store = *a.store;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t nround = (a.n + 63) & ~int64_t(63);  // keep whole waves in the loop (wave-level collectives below)
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < nround; row += stride) {
    bool active = row < a.n;
    if (active && a.only_deferred) active = a.deferred[row] != 0;
    uint32_t slot = kNoSlot;
    bool is_new = false, gave_up = false;
    uint32_t h = 0;
    if (active) {
      h = hash_row32(a.keys, row, 1u);
      uint32_t idx = h & a.mask;
      for (int probes = 0;; probes++) {
        uint32_t cur = __hip_atomic_load(&a.slots[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kSlotEmpty) {
          const uint32_t old = atomicCAS(&a.slots[idx], kSlotEmpty, static_cast<uint32_t>(row));
          if (old == kSlotEmpty) {
            slot = idx;
            is_new = true;
            break;
          }
          cur = old;
        }
        const bool eq = cur == kSlotOccupied ? key_equals_store(a.keys, row, store, idx)
                                             : (cur == static_cast<uint32_t>(row) ||
                                                keys_equal(a.keys, row, a.keys, static_cast<int64_t>(cur), true));
        if (eq) {
          slot = idx;
          break;
        }
        idx = (idx + 1) & a.mask;
        if (probes >= kMaxProbe) {
          gave_up = true;
          break;
        }
      }
    }
    // bookkeeping: representatives of new groups and deferred rows (wave-aggregated counters)
    {
      const uint64_t m_new = __ballot(is_new);
      if (m_new) {
        unsigned int base = 0;
        if (lane_id() == static_cast<uint32_t>(__ffsll(static_cast<long long>(m_new)) - 1))
          base = atomicAdd(&a.counters[0], static_cast<unsigned int>(popc64(m_new)));
        base = __shfl(base, __ffsll(static_cast<long long>(m_new)) - 1, 64);
        if (is_new) {
          const unsigned int p = base + rank_in_mask(m_new);
          a.rep_rows[p] = static_cast<uint32_t>(row);
          a.rep_slots[p] = slot;
          a.hstore[slot] = h;
        }
      }
      const uint64_t m_def = __ballot(gave_up);
      if (m_def) {
        if (lane_id() == static_cast<uint32_t>(__ffsll(static_cast<long long>(m_def)) - 1))
          atomicAdd(&a.counters[1], static_cast<unsigned int>(popc64(m_def)));
      }
      if (active) a.deferred[row] = gave_up ? 1 : 0;
    }
    if (a.slot_of_row != nullptr && row < a.n) a.slot_of_row[row] = active ? slot : kNoSlot;
    const unsigned long long seq = a.seq_base + static_cast<unsigned long long>(row);
    if (!a.combine) {
      if (slot != kNoSlot)
        for (int f = 0; f < a.naggs; f++) accumulate_direct(a.ops[f], row, slot, seq);
    } else {
      uint64_t todo = __ballot(slot != kNoSlot);
      while (todo) {  // wave-uniform loop: one iteration per distinct slot in the wave
        const int lead = __ffsll(static_cast<long long>(todo)) - 1;
        const uint32_t s = __shfl(slot, lead, 64);
        const uint64_t peers = __ballot(slot == s);
        const bool mine = slot == s;
        const bool leader = lane_id() == static_cast<uint32_t>(lead);
        for (int f = 0; f < a.naggs; f++) accumulate_combined(a.ops[f], row < a.n ? row : 0, s, seq, mine, leader);
        todo &= ~peers;
      }
    }
  }
}

// copy the keys of newly claimed slots into the key store and mark them OCCUPIED
__global__ __launch_bounds__(256) void agg_fixup_kernel(KeyCols keys, const KeyStore *store_p, unsigned int nrep,
                                                        const uint32_t *__restrict__ rep_rows,
                                                        const uint32_t *__restrict__ rep_slots,
                                                        uint32_t *__restrict__ slots,
                                                        unsigned long long *__restrict__ b_tails) {
  const KeyStore &st = *store_p;
  const unsigned int stride = gridDim.x * blockDim.x;
  for (unsigned int r = blockIdx.x * blockDim.x + threadIdx.x; r < nrep; r += stride) {
    const int64_t row = rep_rows[r];
    const uint32_t slot = rep_slots[r];
    for (int c = 0; c < keys.n; c++) {
      const DCol &x = keys.c[c];
      const KeyStoreCol &s = st.c[c];
      const bool isnull = col_is_null(x, row);
      s.present[slot] = isnull ? 0 : 1;
      switch (x.type) {
        case CRGPU_BOOL: s.data[slot] = isnull ? 0 : static_cast<uint8_t>(col_get_bool(x, row)); break;
        case CRGPU_INT16: reinterpret_cast<int16_t *>(s.data)[slot] = isnull ? 0 : reinterpret_cast<const int16_t *>(x.data)[row]; break;
        case CRGPU_INT32: reinterpret_cast<int32_t *>(s.data)[slot] = isnull ? 0 : reinterpret_cast<const int32_t *>(x.data)[row]; break;
        case CRGPU_INT64: reinterpret_cast<int64_t *>(s.data)[slot] = isnull ? 0 : reinterpret_cast<const int64_t *>(x.data)[row]; break;
        case CRGPU_FLOAT64: reinterpret_cast<double *>(s.data)[slot] = isnull ? 0.0 : col_get_f64(x, row); break;
        case CRGPU_BYTES: {
          uint32_t a0 = x.offsets[row], len = isnull ? 0 : x.offsets[row + 1] - a0;
          unsigned long long off = len ? atomicAdd(&b_tails[c], static_cast<unsigned long long>(len)) : 0ull;
          s.b_off[slot] = off;
          s.b_len[slot] = len;
          const uint8_t *src = reinterpret_cast<const uint8_t *>(x.data) + a0;
          for (uint32_t t = 0; t < len; t++) s.b_buf[off + t] = src[t];
          break;
        }
        default: break;
      }
    }
    slots[slot] = kSlotOccupied;
  }
}

// ANY_NOT_NULL over a non-group column: the row whose arrival number won the
// atomicMin stores its value (fixed-width: w1 word; BYTES: per-op append buffer).
struct AnyFixArgs {
  const AggOpDev *ops;
  int32_t naggs;
  int32_t _pad;
  int64_t n;
  unsigned long long seq_base;
  const uint32_t *slot_of_row;
  // BYTES values of ANY_ROW ops: indexed by op
  uint64_t *const *any_off;
  uint32_t *const *any_len;
  uint8_t *const *any_buf;
  unsigned long long *any_tails;
};
__global__ __launch_bounds__(256) void agg_any_fixup_kernel(AnyFixArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < a.n; row += stride) {
    const uint32_t slot = a.slot_of_row[row];
    if (slot == kNoSlot) continue;
    for (int f = 0; f < a.naggs; f++) {
      const AggOpDev &op = a.ops[f];
      if (op.kind != ACC_ANY_ROW) continue;
      if (op.w0[slot] != a.seq_base + static_cast<unsigned long long>(row)) continue;
      switch (op.col.type) {
        case CRGPU_BOOL: op.w1[slot] = col_get_bool(op.col, row); break;
        case CRGPU_FLOAT64: op.w1[slot] = static_cast<unsigned long long>(__double_as_longlong(col_get_f64(op.col, row))); break;
        case CRGPU_BYTES: {
          uint32_t a0 = op.col.offsets[row], len = op.col.offsets[row + 1] - a0;
          unsigned long long off = len ? atomicAdd(&a.any_tails[f], static_cast<unsigned long long>(len)) : 0ull;
          a.any_off[f][slot] = off;
          a.any_len[f][slot] = len;
          const uint8_t *src = reinterpret_cast<const uint8_t *>(op.col.data) + a0;
          for (uint32_t t = 0; t < len; t++) a.any_buf[f][off + t] = src[t];
          break;
        }
        default: op.w1[slot] = static_cast<unsigned long long>(col_get_int(op.col, row)); break;
      }
    }
  }
}

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long *p, uint64_t n, unsigned long long v) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// rebuild into a larger table: every OCCUPIED slot moves with its key-store entry and accumulators
struct RehashArgs {
  const uint32_t *old_slots;
  const uint32_t *old_h;
  uint32_t old_cap;
  uint32_t *new_slots;
  uint32_t *new_h;
  uint32_t new_mask;
  uint32_t *move_to;  // [old_cap]: new slot of each old slot (kNoSlot if empty)
};
__global__ __launch_bounds__(256) void agg_rehash_kernel(RehashArgs a) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < a.old_cap; s += stride) {
    uint32_t to = kNoSlot;
    if (a.old_slots[s] == kSlotOccupied) {
      const uint32_t h = a.old_h[s];
      uint32_t idx = h & a.new_mask;
      while (atomicCAS(&a.new_slots[idx], kSlotEmpty, kSlotOccupied) != kSlotEmpty) idx = (idx + 1) & a.new_mask;
      a.new_h[idx] = h;
      to = idx;
    }
    a.move_to[s] = to;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void move_slots_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                         const uint32_t *__restrict__ move_to, uint32_t old_cap) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < old_cap; s += stride) {
    const uint32_t to = move_to[s];
    if (to != kNoSlot) dst[to] = src[s];
  }
}

__global__ __launch_bounds__(256) void occupied_flags_kernel(const uint32_t *__restrict__ slots, uint32_t cap,
                                                             uint8_t *__restrict__ flags) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += stride) flags[s] = slots[s] == kSlotOccupied;
}

// finalisation: one output value per group (Flush, e.g. hash_sum_agg.eg.go:562-572)
struct FinalArgs {
  AggOpDev op;
  int32_t out_type;
  int32_t _pad;
  const int32_t *slot_list;
  int64_t ngroups;
  uint8_t *out_data;
  uint8_t *out_present;  // byte per group
  unsigned int *err;     // set to 1 on SUM_INT out of range
  const uint8_t *key_data;
  const uint8_t *key_present;
};
__global__ __launch_bounds__(256) void agg_finalize_kernel(FinalArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < a.ngroups; g += stride) {
    const uint32_t s = static_cast<uint32_t>(a.slot_list[g]);
    uint8_t present = 1;
    unsigned long long v = 0;
    switch (a.op.kind) {
      case ACC_COUNT_ROWS:
      case ACC_COUNT: v = a.op.w0[s]; break;
      case ACC_SUM_F64: present = a.op.seen[s]; v = a.op.w0[s]; break;
      case ACC_AVG_F64: {
        const unsigned long long c = a.op.w1[s];
        present = c != 0;
        double q = present ? __longlong_as_double(static_cast<long long>(a.op.w0[s])) / static_cast<double>(c) : 0.0;
        v = static_cast<unsigned long long>(__double_as_longlong(q));
        break;
      }
      case ACC_SUM_I128: {
        present = a.op.seen[s];
        const unsigned long long lo = a.op.w0[s], hi = a.op.w1[s];
        const unsigned long long want_hi = (lo >> 63) ? ~0ull : 0ull;
        if (present && hi != want_hi) *a.err = 1;  // exact total does not fit int64
        v = lo;
        break;
      }
      case ACC_MIN_I64:
      case ACC_MAX_I64: present = a.op.seen[s]; v = a.op.w0[s]; break;
      case ACC_MIN_F64:
      case ACC_MAX_F64:
        present = a.op.seen[s];
        v = static_cast<unsigned long long>(__double_as_longlong(ordered_to_f64(a.op.w0[s])));
        break;
      case ACC_BOOL_AND:
      case ACC_BOOL_OR: present = a.op.seen[s]; v = a.op.w0[s] != 0; break;
      case ACC_ANY_ROW: present = a.op.w0[s] != ~0ull; v = a.op.w1[s]; break;
      case ACC_ANY_KEY: {
        present = a.key_present[s];
        switch (a.out_type) {
          case CRGPU_BOOL: v = a.key_data[s]; break;
          case CRGPU_INT16: v = static_cast<unsigned long long>(static_cast<long long>(reinterpret_cast<const int16_t *>(a.key_data)[s])); break;
          case CRGPU_INT32: v = static_cast<unsigned long long>(static_cast<long long>(reinterpret_cast<const int32_t *>(a.key_data)[s])); break;
          default: v = reinterpret_cast<const unsigned long long *>(a.key_data)[s]; break;
        }
        break;
      }
      default: break;
    }
    if (!present) v = 0;
    switch (a.out_type) {
      case CRGPU_BOOL: a.out_data[g] = static_cast<uint8_t>(v != 0); break;
      case CRGPU_INT16: reinterpret_cast<uint16_t *>(a.out_data)[g] = static_cast<uint16_t>(v); break;
      case CRGPU_INT32: reinterpret_cast<uint32_t *>(a.out_data)[g] = static_cast<uint32_t>(v); break;
      default: reinterpret_cast<unsigned long long *>(a.out_data)[g] = v; break;
    }
    a.out_present[g] = present;
  }
}

// BYTES outputs (ANY_NOT_NULL of a BYTES group key or column): build a temporary
// {offsets64, len, buf} view as an Arrow column and reuse gather_bytes.
__global__ __launch_bounds__(256) void bytes_view_kernel(const uint64_t *__restrict__ off, const uint32_t *__restrict__ len,
                                                         const uint8_t *__restrict__ present_in, bool any_row,
                                                         const unsigned long long *__restrict__ w0,
                                                         const int32_t *__restrict__ slot_list, int64_t ngroups,
                                                         uint32_t *__restrict__ lens_out, uint8_t *__restrict__ present_out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < ngroups; g += stride) {
    const uint32_t s = static_cast<uint32_t>(slot_list[g]);
    const uint8_t p = any_row ? (w0[s] != ~0ull) : present_in[s];
    lens_out[g] = p ? len[s] : 0;
    present_out[g] = p;
  }
}
__global__ __launch_bounds__(256) void bytes_copy_slots_kernel(const uint64_t *__restrict__ off, const uint8_t *__restrict__ buf,
                                                               const int32_t *__restrict__ slot_list, int64_t ngroups,
                                                               const uint32_t *__restrict__ dst_off, uint8_t *__restrict__ dst) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < ngroups; g += stride) {
    const uint32_t s = static_cast<uint32_t>(slot_list[g]);
    const uint32_t o = dst_off[g], len = dst_off[g + 1] - o;
    const uint8_t *src = buf + off[s];
    for (uint32_t t = 0; t < len; t++) dst[o + t] = src[t];
  }
}
__global__ __launch_bounds__(256) void narrow_u64_u32_kernel(const uint64_t *__restrict__ in, int64_t n, uint32_t *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = static_cast<uint32_t>(in[i]);
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
struct AggOpHost {
  int func = 0, col = -1, filter_col = -1, kind = 0, key_idx = -1, in_type = 0, out_type = 0;
  DevBuf w0, w1, seen;
  // BYTES value store for ACC_ANY_ROW
  DevBuf any_off, any_len, any_buf;
  uint64_t any_tail = 0;
};

struct crgpu_agg {
  std::vector<int> types, group_cols;
  std::vector<AggOpHost> ops;
  bool need_slot_of_row = false;

  uint32_t cap = 0;
  int64_t ngroups = 0;
  unsigned long long seq = 0;
  DevBuf slots, hstore;
  // key store
  struct KS { DevBuf data, present, b_off, b_len, b_buf; uint64_t tail = 0; };
  std::vector<KS> ks;
  DevBuf store_dev, ops_dev, tails_dev, any_tails_dev, any_ptrs_dev;
  DevBuf deferred, counters, rep_rows, rep_slots, slot_of_row, move_to, flags, slot_list, scratch, err_dev;
  DevBuf lens_tmp, offs64_tmp, present_tmp;
  std::vector<OutCol> out;
  std::vector<crgpu_dcol> views;
};

static unsigned long long acc_init_w0(int kind) {
  switch (kind) {
    case ACC_MIN_I64: return 0x7FFFFFFFFFFFFFFFull;
    case ACC_MAX_I64: return 0x8000000000000000ull;
    case ACC_MIN_F64: return ~0ull;
    case ACC_BOOL_AND: return 1ull;
    case ACC_ANY_ROW: return ~0ull;
    default: return 0ull;
  }
}

static crgpu_status fill_u64(void *p, uint64_t n, unsigned long long v) {
  if (n == 0) return CRGPU_OK;
  if (v == 0) {
    CRGPU_HIP(hipMemsetAsync(p, 0, n * 8, stream()));
    return CRGPU_OK;
  }
  hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(static_cast<int64_t>(n), 256)), dim3(256), 0, stream(),
                     static_cast<unsigned long long *>(p), n, v);
  CRGPU_HIP(hipGetLastError());
  return CRGPU_OK;
}

// (Re)allocate per-slot arrays for capacity `cap` into fresh buffers and initialise them.
struct SlotArrays {
  DevBuf slots, hstore;
  std::vector<DevBuf> w0, w1, seen, any_off, any_len;
  std::vector<DevBuf> k_data, k_present, k_off, k_len;
};

static crgpu_status alloc_slot_arrays(crgpu_agg *a, uint32_t cap, SlotArrays &sa) {
  CRGPU_TRY(sa.slots.ensure(static_cast<size_t>(cap) * 4));
  CRGPU_TRY(sa.hstore.ensure(static_cast<size_t>(cap) * 4));
  CRGPU_HIP(hipMemsetAsync(sa.slots.p, 0xFF, static_cast<size_t>(cap) * 4, stream()));
  const size_t nops = a->ops.size(), nk = a->group_cols.size();
  sa.w0 = std::vector<DevBuf>(nops);
  sa.w1 = std::vector<DevBuf>(nops);
  sa.seen = std::vector<DevBuf>(nops);
  sa.any_off = std::vector<DevBuf>(nops);
  sa.any_len = std::vector<DevBuf>(nops);
  for (size_t f = 0; f < nops; f++) {
    const AggOpHost &op = a->ops[f];
    if (op.kind == ACC_ANY_KEY) continue;
    CRGPU_TRY(sa.w0[f].ensure(static_cast<size_t>(cap) * 8));
    CRGPU_TRY(fill_u64(sa.w0[f].p, cap, acc_init_w0(op.kind)));
    CRGPU_TRY(sa.w1[f].ensure(static_cast<size_t>(cap) * 8));
    CRGPU_HIP(hipMemsetAsync(sa.w1[f].p, 0, static_cast<size_t>(cap) * 8, stream()));
    CRGPU_TRY(sa.seen[f].ensure(cap));
    CRGPU_HIP(hipMemsetAsync(sa.seen[f].p, 0, cap, stream()));
    if (op.kind == ACC_ANY_ROW && op.in_type == CRGPU_BYTES) {
      CRGPU_TRY(sa.any_off[f].ensure(static_cast<size_t>(cap) * 8));
      CRGPU_TRY(sa.any_len[f].ensure(static_cast<size_t>(cap) * 4));
      CRGPU_HIP(hipMemsetAsync(sa.any_off[f].p, 0, static_cast<size_t>(cap) * 8, stream()));
      CRGPU_HIP(hipMemsetAsync(sa.any_len[f].p, 0, static_cast<size_t>(cap) * 4, stream()));
    }
  }
  sa.k_data = std::vector<DevBuf>(nk);
  sa.k_present = std::vector<DevBuf>(nk);
  sa.k_off = std::vector<DevBuf>(nk);
  sa.k_len = std::vector<DevBuf>(nk);
  for (size_t k = 0; k < nk; k++) {
    const int t = a->types[a->group_cols[k]];
    CRGPU_TRY(sa.k_present[k].ensure(cap));
    CRGPU_HIP(hipMemsetAsync(sa.k_present[k].p, 0, cap, stream()));
    if (t == CRGPU_BYTES) {
      CRGPU_TRY(sa.k_off[k].ensure(static_cast<size_t>(cap) * 8));
      CRGPU_TRY(sa.k_len[k].ensure(static_cast<size_t>(cap) * 4));
    } else {
      CRGPU_TRY(sa.k_data[k].ensure(static_cast<size_t>(cap) * type_width(t)));
    }
  }
  return CRGPU_OK;
}

static void swap_buf(DevBuf &a, DevBuf &b) {
  std::swap(a.p, b.p);
  std::swap(a.cap, b.cap);
}

static void install_slot_arrays(crgpu_agg *a, SlotArrays &sa, uint32_t cap) {
  swap_buf(a->slots, sa.slots);
  swap_buf(a->hstore, sa.hstore);
  for (size_t f = 0; f < a->ops.size(); f++) {
    swap_buf(a->ops[f].w0, sa.w0[f]);
    swap_buf(a->ops[f].w1, sa.w1[f]);
    swap_buf(a->ops[f].seen, sa.seen[f]);
    swap_buf(a->ops[f].any_off, sa.any_off[f]);
    swap_buf(a->ops[f].any_len, sa.any_len[f]);
  }
  for (size_t k = 0; k < a->ks.size(); k++) {
    swap_buf(a->ks[k].data, sa.k_data[k]);
    swap_buf(a->ks[k].present, sa.k_present[k]);
    swap_buf(a->ks[k].b_off, sa.k_off[k]);
    swap_buf(a->ks[k].b_len, sa.k_len[k]);
  }
  a->cap = cap;
}

// upload device descriptors (ops + key store) for the current buffers and the given chunk columns
static crgpu_status upload_descriptors(crgpu_agg *a, const std::vector<DCol> *chunk) {
  std::vector<AggOpDev> ops(a->ops.size());
  for (size_t f = 0; f < a->ops.size(); f++) {
    const AggOpHost &h = a->ops[f];
    AggOpDev &d = ops[f];
    std::memset(&d, 0, sizeof(d));
    d.kind = h.kind;
    d.has_filter = h.filter_col >= 0;
    d.key_idx = h.key_idx;
    if (chunk != nullptr) {
      if (h.col >= 0) d.col = (*chunk)[h.col];
      if (h.filter_col >= 0) d.filter = (*chunk)[h.filter_col];
    } else if (h.col >= 0) {
      d.col.type = h.in_type;
    }
    d.w0 = h.w0.as<unsigned long long>();
    d.w1 = h.w1.as<unsigned long long>();
    d.seen = h.seen.as<uint8_t>();
  }
  CRGPU_TRY(a->ops_dev.ensure(sizeof(AggOpDev) * (ops.size() ? ops.size() : 1)));
  if (!ops.empty())
    CRGPU_HIP(hipMemcpyAsync(a->ops_dev.p, ops.data(), sizeof(AggOpDev) * ops.size(), hipMemcpyHostToDevice, stream()));
  KeyStore st;
  std::memset(&st, 0, sizeof(st));
  st.n = static_cast<int>(a->ks.size());
  for (size_t k = 0; k < a->ks.size(); k++) {
    st.c[k].type = a->types[a->group_cols[k]];
    st.c[k].data = a->ks[k].data.as<uint8_t>();
    st.c[k].present = a->ks[k].present.as<uint8_t>();
    st.c[k].b_off = a->ks[k].b_off.as<uint64_t>();
    st.c[k].b_len = a->ks[k].b_len.as<uint32_t>();
    st.c[k].b_buf = a->ks[k].b_buf.as<uint8_t>();
  }
  CRGPU_TRY(a->store_dev.ensure(sizeof(KeyStore)));
  CRGPU_HIP(hipMemcpyAsync(a->store_dev.p, &st, sizeof(KeyStore), hipMemcpyHostToDevice, stream()));
  CRGPU_HIP(hipStreamSynchronize(stream()));  // `ops`/`st` are stack/heap temporaries
  return CRGPU_OK;
}

static crgpu_status agg_grow(crgpu_agg *a, uint32_t new_cap) {
  SlotArrays sa;
  CRGPU_TRY(alloc_slot_arrays(a, new_cap, sa));
  if (a->cap > 0 && a->ngroups > 0) {
    CRGPU_TRY(a->move_to.ensure(static_cast<size_t>(a->cap) * 4));
    RehashArgs r{a->slots.as<uint32_t>(), a->hstore.as<uint32_t>(), a->cap, sa.slots.as<uint32_t>(),
                 sa.hstore.as<uint32_t>(), new_cap - 1, a->move_to.as<uint32_t>()};
    const int grid = grid_for(a->cap, 256);
    hipLaunchKernelGGL(agg_rehash_kernel, dim3(grid), dim3(256), 0, stream(), r);
    const uint32_t *mv = a->move_to.as<uint32_t>();
    auto mv64 = [&](DevBuf &src, DevBuf &dst) {
      if (src.p && dst.p)
        hipLaunchKernelGGL(move_slots_kernel<unsigned long long>, dim3(grid), dim3(256), 0, stream(),
                           src.as<unsigned long long>(), dst.as<unsigned long long>(), mv, a->cap);
    };
    auto mv32 = [&](DevBuf &src, DevBuf &dst) {
      if (src.p && dst.p)
        hipLaunchKernelGGL(move_slots_kernel<uint32_t>, dim3(grid), dim3(256), 0, stream(), src.as<uint32_t>(),
                           dst.as<uint32_t>(), mv, a->cap);
    };
    auto mv16 = [&](DevBuf &src, DevBuf &dst) {
      if (src.p && dst.p)
        hipLaunchKernelGGL(move_slots_kernel<uint16_t>, dim3(grid), dim3(256), 0, stream(), src.as<uint16_t>(),
                           dst.as<uint16_t>(), mv, a->cap);
    };
    auto mv8 = [&](DevBuf &src, DevBuf &dst) {
      if (src.p && dst.p)
        hipLaunchKernelGGL(move_slots_kernel<uint8_t>, dim3(grid), dim3(256), 0, stream(), src.as<uint8_t>(),
                           dst.as<uint8_t>(), mv, a->cap);
    };
    for (size_t f = 0; f < a->ops.size(); f++) {
      mv64(a->ops[f].w0, sa.w0[f]);
      mv64(a->ops[f].w1, sa.w1[f]);
      mv8(a->ops[f].seen, sa.seen[f]);
      mv64(a->ops[f].any_off, sa.any_off[f]);
      mv32(a->ops[f].any_len, sa.any_len[f]);
    }
    for (size_t k = 0; k < a->ks.size(); k++) {
      const int t = a->types[a->group_cols[k]];
      mv8(a->ks[k].present, sa.k_present[k]);
      if (t == CRGPU_BYTES) {
        mv64(a->ks[k].b_off, sa.k_off[k]);
        mv32(a->ks[k].b_len, sa.k_len[k]);
      } else {
        switch (type_width(t)) {
          case 1: mv8(a->ks[k].data, sa.k_data[k]); break;
          case 2: mv16(a->ks[k].data, sa.k_data[k]); break;
          case 4: mv32(a->ks[k].data, sa.k_data[k]); break;
          default: mv64(a->ks[k].data, sa.k_data[k]); break;
        }
      }
    }
    CRGPU_HIP(hipGetLastError());
  }
  CRGPU_HIP(hipStreamSynchronize(stream()));
  install_slot_arrays(a, sa, new_cap);  // old buffers are released when `sa` goes out of scope
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_agg_create(const crgpu_agg_spec *spec, crgpu_agg **out) {
  CRGPU_TRY(ensure_init());
  if (spec == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: null argument");
  if (spec->n_cols < 0 || spec->n_cols > kMaxCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: column count out of range");
  if (spec->n_group < 0 || spec->n_group > kMaxKeyCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: at most 8 grouping columns");
  if (spec->n_aggs < 1 || spec->n_aggs > kMaxAggs) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: between 1 and 64 aggregations");
  std::unique_ptr<crgpu_agg> a(new crgpu_agg());
  for (int i = 0; i < spec->n_cols; i++) {
    if (!valid_type(spec->types[i])) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: unsupported column type");
    a->types.push_back(spec->types[i]);
  }
  for (int i = 0; i < spec->n_group; i++) {
    if (spec->group_cols[i] < 0 || spec->group_cols[i] >= spec->n_cols) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: grouping column ordinal out of range");
    a->group_cols.push_back(spec->group_cols[i]);
  }
  a->ks = std::vector<crgpu_agg::KS>(spec->n_group);
  a->ops = std::vector<AggOpHost>(spec->n_aggs);
  for (int f = 0; f < spec->n_aggs; f++) {
    const crgpu_agg_desc &d = spec->aggs[f];
    AggOpHost &op = a->ops[f];
    op.func = d.func;
    op.col = d.col;
    op.filter_col = d.filter_col;
    if (d.distinct) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: DISTINCT aggregation is planned on the CPU operator");
    if (d.filter_col >= 0 && (d.filter_col >= spec->n_cols || a->types[d.filter_col] != CRGPU_BOOL))
      return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: FILTER column must be a BOOL column");
    if (d.func != CRGPU_COUNT_ROWS && (d.col < 0 || d.col >= spec->n_cols))
      return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: aggregation column ordinal out of range");
    const int t = d.func == CRGPU_COUNT_ROWS ? -1 : a->types[d.col];
    op.in_type = t;
    switch (d.func) {
      case CRGPU_COUNT_ROWS: op.kind = ACC_COUNT_ROWS; op.out_type = CRGPU_INT64; op.col = -1; break;
      case CRGPU_COUNT: op.kind = ACC_COUNT; op.out_type = CRGPU_INT64; break;
      case CRGPU_SUM:
        if (t != CRGPU_FLOAT64) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: SUM over INT yields DECIMAL (apd): CPU operator");
        op.kind = ACC_SUM_F64; op.out_type = CRGPU_FLOAT64; break;
      case CRGPU_AVG:
        if (t != CRGPU_FLOAT64) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: AVG over INT yields DECIMAL (apd): CPU operator");
        op.kind = ACC_AVG_F64; op.out_type = CRGPU_FLOAT64; break;
      case CRGPU_SUM_INT:
        if (!is_int_type(t)) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: SUM_INT needs an integer column");
        op.kind = ACC_SUM_I128; op.out_type = CRGPU_INT64; break;
      case CRGPU_MIN:
      case CRGPU_MAX:
        if (t == CRGPU_BYTES) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: MIN/MAX over BYTES is planned on the CPU operator");
        if (t == CRGPU_FLOAT64) op.kind = d.func == CRGPU_MIN ? ACC_MIN_F64 : ACC_MAX_F64;
        else op.kind = d.func == CRGPU_MIN ? ACC_MIN_I64 : ACC_MAX_I64;
        op.out_type = t; break;
      case CRGPU_BOOL_AND:
      case CRGPU_BOOL_OR:
        if (t != CRGPU_BOOL) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: BOOL_AND/BOOL_OR need a BOOL column");
        op.kind = d.func == CRGPU_BOOL_AND ? ACC_BOOL_AND : ACC_BOOL_OR; op.out_type = CRGPU_BOOL; break;
      case CRGPU_ANY_NOT_NULL: {
        op.out_type = t;
        op.kind = ACC_ANY_ROW;
        if (d.filter_col < 0)
          for (int k = 0; k < spec->n_group; k++)
            if (spec->group_cols[k] == d.col) { op.kind = ACC_ANY_KEY; op.key_idx = k; break; }
        if (op.kind == ACC_ANY_ROW) a->need_slot_of_row = true;
        break;
      }
      default: return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: aggregate function outside the GPU path");
    }
  }
  a->out = std::vector<OutCol>(spec->n_aggs);
  a->views.resize(spec->n_aggs);
  CRGPU_TRY(a->counters.ensure(16));
  CRGPU_TRY(a->err_dev.ensure(4));
  CRGPU_TRY(agg_grow(a.get(), 256));  // hashAggregator starts with 256 buckets too (hash_aggregator.go:217)
  *out = a.release();
  return CRGPU_OK;
}

extern "C" void crgpu_dev_agg_destroy(crgpu_agg *a) {
  if (a == nullptr) return;
  if (stream() != nullptr) (void)hipStreamSynchronize(stream());
  delete a;
}

extern "C" crgpu_status crgpu_dev_agg_reset(crgpu_agg *a) {
  if (a == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_reset: null handle");
  a->ngroups = 0;
  a->seq = 0;
  for (auto &k : a->ks) k.tail = 0;
  for (auto &o : a->ops) o.any_tail = 0;
  const uint32_t cap = a->cap;
  a->cap = 0;
  return agg_grow(a, cap ? cap : 256);
}

static crgpu_status consume_chunk(crgpu_agg *a, const std::vector<DCol> &cols, int64_t n) {
  hipStream_t st = stream();
  // keep the load factor <= 0.5 at chunk start
  if (static_cast<uint64_t>(a->ngroups) * 2 > a->cap) CRGPU_TRY(agg_grow(a, static_cast<uint32_t>(next_pow2(static_cast<uint64_t>(a->ngroups) * 4))));
  CRGPU_TRY(a->deferred.ensure(static_cast<size_t>(n)));
  CRGPU_TRY(a->rep_rows.ensure(static_cast<size_t>(n) * 4));
  CRGPU_TRY(a->rep_slots.ensure(static_cast<size_t>(n) * 4));
  if (a->need_slot_of_row) CRGPU_TRY(a->slot_of_row.ensure(static_cast<size_t>(n) * 4));
  // BYTES key payload: reserve room for every key byte of the chunk
  std::vector<unsigned long long> tails(kMaxKeyCols, 0);
  for (size_t k = 0; k < a->ks.size(); k++) {
    const DCol &c = cols[a->group_cols[k]];
    if (c.type != CRGPU_BYTES) continue;
    uint32_t ends[1] = {0}, begins[1] = {0};
    CRGPU_HIP(hipMemcpyAsync(begins, c.offsets, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipMemcpyAsync(ends, c.offsets + n, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    CRGPU_TRY(a->ks[k].b_buf.ensure(static_cast<size_t>(a->ks[k].tail + (ends[0] - begins[0]) + 16), true));
    tails[k] = a->ks[k].tail;
  }
  std::vector<unsigned long long> any_tails(a->ops.size() ? a->ops.size() : 1, 0);
  for (size_t f = 0; f < a->ops.size(); f++) {
    AggOpHost &op = a->ops[f];
    if (op.kind != ACC_ANY_ROW || op.in_type != CRGPU_BYTES) continue;
    const DCol &c = cols[op.col];
    uint32_t ends[1] = {0}, begins[1] = {0};
    CRGPU_HIP(hipMemcpyAsync(begins, c.offsets, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipMemcpyAsync(ends, c.offsets + n, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    CRGPU_TRY(op.any_buf.ensure(static_cast<size_t>(op.any_tail + (ends[0] - begins[0]) + 16), true));
    any_tails[f] = op.any_tail;
  }
  CRGPU_TRY(a->tails_dev.ensure(kMaxKeyCols * 8));
  CRGPU_HIP(hipMemcpyAsync(a->tails_dev.p, tails.data(), kMaxKeyCols * 8, hipMemcpyHostToDevice, st));
  CRGPU_TRY(a->any_tails_dev.ensure(any_tails.size() * 8));
  CRGPU_HIP(hipMemcpyAsync(a->any_tails_dev.p, any_tails.data(), any_tails.size() * 8, hipMemcpyHostToDevice, st));
  CRGPU_HIP(hipStreamSynchronize(st));

  KeyCols keys;
  keys.n = static_cast<int>(a->group_cols.size());
  keys._pad = 0;
  for (int k = 0; k < keys.n; k++) keys.c[k] = cols[a->group_cols[k]];

  bool only_deferred = false;
  int64_t groups_before = a->ngroups;
  for (int round = 0; round < 40; round++) {
    CRGPU_TRY(upload_descriptors(a, &cols));
    CRGPU_HIP(hipMemsetAsync(a->counters.p, 0, 16, st));
    AggArgs args;
    std::memset(&args, 0, sizeof(args));
    args.keys = keys;
    args.store = a->store_dev.as<KeyStore>();
    args.ops = a->ops_dev.as<AggOpDev>();
    args.naggs = static_cast<int>(a->ops.size());
    args.combine = a->ngroups <= 64 && (a->seq > 0 || only_deferred || n >= 4096) ? 1 : 0;
    if (a->ngroups == 0 && a->seq == 0 && n > 4096) args.combine = 0;  // unknown cardinality: probe with a small first chunk
    args.n = n;
    args.seq_base = a->seq;
    args.slots = a->slots.as<uint32_t>();
    args.mask = a->cap - 1;
    args.only_deferred = only_deferred ? 1 : 0;
    args.hstore = a->hstore.as<uint32_t>();
    args.deferred = a->deferred.as<uint8_t>();
    args.counters = a->counters.as<unsigned int>();
    args.rep_rows = a->rep_rows.as<uint32_t>();
    args.rep_slots = a->rep_slots.as<uint32_t>();
    args.slot_of_row = a->need_slot_of_row ? a->slot_of_row.as<uint32_t>() : nullptr;
    {
      ProfScope prof("agg");
      hipLaunchKernelGGL(agg_consume_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, args);
      CRGPU_HIP(hipGetLastError());
    }
    unsigned int counters[4] = {0, 0, 0, 0};
    CRGPU_HIP(hipMemcpyAsync(counters, a->counters.p, 16, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    const unsigned int nrep = counters[0], ndef = counters[1];
    if (nrep > 0) {
      hipLaunchKernelGGL(agg_fixup_kernel, dim3(grid_for(nrep, 256)), dim3(256), 0, st, keys,
                         a->store_dev.as<KeyStore>(), nrep, a->rep_rows.as<uint32_t>(), a->rep_slots.as<uint32_t>(),
                         a->slots.as<uint32_t>(), a->tails_dev.as<unsigned long long>());
      CRGPU_HIP(hipGetLastError());
      a->ngroups += nrep;
    }
    if (a->need_slot_of_row) {
      // ANY_NOT_NULL over non-group columns: the rows that won the arrival-order atomicMin in THIS
      // round store their values now (slot ids are only valid until the table is rebuilt)
      const size_t nops = a->ops.size();
      std::vector<void *> ptrs(nops * 3, nullptr);
      for (size_t f = 0; f < nops; f++) {
        ptrs[f] = a->ops[f].any_off.p;
        ptrs[nops + f] = a->ops[f].any_len.p;
        ptrs[2 * nops + f] = a->ops[f].any_buf.p;
      }
      CRGPU_TRY(a->any_ptrs_dev.ensure(ptrs.size() * 8));
      CRGPU_HIP(hipMemcpyAsync(a->any_ptrs_dev.p, ptrs.data(), ptrs.size() * 8, hipMemcpyHostToDevice, st));
      AnyFixArgs fa;
      std::memset(&fa, 0, sizeof(fa));
      fa.ops = a->ops_dev.as<AggOpDev>();
      fa.naggs = static_cast<int>(nops);
      fa.n = n;
      fa.seq_base = a->seq;
      fa.slot_of_row = a->slot_of_row.as<uint32_t>();
      fa.any_off = reinterpret_cast<uint64_t *const *>(a->any_ptrs_dev.as<void *>());
      fa.any_len = reinterpret_cast<uint32_t *const *>(a->any_ptrs_dev.as<void *>() + nops);
      fa.any_buf = reinterpret_cast<uint8_t *const *>(a->any_ptrs_dev.as<void *>() + 2 * nops);
      fa.any_tails = a->any_tails_dev.as<unsigned long long>();
      hipLaunchKernelGGL(agg_any_fixup_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, fa);
      CRGPU_HIP(hipGetLastError());
      CRGPU_HIP(hipStreamSynchronize(st));  // `ptrs` is a host temporary
    }
    if (ndef == 0) break;
    // table too full for some rows: grow for the worst case and replay the deferred rows
    CRGPU_HIP(hipStreamSynchronize(st));
    uint64_t want = next_pow2((static_cast<uint64_t>(a->ngroups) + ndef) * 2);
    if (want <= a->cap) want = static_cast<uint64_t>(a->cap) * 2;
    if (want > 0x80000000ull) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_consume: more than 2^30 groups per GPU");
    CRGPU_TRY(agg_grow(a, static_cast<uint32_t>(want)));
    only_deferred = true;
    if (round == 39) return fail(CRGPU_ERR_HIP, "crgpu_dev_agg_consume: deferred rows did not converge");
  }
  (void)groups_before;
  // read back the BYTES tails
  if (!a->ks.empty()) {
    CRGPU_HIP(hipMemcpyAsync(tails.data(), a->tails_dev.p, kMaxKeyCols * 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    for (size_t k = 0; k < a->ks.size(); k++) a->ks[k].tail = tails[k];
  }
  if (a->need_slot_of_row) {
    CRGPU_HIP(hipMemcpyAsync(any_tails.data(), a->any_tails_dev.p, any_tails.size() * 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    for (size_t f = 0; f < a->ops.size(); f++)
      if (a->ops[f].kind == ACC_ANY_ROW && a->ops[f].in_type == CRGPU_BYTES) a->ops[f].any_tail = any_tails[f];
  }
  CRGPU_HIP(hipStreamSynchronize(st));
  a->seq += static_cast<unsigned long long>(n);
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_agg_consume(crgpu_agg *a, const crgpu_dcol *cols, int64_t n) {
  if (a == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: null handle");
  if (n < 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: negative row count");
  if (n == 0) return CRGPU_OK;
  if (cols == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: null columns");
  const int nc = static_cast<int>(a->types.size());
  std::vector<DCol> in;
  for (int c = 0; c < nc; c++) {
    if (cols[c].type != a->types[c]) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: column type differs from the spec");
    if (cols[c].type == CRGPU_BYTES && cols[c].offsets == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: BYTES column without offsets");
    if (cols[c].type != CRGPU_BYTES && cols[c].data == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: null data pointer");
    in.push_back(to_dcol(cols[c]));
  }
  // chunking: a small first chunk discovers the cardinality regime, then 16M-row chunks
  const int64_t kChunk = int64_t(1) << 24;
  int64_t pos = 0;
  while (pos < n) {
    int64_t len = n - pos < kChunk ? n - pos : kChunk;
    if (a->seq == 0 && len > 65536) len = 65536;
    std::vector<DCol> chunk = in;
    for (int c = 0; c < nc; c++) {
      DCol &d = chunk[c];
      if (d.valid != nullptr && (pos & 7) != 0)
        return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: internal chunk offset not byte aligned");
      if (d.type == CRGPU_BYTES) d.offsets += pos;
      else d.data = static_cast<const uint8_t *>(d.data) + pos * type_width(d.type);
      if (d.valid != nullptr) d.valid += pos >> 3;
    }
    CRGPU_TRY(consume_chunk(a, chunk, len));
    pos += len;
  }
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_agg_finish(crgpu_agg *a, crgpu_dev_result *res) {
  if (a == nullptr || res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_finish: null argument");
  hipStream_t st = stream();
  int64_t ng = 0;
  CRGPU_TRY(a->flags.ensure(a->cap));
  CRGPU_TRY(a->slot_list.ensure(static_cast<size_t>(a->cap) * 4));
  hipLaunchKernelGGL(occupied_flags_kernel, dim3(grid_for(a->cap, 256)), dim3(256), 0, st, a->slots.as<uint32_t>(),
                     a->cap, a->flags.as<uint8_t>());
  CRGPU_HIP(hipGetLastError());
  CRGPU_TRY(select_flagged(a->flags.as<uint8_t>(), a->cap, a->slot_list.as<int32_t>(), a->scratch, &ng));
  if (ng != a->ngroups) return fail(CRGPU_ERR_HIP, "crgpu_dev_agg_finish: group count mismatch (internal error)");
  CRGPU_TRY(upload_descriptors(a, nullptr));
  CRGPU_HIP(hipMemsetAsync(a->err_dev.p, 0, 4, st));
  CRGPU_TRY(a->present_tmp.ensure(static_cast<size_t>(ng > 0 ? ng : 1)));
  std::vector<AggOpDev> ops(a->ops.size());
  CRGPU_HIP(hipMemcpyAsync(ops.data(), a->ops_dev.p, sizeof(AggOpDev) * ops.size(), hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  for (size_t f = 0; f < a->ops.size(); f++) {
    AggOpHost &op = a->ops[f];
    OutCol &o = a->out[f];
    o.type = op.out_type;
    o.nullable = true;
    CRGPU_TRY(o.valid.ensure(static_cast<size_t>((ng + 7) / 8 + 1)));
    if (op.out_type == CRGPU_BYTES) {
      // ANY_NOT_NULL over a BYTES group key or BYTES column
      const bool any_row = op.kind == ACC_ANY_ROW;
      const uint64_t *off = any_row ? op.any_off.as<uint64_t>() : a->ks[op.key_idx].b_off.as<uint64_t>();
      const uint32_t *len = any_row ? op.any_len.as<uint32_t>() : a->ks[op.key_idx].b_len.as<uint32_t>();
      const uint8_t *buf = any_row ? op.any_buf.as<uint8_t>() : a->ks[op.key_idx].b_buf.as<uint8_t>();
      const uint8_t *pres = any_row ? nullptr : a->ks[op.key_idx].present.as<uint8_t>();
      CRGPU_TRY(a->lens_tmp.ensure(static_cast<size_t>(ng > 0 ? ng : 1) * 4));
      CRGPU_TRY(a->offs64_tmp.ensure(static_cast<size_t>(ng + 1) * 8));
      CRGPU_TRY(o.offsets.ensure(static_cast<size_t>(ng + 1) * 4));
      uint64_t total = 0;
      if (ng > 0) {
        hipLaunchKernelGGL(bytes_view_kernel, dim3(grid_for(ng, 256)), dim3(256), 0, st, off, len, pres, any_row,
                           op.w0.as<unsigned long long>(), a->slot_list.as<int32_t>(), ng, a->lens_tmp.as<uint32_t>(),
                           a->present_tmp.as<uint8_t>());
        CRGPU_HIP(hipGetLastError());
      }
      CRGPU_TRY(exclusive_scan_u32(a->lens_tmp.as<uint32_t>(), ng, a->offs64_tmp.as<uint64_t>(), a->scratch, &total));
      if (total > 0xFFFFFFFFull) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_finish: BYTES output exceeds 4 GiB");
      hipLaunchKernelGGL(narrow_u64_u32_kernel, dim3(grid_for(ng + 1, 256)), dim3(256), 0, st, a->offs64_tmp.as<uint64_t>(),
                         ng + 1, o.offsets.as<uint32_t>());
      CRGPU_TRY(o.data.ensure(static_cast<size_t>(total ? total : 1)));
      if (ng > 0) {
        hipLaunchKernelGGL(bytes_copy_slots_kernel, dim3(grid_for(ng, 256)), dim3(256), 0, st, off, buf,
                           a->slot_list.as<int32_t>(), ng, o.offsets.as<uint32_t>(), o.data.as<uint8_t>());
        CRGPU_HIP(hipGetLastError());
      }
    } else {
      CRGPU_TRY(o.data.ensure(static_cast<size_t>(ng > 0 ? ng : 1) * type_width(op.out_type)));
      if (ng > 0) {
        FinalArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.op = ops[f];
        fa.out_type = op.out_type;
        fa.slot_list = a->slot_list.as<int32_t>();
        fa.ngroups = ng;
        fa.out_data = o.data.as<uint8_t>();
        fa.out_present = a->present_tmp.as<uint8_t>();
        fa.err = a->err_dev.as<unsigned int>();
        if (op.kind == ACC_ANY_KEY) {
          fa.key_data = a->ks[op.key_idx].data.as<uint8_t>();
          fa.key_present = a->ks[op.key_idx].present.as<uint8_t>();
        }
        hipLaunchKernelGGL(agg_finalize_kernel, dim3(grid_for(ng, 256)), dim3(256), 0, st, fa);
        CRGPU_HIP(hipGetLastError());
      }
    }
    CRGPU_TRY(pack_bitmap(a->present_tmp.as<uint8_t>(), ng, o.valid.as<uint8_t>()));
  }
  unsigned int err = 0;
  CRGPU_HIP(hipMemcpyAsync(&err, a->err_dev.p, 4, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  if (err) return fail(CRGPU_ERR_INT_OUT_OF_RANGE, "integer out of range");  // tree.ErrIntOutOfRange
  for (size_t f = 0; f < a->out.size(); f++) a->views[f] = a->out[f].view();
  res->n = ng;
  res->ncols = static_cast<int32_t>(a->out.size());
  res->_pad = 0;
  res->cols = a->views.data();
  return CRGPU_OK;
}
