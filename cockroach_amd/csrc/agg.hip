// agg.hip -- device-resident online hash aggregation.
//
// Replaces colexec.hashAggregator.onlineAgg (pkg/sql/colexec/hash_aggregator.go:354-453)
// and the per-(group, function) accumulator objects of colexecagg
// (pkg/sql/colexec/colexecagg/hash_{sum,sum_int,avg,count,min_max,any_not_null,bool_and_or}_agg.eg.go).
//
// Design (not the reference's 4096-tuple buffer + equality chains + virtual Compute calls):
//  * Groups are numbered densely in creation order (group id, "gid").  Everything that
//    belongs to a group -- its key (key store), its accumulator words, its hash word --
//    lives in arrays indexed by gid; they only ever grow at the end, so growing the hash
//    index never moves them and the result columns are read straight out of them.
//  * The hash INDEX is an open-addressing array of 64-bit entries
//    {tag:23 | null mask:8 | gid:32}.  The tag filters nearly all non-matching keys before
//    the key store is touched; the null mask makes NULL == NULL (GROUP BY semantics,
//    hash_aggregator.go:269 allowNullEquality=true) a register compare.  While a chunk is
//    resolved an entry may instead hold {claim bit | tag | mask | chunk position} of the
//    row that claimed it: input columns are immutable during the kernel, so other rows
//    compare against the claimant's ROW and no publication protocol is needed.
//  * Measured on MI355X (experiments/atomics.hip): random 8-byte HBM atomics run at
//    ~24 G/s whatever the table size or scope, while LDS atomics sustain > 1 T rows/s.
//    So accumulation is LDS-privatised whenever the groups fit: each workgroup owns a
//    partial table in LDS ([word][group][replica]; replicas spread same-group lanes over
//    banks), accumulates with ds_add_f64 / ds_add_u64 / ds_min / ds_max, and merges it into
//    the HBM accumulators with one atomic per (workgroup, group, word) at the end.  When the
//    groups exceed one LDS table the kernel sweeps gid ranges (re-reading the 4-byte gid per
//    row) as long as that is cheaper than HBM atomics; beyond that it falls back to them.
//  * Steady state is ONE fused pass (hash -> index lookup -> accumulate); only rows whose
//    group does not exist yet go through find-or-insert (resolve -> fixup -> accumulate).
//  * SUM_INT accumulates in 128 bits (two words) so the overflow test does not
//    depend on summation order: CRGPU_ERR_INT_OUT_OF_RANGE iff the exact group total
//    leaves int64 (the reference errors on transient overflow in arrival order,
//    hash_sum_int_agg.eg.go:295-298; see DESIGN.md "Divergences").
//  * float SUM/AVG use fp64 atomics: order-dependent in the last bits (stated
//    tolerance rel 1e-9).
//
// Roofline: HBM-bound (streaming input; LDS atomics are far from their limit).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>

#include "device.h"
#include "radix.h"

namespace crgpu {

static constexpr int kMaxAggs = 64;
static constexpr int kMaxProbe = 128;                  // index slots inspected before a row is deferred / reported as a miss
static constexpr unsigned long long kMetaEmpty = ~0ull;
static constexpr uint32_t kClaimBit = 0x80000000u;     // upper word of an index entry: claimed during the current launch
static constexpr uint32_t kGidDefer = 0xFFFFFFFFu;     // out_gid: row could not be placed in this round
static constexpr uint32_t kGidSlotRef = 0x80000000u;   // out_gid: low 31 bits name an index slot whose gid the fixup assigns
static constexpr uint32_t kMaxCap = 1u << 30;          // index entries (=> at most 2^29 groups per GPU)
static constexpr uint32_t kIdxHashMask = kMaxCap - 1;
static constexpr size_t kLdsReplicated = 20 * 1024;    // replicate small partial tables up to this size (8 workgroups per CU)
static constexpr size_t kLdsLarge = 152 * 1024;        // largest partial table: one 1024-thread workgroup per CU
static constexpr size_t kLdsMax = 160 * 1024;
static constexpr int kMaxReplicas = 32;
#ifndef CRGPU_OP_GROUP
#define CRGPU_OP_GROUP 2
#endif
#ifndef CRGPU_ACC_ROWS
#define CRGPU_ACC_ROWS 4
#endif
static constexpr int kOpGroup = CRGPU_OP_GROUP;                     // aggregations whose input loads are in flight together
static constexpr int kAccRows = CRGPU_ACC_ROWS;                     // rows per thread and iteration in the accumulate kernels

enum AccKind : int32_t {
  ACC_COUNT_ROWS = 0, ACC_COUNT, ACC_SUM_F64, ACC_AVG_F64, ACC_SUM_I128, ACC_MIN_I64, ACC_MAX_I64, ACC_MIN_F64,
  ACC_MAX_F64, ACC_BOOL_AND, ACC_BOOL_OR, ACC_ANY_KEY, ACC_ANY_ROW,
  ACC_MIN_BYTES, ACC_MAX_BYTES  // w0: best row of the current pass (~0: none), w1: 8-byte prefix of the best value, value in the any store
};
__host__ __device__ __forceinline__ bool is_minmax_bytes(int kind) { return kind == ACC_MIN_BYTES || kind == ACC_MAX_BYTES; }

struct AggOpDev {
  int32_t kind;
  int32_t has_filter;        // 0 none, 1 `filter` is a BOOL column, 2 `filter` is an INT64 column compared with f_cst (f_cmp)
  int32_t key_idx;           // ACC_ANY_KEY: which group column
  int32_t lw0, lw1, lseen;   // LDS partial table: word / seen-byte plane numbers (-1: none)
  DCol col;
  DCol filter;
  // computed input (crgpu_dev_agg_create_fused): value = e_op2(e_op1(col, col2), col3); e_op1 < 0: the plain column
  int32_t e_op1, e_op2;
  int32_t f_cmp, _pad;
  long long f_cst;
  DCol col2, col3;
  unsigned long long *w0;    // [gid]
  unsigned long long *w1;    // [gid] (AVG count, SUM_INT high word, ANY_NOT_NULL value)
  uint8_t *seen;             // [gid] a non-NULL input was accumulated
};

struct KeyStoreCol {
  int32_t type;
  int32_t _pad;
  uint8_t *data;       // fixed-width: value per gid
  uint8_t *present;    // one byte per gid (0 => NULL key); only read when results are produced
  uint64_t *b_off;     // BYTES: offset into b_buf per gid
  uint32_t *b_len;     // BYTES: length per gid
  uint8_t *b_buf;      // BYTES: append-only payload buffer
};
struct KeyStore {
  int32_t n;
  int32_t _pad;
  KeyStoreCol c[kMaxKeyCols];
};

// Accumulate-kernel form of an aggregation: lives in the KERNEL ARGUMENTS (constant memory), so the
// per-op fields are scalar loads that no store or atomic of the kernel can invalidate.  (As loads from
// a device array they were re-fetched per lane after every atomic: 68 vector loads per row.)
struct OpK {
  int8_t kind, in_type, has_filter, vex;  // has_filter: 0 none, 1 BOOL column, 2 selection fsel - 1; vex: computed input vex - 1 (0: plain column)
  int8_t lw0, lw1, lseen, fsel;  // LDS partial table: word / seen-byte plane numbers (-1: none)
  const void *data;             // input column
  const uint8_t *valid;
  const uint8_t *fdata;         // FILTER column (BOOL)
  const uint8_t *fvalid;
  unsigned long long *w0;       // HBM accumulators [gid]
  unsigned long long *w1;
  uint8_t *seen;
};
static constexpr int kMaxOpsPerLaunch = 32;  // 2 KiB of the 4 KiB kernel-argument budget

// Computed inputs of the fused scan -> selection -> projection -> aggregation form (crgpu_dev_agg_create_fused; the
// reference plans colexecsel / colexecproj operators in front of the aggregator, tpch_vec:20537-20546): evaluated in
// the accumulate kernels' row loop, so neither the selection's BOOL column nor the projected FLOAT64 columns ever exist
// in HBM.  value = op2(op1(x, y), z) with the roundings of two (three) separate projection operators; a NULL operand
// makes the value NULL (proj_non_const_ops.eg.go:12253).  The tables travel in the kernel arguments like the ops.
enum { EX_PLUS = 0, EX_MINUS, EX_MULT, EX_MULT_ONE_MINUS, EX_MULT_ONE_PLUS };  // = crgpu_dev_project_f64's op codes
enum { SEL_EQ = 0, SEL_NE, SEL_LT, SEL_LE, SEL_GT, SEL_GE };                      // = crgpu_dev_select_i64_const's
struct VExpr {
  const double *x, *y, *z;
  const uint8_t *vx, *vy, *vz;
  int8_t op1, op2;  // op2 < 0: one operator only
  int8_t _pad[6];
};
struct VSel {
  const long long *data;
  const uint8_t *valid;
  long long cst;
  int32_t cmp, _pad;
};
static constexpr int kMaxVex = 8, kMaxVsel = 4;
__device__ __forceinline__ double ex_apply(int op, double a, double b) {
  switch (op) {
    case EX_PLUS: return __dadd_rn(a, b);
    case EX_MINUS: return __dsub_rn(a, b);
    case EX_MULT: return __dmul_rn(a, b);
    case EX_MULT_ONE_MINUS: return __dmul_rn(a, __dsub_rn(1.0, b));
    default: return __dmul_rn(a, __dadd_rn(1.0, b));
  }
}
__device__ __forceinline__ bool sel_apply(int cmp, long long v, long long c) {
  switch (cmp) {
    case SEL_EQ: return v == c;
    case SEL_NE: return v != c;
    case SEL_LT: return v < c;
    case SEL_LE: return v <= c;
    case SEL_GT: return v > c;
    default: return v >= c;
  }
}

// total-order map for float MIN/MAX: NaN is the smallest value
// (hash_min_max_agg.eg.go:966-981) -> 0; everything else order-preserving and > 0.
// -0 ranks below +0 here; the reference's comparison treats them as equal and keeps the one it saw first.  Which zero
// comes out therefore differs by sign only (documented in DESIGN.md "divergences"; the oracle compares with ==).
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

// ------------------------------------------------------------------ row hash (internal: only results must match the reference)
struct RowHash {
  uint32_t idx;    // 30 bits: index position before masking
  uint32_t upper;  // tag << 8 | null mask (claim bit clear)
};
__device__ __forceinline__ uint64_t hash_step(uint64_t h, uint64_t v) {
  return (((h << 27) | (h >> 37)) ^ v) * 0x9FB21C651E98DF25ull;
}
__device__ inline RowHash agg_hash(const KeyCols &k, int64_t row) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  uint32_t nm = 0;
  for (int c = 0; c < k.n; c++) {
    const DCol &x = k.c[c];
    if (col_is_null(x, row)) {
      nm |= 1u << c;
      h = hash_step(h, 0x6E756C6Cull);
      continue;
    }
    if (x.type == CRGPU_BYTES) {
      const uint32_t a0 = x.offsets[row], len = x.offsets[row + 1] - a0;
      const uint8_t *p = reinterpret_cast<const uint8_t *>(x.data) + a0;
      uint32_t t = 0;
      for (; t + 8 <= len; t += 8) {  // unaligned 8-byte loads (little endian: same word as the byte loop below builds)
        uint64_t w;
        __builtin_memcpy(&w, p + t, 8);
        h = hash_step(h, w);
      }
      if (t < len) {
        uint64_t w = 0;
        for (uint32_t j = 0; t + j < len; j++) w |= static_cast<uint64_t>(p[t + j]) << (8 * j);
        h = hash_step(h, w);
      }
      h = hash_step(h, 0xB17E5ull + len);
    } else {
      h = hash_step(h, pack_key(x, row));  // ints sign-extended, floats canonical (NaN == NaN, +0 == -0)
    }
  }
  h = fmix64(h);
  RowHash r;
  r.idx = static_cast<uint32_t>(h) & kIdxHashMask;
  r.upper = (static_cast<uint32_t>(h >> 30) & 0x7FFFFFu) << 8 | nm;
  return r;
}

// chunk row vs stored key of group `gid`; `nm` is the (equal) null mask of both sides
__device__ inline bool key_equals_store(const KeyCols &k, int64_t row, const KeyStore &st, uint32_t gid, uint32_t nm) {
  for (int c = 0; c < k.n; c++) {
    if ((nm >> c) & 1u) continue;
    const DCol &x = k.c[c];
    const KeyStoreCol &s = st.c[c];
    switch (x.type) {
      case CRGPU_BOOL:
        if (col_get_bool(x, row) != (s.data[gid] != 0)) return false;
        break;
      case CRGPU_INT16:
        if (reinterpret_cast<const int16_t *>(x.data)[row] != reinterpret_cast<const int16_t *>(s.data)[gid]) return false;
        break;
      case CRGPU_INT32:
        if (reinterpret_cast<const int32_t *>(x.data)[row] != reinterpret_cast<const int32_t *>(s.data)[gid]) return false;
        break;
      case CRGPU_INT64:
        if (reinterpret_cast<const int64_t *>(x.data)[row] != reinterpret_cast<const int64_t *>(s.data)[gid]) return false;
        break;
      case CRGPU_FLOAT64: {
        double p = col_get_f64(x, row), q = reinterpret_cast<const double *>(s.data)[gid];
        if (!(p == q || (p != p && q != q))) return false;
        break;
      }
      case CRGPU_BYTES: {
        uint32_t xa = x.offsets[row], xl = x.offsets[row + 1] - xa;
        if (xl != s.b_len[gid]) return false;
        const uint8_t *px = reinterpret_cast<const uint8_t *>(x.data) + xa;
        const uint8_t *ps = s.b_buf + s.b_off[gid];
        for (uint32_t t = 0; t < xl; t++)
          if (px[t] != ps[t]) return false;
        break;
      }
      default: return false;
    }
  }
  return true;
}

// ------------------------------------------------------------------ packed keys (LDS key table of the fused pass)
// When the group columns fit 128 bits, a row's key packs injectively into two words:
// fixed-width columns by value (floats canonical), BYTES as {length:4, up to 7 bytes}, plus one NULL
// bit per column when there is room.  Rows that do not fit (longer BYTES, or a NULL without a NULL
// bit) are "unpackable": they and their groups use the HBM index instead.
struct PackCol {
  int8_t word, shift;            // where the value field starts
  int8_t null_word, null_shift;  // where the NULL bit lives (null_word < 0: none)
  int8_t bbytes;                 // BYTES: longest packable value
  int8_t _pad[3];
};
struct PackSpec {
  int32_t enabled;
  int32_t two_words;
  PackCol c[kMaxKeyCols];
};
// packed key of the stored key of group gid (false: the group's key is unpackable)
__device__ inline bool pack_group(const KeyStore &st, const PackSpec &ps, uint32_t gid, unsigned long long &k0,
                                  unsigned long long &k1) {
  k0 = 0;
  k1 = 0;
  for (int c = 0; c < st.n; c++) {
    const KeyStoreCol &s = st.c[c];
    const PackCol &pc = ps.c[c];
    unsigned long long v;
    int word = pc.word, shift = pc.shift;
    if (s.present[gid] == 0) {
      if (pc.null_word < 0) return false;
      v = 1;
      word = pc.null_word;
      shift = pc.null_shift;
    } else {
      switch (s.type) {
        case CRGPU_BYTES: {
          const uint32_t len = s.b_len[gid];
          if (len > static_cast<uint32_t>(pc.bbytes)) return false;
          const uint8_t *p = s.b_buf + s.b_off[gid];
          v = len;
          for (uint32_t j = 0; j < len; j++) v |= static_cast<unsigned long long>(p[j]) << (4 + 8 * j);
          break;
        }
        case CRGPU_BOOL: v = s.data[gid] != 0; break;
        case CRGPU_INT16: v = reinterpret_cast<const uint16_t *>(s.data)[gid]; break;
        case CRGPU_INT32: v = reinterpret_cast<const uint32_t *>(s.data)[gid]; break;
        case CRGPU_FLOAT64: {
          const double f = reinterpret_cast<const double *>(s.data)[gid];
          v = f != f ? 0x7FF8000000000000ull : (f == 0 ? 0ull : static_cast<unsigned long long>(__double_as_longlong(f)));
          break;
        }
        default: v = reinterpret_cast<const unsigned long long *>(s.data)[gid]; break;
      }
    }
    if (word) k1 |= v << shift;
    else k0 |= v << shift;
  }
  return true;
}

// pack_row for the kAccRows rows of a thread, column-wise (see load_inputs): fixed-width columns load
// branch-free; BYTES columns keep a per-row branch for their variable length.
template <int K>
__device__ __forceinline__ void pack_rows(const KeyCols &keys, const PackSpec &ps, const int64_t (&row)[K], bool (&ok)[K],
                                          unsigned long long (&k0)[K], unsigned long long (&k1)[K]) {
#pragma unroll
  for (int k = 0; k < K; k++) {
    k0[k] = 0;
    k1[k] = 0;
  }
  for (int c = 0; c < keys.n; c++) {
    const DCol &x = keys.c[c];
    const PackCol &pc = ps.c[c];
    bool isnull[K];
#pragma unroll
    for (int k = 0; k < K; k++) isnull[k] = false;
    if (x.valid != nullptr) {
      uint8_t vb[K];
#pragma unroll
      for (int k = 0; k < K; k++) vb[k] = x.valid[row[k] >> 3];
#pragma unroll
      for (int k = 0; k < K; k++) isnull[k] = !((vb[k] >> (row[k] & 7)) & 1);
    }
    unsigned long long v[K];
    if (x.type == CRGPU_BYTES) {
      uint32_t a0[K], len[K];
#pragma unroll
      for (int k = 0; k < K; k++) {
        a0[k] = x.offsets[row[k]];
        len[k] = x.offsets[row[k] + 1];
      }
      uint32_t mx = 0;
#pragma unroll
      for (int k = 0; k < K; k++) {
        len[k] -= a0[k];
        if (len[k] > static_cast<uint32_t>(pc.bbytes)) {
          if (!isnull[k]) ok[k] = false;
          len[k] = 0;
        }
        if (isnull[k] || !ok[k]) len[k] = 0;
        v[k] = len[k];
        mx = len[k] > mx ? len[k] : mx;
      }
      // longest packable value in the wave (wave-uniform loop bound); byte j of every row is one unconditional
      // load: rows shorter than j + 1 read a byte that is certainly mapped (their own last byte, or the offsets array)
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(mx, d, 64);
        mx = o > mx ? o : mx;
      }
      for (uint32_t j = 0; j < mx; j++) {
        uint8_t b[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
          const uint8_t *src = len[k] ? reinterpret_cast<const uint8_t *>(x.data) + a0[k] + (j < len[k] ? j : len[k] - 1)
                                      : reinterpret_cast<const uint8_t *>(x.offsets);
          b[k] = *src;
        }
#pragma unroll
        for (int k = 0; k < K; k++)
          if (j < len[k]) v[k] |= static_cast<unsigned long long>(b[k]) << (4 + 8 * j);
      }
    } else {
      switch (x.type) {
        case CRGPU_BOOL:
#pragma unroll
          for (int k = 0; k < K; k++) v[k] = reinterpret_cast<const uint8_t *>(x.data)[row[k]] != 0;
          break;
        case CRGPU_INT16:
#pragma unroll
          for (int k = 0; k < K; k++) v[k] = reinterpret_cast<const uint16_t *>(x.data)[row[k]];
          break;
        case CRGPU_INT32:
#pragma unroll
          for (int k = 0; k < K; k++) v[k] = reinterpret_cast<const uint32_t *>(x.data)[row[k]];
          break;
        case CRGPU_FLOAT64:
#pragma unroll
          for (int k = 0; k < K; k++) {
            const double f = reinterpret_cast<const double *>(x.data)[row[k]];
            v[k] = f != f ? 0x7FF8000000000000ull : (f == 0 ? 0ull : static_cast<unsigned long long>(__double_as_longlong(f)));
          }
          break;
        default:
#pragma unroll
          for (int k = 0; k < K; k++) v[k] = reinterpret_cast<const unsigned long long *>(x.data)[row[k]];
          break;
      }
    }
    if (x.valid == nullptr) {  // no NULLs in this column: no per-row branch at all
      if (pc.word) {
#pragma unroll
        for (int k = 0; k < K; k++) k1[k] |= v[k] << pc.shift;
      } else {
#pragma unroll
        for (int k = 0; k < K; k++) k0[k] |= v[k] << pc.shift;
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      unsigned long long f0 = 0, f1 = 0;
      if (isnull[k]) {
        if (pc.null_word < 0) ok[k] = false;
        else if (pc.null_word) f1 = 1ull << pc.null_shift;
        else f0 = 1ull << pc.null_shift;
      } else if (pc.word) {
        f1 = v[k] << pc.shift;
      } else {
        f0 = v[k] << pc.shift;
      }
      k0[k] |= f0;
      k1[k] |= f1;
    }
  }
}
// A load at a 32-bit byte offset from a wave-uniform base: the scalar-base + 32-bit-vector-offset form of global_load,
// one VGPR per address and no 64-bit address arithmetic (the offset of row r is shared by every column of one width).
template <typename T>
__device__ __forceinline__ T ld_at(const void *base, uint32_t byte_off) {
  return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ T ld_at_nt(const void *base, uint32_t byte_off) {  // streamed once: do not keep the line
  return __builtin_nontemporal_load(reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off));
}
// Bounds-checked buffer loads: a column that is absent (a slot without an operand, a FILTER without a validity bitmap)
// gets a descriptor of zero records -- its loads return 0 and touch no memory -- so the load sequence of a kernel is
// the same straight-line code whatever the plan is.  (A load behind a wave-uniform branch would do as well for the
// memory system, but its result merges with the other path's value at the join, and the compiler waits for the data
// right there: every such branch ends the run of loads in flight.)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p != nullptr ? bytes : 0u, 0x00020000);
}
__device__ __forceinline__ unsigned long long buf_ld8(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, static_cast<int>(byte_off), 0, 0);
  return (static_cast<unsigned long long>(v.y) << 32) | v.x;
}
__device__ __forceinline__ uint32_t buf_ld1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b8(r, static_cast<int>(byte_off), 0, 0);
}
__device__ __forceinline__ uint32_t pk_hash(unsigned long long k0, unsigned long long k1) {
#ifdef CRGPU_PK_HASH_FMIX
  return static_cast<uint32_t>(fmix64(k0 ^ (k1 * 0x9FB21C651E98DF25ull)) >> 20);
#endif
  // 32-bit mixing with two multiplies ("lowbias32"): the table is at most half full and only has to spread the keys
  uint32_t h = static_cast<uint32_t>(k0) ^ (static_cast<uint32_t>(k0 >> 32) * 0x9E3779B1u) ^
               (static_cast<uint32_t>(k1) * 0x85EBCA77u) ^ (static_cast<uint32_t>(k1 >> 32) * 0xC2B2AE3Du);
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}

// read-only index lookup: gid of the row's group or kGidDefer when it does not exist (yet)
__device__ inline uint32_t index_lookup(const KeyCols &keys, int64_t row, const KeyStore &store,
                                        const unsigned long long *__restrict__ meta, uint32_t mask) {
  const RowHash rh = agg_hash(keys, row);
  uint32_t idx = rh.idx & mask;
  for (int probes = 0; probes < kMaxProbe; probes++) {
    const unsigned long long e = meta[idx];
    if (e == kMetaEmpty) return kGidDefer;
    if (static_cast<uint32_t>(e >> 32) == rh.upper &&
        key_equals_store(keys, row, store, static_cast<uint32_t>(e), rh.upper & 0xFFu))
      return static_cast<uint32_t>(e);
    idx = (idx + 1) & mask;
  }
  return kGidDefer;
}

// ------------------------------------------------------------------ accumulation (one atomic per lane; HBM or LDS words)
__device__ __forceinline__ void add_i128(unsigned long long *lo, unsigned long long *hi, unsigned long long vlo,
                                         unsigned long long vhi) {
  const unsigned long long old = atomicAdd(lo, vlo);
  const unsigned long long carry = (old + vlo) < old ? 1ull : 0ull;
  const unsigned long long d = vhi + carry;
  if (d != 0ull) atomicAdd(hi, d);
}

// Accumulating one input row is split in two so that a thread can issue the loads of all its rows
// before the first atomic: (1) does the row contribute to `op`, and with which 64-bit value
// (float bits / sign-extended integer / bool); (2) the atomic update.
__device__ __forceinline__ bool bit_valid(const uint8_t *valid, int64_t i) {
  return valid == nullptr || ((valid[i >> 3] >> (i & 7)) & 1);  // coldata.Nulls: bit = 1 => present
}
__device__ __forceinline__ void apply_input(const OpK &op, unsigned long long v, unsigned long long seq,
                                            unsigned long long *w0, unsigned long long *w1, uint8_t *seen) {
  switch (op.kind) {
    case ACC_COUNT_ROWS:
    case ACC_COUNT: atomicAdd(w0, 1ull); break;
    case ACC_SUM_F64:
      unsafeAtomicAdd(reinterpret_cast<double *>(w0), __longlong_as_double(static_cast<long long>(v)));
      *seen = 1;
      break;
    case ACC_AVG_F64:
      unsafeAtomicAdd(reinterpret_cast<double *>(w0), __longlong_as_double(static_cast<long long>(v)));
      atomicAdd(w1, 1ull);
      break;
    case ACC_SUM_I128:
      add_i128(w0, w1, v, (v >> 63) ? ~0ull : 0ull);
      *seen = 1;
      break;
    case ACC_MIN_I64: atomicMin(reinterpret_cast<long long *>(w0), static_cast<long long>(v)); *seen = 1; break;
    case ACC_MAX_I64: atomicMax(reinterpret_cast<long long *>(w0), static_cast<long long>(v)); *seen = 1; break;
    case ACC_MIN_F64: atomicMin(w0, f64_to_ordered(__longlong_as_double(static_cast<long long>(v)))); *seen = 1; break;
    case ACC_MAX_F64: atomicMax(w0, f64_to_ordered(__longlong_as_double(static_cast<long long>(v)))); *seen = 1; break;
    case ACC_BOOL_AND: if (!v) atomicAnd(w0, 0ull); *seen = 1; break;
    case ACC_BOOL_OR: if (v) atomicOr(w0, 1ull); *seen = 1; break;
    case ACC_ANY_ROW: atomicMin(w0, seq); break;  // first non-NULL in arrival order
    default: break;
  }
}

// LDS form of apply_input for the kAccRows rows of a thread: the switch on the aggregation runs once, and
// rows that do not contribute are not branched around -- they update a trash group (index G of every plane)
// instead, so the whole update is straight-line code (a divergent branch per (aggregation, row) cost more
// scalar instructions than the atomics themselves: 313 SALU instructions per row in the first profile).
template <int K>
__device__ __forceinline__ void apply_rows_lds(const OpK &op, const unsigned long long (&v)[K], uint32_t takem,
                                               const uint32_t (&gr)[K], uint32_t trash, uint32_t plane,
                                               unsigned long long *lds_words, uint8_t *lds_seen, const int64_t (&row)[K],
                                               unsigned long long seq_base) {
  uint32_t idx[K];
#pragma unroll
  for (int k = 0; k < K; k++) idx[k] = ((takem >> k) & 1u) ? gr[k] : trash;
  unsigned long long *p0 = lds_words + static_cast<size_t>(op.lw0) * plane;
  unsigned long long *p1 = lds_words + static_cast<size_t>(op.lw1) * plane;
  uint8_t *ps = lds_seen + static_cast<size_t>(op.lseen) * plane;
  switch (op.kind) {
    case ACC_COUNT_ROWS:
    case ACC_COUNT:
#pragma unroll
      for (int k = 0; k < K; k++) atomicAdd(p0 + idx[k], 1ull);
      break;
    case ACC_SUM_F64:
#pragma unroll
      for (int k = 0; k < K; k++) unsafeAtomicAdd(reinterpret_cast<double *>(p0 + idx[k]), __longlong_as_double(static_cast<long long>(v[k])));
#pragma unroll
      for (int k = 0; k < K; k++) ps[idx[k]] = 1;
      break;
    case ACC_AVG_F64:
#pragma unroll
      for (int k = 0; k < K; k++) unsafeAtomicAdd(reinterpret_cast<double *>(p0 + idx[k]), __longlong_as_double(static_cast<long long>(v[k])));
#pragma unroll
      for (int k = 0; k < K; k++) atomicAdd(p1 + idx[k], 1ull);
      break;
    case ACC_SUM_I128:
#pragma unroll
      for (int k = 0; k < K; k++) add_i128(p0 + idx[k], p1 + idx[k], v[k], (v[k] >> 63) ? ~0ull : 0ull);
#pragma unroll
      for (int k = 0; k < K; k++) ps[idx[k]] = 1;
      break;
    case ACC_MIN_I64:
#pragma unroll
      for (int k = 0; k < K; k++) atomicMin(reinterpret_cast<long long *>(p0 + idx[k]), static_cast<long long>(v[k]));
#pragma unroll
      for (int k = 0; k < K; k++) ps[idx[k]] = 1;
      break;
    case ACC_MAX_I64:
#pragma unroll
      for (int k = 0; k < K; k++) atomicMax(reinterpret_cast<long long *>(p0 + idx[k]), static_cast<long long>(v[k]));
#pragma unroll
      for (int k = 0; k < K; k++) ps[idx[k]] = 1;
      break;
    case ACC_MIN_F64:
#pragma unroll
      for (int k = 0; k < K; k++) atomicMin(p0 + idx[k], f64_to_ordered(__longlong_as_double(static_cast<long long>(v[k]))));
#pragma unroll
      for (int k = 0; k < K; k++) ps[idx[k]] = 1;
      break;
    case ACC_MAX_F64:
#pragma unroll
      for (int k = 0; k < K; k++) atomicMax(p0 + idx[k], f64_to_ordered(__longlong_as_double(static_cast<long long>(v[k]))));
#pragma unroll
      for (int k = 0; k < K; k++) ps[idx[k]] = 1;
      break;
    case ACC_BOOL_AND:
#pragma unroll
      for (int k = 0; k < K; k++) atomicAnd(p0 + idx[k], v[k] ? 1ull : 0ull);
#pragma unroll
      for (int k = 0; k < K; k++) ps[idx[k]] = 1;
      break;
    case ACC_BOOL_OR:
#pragma unroll
      for (int k = 0; k < K; k++) atomicOr(p0 + idx[k], v[k] ? 1ull : 0ull);
#pragma unroll
      for (int k = 0; k < K; k++) ps[idx[k]] = 1;
      break;
    case ACC_ANY_ROW:  // first non-NULL in arrival order
#pragma unroll
      for (int k = 0; k < K; k++) atomicMin(p0 + idx[k], seq_base + static_cast<unsigned long long>(row[k]));
      break;
    default: break;
  }
}

// The same test-and-load for the kAccRows rows of a thread, written column-wise: every branch is on a
// per-op (wave-uniform) property and every loop body is a plain load, so the compiler issues the loads of
// all rows back to back and waits once (row-wise code waited after every single load).
// FILTER given as a selection on an INT64 column: `col cmp constant`, a NULL never passes (selection_ops.eg.go:34971)
template <int K>
__device__ __forceinline__ void apply_vsel(const VSel &s, const int64_t (&row)[K], bool (&take)[K]) {
  long long x[K];
#pragma unroll
  for (int k = 0; k < K; k++) x[k] = s.data[row[k]];
  if (s.valid != nullptr) {
    uint8_t vb[K];
#pragma unroll
    for (int k = 0; k < K; k++) vb[k] = s.valid[row[k] >> 3];
#pragma unroll
    for (int k = 0; k < K; k++) take[k] = take[k] && ((vb[k] >> (row[k] & 7)) & 1);
  }
#pragma unroll
  for (int k = 0; k < K; k++) take[k] = take[k] && sel_apply(s.cmp, x[k], s.cst);
}
template <int K>
__device__ __forceinline__ void and_valid(const uint8_t *valid, const int64_t (&row)[K], bool (&take)[K]) {
  if (valid == nullptr) return;
  uint8_t vb[K];
#pragma unroll
  for (int k = 0; k < K; k++) vb[k] = valid[row[k] >> 3];
#pragma unroll
  for (int k = 0; k < K; k++) take[k] = take[k] && ((vb[k] >> (row[k] & 7)) & 1);
}

template <int K, bool VC = false>
__device__ __forceinline__ void load_inputs(const OpK &op, const int64_t (&row)[K], bool (&take)[K],
                                            unsigned long long (&v)[K], const VExpr *vex = nullptr, const VSel *vsel = nullptr) {
  if (VC && op.has_filter == 2) {
    apply_vsel<K>(vsel[op.fsel - 1], row, take);
  } else if (op.has_filter) {  // aggregators_util.go:216-234
    uint8_t fb[K];
#pragma unroll
    for (int k = 0; k < K; k++) fb[k] = op.fdata[row[k]];
    if (op.fvalid != nullptr) {
      uint8_t fv[K];
#pragma unroll
      for (int k = 0; k < K; k++) fv[k] = op.fvalid[row[k] >> 3];
#pragma unroll
      for (int k = 0; k < K; k++) take[k] = take[k] && ((fv[k] >> (row[k] & 7)) & 1);
    }
#pragma unroll
    for (int k = 0; k < K; k++) take[k] = take[k] && fb[k] != 0;
  }
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = 0;
  if (op.kind == ACC_COUNT_ROWS) return;
  if (VC && op.vex) {  // computed FLOAT64 input
    const VExpr &e = vex[op.vex - 1];
    double x[K], y[K], z[K];
#pragma unroll
    for (int k = 0; k < K; k++) x[k] = e.x[row[k]];
#pragma unroll
    for (int k = 0; k < K; k++) y[k] = e.y[row[k]];
    if (e.op2 >= 0) {
#pragma unroll
      for (int k = 0; k < K; k++) z[k] = e.z[row[k]];
    }
    and_valid<K>(e.vx, row, take);
    and_valid<K>(e.vy, row, take);
    if (e.op2 >= 0) and_valid<K>(e.vz, row, take);
#pragma unroll
    for (int k = 0; k < K; k++) {
      double r = ex_apply(e.op1, x[k], y[k]);
      if (e.op2 >= 0) r = ex_apply(e.op2, r, z[k]);
      v[k] = static_cast<unsigned long long>(__double_as_longlong(r));
    }
    return;
  }
  if (op.valid != nullptr) {  // aggregates skip NULL inputs
    uint8_t vb[K];
#pragma unroll
    for (int k = 0; k < K; k++) vb[k] = op.valid[row[k] >> 3];
#pragma unroll
    for (int k = 0; k < K; k++) take[k] = take[k] && ((vb[k] >> (row[k] & 7)) & 1);
  }
  switch (op.in_type) {
    case CRGPU_BOOL:
#pragma unroll
      for (int k = 0; k < K; k++) v[k] = reinterpret_cast<const uint8_t *>(op.data)[row[k]] != 0;
      break;
    case CRGPU_INT16:
#pragma unroll
      for (int k = 0; k < K; k++) v[k] = static_cast<unsigned long long>(static_cast<long long>(reinterpret_cast<const int16_t *>(op.data)[row[k]]));
      break;
    case CRGPU_INT32:
#pragma unroll
      for (int k = 0; k < K; k++) v[k] = static_cast<unsigned long long>(static_cast<long long>(reinterpret_cast<const int32_t *>(op.data)[row[k]]));
      break;
    case CRGPU_BYTES: break;  // COUNT / ANY_NOT_NULL: the payload is not needed here
    default:
#pragma unroll
      for (int k = 0; k < K; k++) v[k] = reinterpret_cast<const unsigned long long *>(op.data)[row[k]];  // INT64 / FLOAT64 bits
      break;
  }
}

__device__ __forceinline__ unsigned long long acc_init_w0(int kind) {
  switch (kind) {
    case ACC_MIN_I64: return 0x7FFFFFFFFFFFFFFFull;
    case ACC_MAX_I64: return 0x8000000000000000ull;
    case ACC_MIN_F64: return ~0ull;
    case ACC_BOOL_AND: return 1ull;
    case ACC_ANY_ROW:
    case ACC_MIN_BYTES:
    case ACC_MAX_BYTES: return ~0ull;
    default: return 0ull;
  }
}

// merge the replicas of group g (LDS) into the HBM accumulators of group gid
__device__ inline void lds_flush_op(const OpK &op, const unsigned long long *ldsw, const uint8_t *ldss, uint32_t G,
                                    uint32_t R, uint32_t g, uint32_t gid) {
  if (op.kind == ACC_ANY_KEY) return;
  const unsigned long long *p0 = ldsw + (static_cast<size_t>(op.lw0) * G + g) * R;
  const unsigned long long *p1 = op.lw1 >= 0 ? ldsw + (static_cast<size_t>(op.lw1) * G + g) * R : nullptr;
  bool seen = false;
  if (op.lseen >= 0) {
    const uint8_t *ps = ldss + (static_cast<size_t>(op.lseen) * G + g) * R;
    for (uint32_t r = 0; r < R; r++) seen |= ps[r] != 0;
    if (!seen) return;
  }
  switch (op.kind) {
    case ACC_COUNT_ROWS:
    case ACC_COUNT: {
      unsigned long long c = 0;
      for (uint32_t r = 0; r < R; r++) c += p0[r];
      if (c) atomicAdd(&op.w0[gid], c);
      break;
    }
    case ACC_SUM_F64: {
      double s = 0;
      for (uint32_t r = 0; r < R; r++) s += __longlong_as_double(static_cast<long long>(p0[r]));
      unsafeAtomicAdd(reinterpret_cast<double *>(&op.w0[gid]), s);
      op.seen[gid] = 1;
      break;
    }
    case ACC_AVG_F64: {
      double s = 0;
      unsigned long long c = 0;
      for (uint32_t r = 0; r < R; r++) {
        s += __longlong_as_double(static_cast<long long>(p0[r]));
        c += p1[r];
      }
      if (c) {
        unsafeAtomicAdd(reinterpret_cast<double *>(&op.w0[gid]), s);
        atomicAdd(&op.w1[gid], c);
      }
      break;
    }
    case ACC_SUM_I128: {
      unsigned long long lo = 0, hi = 0;
      for (uint32_t r = 0; r < R; r++) {
        const unsigned long long s = lo + p0[r];
        hi += p1[r] + (s < lo ? 1ull : 0ull);
        lo = s;
      }
      add_i128(&op.w0[gid], &op.w1[gid], lo, hi);
      op.seen[gid] = 1;
      break;
    }
    case ACC_MIN_I64:
    case ACC_MAX_I64: {
      long long v = static_cast<long long>(p0[0]);
      for (uint32_t r = 1; r < R; r++) {
        const long long t = static_cast<long long>(p0[r]);
        v = op.kind == ACC_MIN_I64 ? (t < v ? t : v) : (t > v ? t : v);
      }
      if (op.kind == ACC_MIN_I64) atomicMin(reinterpret_cast<long long *>(&op.w0[gid]), v);
      else atomicMax(reinterpret_cast<long long *>(&op.w0[gid]), v);
      op.seen[gid] = 1;
      break;
    }
    case ACC_MIN_F64:
    case ACC_MAX_F64: {
      unsigned long long v = p0[0];
      for (uint32_t r = 1; r < R; r++) v = op.kind == ACC_MIN_F64 ? (p0[r] < v ? p0[r] : v) : (p0[r] > v ? p0[r] : v);
      if (op.kind == ACC_MIN_F64) atomicMin(&op.w0[gid], v);
      else atomicMax(&op.w0[gid], v);
      op.seen[gid] = 1;
      break;
    }
    case ACC_BOOL_AND: {
      unsigned long long v = 1;
      for (uint32_t r = 0; r < R; r++) v &= p0[r];
      if (!v) atomicAnd(&op.w0[gid], 0ull);
      op.seen[gid] = 1;
      break;
    }
    case ACC_BOOL_OR: {
      unsigned long long v = 0;
      for (uint32_t r = 0; r < R; r++) v |= p0[r];
      if (v) atomicOr(&op.w0[gid], 1ull);
      op.seen[gid] = 1;
      break;
    }
    case ACC_ANY_ROW: {
      unsigned long long v = ~0ull;
      for (uint32_t r = 0; r < R; r++) v = p0[r] < v ? p0[r] : v;
      if (v != ~0ull) atomicMin(&op.w0[gid], v);
      break;
    }
    default: break;
  }
}

// merge group g of a single-replica partial table into the HBM accumulators when no other workgroup touches
// these groups during the kernel: plain read-modify-write, no atomics
__device__ inline void lds_flush_exclusive(const OpK &op, const unsigned long long *ldsw, const uint8_t *ldss, uint32_t G,
                                           uint32_t g, uint32_t gid) {
  if (op.kind == ACC_ANY_KEY) return;
  const unsigned long long v0 = ldsw[static_cast<size_t>(op.lw0) * G + g];
  const unsigned long long v1 = op.lw1 >= 0 ? ldsw[static_cast<size_t>(op.lw1) * G + g] : 0ull;
  if (op.lseen >= 0 && ldss[static_cast<size_t>(op.lseen) * G + g] == 0) return;
  switch (op.kind) {
    case ACC_COUNT_ROWS:
    case ACC_COUNT: if (v0) op.w0[gid] += v0; break;
    case ACC_SUM_F64:
      op.w0[gid] = static_cast<unsigned long long>(__double_as_longlong(
          __longlong_as_double(static_cast<long long>(op.w0[gid])) + __longlong_as_double(static_cast<long long>(v0))));
      op.seen[gid] = 1;
      break;
    case ACC_AVG_F64:
      if (v1) {
        op.w0[gid] = static_cast<unsigned long long>(__double_as_longlong(
            __longlong_as_double(static_cast<long long>(op.w0[gid])) + __longlong_as_double(static_cast<long long>(v0))));
        op.w1[gid] += v1;
      }
      break;
    case ACC_SUM_I128: {
      const unsigned long long lo = op.w0[gid], s = lo + v0;
      op.w0[gid] = s;
      op.w1[gid] += v1 + (s < lo ? 1ull : 0ull);
      op.seen[gid] = 1;
      break;
    }
    case ACC_MIN_I64: {
      const long long o = static_cast<long long>(op.w0[gid]), t = static_cast<long long>(v0);
      op.w0[gid] = static_cast<unsigned long long>(t < o ? t : o);
      op.seen[gid] = 1;
      break;
    }
    case ACC_MAX_I64: {
      const long long o = static_cast<long long>(op.w0[gid]), t = static_cast<long long>(v0);
      op.w0[gid] = static_cast<unsigned long long>(t > o ? t : o);
      op.seen[gid] = 1;
      break;
    }
    case ACC_MIN_F64: { const unsigned long long o = op.w0[gid]; op.w0[gid] = v0 < o ? v0 : o; op.seen[gid] = 1; break; }
    case ACC_MAX_F64: { const unsigned long long o = op.w0[gid]; op.w0[gid] = v0 > o ? v0 : o; op.seen[gid] = 1; break; }
    case ACC_BOOL_AND: op.w0[gid] &= v0; op.seen[gid] = 1; break;
    case ACC_BOOL_OR: op.w0[gid] |= v0; op.seen[gid] = 1; break;
    case ACC_ANY_ROW: { const unsigned long long o = op.w0[gid]; op.w0[gid] = v0 < o ? v0 : o; break; }
    default: break;
  }
}

struct AccArgs {
  KeyCols keys;
  KeyStore store;
  OpK ops[kMaxOpsPerLaunch];
  int32_t naggs;           // ops of this launch
  int32_t record_misses;   // FUSED: list rows whose group does not exist (first launch of a pass only)
  int64_t m;               // rows of this pass (positions into `list`, or chunk rows when list == nullptr)
  const uint32_t *list;    // chunk row of each position, or nullptr for the identity
  const uint32_t *out_gid; // !FUSED: group of each position (resolve kernel)
  const unsigned long long *meta;
  uint32_t mask;
  uint32_t gid_lo;         // LDS: first group of this sweep
  uint32_t G;              // LDS: groups held by the partial table
  uint32_t R;              // LDS: replicas (power of two)
  uint32_t nwords, nseen;  // LDS: planes
  unsigned long long seq_base;  // global arrival number of row 0 of this chunk
  uint32_t *miss_list;     // FUSED: chunk rows whose group does not exist yet
  unsigned int *counters;  // [0] new groups, [1] deferred / missed rows
  // PK: LDS key table over the packed keys of all groups
  PackSpec pack;
  uint32_t C;                             // key table entries (power of two, >= 2 G)
  int32_t cf_kind;                        // a FILTER shared by every op of the launch, tested once per row: 0 none, 1 BOOL column, 2 vsel[0]
  const uint8_t *cf_data, *cf_valid;
  VExpr vex[kMaxVex];
  VSel vsel[kMaxVsel];
};
static_assert(sizeof(AccArgs) <= 4096, "kernel arguments are limited to 4 KiB");

// FUSED: hash -> index lookup -> accumulate (misses are listed);  !FUSED: the group comes from out_gid.
// LDS: accumulate into the workgroup's partial table and merge it at the end;  !LDS: HBM atomics.
// PK (with FUSED and LDS): the lookup runs against an LDS table of the groups' packed keys; the HBM index
// only serves unpackable rows and rows whose group does not exist yet.
// VC: computed inputs / selections / a common FILTER are in play (the plain form keeps them out of its instruction stream:
// the pass is instruction-issue bound, and the extra uniform tests cost the Q1 shape 2.22 -> 2.60 ms).
template <bool FUSED, bool LDS, bool PK, bool VC = false>
__global__ __launch_bounds__(1024) void agg_accum_kernel(AccArgs a) {
  extern __shared__ unsigned long long lds_words[];
  const uint32_t plane = (a.G + 1) * a.R;  // one plane per accumulator word: G groups + the trash group, R replicas each
  uint8_t *lds_seen = reinterpret_cast<uint8_t *>(lds_words + static_cast<size_t>(a.nwords) * plane);
  // key table: k0[C] | k1[C] (two-word keys) | gid[C]
  unsigned long long *t_k0 = reinterpret_cast<unsigned long long *>(
      lds_seen + ((static_cast<size_t>(a.nseen) * plane + 7) & ~size_t(7)));
  unsigned long long *t_k1 = t_k0 + a.C;
  uint32_t *t_gid = reinterpret_cast<uint32_t *>(a.pack.two_words ? t_k1 + a.C : t_k1);
  const KeyStore &store = a.store;
  if (PK) {
    for (uint32_t i = threadIdx.x; i < a.C; i += blockDim.x) t_gid[i] = kGidDefer;
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < a.G; g += blockDim.x) {
      unsigned long long k0, k1;
      if (!pack_group(store, a.pack, g, k0, k1)) continue;
      uint32_t idx = pk_hash(k0, k1) & (a.C - 1);
      while (atomicCAS(&t_gid[idx], kGidDefer, g) != kGidDefer) idx = (idx + 1) & (a.C - 1);  // groups are distinct keys
      t_k0[idx] = k0;
      if (a.pack.two_words) t_k1[idx] = k1;
    }
  }
  if (LDS) {
    const uint32_t per = plane;
    for (int f = 0; f < a.naggs; f++) {
      const OpK &op = a.ops[f];
      if (op.lw0 >= 0) {
        const unsigned long long init = acc_init_w0(op.kind);
        for (uint32_t i = threadIdx.x; i < per; i += blockDim.x) lds_words[static_cast<size_t>(op.lw0) * per + i] = init;
      }
      if (op.lw1 >= 0)
        for (uint32_t i = threadIdx.x; i < per; i += blockDim.x) lds_words[static_cast<size_t>(op.lw1) * per + i] = 0ull;
    }
    for (uint32_t i = threadIdx.x; i < a.nseen * per; i += blockDim.x) lds_seen[i] = 0;
    __syncthreads();
  }
  const uint32_t rep = threadIdx.x & (a.R - 1);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t mround = (a.m + 63) & ~int64_t(63);  // whole waves stay in the loop (wave-level collectives below)
  // kAccRows rows per thread and iteration: their loads overlap, and each op descriptor is fetched once for all of them
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i0 < mround; i0 += stride * kAccRows) {
    int64_t row[kAccRows];
    uint32_t gid[kAccRows];
#pragma unroll
    for (int k = 0; k < kAccRows; k++) {
      const int64_t i = i0 + k * stride;
      const bool active = i < a.m;
      row[k] = active ? (a.list != nullptr ? static_cast<int64_t>(a.list[i]) : i) : 0;
      gid[k] = kGidDefer;
      if (!FUSED && active) gid[k] = a.out_gid[i];
    }
    if (PK) {
      bool ok[kAccRows];
      unsigned long long k0[kAccRows], k1[kAccRows];
#pragma unroll
      for (int k = 0; k < kAccRows; k++) ok[k] = i0 + k * stride < a.m;
      pack_rows<kAccRows>(a.keys, a.pack, row, ok, k0, k1);
      // First probe of all rows as straight-line code (unconditional LDS reads, selects): the key table is at most
      // half full, so nearly every row is decided here; the probe loop runs only behind a wave-uniform test.  (The
      // per-row loop with its breaks cost more scalar exec-mask instructions than everything else in the pass.)
      uint32_t idx[kAccRows], g0[kAccRows];
      unsigned long long q0[kAccRows], q1[kAccRows];
      bool need[kAccRows], more = false;
#pragma unroll
      for (int k = 0; k < kAccRows; k++) idx[k] = pk_hash(k0[k], k1[k]) & (a.C - 1);
#pragma unroll
      for (int k = 0; k < kAccRows; k++) g0[k] = t_gid[idx[k]];
#pragma unroll
      for (int k = 0; k < kAccRows; k++) q0[k] = t_k0[idx[k]];
#pragma unroll
      for (int k = 0; k < kAccRows; k++) q1[k] = a.pack.two_words ? t_k1[idx[k]] : k1[k];
#pragma unroll
      for (int k = 0; k < kAccRows; k++) {
        const bool occupied = ok[k] && g0[k] != kGidDefer;
        const bool hit = occupied && q0[k] == k0[k] && q1[k] == k1[k];
        gid[k] = hit ? g0[k] : kGidDefer;
        need[k] = occupied && !hit;
        more |= need[k];
      }
      if (__any(more)) {
#pragma unroll
        for (int k = 0; k < kAccRows; k++) {
          if (!need[k]) continue;
          uint32_t ix = (idx[k] + 1) & (a.C - 1);
          for (;;) {
            const uint32_t g = t_gid[ix];
            if (g == kGidDefer) break;
            if (t_k0[ix] == k0[k] && (!a.pack.two_words || t_k1[ix] == k1[k])) {
              gid[k] = g;
              break;
            }
            ix = (ix + 1) & (a.C - 1);
          }
        }
      }
    }
    if (FUSED) {
      // rows the key table did not decide (unpackable key, group not in the table, new key): rare, so the whole
      // block sits behind one wave-uniform test instead of one divergent region per row
      bool open_rows = false;
#pragma unroll
      for (int k = 0; k < kAccRows; k++) open_rows |= i0 + k * stride < a.m && gid[k] == kGidDefer;
      if (__any(open_rows)) {
#pragma unroll
        for (int k = 0; k < kAccRows; k++) {
          const bool active = i0 + k * stride < a.m;
          if (active && gid[k] == kGidDefer) gid[k] = index_lookup(a.keys, row[k], store, a.meta, a.mask);
          const bool miss = active && gid[k] == kGidDefer;
          const uint64_t mm = a.record_misses ? __ballot(miss) : 0;
          if (mm) {
            const int lead = __ffsll(static_cast<long long>(mm)) - 1;
            unsigned int base = 0;
            if (lane_id() == static_cast<uint32_t>(lead)) base = atomicAdd(&a.counters[1], static_cast<unsigned int>(popc64(mm)));
            base = __shfl(base, lead, 64);
            if (miss) a.miss_list[base + rank_in_mask(mm)] = static_cast<uint32_t>(row[k]);
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < kAccRows; k++)
        if (gid[k] != kGidDefer && (gid[k] & kGidSlotRef)) gid[k] = static_cast<uint32_t>(a.meta[gid[k] & ~kGidSlotRef]);
    }
    if (LDS) {
#pragma unroll
      for (int k = 0; k < kAccRows; k++) {
        const uint32_t rel = gid[k] - a.gid_lo;  // another sweep's group: the unsigned compare also covers gid < gid_lo
        gid[k] = gid[k] != kGidDefer && rel < a.G ? rel : kGidDefer;
      }
    }
    uint32_t gr[kAccRows];  // LDS: position of the row's group inside a plane
#pragma unroll
    for (int k = 0; k < kAccRows; k++) gr[k] = LDS && gid[k] != kGidDefer ? gid[k] * a.R + rep : 0;
    bool pass[kAccRows];
#pragma unroll
    for (int k = 0; k < kAccRows; k++) pass[k] = gid[k] != kGidDefer;
    if (VC && a.cf_kind == 2) {
      apply_vsel<kAccRows>(a.vsel[0], row, pass);
    } else if (VC && a.cf_kind == 1) {
      uint8_t fb[kAccRows];
#pragma unroll
      for (int k = 0; k < kAccRows; k++) fb[k] = a.cf_data[row[k]];
      and_valid<kAccRows>(a.cf_valid, row, pass);
#pragma unroll
      for (int k = 0; k < kAccRows; k++) pass[k] = pass[k] && fb[k] != 0;
    }
    // kOpGroup aggregations at a time: all their input loads are issued before the first atomic
    for (int f0 = 0; f0 < a.naggs; f0 += kOpGroup) {
      unsigned long long v[kOpGroup][kAccRows];
      uint32_t takem[kOpGroup];
#pragma unroll
      for (int u = 0; u < kOpGroup; u++) {
        takem[u] = 0;
        if (f0 + u < a.naggs) {
          bool take[kAccRows];
#pragma unroll
          for (int k = 0; k < kAccRows; k++) take[k] = pass[k];
          load_inputs<kAccRows, VC>(a.ops[f0 + u], row, take, v[u], a.vex, a.vsel);
#pragma unroll
          for (int k = 0; k < kAccRows; k++) takem[u] |= (take[k] ? 1u : 0u) << k;
        }
      }
#pragma unroll
      for (int u = 0; u < kOpGroup; u++) {
        if (f0 + u >= a.naggs) break;
        const OpK &op = a.ops[f0 + u];
        if (LDS) {
          apply_rows_lds<kAccRows>(op, v[u], takem[u], gr, a.G * a.R + rep, plane, lds_words, lds_seen, row, a.seq_base);
        } else {
#pragma unroll
          for (int k = 0; k < kAccRows; k++) {
            if (!((takem[u] >> k) & 1u)) continue;
            const unsigned long long seq = a.seq_base + static_cast<unsigned long long>(row[k]);
            apply_input(op, v[u][k], seq, op.w0 + gid[k], op.w1 + gid[k], op.seen + gid[k]);
          }
        }
      }
    }
  }
  if (LDS) {
    __syncthreads();
    const uint32_t pairs = a.G * static_cast<uint32_t>(a.naggs);
    for (uint32_t t = threadIdx.x; t < pairs; t += blockDim.x) {
      const uint32_t f = t / a.G, g = t - f * a.G;
      lds_flush_op(a.ops[f], lds_words, lds_seen, a.G + 1, a.R, g, a.gid_lo + g);
    }
  }
}

// ------------------------------------------------------------------ deduplicated inputs (fused pass, LDS, packed keys)
// Q1-style aggregation lists name few INPUTS many times: SUM(qty) and AVG(qty), SUM(price) and AVG(price), every AVG
// with its own count, COUNT_ROWS -- eleven accumulator words over five distinct NULL-free FLOAT64 inputs under one
// WHERE; SUM_INT(x), MIN(x), MAX(x), COUNT_ROWS is one input with four.  When every aggregation of the handle is
// COUNT_ROWS / COUNT / SUM / AVG / SUM_INT / MIN / MAX / BOOL_AND / BOOL_OR over NULL-free inputs (plain columns or
// computed FLOAT64 ones) and they share their FILTER, the generic pass (one load + one or two LDS atomics + a uniform
// switch PER AGGREGATION and row) is replaced by this one: every distinct input is loaded ONCE per row, with the inputs
// unrolled at compile time, and feeds only the accumulations somebody needs of it (sum, min, max: one LDS atomic each)
// beside ONE row count; the aggregations are told apart only when the partial table is merged: SUM / AVG take their
// input's sum, AVG and the counts the row count, BOOL_AND / BOOL_OR the min / max of a 0/1 input.
static constexpr int kSfMaxIn = 8;
static constexpr int kSfMaxPlanes = 24;
struct SfIn {
  const void *x;        // typed by `type`
  const double *y, *z;  // computed FLOAT64 inputs: value = op2(op1(x, y), z)
  int8_t op1, op2, type;
  int8_t p_sum, p_hi, p_min, p_max;  // planes of the partial table (-1: nobody needs that accumulation of this input)
  int8_t chain;  // agg_sf8_kernel: x is the value of the previous slot (no load)
};
struct SfOp {
  int8_t kind, in;  // in: index of the distinct input
  int8_t _p[6];
  unsigned long long *w0, *w1;
  uint8_t *seen;
};
struct SfArgs {
  KeyCols keys;
  KeyStore store;
  PackSpec pack;
  SfIn in[kSfMaxIn];
  VSel cf_sel;
  const uint8_t *cf_data, *cf_valid;
  int32_t cf_kind;     // 0 none, 1 BOOL column, 2 cf_sel
  int32_t nops;
  SfOp ops[kMaxOpsPerLaunch];
  int64_t m;
  const unsigned long long *meta;
  uint32_t mask, G, R, C;
  uint32_t nplanes, _pad;
  uint32_t row_base, _pad2;  // agg_sf8_kernel: the launch covers rows [row_base, row_base + m) of the chunk
  unsigned long long plane_init[kSfMaxPlanes];
  uint32_t *miss_list;
  unsigned int *counters;
};
static_assert(sizeof(SfArgs) <= 4096, "kernel arguments are limited to 4 KiB");

// LDS layout of the sum-family kernels: planes [nplanes][(G + 1) * R] of 64-bit words (plane 0: rows), then the key table.
struct SfLds {
  unsigned long long *cnt, *t_k0, *t_k1;
  uint32_t *t_gid;
  uint32_t plane;
};
__device__ __forceinline__ SfLds sf_prologue(const SfArgs &a, unsigned long long *lds_words) {
  SfLds L;
  L.plane = (a.G + 1) * a.R;  // G groups + the trash group, R replicas each
  L.cnt = lds_words;
  L.t_k0 = lds_words + static_cast<size_t>(a.nplanes) * L.plane;
  L.t_k1 = L.t_k0 + a.C;
  L.t_gid = reinterpret_cast<uint32_t *>(a.pack.two_words ? L.t_k1 + a.C : L.t_k1);
  for (uint32_t i = threadIdx.x; i < a.C; i += blockDim.x) L.t_gid[i] = kGidDefer;
  for (uint32_t p = 0; p < a.nplanes; p++) {
    const unsigned long long init = a.plane_init[p];
    for (uint32_t i = threadIdx.x; i < L.plane; i += blockDim.x) lds_words[static_cast<size_t>(p) * L.plane + i] = init;
  }
  __syncthreads();
  for (uint32_t g = threadIdx.x; g < a.G; g += blockDim.x) {
    unsigned long long k0, k1;
    if (!pack_group(a.store, a.pack, g, k0, k1)) continue;
    uint32_t idx = pk_hash(k0, k1) & (a.C - 1);
    while (atomicCAS(&L.t_gid[idx], kGidDefer, g) != kGidDefer) idx = (idx + 1) & (a.C - 1);
    L.t_k0[idx] = k0;
    if (a.pack.two_words) L.t_k1[idx] = k1;
  }
  __syncthreads();
  return L;
}
__device__ __forceinline__ void sf_merge(const SfArgs &a, unsigned long long *lds_words, const unsigned long long *cnt, uint32_t plane) {
  __syncthreads();
  // merge: the aggregations are told apart only here
  const uint32_t pairs = a.G * static_cast<uint32_t>(a.nops);
  for (uint32_t t = threadIdx.x; t < pairs; t += blockDim.x) {
    const uint32_t f = t / a.G, g = t - f * a.G;
    const SfOp &op = a.ops[f];
    unsigned long long c = 0;
    for (uint32_t r = 0; r < a.R; r++) c += cnt[g * a.R + r];
    if (c == 0) continue;
    const SfIn &e = a.in[op.in >= 0 ? op.in : 0];
    switch (op.kind) {
      case ACC_COUNT_ROWS:
      case ACC_COUNT: atomicAdd(&op.w0[g], c); break;
      case ACC_SUM_F64:
      case ACC_AVG_F64: {
        const unsigned long long *ps = lds_words + static_cast<size_t>(e.p_sum) * plane + g * a.R;
        double sm = 0;
        for (uint32_t r = 0; r < a.R; r++) sm += __longlong_as_double(static_cast<long long>(ps[r]));
        unsafeAtomicAdd(reinterpret_cast<double *>(&op.w0[g]), sm);
        if (op.kind == ACC_AVG_F64) atomicAdd(&op.w1[g], c);
        else op.seen[g] = 1;
        break;
      }
      case ACC_SUM_I128: {
        const unsigned long long *pl = lds_words + static_cast<size_t>(e.p_sum) * plane + g * a.R;
        const unsigned long long *ph = lds_words + static_cast<size_t>(e.p_hi) * plane + g * a.R;
        unsigned long long lo = 0, hi = 0;
        for (uint32_t r = 0; r < a.R; r++) {
          const unsigned long long sl = lo + pl[r];
          hi += ph[r] + (sl < lo ? 1ull : 0ull);
          lo = sl;
        }
        add_i128(&op.w0[g], &op.w1[g], lo, hi);
        op.seen[g] = 1;
        break;
      }
      case ACC_MIN_I64:
      case ACC_MAX_I64:
      case ACC_BOOL_AND:
      case ACC_BOOL_OR: {
        const bool is_min = op.kind == ACC_MIN_I64 || op.kind == ACC_BOOL_AND;
        const unsigned long long *pm = lds_words + static_cast<size_t>(is_min ? e.p_min : e.p_max) * plane + g * a.R;
        long long m = static_cast<long long>(pm[0]);
        for (uint32_t r = 1; r < a.R; r++) {
          const long long x = static_cast<long long>(pm[r]);
          m = is_min ? (x < m ? x : m) : (x > m ? x : m);
        }
        if (op.kind == ACC_MIN_I64) atomicMin(reinterpret_cast<long long *>(&op.w0[g]), m);
        else if (op.kind == ACC_MAX_I64) atomicMax(reinterpret_cast<long long *>(&op.w0[g]), m);
        else if (op.kind == ACC_BOOL_AND) { if (m == 0) atomicAnd(&op.w0[g], 0ull); }
        else { if (m != 0) atomicOr(&op.w0[g], 1ull); }
        op.seen[g] = 1;
        break;
      }
      default: {  // MIN / MAX over FLOAT64 (ordered words)
        const bool is_min = op.kind == ACC_MIN_F64;
        const unsigned long long *pm = lds_words + static_cast<size_t>(is_min ? e.p_min : e.p_max) * plane + g * a.R;
        unsigned long long m = pm[0];
        for (uint32_t r = 1; r < a.R; r++) m = is_min ? (pm[r] < m ? pm[r] : m) : (pm[r] > m ? pm[r] : m);
        if (is_min) atomicMin(&op.w0[g], m);
        else atomicMax(&op.w0[g], m);
        op.seen[g] = 1;
        break;
      }
    }
  }
}

template <int NIN>
__global__ __launch_bounds__(1024) void agg_sumfam_kernel(SfArgs a) {
  extern __shared__ unsigned long long lds_words[];
  const SfLds L = sf_prologue(a, lds_words);
  const uint32_t plane = L.plane;
  unsigned long long *cnt = L.cnt, *t_k0 = L.t_k0, *t_k1 = L.t_k1;
  uint32_t *t_gid = L.t_gid;
  const KeyStore &store = a.store;
  const uint32_t rep = threadIdx.x & (a.R - 1);
  const uint32_t trash = a.G * a.R + rep;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t mround = (a.m + 63) & ~int64_t(63);
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i0 < mround; i0 += stride * kAccRows) {
    int64_t row[kAccRows];
    uint32_t gid[kAccRows];
    bool ok[kAccRows];
    unsigned long long k0[kAccRows], k1[kAccRows];
#pragma unroll
    for (int k = 0; k < kAccRows; k++) {
      const int64_t i = i0 + k * stride;
      ok[k] = i < a.m;
      row[k] = ok[k] ? i : 0;
      gid[k] = kGidDefer;
    }
    // every input column of the rows is requested before the key lookup resolves; values travel as 64-bit words
    // (FLOAT64 bits, sign-extended integers, 0/1)
    unsigned long long v[NIN][kAccRows];
#pragma unroll
    for (int c = 0; c < NIN; c++) {
      const SfIn &e = a.in[c];
      switch (e.type) {
        case CRGPU_BOOL:
#pragma unroll
          for (int k = 0; k < kAccRows; k++) v[c][k] = reinterpret_cast<const uint8_t *>(e.x)[row[k]] != 0;
          break;
        case CRGPU_INT16:
#pragma unroll
          for (int k = 0; k < kAccRows; k++) v[c][k] = static_cast<unsigned long long>(static_cast<long long>(reinterpret_cast<const int16_t *>(e.x)[row[k]]));
          break;
        case CRGPU_INT32:
#pragma unroll
          for (int k = 0; k < kAccRows; k++) v[c][k] = static_cast<unsigned long long>(static_cast<long long>(reinterpret_cast<const int32_t *>(e.x)[row[k]]));
          break;
        default:
#pragma unroll
          for (int k = 0; k < kAccRows; k++) v[c][k] = reinterpret_cast<const unsigned long long *>(e.x)[row[k]];
          break;
      }
      if (e.op1 >= 0) {
        double y[kAccRows];
#pragma unroll
        for (int k = 0; k < kAccRows; k++) y[k] = e.y[row[k]];
#pragma unroll
        for (int k = 0; k < kAccRows; k++)
          v[c][k] = static_cast<unsigned long long>(__double_as_longlong(ex_apply(e.op1, __longlong_as_double(static_cast<long long>(v[c][k])), y[k])));
        if (e.op2 >= 0) {
#pragma unroll
          for (int k = 0; k < kAccRows; k++) y[k] = e.z[row[k]];
#pragma unroll
          for (int k = 0; k < kAccRows; k++)
            v[c][k] = static_cast<unsigned long long>(__double_as_longlong(ex_apply(e.op2, __longlong_as_double(static_cast<long long>(v[c][k])), y[k])));
        }
      }
    }
    bool pass[kAccRows];
#pragma unroll
    for (int k = 0; k < kAccRows; k++) pass[k] = ok[k];
    if (a.cf_kind == 2) {
      apply_vsel<kAccRows>(a.cf_sel, row, pass);
    } else if (a.cf_kind == 1) {
      uint8_t fb[kAccRows];
#pragma unroll
      for (int k = 0; k < kAccRows; k++) fb[k] = a.cf_data[row[k]];
      and_valid<kAccRows>(a.cf_valid, row, pass);
#pragma unroll
      for (int k = 0; k < kAccRows; k++) pass[k] = pass[k] && fb[k] != 0;
    }
    pack_rows<kAccRows>(a.keys, a.pack, row, ok, k0, k1);
    uint32_t idx[kAccRows], g0[kAccRows];
    unsigned long long q0[kAccRows], q1[kAccRows];
    bool need[kAccRows], more = false;
#pragma unroll
    for (int k = 0; k < kAccRows; k++) idx[k] = pk_hash(k0[k], k1[k]) & (a.C - 1);
#pragma unroll
    for (int k = 0; k < kAccRows; k++) g0[k] = t_gid[idx[k]];
#pragma unroll
    for (int k = 0; k < kAccRows; k++) q0[k] = t_k0[idx[k]];
#pragma unroll
    for (int k = 0; k < kAccRows; k++) q1[k] = a.pack.two_words ? t_k1[idx[k]] : k1[k];
#pragma unroll
    for (int k = 0; k < kAccRows; k++) {
      const bool occupied = ok[k] && g0[k] != kGidDefer;
      const bool hit = occupied && q0[k] == k0[k] && q1[k] == k1[k];
      gid[k] = hit ? g0[k] : kGidDefer;
      need[k] = occupied && !hit;
      more |= need[k];
    }
    if (__any(more)) {
#pragma unroll
      for (int k = 0; k < kAccRows; k++) {
        if (!need[k]) continue;
        uint32_t ix = (idx[k] + 1) & (a.C - 1);
        for (;;) {
          const uint32_t g = t_gid[ix];
          if (g == kGidDefer) break;
          if (t_k0[ix] == k0[k] && (!a.pack.two_words || t_k1[ix] == k1[k])) {
            gid[k] = g;
            break;
          }
          ix = (ix + 1) & (a.C - 1);
        }
      }
    }
    bool open_rows = false;
#pragma unroll
    for (int k = 0; k < kAccRows; k++) open_rows |= i0 + k * stride < a.m && gid[k] == kGidDefer;
    if (__any(open_rows)) {  // unpackable key, group not in the table, new key: rare
#pragma unroll
      for (int k = 0; k < kAccRows; k++) {
        const bool active = i0 + k * stride < a.m;
        if (active && gid[k] == kGidDefer) gid[k] = index_lookup(a.keys, row[k], store, a.meta, a.mask);
        const bool miss = active && gid[k] == kGidDefer;
        const uint64_t mm = __ballot(miss);
        if (mm) {
          const int lead = __ffsll(static_cast<long long>(mm)) - 1;
          unsigned int base = 0;
          if (lane_id() == static_cast<uint32_t>(lead)) base = atomicAdd(&a.counters[1], static_cast<unsigned int>(popc64(mm)));
          base = __shfl(base, lead, 64);
          if (miss) a.miss_list[base + rank_in_mask(mm)] = static_cast<uint32_t>(row[k]);
        }
      }
    }
    uint32_t at[kAccRows];
#pragma unroll
    for (int k = 0; k < kAccRows; k++) at[k] = (pass[k] && gid[k] < a.G) ? gid[k] * a.R + rep : trash;
#pragma unroll
    for (int k = 0; k < kAccRows; k++) atomicAdd(cnt + at[k], 1ull);
#pragma unroll
    for (int c = 0; c < NIN; c++) {
      const SfIn &e = a.in[c];
      const bool f64 = e.type == CRGPU_FLOAT64;
      if (e.p_sum >= 0) {
        unsigned long long *ps = lds_words + static_cast<size_t>(e.p_sum) * plane;
        if (f64) {
#pragma unroll
          for (int k = 0; k < kAccRows; k++) unsafeAtomicAdd(reinterpret_cast<double *>(ps + at[k]), __longlong_as_double(static_cast<long long>(v[c][k])));
        } else {
          unsigned long long *ph = lds_words + static_cast<size_t>(e.p_hi) * plane;
#pragma unroll
          for (int k = 0; k < kAccRows; k++) add_i128(ps + at[k], ph + at[k], v[c][k], (v[c][k] >> 63) ? ~0ull : 0ull);
        }
      }
      if (e.p_min >= 0) {
        unsigned long long *pm = lds_words + static_cast<size_t>(e.p_min) * plane;
        if (f64) {
#pragma unroll
          for (int k = 0; k < kAccRows; k++) atomicMin(pm + at[k], f64_to_ordered(__longlong_as_double(static_cast<long long>(v[c][k]))));
        } else {
#pragma unroll
          for (int k = 0; k < kAccRows; k++) atomicMin(reinterpret_cast<long long *>(pm + at[k]), static_cast<long long>(v[c][k]));
        }
      }
      if (e.p_max >= 0) {
        unsigned long long *pm = lds_words + static_cast<size_t>(e.p_max) * plane;
        if (f64) {
#pragma unroll
          for (int k = 0; k < kAccRows; k++) atomicMax(pm + at[k], f64_to_ordered(__longlong_as_double(static_cast<long long>(v[c][k]))));
        } else {
#pragma unroll
          for (int k = 0; k < kAccRows; k++) atomicMax(reinterpret_cast<long long *>(pm + at[k]), static_cast<long long>(v[c][k]));
        }
      }
    }
  }
  sf_merge(a, lds_words, cnt, plane);
}

// ------------------------------------------------------------------ the same pass, every load requested up front
// agg_sumfam_kernel walks its inputs one after the other: a projected input waits for its operands before the next
// input's loads are issued, then the FILTER column, then each key column (BYTES: offsets, then payload) -- about
// eight dependent memory round trips per iteration, and the Q1 pass ran at 1.7 us per trip (3.2 TB/s).  This form
// covers the common shape -- one or two NULL-free key columns, all BYTES (KB) or all 8 bytes wide; COUNT_ROWS, COUNT,
// SUM and AVG over NULL-free FLOAT64 inputs -- with two trips per iteration:
//   trip 1  key values / BYTES offsets, the FILTER column, every input and operand column
//   trip 2  the BYTES payload of the keys
// The inputs arrive as SLOTS (sf8_slots): a slot loads its x or takes the previous slot's value (`chain`), then
// applies at most one operation with a loaded operand y -- price, price * (1 - disc), price * (1 - disc) * (1 + tax)
// are three consecutive slots and four loads.  Absent columns are zero-record buffer descriptors (buf_rsrc); rows
// are addressed by 32-bit byte offsets from wave-uniform bases: one address register per row and column width.
static constexpr int kSf8Rows = 2;  // rows per thread and iteration
__host__ __device__ constexpr uint32_t kSf8Planes(int nin) { return static_cast<uint32_t>((nin + 1) | 1); }  // count + one sum per slot, odd
template <int NIN, int NK, bool KB>
__global__ __launch_bounds__(1024) void agg_sf8_kernel(SfArgs a) {
  // LDS: the partial table, interleaved -- word (g * R + rep) * NPL + p, p = 0 the row count, p = 1 + c the sum of slot c
  // (NPL odd: lanes 56 bytes apart spread over the banks) -- so that the atomics of a row share ONE address register
  // and differ in the instruction's immediate offset; then the key table as in sf_prologue.
  extern __shared__ unsigned long long lds_words[];
  constexpr uint32_t NPL = kSf8Planes(NIN);
  const uint32_t cells = (a.G + 1) * a.R;  // G groups + the trash group, R replicas each
  unsigned long long *t_k0 = lds_words + static_cast<size_t>(cells) * NPL;
  unsigned long long *t_k1 = t_k0 + a.C;
  uint32_t *t_gid = reinterpret_cast<uint32_t *>(a.pack.two_words ? t_k1 + a.C : t_k1);
  for (uint32_t i = threadIdx.x; i < a.C; i += blockDim.x) t_gid[i] = kGidDefer;
  for (uint32_t i = threadIdx.x; i < cells * NPL; i += blockDim.x) lds_words[i] = 0;
  __syncthreads();
  for (uint32_t g = threadIdx.x; g < a.G; g += blockDim.x) {
    unsigned long long k0, k1;
    if (!pack_group(a.store, a.pack, g, k0, k1)) continue;
    uint32_t idx = pk_hash(k0, k1) & (a.C - 1);
    while (atomicCAS(&t_gid[idx], kGidDefer, g) != kGidDefer) idx = (idx + 1) & (a.C - 1);
    t_k0[idx] = k0;
    if (a.pack.two_words) t_k1[idx] = k1;
  }
  __syncthreads();
  const uint32_t rep = threadIdx.x & (a.R - 1);
  const uint32_t trash = a.G * a.R + rep;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t mround = (a.m + 63) & ~int64_t(63);
  // FILTER: an INT64 column compared with a constant (cf_kind 2) or a BOOL column (1), each with an optional bitmap;
  // the comparison as three wave-uniform masks (which of <, ==, > pass)
  const uint8_t *fvalid = a.cf_kind == 2 ? a.cf_sel.valid : (a.cf_kind == 1 ? a.cf_valid : nullptr);
  const int cmp = a.cf_sel.cmp;
  const bool want_lt = cmp == SEL_LT || cmp == SEL_LE || cmp == SEL_NE, want_eq = cmp == SEL_EQ || cmp == SEL_LE || cmp == SEL_GE,
             want_gt = cmp == SEL_GT || cmp == SEL_GE || cmp == SEL_NE;
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i0 < mround; i0 += stride * kSf8Rows) {
    uint32_t row[kSf8Rows], off8[kSf8Rows];  // (a chunk has at most 2^28 rows)
    bool ok[kSf8Rows], live[kSf8Rows];
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) {
      const int64_t i = i0 + k * stride;
      live[k] = ok[k] = i < a.m;
      row[k] = ok[k] ? static_cast<uint32_t>(i) : 0u;
      off8[k] = row[k] << 3;
    }
    // ---- trip 1
    unsigned long long kv[NK][kSf8Rows];
    uint32_t a0[NK][kSf8Rows], len[NK][kSf8Rows];
#pragma unroll
    for (int c = 0; c < NK; c++) {
#pragma unroll
      for (int k = 0; k < kSf8Rows; k++) {
        if (KB) {
          a0[c][k] = ld_at<uint32_t>(a.keys.c[c].offsets, row[k] << 2);
          len[c][k] = ld_at<uint32_t>(a.keys.c[c].offsets + 1, row[k] << 2);
          kv[c][k] = 0;
        } else {
          kv[c][k] = ld_at_nt<unsigned long long>(a.keys.c[c].data, off8[k]);
          a0[c][k] = len[c][k] = 0;
        }
      }
    }
    // (a column that is absent leaves its registers undefined -- __builtin_nondeterministic_value -- so that the join
    // behind the wave-uniform branch needs no copy of the loaded value: a copy there would wait for the load)
    long long fx[kSf8Rows];
    uint32_t fb[kSf8Rows], fvb[kSf8Rows];
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) {
      fx[k] = __builtin_nondeterministic_value(fx[k]);
      fb[k] = __builtin_nondeterministic_value(fb[k]);
      fvb[k] = __builtin_nondeterministic_value(fvb[k]);
    }
    if (a.cf_kind == 2) {
#pragma unroll
      for (int k = 0; k < kSf8Rows; k++) fx[k] = ld_at_nt<long long>(a.cf_sel.data, off8[k]);
    } else if (a.cf_kind == 1) {
#pragma unroll
      for (int k = 0; k < kSf8Rows; k++) fb[k] = ld_at<uint8_t>(a.cf_data, row[k]);
    }
    if (fvalid != nullptr) {
#pragma unroll
      for (int k = 0; k < kSf8Rows; k++) fvb[k] = ld_at<uint8_t>(fvalid, row[k] >> 3);
    }
    unsigned long long v[NIN][kSf8Rows];
    double y[NIN][kSf8Rows];
#pragma unroll
    for (int c = 0; c < NIN; c++) {
#pragma unroll
      for (int k = 0; k < kSf8Rows; k++) {
        v[c][k] = __builtin_nondeterministic_value(v[c][k]);
        y[c][k] = __builtin_nondeterministic_value(y[c][k]);
      }
      if (c == 0 || !a.in[c].chain) {
#pragma unroll
        for (int k = 0; k < kSf8Rows; k++) v[c][k] = ld_at_nt<unsigned long long>(a.in[c].x, off8[k]);
      }
      if (a.in[c].op1 >= 0) {
#pragma unroll
        for (int k = 0; k < kSf8Rows; k++) y[c][k] = ld_at_nt<double>(a.in[c].y, off8[k]);
      }
    }
    // ---- trip 2: the payload of BYTES keys, {length:4, up to 7 bytes} (pack_rows); byte j of every row is one
    // unconditional load (rows shorter than j + 1 re-read their last byte, empty ones the offsets array)
    if (KB) {
#pragma unroll
      for (int c = 0; c < NK; c++) {
        const uint32_t bb = static_cast<uint32_t>(a.pack.c[c].bbytes);
#pragma unroll
        for (int k = 0; k < kSf8Rows; k++) {
          len[c][k] -= a0[c][k];
          if (len[c][k] > bb) ok[k] = false;
        }
      }
      bool any_left = false;
#pragma unroll
      for (int c = 0; c < NK; c++) {
#pragma unroll
        for (int k = 0; k < kSf8Rows; k++) {
          if (!ok[k]) len[c][k] = 0;  // (also the rows another column made unpackable, and the rows past the end)
          kv[c][k] = len[c][k];
          any_left |= len[c][k] > 0;
        }
      }
      for (uint32_t j = 0; __any(any_left); j++) {
        uint32_t b[NK][kSf8Rows];
#pragma unroll
        for (int c = 0; c < NK; c++) {
#pragma unroll
          for (int k = 0; k < kSf8Rows; k++) {
            const uint32_t l = len[c][k];
            const uint32_t at_data = a0[c][k] + (j < l ? j : l - 1);
            b[c][k] = l ? ld_at<uint8_t>(a.keys.c[c].data, at_data) : ld_at<uint8_t>(a.keys.c[c].offsets, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // every byte load is in flight before the first one is used
        any_left = false;
#pragma unroll
        for (int c = 0; c < NK; c++) {
#pragma unroll
          for (int k = 0; k < kSf8Rows; k++) {
            if (j < len[c][k]) kv[c][k] |= static_cast<unsigned long long>(b[c][k]) << (4 + 8 * j);
            any_left |= j + 1 < len[c][k];
          }
        }
      }
    }
    // ---- values, FILTER, packed key
#pragma unroll
    for (int c = 0; c < NIN; c++) {
      const int op = a.in[c].op1;
      if (c > 0 && a.in[c].chain) {
#pragma unroll
        for (int k = 0; k < kSf8Rows; k++) v[c][k] = v[c > 0 ? c - 1 : 0][k];
      }
      if (op >= 0) {
#pragma unroll
        for (int k = 0; k < kSf8Rows; k++)
          v[c][k] = static_cast<unsigned long long>(__double_as_longlong(ex_apply(op, __longlong_as_double(static_cast<long long>(v[c][k])), y[c][k])));
      }
    }
    bool pass[kSf8Rows];
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) {
      pass[k] = live[k];
      if (fvalid != nullptr) pass[k] = pass[k] && ((fvb[k] >> (row[k] & 7)) & 1u) != 0;
      if (a.cf_kind == 2) pass[k] = pass[k] && ((want_lt && fx[k] < a.cf_sel.cst) || (want_eq && fx[k] == a.cf_sel.cst) || (want_gt && fx[k] > a.cf_sel.cst));
      else if (a.cf_kind == 1) pass[k] = pass[k] && fb[k] != 0;
    }
    unsigned long long k0[kSf8Rows], k1[kSf8Rows];
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) k0[k] = k1[k] = 0;
#pragma unroll
    for (int c = 0; c < NK; c++) {
      const PackCol &pc = a.pack.c[c];
      const bool canon = !KB && a.keys.c[c].type == CRGPU_FLOAT64;
#pragma unroll
      for (int k = 0; k < kSf8Rows; k++) {
        unsigned long long w = kv[c][k];
        const double f = __longlong_as_double(static_cast<long long>(w));
        if (canon) w = f != f ? 0x7FF8000000000000ull : (f == 0 ? 0ull : w);
        if (pc.word) k1[k] |= w << pc.shift;
        else k0[k] |= w << pc.shift;
      }
    }
    uint32_t gid[kSf8Rows], idx[kSf8Rows], g0[kSf8Rows];
    unsigned long long q0[kSf8Rows], q1[kSf8Rows];
    bool need[kSf8Rows], more = false;
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) idx[k] = pk_hash(k0[k], k1[k]) & (a.C - 1);
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) g0[k] = t_gid[idx[k]];
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) q0[k] = t_k0[idx[k]];
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) q1[k] = a.pack.two_words ? t_k1[idx[k]] : k1[k];
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) {
      const bool occupied = ok[k] && g0[k] != kGidDefer;
      const bool hit = occupied && q0[k] == k0[k] && q1[k] == k1[k];
      gid[k] = hit ? g0[k] : kGidDefer;
      need[k] = occupied && !hit;
      more |= need[k];
    }
    if (__any(more)) {
#pragma unroll
      for (int k = 0; k < kSf8Rows; k++) {
        if (!need[k]) continue;
        uint32_t ix = (idx[k] + 1) & (a.C - 1);
        for (;;) {
          const uint32_t g = t_gid[ix];
          if (g == kGidDefer) break;
          if (t_k0[ix] == k0[k] && (!a.pack.two_words || t_k1[ix] == k1[k])) {
            gid[k] = g;
            break;
          }
          ix = (ix + 1) & (a.C - 1);
        }
      }
    }
    // rows the LDS key table does not resolve (unpackable key, new group) go to the resolve pass through the miss list
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) {
      const bool miss = live[k] && gid[k] == kGidDefer;
      const uint64_t mm = __ballot(miss);
      if (mm) {
        const int lead = __ffsll(static_cast<long long>(mm)) - 1;
        unsigned int base = 0;
        if (lane_id() == static_cast<uint32_t>(lead)) base = atomicAdd(&a.counters[1], static_cast<unsigned int>(popc64(mm)));
        base = __shfl(base, lead, 64);
        if (miss) a.miss_list[base + rank_in_mask(mm)] = row[k] + a.row_base;
      }
    }
    // rows that do not contribute update the trash group: straight-line code
#pragma unroll
    for (int k = 0; k < kSf8Rows; k++) {
      const uint32_t at = ((pass[k] && gid[k] < a.G) ? gid[k] * a.R + rep : trash) * NPL;
      atomicAdd(lds_words + at, 1ull);
#pragma unroll
      for (int c = 0; c < NIN; c++)
        unsafeAtomicAdd(reinterpret_cast<double *>(lds_words + at + 1 + c), __longlong_as_double(static_cast<long long>(v[c][k])));
    }
  }
  __syncthreads();
  // merge: the aggregations are told apart only here (COUNT_ROWS / COUNT / AVG's count: plane 0; SUM / AVG: their slot's)
  const uint32_t pairs = a.G * static_cast<uint32_t>(a.nops);
  for (uint32_t t = threadIdx.x; t < pairs; t += blockDim.x) {
    const uint32_t f = t / a.G, g = t - f * a.G;
    const SfOp &op = a.ops[f];
    const unsigned long long *w = lds_words + static_cast<size_t>(g) * a.R * NPL;
    unsigned long long c = 0;
    for (uint32_t r = 0; r < a.R; r++) c += w[r * NPL];
    if (c == 0) continue;
    if (op.kind == ACC_COUNT_ROWS || op.kind == ACC_COUNT) {
      atomicAdd(&op.w0[g], c);
    } else {
      double sm = 0;
      for (uint32_t r = 0; r < a.R; r++) sm += __longlong_as_double(static_cast<long long>(w[r * NPL + 1 + op.in]));
      unsafeAtomicAdd(reinterpret_cast<double *>(&op.w0[g]), sm);
      if (op.kind == ACC_AVG_F64) atomicAdd(&op.w1[g], c);
      else op.seen[g] = 1;
    }
  }
}

// ------------------------------------------------------------------ gid-partitioned accumulation
// Too many groups for one LDS partial table, and HBM atomics run at ~24 G/s however they are laid out:
// so the rows are first regrouped by gid RANGE (partition p = gids [p << shift, (p + 1) << shift)), together
// with everything accumulation needs -- the gid, one "takes this row" bit per aggregation (FILTER and NULL
// tests evaluated here, once) and the input values widened to 64 bits, each as a dense column.  Every
// partition is then accumulated by a few workgroups in an LDS partial table exactly like the small case.
static constexpr int kPartTileMin = 16384;    // rows per tile; tiles grow with the partition count so that a (tile, partition) run
                                              // stays >= ~32 rows and its stores combine in L2
static constexpr int kPartRows = 8;           // rows per thread and step (two steps per tile)
static constexpr int kMaxPartCols = 12;
static constexpr uint32_t kMaxParts = 8192;    // 2 x 32 KiB of LDS counters

struct PartArgs {
  int64_t m;
  const uint32_t *list;
  const uint32_t *out_gid;
  const unsigned long long *meta;
  uint32_t shift, P, ntiles;
  int32_t naggs, ncols;
  int32_t tile_rows;          // multiple of 256 * kPartRows
  uint32_t *gid_res;          // [m] resolved gid (kGidDefer: row is not part of this pass)
  uint32_t *tile_hist;        // [P][ntiles]
  const uint64_t *offsets;    // exclusive scan of tile_hist
  uint32_t *gid_out;          // [rows of the pass] partition-ordered
  uint32_t *flags_out;
  const void *src[kMaxPartCols];
  unsigned long long *dst[kMaxPartCols];
  int8_t src_type[kMaxPartCols];
  int32_t _pad2;
  OpK ops[kMaxOpsPerLaunch];  // FILTER / validity pointers of each aggregation
};

__global__ __launch_bounds__(1024) void agg_part_hist_kernel(PartArgs a) {
  __shared__ uint32_t cnt[kMaxParts];
  for (uint32_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    for (uint32_t p = threadIdx.x; p < a.P; p += blockDim.x) cnt[p] = 0;
    __syncthreads();
    const int64_t base = static_cast<int64_t>(tile) * a.tile_rows;
    for (int step = 0; step < a.tile_rows / static_cast<int>(blockDim.x) / kPartRows; step++) {
      uint32_t g[kPartRows];
#pragma unroll
      for (int k = 0; k < kPartRows; k++) {
        const int64_t i = base + (step * kPartRows + k) * blockDim.x + threadIdx.x;
        g[k] = i < a.m ? a.out_gid[i] : kGidDefer;
      }
#pragma unroll
      for (int k = 0; k < kPartRows; k++)
        if (g[k] != kGidDefer && (g[k] & kGidSlotRef)) g[k] = static_cast<uint32_t>(a.meta[g[k] & ~kGidSlotRef]);
#pragma unroll
      for (int k = 0; k < kPartRows; k++) {
        const int64_t i = base + (step * kPartRows + k) * blockDim.x + threadIdx.x;
        if (i < a.m) a.gid_res[i] = g[k];
        if (g[k] != kGidDefer) atomicAdd(&cnt[g[k] >> a.shift], 1u);
      }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < a.P; p += blockDim.x) a.tile_hist[static_cast<size_t>(p) * a.ntiles + tile] = cnt[p];
    __syncthreads();
  }
}

// bit f of the result: aggregation f accumulates this row (FILTER true and, except COUNT_ROWS, input not NULL)
template <int K>
__device__ __forceinline__ void take_bits(const OpK *ops, int naggs, const int64_t (&row)[K], uint32_t (&bits)[K]) {
#pragma unroll
  for (int k = 0; k < K; k++) bits[k] = 0;
  for (int f = 0; f < naggs; f++) {
    const OpK &op = ops[f];
    bool take[K];
#pragma unroll
    for (int k = 0; k < K; k++) take[k] = true;
    if (op.has_filter) {
      uint8_t fb[K];
#pragma unroll
      for (int k = 0; k < K; k++) fb[k] = op.fdata[row[k]];
      if (op.fvalid != nullptr) {
        uint8_t fv[K];
#pragma unroll
        for (int k = 0; k < K; k++) fv[k] = op.fvalid[row[k] >> 3];
#pragma unroll
        for (int k = 0; k < K; k++) take[k] = take[k] && ((fv[k] >> (row[k] & 7)) & 1);
      }
#pragma unroll
      for (int k = 0; k < K; k++) take[k] = take[k] && fb[k] != 0;
    }
    if (op.kind != ACC_COUNT_ROWS && op.valid != nullptr) {
      uint8_t vb[K];
#pragma unroll
      for (int k = 0; k < K; k++) vb[k] = op.valid[row[k] >> 3];
#pragma unroll
      for (int k = 0; k < K; k++) take[k] = take[k] && ((vb[k] >> (row[k] & 7)) & 1);
    }
#pragma unroll
    for (int k = 0; k < K; k++) bits[k] |= (take[k] ? 1u : 0u) << f;
  }
}

__global__ __launch_bounds__(1024) void agg_part_scatter_kernel(PartArgs a) {
  __shared__ uint32_t cnt[kMaxParts], start[kMaxParts];
  for (uint32_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    for (uint32_t p = threadIdx.x; p < a.P; p += blockDim.x) {
      cnt[p] = 0;
      start[p] = static_cast<uint32_t>(a.offsets[static_cast<size_t>(p) * a.ntiles + tile]);
    }
    __syncthreads();
    const int64_t base = static_cast<int64_t>(tile) * a.tile_rows;
    for (int step = 0; step < a.tile_rows / static_cast<int>(blockDim.x) / kPartRows; step++) {
      int64_t row[kPartRows];
      uint32_t g[kPartRows], dest[kPartRows], bits[kPartRows];
#pragma unroll
      for (int k = 0; k < kPartRows; k++) {
        const int64_t i = base + (step * kPartRows + k) * blockDim.x + threadIdx.x;
        const bool active = i < a.m;
        g[k] = active ? a.gid_res[i] : kGidDefer;
        row[k] = active ? (a.list != nullptr ? static_cast<int64_t>(a.list[i]) : i) : 0;
      }
#pragma unroll
      for (int k = 0; k < kPartRows; k++) {
        dest[k] = 0;
        if (g[k] != kGidDefer) {
          const uint32_t p = g[k] >> a.shift;
          dest[k] = start[p] + atomicAdd(&cnt[p], 1u);
        }
      }
      take_bits<kPartRows>(a.ops, a.naggs, row, bits);
#pragma unroll
      for (int k = 0; k < kPartRows; k++) {
        if (g[k] == kGidDefer) continue;
        a.gid_out[dest[k]] = g[k];
        a.flags_out[dest[k]] = bits[k];
      }
      for (int c = 0; c < a.ncols; c++) {
        unsigned long long v[kPartRows];
        switch (a.src_type[c]) {
          case CRGPU_BOOL:
#pragma unroll
            for (int k = 0; k < kPartRows; k++) v[k] = reinterpret_cast<const uint8_t *>(a.src[c])[row[k]] != 0;
            break;
          case CRGPU_INT16:
#pragma unroll
            for (int k = 0; k < kPartRows; k++) v[k] = static_cast<unsigned long long>(static_cast<long long>(reinterpret_cast<const int16_t *>(a.src[c])[row[k]]));
            break;
          case CRGPU_INT32:
#pragma unroll
            for (int k = 0; k < kPartRows; k++) v[k] = static_cast<unsigned long long>(static_cast<long long>(reinterpret_cast<const int32_t *>(a.src[c])[row[k]]));
            break;
          default:
#pragma unroll
            for (int k = 0; k < kPartRows; k++) v[k] = reinterpret_cast<const unsigned long long *>(a.src[c])[row[k]];
            break;
        }
#pragma unroll
        for (int k = 0; k < kPartRows; k++)
          if (g[k] != kGidDefer) a.dst[c][dest[k]] = v[k];
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void agg_part_starts_kernel(const uint64_t *__restrict__ offsets, uint32_t P, uint32_t ntiles,
                                                              uint64_t *__restrict__ starts) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p <= P) starts[p] = offsets[static_cast<size_t>(p) * ntiles];  // p == P: the grand total
}

struct PartWg {  // one workgroup of the partitioned accumulate: rows [begin, end) of partition `part`
  uint32_t part;
  uint32_t exclusive;  // the only workgroup of its partition: merges with plain read-modify-writes instead of atomics
  uint64_t begin, end;
};
struct PartAccArgs {
  OpK ops[kMaxOpsPerLaunch];  // data: the partition-ordered 64-bit value column of the op's input
  int32_t naggs;
  uint32_t shift;
  uint32_t nwords, nseen;
  int64_t ngroups;
  const uint32_t *gid;
  const uint32_t *flags;
  const PartWg *wgs;
};
__global__ __launch_bounds__(1024) void agg_part_accum_kernel(PartAccArgs a) {
  extern __shared__ unsigned long long lds_words[];
  const PartWg d = a.wgs[blockIdx.x];
  const uint32_t gid_lo = d.part << a.shift;
  const int64_t left = a.ngroups - static_cast<int64_t>(gid_lo);
  const uint32_t G = static_cast<uint32_t>(left < (int64_t(1) << a.shift) ? left : (int64_t(1) << a.shift));
  const uint32_t plane = G + 1;  // + the trash group (see apply_rows_lds)
  uint8_t *lds_seen = reinterpret_cast<uint8_t *>(lds_words + static_cast<size_t>(a.nwords) * plane);
  for (int f = 0; f < a.naggs; f++) {
    const OpK &op = a.ops[f];
    if (op.lw0 >= 0) {
      const unsigned long long init = acc_init_w0(op.kind);
      for (uint32_t i = threadIdx.x; i < plane; i += blockDim.x) lds_words[static_cast<size_t>(op.lw0) * plane + i] = init;
    }
    if (op.lw1 >= 0)
      for (uint32_t i = threadIdx.x; i < plane; i += blockDim.x) lds_words[static_cast<size_t>(op.lw1) * plane + i] = 0ull;
  }
  for (uint32_t i = threadIdx.x; i < a.nseen * plane; i += blockDim.x) lds_seen[i] = 0;
  __syncthreads();
  for (uint64_t j0 = d.begin + threadIdx.x; j0 < d.end; j0 += static_cast<uint64_t>(blockDim.x) * kAccRows) {
    int64_t j[kAccRows];
    uint32_t g[kAccRows], fl[kAccRows];
#pragma unroll
    for (int k = 0; k < kAccRows; k++) {
      const uint64_t jj = j0 + static_cast<uint64_t>(k) * blockDim.x;
      j[k] = static_cast<int64_t>(jj < d.end ? jj : d.end - 1);  // clamped: the loads below stay unconditional
      g[k] = a.gid[j[k]] - gid_lo;
      fl[k] = jj < d.end ? a.flags[j[k]] : 0u;
    }
    for (int f = 0; f < a.naggs; f++) {
      const OpK &op = a.ops[f];
      unsigned long long v[kAccRows];
      if (op.data != nullptr) {
#pragma unroll
        for (int k = 0; k < kAccRows; k++) v[k] = reinterpret_cast<const unsigned long long *>(op.data)[j[k]];
      } else {
#pragma unroll
        for (int k = 0; k < kAccRows; k++) v[k] = 0;
      }
      uint32_t takem = 0;
#pragma unroll
      for (int k = 0; k < kAccRows; k++) takem |= ((fl[k] >> f) & 1u) << k;
      apply_rows_lds<kAccRows>(op, v, takem, g, G, plane, lds_words, lds_seen, j, 0ull);
    }
  }
  __syncthreads();
  const uint32_t pairs = G * static_cast<uint32_t>(a.naggs);
  for (uint32_t t = threadIdx.x; t < pairs; t += blockDim.x) {
    const uint32_t f = t / G, g = t - f * G;
    if (d.exclusive) lds_flush_exclusive(a.ops[f], lds_words, lds_seen, plane, g, gid_lo + g);
    else lds_flush_op(a.ops[f], lds_words, lds_seen, plane, 1, g, gid_lo + g);
  }
}

// ------------------------------------------------------------------ bulk aggregation of a high-cardinality input
// A fresh table fed one large input of mostly distinct keys would pay one random HBM CAS (~10 G/s with the
// returned value) plus a random 8-byte write per row in find-or-insert.  When the shape allows it (one INT64 key
// column and at most one 8-byte value column, no NULLs, no FILTER) the rows are instead radix-partitioned by key
// hash with the join's build-side kernels (radix.hip: 1024 partitions, then <= 128 sub-partitions of ~1024 rows)
// and every sub-partition is grouped and aggregated ENTIRELY IN LDS by one workgroup: equal keys share a
// sub-partition, so its groups are final and are appended to the per-group arrays (one HBM atomic per workgroup
// reserves their group ids).  The hash index is not touched; it is rebuilt from the hash words if more input follows.
// Two forms, chosen from a key sample:
//   sub : ~all keys distinct up to ~32 rows per key: the level-2 split gives fixed regions of <= kBulkRows rows;
//   part: ~16 ... ~2000 groups per level-1 partition (2^14 ... ~2^21 groups in all): no level-2 split; a workgroup
//         walks the variable-length partition in tiles of kBulkRows rows and its groups stay in LDS across the tiles.
static constexpr uint32_t kBulkRows = 2048;   // rows per tile = rows a sub-partition region holds (8192 made the
                                              // kernel 1.5x slower: 8 rows per thread and twice the region traffic)
static constexpr uint32_t kBulkSubSlots = 2048;   // sub form: LDS index slots / groups a sub-partition may hold
static constexpr uint32_t kBulkSubGroups = 1536;
static constexpr uint32_t kBulkSlimGroups = 1280;  // groups a region may hold in the slim form of the general pass
struct BulkArgs {
  const void *rows;             // {key, value} rows of 16 bytes, or (WIDE) {k0, k1, value, length} rows of 32 bytes
  const uint32_t *starts;       // part form: first row of region q (nullptr: q * kBulkRows)
  const uint32_t *counts;       // rows of region q
  OpK ops[kMaxOpsPerLaunch];    // accumulator pointers and LDS plane numbers; inputs all come from the row's value
  int32_t naggs;
  uint32_t nwords, nseen;
  int32_t key_type;
  uint8_t *key_data;            // key store of the single group column
  uint8_t *key_present;
  int32_t key_type2, _pad2;     // WIDE over two INT64 key columns: the second column's key store
  uint8_t *key_data2;
  uint8_t *key_present2;
  uint64_t *b_off;              // WIDE: the BYTES key store; group g's bytes go to b_buf[16 g, 16 g + 16)
  uint32_t *b_len;
  uint8_t *b_buf;
  unsigned long long *gword;
  unsigned int *ngroups;        // running group count (HBM)
  unsigned int *failed;         // a region held more than gmax groups
  unsigned int *not16;          // WIDE: some group's key is not exactly 16 bytes long
  unsigned long long wmask;     // WIDE: all ones when the row's fourth word belongs to the key (its length), zero when it
                                // is a second value (one integer key, two value columns)
  uint8_t op_val[kMaxOpsPerLaunch];  // which of the row's values an aggregation reads (0: the first, 1: the second)
  uint32_t g0;
  uint32_t gmax;                // groups a region may hold (sized to the LDS the planes need)
  uint32_t slots;               // LDS index size (power of two, > gmax)
  uint32_t lkey;                // the part form of narrow rows: the slots' keys sit in LDS too (see agg_bulk_kernel)
  // the all-distinct attempt (TRY): region q's groups are numbered from try_base[q] (prefix sums of the row counts: no
  // group-id atomic); *try_failed is raised by a region with a repeated key -- the attempt is then void as a whole
  const uint64_t *try_base;
  unsigned int *try_failed;     // (a cache line of its own: it is read by every workgroup)
};
static constexpr int kBulkTryThreads = 512;
template <bool WIDE>
struct BulkRow;
template <>
struct BulkRow<false> {
  typedef ulonglong2 T;
  static __device__ __forceinline__ bool same_key(const BulkArgs &, const T &a, const T &b) { return a.x == b.x; }
  static __device__ __forceinline__ unsigned long long key_word(const T &r) { return r.x; }  // the whole key (LKEY)
  static __device__ __forceinline__ unsigned long long value(const T &r) { return r.y; }
  static __device__ __forceinline__ unsigned long long value2(const T &) { return 0ull; }
  static __device__ __forceinline__ uint32_t slot_hash(const BulkArgs &, const T &r) { return static_cast<uint32_t>(fmix64(r.x ^ 0x5851F42D4C957F2Dull) >> 20); }
  // the group's key and the hash word of the index entry it gets if the index is ever rebuilt (agg_hash of one INT64 column)
  static __device__ __forceinline__ void store_key(const BulkArgs &a, uint32_t gid, const T &r) {
    reinterpret_cast<unsigned long long *>(a.key_data)[gid] = r.x;
    a.key_present[gid] = 1;
    const uint64_t h = fmix64(hash_step(0x9E3779B97F4A7C15ull, r.x));
    a.gword[gid] = (static_cast<unsigned long long>((static_cast<uint32_t>(h >> 30) & 0x7FFFFFu) << 8) << 32) |
                   (static_cast<uint32_t>(h) & kIdxHashMask);
  }
};
template <>
struct BulkRow<true> {
  typedef ulonglong4 T;
  static __device__ __forceinline__ bool same_key(const BulkArgs &g, const T &a, const T &b) {
    return a.x == b.x && a.y == b.y && ((a.w ^ b.w) & g.wmask) == 0;
  }
  static __device__ __forceinline__ unsigned long long key_word(const T &r) { return r.x; }  // (never used: LKEY is narrow only)
  static __device__ __forceinline__ unsigned long long value(const T &r) { return r.z; }
  static __device__ __forceinline__ unsigned long long value2(const T &r) { return r.w; }
  static __device__ __forceinline__ uint32_t slot_hash(const BulkArgs &g, const T &r) {
    return static_cast<uint32_t>(fmix64(wide_key_hash(r.x, r.y, r.w & g.wmask) ^ 0x5851F42D4C957F2Dull) >> 20);
  }
  // agg_hash of one BYTES column: the full 8-byte words, the zero-padded rest, then the length; of two integer columns:
  // their sign-extended values one after the other
  static __device__ __forceinline__ void store_key(const BulkArgs &a, uint32_t gid, const T &r) {
    if (a.key_type != CRGPU_BYTES) {
      // INT64 keys only: a per-type switch on the store width here was MISCOMPILED by hipcc 7.2 for gfx950 (the group
      // number of the present-byte store after it was an undefined register on the 8-byte path of the all-distinct
      // branch, so one byte per region was written; found with a first-run-in-a-fresh-process check, since pooled
      // memory holds the 1s of earlier runs).  tests/test_gpu_agg.py::test_agg_bulk_paths_in_a_fresh_process pins it.
      reinterpret_cast<unsigned long long *>(a.key_data)[gid] = r.x;
      a.key_present[gid] = 1;
      uint64_t h = hash_step(0x9E3779B97F4A7C15ull, r.x);
      if (a.key_data2 != nullptr) {  // two integer keys (else: one key, the row carries two values)
        reinterpret_cast<unsigned long long *>(a.key_data2)[gid] = r.y;
        a.key_present2[gid] = 1;
        h = hash_step(h, r.y);
      }
      h = fmix64(h);
      a.gword[gid] = (static_cast<unsigned long long>((static_cast<uint32_t>(h >> 30) & 0x7FFFFFu) << 8) << 32) |
                     (static_cast<uint32_t>(h) & kIdxHashMask);
      return;
    }
    reinterpret_cast<ulonglong2 *>(a.b_buf)[gid] = make_ulonglong2(r.x, r.y);
    a.b_off[gid] = static_cast<uint64_t>(gid) * 16u;
    a.b_len[gid] = static_cast<uint32_t>(r.w);
    if (r.w != 16) *a.not16 = 1u;
    a.key_present[gid] = 1;
    uint64_t h = 0x9E3779B97F4A7C15ull;
    if (r.w > 0) h = hash_step(h, r.x);
    if (r.w > 8) h = hash_step(h, r.y);
    h = fmix64(hash_step(h, 0xB17E5ull + r.w));
    a.gword[gid] = (static_cast<unsigned long long>((static_cast<uint32_t>(h >> 30) & 0x7FFFFFu) << 8) << 32) |
                   (static_cast<uint32_t>(h) & kIdxHashMask);
  }
};
// TRY (the sub form's first pass when a key sample held no repeated key): every region is expected to hold distinct keys
// only, so region q's groups are rows q's rows and their ids are known beforehand -- try_base[q], the prefix sum of the
// regions' row counts: no returning atomic on ONE group counter per region (131072 of them for 2^27 rows, serialised
// at the counter's L2 channel: what held the all-distinct case at 1.8 TB/s), an LDS index only (no accumulator planes)
// and THREADS < 1024, so many regions are in flight per CU.  A region with a repeated key raises *try_failed and the
// whole attempt is void (its writes are overwritten by the general form, which then runs over every region); regions
// that start after the flag is up exit at once.
// SLIM (the sub form's general pass: one tile per region): no slot -> group and group -> row arrays in LDS -- after the
// insert phase a slot's entry is overwritten with its group's local number (the claimant's row number is only needed
// while rows are compared) and the claimants store their groups' keys from registers -- so that the planes for 1280
// groups, the index and the seen bytes fit 44 KiB: three 512-thread workgroups per CU instead of two of 1024 threads
// (16-bit index entries, which make room for a fourth, gain nothing: 2.56 against 2.51 ms on config G's 4 x input).
// LKEY (the part form, one INT64 key): the index holds the KEYS themselves.  A partition is walked in many tiles, and
// the row that claimed a slot tiles ago is in HBM or L2 by now: comparing against it put a global load into every row's
// chain of dependent steps.  A slot is claimed by a 64-bit compare-and-swap of its key against the empty marker (so a
// claimed slot's key is there for whoever sees the claim); the one key equal to the marker gets a slot of its own.
// (a.lkey, a launch-time flag of the narrow 1024-thread form: one more instantiation of this kernel could not have its
// dynamic LDS limit raised -- hipFuncSetAttribute returned "invalid argument" for it on this ROCm --, and a uniform
// branch per lookup costs nothing measurable)
template <bool WIDE, int THREADS, bool TRY, bool SLIM = false>
__global__ __launch_bounds__(THREADS) void agg_bulk_kernel(BulkArgs a) {
  typedef BulkRow<WIDE> R;
  typedef typename R::T Row;
  // dynamic LDS: [nwords][gmax + 1] accumulator words, [slots] slot -> representative row, [slots] slot -> local group
  // id, [gmax] local group id -> representative row, then the seen planes (bytes)
  extern __shared__ unsigned long long lds_words[];
  __shared__ uint32_t s_base, s_claims;
  __shared__ uint32_t s_special;  // LKEY: the key equal to the empty marker has been claimed
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  constexpr unsigned long long kNoKey = ~0ull;
  // (the part form and the all-distinct attempt over narrow rows; in the slim general form, whose compares hit rows its
  // workgroup has just loaded, the key index changed nothing: 2.29 against 2.27 ms on config G's 4 x input)
  const bool LKEY = !WIDE && !SLIM && a.lkey != 0;
  const uint32_t plane = a.gmax + 1;
  // LKEY: [slots] keys instead of the slot -> row index, then [slots + 1] slot -> local group (the last one: the marker's key)
  unsigned long long *kkey = TRY ? lds_words : lds_words + static_cast<size_t>(a.nwords) * plane;
  uint32_t *idx = TRY ? reinterpret_cast<uint32_t *>(lds_words) : reinterpret_cast<uint32_t *>(lds_words + static_cast<size_t>(a.nwords) * plane);
  uint32_t *lid = LKEY ? reinterpret_cast<uint32_t *>(kkey + a.slots) : idx + a.slots;
  uint32_t *rep = lid + a.slots + (LKEY ? 1 : 0);
  uint8_t *lds_seen = SLIM ? reinterpret_cast<uint8_t *>(idx + a.slots) : reinterpret_cast<uint8_t *>(rep + a.gmax);
  const int q = blockIdx.x;
  // TRY: the first ~1024 rows of the region (what a region holds on average; its capacity is kBulkRows) are requested
  // before the row count is known, and the flag and the region's first group id with them: one memory latency in
  // front of the LDS phase instead of two
  constexpr int kPre = TRY ? 1024 / THREADS : 1;
  Row pre[kPre];
  unsigned int flag_seen = 0;
  uint64_t try_first = 0;
  if (TRY) {
    const Row *rq = static_cast<const Row *>(a.rows) + static_cast<int64_t>(q) * kBulkRows;
#pragma unroll
    for (int k = 0; k < kPre; k++) pre[k] = rq[k * THREADS + threadIdx.x];
    if (threadIdx.x == 0) flag_seen = __hip_atomic_load(a.try_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    try_first = a.try_base[q];
  }
  const uint32_t n = a.counts[q];
  if (n == 0) return;
  const Row *rows = static_cast<const Row *>(a.rows) + (a.starts != nullptr ? static_cast<int64_t>(a.starts[q]) : static_cast<int64_t>(q) * kBulkRows);
  const uint32_t smask = a.slots - 1;
  // A region of one tile (the sub form: <= kBulkRows rows) whose keys all turn out distinct needs no accumulator planes:
  // every row IS its group's result.  The planes are therefore set up after the first tile's find-or-insert.
  const bool single = n <= kBulkRows;
  auto init_planes = [&]() {
    for (int f = 0; f < a.naggs; f++) {
      const OpK &op = a.ops[f];
      if (op.lw0 >= 0) {
        const unsigned long long init = acc_init_w0(op.kind);
        for (uint32_t i = threadIdx.x; i < plane; i += blockDim.x) lds_words[static_cast<size_t>(op.lw0) * plane + i] = init;
      }
      if (op.lw1 >= 0)
        for (uint32_t i = threadIdx.x; i < plane; i += blockDim.x) lds_words[static_cast<size_t>(op.lw1) * plane + i] = 0ull;
    }
    for (uint32_t i = threadIdx.x; i < a.nseen * plane; i += blockDim.x) lds_seen[i] = 0;
  };
  if (LKEY) {
    for (uint32_t i = threadIdx.x; i < a.slots; i += blockDim.x) kkey[i] = kNoKey;
    if (threadIdx.x == 0) s_special = 0;
  } else {
    for (uint32_t i = threadIdx.x; i < a.slots; i += blockDim.x) idx[i] = kNone;
  }
  if (!TRY && !single) init_planes();
  if (threadIdx.x == 0) s_claims = 0;
  __syncthreads();
  constexpr int kRows = kBulkRows / THREADS;  // rows per thread and tile
  Row keep_rr[SLIM ? kRows : 1];        // SLIM: the claimants' rows and local group numbers, for the key stores at the end
  uint32_t keep_l[SLIM ? kRows : 1];
  // The 1024-thread form walks partitions of many tiles (the part form): the next tile's rows are requested while this
  // one is looked up and accumulated.  (Not in the 512-thread forms: one tile per region, and no registers to spare.)
  constexpr bool PF = !TRY && !SLIM;
  Row nxt[PF ? kRows : 1];
  if (PF) {
#pragma unroll
    for (int k = 0; k < kRows; k++) {
      const uint32_t i = k * blockDim.x + threadIdx.x;
      if (i < n) nxt[PF ? k : 0] = rows[i];
    }
  }
  for (uint32_t t0 = 0; t0 < n; t0 += kBulkRows) {
    // 1. find-or-insert by key: a slot holds the row that claimed it (its key is the group's key); the claimant
    //    numbers the group
    uint32_t slot_of[kRows];
    uint32_t my_l[kRows];  // SLIM: the local group number a claimant gave its group (kNone: not a claimant)
    unsigned long long val[kRows], val2[kRows];
    Row rr[kRows];
    // (all of the tile's rows are requested before the first one is looked up: the lookups are loops the loads of the
    // later rows would otherwise wait behind)
#pragma unroll
    for (int k = 0; k < kRows; k++) {
      const uint32_t i = t0 + k * blockDim.x + threadIdx.x;
      if (PF) {
        rr[k] = nxt[PF ? k : 0];
        const uint32_t j = i + kBulkRows;
        if (j < n) nxt[PF ? k : 0] = rows[j];
      } else if (i < n) {
        rr[k] = (TRY && k < kPre) ? pre[k < kPre ? k : 0] : rows[i];
      }
    }
#pragma unroll
    for (int k = 0; k < kRows; k++) {
      const uint32_t i = t0 + k * blockDim.x + threadIdx.x;
      slot_of[k] = kNone;
      my_l[k] = kNone;
      val[k] = 0;
      val2[k] = 0;
      bool claimed = false;
      if (i < n) {
        const Row r = rr[k];
        val[k] = R::value(r);
        val2[k] = R::value2(r);
        uint32_t sl = R::slot_hash(a, r) & smask;
        if (LKEY) {
          const unsigned long long key = R::key_word(r);
          if (key == kNoKey) {
            sl = a.slots;
            claimed = atomicCAS(&s_special, 0u, 1u) == 0u;
          } else {
            for (uint32_t probes = 0;; probes++) {
              if (probes == a.slots) {  // the index is full of other keys: far more groups than the planes hold
                atomicAdd(&s_claims, a.gmax + 1);
                sl = kNone;
                break;
              }
              unsigned long long old = kkey[sl];
              if (old == kNoKey) {
                old = atomicCAS(&kkey[sl], kNoKey, key);
                if (old == kNoKey) {
                  claimed = true;
                  break;
                }
              }
              if (old == key) break;
              sl = (sl + 1) & smask;
            }
          }
        } else {
          for (uint32_t probes = 0;; probes++) {
            if (probes == a.slots) {  // the index is full of other keys: far more groups than the planes hold
              atomicAdd(&s_claims, a.gmax + 1);
              sl = kNone;
              break;
            }
            uint32_t cur = idx[sl];
            if (cur == kNone) {
              cur = atomicCAS(&idx[sl], kNone, i);
              if (cur == kNone) {
                claimed = true;
                break;
              }
            }
            if (R::same_key(a, rows[cur], r)) break;
            sl = (sl + 1) & smask;
          }
        }
        slot_of[k] = sl;
      }
      // the claimants number their groups: one LDS atomic per wave (a same-address atomic per claimant serialised the
      // ~1024 claims of an all-distinct region)
      const uint64_t cm = __ballot(claimed);
      if (cm) {
        const int lead = __ffsll(static_cast<long long>(cm)) - 1;
        uint32_t wb = 0;
        if (lane_id() == static_cast<uint32_t>(lead)) wb = atomicAdd(&s_claims, static_cast<uint32_t>(popc64(cm)));
        wb = __shfl(wb, lead, 64);
        if (!TRY && claimed) {
          const uint32_t l = wb + rank_in_mask(cm);
          if (SLIM) {
            my_l[k] = l;
          } else {
            lid[slot_of[k]] = l;  // read after the barrier below
            if (l < a.gmax) rep[l] = i;
          }
        }
      }
    }
    if (TRY && threadIdx.x == 0 && flag_seen != 0) atomicAdd(&s_claims, kBulkRows + 1);  // void already: not worth writing
    __syncthreads();
    if (TRY && s_claims != n) {  // some key repeats (or the index overflowed, or the attempt is void)
      if (threadIdx.x == 0 && flag_seen == 0) atomicOr(a.try_failed, 1u);
      return;
    }
    if (!TRY && s_claims > a.gmax) {  // more groups than the accumulator planes hold: the whole bulk attempt is abandoned
      if (threadIdx.x == 0) atomicOr(a.failed, 1u);
      return;
    }
    if (TRY || (single && s_claims == n)) {
      // every key of the region is distinct: row i of the region is group i (consecutive lanes write consecutive
      // groups); its accumulators are its value
      if (!TRY) {
        if (threadIdx.x == 0) s_base = atomicAdd(a.ngroups, n);
        __syncthreads();
      }
      const uint32_t base = a.g0 + (TRY ? static_cast<uint32_t>(try_first) : s_base);
#pragma unroll
      for (int k = 0; k < kRows; k++) {
        if (slot_of[k] == kNone) continue;
        const uint32_t gid = base + k * blockDim.x + threadIdx.x;
        R::store_key(a, gid, rr[k]);
        for (int f = 0; f < a.naggs; f++) {
          const OpK &op = a.ops[f];
          const unsigned long long v = (WIDE && a.op_val[f]) ? val2[k] : val[k];
          unsigned long long w0 = v, w1 = 0;
          uint8_t seen = 1;
          switch (op.kind) {  // what apply_rows_lds leaves in a group's words after this one row
            case ACC_COUNT_ROWS:
            case ACC_COUNT: w0 = 1; seen = 0; break;
            case ACC_AVG_F64: w1 = 1; seen = 0; break;
            case ACC_SUM_I128: w1 = (v >> 63) ? ~0ull : 0ull; break;
            case ACC_MIN_F64:
            case ACC_MAX_F64: w0 = f64_to_ordered(__longlong_as_double(static_cast<long long>(v))); break;
            case ACC_BOOL_AND:
            case ACC_BOOL_OR: w0 = v ? 1ull : 0ull; break;
            default: break;  // SUM_F64, MIN / MAX over integers: the value itself
          }
          op.w0[gid] = w0;
          if (op.w1 != nullptr) op.w1[gid] = op.lw1 >= 0 ? w1 : 0ull;
          if (op.seen != nullptr) op.seen[gid] = op.lseen >= 0 ? seen : 0;
        }
      }
      return;
    }
    if (SLIM) {  // every compare is done: the claimants' slots now name their groups
#pragma unroll
      for (int k = 0; k < kRows; k++)
        if (my_l[k] != kNone) idx[slot_of[k]] = my_l[k];
    }
    if (single) {
      init_planes();
      __syncthreads();
    } else if (SLIM) {
      __syncthreads();
    }
    // 2. accumulate in LDS (apply_rows_lds: straight-line updates; rows past the end go to the trash group)
    uint32_t gr[kRows];
    int64_t rowi[kRows];
    uint32_t takem = 0;
#pragma unroll
    for (int k = 0; k < kRows; k++) {
      gr[k] = slot_of[k] != kNone ? (SLIM ? idx[slot_of[k]] : lid[slot_of[k]]) : a.gmax;
      if (gr[k] > a.gmax) gr[k] = a.gmax;  // only while a faster wave has already overflowed the planes (attempt abandoned)
      rowi[k] = 0;
      takem |= (slot_of[k] != kNone ? 1u : 0u) << k;
    }
    for (int f = 0; f < a.naggs; f++) {
      // (a select per row: choosing between the two arrays by pointer would move both into scratch memory)
      unsigned long long vsel[kRows];
      const bool second = WIDE && a.op_val[f] != 0;
#pragma unroll
      for (int k = 0; k < kRows; k++) vsel[k] = second ? val2[k] : val[k];
      apply_rows_lds<kRows>(a.ops[f], vsel, takem, gr, a.gmax, plane, lds_words, lds_seen, rowi, 0ull);
    }
    // no barrier needed before the next tile's inserts: they touch idx / lid / rep / s_claims, the updates above touch
    // the planes; lid entries read above were written before the barrier and are never rewritten
    if (SLIM) {
#pragma unroll
      for (int k = 0; k < kRows; k++) {
        keep_rr[SLIM ? k : 0] = rr[k];
        keep_l[SLIM ? k : 0] = my_l[k];
      }
    }
  }
  __syncthreads();
  if (s_claims > a.gmax) {  // a wave that was ahead overflowed after the others had passed the check above
    if (threadIdx.x == 0) atomicOr(a.failed, 1u);
    return;
  }
  // 3. append the groups: ids [base, base + ngrp), everything written with plain coalesced stores
  const uint32_t ngrp = s_claims;
  if (threadIdx.x == 0) s_base = atomicAdd(a.ngroups, ngrp);
  __syncthreads();
  const uint32_t base = a.g0 + s_base;
  if (SLIM) {  // (one tile: the claimants still hold their rows; consecutive claimants of a wave own consecutive groups)
#pragma unroll
    for (int k = 0; k < kRows; k++)
      if (keep_l[SLIM ? k : 0] != kNone) R::store_key(a, base + keep_l[SLIM ? k : 0], keep_rr[SLIM ? k : 0]);
  }
  for (uint32_t l = threadIdx.x; l < ngrp; l += blockDim.x) {
    const uint32_t gid = base + l;
    if (!SLIM) R::store_key(a, gid, rows[rep[l]]);
    for (int f = 0; f < a.naggs; f++) {
      const OpK &op = a.ops[f];
      op.w0[gid] = lds_words[static_cast<size_t>(op.lw0) * plane + l];
      if (op.w1 != nullptr) op.w1[gid] = op.lw1 >= 0 ? lds_words[static_cast<size_t>(op.lw1) * plane + l] : 0ull;
      if (op.seen != nullptr) op.seen[gid] = op.lseen >= 0 ? lds_seen[static_cast<size_t>(op.lseen) * plane + l] : 0;
    }
  }
}

// ------------------------------------------------------------------ find-or-insert (rows whose group may not exist yet)
struct ResolveArgs {
  KeyCols keys;
  KeyStore store;
  int64_t m;
  const uint32_t *list;
  unsigned long long *meta;
  uint32_t mask;
  uint32_t cas_first;             // most rows belong to groups that do not exist yet: claim without looking first
  uint32_t *out_gid;              // [m]
  uint32_t *rep_pos, *rep_slot;   // [new groups]: claimant position / index slot
  unsigned long long *rep_word;   // [new groups]: {upper:32 | idx hash:32}
  uint32_t *next_list;            // deferred chunk rows
  unsigned int *counters;         // [0] new groups, [1] deferred rows
};
static constexpr int kResolveRows = 4;  // rows per thread and tile: one pair of global counter updates per 1024 rows
__global__ __launch_bounds__(256) void agg_resolve_kernel(ResolveArgs a) {
  // Same-address HBM atomics serialise (~10 ns each), so new-group / deferred-row slots are
  // reserved per workgroup tile: waves count in LDS, one thread reserves for the whole tile.
  __shared__ unsigned int s_cnt[2], s_base[2];
  __shared__ unsigned int s_cas_first;  // claim without looking first: follows the share of new groups in the previous tile
  const KeyStore &store = a.store;
  const int64_t tile_rows = 256 * kResolveRows;
  const int64_t ntiles = (a.m + tile_rows - 1) / tile_rows;
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_cas_first = a.cas_first;
  __syncthreads();
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool cas_first = s_cas_first != 0;
    uint32_t res[kResolveRows], slot[kResolveRows], loc[kResolveRows];
    unsigned long long word[kResolveRows];
    int64_t row[kResolveRows];
    bool is_new[kResolveRows], deferred[kResolveRows];
#pragma unroll
    for (int k = 0; k < kResolveRows; k++) {
      const int64_t i = tile * tile_rows + k * 256 + threadIdx.x;
      const bool active = i < a.m;
      row[k] = active ? (a.list != nullptr ? static_cast<int64_t>(a.list[i]) : i) : 0;
      res[k] = kGidDefer;
      slot[k] = 0;
      word[k] = 0;
      is_new[k] = false;
      if (active) {
        const RowHash rh = agg_hash(a.keys, row[k]);
        word[k] = (static_cast<unsigned long long>(rh.upper) << 32) | rh.idx;
        const unsigned long long claim = (static_cast<unsigned long long>(rh.upper | kClaimBit) << 32) | static_cast<uint32_t>(i);
        uint32_t idx = rh.idx & a.mask;
        for (int probes = 0; probes < kMaxProbe; probes++) {
          // A failed CAS still costs an HBM atomic (~24 G/s), a load does not: look first unless most rows are new groups
          unsigned long long e = cas_first ? kMetaEmpty : __hip_atomic_load(&a.meta[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (e == kMetaEmpty) e = atomicCAS(&a.meta[idx], kMetaEmpty, claim);
          if (e == kMetaEmpty) {
            is_new[k] = true;
            slot[k] = idx;
            break;
          }
          const uint32_t up = static_cast<uint32_t>(e >> 32), lo = static_cast<uint32_t>(e);
          if ((up & ~kClaimBit) == rh.upper) {
            if (up & kClaimBit) {
              const int64_t other = a.list != nullptr ? static_cast<int64_t>(a.list[lo]) : static_cast<int64_t>(lo);
              if (keys_equal(a.keys, row[k], a.keys, other, true)) {
                res[k] = kGidSlotRef | idx;
                break;
              }
            } else if (key_equals_store(a.keys, row[k], store, lo, rh.upper & 0xFFu)) {
              res[k] = lo;
              break;
            }
          }
          idx = (idx + 1) & a.mask;
        }
      }
      deferred[k] = active && !is_new[k] && res[k] == kGidDefer;
      // position inside the tile's reservation
      const uint64_t m_new = __ballot(is_new[k]), m_def = __ballot(deferred[k]);
      unsigned int wb = 0;
      if (m_new) {
        const int lead = __ffsll(static_cast<long long>(m_new)) - 1;
        if (lane_id() == static_cast<uint32_t>(lead)) wb = atomicAdd(&s_cnt[0], static_cast<unsigned int>(popc64(m_new)));
        wb = __shfl(wb, lead, 64);
        if (is_new[k]) loc[k] = wb + rank_in_mask(m_new);
      }
      if (m_def) {
        const int lead = __ffsll(static_cast<long long>(m_def)) - 1;
        if (lane_id() == static_cast<uint32_t>(lead)) wb = atomicAdd(&s_cnt[1], static_cast<unsigned int>(popc64(m_def)));
        wb = __shfl(wb, lead, 64);
        if (deferred[k]) loc[k] = wb + rank_in_mask(m_def);
      }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      const unsigned int c = s_cnt[threadIdx.x];
      s_base[threadIdx.x] = c ? atomicAdd(&a.counters[threadIdx.x], c) : 0;
      s_cnt[threadIdx.x] = 0;
      if (threadIdx.x == 0) s_cas_first = c * 2 > static_cast<unsigned int>(tile_rows) ? 1u : 0u;
    }
    __syncthreads();
    const unsigned int base_new = s_base[0], base_def = s_base[1];
#pragma unroll
    for (int k = 0; k < kResolveRows; k++) {
      const int64_t i = tile * tile_rows + k * 256 + threadIdx.x;
      if (is_new[k]) {
        const unsigned int p = base_new + loc[k];
        a.rep_pos[p] = static_cast<uint32_t>(i);
        a.rep_slot[p] = slot[k];
        a.rep_word[p] = word[k];
      }
      if (deferred[k]) a.next_list[base_def + loc[k]] = static_cast<uint32_t>(row[k]);
      // claimants are accumulated by the fixup kernel, deferred rows by a later round
      if (i < a.m) a.out_gid[i] = res[k];
    }
    __syncthreads();  // s_base is rewritten by the next tile
  }
}

// MIN/MAX over BYTES (hash_min_max_agg.eg.go:247-356: bytes.Compare decides): the first 8 bytes of a value, big-endian and
// zero-padded, order values the way bytes.Compare does whenever they differ (a strictly larger prefix means a strictly
// larger value; equal prefixes decide nothing), so one 64-bit atomic min/max per row leaves only the rows that share the
// best prefix for the full comparison.
__device__ __forceinline__ unsigned long long bytes_prefix(const uint8_t *p, uint32_t len) {
  unsigned long long v = 0;
#pragma unroll
  for (uint32_t t = 0; t < 8; t++) v = (v << 8) | (t < len ? p[t] : 0u);
  return v;
}
__device__ __forceinline__ unsigned long long bytes_prefix_of(const DCol &c, int64_t row) {
  const uint32_t a0 = c.offsets[row];
  return bytes_prefix(reinterpret_cast<const uint8_t *>(c.data) + a0, c.offsets[row + 1] - a0);
}
__device__ inline int bytes_compare(const uint8_t *x, uint32_t lx, const uint8_t *y, uint32_t ly) {
  const uint32_t m = lx < ly ? lx : ly;
  for (uint32_t t = 0; t < m; t++)
    if (x[t] != y[t]) return x[t] < y[t] ? -1 : 1;
  return lx < ly ? -1 : (lx > ly ? 1 : 0);
}

// ANY_NOT_NULL over a non-group column: store the value of `row` as group gid's value
// (fixed-width: w1 word; BYTES: per-op append buffer)
__device__ inline void any_store_value(const AggOpDev &op, int f, int64_t row, uint32_t gid, uint64_t *const *any_off,
                                       uint32_t *const *any_len, uint8_t *const *any_buf, unsigned long long *any_tails) {
  switch (op.col.type) {
    case CRGPU_BOOL: op.w1[gid] = col_get_bool(op.col, row); break;
    case CRGPU_FLOAT64: op.w1[gid] = static_cast<unsigned long long>(__double_as_longlong(col_get_f64(op.col, row))); break;
    case CRGPU_BYTES: {
      uint32_t a0 = op.col.offsets[row], len = op.col.offsets[row + 1] - a0;
      unsigned long long off = len ? atomicAdd(&any_tails[f], static_cast<unsigned long long>(len)) : 0ull;
      any_off[f][gid] = off;
      any_len[f][gid] = len;
      const uint8_t *src = reinterpret_cast<const uint8_t *>(op.col.data) + a0;
      for (uint32_t t = 0; t < len; t++) any_buf[f][off + t] = src[t];
      break;
    }
    default: op.w1[gid] = static_cast<unsigned long long>(col_get_int(op.col, row)); break;
  }
}

// FILTER / input of an aggregation in its device-array form (fixup kernels), computed inputs included
__device__ inline bool op_filter_pass(const AggOpDev &op, int64_t row) {
  if (op.has_filter == 0) return true;
  if (col_is_null(op.filter, row)) return false;
  if (op.has_filter == 2) return sel_apply(op.f_cmp, col_get_int(op.filter, row), op.f_cst);
  return col_get_bool(op.filter, row);
}
__device__ inline bool op_input_null(const AggOpDev &op, int64_t row) {
  if (col_is_null(op.col, row)) return true;
  if (op.e_op1 < 0) return false;
  return col_is_null(op.col2, row) || (op.e_op2 >= 0 && col_is_null(op.col3, row));
}
__device__ inline double op_input_f64(const AggOpDev &op, int64_t row) {
  double r = col_get_f64(op.col, row);
  if (op.e_op1 < 0) return r;
  r = ex_apply(op.e_op1, r, col_get_f64(op.col2, row));
  if (op.e_op2 >= 0) r = ex_apply(op.e_op2, r, col_get_f64(op.col3, row));
  return r;
}

// A new group's accumulators after its first row (the claimant), written with plain stores:
// nobody else touches the group before the fixup kernel has finished.
__device__ inline void first_row_store(const AggOpDev &op, int f, int64_t row, unsigned long long seq, uint32_t gid,
                                       uint64_t *const *any_off, uint32_t *const *any_len, uint8_t *const *any_buf,
                                       unsigned long long *any_tails) {
  if (op.kind == ACC_ANY_KEY) return;
  unsigned long long w0 = acc_init_w0(op.kind), w1 = 0;
  uint8_t seen = 0;
  bool take = op_filter_pass(op, row);
  if (take && op.kind != ACC_COUNT_ROWS && op_input_null(op, row)) take = false;
  if (take) {
    switch (op.kind) {
      case ACC_COUNT_ROWS:
      case ACC_COUNT: w0 = 1; break;
      case ACC_SUM_F64: w0 = static_cast<unsigned long long>(__double_as_longlong(op_input_f64(op, row))); seen = 1; break;
      case ACC_AVG_F64: w0 = static_cast<unsigned long long>(__double_as_longlong(op_input_f64(op, row))); w1 = 1; break;
      case ACC_SUM_I128: {
        const long long v = col_get_int(op.col, row);
        w0 = static_cast<unsigned long long>(v);
        w1 = v < 0 ? ~0ull : 0ull;
        seen = 1;
        break;
      }
      case ACC_MIN_I64:
      case ACC_MAX_I64:
        w0 = static_cast<unsigned long long>(op.col.type == CRGPU_BOOL ? static_cast<long long>(col_get_bool(op.col, row)) : col_get_int(op.col, row));
        seen = 1;
        break;
      case ACC_MIN_F64:
      case ACC_MAX_F64: w0 = f64_to_ordered(op_input_f64(op, row)); seen = 1; break;
      case ACC_BOOL_AND:
      case ACC_BOOL_OR: w0 = col_get_bool(op.col, row) ? 1ull : 0ull; seen = 1; break;
      case ACC_ANY_ROW: w0 = seq; break;
      case ACC_MIN_BYTES:
      case ACC_MAX_BYTES: seen = 1; break;  // the claimant's value becomes the stored one below; w0 stays "no candidate"
      default: break;
    }
  }
  if (is_minmax_bytes(op.kind)) w1 = take ? bytes_prefix_of(op.col, row) : (op.kind == ACC_MIN_BYTES ? ~0ull : 0ull);
  op.w0[gid] = w0;
  if (op.w1 != nullptr) op.w1[gid] = w1;
  if (op.seen != nullptr) op.seen[gid] = seen;
  if (op.kind == ACC_ANY_ROW || is_minmax_bytes(op.kind)) {
    if (any_off[f] != nullptr) {
      any_off[f][gid] = 0;
      any_len[f][gid] = 0;
    }
    if (take) any_store_value(op, f, row, gid, any_off, any_len, any_buf, any_tails);
  }
}

// new groups: gid = g0 + p; copy the claimant's key into the key store, initialise the
// accumulators, publish the gid in the index entry
struct FixupArgs {
  KeyCols keys;
  KeyStore store;
  const AggOpDev *ops;
  int32_t naggs;
  uint32_t nnew;
  uint32_t g0;
  uint32_t _pad;
  const uint32_t *list;
  const uint32_t *rep_pos, *rep_slot;
  const unsigned long long *rep_word;
  unsigned long long *meta;
  unsigned long long *gword;
  unsigned long long *b_tails;
  uint64_t *const *any_off;  // [naggs] BYTES ANY_NOT_NULL value store (or nullptr entries)
  uint32_t *const *any_len;
  uint8_t *const *any_buf;
  unsigned long long *any_tails;
  unsigned long long seq_base;
};
// Per-lane byte copy with unaligned 8-byte accesses (global memory allows them), bytes for the tail.
__device__ __forceinline__ void copy_bytes_lane(uint8_t *dst, const uint8_t *src, uint32_t len) {
  uint32_t t = 0;
  for (; t + 8 <= len; t += 8) {
    unsigned long long v;
    __builtin_memcpy(&v, src + t, 8);
    __builtin_memcpy(dst + t, &v, 8);
  }
  for (; t < len; t++) dst[t] = src[t];
}

// Exclusive scan of `v` over the workgroup (256 threads) + one atomicAdd of the total on *tail by thread 0: returns
// tail_before + (sum of v over lower threads).  A same-address HBM atomic costs ~10 ns, so the key bytes of new groups
// are reserved once per workgroup pass, not per group or wave; the groups' payloads also end up adjacent in gid order.
__device__ __forceinline__ unsigned long long block_reserve(uint32_t v, unsigned long long *tail, uint32_t *s_wave,
                                                            unsigned long long *s_base) {
  const uint32_t incl = wave_incl_scan_u32(v);
  const uint32_t w = threadIdx.x >> 6;
  if (lane_id() == 63) s_wave[w] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (uint32_t i = 0; i < (blockDim.x >> 6); i++) {
      const uint32_t c = s_wave[i];
      s_wave[i] = tot;
      tot += c;
    }
    *s_base = tot ? atomicAdd(tail, static_cast<unsigned long long>(tot)) : 0ull;
  }
  __syncthreads();
  const unsigned long long off = *s_base + s_wave[w] + (incl - v);
  __syncthreads();  // s_wave / s_base are reused by the next column
  return off;
}

static constexpr int kFixupGroups = 1;  // new groups per thread and pass (4 measured slower: 5.3 vs 4.6 ms per 2^26 groups with
                                        // 16-byte keys; the reservation atomics are not what bounds the kernel)
__global__ __launch_bounds__(256) void agg_fixup_kernel(FixupArgs a) {
  __shared__ uint32_t s_wave[4];
  __shared__ unsigned long long s_base;
  const KeyStore &st = a.store;
  const unsigned int per_pass = 256 * kFixupGroups;
  const unsigned int npass = (a.nnew + per_pass - 1) / per_pass;
  for (unsigned int pass = blockIdx.x; pass < npass; pass += gridDim.x) {  // uniform per workgroup (block_reserve syncs)
    unsigned int p[kFixupGroups];
    int64_t row[kFixupGroups];
    bool active[kFixupGroups];
#pragma unroll
    for (int j = 0; j < kFixupGroups; j++) {
      p[j] = pass * per_pass + j * 256 + threadIdx.x;
      active[j] = p[j] < a.nnew;
      const uint32_t pos = active[j] ? a.rep_pos[p[j]] : 0;
      row[j] = !active[j] ? 0 : a.list != nullptr ? static_cast<int64_t>(a.list[pos]) : static_cast<int64_t>(pos);
    }
    for (int c = 0; c < a.keys.n; c++) {
      const DCol &x = a.keys.c[c];
      const KeyStoreCol &s = st.c[c];
      if (x.type == CRGPU_BYTES) {
        uint32_t a0[kFixupGroups], len[kFixupGroups], sum = 0;
        bool isnull[kFixupGroups];
#pragma unroll
        for (int j = 0; j < kFixupGroups; j++) {
          a0[j] = 0;
          len[j] = 0;
          isnull[j] = true;
          if (active[j]) {
            isnull[j] = col_is_null(x, row[j]);
            a0[j] = x.offsets[row[j]];
            len[j] = isnull[j] ? 0 : x.offsets[row[j] + 1] - a0[j];
          }
          sum += len[j];
        }
        unsigned long long off = block_reserve(sum, &a.b_tails[c], s_wave, &s_base);
#pragma unroll
        for (int j = 0; j < kFixupGroups; j++) {
          if (!active[j]) continue;
          const uint32_t gid = a.g0 + p[j];
          s.present[gid] = isnull[j] ? 0 : 1;
          s.b_off[gid] = len[j] ? off : 0ull;
          s.b_len[gid] = len[j];
          copy_bytes_lane(s.b_buf + off, reinterpret_cast<const uint8_t *>(x.data) + a0[j], len[j]);
          off += len[j];
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < kFixupGroups; j++) {
        if (!active[j]) continue;
        const uint32_t gid = a.g0 + p[j];
        const bool isnull = col_is_null(x, row[j]);
        s.present[gid] = isnull ? 0 : 1;
        switch (x.type) {
          case CRGPU_BOOL: s.data[gid] = isnull ? 0 : static_cast<uint8_t>(col_get_bool(x, row[j])); break;
          case CRGPU_INT16: reinterpret_cast<int16_t *>(s.data)[gid] = isnull ? 0 : reinterpret_cast<const int16_t *>(x.data)[row[j]]; break;
          case CRGPU_INT32: reinterpret_cast<int32_t *>(s.data)[gid] = isnull ? 0 : reinterpret_cast<const int32_t *>(x.data)[row[j]]; break;
          case CRGPU_INT64: reinterpret_cast<int64_t *>(s.data)[gid] = isnull ? 0 : reinterpret_cast<const int64_t *>(x.data)[row[j]]; break;
          case CRGPU_FLOAT64: reinterpret_cast<double *>(s.data)[gid] = isnull ? 0.0 : col_get_f64(x, row[j]); break;
          default: break;
        }
      }
    }
    for (int j = 0; j < kFixupGroups; j++) {
      if (!active[j]) continue;
      const uint32_t gid = a.g0 + p[j];
      const unsigned long long seq = a.seq_base + static_cast<unsigned long long>(row[j]);
      for (int f = 0; f < a.naggs; f++)
        first_row_store(a.ops[f], f, row[j], seq, gid, a.any_off, a.any_len, a.any_buf, a.any_tails);
      const unsigned long long w = a.rep_word[p[j]];
      a.gword[gid] = w;
      a.meta[a.rep_slot[p[j]]] = (w & 0xFFFFFFFF00000000ull) | gid;
    }
  }
}

// rebuild the index at a new capacity from the per-group hash words
__global__ __launch_bounds__(256) void agg_reindex_kernel(const unsigned long long *__restrict__ gword, uint32_t ngroups,
                                                          unsigned long long *__restrict__ meta, uint32_t mask) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x; gid < ngroups; gid += stride) {
    const unsigned long long w = gword[gid];
    const unsigned long long e = (w & 0xFFFFFFFF00000000ull) | gid;
    uint32_t idx = static_cast<uint32_t>(w) & mask;
    while (atomicCAS(&meta[idx], kMetaEmpty, e) != kMetaEmpty) idx = (idx + 1) & mask;
  }
}

// ANY_NOT_NULL over a non-group column: the row whose arrival number won the
// atomicMin stores its value (fixed-width: w1 word; BYTES: per-op append buffer).
struct AnyFixArgs {
  const AggOpDev *ops;
  int32_t naggs;
  int32_t _pad;
  int64_t m;
  const uint32_t *list;
  const uint32_t *out_gid;
  const unsigned long long *meta;
  unsigned long long seq_base;
  uint64_t *const *any_off;
  uint32_t *const *any_len;
  uint8_t *const *any_buf;
  unsigned long long *any_tails;
};
__global__ __launch_bounds__(256) void agg_any_fixup_kernel(AnyFixArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.m; i += stride) {
    uint32_t gid = a.out_gid[i];
    if (gid == kGidDefer) continue;
    if (gid & kGidSlotRef) gid = static_cast<uint32_t>(a.meta[gid & ~kGidSlotRef]);
    const int64_t row = a.list != nullptr ? static_cast<int64_t>(a.list[i]) : i;
    for (int f = 0; f < a.naggs; f++) {
      const AggOpDev &op = a.ops[f];
      if (op.kind != ACC_ANY_ROW) continue;
      if (op.w0[gid] != a.seq_base + static_cast<unsigned long long>(row)) continue;
      any_store_value(op, f, row, gid, a.any_off, a.any_len, a.any_buf, a.any_tails);
    }
  }
}

// MIN/MAX over BYTES, three passes over the rows of a round (the grid-wide order between them is the kernel boundary):
//   phase 0: w1[gid] = min/max of the rows' 8-byte prefixes (it persists across chunks: it also is the stored value's);
//   phase 1: the rows whose prefix equals w1[gid] elect the best row of this round with a CAS loop on w0[gid], comparing
//            the full values of the contender and the holder (both rows of the current chunk, immutable);
//   phase 2: the elected row compares itself with the group's stored value, replaces it when it is better (appending to
//            the op's value buffer: a group appends at most once per round, a claimant once per group, so the bytes of
//            the chunk bound what a chunk can append) and clears w0[gid] for the next round.
__global__ __launch_bounds__(256) void agg_minmax_bytes_kernel(AnyFixArgs a, int phase) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.m; i += stride) {
    uint32_t gid = a.out_gid[i];
    if (gid == kGidDefer) continue;
    if (gid & kGidSlotRef) gid = static_cast<uint32_t>(a.meta[gid & ~kGidSlotRef]);
    const int64_t row = a.list != nullptr ? static_cast<int64_t>(a.list[i]) : i;
    for (int f = 0; f < a.naggs; f++) {
      const AggOpDev &op = a.ops[f];
      if (!is_minmax_bytes(op.kind)) continue;
      if (!op_filter_pass(op, row)) continue;
      if (col_is_null(op.col, row)) continue;
      const bool is_min = op.kind == ACC_MIN_BYTES;
      const uint8_t *base = reinterpret_cast<const uint8_t *>(op.col.data);
      const uint32_t a0 = op.col.offsets[row], len = op.col.offsets[row + 1] - a0;
      const unsigned long long pre = bytes_prefix(base + a0, len);
      if (phase == 0) {
        const unsigned long long cur = __hip_atomic_load(&op.w1[gid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (is_min ? pre < cur : pre > cur) {
          if (is_min) atomicMin(&op.w1[gid], pre);
          else atomicMax(&op.w1[gid], pre);
        }
        continue;
      }
      if (pre != op.w1[gid]) continue;  // (final since phase 0 ended)
      if (phase == 1) {
        unsigned long long cur = __hip_atomic_load(&op.w0[gid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (true) {
          if (cur != ~0ull) {
            const uint32_t b0 = op.col.offsets[cur], blen = op.col.offsets[cur + 1] - b0;
            const int c = bytes_compare(base + a0, len, base + b0, blen);
            if (!(is_min ? c < 0 : c > 0)) break;  // the holder is at least as good
          }
          const unsigned long long old = atomicCAS(&op.w0[gid], cur, static_cast<unsigned long long>(row));
          if (old == cur) break;
          cur = old;
        }
        continue;
      }
      if (op.w0[gid] != static_cast<unsigned long long>(row)) continue;
      bool better = op.seen[gid] == 0;
      if (!better) {
        const int c = bytes_compare(base + a0, len, a.any_buf[f] + a.any_off[f][gid], a.any_len[f][gid]);
        better = is_min ? c < 0 : c > 0;
      }
      if (better) {
        any_store_value(op, f, row, gid, a.any_off, a.any_len, a.any_buf, a.any_tails);
        op.seen[gid] = 1;
      }
      op.w0[gid] = ~0ull;
    }
  }
}

// finalisation: one output value per group (Flush, e.g. hash_sum_agg.eg.go:562-572); group g is row g
struct FinalArgs {
  AggOpDev op;
  int32_t out_type;
  int32_t _pad;
  int64_t ngroups;
  uint8_t *out_data;
  uint8_t *out_present;  // byte per group
  unsigned int *err;     // set to 1 on SUM_INT out of range
  const uint8_t *key_data;
  const uint8_t *key_present;
};
// value and presence of output row s of one aggregation (the result column's 64-bit word before narrowing)
__device__ __forceinline__ unsigned long long finalize_value(const AggOpDev &op, int out_type, const uint8_t *key_data,
                                                             const uint8_t *key_present, unsigned int *err, int64_t s, uint8_t *present_out) {
  uint8_t present = 1;
  unsigned long long v = 0;
  switch (op.kind) {
    case ACC_COUNT_ROWS:
    case ACC_COUNT: v = op.w0[s]; break;
    case ACC_SUM_F64: present = op.seen[s]; v = op.w0[s]; break;
    case ACC_AVG_F64: {
      const unsigned long long c = op.w1[s];
      present = c != 0;
      double q = present ? __longlong_as_double(static_cast<long long>(op.w0[s])) / static_cast<double>(c) : 0.0;
      v = static_cast<unsigned long long>(__double_as_longlong(q));
      break;
    }
    case ACC_SUM_I128: {
      present = op.seen[s];
      const unsigned long long lo = op.w0[s], hi = op.w1[s];
      const unsigned long long want_hi = (lo >> 63) ? ~0ull : 0ull;
      if (present && hi != want_hi) *err = 1;  // exact total does not fit int64
      v = lo;
      break;
    }
    case ACC_MIN_I64:
    case ACC_MAX_I64: present = op.seen[s]; v = op.w0[s]; break;
    case ACC_MIN_F64:
    case ACC_MAX_F64:
      present = op.seen[s];
      v = static_cast<unsigned long long>(__double_as_longlong(ordered_to_f64(op.w0[s])));
      break;
    case ACC_BOOL_AND:
    case ACC_BOOL_OR: present = op.seen[s]; v = op.w0[s] != 0; break;
    case ACC_ANY_ROW: present = op.w0[s] != ~0ull; v = op.w1[s]; break;
    case ACC_ANY_KEY: {
      present = key_present[s];
      switch (out_type) {
        case CRGPU_BOOL: v = key_data[s]; break;
        case CRGPU_INT16: v = static_cast<unsigned long long>(static_cast<long long>(reinterpret_cast<const int16_t *>(key_data)[s])); break;
        case CRGPU_INT32: v = static_cast<unsigned long long>(static_cast<long long>(reinterpret_cast<const int32_t *>(key_data)[s])); break;
        default: v = reinterpret_cast<const unsigned long long *>(key_data)[s]; break;
      }
      break;
    }
    default: break;
  }
  *present_out = present;
  return present ? v : 0ull;
}
__device__ __forceinline__ void store_narrow(uint8_t *out_data, int out_type, int64_t s, unsigned long long v) {
  switch (out_type) {
    case CRGPU_BOOL: out_data[s] = static_cast<uint8_t>(v != 0); break;
    case CRGPU_INT16: reinterpret_cast<uint16_t *>(out_data)[s] = static_cast<uint16_t>(v); break;
    case CRGPU_INT32: reinterpret_cast<uint32_t *>(out_data)[s] = static_cast<uint32_t>(v); break;
    default: reinterpret_cast<unsigned long long *>(out_data)[s] = v; break;
  }
}
__global__ __launch_bounds__(256) void agg_finalize_kernel(FinalArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; s < a.ngroups; s += stride) {
    uint8_t present;
    const unsigned long long v = finalize_value(a.op, a.out_type, a.key_data, a.key_present, a.err, s, &present);
    store_narrow(a.out_data, a.out_type, s, v);
    a.out_present[s] = present;
  }
}

// The whole result of a small aggregation (<= kFinishSmallGroups groups) in one launch: workgroup f builds output
// column f -- values, validity bitmap, and for BYTES outputs lengths -> offsets -> payload -- where the general path
// spends three to six launches and a read-back per column (0.29 ms for Q1's ten columns and six groups).
static constexpr int kFinishSmallGroups = 2048, kFinishSmallOps = 40;
struct FinSmallCol {
  uint8_t *out_data, *out_valid;
  uint32_t *out_offsets;        // BYTES outputs
  const uint8_t *key_data, *key_present;
  const uint64_t *b_off;        // BYTES: where group s's value starts in b_buf
  const uint32_t *b_len;
  const uint8_t *b_buf, *b_present;  // b_present == nullptr: presence is w0 != ~0 (ANY_NOT_NULL of a column)
  int32_t out_type, bytes;
};
struct FinSmallArgs {
  const AggOpDev *ops;
  unsigned int *err;
  int32_t ngroups, _pad;
  FinSmallCol c[kFinishSmallOps];
};
static_assert(sizeof(FinSmallArgs) <= 4096, "kernel arguments are limited to 4 KiB");
__global__ __launch_bounds__(256) void agg_finish_small_kernel(FinSmallArgs a) {
  __shared__ uint8_t present[kFinishSmallGroups];
  __shared__ uint32_t lens[kFinishSmallGroups + 1];
  __shared__ uint32_t part[257];
  const FinSmallCol &c = a.c[blockIdx.x];
  const AggOpDev &op = a.ops[blockIdx.x];
  const int ng = a.ngroups;
  if (c.bytes == 2) {  // SUM over INT: 16 bytes per group
    for (int s = threadIdx.x; s < ng; s += 256) {
      const uint8_t p = op.seen[s];
      reinterpret_cast<unsigned long long *>(c.out_data)[2 * s] = p ? op.w0[s] : 0ull;
      reinterpret_cast<unsigned long long *>(c.out_data)[2 * s + 1] = p ? op.w1[s] : 0ull;
      c.out_offsets[s] = 16u * static_cast<uint32_t>(s);
      present[s] = p;
    }
    if (threadIdx.x == 0) c.out_offsets[ng] = 16u * static_cast<uint32_t>(ng);
  } else if (!c.bytes) {
    for (int s = threadIdx.x; s < ng; s += 256) {
      uint8_t p;
      const unsigned long long v = finalize_value(op, c.out_type, c.key_data, c.key_present, a.err, s, &p);
      store_narrow(c.out_data, c.out_type, s, v);
      present[s] = p;
    }
  } else {
    for (int s = threadIdx.x; s < ng; s += 256) {
      const uint8_t p = c.b_present == nullptr ? (op.w0[s] != ~0ull) : c.b_present[s];
      present[s] = p;
      lens[s] = p ? c.b_len[s] : 0;
    }
    __syncthreads();
    // exclusive scan of the lengths: thread t owns groups [t * per, (t + 1) * per)
    const int per = (ng + 255) / 256;
    uint32_t sum = 0;
    for (int s = threadIdx.x * per; s < (threadIdx.x + 1) * per && s < ng; s++) sum += lens[s];
    part[threadIdx.x + 1] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      part[0] = 0;
      for (int t = 1; t <= 256; t++) part[t] += part[t - 1];
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (int s = threadIdx.x * per; s < (threadIdx.x + 1) * per && s < ng; s++) {
      const uint32_t len = lens[s];
      c.out_offsets[s] = run;
      const uint8_t *src = c.b_buf + c.b_off[s];
      for (uint32_t t = 0; t < len; t++) c.out_data[run + t] = src[t];
      run += len;
    }
    if (threadIdx.x == 0) c.out_offsets[ng] = part[256];
  }
  __syncthreads();
  for (int b = threadIdx.x; b < (ng + 7) / 8; b += 256) {
    uint32_t bits = 0;
    for (int j = 0; j < 8 && b * 8 + j < ng; j++) bits |= (present[b * 8 + j] ? 1u : 0u) << j;
    c.out_valid[b] = static_cast<uint8_t>(bits);
  }
}

// SUM over INT (AggOpHost::dec128): row s of the BYTES output is the 16 bytes {lo, hi} of group s's exact total
__global__ __launch_bounds__(256) void agg_sum128_out_kernel(const unsigned long long *__restrict__ lo, const unsigned long long *__restrict__ hi,
                                                             const uint8_t *__restrict__ seen, int64_t ngroups,
                                                             unsigned long long *__restrict__ out, uint8_t *__restrict__ present_out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; s < ngroups; s += stride) {
    const uint8_t p = seen[s];
    out[2 * s] = p ? lo[s] : 0ull;
    out[2 * s + 1] = p ? hi[s] : 0ull;
    present_out[s] = p;
  }
}

// SUM_INT whose low word is handed out in place (see crgpu_dev_agg_finish): only the range check remains
__global__ __launch_bounds__(256) void agg_sum_check_kernel(const unsigned long long *__restrict__ lo,
                                                            const unsigned long long *__restrict__ hi,
                                                            const uint8_t *__restrict__ seen, int64_t ngroups,
                                                            unsigned int *__restrict__ err) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  bool bad = false;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; s < ngroups; s += stride) {
    const unsigned long long want_hi = (lo[s] >> 63) ? ~0ull : 0ull;
    bad |= seen[s] && hi[s] != want_hi;
  }
  if (bad) *err = 1;
}

// BYTES outputs (ANY_NOT_NULL of a BYTES group key or column): lengths -> offsets -> payload copy
__global__ __launch_bounds__(256) void bytes_view_kernel(const uint32_t *__restrict__ len, const uint8_t *__restrict__ present_in,
                                                         bool any_row, const unsigned long long *__restrict__ w0,
                                                         int64_t ngroups, uint32_t *__restrict__ lens_out,
                                                         uint8_t *__restrict__ present_out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; s < ngroups; s += stride) {
    const uint8_t p = any_row ? (w0[s] != ~0ull) : present_in[s];
    lens_out[s] = p ? len[s] : 0;
    present_out[s] = p;
  }
}
__global__ __launch_bounds__(256) void bytes_copy_groups_kernel(const uint64_t *__restrict__ off, const uint8_t *__restrict__ buf,
                                                                int64_t ngroups, const uint32_t *__restrict__ dst_off,
                                                                uint8_t *__restrict__ dst) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; s < ngroups; s += stride) {
    const uint32_t o = dst_off[s], len = dst_off[s + 1] - o;
    const uint8_t *src = buf + off[s];
    for (uint32_t t = 0; t < len; t++) dst[o + t] = src[t];
  }
}
__global__ __launch_bounds__(256) void bool_and_valid_kernel(const uint8_t *__restrict__ b, const uint8_t *__restrict__ valid, int64_t n,
                                                             uint8_t *__restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256)
    out[i] = b[i] != 0 && ((valid[i >> 3] >> (i & 7)) & 1);
}
__global__ __launch_bounds__(256) void iota64_kernel(int64_t n, long long *__restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) out[i] = i;
}
__global__ __launch_bounds__(256) void mask_rows_kernel(const long long *__restrict__ rows, int64_t m, uint8_t *__restrict__ mask) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < m; i += static_cast<int64_t>(gridDim.x) * 256) mask[rows[i]] = 1;
}
__global__ __launch_bounds__(256) void iota16_kernel(int64_t n, uint32_t *__restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256)
    out[i] = static_cast<uint32_t>(i) * 16u;
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
  // computed input (crgpu_dev_agg_create_fused): value = op2(op1(col, col2), col3), physical ordinals; e_op1 < 0: plain column
  int e_op1 = -1, e_op2 = -1, col2 = -1, col3 = -1;
  // FILTER as a selection: filter_col is then an INT64 column compared with f_cst
  int f_cmp = -1;
  long long f_cst = 0;
  bool dec128 = false;  // SUM over an integer column: the exact 128-bit total goes out as 16 bytes (little endian)
  int lw0 = -1, lw1 = -1, lseen = -1;
  bool has_w0 = false, has_w1 = false, has_seen = false;
  DevBuf w0, w1, seen;
  // BYTES value store for ACC_ANY_ROW and ACC_MIN_BYTES / ACC_MAX_BYTES
  DevBuf any_off, any_len, any_buf;
  uint64_t any_tail = 0;
  bool uses_any_store() const { return (kind == ACC_ANY_ROW && in_type == CRGPU_BYTES) || is_minmax_bytes(kind); }
};

// Partially ordered input (AggregatorSpec.OrderedGroupCols; hash_aggregator_tmpl.go:160-169, 256-306): see the section
// at the end of this file.
struct OrderedPiece {
  std::vector<OutCol> cols;
  int64_t n = 0;
};
struct OrderedState {
  crgpu_agg *inner = nullptr;   // aggregates (the caller's columns..., row number) with MIN(row number) behind the caller's aggregations
  std::vector<int> ordered_cols;  // input ordinals of the ordered grouping columns
  int n_cols = 0, n_aggs = 0;
  int64_t rows_since_flush = 0;   // row numbers restart at every flush
  bool has_prev = false;
  int prev_slot = 0;
  DevBuf hashes, rowid, prev_hash, last_b, flags, pos, perm, scratch;
  std::vector<DevBuf> shifted;    // validity bitmaps of a sub-batch that does not start at a byte boundary
  std::deque<std::unique_ptr<OrderedPiece>> ready;
  std::unique_ptr<OrderedPiece> handed;  // the piece the caller holds (valid until the next call on the handle)
  std::vector<crgpu_dcol> views;
};

struct crgpu_agg : crgpu::Handle {
  std::unique_ptr<OrderedState> ord;  // set: this handle only routes to ord->inner
  std::vector<int> types, group_cols;
  std::vector<AggOpHost> ops;
  bool need_any_fix = false;
  bool need_minmax_bytes = false;
  int n_phys = 0;            // columns consume() receives; the spec's further ordinals are computed columns
  bool has_computed = false;
  bool bytes_keys = false;
  int lds_words = 0, lds_seen = 0;  // planes of the LDS partial table
  int acc_words = 0;                // HBM atomics per row on the fallback path (cost model)

  // hash index
  uint32_t cap = 0;
  DevBuf meta;
  // per-group arrays
  int64_t ngroups = 0;
  size_t gcap = 0;
  DevBuf gword;
  PackSpec pack;                    // packed-key layout (enabled = 0: keys do not fit 128 bits)
  struct KS { DevBuf data, present, b_off, b_len, b_buf; uint64_t tail = 0; };
  std::vector<KS> ks;

  unsigned long long seq = 0;
  double new_ratio = 1.0;  // new groups per input row in the previous chunk (1: unknown)
  bool descriptors_dirty = true;

  KeyStore store_host;
  DevBuf ops_dev, tails_dev, any_tails_dev, any_ptrs_dev;
  DevBuf counters, out_gid, list_a, list_b, rep_pos, rep_slot, rep_word, scratch, err_dev;
  DevBuf lens_tmp, offs64_tmp, present_tmp;
  // bulk aggregation of a high-cardinality first input (radix.hip buffers); the index is rebuilt lazily afterwards
  RadixTable radix;
  bool index_valid = true;
  bool dense16 = false;  // every group came from the wide bulk path with a 16-byte key: group g's key is b_buf[16 g, 16 g + 16)
  bool try_voided = false;      // a piece of this input voided its all-distinct attempt: the later pieces (same keys' distribution) skip theirs
  bool one_row_groups = false;  // every group is ONE input row (the all-distinct bulk attempt succeeded and nothing followed): an
                                // integer sum cannot have left the int64 range, finish skips the pass that checks
  DevBuf bulk_count;
  // gid-partitioned accumulation
  DevBuf p_gid_res, p_hist, p_offsets, p_gid, p_flags, p_wgs, p_starts;
  DevBuf p_cols[kMaxPartCols];
  std::vector<OutCol> out;
  std::vector<crgpu_dcol> views;
  std::vector<AggOpDev> ops_stage;  // host staging of upload_descriptors
  // DISTINCT aggregations (aggregators_util.go:298-373: a row contributes iff its (group, input) tuple was not seen
  // before): one unordered-DISTINCT table per (input column, FILTER) over (group columns..., input, row number); the
  // rows it lets through become a BOOL column that is the aggregation's FILTER -- a hidden input column behind the
  // physical ones
  struct DistinctSet {
    int col = -1, filter_col = -1, hidden = -1;
    crgpu_distinct *h = nullptr;
    crgpu_deselector *desel = nullptr;
    DevBuf mask;
  };
  std::vector<std::unique_ptr<DistinctSet>> dsets;
  DevBuf d_rowid;
  ~crgpu_agg() {
    if (ord && ord->inner != nullptr) crgpu_dev_agg_destroy(ord->inner);
    for (auto &d : dsets) {
      if (d->h != nullptr) crgpu_dev_distinct_destroy(d->h);
      if (d->desel != nullptr) crgpu_dev_deselector_destroy(d->desel);
    }
  }
  std::vector<void *> ptrs_stage;
};

// upload device descriptors (ops + key store) for the current buffers and the given chunk columns
static crgpu_status upload_descriptors(crgpu_agg *a, const std::vector<DCol> *chunk) {
  // (the staging vectors live in the handle, so the copies need not be waited for: they are rewritten by the next
  // upload only, which every caller reaches after having waited for the stream)
  std::vector<AggOpDev> &ops = a->ops_stage;
  ops.assign(a->ops.size(), AggOpDev());
  for (size_t f = 0; f < a->ops.size(); f++) {
    const AggOpHost &h = a->ops[f];
    AggOpDev &d = ops[f];
    std::memset(&d, 0, sizeof(d));
    d.kind = h.kind;
    d.has_filter = h.filter_col < 0 ? 0 : (h.f_cmp >= 0 ? 2 : 1);
    d.f_cmp = h.f_cmp;
    d.f_cst = h.f_cst;
    d.e_op1 = h.e_op1;
    d.e_op2 = h.e_op2;
    d.key_idx = h.key_idx;
    d.lw0 = h.lw0;
    d.lw1 = h.lw1;
    d.lseen = h.lseen;
    if (chunk != nullptr) {
      if (h.col >= 0) d.col = (*chunk)[h.col];
      if (h.col2 >= 0) d.col2 = (*chunk)[h.col2];
      if (h.col3 >= 0) d.col3 = (*chunk)[h.col3];
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
  KeyStore &st = a->store_host;
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
  // pointer table of the BYTES ANY_NOT_NULL value stores: [off | len | buf] x naggs
  const size_t nops = a->ops.size();
  std::vector<void *> &ptrs = a->ptrs_stage;
  ptrs.assign(nops * 3, nullptr);
  for (size_t f = 0; f < nops; f++) {
    ptrs[f] = a->ops[f].any_off.p;
    ptrs[nops + f] = a->ops[f].any_len.p;
    ptrs[2 * nops + f] = a->ops[f].any_buf.p;
  }
  CRGPU_TRY(a->any_ptrs_dev.ensure(ptrs.size() * 8));
  CRGPU_HIP(hipMemcpyAsync(a->any_ptrs_dev.p, ptrs.data(), ptrs.size() * 8, hipMemcpyHostToDevice, stream()));
  a->descriptors_dirty = false;
  return CRGPU_OK;
}

// per-group arrays: capacity for `need` groups (contents kept; new entries are initialised by the fixup)
static crgpu_status ensure_groups(crgpu_agg *a, size_t need) {
  if (need <= a->gcap) return CRGPU_OK;
  size_t ncap = a->gcap ? a->gcap : 1024;
  while (ncap < need) ncap *= 2;
  const bool keep = a->ngroups > 0;  // an empty table has nothing to carry over: no copy, no wait per array
  CRGPU_TRY(a->gword.ensure(ncap * 8, keep));
  for (auto &op : a->ops) {
    if (op.has_w0) CRGPU_TRY(op.w0.ensure(ncap * 8, keep));
    if (op.has_w1) CRGPU_TRY(op.w1.ensure(ncap * 8, keep));
    if (op.has_seen) CRGPU_TRY(op.seen.ensure(ncap, keep));
    if (op.uses_any_store()) {
      CRGPU_TRY(op.any_off.ensure(ncap * 8, keep));
      CRGPU_TRY(op.any_len.ensure(ncap * 4, keep));
    }
  }
  for (size_t k = 0; k < a->ks.size(); k++) {
    const int t = a->types[a->group_cols[k]];
    CRGPU_TRY(a->ks[k].present.ensure(ncap, keep));
    if (t == CRGPU_BYTES) {
      CRGPU_TRY(a->ks[k].b_off.ensure(ncap * 8, keep));
      CRGPU_TRY(a->ks[k].b_len.ensure(ncap * 4, keep));
    } else {
      CRGPU_TRY(a->ks[k].data.ensure(ncap * type_width(t), keep));
    }
  }
  a->gcap = ncap;
  a->descriptors_dirty = true;
  return CRGPU_OK;
}

// (re)build the index with `new_cap` entries from the per-group hash words
static crgpu_status reindex(crgpu_agg *a, uint32_t new_cap) {
  DevBuf fresh;
  CRGPU_TRY(fresh.ensure(static_cast<size_t>(new_cap) * 8));
  CRGPU_HIP(hipMemsetAsync(fresh.p, 0xFF, static_cast<size_t>(new_cap) * 8, stream()));
  if (a->ngroups > 0) {
    ProfScope prof("agg_reindex");
    hipLaunchKernelGGL(agg_reindex_kernel, dim3(grid_for(a->ngroups, 256)), dim3(256), 0, stream(),
                       a->gword.as<unsigned long long>(), static_cast<uint32_t>(a->ngroups),
                       fresh.as<unsigned long long>(), new_cap - 1);
    CRGPU_HIP(hipGetLastError());
  }
  CRGPU_HIP(hipStreamSynchronize(stream()));
  std::swap(a->meta.p, fresh.p);
  std::swap(a->meta.cap, fresh.cap);
  std::swap(a->meta.device, fresh.device);
  a->cap = new_cap;
  return CRGPU_OK;
}

// Packed-key layout: first-fit the value fields (and, when they fit too, one NULL bit per column)
// into two 64-bit words.  BYTES columns get {length:4, B bytes}, B as large as fits (at most 7).
static PackSpec plan_pack(const std::vector<int> &types, const std::vector<int> &group_cols) {
  PackSpec best;
  std::memset(&best, 0, sizeof(best));
  const int n = static_cast<int>(group_cols.size());
  if (n == 0) return best;
  for (int with_nulls = 1; with_nulls >= 0; with_nulls--) {
    for (int bbytes = 7; bbytes >= 1; bbytes--) {
      PackSpec ps;
      std::memset(&ps, 0, sizeof(ps));
      int used[2] = {0, 0};
      bool ok = true, any_bytes = false;
      auto place = [&](int bits, int8_t &word, int8_t &shift) {
        for (int w = 0; w < 2; w++)
          if (used[w] + bits <= 64) {
            word = static_cast<int8_t>(w);
            shift = static_cast<int8_t>(used[w]);
            used[w] += bits;
            return true;
          }
        return false;
      };
      for (int c = 0; c < n && ok; c++) {
        const int t = types[group_cols[c]];
        int bits = 64;
        if (t == CRGPU_BOOL) bits = 1;
        else if (t == CRGPU_INT16) bits = 16;
        else if (t == CRGPU_INT32) bits = 32;
        else if (t == CRGPU_BYTES) { bits = 4 + 8 * bbytes; any_bytes = true; }
        ps.c[c].bbytes = static_cast<int8_t>(t == CRGPU_BYTES ? bbytes : 0);
        ps.c[c].null_word = -1;
        ok = place(bits, ps.c[c].word, ps.c[c].shift);
      }
      for (int c = 0; c < n && ok && with_nulls; c++) ok = place(1, ps.c[c].null_word, ps.c[c].null_shift);
      if (ok) {
        ps.enabled = 1;
        ps.two_words = used[1] > 0;
        return ps;
      }
      if (!any_bytes) break;  // the BYTES budget is the only thing the inner loop varies
    }
  }
  return best;
}

static crgpu_status agg_create_impl(const crgpu_agg_spec *spec, int32_t n_virtual, const crgpu_agg_vcol *vcols, crgpu_agg **out);
static crgpu_status ordered_create(const crgpu_agg_spec *spec, crgpu_agg **out);
extern "C" crgpu_status crgpu_dev_agg_create(const crgpu_agg_spec *spec, crgpu_agg **out) {
  if (spec != nullptr && spec->n_ordered != 0) return ordered_create(spec, out);
  return agg_create_impl(spec, 0, nullptr, out);
}
extern "C" crgpu_status crgpu_dev_agg_create_fused(const crgpu_agg_spec *spec, int32_t n_virtual, const crgpu_agg_vcol *vcols,
                                                   crgpu_agg **out) {
  if (n_virtual < 0 || (n_virtual > 0 && vcols == nullptr)) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create_fused: null computed columns");
  if (spec != nullptr && spec->n_ordered != 0) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create_fused: ordered grouping columns with computed columns");
  return agg_create_impl(spec, n_virtual, vcols, out);
}

// A computed FLOAT64 column resolved to physical operands: value = op2(op1(x, y), z).
struct ResolvedExpr {
  int op1 = -1, op2 = -1, x = -1, y = -1, z = -1;
};
static crgpu_status resolve_vcol(const crgpu_agg_spec *spec, int n_phys, const crgpu_agg_vcol *vcols, int ord, ResolvedExpr *r) {
  const crgpu_agg_vcol &v = vcols[ord - n_phys];
  if (v.kind != CRGPU_VCOL_PROJECT_F64) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create_fused: aggregation input must be a projection");
  if (v.op < EX_PLUS || v.op > EX_MULT_ONE_PLUS) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create_fused: unknown projection operator");
  if (v.b < 0 || v.b >= n_phys || spec->types[v.b] != CRGPU_FLOAT64)
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create_fused: the right operand of a projection must be a FLOAT64 input column");
  if (v.a < 0 || v.a >= ord) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create_fused: operand ordinal out of range (a computed column may use earlier ones)");
  if (v.a < n_phys) {
    if (spec->types[v.a] != CRGPU_FLOAT64) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create_fused: FLOAT64 operands only");
    r->op1 = v.op, r->x = v.a, r->y = v.b;
    return CRGPU_OK;
  }
  ResolvedExpr inner;
  CRGPU_TRY(resolve_vcol(spec, n_phys, vcols, v.a, &inner));
  if (inner.op2 >= 0) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create_fused: at most two chained projections per aggregation input");
  r->op1 = inner.op1, r->x = inner.x, r->y = inner.y, r->op2 = v.op, r->z = v.b;
  return CRGPU_OK;
}

static crgpu_status agg_create_impl(const crgpu_agg_spec *spec, int32_t n_virtual, const crgpu_agg_vcol *vcols, crgpu_agg **out) {
  CRGPU_TRY(ensure_init());
  if (spec == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: null argument");
  if (n_virtual > spec->n_cols) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create_fused: more computed columns than columns");
  const int n_phys = spec->n_cols - n_virtual;
  if (spec->n_cols < 0 || spec->n_cols > kMaxCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: column count out of range");
  if (spec->n_group < 0 || spec->n_group > kMaxKeyCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: at most 8 grouping columns");
  if (spec->n_aggs < 1 || spec->n_aggs > kMaxAggs) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: between 1 and 64 aggregations");
  std::unique_ptr<crgpu_agg> a(new crgpu_agg());
  {
    crgpu_status _os = a->open();
    if (_os != CRGPU_OK) {  return _os; }
  }
  CRGPU_ENTER(a.get());
  for (int i = 0; i < spec->n_cols; i++) {
    if (!valid_type(spec->types[i])) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: unsupported column type");
    a->types.push_back(spec->types[i]);
  }
  a->n_phys = n_phys;
  a->has_computed = n_virtual > 0;
  for (int i = 0; i < n_virtual; i++) {
    const int want = vcols[i].kind == CRGPU_VCOL_SELECT_I64_CONST ? CRGPU_BOOL : CRGPU_FLOAT64;
    if (vcols[i].kind != CRGPU_VCOL_SELECT_I64_CONST && vcols[i].kind != CRGPU_VCOL_PROJECT_F64)
      return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create_fused: unknown computed column kind");
    if (a->types[n_phys + i] != want) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create_fused: a selection is a BOOL column, a projection a FLOAT64 column");
  }
  for (int i = 0; i < spec->n_group; i++) {
    if (spec->group_cols[i] >= n_phys && spec->group_cols[i] < spec->n_cols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create_fused: grouping by a computed column");
    if (spec->group_cols[i] < 0 || spec->group_cols[i] >= spec->n_cols) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: grouping column ordinal out of range");
    a->group_cols.push_back(spec->group_cols[i]);
    if (a->types[spec->group_cols[i]] == CRGPU_BYTES) a->bytes_keys = true;
  }
  a->ks = std::vector<crgpu_agg::KS>(spec->n_group);
  a->pack = plan_pack(a->types, a->group_cols);
  a->ops = std::vector<AggOpHost>(spec->n_aggs);
  for (int f = 0; f < spec->n_aggs; f++) {
    const crgpu_agg_desc &d = spec->aggs[f];
    AggOpHost &op = a->ops[f];
    op.func = d.func;
    op.col = d.col;
    op.filter_col = d.filter_col;
    if (d.distinct && (n_virtual > 0 || d.func == CRGPU_COUNT_ROWS))
      return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: DISTINCT with computed columns / on COUNT_ROWS");
    if (d.filter_col >= 0 && (d.filter_col >= spec->n_cols || a->types[d.filter_col] != CRGPU_BOOL))
      return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: FILTER column must be a BOOL column");
    if (d.func != CRGPU_COUNT_ROWS && (d.col < 0 || d.col >= spec->n_cols))
      return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: aggregation column ordinal out of range");
    const int t = d.func == CRGPU_COUNT_ROWS ? -1 : a->types[d.col];
    op.in_type = t;
    if (d.filter_col >= n_phys) {  // FILTER = a computed selection: test the INT64 column itself
      const crgpu_agg_vcol &v = vcols[d.filter_col - n_phys];
      if (v.kind != CRGPU_VCOL_SELECT_I64_CONST) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create_fused: FILTER must be a selection");
      if (v.op < SEL_EQ || v.op > SEL_GE) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create_fused: unknown comparison");
      if (v.a < 0 || v.a >= n_phys || a->types[v.a] != CRGPU_INT64) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create_fused: selections compare an INT64 input column with a constant");
      op.filter_col = v.a;
      op.f_cmp = v.op;
      op.f_cst = v.cst;
    }
    if (d.func != CRGPU_COUNT_ROWS && d.col >= n_phys) {  // computed input
      if (d.func != CRGPU_SUM && d.func != CRGPU_AVG && d.func != CRGPU_MIN && d.func != CRGPU_MAX && d.func != CRGPU_COUNT)
        return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create_fused: SUM / AVG / MIN / MAX / COUNT over computed columns");
      ResolvedExpr r;
      CRGPU_TRY(resolve_vcol(spec, n_phys, vcols, d.col, &r));
      op.col = r.x, op.col2 = r.y, op.col3 = r.z, op.e_op1 = r.op1, op.e_op2 = r.op2;
    }
    switch (d.func) {
      case CRGPU_COUNT_ROWS: op.kind = ACC_COUNT_ROWS; op.out_type = CRGPU_INT64; op.col = -1; break;
      case CRGPU_COUNT: op.kind = ACC_COUNT; op.out_type = CRGPU_INT64; break;
      case CRGPU_SUM:
        if (is_int_type(t) && d.col < n_phys) {
          // SUM over INT yields DECIMAL in the reference (apd additions, hash_sum_agg.eg.go:30-373).  The total of
          // 64-bit integers is exact in 128 bits: the SUM_INT accumulators without the range check, handed out as a
          // BYTES column of 16-byte little-endian two's-complement values the shell turns into apd.Decimal.
          op.kind = ACC_SUM_I128; op.out_type = CRGPU_BYTES; op.dec128 = true; break;
        }
        if (t != CRGPU_FLOAT64) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: SUM over this type: CPU operator");
        op.kind = ACC_SUM_F64; op.out_type = CRGPU_FLOAT64; break;
      case CRGPU_AVG:
        if (t != CRGPU_FLOAT64) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: AVG over INT yields DECIMAL (apd): ask for SUM and COUNT and divide in the shell, or keep the CPU operator");
        op.kind = ACC_AVG_F64; op.out_type = CRGPU_FLOAT64; break;
      case CRGPU_SUM_INT:
        if (!is_int_type(t)) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: SUM_INT needs an integer column");
        op.kind = ACC_SUM_I128; op.out_type = CRGPU_INT64; break;
      case CRGPU_MIN:
      case CRGPU_MAX:
        if (t == CRGPU_BYTES) {
          op.kind = d.func == CRGPU_MIN ? ACC_MIN_BYTES : ACC_MAX_BYTES;
          a->need_any_fix = true;
          a->need_minmax_bytes = true;
        } else if (t == CRGPU_FLOAT64) op.kind = d.func == CRGPU_MIN ? ACC_MIN_F64 : ACC_MAX_F64;
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
        if (op.kind == ACC_ANY_ROW) a->need_any_fix = true;
        break;
      }
      default: return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: aggregate function outside the GPU path");
    }
    // accumulator words: HBM arrays and planes of the LDS partial table
    op.has_w0 = op.kind != ACC_ANY_KEY;
    op.has_w1 = op.kind == ACC_AVG_F64 || op.kind == ACC_SUM_I128 || op.kind == ACC_ANY_ROW || is_minmax_bytes(op.kind);
    op.has_seen = op.kind == ACC_SUM_F64 || op.kind == ACC_SUM_I128 || (op.kind >= ACC_MIN_I64 && op.kind <= ACC_BOOL_OR) || is_minmax_bytes(op.kind);
    if (op.has_w0) {
      op.lw0 = a->lds_words++;
      a->acc_words++;
    }
    if (op.kind == ACC_AVG_F64 || op.kind == ACC_SUM_I128) {
      op.lw1 = a->lds_words++;
      if (op.kind == ACC_AVG_F64) a->acc_words++;
    }
    if (op.has_seen) op.lseen = a->lds_seen++;
  }
  // DISTINCT: COUNT / SUM / SUM_INT / AVG only see the first row of every (group, input) tuple (it changes nothing for
  // MIN / MAX / ANY_NOT_NULL / BOOL_AND / BOOL_OR); aggregations over the same (input, FILTER) share one table
  for (int f = 0; f < spec->n_aggs; f++) {
    const crgpu_agg_desc &d = spec->aggs[f];
    if (!d.distinct || !(d.func == CRGPU_COUNT || d.func == CRGPU_SUM || d.func == CRGPU_SUM_INT || d.func == CRGPU_AVG)) continue;
    crgpu_agg::DistinctSet *ds = nullptr;
    for (auto &x : a->dsets)
      if (x->col == d.col && x->filter_col == d.filter_col) ds = x.get();
    if (ds == nullptr) {
      a->dsets.emplace_back(new crgpu_agg::DistinctSet());
      ds = a->dsets.back().get();
      ds->col = d.col;
      ds->filter_col = d.filter_col;
      ds->hidden = static_cast<int>(a->types.size());
      a->types.push_back(CRGPU_BOOL);
      std::vector<int32_t> dt, dc;
      for (int g : a->group_cols) dt.push_back(a->types[g]);
      dt.push_back(a->types[d.col]);
      for (size_t i = 0; i < dt.size(); i++) dc.push_back(static_cast<int32_t>(i));
      dt.push_back(CRGPU_INT64);  // the row number rides along
      CRGPU_TRY(crgpu_dev_distinct_create(dt.data(), static_cast<int32_t>(dt.size()), dc.data(), static_cast<int32_t>(dc.size()), 0, 0, &ds->h));
      if (d.filter_col >= 0) CRGPU_TRY(crgpu_dev_deselector_create(dt.data(), static_cast<int32_t>(dt.size()), &ds->desel));
    }
    a->ops[f].filter_col = ds->hidden;  // (the user's FILTER is applied before the DISTINCT test: it is part of the mask)
  }
  a->out = std::vector<OutCol>(spec->n_aggs);
  a->views.resize(spec->n_aggs);
  CRGPU_TRY(a->counters.ensure(16));
  CRGPU_TRY(a->err_dev.ensure(4));
  CRGPU_TRY(a->tails_dev.ensure(kMaxKeyCols * 8));
  CRGPU_TRY(a->any_tails_dev.ensure(static_cast<size_t>(spec->n_aggs) * 8));
  CRGPU_HIP(hipMemsetAsync(a->tails_dev.p, 0, kMaxKeyCols * 8, stream()));
  CRGPU_HIP(hipMemsetAsync(a->any_tails_dev.p, 0, static_cast<size_t>(spec->n_aggs) * 8, stream()));
  CRGPU_TRY(ensure_groups(a.get(), 1024));
  CRGPU_TRY(reindex(a.get(), 256));  // hashAggregator starts with 256 buckets too (hash_aggregator.go:217)
  static DeviceOnce lds_attr_set;
  if (lds_attr_set.need()) {
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_accum_kernel<true, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsMax)));
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_accum_kernel<true, true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsMax)));
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_accum_kernel<false, true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsMax)));
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_accum_kernel<true, true, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsMax)));
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_accum_kernel<true, true, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsMax)));
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_accum_kernel<false, true, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsMax)));
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_bulk_kernel<false, 1024, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsMax - 64)));
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_bulk_kernel<true, 1024, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsMax - 64)));
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_part_accum_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kLdsMax)));
    lds_attr_set.mark();
  }
  *out = a.release();
  return CRGPU_OK;
}

extern "C" void crgpu_dev_agg_destroy(crgpu_agg *a) {
  if (a == nullptr) return;
  if (a->st != nullptr) (void)hipStreamSynchronize(a->st);  // its buffers return to the pool: no kernel may still use them
  delete a;
}

static crgpu_status ordered_reset(crgpu_agg *a);
extern "C" crgpu_status crgpu_dev_agg_reset(crgpu_agg *a) {
  if (a == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_reset: null handle");
  CRGPU_ENTER(a);
  a->cancelled = 0;
  if (a->ord) return ordered_reset(a);
  a->ngroups = 0;
  a->seq = 0;
  a->new_ratio = 1.0;
  a->index_valid = true;
  a->dense16 = false;
  a->one_row_groups = false;
  for (auto &k : a->ks) k.tail = 0;
  for (auto &o : a->ops) o.any_tail = 0;
  for (auto &d : a->dsets) CRGPU_TRY(crgpu_dev_distinct_reset(d->h));
  CRGPU_HIP(hipMemsetAsync(a->tails_dev.p, 0, kMaxKeyCols * 8, stream()));
  CRGPU_HIP(hipMemsetAsync(a->any_tails_dev.p, 0, a->ops.size() * 8, stream()));
  CRGPU_HIP(hipMemsetAsync(a->meta.p, 0xFF, static_cast<size_t>(a->cap) * 8, stream()));
  return CRGPU_OK;
}

// ---------------------------------------------------------------------- accumulation passes
struct AccPlan {
  bool lds = false;
  uint32_t G = 0, R = 1, sweeps = 1;
  int block = 256;
  int wgs_per_cu = 8;
  size_t lds_bytes = 0;
  bool pk = false;   // fused pass: LDS key table over the packed keys
  uint32_t C = 0;    // its capacity
  bool part = false; // regroup the rows by gid range first, then LDS partial tables per range
  uint32_t shift = 0, P = 0;
};
// How to accumulate `m` rows into `ngroups` groups: LDS partial tables (one sweep, or a few gid
// ranges while that stays cheaper than `acc_words` HBM atomics per row), else HBM atomics.
static AccPlan plan_accumulate(const crgpu_agg *a, int64_t ngroups, int64_t m, bool fused, bool fused_only_lds = false) {
  AccPlan p;
  const size_t bpg = static_cast<size_t>(a->lds_words) * 8 + static_cast<size_t>(a->lds_seen);
  if (bpg == 0 || ngroups <= 0) return p;
  const size_t total = bpg * static_cast<size_t>(ngroups + 1);  // + the trash group rows that do not contribute update
  if (total <= kLdsLarge) {
    p.lds = true;
    p.G = static_cast<uint32_t>(ngroups);
    size_t table = 0;
    if (fused && a->pack.enabled) {
      // at most half full; small tables are kept sparser (<= 1/4, >= 64 slots) so that the first probe decides
      uint64_t want_c = static_cast<uint64_t>(ngroups) * 2;
      if (ngroups <= 2048) want_c = static_cast<uint64_t>(ngroups) * 4 < 64 ? 64 : static_cast<uint64_t>(ngroups) * 4;
      const uint32_t C = static_cast<uint32_t>(next_pow2(want_c));
      const size_t bytes = static_cast<size_t>(C) * (a->pack.two_words ? 20 : 12) + 8;
      if (total + bytes <= kLdsMax - 1024) {
        p.pk = true;
        p.C = C;
        table = bytes;
      }
    }
    uint32_t r = 1;  // replicas: spread the lanes of a wave that hit one group, while 8 workgroups still fit a CU
    while (r < kMaxReplicas && total * (r * 2) + table <= kLdsReplicated) r *= 2;
    p.R = r;
    p.lds_bytes = total * r + table;
    const size_t fit = (160 * 1024) / (p.lds_bytes ? p.lds_bytes : 1);
    p.wgs_per_cu = fit >= 8 ? 8 : fit >= 4 ? 4 : fit >= 2 ? 2 : 1;
    p.block = p.wgs_per_cu == 8 ? 256 : p.wgs_per_cu == 1 ? 1024 : 512;  // 16 waves per CU whenever LDS allows
    return p;
  }
  // gid-partitioned: ranges of 2^shift groups, each with a partial table of at most kLdsLarge bytes
  {
    uint32_t shift = 0;
    while ((bpg << (shift + 1)) <= kLdsLarge) shift++;
    const uint64_t P = (static_cast<uint64_t>(ngroups) + (1ull << shift) - 1) >> shift;
    int nops = 0;
    std::vector<int> cols;
    for (const AggOpHost &h : a->ops) {
      if (h.kind == ACC_ANY_KEY) continue;
      nops++;
      if (h.kind != ACC_COUNT_ROWS && h.in_type != CRGPU_BYTES && std::find(cols.begin(), cols.end(), h.col) == cols.end())
        cols.push_back(h.col);
    }
    // moving a row costs ~(8 + 8 per value column) bytes written and read again; an HBM atomic costs 1/24 ns
    const double move_cost = (16.0 + 16.0 * static_cast<double>(cols.size())) / 3e12;
    const double atomic_cost = static_cast<double>(a->acc_words > 0 ? a->acc_words : 1) / 24e9;
    if (!fused_only_lds && !a->need_any_fix && !a->has_computed && m >= (int64_t(1) << 18) && P <= kMaxParts && nops <= kMaxOpsPerLaunch &&
        static_cast<int>(cols.size()) <= kMaxPartCols && move_cost < atomic_cost) {
      p.part = true;
      p.shift = shift;
      p.P = static_cast<uint32_t>(P);
      p.block = 512;
      p.lds_bytes = bpg * ((size_t(1) << shift) + 1);
    }
  }
  return p;
}

struct PassInput {
  const std::vector<DCol> *cols;
  KeyCols keys;
  const uint32_t *list;
  int64_t m;
};

static crgpu_status launch_accumulate(crgpu_agg *a, const PassInput &in, bool fused, uint32_t *miss_list, const AccPlan &plan) {
  AccArgs args;
  std::memset(&args, 0, sizeof(args));
  args.keys = in.keys;
  args.store = a->store_host;
  args.m = in.m;
  args.list = in.list;
  args.out_gid = a->out_gid.as<uint32_t>();
  args.meta = a->meta.as<unsigned long long>();
  args.mask = a->cap - 1;
  args.R = 1;
  args.seq_base = a->seq;
  args.miss_list = miss_list;
  args.counters = a->counters.as<unsigned int>();
  args.record_misses = fused ? 1 : 0;
  args.pack = a->pack;
  args.C = plan.pk ? plan.C : 0;
  hipStream_t st = stream();
  dim3 grid, block;
  size_t lds = 0;
  if (!plan.lds) {
    grid = dim3(grid_for((in.m + kAccRows - 1) / kAccRows, 256));
    block = dim3(256);
  } else {
    args.R = plan.R;
    args.nwords = static_cast<uint32_t>(a->lds_words);
    args.nseen = static_cast<uint32_t>(a->lds_seen);
    // persistent workgroups: every one initialises and merges a whole partial table
    const int64_t per_wg = static_cast<int64_t>(plan.block) * kAccRows * 2;
    int64_t want = (in.m + per_wg - 1) / per_wg;
    if (want < 1) want = 1;
    const int64_t max_grid = static_cast<int64_t>(num_cus()) * plan.wgs_per_cu;
    grid = dim3(static_cast<unsigned int>(want < max_grid ? want : max_grid));
    block = dim3(plan.block);
    lds = (plan.lds_bytes + 15) & ~size_t(15);
  }
  // the op descriptors travel in the kernel arguments, kMaxOpsPerLaunch at a time (one launch in practice)
  std::vector<OpK> ops;
  // a FILTER that every aggregation of the pass shares is tested once per row, not once per aggregation
  bool common_filter = true;
  int cf_col = -2, cf_cmp = -2;
  long long cf_cst = 0;
  for (const AggOpHost &h : a->ops) {
    if (h.kind == ACC_ANY_KEY) continue;
    if (cf_col == -2) cf_col = h.filter_col, cf_cmp = h.f_cmp, cf_cst = h.f_cst;
    common_filter = common_filter && h.filter_col == cf_col && h.f_cmp == cf_cmp && h.f_cst == cf_cst;
  }
  common_filter = common_filter && cf_col >= 0;
  int nvex = 0, nvsel = 0;
  if (common_filter) {
    const DCol &fc = (*in.cols)[cf_col];
    if (cf_cmp >= 0) {
      args.cf_kind = 2;
      args.vsel[0].data = static_cast<const long long *>(fc.data);
      args.vsel[0].valid = fc.valid;
      args.vsel[0].cst = cf_cst;
      args.vsel[0].cmp = cf_cmp;
      nvsel = 1;
    } else {
      args.cf_kind = 1;
      args.cf_data = static_cast<const uint8_t *>(fc.data);
      args.cf_valid = fc.valid;
    }
  }
  for (const AggOpHost &h : a->ops) {
    if (h.kind == ACC_ANY_KEY) continue;
    OpK k;
    std::memset(&k, 0, sizeof(k));
    k.kind = static_cast<int8_t>(h.kind);
    k.in_type = static_cast<int8_t>(h.in_type);
    k.lw0 = static_cast<int8_t>(h.lw0);
    k.lw1 = static_cast<int8_t>(h.lw1);
    k.lseen = static_cast<int8_t>(h.lseen);
    if (h.col >= 0) {
      k.data = (*in.cols)[h.col].data;
      k.valid = (*in.cols)[h.col].valid;
    }
    if (h.e_op1 >= 0) {  // computed input: one table entry per distinct expression
      VExpr e;
      std::memset(&e, 0, sizeof(e));
      e.x = static_cast<const double *>((*in.cols)[h.col].data);
      e.vx = (*in.cols)[h.col].valid;
      e.y = static_cast<const double *>((*in.cols)[h.col2].data);
      e.vy = (*in.cols)[h.col2].valid;
      e.op1 = static_cast<int8_t>(h.e_op1);
      e.op2 = static_cast<int8_t>(h.e_op2);
      if (h.e_op2 >= 0) {
        e.z = static_cast<const double *>((*in.cols)[h.col3].data);
        e.vz = (*in.cols)[h.col3].valid;
      }
      int at = -1;
      for (int i = 0; i < nvex && at < 0; i++)
        if (std::memcmp(&args.vex[i], &e, sizeof(e)) == 0) at = i;
      if (at < 0) {
        if (nvex == kMaxVex) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg: more than 8 distinct computed inputs");
        at = nvex;
        args.vex[nvex++] = e;
      }
      k.vex = static_cast<int8_t>(at + 1);
    }
    if (h.filter_col >= 0 && !common_filter) {
      const DCol &fc = (*in.cols)[h.filter_col];
      if (h.f_cmp >= 0) {
        VSel sl;
        std::memset(&sl, 0, sizeof(sl));
        sl.data = static_cast<const long long *>(fc.data);
        sl.valid = fc.valid;
        sl.cst = h.f_cst;
        sl.cmp = h.f_cmp;
        int at = -1;
        for (int i = 0; i < nvsel && at < 0; i++)
          if (std::memcmp(&args.vsel[i], &sl, sizeof(sl)) == 0) at = i;
        if (at < 0) {
          if (nvsel == kMaxVsel) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg: more than 4 distinct selections as FILTER");
          at = nvsel;
          args.vsel[nvsel++] = sl;
        }
        k.has_filter = 2;
        k.fsel = static_cast<int8_t>(at + 1);
      } else {
        k.has_filter = 1;
        k.fdata = static_cast<const uint8_t *>(fc.data);
        k.fvalid = fc.valid;
      }
    }
    k.w0 = h.w0.as<unsigned long long>();
    k.w1 = h.w1.as<unsigned long long>();
    k.seen = h.seen.as<uint8_t>();
    ops.push_back(k);
  }
  const size_t nbatches = ops.empty() ? 1 : (ops.size() + kMaxOpsPerLaunch - 1) / kMaxOpsPerLaunch;
  for (size_t b = 0; b < nbatches; b++) {
    const size_t o0 = b * kMaxOpsPerLaunch;
    const size_t cnt = ops.size() - o0 < static_cast<size_t>(kMaxOpsPerLaunch) ? ops.size() - o0 : kMaxOpsPerLaunch;
    args.naggs = static_cast<int32_t>(ops.empty() ? 0 : cnt);
    if (args.naggs > 0) std::memcpy(args.ops, ops.data() + o0, sizeof(OpK) * cnt);
    for (uint32_t s = 0; s < plan.sweeps; s++) {
      if (plan.lds) {
        args.gid_lo = s * plan.G;
        const int64_t left = a->ngroups - static_cast<int64_t>(args.gid_lo);
        args.G = plan.sweeps == 1 ? plan.G : static_cast<uint32_t>(left < static_cast<int64_t>(plan.G) ? left : plan.G);
      }
      ProfScope prof(plan.lds ? (fused ? "agg_fused_lds" : "agg_accum_lds") : (fused ? "agg_fused_hbm" : "agg_accum_hbm"));
      const bool vc = args.cf_kind != 0 || nvex > 0 || nvsel > 0;
      if (plan.lds) {
        if (fused && plan.pk && vc) hipLaunchKernelGGL((agg_accum_kernel<true, true, true, true>), grid, block, lds, st, args);
        else if (fused && plan.pk) hipLaunchKernelGGL((agg_accum_kernel<true, true, true>), grid, block, lds, st, args);
        else if (fused && vc) hipLaunchKernelGGL((agg_accum_kernel<true, true, false, true>), grid, block, lds, st, args);
        else if (fused) hipLaunchKernelGGL((agg_accum_kernel<true, true, false>), grid, block, lds, st, args);
        else if (vc) hipLaunchKernelGGL((agg_accum_kernel<false, true, false, true>), grid, block, lds, st, args);
        else hipLaunchKernelGGL((agg_accum_kernel<false, true, false>), grid, block, lds, st, args);
      } else {
        if (fused && vc) hipLaunchKernelGGL((agg_accum_kernel<true, false, false, true>), grid, block, 0, st, args);
        else if (fused) hipLaunchKernelGGL((agg_accum_kernel<true, false, false>), grid, block, 0, st, args);
        else if (vc) hipLaunchKernelGGL((agg_accum_kernel<false, false, false, true>), grid, block, 0, st, args);
        else hipLaunchKernelGGL((agg_accum_kernel<false, false, false>), grid, block, 0, st, args);
      }
      CRGPU_HIP(hipGetLastError());
    }
    args.record_misses = 0;
  }
  return CRGPU_OK;
}

// Regroup the rows of the pass by gid range and accumulate every range in LDS (see the kernels above).
static crgpu_status launch_partitioned(crgpu_agg *a, const PassInput &in, const AccPlan &plan) {
  hipStream_t st = stream();
  const uint32_t P = plan.P;
  int64_t tile_rows = kPartTileMin;
  while (tile_rows < static_cast<int64_t>(P) * 32 && (in.m + tile_rows - 1) / tile_rows > 512) tile_rows *= 2;
  const uint32_t ntiles = static_cast<uint32_t>((in.m + tile_rows - 1) / tile_rows);
  const size_t m = static_cast<size_t>(in.m);
  // aggregation descriptors + the distinct value columns they read
  std::vector<OpK> ops;
  std::vector<int> cols, op_slot;
  for (const AggOpHost &h : a->ops) {
    if (h.kind == ACC_ANY_KEY) continue;
    OpK k;
    std::memset(&k, 0, sizeof(k));
    k.kind = static_cast<int8_t>(h.kind);
    k.in_type = static_cast<int8_t>(h.in_type);
    k.has_filter = h.filter_col >= 0;
    k.lw0 = static_cast<int8_t>(h.lw0);
    k.lw1 = static_cast<int8_t>(h.lw1);
    k.lseen = static_cast<int8_t>(h.lseen);
    if (h.col >= 0) k.valid = (*in.cols)[h.col].valid;
    if (h.filter_col >= 0) {
      k.fdata = static_cast<const uint8_t *>((*in.cols)[h.filter_col].data);
      k.fvalid = (*in.cols)[h.filter_col].valid;
    }
    k.w0 = h.w0.as<unsigned long long>();
    k.w1 = h.w1.as<unsigned long long>();
    k.seen = h.seen.as<uint8_t>();
    int slot = -1;
    if (h.kind != ACC_COUNT_ROWS && h.in_type != CRGPU_BYTES) {
      auto it = std::find(cols.begin(), cols.end(), h.col);
      slot = static_cast<int>(it - cols.begin());
      if (it == cols.end()) cols.push_back(h.col);
    }
    ops.push_back(k);
    op_slot.push_back(slot);
  }
  CRGPU_TRY(a->p_gid_res.ensure(m * 4));
  CRGPU_TRY(a->p_hist.ensure(static_cast<size_t>(P) * ntiles * 4));
  CRGPU_TRY(a->p_offsets.ensure((static_cast<size_t>(P) * ntiles + 1) * 8));
  CRGPU_TRY(a->p_gid.ensure(m * 4));
  CRGPU_TRY(a->p_flags.ensure(m * 4));
  for (size_t c = 0; c < cols.size(); c++) CRGPU_TRY(a->p_cols[c].ensure(m * 8));

  PartArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.m = in.m;
  pa.list = in.list;
  pa.out_gid = a->out_gid.as<uint32_t>();
  pa.meta = a->meta.as<unsigned long long>();
  pa.shift = plan.shift;
  pa.P = P;
  pa.ntiles = ntiles;
  pa.tile_rows = static_cast<int32_t>(tile_rows);
  pa.naggs = static_cast<int32_t>(ops.size());
  pa.ncols = static_cast<int32_t>(cols.size());
  pa.gid_res = a->p_gid_res.as<uint32_t>();
  pa.tile_hist = a->p_hist.as<uint32_t>();
  pa.offsets = a->p_offsets.as<uint64_t>();
  pa.gid_out = a->p_gid.as<uint32_t>();
  pa.flags_out = a->p_flags.as<uint32_t>();
  for (size_t c = 0; c < cols.size(); c++) {
    pa.src[c] = (*in.cols)[cols[c]].data;
    pa.src_type[c] = static_cast<int8_t>((*in.cols)[cols[c]].type);
    pa.dst[c] = a->p_cols[c].as<unsigned long long>();
  }
  if (!ops.empty()) std::memcpy(pa.ops, ops.data(), sizeof(OpK) * ops.size());
  const dim3 grid(static_cast<unsigned int>(ntiles < static_cast<uint32_t>(num_cus()) * 8 ? ntiles : num_cus() * 8));
  // few, large tiles when there are many ranges: 16 waves per workgroup keep the CU busy
  const dim3 pblock(tile_rows >= 65536 ? 1024 : 256);
  uint64_t total = 0;
  {
    ProfScope prof("agg_part_hist");
    hipLaunchKernelGGL(agg_part_hist_kernel, grid, pblock, 0, st, pa);
    CRGPU_HIP(hipGetLastError());
  }
  CRGPU_TRY(exclusive_scan_u32(a->p_hist.as<uint32_t>(), static_cast<int64_t>(P) * ntiles, a->p_offsets.as<uint64_t>(),
                               a->scratch, &total));
  {
    ProfScope prof("agg_part_scatter");
    hipLaunchKernelGGL(agg_part_scatter_kernel, grid, pblock, 0, st, pa);
    CRGPU_HIP(hipGetLastError());
  }
  // partition boundaries: offsets[p * ntiles] for every p, then `total`
  std::vector<uint64_t> starts(static_cast<size_t>(P) + 1, 0);
  CRGPU_TRY(a->p_starts.ensure((static_cast<size_t>(P) + 1) * 8));
  hipLaunchKernelGGL(agg_part_starts_kernel, dim3((P + 256) / 256), dim3(256), 0, st, a->p_offsets.as<uint64_t>(), P, ntiles,
                     a->p_starts.as<uint64_t>());
  CRGPU_HIP(hipGetLastError());
  CRGPU_HIP(hipMemcpyAsync(starts.data(), a->p_starts.p, (static_cast<size_t>(P) + 1) * 8, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  // a few fat workgroups per partition: every one merges a whole partial table at the end
  const uint64_t target_wgs = static_cast<uint64_t>(num_cus()) * 2;
  uint64_t rows_per_wg = (total + target_wgs - 1) / target_wgs;
  if (rows_per_wg < 8192) rows_per_wg = 8192;
  std::vector<PartWg> wgs;
  for (uint32_t q = 0; q < P; q++) {
    const uint64_t b = starts[q], e = starts[q + 1];
    if (e <= b) continue;
    const uint64_t nw = (e - b + rows_per_wg - 1) / rows_per_wg;
    const uint64_t per = (e - b + nw - 1) / nw;
    for (uint64_t w = 0; w < nw; w++) {
      PartWg d;
      d.part = q;
      d.exclusive = nw == 1 ? 1u : 0u;
      d.begin = b + w * per;
      d.end = d.begin + per < e ? d.begin + per : e;
      if (d.begin < d.end) wgs.push_back(d);
    }
  }
  if (wgs.empty()) return CRGPU_OK;
  CRGPU_TRY(a->p_wgs.ensure(wgs.size() * sizeof(PartWg)));
  CRGPU_HIP(hipMemcpyAsync(a->p_wgs.p, wgs.data(), wgs.size() * sizeof(PartWg), hipMemcpyHostToDevice, st));
  PartAccArgs aa;
  std::memset(&aa, 0, sizeof(aa));
  for (size_t f = 0; f < ops.size(); f++) {
    aa.ops[f] = ops[f];
    aa.ops[f].data = op_slot[f] >= 0 ? a->p_cols[op_slot[f]].p : nullptr;
  }
  aa.naggs = static_cast<int32_t>(ops.size());
  aa.shift = plan.shift;
  aa.nwords = static_cast<uint32_t>(a->lds_words);
  aa.nseen = static_cast<uint32_t>(a->lds_seen);
  aa.ngroups = a->ngroups;
  aa.gid = a->p_gid.as<uint32_t>();
  aa.flags = a->p_flags.as<uint32_t>();
  aa.wgs = a->p_wgs.as<PartWg>();
  {
    ProfScope prof("agg_part_accum");
    // one workgroup per CU when the table is large: make it 16 waves
    hipLaunchKernelGGL(agg_part_accum_kernel, dim3(static_cast<unsigned int>(wgs.size())), dim3(plan.lds_bytes > 76 * 1024 ? 1024 : 512),
                       (plan.lds_bytes + 15) & ~size_t(15), st, aa);
    CRGPU_HIP(hipGetLastError());
  }
  CRGPU_HIP(hipStreamSynchronize(st));  // `wgs` is a host temporary
  return CRGPU_OK;
}

// The deduplicated sum-family pass (agg_sumfam_kernel): *done = false when the handle / chunk is outside its form.
template <int NIN>
static void launch_sumfam_n(const SfArgs &args, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  static DeviceOnce attr_set;  // per instantiation
  if (attr_set.need()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_sumfam_kernel<NIN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(kLdsMax));
    attr_set.mark();
  }
  hipLaunchKernelGGL(agg_sumfam_kernel<NIN>, grid, block, lds, st, args);
}
template <int NIN, int NK, bool KB>
static void launch_sf8_nk(const SfArgs &args, int blk, int wgs_per_cu, size_t lds, hipStream_t st) {
  static DeviceOnce attr_set;  // per instantiation
  if (attr_set.need()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_sf8_kernel<NIN, NK, KB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(kLdsMax));
    attr_set.mark();
  }
  const int64_t per_wg = static_cast<int64_t>(blk) * kSf8Rows * 2;
  int64_t want = (args.m + per_wg - 1) / per_wg;
  if (want < 1) want = 1;
  const int64_t max_grid = static_cast<int64_t>(num_cus()) * wgs_per_cu;  // resident workgroups only: each strides over the chunk
  hipLaunchKernelGGL((agg_sf8_kernel<NIN, NK, KB>), dim3(static_cast<unsigned int>(want < max_grid ? want : max_grid)), dim3(blk), lds, st, args);
}
template <int NIN>
static void launch_sf8_n(const SfArgs &args, int nk, bool kb, int blk, int wgs_per_cu, size_t lds, hipStream_t st) {
  if (nk == 1 && kb) launch_sf8_nk<NIN, 1, true>(args, blk, wgs_per_cu, lds, st);
  else if (nk == 1) launch_sf8_nk<NIN, 1, false>(args, blk, wgs_per_cu, lds, st);
  else if (kb) launch_sf8_nk<NIN, 2, true>(args, blk, wgs_per_cu, lds, st);
  else launch_sf8_nk<NIN, 2, false>(args, blk, wgs_per_cu, lds, st);
}
// Slots of agg_sf8_kernel from the deduplicated inputs: plain columns first, each followed by the projections that
// extend it (x, then x op1 y, then (x op1 y) op2 z as consecutive slots; a two-operation input without its one-operation
// prefix among the inputs gets a slot nobody accumulates).  false: more than kSfMaxIn slots.
static bool sf8_slots(SfArgs &args, int nin) {
  SfIn slots[kSfMaxIn];
  int slot_of[kSfMaxIn];
  bool placed[kSfMaxIn] = {};
  int n = 0;
  auto emit = [&](const void *x, bool chain, int op, const double *y, int type, int acc) -> bool {
    if (n == kSfMaxIn) return false;
    SfIn &t = slots[n];
    std::memset(&t, 0, sizeof(t));
    t.x = x;
    t.y = y;
    t.op1 = static_cast<int8_t>(op);
    t.op2 = -1;
    t.type = static_cast<int8_t>(type);
    t.chain = chain ? 1 : 0;
    t.p_sum = t.p_hi = t.p_min = t.p_max = -1;
    if (acc >= 0) {
      const SfIn &e = args.in[acc];
      t.p_sum = e.p_sum, t.p_hi = e.p_hi, t.p_min = e.p_min, t.p_max = e.p_max;
      slot_of[acc] = n;
      placed[acc] = true;
    }
    n++;
    return true;
  };
  auto one_op_over = [&](const void *x) {
    for (int i = 0; i < nin; i++)
      if (!placed[i] && args.in[i].x == x && args.in[i].op1 >= 0 && args.in[i].op2 < 0) return i;
    return -1;
  };
  auto two_op_over = [&](const void *x, int op1, const double *y) {  // op1 < 0: any first operation
    for (int i = 0; i < nin; i++)
      if (!placed[i] && args.in[i].x == x && args.in[i].op2 >= 0 && (op1 < 0 || (args.in[i].op1 == op1 && args.in[i].y == y))) return i;
    return -1;
  };
  auto extend = [&](const void *x, int type) -> bool {  // the previous slot holds the column x
    const int j = one_op_over(x);
    if (j >= 0) {
      if (!emit(nullptr, true, args.in[j].op1, args.in[j].y, type, j)) return false;
      const int k = two_op_over(x, args.in[j].op1, args.in[j].y);
      return k < 0 || emit(nullptr, true, args.in[k].op2, args.in[k].z, type, k);
    }
    const int k = two_op_over(x, -1, nullptr);
    if (k < 0) return true;
    return emit(nullptr, true, args.in[k].op1, args.in[k].y, type, -1) && emit(nullptr, true, args.in[k].op2, args.in[k].z, type, k);
  };
  for (int i = 0; i < nin; i++) {
    if (placed[i] || args.in[i].op1 >= 0) continue;
    if (!emit(args.in[i].x, false, -1, nullptr, args.in[i].type, i) || !extend(args.in[i].x, args.in[i].type)) return false;
  }
  for (int i = 0; i < nin; i++) {
    if (placed[i] || args.in[i].op2 >= 0) continue;
    const SfIn e = args.in[i];
    if (!emit(e.x, false, e.op1, e.y, e.type, i)) return false;
    const int k = two_op_over(e.x, e.op1, e.y);
    if (k >= 0 && !emit(nullptr, true, args.in[k].op2, args.in[k].z, e.type, k)) return false;
  }
  for (int i = 0; i < nin; i++) {
    if (placed[i]) continue;
    const SfIn e = args.in[i];
    if (!emit(e.x, false, e.op1, e.y, e.type, -1) || !emit(nullptr, true, e.op2, e.z, e.type, i)) return false;
  }
  for (int f = 0; f < args.nops; f++)
    if (args.ops[f].in >= 0) args.ops[f].in = static_cast<int8_t>(slot_of[args.ops[f].in]);
  for (int i = 0; i < n; i++) args.in[i] = slots[i];
  args._pad = static_cast<uint32_t>(n);  // (the planes are unchanged: slots only reorder who feeds them)
  return n > 0;
}
static crgpu_status try_sumfam(crgpu_agg *a, const PassInput &in, uint32_t *miss_list, const AccPlan &plan, bool *done) {
  *done = false;
  static const bool enabled = [] { const char *v = std::getenv("CRGPU_AGG_SUMFAM"); return v == nullptr || v[0] != '0'; }();
  if (!enabled || !plan.lds || !plan.pk || plan.sweeps != 1 || in.list != nullptr) return CRGPU_OK;
  SfArgs args;
  std::memset(&args, 0, sizeof(args));
  int nin = 0, nops = 0, nplanes = 1;  // plane 0: rows
  int cf_col = -2, cf_cmp = -2;
  long long cf_cst = 0;
  for (const AggOpHost &h : a->ops) {
    if (h.kind == ACC_ANY_KEY) continue;
    if (h.kind == ACC_ANY_ROW || is_minmax_bytes(h.kind)) return CRGPU_OK;
    if (nops == kMaxOpsPerLaunch) return CRGPU_OK;
    if (cf_col == -2) cf_col = h.filter_col, cf_cmp = h.f_cmp, cf_cst = h.f_cst;
    if (h.filter_col != cf_col || h.f_cmp != cf_cmp || h.f_cst != cf_cst) return CRGPU_OK;  // one FILTER for all
    SfOp &op = args.ops[nops++];
    op.kind = static_cast<int8_t>(h.kind);
    op.in = -1;
    op.w0 = h.w0.as<unsigned long long>();
    op.w1 = h.w1.as<unsigned long long>();
    op.seen = h.seen.as<uint8_t>();
    if (h.kind == ACC_COUNT_ROWS) continue;
    // NULL-free inputs only: then COUNT(col) is the row count and every AVG shares it
    const int cs[3] = {h.col, h.e_op1 >= 0 ? h.col2 : -1, h.e_op2 >= 0 ? h.col3 : -1};
    for (int c : cs)
      if (c >= 0 && (*in.cols)[c].valid != nullptr) return CRGPU_OK;
    if (h.kind == ACC_COUNT) continue;
    if (h.in_type == CRGPU_BYTES) return CRGPU_OK;
    SfIn e;
    std::memset(&e, 0, sizeof(e));
    e.x = (*in.cols)[h.col].data;
    e.type = static_cast<int8_t>(h.in_type);
    e.op1 = static_cast<int8_t>(h.e_op1 >= 0 ? h.e_op1 : -1);
    e.op2 = static_cast<int8_t>(h.e_op1 >= 0 && h.e_op2 >= 0 ? h.e_op2 : -1);
    if (e.op1 >= 0) e.y = static_cast<const double *>((*in.cols)[h.col2].data);
    if (e.op2 >= 0) e.z = static_cast<const double *>((*in.cols)[h.col3].data);
    e.p_sum = e.p_hi = e.p_min = e.p_max = -1;
    int at = -1;
    for (int i = 0; i < nin && at < 0; i++)
      if (args.in[i].x == e.x && args.in[i].y == e.y && args.in[i].z == e.z && args.in[i].op1 == e.op1 && args.in[i].op2 == e.op2 &&
          args.in[i].type == e.type)
        at = i;
    if (at < 0) {
      if (nin == kSfMaxIn) return CRGPU_OK;
      at = nin;
      args.in[nin++] = e;
    }
    op.in = static_cast<int8_t>(at);
    SfIn &si = args.in[at];
    const bool f64 = h.in_type == CRGPU_FLOAT64;
    auto new_plane = [&](unsigned long long init) -> int8_t {
      if (nplanes >= kSfMaxPlanes) return -1;
      args.plane_init[nplanes] = init;
      return static_cast<int8_t>(nplanes++);
    };
    switch (h.kind) {
      case ACC_SUM_F64:
      case ACC_AVG_F64:
        if (si.p_sum < 0) si.p_sum = new_plane(0);
        break;
      case ACC_SUM_I128:
        if (si.p_sum < 0) { si.p_sum = new_plane(0); si.p_hi = new_plane(0); }
        break;
      case ACC_MIN_I64:
      case ACC_BOOL_AND:
        if (si.p_min < 0) si.p_min = new_plane(0x7FFFFFFFFFFFFFFFull);
        break;
      case ACC_MAX_I64:
      case ACC_BOOL_OR:
        if (si.p_max < 0) si.p_max = new_plane(0x8000000000000000ull);
        break;
      case ACC_MIN_F64:
        if (si.p_min < 0) si.p_min = new_plane(~0ull);
        break;
      case ACC_MAX_F64:
        if (si.p_max < 0) si.p_max = new_plane(0ull);
        break;
      default: return CRGPU_OK;
    }
    (void)f64;
    if ((h.kind == ACC_SUM_F64 || h.kind == ACC_AVG_F64) && si.p_sum < 0) return CRGPU_OK;
    if (h.kind == ACC_SUM_I128 && (si.p_sum < 0 || si.p_hi < 0)) return CRGPU_OK;
    if ((h.kind == ACC_MIN_I64 || h.kind == ACC_BOOL_AND || h.kind == ACC_MIN_F64) && si.p_min < 0) return CRGPU_OK;
    if ((h.kind == ACC_MAX_I64 || h.kind == ACC_BOOL_OR || h.kind == ACC_MAX_F64) && si.p_max < 0) return CRGPU_OK;
  }
  if (nops == 0 || nin == 0) return CRGPU_OK;
  if (cf_col >= 0) {
    const DCol &fc = (*in.cols)[cf_col];
    if (cf_cmp >= 0) {
      args.cf_kind = 2;
      args.cf_sel.data = static_cast<const long long *>(fc.data);
      args.cf_sel.valid = fc.valid;
      args.cf_sel.cst = cf_cst;
      args.cf_sel.cmp = cf_cmp;
    } else {
      args.cf_kind = 1;
      args.cf_data = static_cast<const uint8_t *>(fc.data);
      args.cf_valid = fc.valid;
    }
  }
  // partial table: nplanes planes of (G + 1) x R words + the key table
  const size_t G = plan.G;
  const size_t table = static_cast<size_t>(plan.C) * (a->pack.two_words ? 20 : 12) + 8;
  const size_t per_rep = static_cast<size_t>(nplanes) * (G + 1) * 8;
  if (per_rep + table > kLdsLarge) return CRGPU_OK;
  uint32_t R = 1;
  while (R < static_cast<uint32_t>(kMaxReplicas) && per_rep * (R * 2) + table <= kLdsReplicated) R *= 2;
  const size_t lds = ((per_rep * R + table) + 15) & ~size_t(15);
  const size_t fit = (160 * 1024) / lds;
  const int wgs_per_cu = fit >= 8 ? 8 : fit >= 4 ? 4 : fit >= 2 ? 2 : 1;
  const int blk = wgs_per_cu == 8 ? 256 : wgs_per_cu == 1 ? 1024 : 512;
  args.keys = in.keys;
  args.store = a->store_host;
  args.pack = a->pack;
  args.nops = nops;
  args.nplanes = static_cast<uint32_t>(nplanes);
  args.m = in.m;
  args.meta = a->meta.as<unsigned long long>();
  args.mask = a->cap - 1;
  args.G = static_cast<uint32_t>(G);
  args.R = R;
  args.C = plan.C;
  args.miss_list = miss_list;
  args.counters = a->counters.as<unsigned int>();
  hipStream_t st = stream();
  // the loads-up-front form: one or two NULL-free key columns, 8-byte inputs, rows addressable by 32-bit byte offsets
  static const bool sf8_on = [] { const char *v = std::getenv("CRGPU_AGG_SF8"); return v == nullptr || v[0] != '0'; }();
  bool sf8 = sf8_on && in.m < (int64_t(1) << 32) && (in.keys.n == 1 || in.keys.n == 2);
  bool key_bytes = false;  // key form: every column BYTES, or every column 8 bytes wide
  for (int c = 0; sf8 && c < in.keys.n; c++) {
    const DCol &kc = in.keys.c[c];
    const bool is_b = kc.type == CRGPU_BYTES;
    sf8 = kc.valid == nullptr && (is_b || type_width(kc.type) == 8) && (c == 0 || is_b == key_bytes);
    key_bytes = is_b;
  }
  for (int i = 0; sf8 && i < nin; i++) sf8 = args.in[i].type == CRGPU_FLOAT64;
  for (int f = 0; sf8 && f < nops; f++) {
    const int kd = args.ops[f].kind;
    sf8 = kd == ACC_COUNT_ROWS || kd == ACC_COUNT || kd == ACC_SUM_F64 || kd == ACC_AVG_F64;
  }
  if (sf8) {
    SfArgs a8 = args;
    a8.nops = nops;
    if (sf8_slots(a8, nin)) {
      const int nslots = static_cast<int>(a8._pad);
      a8._pad = 0;
      // 16 waves per CU (measured: 46 B/row streams at 5.8 TB/s with 4 waves per SIMD, 4.6-4.8 with 7-8): 4 workgroups of
      // 256 threads while the partial table fits a quarter of the LDS; lane-private replicas (64) when they fit
      const size_t per_rep8 = static_cast<size_t>(kSf8Planes(nslots)) * (G + 1) * 8;
      if (per_rep8 + table > kLdsLarge) return CRGPU_OK;  // (cannot happen before the generic form refuses: it has more planes)
      uint32_t R8 = 1;
      while (R8 < 64u && per_rep8 * (R8 * 2) + table <= 36u * 1024u) R8 *= 2;
      const size_t lds8 = ((per_rep8 * R8 + table) + 15) & ~size_t(15);
      const int wgs8 = lds8 <= 36u * 1024u ? 4 : lds8 <= 76u * 1024u ? 2 : 1;
      const int blk8 = 1024 / wgs8;
      a8.R = R8;
      ProfScope prof("agg_fused_sumfam");
      // the kernel addresses rows by 32-bit byte offsets: one launch per 2^28 rows, the column bases moved along
      // (the launches share the miss list and its counter; nothing is read back in between)
      const int64_t kSeg = int64_t(1) << 28;
      for (int64_t base = 0; base < in.m; base += kSeg) {
        SfArgs s8 = a8;
        s8.m = in.m - base < kSeg ? in.m - base : kSeg;
        s8.row_base = static_cast<uint32_t>(base);
        for (int c = 0; c < in.keys.n; c++) {
          DCol &kc = s8.keys.c[c];
          if (kc.type == CRGPU_BYTES) kc.offsets += base;
          else kc.data = static_cast<const uint8_t *>(kc.data) + base * 8;
        }
        for (int i = 0; i < nslots; i++) {
          if (s8.in[i].x != nullptr) s8.in[i].x = static_cast<const uint8_t *>(s8.in[i].x) + base * 8;
          if (s8.in[i].y != nullptr) s8.in[i].y += base;
        }
        if (s8.cf_kind == 2) {
          s8.cf_sel.data += base;
          if (s8.cf_sel.valid != nullptr) s8.cf_sel.valid += base >> 3;
        } else if (s8.cf_kind == 1) {
          s8.cf_data += base;
          if (s8.cf_valid != nullptr) s8.cf_valid += base >> 3;
        }
        switch (nslots) {
          case 1: launch_sf8_n<1>(s8, in.keys.n, key_bytes, blk8, wgs8, lds8, st); break;
          case 2: launch_sf8_n<2>(s8, in.keys.n, key_bytes, blk8, wgs8, lds8, st); break;
          case 3: launch_sf8_n<3>(s8, in.keys.n, key_bytes, blk8, wgs8, lds8, st); break;
          case 4: launch_sf8_n<4>(s8, in.keys.n, key_bytes, blk8, wgs8, lds8, st); break;
          case 5: launch_sf8_n<5>(s8, in.keys.n, key_bytes, blk8, wgs8, lds8, st); break;
          case 6: launch_sf8_n<6>(s8, in.keys.n, key_bytes, blk8, wgs8, lds8, st); break;
          case 7: launch_sf8_n<7>(s8, in.keys.n, key_bytes, blk8, wgs8, lds8, st); break;
          default: launch_sf8_n<8>(s8, in.keys.n, key_bytes, blk8, wgs8, lds8, st); break;
        }
        CRGPU_HIP(hipGetLastError());
      }
      *done = true;
      return CRGPU_OK;
    }
  }
  const int64_t per_wg = static_cast<int64_t>(blk) * kAccRows * 2;
  int64_t want = (in.m + per_wg - 1) / per_wg;
  if (want < 1) want = 1;
  const int64_t max_grid = static_cast<int64_t>(num_cus()) * wgs_per_cu;
  const dim3 grid(static_cast<unsigned int>(want < max_grid ? want : max_grid)), block(blk);
  ProfScope prof("agg_fused_sumfam");
  switch (nin) {
    case 1: launch_sumfam_n<1>(args, grid, block, lds, st); break;
    case 2: launch_sumfam_n<2>(args, grid, block, lds, st); break;
    case 3: launch_sumfam_n<3>(args, grid, block, lds, st); break;
    case 4: launch_sumfam_n<4>(args, grid, block, lds, st); break;
    case 5: launch_sumfam_n<5>(args, grid, block, lds, st); break;
    case 6: launch_sumfam_n<6>(args, grid, block, lds, st); break;
    case 7: launch_sumfam_n<7>(args, grid, block, lds, st); break;
    default: launch_sumfam_n<8>(args, grid, block, lds, st); break;
  }
  CRGPU_HIP(hipGetLastError());
  *done = true;
  return CRGPU_OK;
}

static crgpu_status read_counters(crgpu_agg *a, unsigned int *out2) {
  unsigned int c[4] = {0, 0, 0, 0};
  CRGPU_HIP(hipMemcpyAsync(c, a->counters.p, 16, hipMemcpyDeviceToHost, stream()));
  CRGPU_HIP(hipStreamSynchronize(stream()));
  out2[0] = c[0];
  out2[1] = c[1];
  return CRGPU_OK;
}

static crgpu_status consume_chunk(crgpu_agg *a, const std::vector<DCol> &cols, int64_t n, int64_t remaining) {
  hipStream_t st = stream();
  // BYTES key payload / BYTES ANY_NOT_NULL values: room for every byte of the chunk, reserved when the chunk turns out
  // to have rows the fused pass did not place (only those can append): a steady-state chunk skips the read-back
  // (all the boundary offsets are requested together: one wait per chunk, not one per column)
  bool reserved = false;
  auto reserve_bytes = [&]() -> crgpu_status {
    reserved = true;
    std::vector<const DCol *> bcols;
    for (size_t k = 0; k < a->ks.size(); k++)
      if (cols[a->group_cols[k]].type == CRGPU_BYTES) bcols.push_back(&cols[a->group_cols[k]]);
    for (size_t f = 0; f < a->ops.size(); f++)
      if (a->ops[f].uses_any_store()) bcols.push_back(&cols[a->ops[f].col]);
    std::vector<uint32_t> bounds(bcols.size() * 2 + 2, 0);
    for (size_t i = 0; i < bcols.size(); i++) {
      CRGPU_HIP(hipMemcpyAsync(&bounds[2 * i], bcols[i]->offsets, 4, hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipMemcpyAsync(&bounds[2 * i + 1], bcols[i]->offsets + n, 4, hipMemcpyDeviceToHost, st));
    }
    if (!bcols.empty()) CRGPU_HIP(hipStreamSynchronize(st));
    size_t i = 0;
    for (size_t k = 0; k < a->ks.size(); k++) {
      if (cols[a->group_cols[k]].type != CRGPU_BYTES) continue;
      const size_t before = a->ks[k].b_buf.cap;
      CRGPU_TRY(a->ks[k].b_buf.ensure(static_cast<size_t>(a->ks[k].tail + (bounds[2 * i + 1] - bounds[2 * i]) + 16), true));
      if (a->ks[k].b_buf.cap != before) a->descriptors_dirty = true;
      i++;
    }
    for (size_t f = 0; f < a->ops.size(); f++) {
      AggOpHost &op = a->ops[f];
      if (!op.uses_any_store()) continue;
      CRGPU_TRY(op.any_buf.ensure(static_cast<size_t>(op.any_tail + (bounds[2 * i + 1] - bounds[2 * i]) + 16), true));
      i++;
    }
    return CRGPU_OK;
  };
  if (a->need_any_fix || a->ngroups == 0) CRGPU_TRY(reserve_bytes());  // (value stores may be appended to by any pass)
  CRGPU_TRY(upload_descriptors(a, &cols));  // the chunk's column pointers are part of the op descriptors

  PassInput in;
  in.cols = &cols;
  in.keys.n = static_cast<int>(a->group_cols.size());
  in.keys._pad = 0;
  for (int k = 0; k < in.keys.n; k++) in.keys.c[k] = cols[a->group_cols[k]];
  in.list = nullptr;
  in.m = n;

  CRGPU_TRY(a->list_a.ensure(static_cast<size_t>(n) * 4));
  uint32_t *cur_list = nullptr;                       // list backing `in.list`
  uint32_t *next_list = a->list_a.as<uint32_t>();     // where the next pass's rows are collected
  const int64_t groups_before = a->ngroups;

  // steady state: one fused pass; only rows of groups that do not exist yet remain
  const AccPlan fused_plan = plan_accumulate(a, a->ngroups, n, true);
  const bool fused = !a->need_any_fix && a->ngroups > 0 && a->new_ratio < 0.5 && fused_plan.sweeps == 1 && !fused_plan.part;
  if (fused) {
    CRGPU_HIP(hipMemsetAsync(a->counters.p, 0, 16, st));
    bool sumfam = false;
    CRGPU_TRY(try_sumfam(a, in, next_list, fused_plan, &sumfam));
    if (!sumfam) CRGPU_TRY(launch_accumulate(a, in, true, next_list, fused_plan));
    unsigned int c[2];
    CRGPU_TRY(read_counters(a, c));
    in.m = c[1];
    cur_list = next_list;
    in.list = cur_list;
  }
  for (int round = 0; in.m > 0; round++) {
    if (round >= 40) return fail(CRGPU_ERR_HIP, "crgpu_dev_agg_consume: deferred rows did not converge");
    if (!reserved) {
      CRGPU_TRY(reserve_bytes());
      if (a->descriptors_dirty) CRGPU_TRY(upload_descriptors(a, &cols));
    }
    // index capacity: at most half full after the expected (later rounds: worst-case) number of new groups
    uint64_t est = static_cast<uint64_t>(in.m);
    if (round == 0 && a->new_ratio < 1.0) {
      const uint64_t guess = static_cast<uint64_t>(static_cast<double>(in.m) * a->new_ratio * 1.5) + 4096;
      if (guess < est) est = guess;
    }
    if (round == 0 && a->seq > 0 && a->new_ratio >= 0.25 && remaining > in.m) {
      // high-cardinality input: size the index once for the rest of this call instead of rebuilding it chunk by chunk
      const uint64_t rest = static_cast<uint64_t>(static_cast<double>(remaining) * (a->new_ratio < 1.0 ? a->new_ratio : 1.0));
      if (rest > est) est = rest;
    }
    const uint64_t want = next_pow2((static_cast<uint64_t>(a->ngroups) + est) * 2);
    if (want > kMaxCap) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_consume: more than 2^29 groups per GPU");
    if (want > a->cap) CRGPU_TRY(reindex(a, static_cast<uint32_t>(want)));

    const size_t m = static_cast<size_t>(in.m);
    CRGPU_TRY(a->out_gid.ensure(m * 4));
    CRGPU_TRY(a->rep_pos.ensure(m * 4));
    CRGPU_TRY(a->rep_slot.ensure(m * 4));
    CRGPU_TRY(a->rep_word.ensure(m * 8));
    if (cur_list == a->list_a.as<uint32_t>()) {
      CRGPU_TRY(a->list_b.ensure(m * 4));
      next_list = a->list_b.as<uint32_t>();
    } else {
      next_list = a->list_a.as<uint32_t>();
    }
    CRGPU_HIP(hipMemsetAsync(a->counters.p, 0, 16, st));
    ResolveArgs r;
    std::memset(&r, 0, sizeof(r));
    r.keys = in.keys;
    r.store = a->store_host;
    r.m = in.m;
    r.list = in.list;
    r.meta = a->meta.as<unsigned long long>();
    r.mask = a->cap - 1;
    r.cas_first = (cur_list != nullptr || a->new_ratio >= 0.5) ? 1u : 0u;  // misses of a fused pass, or mostly new groups
    r.out_gid = a->out_gid.as<uint32_t>();
    r.rep_pos = a->rep_pos.as<uint32_t>();
    r.rep_slot = a->rep_slot.as<uint32_t>();
    r.rep_word = a->rep_word.as<unsigned long long>();
    r.next_list = next_list;
    r.counters = a->counters.as<unsigned int>();
    {
      ProfScope prof("agg_resolve");
      hipLaunchKernelGGL(agg_resolve_kernel, dim3(grid_for((in.m + kResolveRows - 1) / kResolveRows, 256)), dim3(256), 0, st, r);
      CRGPU_HIP(hipGetLastError());
    }
    unsigned int c[2];
    CRGPU_TRY(read_counters(a, c));
    const unsigned int nnew = c[0], ndef = c[1];
    if (nnew > 0) {
      CRGPU_TRY(ensure_groups(a, static_cast<size_t>(a->ngroups) + nnew));
      if (a->descriptors_dirty) CRGPU_TRY(upload_descriptors(a, &cols));
      FixupArgs f;
      std::memset(&f, 0, sizeof(f));
      f.keys = in.keys;
      f.store = a->store_host;
      f.ops = a->ops_dev.as<AggOpDev>();
      f.naggs = static_cast<int>(a->ops.size());
      f.nnew = nnew;
      f.g0 = static_cast<uint32_t>(a->ngroups);
      f.list = in.list;
      f.rep_pos = a->rep_pos.as<uint32_t>();
      f.rep_slot = a->rep_slot.as<uint32_t>();
      f.rep_word = a->rep_word.as<unsigned long long>();
      f.meta = a->meta.as<unsigned long long>();
      f.gword = a->gword.as<unsigned long long>();
      f.b_tails = a->tails_dev.as<unsigned long long>();
      const size_t nops = a->ops.size();
      f.any_off = reinterpret_cast<uint64_t *const *>(a->any_ptrs_dev.as<void *>());
      f.any_len = reinterpret_cast<uint32_t *const *>(a->any_ptrs_dev.as<void *>() + nops);
      f.any_buf = reinterpret_cast<uint8_t *const *>(a->any_ptrs_dev.as<void *>() + 2 * nops);
      f.any_tails = a->any_tails_dev.as<unsigned long long>();
      f.seq_base = a->seq;
      ProfScope prof("agg_fixup");
      hipLaunchKernelGGL(agg_fixup_kernel, dim3(grid_for((static_cast<int64_t>(nnew) + kFixupGroups - 1) / kFixupGroups, 256)), dim3(256), 0, st, f);
      CRGPU_HIP(hipGetLastError());
      a->ngroups += nnew;
    }
    if (static_cast<int64_t>(ndef) + static_cast<int64_t>(nnew) < in.m) {  // rows besides claimants (fixup) and deferred ones
      const AccPlan plan = plan_accumulate(a, a->ngroups, in.m, false);
      if (plan.part) CRGPU_TRY(launch_partitioned(a, in, plan));
      else CRGPU_TRY(launch_accumulate(a, in, false, nullptr, plan));
      if (a->need_any_fix) {
        const size_t nops = a->ops.size();
        AnyFixArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.ops = a->ops_dev.as<AggOpDev>();
        fa.naggs = static_cast<int>(nops);
        fa.m = in.m;
        fa.list = in.list;
        fa.out_gid = a->out_gid.as<uint32_t>();
        fa.meta = a->meta.as<unsigned long long>();
        fa.seq_base = a->seq;
        fa.any_off = reinterpret_cast<uint64_t *const *>(a->any_ptrs_dev.as<void *>());
        fa.any_len = reinterpret_cast<uint32_t *const *>(a->any_ptrs_dev.as<void *>() + nops);
        fa.any_buf = reinterpret_cast<uint8_t *const *>(a->any_ptrs_dev.as<void *>() + 2 * nops);
        fa.any_tails = a->any_tails_dev.as<unsigned long long>();
        ProfScope prof("agg_any_fixup");
        hipLaunchKernelGGL(agg_any_fixup_kernel, dim3(grid_for(in.m, 256)), dim3(256), 0, st, fa);
        CRGPU_HIP(hipGetLastError());
        if (a->need_minmax_bytes) {
          for (int phase = 0; phase < 3; phase++) hipLaunchKernelGGL(agg_minmax_bytes_kernel, dim3(grid_for(in.m, 256)), dim3(256), 0, st, fa, phase);
          CRGPU_HIP(hipGetLastError());
        }
      }
    }
    CRGPU_HIP(hipStreamSynchronize(st));
    cur_list = next_list;
    in.list = cur_list;
    in.m = ndef;
  }
  // the BYTES append positions live on the device; the host copy sizes the next reservation
  if (a->bytes_keys && a->ngroups != groups_before) {  // (no new group: no key bytes were appended)
    unsigned long long tails[kMaxKeyCols];
    CRGPU_HIP(hipMemcpyAsync(tails, a->tails_dev.p, kMaxKeyCols * 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    for (size_t k = 0; k < a->ks.size(); k++) a->ks[k].tail = tails[k];
  }
  if (a->need_any_fix) {
    std::vector<unsigned long long> any_tails(a->ops.size(), 0);
    CRGPU_HIP(hipMemcpyAsync(any_tails.data(), a->any_tails_dev.p, any_tails.size() * 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    for (size_t f = 0; f < a->ops.size(); f++)
      if (a->ops[f].uses_any_store()) a->ops[f].any_tail = any_tails[f];
  }
  CRGPU_HIP(hipStreamSynchronize(st));  // `cols` descriptors and host temporaries are done
  a->new_ratio = static_cast<double>(a->ngroups - groups_before) / static_cast<double>(n);
  a->seq += static_cast<unsigned long long>(n);
  return CRGPU_OK;
}

static constexpr int64_t kBulkMinRows = int64_t(1) << 23;

__global__ void agg_sample_keys_kernel(const long long *__restrict__ keys, int64_t stride, int nsample, long long *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nsample) out[i] = keys[static_cast<int64_t>(i) * stride];
}

// Does a larger sample (65536 keys at a fixed stride) hold a repeated key?  The 8192-key sample that picks the form
// cannot tell "every key 2-4 times" from "all distinct" (0.2-0.8 expected pairs in 2^27 rows); this one has tens of
// pairs then, so the all-distinct attempt is only made when it will stand.  Keys are inserted into a 2^17-slot table
// in global memory with a CAS each; `wide`: 64-bit hashes of the keys are used (a collision only costs the attempt).
static constexpr int kDupSample = 1 << 16;
__global__ __launch_bounds__(256) void agg_sample_dups_kernel(DCol key, DCol key2, int wide, int64_t stride, unsigned long long *__restrict__ table,
                                                              unsigned int *__restrict__ dups) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kDupSample) return;
  const int64_t r = static_cast<int64_t>(i) * stride;
  unsigned long long k;
  if (!wide) {
    k = static_cast<unsigned long long>(static_cast<const long long *>(key.data)[r]);
  } else if (key.type != CRGPU_BYTES) {
    const uint64_t k1 = key2.data != nullptr ? static_cast<uint64_t>(col_get_int(key2, r)) : 0ull;
    k = fmix64(hash_step(hash_step(0x9E3779B97F4A7C15ull, static_cast<uint64_t>(col_get_int(key, r))), k1));
  } else {
    const uint32_t b = key.offsets[r], len = key.offsets[r + 1] - b;
    uint64_t h = 0x9E3779B97F4A7C15ull + len;
    const uint8_t *p = static_cast<const uint8_t *>(key.data) + b;
    for (uint32_t j = 0; j < len && j < 64; j++) h = hash_step(h, p[j]);
    k = fmix64(h);
  }
  constexpr unsigned long long kEmpty = ~0ull;
  if (k == kEmpty) return;
  uint32_t sl = static_cast<uint32_t>(fmix64(k ^ 0xD6E8FEB86659FD93ull)) & (2 * kDupSample - 1);
  for (int probes = 0; probes < 2 * kDupSample; probes++) {
    const unsigned long long old = atomicCAS(&table[sl], kEmpty, k);
    if (old == kEmpty) return;
    if (old == k) {
      atomicAdd(dups, 1u);
      return;
    }
    sl = (sl + 1) & (2 * kDupSample - 1);
  }
}

// BYTES keys: the sample holds a 64-bit hash of each sampled key; out[nsample] is set when a sampled key is longer than
// the 16 bytes a wide row holds
__global__ void agg_sample_wide_kernel(DCol key, DCol key2, int64_t stride, int nsample, long long *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsample) return;
  const int64_t r = static_cast<int64_t>(i) * stride;
  if (key.type != CRGPU_BYTES) {
    const uint64_t k1 = key2.data != nullptr ? static_cast<uint64_t>(col_get_int(key2, r)) : 0ull;  // (one key, two values: no key2)
    out[i] = static_cast<long long>(wide_key_hash(static_cast<uint64_t>(col_get_int(key, r)), k1, 16));
    return;
  }
  const uint32_t a0 = key.offsets[r], len = key.offsets[r + 1] - a0;
  if (len > 16u) {
    out[nsample] = 1;
    out[i] = 0;
    return;
  }
  const uint8_t *p = static_cast<const uint8_t *>(key.data) + a0;
  uint64_t k0 = 0, k1 = 0;
  for (uint32_t j = 0; j < len; j++) {
    if (j < 8) k0 |= static_cast<uint64_t>(p[j]) << (8u * j);
    else k1 |= static_cast<uint64_t>(p[j]) << (8u * (j - 8));
  }
  out[i] = static_cast<long long>(wide_key_hash(k0, k1, len));
}

// sets *flag when one of the first n bits of a validity bitmap is clear
__global__ __launch_bounds__(256) void agg_bitmap_has_null_kernel(const uint8_t *__restrict__ valid, int64_t n, unsigned int *__restrict__ flag) {
  const int64_t nbytes = (n + 7) / 8;
  bool bad = false;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < nbytes; i += static_cast<int64_t>(gridDim.x) * 256) {
    const uint32_t want = (i == nbytes - 1 && (n & 7)) ? ((1u << (n & 7)) - 1u) : 0xFFu;
    bad |= (valid[i] & want) != want;
  }
  if (bad) *flag = 1u;
}

// Bulk path (see agg_bulk_kernel).  *done = true when the whole input was aggregated; otherwise nothing was changed
// and the caller continues with the incremental path.
// `append`: the table already holds groups, all of them from earlier bulk calls, and the caller vouches that no key of
// this input is among them (try_bulk_huge: the pieces of one input split by key hash): the new groups are numbered
// behind the old ones and nothing else changes -- the hash index is not in use yet.
static crgpu_status try_bulk(crgpu_agg *a, const std::vector<DCol> &cols, int64_t n, bool *done, bool append = false) {
  *done = false;
  if (!append && (a->ngroups != 0 || a->seq != 0)) return CRGPU_OK;
  if (append && a->index_valid && a->ngroups != 0) return CRGPU_OK;  // (groups that live in the index: not this path)
  // (a piece of a split input may fall a little short of the size a first input needs to be worth the partitioning)
  if (n < (append ? kBulkMinRows / 2 : kBulkMinRows) || n > (int64_t(1) << 27) || a->need_any_fix || a->has_computed) return CRGPU_OK;
  if (a->group_cols.size() != 1 && a->group_cols.size() != 2) return CRGPU_OK;
  const int key_type = a->types[a->group_cols[0]];
  // One INT64 key travels as 16-byte {key, value} rows; one BYTES key of at most 16 bytes, or two integer keys, as
  // 32-byte {k0, k1, value, length} rows (wide form, the sub form only: it exists for config G's all-distinct keys)
  const bool two_ints = a->group_cols.size() == 2;
  if (two_ints && (key_type != CRGPU_INT64 || a->types[a->group_cols[1]] != CRGPU_INT64)) return CRGPU_OK;
  int vcol = -1, vcol2 = -1, nops = 0;
  for (const AggOpHost &h : a->ops) {
    if (h.kind == ACC_ANY_KEY) continue;
    nops++;
    if (h.filter_col >= 0) return CRGPU_OK;
    if (h.kind != ACC_COUNT_ROWS) {
      if (h.in_type != CRGPU_INT64 && h.in_type != CRGPU_FLOAT64) return CRGPU_OK;
      if (vcol < 0 || vcol == h.col) vcol = h.col;
      else if (vcol2 < 0 || vcol2 == h.col) vcol2 = h.col;
      else return CRGPU_OK;
    }
  }
  if (nops > kMaxOpsPerLaunch) return CRGPU_OK;
  // a second value column rides in the wide row's fourth word, which is free only when the key is one integer column
  // (the final stage of a distributed GROUP BY: key, partial sum, partial count)
  const bool two_vals = vcol2 >= 0;
  if (two_vals && (two_ints || key_type != CRGPU_INT64)) return CRGPU_OK;
  const bool wide = two_ints || two_vals || key_type == CRGPU_BYTES;
  if (key_type != CRGPU_INT64 && !wide) return CRGPU_OK;
  if (append && wide) return CRGPU_OK;
  static const bool wide_on = getenv("CRGPU_BULK_WIDE") == nullptr || getenv("CRGPU_BULK_WIDE")[0] != '0';
  if (wide && !wide_on) return CRGPU_OK;
  DCol key = cols[a->group_cols[0]];
  DCol key2;
  std::memset(&key2, 0, sizeof(key2));
  if (two_ints) key2 = cols[a->group_cols[1]];
  // (16-byte vector loads of two rows; a piece of a hash-split input may start at an odd row: the row-wise form then)
  bool aligned16 = (reinterpret_cast<uintptr_t>(key.data) & 15u) == 0 && (vcol < 0 || (reinterpret_cast<uintptr_t>(cols[vcol].data) & 15u) == 0);
  if (!wide && !aligned16 && !append) return CRGPU_OK;
  hipStream_t st = stream();
  {
    // Columns that carry a validity bitmap without a NULL in it (the partial rows of a distributed GROUP BY arrive like
    // that: the outputs of the local stage are nullable) are as good as columns without one: a pass over n / 8 bytes
    const uint8_t *maps[4] = {key.valid, key2.valid, vcol >= 0 ? cols[vcol].valid : nullptr, two_vals ? cols[vcol2].valid : nullptr};
    bool any = false;
    for (const uint8_t *m : maps) any = any || m != nullptr;
    if (any) {
      CRGPU_TRY(a->bulk_count.ensure(16));
      CRGPU_HIP(hipMemsetAsync(a->bulk_count.p, 0, 4, st));
      for (const uint8_t *m : maps)
        if (m != nullptr) hipLaunchKernelGGL(agg_bitmap_has_null_kernel, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0, st, m, n, a->bulk_count.as<unsigned int>());
      CRGPU_HIP(hipGetLastError());
      unsigned int has_null = 0;
      CRGPU_HIP(hipMemcpyAsync(&has_null, a->bulk_count.p, 4, hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipStreamSynchronize(st));
      if (has_null) return CRGPU_OK;
      key.valid = nullptr;
      key2.valid = nullptr;
    }
  }
  const size_t bpg = static_cast<size_t>(a->lds_words) * 8 + static_cast<size_t>(a->lds_seen);
  if (bpg == 0) return CRGPU_OK;
  // Which form?  s keys sampled at a fixed stride: for G equally likely groups the sample holds G (1 - exp(-s / G))
  // distinct keys.  The small sample rejects low-cardinality inputs for ~30 us.
  int64_t distinct = 0, maxrun = 0;
  constexpr int kBigSample = 8192, kSmallSample = 512;
  // both samples are taken at once and come back in ONE round trip (each read-back is a wait of ~30 us); the small one
  // is looked at first
  std::vector<int64_t> both(static_cast<size_t>(kSmallSample + 1 + kBigSample + 1));
  CRGPU_TRY(a->scratch.ensure(both.size() * 8));
  {
    long long *dev_small = a->scratch.as<long long>(), *dev_big = dev_small + kSmallSample + 1;
    if (wide) {
      CRGPU_HIP(hipMemsetAsync(dev_small + kSmallSample, 0, 8, st));
      CRGPU_HIP(hipMemsetAsync(dev_big + kBigSample, 0, 8, st));
      hipLaunchKernelGGL(agg_sample_wide_kernel, dim3((kSmallSample + 255) / 256), dim3(256), 0, st, key, key2, n / kSmallSample, kSmallSample, dev_small);
      hipLaunchKernelGGL(agg_sample_wide_kernel, dim3(kBigSample / 256), dim3(256), 0, st, key, key2, n / kBigSample, kBigSample, dev_big);
    } else {
      hipLaunchKernelGGL(agg_sample_keys_kernel, dim3((kSmallSample + 255) / 256), dim3(256), 0, st, static_cast<const long long *>(key.data),
                         n / kSmallSample, kSmallSample, dev_small);
      hipLaunchKernelGGL(agg_sample_keys_kernel, dim3(kBigSample / 256), dim3(256), 0, st, static_cast<const long long *>(key.data),
                         n / kBigSample, kBigSample, dev_big);
    }
    CRGPU_HIP(hipGetLastError());
    CRGPU_HIP(hipMemcpyAsync(both.data(), a->scratch.p, both.size() * 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
  }
  for (const int kSample : {kSmallSample, kBigSample}) {
    const int64_t *sample = both.data() + (kSample == kBigSample ? kSmallSample + 1 : 0);
    if (wide && sample[kSample] != 0) return CRGPU_OK;  // keys longer than 16 bytes
    // distinct keys and the largest multiplicity of the sample, by open addressing (sorting 8192 keys took 0.3 ms)
    const size_t tab = static_cast<size_t>(kSample) * 4;
    std::vector<int64_t> tkey(tab);
    std::vector<int32_t> tcnt(tab, 0);
    distinct = 0;
    maxrun = 0;
    for (int i = 0; i < kSample; i++) {
      size_t at = static_cast<size_t>((static_cast<uint64_t>(sample[i]) * 0x9E3779B97F4A7C15ull) >> 40) & (tab - 1);
      while (tcnt[at] != 0 && tkey[at] != sample[i]) at = (at + 1) & (tab - 1);
      if (tcnt[at] == 0) {
        tkey[at] = sample[i];
        distinct++;
      }
      if (++tcnt[at] > maxrun) maxrun = tcnt[at];
    }
    if (kSample == 512 && distinct < 384) return CRGPU_OK;  // fewer than ~1000 groups: the fused pass is the right tool
  }
  // sub form: up to ~32 rows per group the sub-partition regions hold (1024 rows expected, capacity 2048)
  int64_t dmax = 16 * (int64_t(kBigSample) * kBigSample) / n;
  if (dmax < 8) dmax = 8;
  if (dmax > kBigSample / 4) dmax = kBigSample / 4;
  const bool sub_form = kBigSample - distinct <= dmax;
  if (wide && !sub_form) return CRGPU_OK;
  size_t gmax = 0, slots = 0;
  if (sub_form) {
    slots = kBulkSubSlots;
    gmax = (kLdsMax / 2 - 64 - slots * 8 - bpg) / (bpg + 4);  // two workgroups per CU
    if (gmax < 1280) gmax = (kLdsMax - 64 - slots * 8 - bpg) / (bpg + 4);  // wide accumulator sets: one workgroup per CU
    if (gmax > kBulkSubGroups) gmax = kBulkSubGroups;
    if (gmax < 1280) return CRGPU_OK;  // ~1024 groups are expected for all-distinct keys
  } else {
    // part form: estimate G by bisection on G (1 - exp(-s / G)) = distinct
    double lo = 1.0, hi = 1e12;
    for (int it = 0; it < 60; it++) {
      const double mid = std::sqrt(lo * hi);
      if (mid * (1.0 - std::exp(-static_cast<double>(kBigSample) / mid)) < static_cast<double>(distinct)) lo = mid;
      else hi = mid;
    }
    // a hot key would serialise its partition's workgroup: a key sampled far more often than the s / G expected (or
    // more than 8 times = 0.1 % of the rows, when that is larger) leaves the input to the other paths
    const double mu = static_cast<double>(kBigSample) / hi;
    if (static_cast<double>(maxrun) > std::max(8.0, 3.0 * mu + 6.0)) return CRGPU_OK;
    // few groups per partition: the lanes of a wave collide on the same LDS words (measured: 8 per partition 3.3 ms,
    // 16 3.1 ms, 64 3.0 ms per 2^27 rows; the gid-partitioned path takes 4.5 ms there)
    static const double min_per_part = getenv("CRGPU_BULK_MIN_PER_PART") ? atof(getenv("CRGPU_BULK_MIN_PER_PART")) : 3.0;
    const double per_part = hi / kRadixP1;
    if (per_part < min_per_part) return CRGPU_OK;
    // (12 bytes per slot: the key and the local group number, see LKEY)
    for (const size_t sl : {size_t(2048), size_t(4096), size_t(8192)}) {
      if (kLdsMax < 64 + sl * 12 + bpg) continue;
      size_t g = (kLdsMax - 64 - sl * 12 - bpg) / (bpg + 4);
      if (g > sl * 3 / 4) g = sl * 3 / 4;
      if (g > gmax) gmax = g, slots = sl;
    }
    if (per_part * 1.3 + 4.0 * std::sqrt(per_part) + 16.0 > static_cast<double>(gmax)) return CRGPU_OK;
  }
  static const bool lkey_on = getenv("CRGPU_BULK_LKEY") == nullptr || getenv("CRGPU_BULK_LKEY")[0] != '0';
  const bool lkey = !sub_form && !wide && lkey_on;
  const size_t lds = (static_cast<size_t>(a->lds_words) * 8 * (gmax + 1) + (lkey ? slots * 12 + 16 : slots * 8) + gmax * 4 +
                      static_cast<size_t>(a->lds_seen) * (gmax + 1) + 15) & ~size_t(15);
  // sub form: one region per ~1024 rows, whatever the row count -- the kernel's time follows the number of regions (one
  // workgroup each, bound by its chain of latencies), and a power-of-two count made an input just above 2^26 rows cost
  // what 2^27 rows do
  static const int64_t region_rows = getenv("CRGPU_BULK_REGION_ROWS") ? atoll(getenv("CRGPU_BULK_REGION_ROWS")) : 1024;
  uint32_t any_nsub = 0;
  if (sub_form && region_rows > 0) {
    const int64_t per = static_cast<int64_t>(kRadixP1) * region_rows;
    const int64_t want = (n + per - 1) / per;
    any_nsub = static_cast<uint32_t>(want < 1 ? 1 : (want > 128 ? 128 : want));
    if ((any_nsub & (any_nsub - 1)) == 0) any_nsub = 0;  // a power of two: the split by hash bits gives the same count
  }
  RadixSrc src;
  std::memset(&src, 0, sizeof(src));
  src.kp.n = 1;
  src.kp.c[0] = key;
  src.kp.range[0] = ~0ull;
  src.pay = vcol >= 0 ? static_cast<const uint8_t *>(cols[vcol].data) : nullptr;
  src.pay_width = vcol >= 0 ? 8 : 0;
  src.fast = vcol >= 0 && aligned16 ? 1 : 0;
  if (wide) {
    WideSrc ws;
    ws.key = key;
    ws.key2 = key2;
    ws.val = vcol >= 0 ? static_cast<const unsigned long long *>(cols[vcol].data) : nullptr;
    ws.val2 = two_vals ? static_cast<const unsigned long long *>(cols[vcol2].data) : nullptr;
    bool ok = false;
    CRGPU_TRY(radix_partition_split_wide(a->radix, ws, n, &ok, kBulkRows, 10, 7, any_nsub));
    if (!ok) return CRGPU_OK;  // a sub-partition overflowed, or a key is longer than 16 bytes
    if (key_type == CRGPU_BYTES) CRGPU_TRY(a->ks[0].b_buf.ensure(static_cast<size_t>(n) * 16 + 16));
  } else if (sub_form) {
    bool ok = false;
    CRGPU_TRY(radix_partition_split(a->radix, src, n, &ok, kBulkRows, 10, 7, any_nsub));  // ~1024 rows per sub-partition
    if (!ok) return CRGPU_OK;  // a sub-partition overflowed: many rows per key after all
  } else {
    CRGPU_TRY(radix_partition_level1(a->radix, src, n));
  }
  const uint32_t g0 = static_cast<uint32_t>(a->ngroups);
  CRGPU_TRY(ensure_groups(a, static_cast<size_t>(g0) + static_cast<size_t>(n)));
  CRGPU_TRY(upload_descriptors(a, &cols));
  CRGPU_TRY(a->bulk_count.ensure(16));
  CRGPU_HIP(hipMemsetAsync(a->bulk_count.p, 0, 16, st));
  BulkArgs b;
  std::memset(&b, 0, sizeof(b));
  b.rows = sub_form ? a->radix.rows_b.p : a->radix.rows_a.p;
  b.starts = sub_form ? nullptr : a->radix.starts.as<uint32_t>();
  b.counts = sub_form ? a->radix.sub_count.as<uint32_t>() : a->radix.totals.as<uint32_t>();
  int f = 0;
  for (const AggOpHost &h : a->ops) {
    if (h.kind == ACC_ANY_KEY) continue;
    b.op_val[f] = (two_vals && h.kind != ACC_COUNT_ROWS && h.col == vcol2) ? 1 : 0;
    OpK &k = b.ops[f++];
    k.kind = static_cast<int8_t>(h.kind);
    k.in_type = static_cast<int8_t>(h.in_type);
    k.lw0 = static_cast<int8_t>(h.lw0);
    k.lw1 = static_cast<int8_t>(h.lw1);
    k.lseen = static_cast<int8_t>(h.lseen);
    k.w0 = h.w0.as<unsigned long long>();
    k.w1 = h.w1.as<unsigned long long>();
    k.seen = h.seen.as<uint8_t>();
  }
  b.naggs = f;
  b.nwords = static_cast<uint32_t>(a->lds_words);
  b.nseen = static_cast<uint32_t>(a->lds_seen);
  b.key_type = key_type;
  b.key_data = a->ks[0].data.as<uint8_t>();
  b.key_present = a->ks[0].present.as<uint8_t>();
  if (two_ints) {
    b.key_type2 = a->types[a->group_cols[1]];
    b.key_data2 = a->ks[1].data.as<uint8_t>();
    b.key_present2 = a->ks[1].present.as<uint8_t>();
  }
  b.b_off = a->ks[0].b_off.as<uint64_t>();
  b.b_len = a->ks[0].b_len.as<uint32_t>();
  b.b_buf = a->ks[0].b_buf.as<uint8_t>();
  b.gword = a->gword.as<unsigned long long>();
  b.ngroups = a->bulk_count.as<unsigned int>();
  b.failed = a->bulk_count.as<unsigned int>() + 1;
  b.not16 = a->bulk_count.as<unsigned int>() + 2;
  b.wmask = two_vals ? 0ull : ~0ull;
  b.g0 = g0;
  b.lkey = lkey ? 1u : 0u;
  b.gmax = static_cast<uint32_t>(gmax);
  b.slots = static_cast<uint32_t>(slots);
  const unsigned nregions = sub_form ? static_cast<unsigned>(kRadixP1) * a->radix.nsub : static_cast<unsigned>(kRadixP1);
  // sub form, no repeated key among the 8192 sampled: first the all-distinct attempt (see agg_bulk_kernel)
  static const bool try_on = getenv("CRGPU_BULK_TRY") == nullptr || getenv("CRGPU_BULK_TRY")[0] != '0';
  bool tried_ok = false;
  unsigned int try_groups = 0;
  bool try_it = sub_form && try_on && distinct == kBigSample && !(append && a->try_voided);
  if (try_it && n >= 4 * static_cast<int64_t>(kDupSample)) {
    CRGPU_TRY(a->p_gid.ensure(static_cast<size_t>(2 * kDupSample) * 8 + 16));
    CRGPU_HIP(hipMemsetAsync(a->p_gid.p, 0xFF, static_cast<size_t>(2 * kDupSample) * 8, st));
    unsigned int *dups_dev = reinterpret_cast<unsigned int *>(a->p_gid.as<unsigned long long>() + 2 * kDupSample);
    CRGPU_HIP(hipMemsetAsync(dups_dev, 0, 4, st));
    hipLaunchKernelGGL(agg_sample_dups_kernel, dim3(kDupSample / 256), dim3(256), 0, st, key, key2, wide ? 1 : 0, n / kDupSample,
                       a->p_gid.as<unsigned long long>(), dups_dev);
    CRGPU_HIP(hipGetLastError());
    unsigned int dups = 0;
    CRGPU_HIP(hipMemcpyAsync(&dups, dups_dev, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (dups != 0) try_it = false;
  }
  if (try_it) {
    CRGPU_TRY(a->p_offsets.ensure(static_cast<size_t>(nregions + 1) * 8));
    CRGPU_TRY(a->p_flags.ensure(256));
    CRGPU_HIP(hipMemsetAsync(a->p_flags.p, 0, 256, st));
    // (the regions hold every row -- no NULL key, no region overflowed --, so their counts add up to n: no read-back)
    const uint64_t total_rows = static_cast<uint64_t>(n);
    CRGPU_TRY(exclusive_scan_u32(b.counts, nregions, a->p_offsets.as<uint64_t>(), a->scratch, nullptr));
    BulkArgs bt = b;
    bt.slots = 4096;
    bt.lkey = !wide && lkey_on ? 1u : 0u;
    const size_t try_lds = static_cast<size_t>(bt.slots) * (bt.lkey ? 8 : 4);
    bt.try_base = a->p_offsets.as<uint64_t>();
    bt.try_failed = a->p_flags.as<unsigned int>();
    {
      ProfScope prof("agg_bulk_distinct");
      // (512 threads: 1.59 ms; 256 threads, eight workgroups per CU: 1.71 ms)
      if (wide) hipLaunchKernelGGL((agg_bulk_kernel<true, kBulkTryThreads, true>), dim3(nregions), dim3(kBulkTryThreads), try_lds, st, bt);
      else hipLaunchKernelGGL((agg_bulk_kernel<false, kBulkTryThreads, true>), dim3(nregions), dim3(kBulkTryThreads), try_lds, st, bt);
      CRGPU_HIP(hipGetLastError());
    }
    unsigned int failed_try = 0;
    CRGPU_HIP(hipMemcpyAsync(&failed_try, bt.try_failed, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (failed_try == 0) {  // every region held distinct keys: group ids 0 .. rows - 1 are written
      tried_ok = true;
      try_groups = static_cast<unsigned int>(total_rows);
    } else {
      a->try_voided = true;
    }
  }
  if (!tried_ok) {
    ProfScope prof(sub_form ? "agg_bulk" : "agg_bulk_part");
    static const bool slim_on = getenv("CRGPU_BULK_SLIM") == nullptr || getenv("CRGPU_BULK_SLIM")[0] != '0';
    // the slim form's LDS: planes for kBulkSlimGroups groups (mean 1024 + 8 sigma for distinct keys: a region with more
    // abandons the attempt, as one beyond gmax does in the other form), a 2048-slot index, the seen bytes
    const size_t slim_lds = (static_cast<size_t>(a->lds_words) * 8 * (kBulkSlimGroups + 1) + kBulkSubSlots * 4 +
                             static_cast<size_t>(a->lds_seen) * (kBulkSlimGroups + 1) + 15) & ~size_t(15);
    if (sub_form && slim_on && slim_lds <= 52 * 1024) {
      BulkArgs bs = b;
      bs.gmax = kBulkSlimGroups;
      bs.slots = kBulkSubSlots;
      if (wide) hipLaunchKernelGGL((agg_bulk_kernel<true, 512, false, true>), dim3(nregions), dim3(512), slim_lds, st, bs);
      else hipLaunchKernelGGL((agg_bulk_kernel<false, 512, false, true>), dim3(nregions), dim3(512), slim_lds, st, bs);
    } else if (wide) hipLaunchKernelGGL((agg_bulk_kernel<true, 1024, false>), dim3(nregions), dim3(1024), lds, st, b);
    else hipLaunchKernelGGL((agg_bulk_kernel<false, 1024, false>), dim3(nregions), dim3(1024), lds, st, b);
    CRGPU_HIP(hipGetLastError());
  }
  unsigned int tf[4] = {0, 0, 0, 0};
  CRGPU_HIP(hipMemcpyAsync(tf, a->bulk_count.p, 16, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  if (tf[1]) return CRGPU_OK;  // nothing was committed: the groups written so far lie beyond ngroups == 0
  const unsigned int total = tried_ok ? try_groups : tf[0];
  if (key_type == CRGPU_BYTES) {  // the key bytes of group g lie at 16 g: later appends (find-or-insert) continue behind them
    a->ks[0].tail = static_cast<uint64_t>(total) * 16u;
    CRGPU_HIP(hipMemcpyAsync(a->tails_dev.p, &a->ks[0].tail, 8, hipMemcpyHostToDevice, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    a->descriptors_dirty = true;
    a->dense16 = tf[2] == 0;
  }
  if (static_cast<uint64_t>(g0) + total > (kMaxCap >> 1)) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_consume: more than 2^29 groups per GPU");
  a->ngroups = static_cast<int64_t>(g0) + total;
  a->one_row_groups = tried_ok && (g0 == 0 || a->one_row_groups);
  a->seq += static_cast<unsigned long long>(n);
  a->new_ratio = static_cast<double>(total) / static_cast<double>(n);
  a->index_valid = false;  // the groups exist, their index entries do not
  *done = true;
  return CRGPU_OK;
}

// A first input of more than 2^27 rows (the bulk path's regions are sized for that): the (key, value) rows are split by
// key hash into pieces of ~2^26 rows with the router's partition kernels (crgpu_dev_partition: rehash64 % pieces,
// independent of the radix bits) and every piece goes through the bulk path by itself -- equal keys share a piece, so a
// piece's groups are final and are appended behind those of the pieces before it; nothing is merged and the hash index
// is still not touched.  One INT64 key, at most one 8-byte value column, no validity bitmaps.  A piece the bulk path
// declines (few groups, a hot key, an overflowing region) and every piece after it take the incremental path.
// *pos = rows consumed through pieces (all or nothing per piece; the pieces are consumed in full either way).
static crgpu_status crgpu_consume_rows(crgpu_agg *a, std::vector<DCol> &in, int64_t n);
static crgpu_status try_bulk_huge(crgpu_agg *a, const std::vector<DCol> &cols, int64_t n, bool *done) {
  *done = false;
  const char *hm = getenv("CRGPU_BULK_HUGE_MIN");  // (read per call: tests lower it)
  const int64_t huge_min = hm != nullptr && *hm ? atoll(hm) : (int64_t(1) << 27) + 1;
  if (a->ngroups != 0 || a->seq != 0 || n < huge_min || n < 2 * kBulkMinRows || n > 0x7FFFFFF0ll || a->need_any_fix || a->has_computed || !a->dsets.empty())
    return CRGPU_OK;
  if (a->group_cols.size() != 1 || a->types[a->group_cols[0]] != CRGPU_INT64) return CRGPU_OK;
  int vcol = -1;
  for (const AggOpHost &h : a->ops) {
    if (h.kind == ACC_ANY_KEY) continue;
    if (h.filter_col >= 0) return CRGPU_OK;
    if (h.kind != ACC_COUNT_ROWS) {
      if (h.in_type != CRGPU_INT64 && h.in_type != CRGPU_FLOAT64) return CRGPU_OK;
      if (vcol >= 0 && vcol != h.col) return CRGPU_OK;
      vcol = h.col;
    }
  }
  const int kcol = a->group_cols[0];
  if (cols[kcol].valid != nullptr || (vcol >= 0 && cols[vcol].valid != nullptr) || vcol == kcol) return CRGPU_OK;
  int pieces = 2;
  while (n / pieces > (int64_t(1) << 26)) pieces *= 2;
  if (n / pieces < kBulkMinRows) return CRGPU_OK;
  std::vector<int32_t> ptypes = {CRGPU_INT64};
  std::vector<crgpu_dcol> pcols(1);
  std::memset(pcols.data(), 0, sizeof(crgpu_dcol));
  pcols[0].type = CRGPU_INT64;
  pcols[0].data = cols[kcol].data;
  if (vcol >= 0) {
    crgpu_dcol v;
    std::memset(&v, 0, sizeof(v));
    v.type = a->types[vcol];
    v.data = cols[vcol].data;
    ptypes.push_back(v.type);
    pcols.push_back(v);
  }
  const int32_t hash_col = 0;
  crgpu_partitioner *part = nullptr;
  CRGPU_TRY(crgpu_dev_partitioner_create(ptypes.data(), static_cast<int32_t>(ptypes.size()), &hash_col, 1, 0x9E3779B97F4A7C15ull, pieces, &part));
  struct Guard {
    crgpu_partitioner *p;
    ~Guard() { crgpu_dev_partitioner_destroy(p); }
  } guard{part};
  std::vector<int64_t> counts(static_cast<size_t>(pieces), 0);
  crgpu_dev_result pr;
  CRGPU_TRY(crgpu_dev_partition(part, pcols.data(), n, &pr, counts.data()));
  a->try_voided = false;
  // (one allocation, not one per piece; a piece's rows are written before its groups are counted, so room for one
  // piece beyond the 2^29 groups a table may hold)
  const int64_t room = (int64_t(1) << 29) + (int64_t(1) << 27);
  CRGPU_TRY(ensure_groups(a, static_cast<size_t>(n < room ? n : room)));
  bool incremental = false;
  int64_t off = 0;
  for (int s = 0; s < pieces; s++) {
    const int64_t m = counts[static_cast<size_t>(s)];
    if (m == 0) continue;
    std::vector<DCol> piece(cols.size());
    for (auto &d : piece) std::memset(&d, 0, sizeof(d));
    for (size_t c = 0; c < cols.size(); c++) piece[c].type = cols[c].type;
    piece[static_cast<size_t>(kcol)].data = static_cast<const uint8_t *>(pr.cols[0].data) + off * 8;
    if (vcol >= 0) piece[static_cast<size_t>(vcol)].data = static_cast<const uint8_t *>(pr.cols[1].data) + off * 8;
    // (columns the aggregations do not read stay null: neither path touches them -- ANY_NOT_NULL of the key reads the key store)
    bool piece_done = false;
    if (!incremental) CRGPU_TRY(try_bulk(a, piece, m, &piece_done, true));
    if (!piece_done) {
      incremental = true;
      CRGPU_TRY(crgpu_consume_rows(a, piece, m));
    }
    off += m;
  }
  *done = true;
  return CRGPU_OK;
}

static crgpu_status crgpu_dev_agg_consume_impl(crgpu_agg *a, const crgpu_dcol *cols, int64_t n) {
  if (a == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: null handle");
  CRGPU_ENTER(a);
  CRGPU_CHECK_CANCEL(a);
  if (n < 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: negative row count");
  if (n == 0) return CRGPU_OK;
  if (cols == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: null columns");
  const int nc = a->n_phys;  // computed columns (crgpu_dev_agg_create_fused) are not passed in
  std::vector<DCol> in;
  for (int c = 0; c < nc; c++) {
    if (cols[c].type != a->types[c]) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: column type differs from the spec");
    if (cols[c].type == CRGPU_BYTES && cols[c].offsets == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: BYTES column without offsets");
    if (cols[c].type != CRGPU_BYTES && cols[c].data == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: null data pointer");
    in.push_back(to_dcol(cols[c]));
  }
  if (!a->dsets.empty()) {
    // DISTINCT aggregations: the rows each table lets through (first occurrence of the (group, input) tuple among the
    // rows that pass the aggregation's FILTER) as a BOOL column that is the aggregation's FILTER for this call
    hipStream_t st = stream();
    CRGPU_TRY(a->d_rowid.ensure(static_cast<size_t>(n) * 8));
    hipLaunchKernelGGL(iota64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, n, a->d_rowid.as<long long>());
    CRGPU_HIP(hipGetLastError());
    for (auto &dsp : a->dsets) {
      crgpu_agg::DistinctSet &ds = *dsp;
      CRGPU_TRY(ds.mask.ensure(static_cast<size_t>(n)));
      CRGPU_HIP(hipMemsetAsync(ds.mask.p, 0, static_cast<size_t>(n), st));
      const int64_t kSlice = int64_t(1) << 27;  // (the DISTINCT table takes at most 2^28 rows per call)
      for (int64_t lo = 0; lo < n; lo += kSlice) {
        const int64_t len = n - lo < kSlice ? n - lo : kSlice;
        if ((lo & 7) != 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: internal slice offset not byte aligned");
        std::vector<crgpu_dcol> dc;
        auto slice = [&](const DCol &c) {
          crgpu_dcol o;
          std::memset(&o, 0, sizeof(o));
          o.type = c.type;
          if (c.type == CRGPU_BYTES) {
            o.data = c.data;
            o.offsets = c.offsets + lo;
          } else {
            o.data = static_cast<const uint8_t *>(c.data) + lo * type_width(c.type);
          }
          o.valid = c.valid != nullptr ? c.valid + (lo >> 3) : nullptr;
          return o;
        };
        for (int g : a->group_cols) dc.push_back(slice(in[g]));
        dc.push_back(slice(in[ds.col]));
        dc.push_back(slice(DCol{CRGPU_INT64, 0, a->d_rowid.p, nullptr, nullptr}));
        const crgpu_dcol *cols_d = dc.data();
        int64_t n_d = len;
        crgpu_dev_result sel;
        if (ds.filter_col >= 0) {  // FILTER first (aggregators_util.go: the filtering helper wraps the distinct one): NULL = false
          const DCol &fc = in[ds.filter_col];
          const uint8_t *keep = static_cast<const uint8_t *>(fc.data) + lo;
          DevBuf keep_nn;
          if (fc.valid != nullptr) {  // a NULL filter value does not pass: AND the validity bits in
            CRGPU_TRY(keep_nn.ensure(static_cast<size_t>(len)));
            hipLaunchKernelGGL(bool_and_valid_kernel, dim3(grid_for(len, 256)), dim3(256), 0, st, keep, fc.valid + (lo >> 3), len, keep_nn.as<uint8_t>());
            CRGPU_HIP(hipGetLastError());
            keep = keep_nn.as<uint8_t>();
          }
          CRGPU_TRY(crgpu_dev_deselect(ds.desel, cols_d, n_d, keep, &sel));
          CRGPU_HIP(hipStreamSynchronize(st));  // keep_nn is released here
          cols_d = sel.cols;
          n_d = sel.n;
        }
        crgpu_dev_result kept;
        CRGPU_TRY(crgpu_dev_distinct_consume(ds.h, cols_d, n_d, &kept));
        if (kept.n > 0) {
          hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for(kept.n, 256)), dim3(256), 0, st,
                             static_cast<const long long *>(kept.cols[kept.ncols - 1].data), kept.n, ds.mask.as<uint8_t>());
          CRGPU_HIP(hipGetLastError());
        }
      }
    }
    for (auto &dsp : a->dsets) in.push_back(DCol{CRGPU_BOOL, 0, dsp->mask.p, nullptr, nullptr});
  }
  a->one_row_groups = false;
  {
    bool done = false;
    CRGPU_TRY(try_bulk_huge(a, in, n, &done));
    if (done) return CRGPU_OK;
    CRGPU_TRY(try_bulk(a, in, n, &done));
    if (done) return CRGPU_OK;
  }
  return crgpu_consume_rows(a, in, n);
}

// The incremental path: find-or-insert through the hash index, chunk by chunk.
static crgpu_status crgpu_consume_rows(crgpu_agg *a, std::vector<DCol> &in, int64_t n) {
  a->one_row_groups = false;
  a->dense16 = false;
  if (!a->index_valid) {  // groups appended by the bulk path: give them index entries before looking anything up
    const uint64_t want = next_pow2((static_cast<uint64_t>(a->ngroups) + 1) * 2);
    if (want > kMaxCap) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_consume: more than 2^29 groups per GPU");
    CRGPU_TRY(reindex(a, static_cast<uint32_t>(want > a->cap ? want : a->cap)));
    a->index_valid = true;
  }
  // A fresh table gets its per-group arrays sized for the input at once (every later growth of the ~12 arrays is a copy
  // and a wait each: 0.65 ms of the 1.5 ms Q3's final aggregation took); the index still grows with the groups found.
  if (a->ngroups == 0) CRGPU_TRY(ensure_groups(a, static_cast<size_t>(n < (int64_t(1) << 20) ? n : (int64_t(1) << 20))));
  // chunking: a small first chunk discovers the cardinality regime; after that chunks are large (2^27 rows,
  // ~50 B of temporaries per row in the find-or-insert / partitioned paths): every chunk costs one merge of the
  // LDS partial tables and one pass over all gid ranges, so fewer chunks are cheaper
  int64_t pos = 0;
  while (pos < n) {
    // 2^16 rows, then 2^22 (refines the share of new groups, which picks claim-first or look-first), then 2^27
    // (a first chunk that found hardly any new group per row -- a low-cardinality GROUP BY -- needs no refinement: every
    // chunk costs ~0.15 ms of launches and read-backs, a third of the time of a Q1-shaped pass over 2^27 rows)
    // (settled with few groups: chunks of up to 2^30 rows -- the fused LDS pass has no per-row temporaries besides the
    // 4-byte miss list)
    const bool settled = a->seq > 0 && a->new_ratio < 1e-3;
    const int64_t kChunk = settled && a->ngroups <= 4096 ? int64_t(1) << 30 : (a->seq >= (1u << 22) || settled) ? int64_t(1) << 27 : int64_t(1) << 22;
    int64_t len = n - pos < kChunk ? n - pos : kChunk;
    if (a->seq == 0 && len > 16384 && n > (int64_t(1) << 20)) len = 16384;  // (a small input is one chunk: discovery would cost more than it can save)
    std::vector<DCol> chunk = in;
    for (size_t c = 0; c < chunk.size(); c++) {  // (the hidden DISTINCT masks too)
      DCol &d = chunk[c];
      if (d.valid != nullptr && (pos & 7) != 0)
        return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: internal chunk offset not byte aligned");
      if (d.type == CRGPU_BYTES) d.offsets += pos;
      else d.data = static_cast<const uint8_t *>(d.data) + pos * type_width(d.type);
      if (d.valid != nullptr) d.valid += pos >> 3;
    }
    CRGPU_TRY(consume_chunk(a, chunk, len, n - pos));
    pos += len;
  }
  return CRGPU_OK;
}

static crgpu_status ordered_consume(crgpu_agg *a, const crgpu_dcol *cols, int64_t n);
static crgpu_status ordered_finish(crgpu_agg *a, crgpu_dev_result *res);
extern "C" crgpu_status crgpu_dev_agg_consume(crgpu_agg *a, const crgpu_dcol *cols, int64_t n) {
  const crgpu_status s_ = a != nullptr && a->ord ? ordered_consume(a, cols, n) : crgpu_dev_agg_consume_impl(a, cols, n);
  if (s_ == CRGPU_OK && a != nullptr) note_rows(a, n, 0);
  return s_;
}

// Output columns that need no finalisation pass: the accumulator word (or the stored group key) of group g already is
// the value of output row g.  Below kAliasMinGroups the copy costs nothing and keeps the result independent of the
// handle's arrays.
static constexpr int64_t kAliasMinGroups = int64_t(1) << 16;
static bool alias_source(crgpu_agg *a, const AggOpHost &op, const AggOpDev &dev, const void **data, const uint8_t **present) {
  switch (op.kind) {
    case ACC_COUNT_ROWS:
    case ACC_COUNT: *data = dev.w0; *present = nullptr; return true;
    case ACC_SUM_F64:
    case ACC_SUM_I128:
    case ACC_MIN_I64:
    case ACC_MAX_I64: *data = dev.w0; *present = dev.seen; return true;
    case ACC_ANY_KEY:
      if (type_width(a->types[a->group_cols[op.key_idx]]) != 8) return false;
      *data = a->ks[op.key_idx].data.p;
      *present = a->ks[op.key_idx].present.as<uint8_t>();
      return true;
    default: return false;
  }
}

static crgpu_status crgpu_dev_agg_finish_impl(crgpu_agg *a, crgpu_dev_result *res) {
  if (a == nullptr || res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_finish: null argument");
  CRGPU_ENTER(a);
  CRGPU_CHECK_CANCEL(a);
  hipStream_t st = stream();
  const int64_t ng = a->ngroups;
  CRGPU_TRY(upload_descriptors(a, nullptr));
  CRGPU_HIP(hipMemsetAsync(a->err_dev.p, 0, 4, st));
  CRGPU_TRY(a->present_tmp.ensure(static_cast<size_t>(ng > 0 ? ng : 1)));
  if (ng > 0 && ng <= kFinishSmallGroups && a->ops.size() <= static_cast<size_t>(kFinishSmallOps)) {
    FinSmallArgs fs;
    std::memset(&fs, 0, sizeof(fs));
    fs.ops = a->ops_dev.as<AggOpDev>();
    fs.err = a->err_dev.as<unsigned int>();
    fs.ngroups = static_cast<int32_t>(ng);
    for (size_t f = 0; f < a->ops.size(); f++) {
      AggOpHost &op = a->ops[f];
      OutCol &o = a->out[f];
      FinSmallCol &c = fs.c[f];
      o.type = op.out_type;
      o.nullable = true;
      CRGPU_TRY(o.valid.ensure(static_cast<size_t>((ng + 7) / 8 + 1)));
      c.out_valid = o.valid.as<uint8_t>();
      c.out_type = op.out_type;
      if (op.dec128) {
        CRGPU_TRY(o.offsets.ensure(static_cast<size_t>(ng + 1) * 4));
        CRGPU_TRY(o.data.ensure(static_cast<size_t>(ng) * 16));
        c.bytes = 2;
        c.out_offsets = o.offsets.as<uint32_t>();
      } else if (op.out_type == CRGPU_BYTES) {
        const bool any_row = op.kind == ACC_ANY_ROW, stored = any_row || is_minmax_bytes(op.kind);
        // every group's value lies in the store once: its fill level bounds the output
        const uint64_t bound = stored ? op.any_tail : a->ks[op.key_idx].tail;
        if (bound > 0xFFFFFFFFull) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_finish: BYTES output exceeds 4 GiB");
        CRGPU_TRY(o.offsets.ensure(static_cast<size_t>(ng + 1) * 4));
        CRGPU_TRY(o.data.ensure(static_cast<size_t>(bound ? bound : 1)));
        c.bytes = 1;
        c.out_offsets = o.offsets.as<uint32_t>();
        c.b_off = stored ? op.any_off.as<uint64_t>() : a->ks[op.key_idx].b_off.as<uint64_t>();
        c.b_len = stored ? op.any_len.as<uint32_t>() : a->ks[op.key_idx].b_len.as<uint32_t>();
        c.b_buf = stored ? op.any_buf.as<uint8_t>() : a->ks[op.key_idx].b_buf.as<uint8_t>();
        c.b_present = any_row ? nullptr : (stored ? op.seen.as<uint8_t>() : a->ks[op.key_idx].present.as<uint8_t>());
      } else {
        CRGPU_TRY(o.data.ensure(static_cast<size_t>(ng) * type_width(op.out_type)));
        if (op.kind == ACC_ANY_KEY) {
          c.key_data = a->ks[op.key_idx].data.as<uint8_t>();
          c.key_present = a->ks[op.key_idx].present.as<uint8_t>();
        }
      }
      c.out_data = o.data.as<uint8_t>();
    }
    hipLaunchKernelGGL(agg_finish_small_kernel, dim3(static_cast<unsigned int>(a->ops.size())), dim3(256), 0, st, fs);
    CRGPU_HIP(hipGetLastError());
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
  std::vector<AggOpDev> ops(a->ops.size());
  CRGPU_HIP(hipMemcpyAsync(ops.data(), a->ops_dev.p, sizeof(AggOpDev) * ops.size(), hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  std::vector<const void *> alias(a->ops.size(), nullptr);
  const uint8_t *alias_present = nullptr;
  for (size_t f = 0; f < a->ops.size(); f++) {
    AggOpHost &op = a->ops[f];
    OutCol &o = a->out[f];
    o.type = op.out_type;
    o.nullable = true;
    CRGPU_TRY(o.valid.ensure(static_cast<size_t>((ng + 7) / 8 + 1)));
    if (op.out_type == CRGPU_BYTES && op.kind == ACC_ANY_KEY && a->dense16 && ng >= kAliasMinGroups && ng < (int64_t(1) << 28)) {
      // the wide bulk path left group g's 16-byte key at 16 g of the key store: the store IS the output's data, the
      // offsets are multiples of 16 and no key is NULL
      CRGPU_TRY(o.offsets.ensure(static_cast<size_t>(ng + 1) * 4));
      hipLaunchKernelGGL(iota16_kernel, dim3(grid_for(ng + 1, 256)), dim3(256), 0, st, ng + 1, o.offsets.as<uint32_t>());
      CRGPU_HIP(hipGetLastError());
      CRGPU_HIP(hipMemsetAsync(o.valid.p, 0xFF, static_cast<size_t>((ng + 7) / 8), st));
      alias[f] = a->ks[op.key_idx].b_buf.p;
      continue;
    }
    if (op.dec128) {
      if (ng >= (int64_t(1) << 28)) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_finish: BYTES output exceeds 4 GiB");
      CRGPU_TRY(o.offsets.ensure(static_cast<size_t>(ng + 1) * 4));
      CRGPU_TRY(o.data.ensure(static_cast<size_t>(ng > 0 ? ng : 1) * 16));
      hipLaunchKernelGGL(iota16_kernel, dim3(grid_for(ng + 1, 256)), dim3(256), 0, st, ng + 1, o.offsets.as<uint32_t>());
      if (ng > 0)
        hipLaunchKernelGGL(agg_sum128_out_kernel, dim3(grid_for(ng, 256)), dim3(256), 0, st, ops[f].w0, ops[f].w1, ops[f].seen, ng,
                           o.data.as<unsigned long long>(), a->present_tmp.as<uint8_t>());
      CRGPU_HIP(hipGetLastError());
    } else if (op.out_type == CRGPU_BYTES) {
      // ANY_NOT_NULL over a BYTES group key or BYTES column
      // ... or MIN / MAX over a BYTES column (present = a value was stored)
      const bool any_row = op.kind == ACC_ANY_ROW, stored = any_row || is_minmax_bytes(op.kind);
      const uint64_t *off = stored ? op.any_off.as<uint64_t>() : a->ks[op.key_idx].b_off.as<uint64_t>();
      const uint32_t *len = stored ? op.any_len.as<uint32_t>() : a->ks[op.key_idx].b_len.as<uint32_t>();
      const uint8_t *buf = stored ? op.any_buf.as<uint8_t>() : a->ks[op.key_idx].b_buf.as<uint8_t>();
      const uint8_t *pres = any_row ? nullptr : (stored ? op.seen.as<uint8_t>() : a->ks[op.key_idx].present.as<uint8_t>());
      CRGPU_TRY(a->lens_tmp.ensure(static_cast<size_t>(ng > 0 ? ng : 1) * 4));
      CRGPU_TRY(a->offs64_tmp.ensure(static_cast<size_t>(ng + 1) * 8));
      CRGPU_TRY(o.offsets.ensure(static_cast<size_t>(ng + 1) * 4));
      uint64_t total = 0;
      if (ng > 0) {
        hipLaunchKernelGGL(bytes_view_kernel, dim3(grid_for(ng, 256)), dim3(256), 0, st, len, pres, any_row,
                           op.w0.as<unsigned long long>(), ng, a->lens_tmp.as<uint32_t>(), a->present_tmp.as<uint8_t>());
        CRGPU_HIP(hipGetLastError());
      }
      CRGPU_TRY(exclusive_scan_u32(a->lens_tmp.as<uint32_t>(), ng, a->offs64_tmp.as<uint64_t>(), a->scratch, &total));
      if (total > 0xFFFFFFFFull) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_finish: BYTES output exceeds 4 GiB");
      hipLaunchKernelGGL(narrow_u64_u32_kernel, dim3(grid_for(ng + 1, 256)), dim3(256), 0, st, a->offs64_tmp.as<uint64_t>(),
                         ng + 1, o.offsets.as<uint32_t>());
      CRGPU_TRY(o.data.ensure(static_cast<size_t>(total ? total : 1)));
      if (ng > 0) {
        hipLaunchKernelGGL(bytes_copy_groups_kernel, dim3(grid_for(ng, 256)), dim3(256), 0, st, off, buf, ng,
                           o.offsets.as<uint32_t>(), o.data.as<uint8_t>());
        CRGPU_HIP(hipGetLastError());
      }
    } else if (ng >= kAliasMinGroups && type_width(op.out_type) == 8 && alias_source(a, op, ops[f], &alias[f], &alias_present)) {
      // the per-group array already IS the output column (group g is row g): hand it out in place and only build the
      // validity bitmap; values under a NULL are unspecified (as in the reference's vectors)
      if (op.kind == ACC_SUM_I128 && !a->one_row_groups) {
        hipLaunchKernelGGL(agg_sum_check_kernel, dim3(grid_for(ng, 256)), dim3(256), 0, st, ops[f].w0, ops[f].w1, ops[f].seen, ng,
                           a->err_dev.as<unsigned int>());
        CRGPU_HIP(hipGetLastError());
      }
      if (alias_present == nullptr || a->one_row_groups) {  // (one row per group, from inputs without a NULL: all present)
        CRGPU_HIP(hipMemsetAsync(o.valid.p, 0xFF, static_cast<size_t>((ng + 7) / 8), st));
      } else {
        CRGPU_TRY(pack_bitmap(alias_present, ng, o.valid.as<uint8_t>()));
      }
      continue;
    } else {
      CRGPU_TRY(o.data.ensure(static_cast<size_t>(ng > 0 ? ng : 1) * type_width(op.out_type)));
      if (ng > 0) {
        FinalArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.op = ops[f];
        fa.out_type = op.out_type;
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
  for (size_t f = 0; f < a->out.size(); f++) {
    a->views[f] = a->out[f].view();
    if (alias[f] != nullptr) a->views[f].data = alias[f];
  }
  res->n = ng;
  res->ncols = static_cast<int32_t>(a->out.size());
  res->_pad = 0;
  res->cols = a->views.data();
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_agg_finish(crgpu_agg *a, crgpu_dev_result *res) {
  const crgpu_status s_ = a != nullptr && a->ord ? ordered_finish(a, res) : crgpu_dev_agg_finish_impl(a, res);
  if (s_ == CRGPU_OK && a != nullptr) note_rows(a, 0, res->n);
  return s_;
}

// ====================================================================== partially ordered input
// AggregatorSpec.OrderedGroupCols (hash_aggregator_tmpl.go:160-169, 256-306; cases oneOrderedCol ... spilloverAfterEmit,
// hash_aggregator_test.go:104-397): the input arrives sorted on some of the grouping columns, so every group is complete
// once those columns change, and the reference aggregates one run of equal ordered values at a time, emits its groups and
// starts over with an empty table -- the output is ordered on the ordered columns (PartialOrderedVerifier,
// colexectestutils/utils.go:1337-1400: the runs in input order, any order inside a run) and the table never holds more
// than one run.  Here: per call, the LAST place where the ordered columns change is found (64-bit hashes of the ordered
// columns of neighbouring rows are compared: equal values hash equal, so a run is never cut in two; a collision between
// two neighbouring runs only delays their emission, the table still keys on every grouping column); the rows in front
// of it complete every group seen so far: they are consumed, the whole table is finished into a "piece" whose rows are
// put in the order of each group's first input row (a hidden MIN(row number) aggregation; first-row order is run order),
// the table is reset and the rest of the call is consumed.  Pieces are handed out by crgpu_dev_agg_emit_ready (and by
// crgpu_dev_agg_finish for the last ones), one per call, oldest first.  One flush per call at most: the cost of the
// ordering is a hash pass over the ordered columns and, per call with a change, one finish + one gather of the results.
__global__ __launch_bounds__(256) void ord_last_change_kernel(const uint64_t *__restrict__ h, int64_t n, const uint64_t *__restrict__ prev,
                                                              int has_prev, long long *__restrict__ last, uint64_t *__restrict__ prev_out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  long long mine = -1;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t cur = h[i];
    const uint64_t before = i > 0 ? h[i - 1] : (has_prev ? prev[0] : cur);
    if (cur != before) mine = i;  // (i grows along the loop)
    if (i == n - 1) prev_out[0] = cur;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const long long other = __shfl_xor(mine, off);
    mine = other > mine ? other : mine;
  }
  if ((threadIdx.x & 63) == 0 && mine >= 0) atomicMax(last, mine);
}
// (grid_for caps the grid: every kernel below walks its rows with a grid-stride loop)
__global__ __launch_bounds__(256) void ord_iota_kernel(long long base, int64_t n, long long *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = base + i;
}
// dst bit j = src bit (bit_off + j), j < n (dst bytes beyond are zero-filled up to the last one written)
__global__ __launch_bounds__(256) void ord_bitmap_shift_kernel(const uint8_t *__restrict__ src, int64_t bit_off, int64_t n, uint8_t *__restrict__ dst) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t nbytes = (n + 7) >> 3;
  const int s = static_cast<int>(bit_off & 7);
  const int64_t last_src = (bit_off + n - 1) >> 3;
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < nbytes; j += stride) {
    const int64_t b0 = (bit_off >> 3) + j;
    unsigned v = src[b0] >> s;
    if (s != 0 && b0 + 1 <= last_src) v |= static_cast<unsigned>(src[b0 + 1]) << (8 - s);
    dst[j] = static_cast<uint8_t>(v);
  }
}
__global__ __launch_bounds__(256) void ord_mark_kernel(const long long *__restrict__ first, int64_t ng, int64_t span, uint8_t *__restrict__ flags,
                                                       unsigned int *__restrict__ bad) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < ng; g += stride) {
    const long long r = first[g];
    if (r < 0 || r >= span) atomicOr(bad, 1u);
    else flags[r] = 1;
  }
}
// pos[0..ng) = the marked row numbers in ascending order; group g goes to the place of its first row
__global__ __launch_bounds__(256) void ord_rank_kernel(const long long *__restrict__ first, const int32_t *__restrict__ pos, int64_t ng,
                                                       int32_t *__restrict__ perm) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < ng; g += stride) {
    const int32_t want = static_cast<int32_t>(first[g]);
    int64_t lo = 0, hi = ng - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (pos[mid] < want) lo = mid + 1;
      else hi = mid;
    }
    perm[lo] = static_cast<int32_t>(g);
  }
}

static constexpr int64_t kOrdSlice = int64_t(1) << 27;             // rows hashed / split per step (a multiple of 8)
static constexpr int64_t kOrdMaxRun = (int64_t(1) << 31) - kOrdSlice - 8;  // row numbers are gathered through 32-bit indices

static crgpu_status ordered_create(const crgpu_agg_spec *spec, crgpu_agg **out) {
  CRGPU_TRY(ensure_init());
  if (out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: null argument");
  if (spec->n_ordered < 0 || spec->n_ordered > spec->n_group)
    return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_create: n_ordered counts leading entries of group_cols");
  if (spec->n_cols < 0 || spec->n_cols + 1 > kMaxCols || spec->n_aggs < 1 || spec->n_aggs + 1 > kMaxAggs)
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_create: no room for the row-number column of the ordered form");
  std::vector<int32_t> types(spec->types, spec->types + spec->n_cols);
  types.push_back(CRGPU_INT64);
  std::vector<crgpu_agg_desc> aggs(spec->aggs, spec->aggs + spec->n_aggs);
  crgpu_agg_desc first_row;
  std::memset(&first_row, 0, sizeof(first_row));
  first_row.func = CRGPU_MIN;
  first_row.col = spec->n_cols;
  first_row.filter_col = -1;
  aggs.push_back(first_row);
  crgpu_agg_spec in = *spec;
  in.n_cols = spec->n_cols + 1;
  in.types = types.data();
  in.n_aggs = spec->n_aggs + 1;
  in.aggs = aggs.data();
  in.n_ordered = 0;
  crgpu_agg *inner = nullptr;
  CRGPU_TRY(agg_create_impl(&in, 0, nullptr, &inner));
  std::unique_ptr<crgpu_agg> a(new crgpu_agg());
  a->ord.reset(new OrderedState());
  a->ord->inner = inner;  // (destroyed with `a` from here on)
  {
    crgpu_status os = a->open();
    if (os != CRGPU_OK) return os;
  }
  CRGPU_ENTER(a.get());
  OrderedState &o = *a->ord;
  o.n_cols = spec->n_cols;
  o.n_aggs = spec->n_aggs;
  for (int i = 0; i < spec->n_cols; i++) a->types.push_back(spec->types[i]);
  for (int i = 0; i < spec->n_group; i++) a->group_cols.push_back(spec->group_cols[i]);
  for (int i = 0; i < spec->n_ordered; i++) o.ordered_cols.push_back(spec->group_cols[i]);
  a->n_phys = spec->n_cols;
  o.shifted = std::vector<DevBuf>(static_cast<size_t>(spec->n_cols));
  o.views.resize(static_cast<size_t>(spec->n_aggs));
  CRGPU_TRY(o.prev_hash.ensure(16));
  CRGPU_TRY(o.last_b.ensure(16));
  *out = a.release();
  return CRGPU_OK;
}

static crgpu_status ordered_reset(crgpu_agg *a) {
  OrderedState &o = *a->ord;
  CRGPU_TRY(crgpu_dev_agg_reset(o.inner));
  o.rows_since_flush = 0;
  o.has_prev = false;
  o.ready.clear();
  o.handed.reset();
  return CRGPU_OK;
}

// rows [off, off + m) of the caller's columns + their row numbers -> the inner aggregator
static crgpu_status ordered_feed(crgpu_agg *a, const crgpu_dcol *cols, int64_t off, int64_t m) {
  if (m <= 0) return CRGPU_OK;
  OrderedState &o = *a->ord;
  hipStream_t st = stream();
  std::vector<crgpu_dcol> dc(static_cast<size_t>(o.n_cols) + 1);
  for (int c = 0; c < o.n_cols; c++) {
    crgpu_dcol d = cols[c];
    if (d.type == CRGPU_BYTES) d.offsets = d.offsets + off;
    else d.data = static_cast<const uint8_t *>(d.data) + off * type_width(d.type);
    if (d.valid != nullptr) {
      if ((off & 7) == 0) {
        d.valid = d.valid + (off >> 3);
      } else {
        DevBuf &sb = o.shifted[static_cast<size_t>(c)];
        CRGPU_TRY(sb.ensure(static_cast<size_t>((m + 7) / 8 + 8)));
        hipLaunchKernelGGL(ord_bitmap_shift_kernel, dim3(grid_for((m + 7) / 8, 256)), dim3(256), 0, st, cols[c].valid, off, m, sb.as<uint8_t>());
        CRGPU_HIP(hipGetLastError());
        d.valid = sb.as<uint8_t>();
      }
    }
    dc[static_cast<size_t>(c)] = d;
  }
  CRGPU_TRY(o.rowid.ensure(static_cast<size_t>(m) * 8));
  hipLaunchKernelGGL(ord_iota_kernel, dim3(grid_for(m, 256)), dim3(256), 0, st, static_cast<long long>(o.rows_since_flush), m, o.rowid.as<long long>());
  CRGPU_HIP(hipGetLastError());
  crgpu_dcol &rc = dc[static_cast<size_t>(o.n_cols)];
  std::memset(&rc, 0, sizeof(rc));
  rc.type = CRGPU_INT64;
  rc.data = o.rowid.p;
  CRGPU_TRY(crgpu_dev_agg_consume_impl(o.inner, dc.data(), m));
  // (the inner call may return with its kernels still queued: rowid / shifted are rewritten by this stream only)
  o.rows_since_flush += m;
  return CRGPU_OK;
}

// every group in the table is complete: finish, order by first row, keep as a piece, start over
static crgpu_status ordered_flush(crgpu_agg *a) {
  OrderedState &o = *a->ord;
  hipStream_t st = stream();
  if (o.rows_since_flush == 0) return CRGPU_OK;
  crgpu_dev_result r;
  CRGPU_TRY(crgpu_dev_agg_finish_impl(o.inner, &r));
  const int64_t ng = r.n;
  if (ng > 0) {
    if (r.ncols != o.n_aggs + 1) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg: ordered form lost its row-number column");
    const long long *first = static_cast<const long long *>(r.cols[o.n_aggs].data);
    const int64_t span = o.rows_since_flush;
    CRGPU_TRY(o.flags.ensure(static_cast<size_t>(span)));
    CRGPU_TRY(o.pos.ensure(static_cast<size_t>(ng) * 4));
    CRGPU_TRY(o.perm.ensure(static_cast<size_t>(ng) * 4));
    CRGPU_HIP(hipMemsetAsync(o.flags.p, 0, static_cast<size_t>(span), st));
    CRGPU_HIP(hipMemsetAsync(o.last_b.p, 0, 16, st));
    unsigned int *bad = o.last_b.as<unsigned int>() + 2;
    hipLaunchKernelGGL(ord_mark_kernel, dim3(grid_for(ng, 256)), dim3(256), 0, st, first, ng, span, o.flags.as<uint8_t>(), bad);
    CRGPU_HIP(hipGetLastError());
    int64_t marked = 0;
    CRGPU_TRY(select_flagged(o.flags.as<uint8_t>(), span, o.pos.as<int32_t>(), o.scratch, &marked));
    unsigned int bad_h = 0;
    CRGPU_HIP(hipMemcpyAsync(&bad_h, bad, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (bad_h != 0 || marked != ng) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg: first rows of the groups are not distinct row numbers");
    hipLaunchKernelGGL(ord_rank_kernel, dim3(grid_for(ng, 256)), dim3(256), 0, st, first, o.pos.as<int32_t>(), ng, o.perm.as<int32_t>());
    CRGPU_HIP(hipGetLastError());
    std::unique_ptr<OrderedPiece> piece(new OrderedPiece());
    piece->n = ng;
    piece->cols = std::vector<OutCol>(static_cast<size_t>(o.n_aggs));
    for (int c = 0; c < o.n_aggs; c++)
      CRGPU_TRY(materialize(to_dcol(r.cols[c]), o.perm.as<int32_t>(), ng, false, piece->cols[static_cast<size_t>(c)], o.scratch));
    o.ready.push_back(std::move(piece));
  }
  CRGPU_TRY(crgpu_dev_agg_reset(o.inner));
  o.rows_since_flush = 0;
  return CRGPU_OK;
}

static crgpu_status ordered_consume(crgpu_agg *a, const crgpu_dcol *cols, int64_t n) {
  CRGPU_ENTER(a);
  CRGPU_CHECK_CANCEL(a);
  if (n < 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: negative row count");
  if (n == 0) return CRGPU_OK;
  if (cols == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: null columns");
  OrderedState &o = *a->ord;
  for (int c = 0; c < o.n_cols; c++) {
    if (cols[c].type != a->types[static_cast<size_t>(c)]) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: column type differs from the spec");
    if (cols[c].type == CRGPU_BYTES && cols[c].offsets == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: BYTES column without offsets");
    if (cols[c].type != CRGPU_BYTES && cols[c].data == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_consume: null data pointer");
  }
  hipStream_t st = stream();
  for (int64_t lo = 0; lo < n; lo += kOrdSlice) {
    const int64_t len = n - lo < kOrdSlice ? n - lo : kOrdSlice;
    long long b = -1;
    if (o.ordered_cols.empty()) {
      b = -1;  // no ordered column: one run
    } else {
      KeyCols kc;
      std::memset(&kc, 0, sizeof(kc));
      kc.n = static_cast<int32_t>(o.ordered_cols.size());
      for (int i = 0; i < kc.n; i++) {
        DCol d = to_dcol(cols[o.ordered_cols[static_cast<size_t>(i)]]);
        if (d.type == CRGPU_BYTES) d.offsets += lo;
        else d.data = static_cast<const uint8_t *>(d.data) + lo * type_width(d.type);
        if (d.valid != nullptr) d.valid += lo >> 3;  // (lo is a multiple of 2^27)
        kc.c[i] = d;
      }
      CRGPU_TRY(o.hashes.ensure(static_cast<size_t>(len) * 8));
      CRGPU_TRY(launch_hash(kc, len, 1, 64, 0, o.hashes.p));
      const long long minus1 = -1;
      CRGPU_HIP(hipMemcpyAsync(o.last_b.p, &minus1, 8, hipMemcpyHostToDevice, st));
      const int64_t want = (len + 255) / 256;
      const int grid = static_cast<int>(want < 4096 ? want : 4096);
      hipLaunchKernelGGL(ord_last_change_kernel, dim3(grid), dim3(256), 0, st, o.hashes.as<uint64_t>(), len,
                         o.prev_hash.as<uint64_t>() + o.prev_slot, o.has_prev ? 1 : 0, o.last_b.as<long long>(),
                         o.prev_hash.as<uint64_t>() + (o.prev_slot ^ 1));
      CRGPU_HIP(hipGetLastError());
      CRGPU_HIP(hipMemcpyAsync(&b, o.last_b.p, 8, hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipStreamSynchronize(st));
      o.prev_slot ^= 1;
      o.has_prev = true;
    }
    if (b >= 0) {
      CRGPU_TRY(ordered_feed(a, cols, lo, b));
      CRGPU_TRY(ordered_flush(a));
      CRGPU_TRY(ordered_feed(a, cols, lo + b, len - b));
    } else {
      if (o.rows_since_flush + len > kOrdMaxRun)
        return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_agg_consume: more than 2^31 rows without a change in the ordered columns: plan the unordered form");
      CRGPU_TRY(ordered_feed(a, cols, lo, len));
    }
  }
  return CRGPU_OK;
}

static crgpu_status ordered_hand_out(crgpu_agg *a, crgpu_dev_result *res) {
  OrderedState &o = *a->ord;
  o.handed.reset();
  res->n = 0;
  res->ncols = o.n_aggs;
  res->_pad = 0;
  res->cols = o.views.data();
  if (o.ready.empty()) {
    for (auto &v : o.views) std::memset(&v, 0, sizeof(v));
    return CRGPU_OK;
  }
  // (a piece's buffers go back to the pool when it is dropped: nothing queued may still read them)
  o.handed = std::move(o.ready.front());
  o.ready.pop_front();
  for (int c = 0; c < o.n_aggs; c++) o.views[static_cast<size_t>(c)] = o.handed->cols[static_cast<size_t>(c)].view();
  res->n = o.handed->n;
  return CRGPU_OK;
}

static crgpu_status ordered_finish(crgpu_agg *a, crgpu_dev_result *res) {
  if (res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_finish: null argument");
  CRGPU_ENTER(a);
  CRGPU_CHECK_CANCEL(a);
  CRGPU_TRY(ordered_flush(a));
  a->ord->has_prev = false;
  return ordered_hand_out(a, res);
}

extern "C" crgpu_status crgpu_dev_agg_emit_ready(crgpu_agg *a, crgpu_dev_result *res) {
  if (a == nullptr || res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_emit_ready: null argument");
  CRGPU_ENTER(a);
  CRGPU_CHECK_CANCEL(a);
  if (!a->ord) {  // unordered input: nothing is complete before the input ends
    res->n = 0;
    res->ncols = 0;  // (no column descriptors to look at)
    res->_pad = 0;
    res->cols = nullptr;
    return CRGPU_OK;
  }
  const crgpu_status s_ = ordered_hand_out(a, res);
  if (s_ == CRGPU_OK) note_rows(a, 0, res->n);
  return s_;
}

// Rows consumed so far whose groups have not been handed out: with ordered grouping columns the rows behind the last
// flush (the reference's hash aggregator drops exactly the others from its input tracking when it has emitted a run's
// groups: resetBucketsAndTrackingState, hash_aggregator_tmpl.go:355-378); without, every row since the last reset.
extern "C" crgpu_status crgpu_dev_agg_pending_rows(crgpu_agg *a, int64_t *rows) {
  if (a == nullptr || rows == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_agg_pending_rows: null argument");
  *rows = a->ord ? a->ord->rows_since_flush : static_cast<int64_t>(a->seq);
  return CRGPU_OK;
}

