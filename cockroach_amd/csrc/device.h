// device.h -- device-side building blocks: column access, the reference's
// memhash family (bit-exact), key equality, wave/block primitives.
#pragma once

#include "common.h"

namespace crgpu {

// ------------------------------------------------------------------ column access
__device__ __forceinline__ bool col_is_null(const DCol &c, int64_t i) {
  // coldata.Nulls.NullAt (col/coldata/nulls.go:184-201): bit = 1 => present
  return c.valid != nullptr && !((c.valid[i >> 3] >> (i & 7)) & 1);
}
__device__ __forceinline__ int64_t col_get_int(const DCol &c, int64_t i) {
  switch (c.type) {
    case CRGPU_INT16: return reinterpret_cast<const int16_t *>(c.data)[i];
    case CRGPU_INT32: return reinterpret_cast<const int32_t *>(c.data)[i];
    default: return reinterpret_cast<const int64_t *>(c.data)[i];
  }
}
__device__ __forceinline__ double col_get_f64(const DCol &c, int64_t i) {
  return reinterpret_cast<const double *>(c.data)[i];
}
__device__ __forceinline__ uint32_t col_get_bool(const DCol &c, int64_t i) {
  return reinterpret_cast<const uint8_t *>(c.data)[i] != 0;
}

// ------------------------------------------------------------------ memhash (colexechash/hash.go)
static constexpr uint64_t kM1 = 16877499708836156737ULL;  // hash.go:29
static constexpr uint64_t kM2 = 2820277070424839065ULL;   // hash.go:30
static constexpr uint64_t kM3 = 9497967016996688599ULL;   // hash.go:31
static constexpr uint64_t kM4 = 15839092249703872147ULL;  // hash.go:32
static constexpr uint64_t kC0 = 33054211828000289ULL;     // hash.go:26
static constexpr uint64_t kC1 = 23344194077549503ULL;     // hash.go:27

__device__ __forceinline__ uint64_t rotl31(uint64_t x) { return (x << 31) | (x >> 33); }
__device__ __forceinline__ uint64_t mix(uint64_t h) { return rotl31(h * kM1) * kM2; }
__device__ __forceinline__ uint64_t fin(uint64_t h) {
  h ^= h >> 29;
  h *= kM3;
  h ^= h >> 32;
  return h;
}
// memhash64: hash.go:131-139 (hashKey[0] = 1)
__device__ __forceinline__ uint64_t memhash64(uint64_t v, uint64_t seed) { return fin(mix((seed + 8) ^ v)); }

// little-endian loads at arbitrary alignment (byte-granular global loads)
__device__ __forceinline__ uint64_t ld_u8(const uint8_t *p) { return *p; }
__device__ __forceinline__ uint64_t ld_le32(const uint8_t *p) {
  return ld_u8(p) | (ld_u8(p + 1) << 8) | (ld_u8(p + 2) << 16) | (ld_u8(p + 3) << 24);
}
__device__ __forceinline__ uint64_t ld_le64(const uint8_t *p) { return ld_le32(p) | (ld_le32(p + 4) << 32); }

// memhash: hash.go:73-129
__device__ inline uint64_t memhash(const uint8_t *p, uint64_t seed, uint64_t s) {
  uint64_t h = seed + s;
  if (s > 32) {
    uint64_t v1 = h, v2 = seed, v3 = seed, v4 = seed;
    while (s >= 32) {
      v1 ^= ld_le64(p);      v1 = rotl31(v1 * kM1) * kM2;
      v2 ^= ld_le64(p + 8);  v2 = rotl31(v2 * kM2) * kM3;
      v3 ^= ld_le64(p + 16); v3 = rotl31(v3 * kM3) * kM4;
      v4 ^= ld_le64(p + 24); v4 = rotl31(v4 * kM4) * kM1;
      p += 32;
      s -= 32;
    }
    h = v1 ^ v2 ^ v3 ^ v4;
  }
  if (s == 0) {
  } else if (s < 4) {
    h ^= ld_u8(p);
    h ^= ld_u8(p + (s >> 1)) << 8;
    h ^= ld_u8(p + s - 1) << 16;
    h = mix(h);
  } else if (s <= 8) {
    h ^= ld_le32(p);
    h ^= ld_le32(p + s - 4) << 32;
    h = mix(h);
  } else if (s <= 16) {
    h ^= ld_le64(p);
    h = mix(h);
    h ^= ld_le64(p + s - 8);
    h = mix(h);
  } else {
    h ^= ld_le64(p);
    h = mix(h);
    h ^= ld_le64(p + 8);
    h = mix(h);
    h ^= ld_le64(p + s - 16);
    h = mix(h);
    h ^= ld_le64(p + s - 8);
    h = mix(h);
  }
  return fin(h);
}

// Float customizer + f64hash: overloads_hash.go:94-107, hash.go:153-163.
__device__ __forceinline__ uint64_t float_hash(double f, uint64_t h) {
  if (f != f) f = 0;
  if (f == 0) return kC1 * (kC0 ^ h);
  return memhash64(static_cast<uint64_t>(__double_as_longlong(f)), h);  // memhash(8 bytes) == memhash64
}

// One step of rehash (hash_utils.eg.go:39): fold column value at row i into p.
// NULL leaves p unchanged (:62-64).
__device__ __forceinline__ uint64_t rehash_one(const DCol &c, int64_t i, uint64_t p) {
  if (col_is_null(c, i)) return p;
  switch (c.type) {
    case CRGPU_BOOL: return p * 31 + col_get_bool(c, i);
    case CRGPU_INT16:
    case CRGPU_INT32:
    case CRGPU_INT64: return memhash64(static_cast<uint64_t>(col_get_int(c, i)), p);
    case CRGPU_FLOAT64: return float_hash(col_get_f64(c, i), p);
    case CRGPU_BYTES: {
      uint32_t a = c.offsets[i], b = c.offsets[i + 1];
      return memhash(reinterpret_cast<const uint8_t *>(c.data) + a, p, b - a);
    }
    default: return p;
  }
}

// 64-bit (router) and 32-bit (hash table: truncated after every column) row hashes.
__device__ __forceinline__ uint64_t hash_row64(const KeyCols &k, int64_t i, uint64_t init) {
  uint64_t p = init;
  for (int c = 0; c < k.n; c++) p = rehash_one(k.c[c], i, p);
  return p;
}
__device__ __forceinline__ uint32_t hash_row32(const KeyCols &k, int64_t i, uint32_t init) {
  uint32_t p = init;
  for (int c = 0; c < k.n; c++) p = static_cast<uint32_t>(rehash_one(k.c[c], i, p));
  return p;
}

// ------------------------------------------------------------------ key equality
// _CHECK_COL_BODY / _ASSIGN_NE (hashtable_tmpl.go:96-223; float total order
// hashtable_full_default.eg.go:4073-4091).  Returns true iff all key columns
// are equal.  NULL handling: with null_eq both-NULL is equal, otherwise any
// NULL makes the tuples differ.
__device__ inline bool keys_equal(const KeyCols &a, int64_t i, const KeyCols &b, int64_t j, bool null_eq) {
  for (int c = 0; c < a.n; c++) {
    const DCol &x = a.c[c];
    const DCol &y = b.c[c];
    bool xn = col_is_null(x, i), yn = col_is_null(y, j);
    if (xn || yn) {
      if (null_eq && xn && yn) continue;
      return false;
    }
    switch (x.type) {
      case CRGPU_BOOL:
        if (col_get_bool(x, i) != col_get_bool(y, j)) return false;
        break;
      case CRGPU_INT16:
      case CRGPU_INT32:
      case CRGPU_INT64:
        if (col_get_int(x, i) != col_get_int(y, j)) return false;
        break;
      case CRGPU_FLOAT64: {
        double p = col_get_f64(x, i), q = col_get_f64(y, j);
        if (!(p == q || (p != p && q != q))) return false;
        break;
      }
      case CRGPU_BYTES: {
        uint32_t xa = x.offsets[i], xl = x.offsets[i + 1] - xa;
        uint32_t ya = y.offsets[j], yl = y.offsets[j + 1] - ya;
        if (xl != yl) return false;
        const uint8_t *px = reinterpret_cast<const uint8_t *>(x.data) + xa;
        const uint8_t *py = reinterpret_cast<const uint8_t *>(y.data) + ya;
        for (uint32_t t = 0; t < xl; t++)
          if (px[t] != py[t]) return false;
        break;
      }
      default: return false;
    }
  }
  return true;
}
// keys_equal of probe row i against K build rows at once, column by column: the K loads of a column go out together
// where K calls of keys_equal wait for each candidate's columns in turn.  eq[t] must come in as "candidate t is to be
// tested"; nothing is loaded for the others.
template <int K>
__device__ __forceinline__ void keys_equal_chunk(const KeyCols &a, int64_t i, const KeyCols &b, const uint32_t (&rr)[K],
                                                 bool null_eq, bool (&eq)[K]) {
  for (int c = 0; c < a.n; c++) {
    const DCol &x = a.c[c];
    const DCol &y = b.c[c];
    const bool xn = col_is_null(x, i);
    bool yn[K];
#pragma unroll
    for (int t = 0; t < K; t++) yn[t] = eq[t] && col_is_null(y, rr[t]);
    bool same[K];
    switch (x.type) {
      case CRGPU_BOOL: {
        const uint32_t pv = col_get_bool(x, i);
#pragma unroll
        for (int t = 0; t < K; t++) same[t] = eq[t] && pv == col_get_bool(y, rr[t]);
        break;
      }
      case CRGPU_INT16:
      case CRGPU_INT32:
      case CRGPU_INT64: {
        const int64_t pv = col_get_int(x, i);
        int64_t bv[K];
        switch (y.type) {  // hoisted out of the candidate loop so that its loads are issued back to back
          case CRGPU_INT16:
#pragma unroll
            for (int t = 0; t < K; t++) bv[t] = eq[t] ? reinterpret_cast<const int16_t *>(y.data)[rr[t]] : 0;
            break;
          case CRGPU_INT32:
#pragma unroll
            for (int t = 0; t < K; t++) bv[t] = eq[t] ? reinterpret_cast<const int32_t *>(y.data)[rr[t]] : 0;
            break;
          default:
#pragma unroll
            for (int t = 0; t < K; t++) bv[t] = eq[t] ? reinterpret_cast<const int64_t *>(y.data)[rr[t]] : 0;
            break;
        }
#pragma unroll
        for (int t = 0; t < K; t++) same[t] = pv == bv[t];
        break;
      }
      case CRGPU_FLOAT64: {
        const double pv = col_get_f64(x, i);
#pragma unroll
        for (int t = 0; t < K; t++) {
          const double q = eq[t] ? col_get_f64(y, rr[t]) : 0.0;
          same[t] = pv == q || (pv != pv && q != q);
        }
        break;
      }
      case CRGPU_BYTES: {
#pragma unroll
        for (int t = 0; t < K; t++) {
          same[t] = false;
          if (eq[t] && !xn && !yn[t]) {
            const uint32_t xa = x.offsets[i], xl = x.offsets[i + 1] - xa;
            const uint32_t ya = y.offsets[rr[t]], yl = y.offsets[rr[t] + 1] - ya;
            bool s = xl == yl;
            const uint8_t *px = reinterpret_cast<const uint8_t *>(x.data) + xa;
            const uint8_t *py = reinterpret_cast<const uint8_t *>(y.data) + ya;
            for (uint32_t u = 0; s && u < xl; u++) s = px[u] == py[u];
            same[t] = s;
          }
        }
        break;
      }
      default:
#pragma unroll
        for (int t = 0; t < K; t++) same[t] = false;
        break;
    }
#pragma unroll
    for (int t = 0; t < K; t++) eq[t] = eq[t] && ((xn || yn[t]) ? (null_eq && xn && yn[t]) : same[t]);
  }
}
__device__ __forceinline__ bool any_key_null(const KeyCols &k, int64_t i) {
  for (int c = 0; c < k.n; c++)
    if (col_is_null(k.c[c], i)) return true;
  return false;
}

// ------------------------------------------------------------------ wave / block primitives (wave = 64)
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
__device__ __forceinline__ uint32_t popc64(uint64_t x) { return __popcll(x); }
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t rank_in_mask(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0));
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane_id() >= static_cast<uint32_t>(d)) v += t;
  }
  return v;
}

// murmur3 fmix64: the internal slot hash of the packed-key tables (free choice:
// only RESULTS must match the reference, not its bucket numbers).
__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

// Packed key: ints sign-extended to 64 bits (mixed widths compare equal,
// hashtable_tmpl.go:275-281); floats by bits with NaN and -0 canonicalised so
// that NaN == NaN and +0 == -0 (hashtable_full_default.eg.go:4073-4091).
__device__ __forceinline__ uint64_t pack_key(const DCol &c, int64_t i) {
  switch (c.type) {
    case CRGPU_BOOL: return col_get_bool(c, i);
    case CRGPU_FLOAT64: {
      double f = col_get_f64(c, i);
      if (f != f) return 0x7FF8000000000000ull;
      if (f == 0) return 0;
      return static_cast<uint64_t>(__double_as_longlong(f));
    }
    default: return static_cast<uint64_t>(col_get_int(c, i));
  }
}

// ------------------------------------------------------------------ packed join keys
// The packed / radix / exact-chained join tables compare ONE 64-bit word per row.  A single fixed-width key column is
// its pack_key() value.  Several integer (or BOOL) key columns are range-compressed into one word: the build side's
// [lo, hi] of every column is measured once, column c contributes (v - lo[c]) << shift[c], and the key is usable when
// the columns' bit widths add up to <= 64 (TPC-H Q9: (l_partkey, l_suppkey) needs 21 + 17 bits at SF10).  A probe value
// outside the build side's range cannot equal any build value: such a row has no key, exactly like a NULL.
static constexpr int kMaxPackCols = 4;
struct KeyPack {
  int32_t n;
  int32_t _pad;
  DCol c[kMaxPackCols];
  int64_t lo[kMaxPackCols];
  uint64_t range[kMaxPackCols];  // hi - lo
  uint32_t shift[kMaxPackCols];
};
__device__ __forceinline__ int64_t col_get_ival(const DCol &c, int64_t i) {
  return c.type == CRGPU_BOOL ? static_cast<int64_t>(col_get_bool(c, i)) : col_get_int(c, i);
}
// false: the row has no key (a NULL key column, or a value outside the build side's range) and matches nothing
__device__ __forceinline__ bool key_pack(const KeyPack &kp, int64_t i, uint64_t &k) {
  if (kp.n == 1) {
    if (col_is_null(kp.c[0], i)) return false;
    k = pack_key(kp.c[0], i);
    return true;
  }
  uint64_t v = 0;
  bool ok = true;
#pragma unroll
  for (int c = 0; c < kMaxPackCols; c++) {
    if (c < kp.n) {
      if (col_is_null(kp.c[c], i)) {
        ok = false;
      } else {
        const uint64_t d = static_cast<uint64_t>(col_get_ival(kp.c[c], i)) - static_cast<uint64_t>(kp.lo[c]);  // wraps above range when v < lo
        ok = ok && d <= kp.range[c];
        v |= d << kp.shift[c];
      }
    }
  }
  k = v;
  return ok;
}
// column c of a packed multi-column key (emitting a key column from the key word)
__device__ __forceinline__ uint64_t key_unpack(const KeyPack &kp, int c, uint64_t k) {
  if (kp.n == 1) return k;
  const uint64_t r = kp.range[c];
  const uint64_t mask = r == ~0ull ? ~0ull : ((2ull << (63 - __builtin_clzll(r | 1ull))) - 1ull);
  return ((k >> kp.shift[c]) & mask) + static_cast<uint64_t>(kp.lo[c]);
}

}  // namespace crgpu
