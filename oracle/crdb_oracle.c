/*
 * crdb_oracle.c -- TEST INFRASTRUCTURE ONLY (see crdb_oracle.h).
 *
 * Single-threaded, batch-at-a-time CPU restatement of the reference's
 * colexechash / colexecjoin / hash aggregator algorithms, written from the
 * reference's Go sources (cited per function, paths relative to pkg/).
 * It deliberately keeps the reference's data structures (First/Next/Same/
 * Visited chains, ToCheck/ToCheckID/differs loops, 1024-tuple probe batches,
 * 4096-tuple aggregation buffers, per-group accumulators) so that it is both
 * the parity oracle and an honest stand-in for the reference's CPU cost.
 */
#include "crdb_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* small helpers                                                             */
/* ------------------------------------------------------------------------ */

static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) { fprintf(stderr, "crdb_oracle: out of memory (%zu)\n", n); abort(); }
  return p;
}
static void *xcalloc(size_t n, size_t sz) {
  void *p = calloc(n ? n : 1, sz ? sz : 1);
  if (!p) { fprintf(stderr, "crdb_oracle: out of memory\n"); abort(); }
  return p;
}
static void *xrealloc(void *p, size_t n) {
  p = realloc(p, n ? n : 1);
  if (!p) { fprintf(stderr, "crdb_oracle: out of memory (%zu)\n", n); abort(); }
  return p;
}

static inline int type_width(int t) {
  switch (t) {
  case ORA_BOOL: return 1;
  case ORA_INT16: return 2;
  case ORA_INT32: return 4;
  case ORA_INT64: return 8;
  case ORA_FLOAT64: return 8;
  default: return 0;
  }
}
static inline int is_int_type(int t) { return t == ORA_INT16 || t == ORA_INT32 || t == ORA_INT64; }

/* coldata.Nulls.NullAt (col/coldata/nulls.go:184-201): bit=1 => valid. */
static inline int col_is_null(const ora_col *c, int64_t i) {
  return c->valid != NULL && !((c->valid[i >> 3] >> (i & 7)) & 1);
}
static inline int64_t col_get_int(const ora_col *c, int64_t i) {
  switch (c->type) {
  case ORA_INT16: return ((const int16_t *)c->data)[i];
  case ORA_INT32: return ((const int32_t *)c->data)[i];
  default: return ((const int64_t *)c->data)[i];
  }
}
static inline double col_get_f64(const ora_col *c, int64_t i) { return ((const double *)c->data)[i]; }
static inline uint8_t col_get_bool(const ora_col *c, int64_t i) { return ((const uint8_t *)c->data)[i] != 0; }
static inline const uint8_t *col_get_bytes(const ora_col *c, int64_t i, uint32_t *len) {
  uint32_t a = c->offsets[i], b = c->offsets[i + 1];
  *len = b - a;
  return (const uint8_t *)c->data + a;
}

/* ------------------------------------------------------------------------ */
/* growable vector (stands in for coldata.Vec inside AppendOnlyBufferedBatch) */
/* ------------------------------------------------------------------------ */

typedef struct vec {
  int type;
  int64_t n, cap;
  uint8_t *data;   /* fixed-width values */
  uint8_t *valid;  /* validity bitmap (cap bits), bit=1 valid */
  int has_nulls;   /* Nulls.maybeHasNulls */
  uint32_t *off;   /* BYTES: cap+1 offsets */
  uint8_t *buf;    /* BYTES data */
  int64_t buflen, bufcap;
} vec;

static void vec_init(vec *v, int type) {
  memset(v, 0, sizeof(*v));
  v->type = type;
}
static void vec_free(vec *v) {
  free(v->data); free(v->valid); free(v->off); free(v->buf);
  memset(v, 0, sizeof(*v));
}
static void vec_reserve(vec *v, int64_t need) {
  if (need <= v->cap) return;
  int64_t ncap = v->cap ? v->cap : 64;
  while (ncap < need) ncap *= 2;
  if (v->type == ORA_BYTES) {
    v->off = (uint32_t *)xrealloc(v->off, (size_t)(ncap + 1) * 4);
    if (v->cap == 0) v->off[0] = 0;
  } else {
    v->data = (uint8_t *)xrealloc(v->data, (size_t)ncap * type_width(v->type));
  }
  size_t ob = (size_t)((v->cap + 7) / 8), nb = (size_t)((ncap + 7) / 8);
  v->valid = (uint8_t *)xrealloc(v->valid, nb);
  memset(v->valid + ob, 0xff, nb - ob); /* trailing bits set to 1 (nulls.go:35-41) */
  v->cap = ncap;
}
static inline void vec_set_null(vec *v, int64_t i) {
  v->valid[i >> 3] &= (uint8_t)~(1u << (i & 7));
  v->has_nulls = 1;
}
static inline void vec_set_valid(vec *v, int64_t i) { v->valid[i >> 3] |= (uint8_t)(1u << (i & 7)); }

static void vec_view(const vec *v, ora_col *c) {
  c->type = v->type;
  c->_pad = 0;
  c->data = v->type == ORA_BYTES ? (const void *)v->buf : (const void *)v->data;
  c->valid = v->has_nulls ? v->valid : NULL;
  c->offsets = v->off;
}

/* Append one row src[i] (or a NULL) at the end of v.  This is the per-row body
 * of Vec.Append / Vec.Copy with a selection vector (col/coldata/vec_tmpl.go:143,187). */
static inline void vec_push_from(vec *v, const ora_col *src, int64_t i) {
  vec_reserve(v, v->n + 1);
  int64_t d = v->n;
  int null = col_is_null(src, i);
  if (v->type == ORA_BYTES) {
    uint32_t len = 0;
    const uint8_t *p = NULL;
    if (!null) p = col_get_bytes(src, i, &len);
    if (v->buflen + len > v->bufcap) {
      int64_t nc = v->bufcap ? v->bufcap : 256;
      while (nc < v->buflen + len) nc *= 2;
      v->buf = (uint8_t *)xrealloc(v->buf, (size_t)nc);
      v->bufcap = nc;
    }
    if (len) memcpy(v->buf + v->buflen, p, len);
    v->buflen += len;
    v->off[d + 1] = (uint32_t)v->buflen;
  } else {
    int w = type_width(v->type);
    if (src->type == v->type) {
      memcpy(v->data + (size_t)d * w, (const uint8_t *)src->data + (size_t)i * w, (size_t)w);
    } else {
      /* widening int copy (only used when output type differs; not on the reference path) */
      int64_t x = col_get_int(src, i);
      if (v->type == ORA_INT64) ((int64_t *)v->data)[d] = x;
      else if (v->type == ORA_INT32) ((int32_t *)v->data)[d] = (int32_t)x;
      else ((int16_t *)v->data)[d] = (int16_t)x;
    }
  }
  if (null) vec_set_null(v, d); else vec_set_valid(v, d);
  v->n++;
}
static inline void vec_push_null(vec *v) {
  vec_reserve(v, v->n + 1);
  int64_t d = v->n;
  if (v->type == ORA_BYTES) v->off[d + 1] = (uint32_t)v->buflen;
  else memset(v->data + (size_t)d * type_width(v->type), 0, (size_t)type_width(v->type));
  vec_set_null(v, d);
  v->n++;
}

/* AppendOnlyBufferedBatch.AppendTuples (colexecutils/utils.go:204): copy rows
 * [start,end) of a batch (through its selection vector, if any). */
static void vec_append_range(vec *v, const ora_col *src, const int64_t *sel, int64_t start, int64_t end) {
  vec_reserve(v, v->n + (end - start));
  for (int64_t i = start; i < end; i++) vec_push_from(v, src, sel ? sel[i] : i);
}

/* ------------------------------------------------------------------------ */
/* result object                                                             */
/* ------------------------------------------------------------------------ */

struct ora_result {
  int ncols;
  vec *cols;
};

static ora_result *result_new(int ncols, const int *types) {
  ora_result *r = (ora_result *)xcalloc(1, sizeof(*r));
  r->ncols = ncols;
  r->cols = (vec *)xcalloc((size_t)ncols, sizeof(vec));
  for (int i = 0; i < ncols; i++) {
    vec_init(&r->cols[i], types[i]);
    vec_reserve(&r->cols[i], 1);
  }
  return r;
}
int32_t ora_result_ncols(const ora_result *r) { return r->ncols; }
int64_t ora_result_nrows(const ora_result *r) { return r->ncols ? r->cols[0].n : 0; }
void ora_result_col(const ora_result *r, int32_t i, ora_col *out) { vec_view(&r->cols[i], out); }
void ora_result_free(ora_result *r) {
  if (!r) return;
  for (int i = 0; i < r->ncols; i++) vec_free(&r->cols[i]);
  free(r->cols);
  free(r);
}

/* ------------------------------------------------------------------------ */
/* memhash family: colexechash/hash.go                                       */
/* ------------------------------------------------------------------------ */

#define M1 16877499708836156737ULL /* hash.go:29 */
#define M2 2820277070424839065ULL  /* hash.go:30 */
#define M3 9497967016996688599ULL  /* hash.go:31 */
#define M4 15839092249703872147ULL /* hash.go:32 */
#define C0 33054211828000289ULL    /* hash.go:26 (ptrSize 8) */
#define C1 23344194077549503ULL    /* hash.go:27 */
/* hashKey[0..3] = 1 (hash.go:36-42) */

static inline uint64_t rotl31(uint64_t x) { return (x << 31) | (x >> 33); } /* hash.go:144 */
static inline uint64_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* memhash: hash.go:73-129 */
uint64_t ora_memhash(const void *pv, uint64_t seed, uint64_t s) {
  const uint8_t *p = (const uint8_t *)pv;
  uint64_t h = seed + s; /* seed + s*hashKey[0] */
tail:
  if (s == 0) {
  } else if (s < 4) {
    h ^= (uint64_t)p[0];
    h ^= (uint64_t)p[s >> 1] << 8;
    h ^= (uint64_t)p[s - 1] << 16;
    h = rotl31(h * M1) * M2;
  } else if (s <= 8) {
    h ^= rd32(p);
    h ^= rd32(p + s - 4) << 32;
    h = rotl31(h * M1) * M2;
  } else if (s <= 16) {
    h ^= rd64(p);
    h = rotl31(h * M1) * M2;
    h ^= rd64(p + s - 8);
    h = rotl31(h * M1) * M2;
  } else if (s <= 32) {
    h ^= rd64(p);
    h = rotl31(h * M1) * M2;
    h ^= rd64(p + 8);
    h = rotl31(h * M1) * M2;
    h ^= rd64(p + s - 16);
    h = rotl31(h * M1) * M2;
    h ^= rd64(p + s - 8);
    h = rotl31(h * M1) * M2;
  } else {
    uint64_t v1 = h, v2 = seed, v3 = seed, v4 = seed; /* seed*hashKey[1..3] */
    while (s >= 32) {
      v1 ^= rd64(p);      v1 = rotl31(v1 * M1) * M2; p += 8;
      v2 ^= rd64(p);      v2 = rotl31(v2 * M2) * M3; p += 8;
      v3 ^= rd64(p);      v3 = rotl31(v3 * M3) * M4; p += 8;
      v4 ^= rd64(p);      v4 = rotl31(v4 * M4) * M1; p += 8;
      s -= 32;
    }
    h = v1 ^ v2 ^ v3 ^ v4;
    goto tail;
  }
  h ^= h >> 29;
  h *= M3;
  h ^= h >> 32;
  return h;
}

/* memhash64: hash.go:131-139 */
uint64_t ora_memhash64(uint64_t v, uint64_t seed) {
  uint64_t h = seed + 8;
  h ^= v;
  h = rotl31(h * M1) * M2;
  h ^= h >> 29;
  h *= M3;
  h ^= h >> 32;
  return h;
}

/* f64hash (hash.go:153-163) behind the float customizer, which first maps
 * NaN -> 0 (execgen/cmd/execgen/overloads_hash.go:94-107), so the random NaN
 * branch of f64hash is unreachable. */
static inline uint64_t float_hash(double f, uint64_t h) {
  if (f != f) f = 0;
  if (f == 0) return C1 * (C0 ^ h); /* +0, -0 */
  return ora_memhash(&f, h, 8);
}

/* test hook: injectable hash for the reference's worked examples */
static int g_test_hash_n = 0;
static int64_t g_test_hash_vals[64];
static uint32_t g_test_hash_hashes[64];
void ora_test_set_hash_table(const int64_t *vals, const uint32_t *hashes, int n) {
  if (n > 64) n = 64;
  g_test_hash_n = n;
  for (int i = 0; i < n; i++) { g_test_hash_vals[i] = vals[i]; g_test_hash_hashes[i] = hashes[i]; }
}

/* rehash: hash_utils.eg.go:39-1087 (template hash_utils_tmpl.go:61-152).
 * p = uintptr(buckets[i]); p = hash(value, p); buckets[i] = T(p).
 * NULL rows leave the bucket unchanged (hash_utils.eg.go:62-64). */
#define DEFINE_REHASH(T, NAME)                                                                  \
  static void NAME(T *buckets, const ora_col *col, int64_t n, const int64_t *sel) {             \
    switch (col->type) {                                                                        \
    case ORA_BOOL:                                                                              \
      for (int64_t i = 0; i < n; i++) {                                                         \
        int64_t s = sel ? sel[i] : i;                                                           \
        if (col_is_null(col, s)) continue;                                                      \
        uint64_t p = (uint64_t)buckets[i];                                                      \
        p = p * 31 + (uint64_t)col_get_bool(col, s); /* overloads_hash.go:55-65 */              \
        buckets[i] = (T)p;                                                                      \
      }                                                                                         \
      break;                                                                                    \
    case ORA_INT16:                                                                             \
    case ORA_INT32:                                                                             \
    case ORA_INT64:                                                                             \
      for (int64_t i = 0; i < n; i++) {                                                         \
        int64_t s = sel ? sel[i] : i;                                                           \
        if (col_is_null(col, s)) continue;                                                      \
        uint64_t p = (uint64_t)buckets[i];                                                      \
        p = ora_memhash64((uint64_t)col_get_int(col, s), p); /* overloads_hash.go:109-118 */    \
        buckets[i] = (T)p;                                                                      \
      }                                                                                         \
      break;                                                                                    \
    case ORA_FLOAT64:                                                                           \
      for (int64_t i = 0; i < n; i++) {                                                         \
        int64_t s = sel ? sel[i] : i;                                                           \
        if (col_is_null(col, s)) continue;                                                      \
        uint64_t p = (uint64_t)buckets[i];                                                      \
        p = float_hash(col_get_f64(col, s), p);                                                 \
        buckets[i] = (T)p;                                                                      \
      }                                                                                         \
      break;                                                                                    \
    case ORA_BYTES:                                                                             \
      for (int64_t i = 0; i < n; i++) {                                                         \
        int64_t s = sel ? sel[i] : i;                                                           \
        if (col_is_null(col, s)) continue;                                                      \
        uint64_t p = (uint64_t)buckets[i];                                                      \
        uint32_t len;                                                                           \
        const uint8_t *b = col_get_bytes(col, s, &len);                                         \
        p = ora_memhash(b, p, len); /* overloads_hash.go:71-80 */                               \
        buckets[i] = (T)p;                                                                      \
      }                                                                                         \
      break;                                                                                    \
    default:                                                                                    \
      break;                                                                                    \
    }                                                                                           \
  }
DEFINE_REHASH(uint32_t, rehash32)
DEFINE_REHASH(uint64_t, rehash64)

/* finalizeHash: hash_utils.go:104-123 */
static void finalize_hash32(uint32_t *b, int64_t n, uint32_t nb) {
  if ((nb & (nb - 1)) == 0) for (int64_t i = 0; i < n; i++) b[i] &= nb - 1;
  else for (int64_t i = 0; i < n; i++) b[i] %= nb;
}
static void finalize_hash64(uint64_t *b, int64_t n, uint64_t nb) {
  if ((nb & (nb - 1)) == 0) for (int64_t i = 0; i < n; i++) b[i] &= nb - 1;
  else for (int64_t i = 0; i < n; i++) b[i] %= nb;
}

void ora_hash_cols32(const ora_col *cols, int ncols, int64_t n, const int64_t *sel,
                     uint32_t init, uint32_t nbuckets, uint32_t *out) {
  if (n <= 0) return;
  for (int64_t i = 0; i < n; i++) out[i] = init; /* initHash hash_utils.go:65 */
  for (int c = 0; c < ncols; c++) rehash32(out, &cols[c], n, sel);
  if (nbuckets) finalize_hash32(out, n, nbuckets);
}
void ora_hash_cols64(const ora_col *cols, int ncols, int64_t n, const int64_t *sel,
                     uint64_t init, uint64_t nbuckets, uint64_t *out) {
  if (n <= 0) return;
  for (int64_t i = 0; i < n; i++) out[i] = init; /* initHash64 hash_utils.go:83 */
  for (int c = 0; c < ncols; c++) rehash64(out, &cols[c], n, sel);
  if (nbuckets) finalize_hash64(out, n, nbuckets);
}

/* TupleHashDistributor.Distribute: hash_utils.go:163-208, called once per
 * batch by HashRouter.processNextBatch (colflow/routers.go:683-715). */
void ora_distribute(const ora_col *cols, int ncols, const int32_t *hash_cols, int nhash,
                    int64_t n, uint64_t init, int32_t num_outputs, int32_t batch_size,
                    int32_t *dest, int64_t *counts) {
  (void)ncols;
  uint64_t *buckets = (uint64_t *)xmalloc((size_t)batch_size * 8);
  int64_t *sel = (int64_t *)xmalloc((size_t)batch_size * 8);
  for (int p = 0; p < num_outputs; p++) counts[p] = 0;
  for (int64_t start = 0; start < n; start += batch_size) {
    int64_t bn = n - start < batch_size ? n - start : batch_size;
    for (int64_t i = 0; i < bn; i++) { buckets[i] = init; sel[i] = start + i; }
    for (int k = 0; k < nhash; k++) rehash64(buckets, &cols[hash_cols[k]], bn, sel);
    finalize_hash64(buckets, bn, (uint64_t)num_outputs);
    for (int64_t i = 0; i < bn; i++) {
      dest[start + i] = (int32_t)buckets[i];
      counts[buckets[i]]++;
    }
  }
  free(buckets);
  free(sel);
}

/* ------------------------------------------------------------------------ */
/* HashTable: colexechash/hashtable.go, hashtable_tmpl.go                    */
/* ------------------------------------------------------------------------ */

#define MAX_KEYS 16
enum { BUILD_FULL = 0, BUILD_DISTINCT = 1 };
enum { PROBE_DEFAULT = 0, PROBE_DELETING = 1 };

typedef struct hashtable {
  /* BuildScratch (hashtable.go:79-94) */
  uint32_t *first; int64_t first_cap;
  uint32_t *next; int64_t next_len, next_cap;
  /* ProbeScratch (hashtable.go:105-196) */
  uint32_t *p_first; int64_t p_first_cap;
  uint32_t *p_next; int64_t p_next_cap;
  uint32_t *to_check, *to_check_id, *head_id, *hash_buffer;
  uint8_t *differs, *found_null;
  int64_t lim_cap;
  int64_t hash_buffer_cap;
  uint32_t *same;
  uint8_t *visited;
  int use_same, use_visited;
  /* Vals (hashtable.go:252-257) */
  vec *vals; int nvals;
  int key_cols[MAX_KEYS]; int nkeys;
  ora_col keys[MAX_KEYS];       /* ht.Keys: probe-side key vectors */
  ora_col build_keys[MAX_KEYS]; /* views of vals key columns */
  uint32_t num_buckets;
  double load_factor;
  int allow_null_eq, build_mode, probe_mode;
} hashtable;

static void ht_init(hashtable *ht, double load_factor, uint32_t initial_buckets, const int *types,
                    int ntypes, const int32_t *key_cols, int nkeys, int allow_null_eq,
                    int build_mode, int probe_mode) {
  /* NewHashTable: hashtable.go:308-378 */
  memset(ht, 0, sizeof(*ht));
  ht->num_buckets = initial_buckets;
  ht->load_factor = load_factor;
  ht->allow_null_eq = allow_null_eq;
  ht->build_mode = build_mode;
  ht->probe_mode = probe_mode;
  ht->nvals = ntypes;
  ht->vals = (vec *)xcalloc((size_t)ntypes, sizeof(vec));
  for (int i = 0; i < ntypes; i++) vec_init(&ht->vals[i], types[i]);
  ht->nkeys = nkeys;
  for (int i = 0; i < nkeys; i++) ht->key_cols[i] = key_cols[i];
  ht->first = (uint32_t *)xcalloc(initial_buckets, 4);
  ht->first_cap = initial_buckets;
  if (build_mode == BUILD_DISTINCT) {
    ht->p_first = (uint32_t *)xcalloc(initial_buckets, 4);
    ht->p_first_cap = initial_buckets;
    ht->next = (uint32_t *)xcalloc(1, 4); /* keyID 0 reserved */
    ht->next_len = 1;
    ht->next_cap = 1;
  }
}
static void ht_free(hashtable *ht) {
  free(ht->first); free(ht->next); free(ht->p_first); free(ht->p_next);
  free(ht->to_check); free(ht->to_check_id); free(ht->head_id); free(ht->hash_buffer);
  free(ht->differs); free(ht->found_null); free(ht->same); free(ht->visited);
  for (int i = 0; i < ht->nvals; i++) vec_free(&ht->vals[i]);
  free(ht->vals);
}
static inline int64_t ht_vals_len(const hashtable *ht) {
  /* In distinct build mode only key columns are stored (hashtable.go:331-351). */
  if (ht->build_mode == BUILD_DISTINCT) return ht->nkeys ? ht->vals[ht->key_cols[0]].n : ht->next_len - 1;
  return ht->nvals ? ht->vals[0].n : 0;
}

/* ComputeBuckets: hashtable.go:765-787 */
static void ht_compute_buckets(hashtable *ht, uint32_t *buckets, const ora_col *keys, int64_t n,
                               const int64_t *sel) {
  if (n == 0) return;
  if (g_test_hash_n > 0) {
    /* injected hash for the worked examples (single int64 key) */
    for (int64_t i = 0; i < n; i++) {
      int64_t v = col_get_int(&keys[0], sel ? sel[i] : i);
      uint32_t h = 0;
      for (int j = 0; j < g_test_hash_n; j++) if (g_test_hash_vals[j] == v) h = g_test_hash_hashes[j];
      buckets[i] = h;
    }
    finalize_hash32(buckets, n, ht->num_buckets);
    return;
  }
  for (int64_t i = 0; i < n; i++) buckets[i] = 1; /* initHash(DefaultInitHashValue) */
  for (int k = 0; k < ht->nkeys; k++) rehash32(buckets, &keys[k], n, sel);
  finalize_hash32(buckets, n, ht->num_buckets);
}

/* buildNextChains: hashtable.go:790-817 */
static void ht_build_next_chains(uint32_t *first, uint32_t *next, uint32_t offset, uint32_t batch_size) {
  for (uint32_t id = offset + batch_size - 1; id >= offset; id--) {
    uint32_t hash = next[id];
    uint32_t first_key_id = first[hash];
    if (first_key_id == 0 || id < first_key_id) {
      next[id] = first_key_id;
      first[hash] = id;
    } else {
      next[id] = next[first_key_id];
      next[first_key_id] = id;
    }
  }
}

static void ht_refresh_build_keys(hashtable *ht) {
  for (int k = 0; k < ht->nkeys; k++) vec_view(&ht->vals[ht->key_cols[k]], &ht->build_keys[k]);
}

/* buildFromBufferedTuples(NoAccounting): hashtable.go:424-469 */
static void ht_build_from_buffered(hashtable *ht) {
  int64_t n = ht_vals_len(ht);
  while ((double)n / (double)ht->num_buckets > ht->load_factor) ht->num_buckets *= 2; /* shouldResize :383 */
  if ((int64_t)ht->num_buckets > ht->first_cap) {
    free(ht->first);
    ht->first = (uint32_t *)xcalloc(ht->num_buckets, 4);
    ht->first_cap = ht->num_buckets;
  } else {
    memset(ht->first, 0, (size_t)ht->num_buckets * 4);
  }
  if (ht->p_first != NULL) {
    if ((int64_t)ht->num_buckets > ht->p_first_cap) {
      free(ht->p_first);
      ht->p_first = (uint32_t *)xcalloc(ht->num_buckets, 4);
      ht->p_first_cap = ht->num_buckets;
    } else {
      memset(ht->p_first, 0, (size_t)ht->num_buckets * 4);
    }
  }
  if (n + 1 > ht->next_cap) {
    ht->next = (uint32_t *)xrealloc(ht->next, (size_t)(n + 1) * 4);
    ht->next_cap = n + 1;
  }
  ht->next_len = n + 1;
  ht->next[0] = 0;
  ht_refresh_build_keys(ht);
  ht_compute_buckets(ht, ht->next + 1, ht->build_keys, n, NULL);
  ht_build_next_chains(ht->first, ht->next, 1, (uint32_t)n);
}

/* SetupLimitedSlices: hashtable.go:823-839 (HeadID/differs/foundNull zeroed;
 * ToCheck/ToCheckID reused as-is). */
static void ht_setup_limited(hashtable *ht, int64_t length) {
  if (length > ht->lim_cap) {
    ht->lim_cap = length;
    ht->to_check = (uint32_t *)xrealloc(ht->to_check, (size_t)length * 4);
    ht->to_check_id = (uint32_t *)xrealloc(ht->to_check_id, (size_t)length * 4);
    ht->head_id = (uint32_t *)xrealloc(ht->head_id, (size_t)length * 4);
    ht->differs = (uint8_t *)xrealloc(ht->differs, (size_t)length);
    ht->found_null = (uint8_t *)xrealloc(ht->found_null, (size_t)length);
  }
  memset(ht->head_id, 0, (size_t)length * 4);
  memset(ht->differs, 0, (size_t)length);
  memset(ht->found_null, 0, (size_t)length);
}

/* _ASSIGN_NE for one pair of non-NULL values.  Ints compare after widening
 * (mixed widths allowed, hashtable_tmpl.go:275-281); floats use the total
 * order in which NaN == NaN and +0 == -0 (hashtable_full_default.eg.go:4073-4091);
 * bytes compare byte-wise. */
static inline int values_differ(const ora_col *a, int64_t i, const ora_col *b, int64_t j) {
  switch (a->type) {
  case ORA_BOOL:
    return col_get_bool(a, i) != col_get_bool(b, j);
  case ORA_INT16:
  case ORA_INT32:
  case ORA_INT64:
    return col_get_int(a, i) != col_get_int(b, j);
  case ORA_FLOAT64: {
    double x = col_get_f64(a, i), y = col_get_f64(b, j);
    if (x < y || x > y) return 1;
    if (x == y) return 0;
    if (x != x) return !(y != y);
    return 1;
  }
  case ORA_BYTES: {
    uint32_t la, lb;
    const uint8_t *pa = col_get_bytes(a, i, &la), *pb = col_get_bytes(b, j, &lb);
    return la != lb || (la && memcmp(pa, pb, la) != 0);
  }
  default:
    return 1;
  }
}

/* _CHECK_COL_BODY: hashtable_tmpl.go:96-223.
 *  probe_sel: selection vector of the probing batch (or NULL)
 *  against_itself: buildIdx goes through probe_sel too (:150-158)
 *  deleting: HashTableDeletingProbeMode (:123-132) */
static void ht_check_col(hashtable *ht, const ora_col *probe, const ora_col *build, uint32_t n_to_check,
                         const int64_t *probe_sel, int against_itself, int deleting) {
  for (uint32_t pos = 0; pos < n_to_check; pos++) {
    uint32_t to_check = ht->to_check[pos];
    if (ht->differs[to_check]) continue;
    uint32_t key_id = ht->to_check_id[to_check];
    if (deleting && ht->visited[key_id]) {
      ht->differs[to_check] = 1;
      continue;
    }
    int64_t probe_idx = probe_sel ? probe_sel[to_check] : (int64_t)to_check;
    int64_t build_idx = (probe_sel && against_itself) ? probe_sel[key_id - 1] : (int64_t)key_id - 1;
    int probe_null = col_is_null(probe, probe_idx);
    int build_null = col_is_null(build, build_idx);
    if (probe_null) {
      if (ht->allow_null_eq) {
        if (!build_null) ht->differs[to_check] = 1;
      } else {
        ht->differs[to_check] = 1;
        ht->found_null[to_check] = 1;
      }
      continue;
    }
    if (build_null) {
      ht->differs[to_check] = 1;
      continue;
    }
    ht->differs[to_check] = (uint8_t)values_differ(probe, probe_idx, build, build_idx);
  }
}

/* checkCols: hashtable.go:733-746 */
static void ht_check_cols(hashtable *ht, uint32_t n_to_check, const int64_t *probe_sel) {
  for (int k = 0; k < ht->nkeys; k++)
    ht_check_col(ht, &ht->keys[k], &ht->build_keys[k], n_to_check, probe_sel, 0,
                 ht->probe_mode == PROBE_DELETING);
}

/* Check: hashtable_tmpl.go:732-756 with _CHECK_BODY :440-565 */
static uint32_t ht_check(hashtable *ht, uint32_t n_to_check, const int64_t *probe_sel) {
  ht_check_cols(ht, n_to_check, probe_sel);
  uint32_t n_differs = 0;
  uint32_t *tc = ht->to_check;
  if (ht->probe_mode == PROBE_DEFAULT) {
    int select_same = ht->use_same;
    for (uint32_t pos = 0; pos < n_to_check && n_differs < n_to_check; pos++) {
      uint32_t to_check = tc[pos];
      if (ht->found_null[to_check]) { /* !AllowNullEquality, !SelectDistinct :447-469 */
        ht->to_check_id[to_check] = 0;
        continue;
      }
      if (!ht->differs[to_check]) {
        uint32_t key_id = ht->to_check_id[to_check];
        if (ht->head_id[to_check] == 0) ht->head_id[to_check] = key_id;
        if (select_same && !ht->visited[key_id]) {
          uint32_t first_id = ht->head_id[to_check];
          if (first_id != key_id) {
            ht->same[key_id] = ht->same[first_id];
            ht->same[first_id] = key_id;
          }
          ht->visited[key_id] = 1;
          tc[n_differs++] = to_check;
        }
      } else {
        ht->differs[to_check] = 0;
        tc[n_differs++] = to_check;
      }
    }
  } else { /* HashTableDeletingProbeMode: _CHECK_BODY(false, true, false, true) */
    for (uint32_t pos = 0; pos < n_to_check && n_differs < n_to_check; pos++) {
      uint32_t to_check = tc[pos];
      if (!ht->differs[to_check]) {
        uint32_t key_id = ht->to_check_id[to_check];
        if (!ht->visited[key_id]) {
          ht->head_id[to_check] = key_id;
          ht->visited[key_id] = 1;
        } else if (key_id != 0) {
          tc[n_differs++] = to_check;
        }
      } else {
        ht->differs[to_check] = 0;
        tc[n_differs++] = to_check;
      }
    }
  }
  return n_differs;
}

/* DistinctCheck: hashtable.go:921-943 */
static uint32_t ht_distinct_check(hashtable *ht, uint32_t n_to_check, const int64_t *probe_sel) {
  ht_check_cols(ht, n_to_check, probe_sel);
  uint32_t n_differs = 0;
  uint32_t *tc = ht->to_check;
  for (uint32_t pos = 0; pos < n_to_check && n_differs < n_to_check; pos++) {
    uint32_t to_check = tc[pos];
    if (ht->found_null[to_check]) {
      ht->to_check_id[to_check] = 0;
    } else if (ht->differs[to_check]) {
      ht->differs[to_check] = 0;
      tc[n_differs++] = to_check;
    }
  }
  return n_differs;
}

/* CheckProbeForDistinct: hashtable_tmpl.go:764-775 (allowNullEquality=true
 * variant is the only one the hash aggregator uses). */
static uint32_t ht_check_probe_for_distinct(hashtable *ht, uint32_t n_to_check, const int64_t *sel) {
  for (int k = 0; k < ht->nkeys; k++)
    ht_check_col(ht, &ht->keys[k], &ht->keys[k], n_to_check, sel, 1, 0);
  uint32_t n_differs = 0;
  uint32_t *tc = ht->to_check;
  for (uint32_t pos = 0; pos < n_to_check && n_differs < n_to_check; pos++) {
    uint32_t to_check = tc[pos];
    if (!ht->allow_null_eq && ht->found_null[to_check]) {
      ht->head_id[to_check] = to_check + 1;
      continue;
    }
    if (!ht->differs[to_check]) {
      ht->head_id[to_check] = ht->to_check_id[to_check];
    } else {
      ht->differs[to_check] = 0;
      tc[n_differs++] = to_check;
    }
  }
  return n_differs;
}

/* CheckBuildForAggregation: hashtable.go:884-914 */
static uint32_t ht_check_build_for_aggregation(hashtable *ht, uint32_t n_to_check, const int64_t *sel) {
  for (int k = 0; k < ht->nkeys; k++) /* checkColsForDistinctTuples :751-761 */
    ht_check_col(ht, &ht->keys[k], &ht->build_keys[k], n_to_check, sel, 0, 0);
  uint32_t n_differs = 0;
  uint32_t *tc = ht->to_check;
  for (uint32_t pos = 0; pos < n_to_check && n_differs < n_to_check; pos++) {
    uint32_t to_check = tc[pos];
    if (ht->differs[to_check]) {
      ht->differs[to_check] = 0;
      tc[n_differs++] = to_check;
    } else {
      ht->head_id[to_check] = ht->to_check_id[to_check];
    }
  }
  return n_differs;
}

typedef uint32_t (*dup_checker)(hashtable *, uint32_t, const int64_t *);

/* FindBuckets / findBuckets / handleNextToCheckID: hashtable_tmpl.go:856-977 */
static void ht_find_buckets(hashtable *ht, int64_t batch_len, const int64_t *sel, const uint32_t *first,
                            const uint32_t *next, dup_checker checker, int zero_head_for_distinct,
                            int against_itself) {
  ht_setup_limited(ht, batch_len);
  uint32_t n_to_check = 0;
  for (int64_t i = 0; i < batch_len; i++) {
    uint32_t to_check = (uint32_t)i;
    uint32_t nid = first[ht->hash_buffer[i]];
    if (against_itself) {
      if (to_check + 1 == nid) ht->head_id[to_check] = nid;
      else { ht->to_check_id[to_check] = nid; ht->to_check[n_to_check++] = to_check; }
    } else if (nid != 0) {
      ht->to_check_id[to_check] = nid; ht->to_check[n_to_check++] = to_check;
    } else if (!zero_head_for_distinct) {
      ht->head_id[to_check] = to_check + 1;
    }
  }
  while (n_to_check > 0) {
    uint32_t n = checker(ht, n_to_check, sel);
    n_to_check = 0;
    for (uint32_t p = 0; p < n; p++) {
      uint32_t to_check = ht->to_check[p];
      uint32_t nid = next[ht->to_check_id[to_check]];
      if (against_itself) {
        if (to_check + 1 == nid) ht->head_id[to_check] = nid;
        else { ht->to_check_id[to_check] = nid; ht->to_check[n_to_check++] = to_check; }
      } else if (nid != 0) {
        ht->to_check_id[to_check] = nid; ht->to_check[n_to_check++] = to_check;
      } else if (!zero_head_for_distinct) {
        ht->head_id[to_check] = to_check + 1;
      }
    }
  }
}

/* ComputeHashAndBuildChains: hashtable.go:655-684.  `cols` are the batch's
 * column vectors; keys are picked by ht->key_cols. */
static void ht_compute_hash_and_build_chains(hashtable *ht, const ora_col *cols, int64_t batch_len,
                                             const int64_t *sel) {
  for (int k = 0; k < ht->nkeys; k++) ht->keys[k] = cols[ht->key_cols[k]];
  if (ht->p_next_cap < batch_len + 1) {
    free(ht->p_next);
    ht->p_next = (uint32_t *)xmalloc((size_t)(batch_len + 1) * 4);
    ht->p_next_cap = batch_len + 1;
  }
  ht_compute_buckets(ht, ht->p_next + 1, ht->keys, batch_len, sel);
  if (ht->hash_buffer_cap < batch_len) {
    ht->hash_buffer = (uint32_t *)xrealloc(ht->hash_buffer, (size_t)batch_len * 4);
    ht->hash_buffer_cap = batch_len;
  }
  memcpy(ht->hash_buffer, ht->p_next + 1, (size_t)batch_len * 4);
  if (batch_len < (int64_t)ht->num_buckets) {
    for (int64_t i = 0; i < batch_len; i++) ht->p_first[ht->hash_buffer[i]] = 0;
  } else {
    memset(ht->p_first, 0, (size_t)ht->num_buckets * 4);
  }
  ht_build_next_chains(ht->p_first, ht->p_next, 1, (uint32_t)batch_len);
}

/* AppendAllDistinct: hashtable.go:708-716 */
static void ht_append_all_distinct(hashtable *ht, const ora_col *cols, const int64_t *sel, int64_t len) {
  uint32_t num_buffered = (uint32_t)ht_vals_len(ht);
  for (int k = 0; k < ht->nkeys; k++) {
    int c = ht->key_cols[k];
    /* a column listed twice among the key columns is stored once */
    int dup = 0;
    for (int k2 = 0; k2 < k; k2++) if (ht->key_cols[k2] == c) dup = 1;
    if (!dup) vec_append_range(&ht->vals[c], &cols[c], sel, 0, len);
  }
  if (ht->next_len + len > ht->next_cap) {
    int64_t nc = ht->next_cap ? ht->next_cap : 1;
    while (nc < ht->next_len + len) nc *= 2;
    ht->next = (uint32_t *)xrealloc(ht->next, (size_t)nc * 4);
    ht->next_cap = nc;
  }
  memcpy(ht->next + ht->next_len, ht->hash_buffer, (size_t)len * 4);
  ht->next_len += len;
  ht_build_next_chains(ht->first, ht->next, num_buffered + 1, (uint32_t)len);
  ht_refresh_build_keys(ht);
  if ((double)ht_vals_len(ht) / (double)ht->num_buckets > ht->load_factor) ht_build_from_buffered(ht);
}

/* CheckBuildForDistinct: hashtable.go:839-870 */
static uint32_t ht_check_build_for_distinct(hashtable *ht, uint32_t n_to_check, const int64_t *sel) {
  for (int k = 0; k < ht->nkeys; k++) /* checkColsForDistinctTuples :751-761 */
    ht_check_col(ht, &ht->keys[k], &ht->build_keys[k], n_to_check, sel, 0, 0);
  uint32_t n_differs = 0;
  uint32_t *tc = ht->to_check;
  for (uint32_t pos = 0; pos < n_to_check && n_differs < n_to_check; pos++) {
    uint32_t to_check = tc[pos];
    if (ht->found_null[to_check]) {
      /* only set when allowNullEquality is false: a NULL makes the tuple distinct from all others */
      ht->found_null[to_check] = 0;
      ht->head_id[to_check] = to_check + 1;
    } else if (ht->differs[to_check]) {
      ht->differs[to_check] = 0;
      tc[n_differs++] = to_check;
    }
  }
  return n_differs;
}

/* updateSel: hashtable_tmpl.go:779-838 (the batch always carries a selection vector here).
 * Returns the new batch length; compacts sel and the hash buffer. */
static int64_t ht_update_sel(hashtable *ht, int64_t *sel, int64_t batch_len) {
  if (batch_len == 0) return 0;
  uint8_t *visited = ht->found_null; /* "Reuse the buffer allocated for foundNull" */
  memset(visited, 0, (size_t)batch_len);
  int64_t distinct = 0;
  for (int64_t i = 0; i < batch_len && distinct < batch_len; i++) {
    uint32_t head = ht->head_id[i];
    if (head != 0 && !visited[head - 1]) {
      sel[distinct] = sel[head - 1];
      visited[head - 1] = 1;
      ht->hash_buffer[distinct] = ht->hash_buffer[i];
      distinct++;
    }
  }
  return distinct;
}

/* UnorderedDistinct: colexec/unordered_distinct.go:72-121 (Init: load factor 2.0, distinct build mode,
 * allowNullEquality = !nullsAreDistinct) driving HashTable.DistinctBuild (hashtable.go:628-649) over
 * batches of `batch_size` rows.  Emits every column of the rows that survive, in input order.
 * *dropped = 1 when some batch lost a tuple (MaybeEmitErrorOnDup's condition, origLen > updatedLen). */
ora_result *ora_unordered_distinct(const ora_col *cols, int ncols, int64_t n, const int32_t *distinct_cols, int ndistinct,
                                   int32_t nulls_are_distinct, int32_t batch_size, uint32_t initial_buckets,
                                   int32_t *dropped) {
  int types[64];
  for (int c = 0; c < ncols; c++) types[c] = cols[c].type;
  hashtable ht;
  ht_init(&ht, 2.0, initial_buckets ? initial_buckets : 128, types, ncols, distinct_cols, ndistinct,
          !nulls_are_distinct, BUILD_DISTINCT, PROBE_DEFAULT);
  ora_result *out = result_new(ncols, types);
  int64_t *sel = (int64_t *)xmalloc((size_t)(batch_size > 0 ? batch_size : 1) * 8);
  if (dropped) *dropped = 0;
  for (int64_t start = 0; start < n; start += batch_size) {
    int64_t len = n - start < batch_size ? n - start : batch_size;
    const int64_t orig_len = len;
    for (int64_t i = 0; i < len; i++) sel[i] = start + i;
    /* DistinctBuild */
    ht_compute_hash_and_build_chains(&ht, cols, len, sel);
    ht_find_buckets(&ht, len, sel, ht.p_first, ht.p_next, ht_check_probe_for_distinct, 0, 1); /* RemoveDuplicates :686-702 */
    len = ht_update_sel(&ht, sel, len);
    if (ht_vals_len(&ht) > 0 && len > 0) {
      ht_find_buckets(&ht, len, sel, ht.first, ht.next, ht_check_build_for_distinct, 0, 0);
      len = ht_update_sel(&ht, sel, len);
    }
    if (len > 0) ht_append_all_distinct(&ht, cols, sel, len);
    if (len < orig_len && dropped) *dropped = 1;
    for (int c = 0; c < ncols && len > 0; c++) vec_append_range(&out->cols[c], &cols[c], sel, 0, len);
  }
  free(sel);
  ht_free(&ht);
  return out;
}

/* ------------------------------------------------------------------------ */
/* hashJoiner: colexecjoin/hashjoiner.go + hashjoiner_tmpl.go                */
/* ------------------------------------------------------------------------ */

typedef struct hash_joiner {
  int join_type;
  int right_distinct, track_build_matches;
  const ora_col *left, *right;
  int nleft_cols, nright_cols;
  int64_t nleft, nright;
  int left_eq[MAX_KEYS];
  int neq;
  int32_t coldata_batch_size;
  hashtable ht;
  /* probeState (hashjoiner.go:193-225) */
  int64_t *build_idx, *probe_idx;
  int64_t idx_len, idx_cap;
  uint8_t *probe_row_unmatched;
  uint8_t *build_row_matched;
  uint32_t *buckets;
  int have_prev_batch;
  int64_t prev_batch_resume_idx;
  int64_t emit_row_idx;
  ora_result *out;
} hash_joiner;

static inline int jt_is_set_op(int j) { return j == ORA_INTERSECT_ALL || j == ORA_EXCEPT_ALL; }
static inline int jt_left_cols_in_output(int j) { return !(j == ORA_RIGHT_SEMI || j == ORA_RIGHT_ANTI); }
static inline int jt_right_cols_in_output(int j) {
  return !(j == ORA_LEFT_SEMI || j == ORA_LEFT_ANTI || j == ORA_INTERSECT_ALL || j == ORA_EXCEPT_ALL);
}
static inline int jt_empty_output_when_right_empty(int j) { /* join_type.go:79-88 */
  return j == ORA_INNER || j == ORA_RIGHT_OUTER || j == ORA_LEFT_SEMI || j == ORA_RIGHT_SEMI ||
         j == ORA_INTERSECT_ALL || j == ORA_RIGHT_ANTI;
}
static inline int jt_left_or_full_outer(int j) { return j == ORA_LEFT_OUTER || j == ORA_FULL_OUTER; }
static inline int jt_left_anti_or_except_all(int j) { return j == ORA_LEFT_ANTI || j == ORA_EXCEPT_ALL; }

/* prepareForCollecting: hashjoiner.go:467-497 */
static void hj_prepare_for_collecting(hash_joiner *hj, int64_t batch_size) {
  if (batch_size > hj->idx_cap) {
    hj->idx_cap = batch_size;
    hj->build_idx = (int64_t *)xrealloc(hj->build_idx, (size_t)batch_size * 8);
    hj->probe_idx = (int64_t *)xrealloc(hj->probe_idx, (size_t)batch_size * 8);
    hj->probe_row_unmatched = (uint8_t *)xrealloc(hj->probe_row_unmatched, (size_t)batch_size);
  }
  hj->idx_len = batch_size;
  if (jt_left_or_full_outer(hj->join_type)) memset(hj->probe_row_unmatched, 0, (size_t)batch_size);
}

/* collectProbeOuter: hashjoiner_tmpl.go:16-88 */
static int64_t hj_collect_probe_outer(hash_joiner *hj, int64_t batch_size, const int64_t *sel) {
  hashtable *ht = &hj->ht;
  int64_t n_results = 0, max_results = hj->idx_len;
  for (int64_t i = hj->prev_batch_resume_idx; i < batch_size; i++) {
    uint32_t current_id = ht->head_id[i];
    for (; n_results < max_results; n_results++) {
      int unmatched = current_id == 0;
      hj->probe_row_unmatched[n_results] = (uint8_t)unmatched;
      hj->build_idx[n_results] = unmatched ? 0 : (int64_t)current_id - 1;
      hj->probe_idx[n_results] = sel ? sel[i] : i;
      current_id = ht->same ? ht->same[current_id] : 0;
      ht->head_id[i] = current_id;
      if (current_id == 0) { n_results++; break; }
    }
    if (n_results == max_results) {
      if (current_id != 0) { hj->have_prev_batch = 1; hj->prev_batch_resume_idx = i; }
      else if (i + 1 < batch_size) { hj->have_prev_batch = 1; hj->prev_batch_resume_idx = i + 1; }
      return n_results;
    }
  }
  return n_results;
}

/* collectProbeNoOuter: hashjoiner_tmpl.go:90-142 */
static int64_t hj_collect_probe_no_outer(hash_joiner *hj, int64_t batch_size, const int64_t *sel) {
  hashtable *ht = &hj->ht;
  int64_t n_results = 0, max_results = hj->idx_len;
  for (int64_t i = hj->prev_batch_resume_idx; i < batch_size; i++) {
    uint32_t current_id = ht->head_id[i];
    for (; current_id != 0 && n_results < max_results; n_results++) {
      hj->build_idx[n_results] = (int64_t)current_id - 1;
      hj->probe_idx[n_results] = sel ? sel[i] : i;
      current_id = ht->same ? ht->same[current_id] : 0;
      ht->head_id[i] = current_id;
    }
    if (n_results == max_results) {
      if (current_id != 0) { hj->have_prev_batch = 1; hj->prev_batch_resume_idx = i; }
      else if (i + 1 < batch_size) { hj->have_prev_batch = 1; hj->prev_batch_resume_idx = i + 1; }
      return n_results;
    }
  }
  return n_results;
}

/* collectSingleMatch: hashjoiner_tmpl.go:144-189 */
static int64_t hj_collect_single_match(hash_joiner *hj, int64_t batch_size, const int64_t *sel, int match_zero) {
  hashtable *ht = &hj->ht;
  int64_t n_results = 0;
  for (int64_t i = 0; i < batch_size; i++) {
    uint32_t current_id = ht->head_id[i];
    if ((match_zero && current_id == 0) || (!match_zero && current_id != 0))
      hj->probe_idx[n_results++] = sel ? sel[i] : i;
  }
  return n_results;
}

/* collectRightSemiAnti: hashjoiner_tmpl.go:191-208 */
static void hj_collect_right_semi_anti(hash_joiner *hj, int64_t batch_size) {
  hashtable *ht = &hj->ht;
  for (int64_t i = 0; i < batch_size; i++) {
    uint32_t current_id = ht->head_id[i];
    while (current_id != 0) {
      hj->build_row_matched[current_id - 1] = 1;
      current_id = ht->same[current_id];
    }
  }
}

/* collect: hashjoiner_tmpl.go:276-311 */
static int64_t hj_collect(hash_joiner *hj, int64_t batch_size, const int64_t *sel) {
  if (!jt_left_cols_in_output(hj->join_type)) { hj_collect_right_semi_anti(hj, batch_size); return 0; }
  if (jt_left_or_full_outer(hj->join_type)) return hj_collect_probe_outer(hj, batch_size, sel);
  if (jt_left_anti_or_except_all(hj->join_type)) return hj_collect_single_match(hj, batch_size, sel, 1);
  if (hj->join_type == ORA_INTERSECT_ALL) return hj_collect_single_match(hj, batch_size, sel, 0);
  return hj_collect_probe_no_outer(hj, batch_size, sel);
}

/* distinctCollect: hashjoiner_tmpl.go:211-271,318-337 */
static int64_t hj_distinct_collect(hash_joiner *hj, int64_t batch_size, const int64_t *sel) {
  hashtable *ht = &hj->ht;
  if (jt_left_or_full_outer(hj->join_type)) {
    for (int64_t i = 0; i < batch_size; i++) {
      uint32_t id = ht->to_check_id[i];
      hj->probe_row_unmatched[i] = id == 0;
      hj->build_idx[i] = id == 0 ? 0 : (int64_t)id - 1;
      hj->probe_idx[i] = sel ? sel[i] : i;
    }
    return batch_size;
  }
  int64_t n_results = 0;
  for (int64_t i = 0; i < batch_size; i++) {
    uint32_t id = ht->to_check_id[i];
    if (id != 0) {
      hj->build_idx[n_results] = (int64_t)id - 1;
      hj->probe_idx[n_results] = sel ? sel[i] : i;
      n_results++;
    }
  }
  return n_results;
}

/* congregate: hashjoiner.go:619-712.  The output batch is appended to the
 * result (the caller of Next() would consume it). */
static void hj_congregate(hash_joiner *hj, int64_t n_results) {
  ora_result *out = hj->out;
  int oc = 0;
  for (int c = 0; c < hj->nleft_cols; c++, oc++)
    for (int64_t r = 0; r < n_results; r++) vec_push_from(&out->cols[oc], &hj->left[c], hj->probe_idx[r]);
  if (jt_right_cols_in_output(hj->join_type)) {
    int outer = jt_left_or_full_outer(hj->join_type);
    for (int c = 0; c < hj->nright_cols; c++, oc++) {
      ora_col v;
      vec_view(&hj->ht.vals[c], &v);
      int have_vals = hj->ht.vals[c].n > 0;
      for (int64_t r = 0; r < n_results; r++) {
        if ((outer && hj->probe_row_unmatched[r]) || !have_vals) vec_push_null(&out->cols[oc]);
        else vec_push_from(&out->cols[oc], &v, hj->build_idx[r]);
      }
    }
  }
  if (hj->track_build_matches) {
    if (jt_left_or_full_outer(hj->join_type)) {
      for (int64_t r = 0; r < n_results; r++)
        if (!hj->probe_row_unmatched[r]) hj->build_row_matched[hj->build_idx[r]] = 1;
    } else {
      for (int64_t r = 0; r < n_results; r++) hj->build_row_matched[hj->build_idx[r]] = 1;
    }
  }
}

/* emitRight: hashjoiner.go:405-460 */
static int64_t hj_emit_right(hash_joiner *hj, int matched) {
  ora_result *out = hj->out;
  int64_t nvals = ht_vals_len(&hj->ht);
  int64_t n_results = 0;
  if (hj->coldata_batch_size > hj->idx_cap) hj_prepare_for_collecting(hj, hj->coldata_batch_size);
  while (n_results < hj->coldata_batch_size && hj->emit_row_idx < nvals) {
    if (hj->build_row_matched[hj->emit_row_idx] == matched) hj->build_idx[n_results++] = hj->emit_row_idx;
    hj->emit_row_idx++;
  }
  int oc = 0;
  if (jt_left_cols_in_output(hj->join_type))
    for (int c = 0; c < hj->nleft_cols; c++, oc++)
      for (int64_t r = 0; r < n_results; r++) vec_push_null(&out->cols[oc]);
  for (int c = 0; c < hj->nright_cols; c++, oc++) {
    ora_col v;
    vec_view(&hj->ht.vals[c], &v);
    for (int64_t r = 0; r < n_results; r++) vec_push_from(&out->cols[oc], &v, hj->build_idx[r]);
  }
  return n_results;
}

/* (thread-local: bench.py's P-way CPU baseline runs one joiner per host thread) */
static _Thread_local hash_joiner *g_last_join = NULL;
static _Thread_local uint32_t g_dbg_head[64];
static _Thread_local int g_dbg_nhead = 0;
static void hj_destroy(hash_joiner *hj) {
  if (!hj) return;
  ht_free(&hj->ht);
  free(hj->build_idx); free(hj->probe_idx); free(hj->probe_row_unmatched);
  free(hj->build_row_matched); free(hj->buckets);
  free(hj);
}

ora_result *ora_hash_join(int32_t join_type,
                          const ora_col *left, int nleft_cols, int64_t nleft,
                          const ora_col *right, int nright_cols, int64_t nright,
                          const int32_t *left_eq, const int32_t *right_eq, int neq,
                          int32_t right_eq_cols_are_key,
                          int32_t batch_size, uint32_t initial_buckets) {
  if (g_last_join) { hj_destroy(g_last_join); g_last_join = NULL; }
  hash_joiner *hj = (hash_joiner *)xcalloc(1, sizeof(*hj));
  g_last_join = hj;
  hj->join_type = join_type;
  hj->left = left; hj->right = right;
  hj->nleft_cols = nleft_cols; hj->nright_cols = nright_cols;
  hj->nleft = nleft; hj->nright = nright;
  hj->neq = neq;
  hj->coldata_batch_size = batch_size;
  for (int i = 0; i < neq; i++) hj->left_eq[i] = left_eq[i];

  /* MakeHashJoinerSpec: hashjoiner.go:827-883 */
  int right_distinct = right_eq_cols_are_key;
  if (join_type == ORA_LEFT_SEMI) right_distinct = 1;
  else if (join_type == ORA_LEFT_ANTI || join_type == ORA_RIGHT_ANTI || join_type == ORA_RIGHT_SEMI ||
           join_type == ORA_INTERSECT_ALL || join_type == ORA_EXCEPT_ALL)
    right_distinct = 0;
  hj->right_distinct = right_distinct;
  hj->track_build_matches = join_type == ORA_RIGHT_OUTER || join_type == ORA_FULL_OUTER ||
                            join_type == ORA_RIGHT_SEMI || join_type == ORA_RIGHT_ANTI;

  /* output types: JoinType.MakeOutputTypes (join_type.go:101-140) */
  int out_types[2 * 64];
  int nout = 0;
  if (jt_left_cols_in_output(join_type)) for (int c = 0; c < nleft_cols; c++) out_types[nout++] = left[c].type;
  if (jt_right_cols_in_output(join_type)) for (int c = 0; c < nright_cols; c++) out_types[nout++] = right[c].type;
  hj->out = result_new(nout, out_types);

  /* Init: hashjoiner.go:259-289 */
  int allow_null_eq = 0, probe_mode = PROBE_DEFAULT;
  if (jt_is_set_op(join_type)) { allow_null_eq = 1; probe_mode = PROBE_DELETING; }
  int rtypes[64];
  for (int c = 0; c < nright_cols; c++) rtypes[c] = right[c].type;
  hashtable *ht = &hj->ht;
  ht_init(ht, 1.0, initial_buckets ? initial_buckets : 256, rtypes, nright_cols, right_eq, neq,
          allow_null_eq, BUILD_FULL, probe_mode);

  /* build(): hashjoiner.go:339-400; FullBuild hashtable.go:477-500 */
  for (int64_t start = 0; start < nright; start += batch_size) {
    int64_t end = start + batch_size < nright ? start + batch_size : nright;
    for (int c = 0; c < nright_cols; c++) vec_append_range(&ht->vals[c], &right[c], NULL, start, end);
  }
  ht_build_from_buffered(ht);
  int64_t nvals = ht_vals_len(ht);
  if (nvals == 0 && jt_empty_output_when_right_empty(join_type)) return hj->out; /* hashjoiner.go:300-306 */

  int need_same = 0, need_visited = jt_is_set_op(join_type);
  if (!right_distinct) {
    switch (join_type) {
    case ORA_LEFT_ANTI: case ORA_EXCEPT_ALL: case ORA_INTERSECT_ALL: break;
    default: need_same = 1; need_visited = 1;
    }
  }
  if (need_same) { ht->same = (uint32_t *)xcalloc((size_t)nvals + 1, 4); ht->use_same = 1; }
  if (need_visited) { ht->visited = (uint8_t *)xcalloc((size_t)nvals + 1, 1); ht->visited[0] = 1; ht->use_visited = 1; }
  if (hj->track_build_matches) hj->build_row_matched = (uint8_t *)xcalloc((size_t)nvals + 1, 1);

  /* probing: exec() hashjoiner.go:504-610, one call per output batch */
  int64_t *sel_scratch = NULL; /* inputs here are dense; probe rows addressed absolutely */
  hj->buckets = (uint32_t *)xmalloc((size_t)batch_size * 4);
  sel_scratch = (int64_t *)xmalloc((size_t)batch_size * 8);
  for (int64_t start = 0; start < nleft; start += batch_size) {
    int64_t bsz = start + batch_size < nleft ? batch_size : nleft - start;
    /* the batch is a window [start, start+bsz) of the left input; expressed as a
     * selection vector over the full columns so that probeIdx indexes `left`. */
    for (int64_t i = 0; i < bsz; i++) sel_scratch[i] = start + i;
    const int64_t *sel = sel_scratch;
    for (int k = 0; k < neq; k++) ht->keys[k] = left[hj->left_eq[k]];
    ht_compute_buckets(ht, hj->buckets, ht->keys, bsz, sel);
    ht_setup_limited(ht, bsz);
    uint32_t n_to_check = 0;
    for (int64_t i = 0; i < bsz; i++) {
      uint32_t f = ht->first[hj->buckets[i]];
      ht->to_check_id[i] = f;
      if (f != 0) ht->to_check[n_to_check++] = (uint32_t)i;
    }
    hj_prepare_for_collecting(hj, bsz);
    while (n_to_check > 0) {
      n_to_check = right_distinct ? ht_distinct_check(ht, n_to_check, sel) : ht_check(ht, n_to_check, sel);
      uint32_t n = n_to_check;
      n_to_check = 0;
      for (uint32_t p = 0; p < n; p++) {
        uint32_t to_check = ht->to_check[p];
        uint32_t next_id = ht->next[ht->to_check_id[to_check]];
        ht->to_check_id[to_check] = next_id;
        if (next_id != 0) ht->to_check[n_to_check++] = to_check;
      }
    }
    hj->prev_batch_resume_idx = 0;
    hj->have_prev_batch = 0;
    if (bsz <= 64) { /* test hook: HeadID of the last probed batch, before collect() consumes it */
      g_dbg_nhead = (int)bsz;
      memcpy(g_dbg_head, ht->head_id, (size_t)bsz * 4);
    }
    int64_t n_results = right_distinct ? hj_distinct_collect(hj, bsz, sel) : hj_collect(hj, bsz, sel);
    if (n_results > 0) hj_congregate(hj, n_results);
    while (hj->have_prev_batch) { /* exec() resumption :505-523 */
      hj->have_prev_batch = 0;
      hj_prepare_for_collecting(hj, hj->coldata_batch_size);
      n_results = hj_collect(hj, bsz, sel);
      if (n_results > 0) hj_congregate(hj, n_results);
    }
  }
  free(sel_scratch);

  /* hjEmittingRight: hashjoiner.go:322-328 */
  if (hj->track_build_matches) {
    while (hj->emit_row_idx < nvals) hj_emit_right(hj, join_type == ORA_RIGHT_SEMI);
  }
  return hj->out;
}

void ora_test_last_join_state(uint32_t *first, int32_t *nfirst, uint32_t *next, uint32_t *same,
                              uint8_t *visited, int64_t *nkeys, uint32_t *head_id, int32_t *nhead) {
  hash_joiner *hj = g_last_join;
  if (!hj) return;
  hashtable *ht = &hj->ht;
  int64_t nvals = ht_vals_len(ht);
  if (nfirst) *nfirst = (int32_t)ht->num_buckets;
  if (first) memcpy(first, ht->first, (size_t)ht->num_buckets * 4);
  if (nkeys) *nkeys = nvals;
  if (next && ht->next) memcpy(next, ht->next, (size_t)(nvals + 1) * 4);
  if (same && ht->same) memcpy(same, ht->same, (size_t)(nvals + 1) * 4);
  if (visited && ht->visited) memcpy(visited, ht->visited, (size_t)(nvals + 1));
  if (nhead) *nhead = g_dbg_nhead;
  if (head_id) memcpy(head_id, g_dbg_head, (size_t)g_dbg_nhead * 4);
}

/* ------------------------------------------------------------------------ */
/* aggregate functions: colexecagg/hash_*_agg.eg.go                          */
/* ------------------------------------------------------------------------ */

typedef struct agg_state {
  /* one per (group, aggregate function), like the generated *HashAgg structs */
  double fsum;          /* sumFloat64HashAgg.curAgg / avg curSum */
  int64_t isum;         /* sumIntInt64HashAgg.curAgg, count */
  __int128 dsum;        /* SUM over INT: the exact total (the reference adds apd decimals: hash_sum_agg.eg.go:287-373) */
  int64_t count;        /* numNonNull / curCount */
  uint8_t found;        /* foundNonNullForCurrentGroup */
  uint8_t bval;
  /* current min/max/any value */
  int64_t ival;
  double fval;
  uint8_t *bytes; uint32_t blen, bcap;
  /* DISTINCT: the input values this (group, aggregation) has seen, as [null u8][len u32][bytes] records
   * (aggregators_util.go:298-373 keeps a map of encoded tuples per bucket) */
  uint8_t *seen; int64_t seen_len, seen_cap;
} agg_state;

typedef struct agg_bucket { agg_state *fns; } agg_bucket;

/* total-order float compare used by MIN/MAX (hash_min_max_agg.eg.go:966-981) */
static inline int float_cmp(double a, double b) {
  if (a < b) return -1;
  if (a > b) return 1;
  if (a == b) return 0;
  if (a != a) return (b != b) ? 0 : -1;
  return 1;
}
static void state_set_bytes(agg_state *s, const uint8_t *p, uint32_t len) {
  if (len > s->bcap) { s->bytes = (uint8_t *)xrealloc(s->bytes, len); s->bcap = len; }
  if (len) memcpy(s->bytes, p, len);
  s->blen = len;
}
static inline int bytes_cmp(const uint8_t *a, uint32_t la, const uint8_t *b, uint32_t lb) {
  uint32_t m = la < lb ? la : lb;
  int c = m ? memcmp(a, b, m) : 0;
  if (c) return c;
  return la < lb ? -1 : (la > lb ? 1 : 0);
}

/* fn.Compute(vecs, inputIdxs, 0, n, sel): folds rows sel[0..n) sequentially.
 * Returns 1 on ErrIntOutOfRange (hash_sum_int_agg.eg.go:295-298). */
static int agg_compute(agg_state *s, const ora_agg_spec *spec, const ora_col *cols, const int64_t *sel, int64_t n) {
  const ora_col *col = spec->col >= 0 ? &cols[spec->col] : NULL;
  switch (spec->func) {
  case ORA_COUNT_ROWS: /* hash_count_agg.eg.go:26-57 */
    s->count += n;
    return 0;
  case ORA_COUNT: /* hash_count_agg.eg.go:90-135 */
    for (int64_t k = 0; k < n; k++) if (!col_is_null(col, sel[k])) s->count++;
    return 0;
  case ORA_SUM_INT: /* hash_sum_int_agg.eg.go:265-349 (int16/32/64 inputs) */
    for (int64_t k = 0; k < n; k++) {
      int64_t i = sel[k];
      if (col_is_null(col, i)) continue;
      int64_t v = col_get_int(col, i);
      int64_t result = (int64_t)((uint64_t)s->isum + (uint64_t)v);
      if ((result < s->isum) != (v < 0)) return 1;
      s->isum = result;
      s->count++;
    }
    return 0;
  case ORA_SUM: /* float: hash_sum_agg.eg.go:505-583 */
    if (is_int_type(col->type)) { /* int16/32/64 -> DECIMAL: hash_sum_agg.eg.go:30-373; exact, so a 128-bit integer here */
      for (int64_t k = 0; k < n; k++) {
        int64_t i = sel[k];
        if (col_is_null(col, i)) continue;
        s->dsum += (__int128)col_get_int(col, i);
        s->count++;
      }
      return 0;
    }
    /* fallthrough */
  case ORA_AVG: /* float: hash_avg_agg.eg.go:525-604 */
    for (int64_t k = 0; k < n; k++) {
      int64_t i = sel[k];
      if (col_is_null(col, i)) continue;
      s->fsum += col_get_f64(col, i);
      s->count++;
    }
    return 0;
  case ORA_BOOL_AND: /* hash_bool_and_or_agg.eg.go */
  case ORA_BOOL_OR:
    for (int64_t k = 0; k < n; k++) {
      int64_t i = sel[k];
      if (col_is_null(col, i)) continue;
      uint8_t v = col_get_bool(col, i);
      if (!s->found) { s->bval = v; s->found = 1; }
      else if (spec->func == ORA_BOOL_AND) s->bval = s->bval && v;
      else s->bval = s->bval || v;
    }
    return 0;
  case ORA_ANY_NOT_NULL: /* hash_any_not_null_agg.eg.go: first non-NULL wins */
    for (int64_t k = 0; k < n && !s->found; k++) {
      int64_t i = sel[k];
      if (col_is_null(col, i)) continue;
      s->found = 1;
      switch (col->type) {
      case ORA_BOOL: s->bval = col_get_bool(col, i); break;
      case ORA_FLOAT64: s->fval = col_get_f64(col, i); break;
      case ORA_BYTES: { uint32_t l; const uint8_t *p = col_get_bytes(col, i, &l); state_set_bytes(s, p, l); break; }
      default: s->ival = col_get_int(col, i);
      }
    }
    return 0;
  case ORA_MIN: /* hash_min_max_agg.eg.go */
  case ORA_MAX: {
    int want = spec->func == ORA_MIN ? -1 : 1;
    for (int64_t k = 0; k < n; k++) {
      int64_t i = sel[k];
      if (col_is_null(col, i)) continue;
      switch (col->type) {
      case ORA_BOOL: {
        uint8_t v = col_get_bool(col, i);
        if (s->count == 0) s->bval = v;
        else { int c = (int)v - (int)s->bval; if ((c < 0 ? -1 : (c > 0)) == want) s->bval = v; }
        break;
      }
      case ORA_FLOAT64: {
        double v = col_get_f64(col, i);
        if (s->count == 0) s->fval = v;
        else if (float_cmp(v, s->fval) == want) s->fval = v;
        break;
      }
      case ORA_BYTES: {
        uint32_t l;
        const uint8_t *p = col_get_bytes(col, i, &l);
        if (s->count == 0) state_set_bytes(s, p, l);
        else { int c = bytes_cmp(p, l, s->bytes, s->blen); if ((c < 0 ? -1 : (c > 0)) == want) state_set_bytes(s, p, l); }
        break;
      }
      default: {
        int64_t v = col_get_int(col, i);
        if (s->count == 0) s->ival = v;
        else if ((v < s->ival ? -1 : (v > s->ival)) == want) s->ival = v;
      }
      }
      s->count++;
    }
    return 0;
  }
  default:
    return 2;
  }
}

static int agg_output_type(const ora_agg_spec *spec, const ora_col *cols) {
  /* colexecagg.ProcessAggregations output types (aggregate_funcs.go:587-612) */
  switch (spec->func) {
  case ORA_COUNT: case ORA_COUNT_ROWS: case ORA_SUM_INT: return ORA_INT64;
  case ORA_SUM: if (is_int_type(cols[spec->col].type)) return ORA_BYTES; /* 16 bytes, little endian two's complement */ return ORA_FLOAT64;
  case ORA_AVG: return ORA_FLOAT64;
  case ORA_BOOL_AND: case ORA_BOOL_OR: return ORA_BOOL;
  default: return cols[spec->col].type;
  }
}

/* fn.Flush(outputIdx) */
static void agg_flush(const agg_state *s, const ora_agg_spec *spec, const ora_col *cols, vec *out) {
  vec_reserve(out, out->n + 1);
  int64_t d = out->n;
  int null = 0;
  switch (spec->func) {
  case ORA_COUNT: case ORA_COUNT_ROWS:
    ((int64_t *)out->data)[d] = s->count; break;
  case ORA_SUM_INT:
    if (s->count == 0) null = 1; else ((int64_t *)out->data)[d] = s->isum; break;
  case ORA_SUM:
    if (out->type == ORA_BYTES) { /* SUM over INT */
      if (s->count == 0) { null = 1; break; }
      if (out->buflen + 16 > out->bufcap) {
        int64_t nc = out->bufcap ? out->bufcap : 256;
        while (nc < out->buflen + 16) nc *= 2;
        out->buf = (uint8_t *)xrealloc(out->buf, (size_t)nc);
        out->bufcap = nc;
      }
      memcpy(out->buf + out->buflen, &s->dsum, 16);
      out->buflen += 16;
      break;
    }
    if (s->count == 0) null = 1; else ((double *)out->data)[d] = s->fsum; break;
  case ORA_AVG:
    if (s->count == 0) null = 1; else ((double *)out->data)[d] = s->fsum / (double)s->count; break;
  case ORA_BOOL_AND: case ORA_BOOL_OR:
    if (!s->found) null = 1; else out->data[d] = s->bval; break;
  default: { /* ANY_NOT_NULL, MIN, MAX */
    int have = spec->func == ORA_ANY_NOT_NULL ? s->found : s->count > 0;
    if (!have) { null = 1; break; }
    switch (cols[spec->col].type) {
    case ORA_BOOL: out->data[d] = s->bval; break;
    case ORA_FLOAT64: ((double *)out->data)[d] = s->fval; break;
    case ORA_INT16: ((int16_t *)out->data)[d] = (int16_t)s->ival; break;
    case ORA_INT32: ((int32_t *)out->data)[d] = (int32_t)s->ival; break;
    case ORA_INT64: ((int64_t *)out->data)[d] = s->ival; break;
    case ORA_BYTES:
      if (out->buflen + s->blen > out->bufcap) {
        int64_t nc = out->bufcap ? out->bufcap : 256;
        while (nc < out->buflen + s->blen) nc *= 2;
        out->buf = (uint8_t *)xrealloc(out->buf, (size_t)nc);
        out->bufcap = nc;
      }
      if (s->blen) memcpy(out->buf + out->buflen, s->bytes, s->blen);
      out->buflen += s->blen;
      break;
    }
  }
  }
  if (out->type == ORA_BYTES) out->off[d + 1] = (uint32_t)out->buflen;
  if (null) {
    if (out->type != ORA_BYTES) memset(out->data + (size_t)d * type_width(out->type), 0, (size_t)type_width(out->type));
    vec_set_null(out, d);
  } else {
    vec_set_valid(out, d);
  }
  out->n++;
}

/* ------------------------------------------------------------------------ */
/* hashAggregator: colexec/hash_aggregator.go, hash_aggregator_tmpl.go       */
/* ------------------------------------------------------------------------ */

typedef struct int_list { int64_t *v; int64_t n, cap; } int_list;
static inline void il_push(int_list *l, int64_t x) {
  if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 4; l->v = (int64_t *)xrealloc(l->v, (size_t)l->cap * 8); }
  l->v[l->n++] = x;
}

typedef struct hash_aggregator {
  hashtable ht;
  const ora_agg_spec *aggs; int naggs;
  int ncols;
  agg_bucket *buckets; int64_t nbuckets, buckets_cap;
  int_list *eq_chains; int64_t eq_cap;
  int64_t *int_slice, *another_int_slice, *sel;
  int64_t *filter_sel, *distinct_sel;
  int err;
} hash_aggregator;

static _Thread_local hash_aggregator *g_last_agg = NULL;

/* 1 if row i's value of `col` is new for this state (and remembers it) */
static int distinct_is_new(agg_state *s, const ora_col *col, int64_t i) {
  uint8_t rec[5 + 8];
  const uint8_t *val = NULL;
  uint32_t len = 0;
  uint8_t isnull = col_is_null(col, i) ? 1 : 0;
  int64_t iv; double fv; uint8_t bv;
  if (!isnull) {
    switch (col->type) {
    case ORA_BYTES: val = col_get_bytes(col, i, &len); break;
    case ORA_FLOAT64: fv = col_get_f64(col, i); if (fv == 0) fv = 0; val = (const uint8_t *)&fv; len = 8; break;
    case ORA_BOOL: bv = col_get_bool(col, i); val = &bv; len = 1; break;
    default: iv = col_get_int(col, i); val = (const uint8_t *)&iv; len = 8; break;
    }
  }
  (void)rec;
  for (int64_t p = 0; p < s->seen_len;) {
    uint8_t n0 = s->seen[p];
    uint32_t l0;
    memcpy(&l0, s->seen + p + 1, 4);
    if (n0 == isnull && l0 == len && (len == 0 || memcmp(s->seen + p + 5, val, len) == 0)) return 0;
    p += 5 + l0;
  }
  if (s->seen_len + 5 + len > s->seen_cap) {
    int64_t nc = s->seen_cap ? s->seen_cap * 2 : 64;
    while (nc < s->seen_len + 5 + len) nc *= 2;
    s->seen = (uint8_t *)xrealloc(s->seen, (size_t)nc);
    s->seen_cap = nc;
  }
  s->seen[s->seen_len] = isnull;
  memcpy(s->seen + s->seen_len + 1, &len, 4);
  if (len) memcpy(s->seen + s->seen_len + 5, val, len);
  s->seen_len += 5 + len;
  return 1;
}

/* aggregatorHelper.performAggregation (aggregators_util.go:102-108) and its
 * FILTER flavour (:216-234, filter = rows whose BOOL filter column is true). */
static void ha_perform_aggregation(hash_aggregator *op, const ora_col *cols, const int64_t *chain, int64_t n,
                                   agg_bucket *bucket) {
  for (int f = 0; f < op->naggs && !op->err; f++) {
    const ora_agg_spec *spec = &op->aggs[f];
    const int64_t *sel = chain;
    int64_t len = n;
    if (spec->filter_col >= 0) {
      const ora_col *fc = &cols[spec->filter_col];
      len = 0;
      for (int64_t k = 0; k < n; k++)
        if (!col_is_null(fc, chain[k]) && col_get_bool(fc, chain[k])) op->filter_sel[len++] = chain[k];
      sel = op->filter_sel;
      if (len == 0) continue;
    }
    if (spec->distinct && spec->col >= 0) { /* after the FILTER (the filtering helper wraps the distinct one) */
      int64_t m = 0;
      for (int64_t k = 0; k < len; k++)
        if (distinct_is_new(&bucket->fns[f], &cols[spec->col], sel[k])) op->distinct_sel[m++] = sel[k];
      sel = op->distinct_sel;
      len = m;
      if (len == 0) continue;
    }
    int e = agg_compute(&bucket->fns[f], spec, cols, sel, len);
    if (e) op->err = e;
  }
}

static agg_bucket *ha_new_bucket(hash_aggregator *op) {
  if (op->nbuckets == op->buckets_cap) {
    op->buckets_cap = op->buckets_cap ? op->buckets_cap * 2 : 64;
    op->buckets = (agg_bucket *)xrealloc(op->buckets, (size_t)op->buckets_cap * sizeof(agg_bucket));
  }
  agg_bucket *b = &op->buckets[op->nbuckets++];
  b->fns = (agg_state *)xcalloc((size_t)op->naggs, sizeof(agg_state)); /* one object per (group, fn) */
  return b;
}

/* onlineAgg: hash_aggregator.go:354-453.  `cols` hold the buffered tuples
 * (dense, length n <= hashAggregatorMaxBuffered). */
static void ha_online_agg(hash_aggregator *op, const ora_col *cols, int64_t n) {
  hashtable *ht = &op->ht;
  if (n > op->eq_cap) {
    op->eq_chains = (int_list *)xrealloc(op->eq_chains, (size_t)n * sizeof(int_list));
    memset(op->eq_chains + op->eq_cap, 0, (size_t)(n - op->eq_cap) * sizeof(int_list));
    op->int_slice = (int64_t *)xrealloc(op->int_slice, (size_t)n * 8);
    op->another_int_slice = (int64_t *)xrealloc(op->another_int_slice, (size_t)n * 8);
    op->sel = (int64_t *)xrealloc(op->sel, (size_t)n * 8);
    op->filter_sel = (int64_t *)xrealloc(op->filter_sel, (size_t)n * 8);
    op->distinct_sel = (int64_t *)xrealloc(op->distinct_sel, (size_t)n * 8);
    op->eq_cap = n;
  }
  /* Step 1 */
  ht_compute_hash_and_build_chains(ht, cols, n, NULL);
  ht_find_buckets(ht, n, NULL, ht->p_first, ht->p_next, ht_check_probe_for_distinct, 0, 1);

  /* Step 2: populateEqChains (hash_aggregator_tmpl.go:26-111) */
  int64_t *head_to_chain = op->int_slice;
  memset(head_to_chain, 0, (size_t)n * 8);
  int64_t *sel = op->sel;
  int64_t eq_count = 0;
  for (int64_t i = 0; i < n; i++) {
    uint32_t head = ht->head_id[i];
    int64_t id = head_to_chain[head - 1];
    if (id == 0) {
      ht->hash_buffer[eq_count] = ht->hash_buffer[i];
      sel[eq_count] = i;
      il_push(&op->eq_chains[eq_count], i);
      eq_count++;
      head_to_chain[head - 1] = eq_count;
    } else {
      il_push(&op->eq_chains[id - 1], i);
    }
  }
  int64_t *eq_heads = op->int_slice; /* copy of the heads selection (reuses intSlice) */
  memcpy(eq_heads, sel, (size_t)eq_count * 8);

  /* Step 3 */
  if (op->nbuckets > 0) {
    ht_find_buckets(ht, eq_count, sel, ht->first, ht->next, ht_check_build_for_aggregation, 1, 0);
    for (int64_t slot = 0; slot < eq_count; slot++) {
      uint32_t head = ht->head_id[slot];
      if (head != 0) {
        int_list *chain = &op->eq_chains[slot];
        ha_perform_aggregation(op, cols, chain->v, chain->n, &op->buckets[head - 1]);
        chain->n = 0;
      }
    }
  }
  /* Step 4 */
  int64_t *new_heads = op->another_int_slice;
  int64_t new_count = 0;
  for (int64_t slot = 0; slot < eq_count; slot++) {
    int_list *chain = &op->eq_chains[slot];
    if (chain->n > 0) {
      agg_bucket *b = ha_new_bucket(op);
      ha_perform_aggregation(op, cols, chain->v, chain->n, b);
      new_heads[new_count] = eq_heads[slot];
      ht->hash_buffer[new_count] = ht->hash_buffer[slot];
      new_count++;
      chain->n = 0;
    }
  }
  if (new_count > 0) {
    memcpy(sel, new_heads, (size_t)new_count * 8);
    ht_append_all_distinct(ht, cols, sel, new_count);
  }
}

static void ha_destroy(hash_aggregator *op) {
  if (!op) return;
  ht_free(&op->ht);
  for (int64_t b = 0; b < op->nbuckets; b++) {
    for (int f = 0; f < op->naggs; f++) { free(op->buckets[b].fns[f].bytes); free(op->buckets[b].fns[f].seen); }
    free(op->buckets[b].fns);
  }
  free(op->buckets);
  for (int64_t i = 0; i < op->eq_cap; i++) free(op->eq_chains[i].v);
  free(op->eq_chains); free(op->int_slice); free(op->another_int_slice); free(op->sel); free(op->filter_sel); free(op->distinct_sel);
  free(op);
}

ora_result *ora_hash_agg(const ora_col *cols, int ncols, int64_t n,
                         const int32_t *group_cols, int ngroup,
                         const ora_agg_spec *aggs, int naggs,
                         int32_t batch_size, int32_t max_buffered, uint32_t initial_buckets,
                         int32_t *err) {
  if (g_last_agg) { ha_destroy(g_last_agg); g_last_agg = NULL; }
  if (err) *err = 0;
  for (int f = 0; f < naggs; f++) {
    const ora_agg_spec *s = &aggs[f];
    int t = s->col >= 0 ? cols[s->col].type : -1;
    int ok = 1;
    if (s->func == ORA_SUM && t != ORA_FLOAT64 && !is_int_type(t)) ok = 0;
    if (s->func == ORA_AVG && t != ORA_FLOAT64) ok = 0; /* AVG over INT: SUM + COUNT and a DECIMAL division in the shell */
    if (s->func == ORA_SUM_INT && !is_int_type(t)) ok = 0;
    if ((s->func == ORA_BOOL_AND || s->func == ORA_BOOL_OR) && t != ORA_BOOL) ok = 0;
    if (!ok) { if (err) *err = 2; return NULL; }
  }
  hash_aggregator *op = (hash_aggregator *)xcalloc(1, sizeof(*op));
  g_last_agg = op;
  op->aggs = aggs; op->naggs = naggs; op->ncols = ncols;
  int types[64];
  for (int c = 0; c < ncols; c++) types[c] = cols[c].type;
  /* Init: hash_aggregator.go:250-273 (load factor 0.1, 256 buckets, NULLs equal) */
  ht_init(&op->ht, 0.1, initial_buckets ? initial_buckets : 256, types, ncols, group_cols, ngroup, 1,
          BUILD_DISTINCT, PROBE_DEFAULT);

  /* bufferingState.tuples: AppendOnlyBufferedBatch of all input columns */
  vec *buffered = (vec *)xcalloc((size_t)ncols, sizeof(vec));
  for (int c = 0; c < ncols; c++) vec_init(&buffered[c], types[c]);
  ora_col *bview = (ora_col *)xcalloc((size_t)ncols, sizeof(ora_col));

  /* getNext (hash_aggregator_tmpl.go:171-398), partialOrder=false: buffer input
   * batches until hashAggregatorMaxBuffered tuples, aggregate, repeat. */
  int64_t pos = 0;
  while (pos < n && !op->err) {
    int64_t pend_end = pos + batch_size < n ? pos + batch_size : n; /* pendingBatch */
    int64_t unprocessed = pos;
    while (unprocessed < pend_end && !op->err) {
      int64_t room = max_buffered - buffered[0].n;
      int64_t take = pend_end - unprocessed < room ? pend_end - unprocessed : room;
      for (int c = 0; c < ncols; c++) vec_append_range(&buffered[c], &cols[c], NULL, unprocessed, unprocessed + take);
      unprocessed += take;
      if (buffered[0].n == max_buffered) {
        for (int c = 0; c < ncols; c++) vec_view(&buffered[c], &bview[c]);
        ha_online_agg(op, bview, buffered[0].n);
        for (int c = 0; c < ncols; c++) { buffered[c].n = 0; buffered[c].buflen = 0; buffered[c].has_nulls = 0;
          if (buffered[c].valid) memset(buffered[c].valid, 0xff, (size_t)((buffered[c].cap + 7) / 8)); }
      }
    }
    pos = pend_end;
  }
  if (ncols > 0 && buffered[0].n > 0 && !op->err) {
    for (int c = 0; c < ncols; c++) vec_view(&buffered[c], &bview[c]);
    ha_online_agg(op, bview, buffered[0].n);
  }
  for (int c = 0; c < ncols; c++) vec_free(&buffered[c]);
  free(buffered);
  free(bview);

  if (op->err) { if (err) *err = op->err; return NULL; }

  /* hashAggregatorOutputting: one row per bucket, one column per aggregation */
  int out_types[64];
  for (int f = 0; f < naggs; f++) out_types[f] = agg_output_type(&aggs[f], cols);
  ora_result *out = result_new(naggs, out_types);
  for (int64_t b = 0; b < op->nbuckets; b++)
    for (int f = 0; f < naggs; f++) agg_flush(&op->buckets[b].fns[f], &aggs[f], cols, &out->cols[f]);
  return out;
}

void ora_test_last_agg_state(uint32_t *first, int32_t *nfirst, uint32_t *next, int64_t *nkeys) {
  hash_aggregator *op = g_last_agg;
  if (!op) return;
  hashtable *ht = &op->ht;
  if (nfirst) *nfirst = (int32_t)ht->num_buckets;
  if (first) memcpy(first, ht->first, (size_t)ht->num_buckets * 4);
  if (nkeys) *nkeys = ht->next_len - 1;
  if (next) memcpy(next, ht->next, (size_t)ht->next_len * 4);
}
