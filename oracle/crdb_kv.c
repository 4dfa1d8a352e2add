/*
 * crdb_kv.c -- TEST INFRASTRUCTURE ONLY (see crdb_oracle.h).
 *
 * CPU restatement of the KV -> column decode path of the reference's cFetcher
 * for primary indexes -- one KV per row, or several column families per row with TUPLE and
 * single-column ("legacy") values and the MVCC extended value header --, plus the matching ENCODER so that
 * decode parity can be pinned by round trips (the reference holds no KV byte
 * fixtures for whole rows).  The primitive codecs are
 * pinned by the literal vectors of pkg/util/encoding/encoding_test.go:280-324,
 * 500-515, float_test.go:17-69 and by the annotated dump of
 * docs/tech-notes/encoding.md:187-233,380-384 (tests/golden/encoding_vectors.json).
 *
 * Follows:
 *   pkg/util/encoding/encoding.go   EncodeVarintAscending :317, DecodeVarintAscending :374,
 *                                   EncodeUvarintAscending :417, DecodeUvarintAscending :569,
 *                                   EncodeBytesAscending :645, decodeBytesInternal :788,
 *                                   EncodeNonsortingStdlibVarint :2608, EncodeNonsortingUvarint :2636,
 *                                   DecodeNonsortingUvarint :2663, EncodeValueTag :2718, DecodeValueTag :3032,
 *                                   PeekValueLengthWithOffsetsAndType :3420, Type constants :1857-1911
 *   pkg/util/encoding/float.go      EncodeFloatAscending :31, DecodeFloatAscending :61, Descending :52,89
 *   pkg/storage/enginepb/decode.go  SplitMVCCKey :26, ScanDecodeKeyValueNoTS :108
 *   pkg/roachpb/data.go             checksumSize/tagPos/headerSize :284-286, ValueType_TUPLE = 10
 *   pkg/keys/keys.go                MakeFamilyKey :950-961 (family 0 => one byte 0x88)
 *   pkg/sql/colencoding             DecodeKeyValsToCols key_encoding.go:43, decodeTableKeyToCol :87,
 *                                   DecodeTableValueToCol value_encoding.go:24
 *   pkg/sql/colfetcher/cfetcher.go  NextBatch :758 (MaxKeysPerRow == 1 path :937, unfinished rows :946-1020), processValue :1136,
 *                                   processValueSingle :1280, processValueBytes :1313, fillNulls :1401,
 *                                   getCurrentColumnFamilyID :1447
 *   pkg/sql/colencoding             UnmarshalColumnValueToCol key_encoding.go:226
 *   pkg/roachpb/data.go             extended encoding :288-289,336-338,436-489; GetBytes/GetFloat/GetBool/GetInt :689-790
 *   pkg/sql/rowenc/index_encoding.go EncodePrimaryIndex :1169-1265 (what the encoder writes per family)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "crdb_oracle.h"

enum { IntMin = 0x80, intMaxWidth = 8, intZero = IntMin + intMaxWidth, IntMax = 0xfd, intSmall = IntMax - intZero - intMaxWidth };
enum { encodedNull = 0x00, encodedNullDesc = 0xff, floatNaN = 0x02, floatNeg = 0x03, floatZero = 0x04, floatPos = 0x05,
       floatNaNDesc = 0x06, bytesMarker = 0x12, bytesDescMarker = 0x13 };
enum { T_Null = 1, T_Int = 3, T_Float = 4, T_Decimal = 5, T_Bytes = 6, T_Time = 8, T_Duration = 9, T_True = 10, T_False = 11,
       T_UUID = 12, T_Array = 13, T_IPAddr = 14, T_Sentinel = 15 };
enum { ValueType_TUPLE = 10 };

/* ---------------------------------------------------------------- key side: ints */
int ora_enc_uvarint_asc(uint8_t *b, uint64_t v) { /* encoding.go:417-442 */
  if (v <= (uint64_t)intSmall) { b[0] = (uint8_t)(intZero + v); return 1; }
  int nb = 1;
  while (nb < 8 && (v >> (8 * nb)) != 0) nb++;
  b[0] = (uint8_t)(IntMax - 8 + nb);
  for (int i = 0; i < nb; i++) b[1 + i] = (uint8_t)(v >> (8 * (nb - 1 - i)));
  return 1 + nb;
}
int ora_enc_varint_asc(uint8_t *b, int64_t v) { /* encoding.go:317-343 */
  if (v >= 0) return ora_enc_uvarint_asc(b, (uint64_t)v);
  int nb = 1; /* smallest nb with v >= -(2^(8nb) - 1) */
  while (nb < 8 && v < -(int64_t)((1ull << (8 * nb)) - 1)) nb++;
  b[0] = (uint8_t)(IntMin + 8 - nb);
  for (int i = 0; i < nb; i++) b[1 + i] = (uint8_t)((uint64_t)v >> (8 * (nb - 1 - i)));
  return 1 + nb;
}
int ora_enc_varint_desc(uint8_t *b, int64_t v) { return ora_enc_varint_asc(b, ~v); } /* :347 */

int ora_dec_uvarint_asc(const uint8_t *b, int64_t len, uint64_t *out) { /* :569-591 */
  if (len <= 0) return -1;
  int length = (int)b[0] - intZero;
  if (length <= intSmall) { *out = (uint64_t)(int64_t)length; return length < 0 ? -1 : 1; }
  length -= intSmall;
  if (length < 0 || length > 8 || len - 1 < length) return -1;
  uint64_t v = 0;
  for (int i = 0; i < length; i++) v = (v << 8) | b[1 + i];
  *out = v;
  return 1 + length;
}
int ora_dec_varint_asc(const uint8_t *b, int64_t len, int64_t *out) { /* :374-403 */
  if (len <= 0) return -1;
  int length = (int)b[0] - intZero;
  if (length < 0) {
    length = -length;
    if (len - 1 < length) return -1;
    int64_t v = 0;
    for (int i = 0; i < length; i++) v = (int64_t)(((uint64_t)v << 8) | (uint8_t)~b[1 + i]);
    *out = ~v;
    return 1 + length;
  }
  uint64_t u;
  int n = ora_dec_uvarint_asc(b, len, &u);
  if (n < 0 || u > (uint64_t)INT64_MAX) return -1;
  *out = (int64_t)u;
  return n;
}
int ora_dec_varint_desc(const uint8_t *b, int64_t len, int64_t *out) { /* :407-410 */
  int n = ora_dec_varint_asc(b, len, out);
  if (n > 0) *out = ~*out;
  return n;
}

/* ---------------------------------------------------------------- key side: floats */
static void put_be64(uint8_t *b, uint64_t u) { for (int i = 0; i < 8; i++) b[i] = (uint8_t)(u >> (56 - 8 * i)); }
static uint64_t get_be64(const uint8_t *b) { uint64_t u = 0; for (int i = 0; i < 8; i++) u = (u << 8) | b[i]; return u; }

int ora_enc_float_asc(uint8_t *b, double f) { /* float.go:31-49 */
  if (isnan(f)) { b[0] = floatNaN; return 1; }
  if (f == 0) { b[0] = floatZero; return 1; }
  uint64_t u;
  memcpy(&u, &f, 8);
  if (u & (1ull << 63)) { u = ~u; b[0] = floatNeg; } else b[0] = floatPos;
  put_be64(b + 1, u);
  return 9;
}
int ora_enc_float_desc(uint8_t *b, double f) { /* float.go:52-57 */
  if (isnan(f)) { b[0] = floatNaNDesc; return 1; }
  return ora_enc_float_asc(b, -f);
}
int ora_dec_float_asc(const uint8_t *b, int64_t len, double *out) { /* float.go:61-86 */
  if (len <= 0) return -1;
  uint64_t u;
  switch (b[0]) {
    case floatNaN: case floatNaNDesc: *out = NAN; return 1;
    case floatZero: *out = 0; return 1;
    case floatNeg: if (len < 9) return -1; u = ~get_be64(b + 1); memcpy(out, &u, 8); return 9;
    case floatPos: if (len < 9) return -1; u = get_be64(b + 1); memcpy(out, &u, 8); return 9;
    default: return -1;
  }
}
int ora_dec_float_desc(const uint8_t *b, int64_t len, double *out) { /* float.go:89-98 */
  int n = ora_dec_float_asc(b, len, out);
  if (n > 0 && *out != 0 && !isnan(*out)) *out = -*out;
  return n;
}

/* ---------------------------------------------------------------- key side: bytes */
int64_t ora_enc_bytes_asc(uint8_t *b, const uint8_t *data, int64_t n) { /* encoding.go:645-703 */
  int64_t o = 0;
  b[o++] = bytesMarker;
  for (int64_t i = 0; i < n; i++) {
    b[o++] = data[i];
    if (data[i] == 0x00) b[o++] = 0xff; /* escape, escaped00 */
  }
  b[o++] = 0x00; /* escape */
  b[o++] = 0x01; /* escapedTerm */
  return o;
}
int64_t ora_enc_bytes_desc(uint8_t *b, const uint8_t *data, int64_t n) { /* :709-715 */
  int64_t o = ora_enc_bytes_asc(b, data, n);
  b[0] = bytesDescMarker;
  for (int64_t i = 1; i < o; i++) b[i] = (uint8_t)~b[i];
  return o;
}
/* decodeBytesInternal (:788-824); desc: complemented escapes, result complemented (:776-782).
 * Returns bytes consumed (or -1); *outlen = decoded length; out may be NULL (length only). */
int64_t ora_dec_bytes(const uint8_t *b, int64_t len, int desc, uint8_t *out, int64_t *outlen) {
  const uint8_t marker = desc ? bytesDescMarker : bytesMarker;
  const uint8_t esc = desc ? 0xff : 0x00, term = desc ? 0xfe : 0x01, e00 = desc ? 0x00 : 0xff;
  if (len <= 0 || b[0] != marker) return -1;
  int64_t i = 1, o = 0;
  for (;;) {
    if (i >= len) return -1; /* did not find terminator */
    uint8_t c = b[i];
    if (c != esc) {
      if (out) out[o] = desc ? (uint8_t)~c : c;
      o++, i++;
      continue;
    }
    if (i + 1 >= len) return -1; /* malformed escape */
    uint8_t v = b[i + 1];
    if (v == term) { *outlen = o; return i + 2; }
    if (v != e00) return -1; /* unknown escape sequence */
    if (out) out[o] = desc ? (uint8_t)~0xff : 0x00; /* escapedFF = 0x00; after complement of the desc result: ~0xff */
    o++, i += 2;
  }
}

/* ---------------------------------------------------------------- value side */
int ora_enc_nonsorting_uvarint(uint8_t *b, uint64_t x) { /* encoding.go:2636-2659: 7-bit groups, MSB first */
  int groups = 1;
  while (groups < 10 && (x >> (7 * groups)) != 0) groups++;
  for (int i = 0; i < groups; i++) {
    uint8_t g = (uint8_t)((x >> (7 * (groups - 1 - i))) & 0x7f);
    b[i] = i == groups - 1 ? g : (uint8_t)(0x80 | g);
  }
  return groups;
}
int ora_dec_nonsorting_uvarint(const uint8_t *b, int64_t len, uint64_t *out) { /* :2663-2673 */
  uint64_t v = 0;
  for (int64_t i = 0; i < len; i++) {
    v += (uint64_t)(b[i] & 0x7f);
    if (b[i] < 0x80) { *out = v; return (int)(i + 1); }
    v <<= 7;
  }
  return -1; /* the reference returns length 0 */
}
int ora_enc_stdlib_varint(uint8_t *b, int64_t x) { /* binary.PutVarint: zig-zag, LSB group first (:2608) */
  uint64_t ux = (uint64_t)x << 1;
  if (x < 0) ux = ~ux;
  int i = 0;
  while (ux >= 0x80) { b[i++] = (uint8_t)(ux | 0x80); ux >>= 7; }
  b[i++] = (uint8_t)ux;
  return i;
}
int ora_dec_stdlib_varint(const uint8_t *b, int64_t len, int64_t *out) { /* binary.Varint (:2617) */
  uint64_t ux = 0;
  int s = 0;
  for (int64_t i = 0; i < len && i < 10; i++) {
    if (b[i] < 0x80) {
      if (i == 9 && b[i] > 1) return -1; /* overflow */
      ux |= (uint64_t)b[i] << s;
      int64_t x = (int64_t)(ux >> 1);
      if (ux & 1) x = ~x;
      *out = x;
      return (int)(i + 1);
    }
    ux |= (uint64_t)(b[i] & 0x7f) << s;
    s += 7;
  }
  return -1;
}
int ora_enc_value_tag(uint8_t *b, uint32_t delta, int typ) { /* :2718-2730 */
  if (typ >= T_Sentinel) {
    int n = ora_enc_nonsorting_uvarint(b, ((uint64_t)delta << 4) | T_Sentinel);
    return n + ora_enc_nonsorting_uvarint(b + n, (uint64_t)typ);
  }
  if (delta == 0) { b[0] = (uint8_t)typ; return 1; }
  return ora_enc_nonsorting_uvarint(b, ((uint64_t)delta << 4) | (uint64_t)typ);
}
/* DecodeValueTag (:3032-3060): returns dataOffset or -1 */
static int dec_value_tag(const uint8_t *b, int64_t len, uint32_t *delta, int *typ) {
  uint64_t tag;
  int n = ora_dec_nonsorting_uvarint(b, len, &tag);
  if (n < 0) return -1;
  *delta = (uint32_t)(tag >> 4);
  *typ = (int)(tag & 0xf);
  if (*typ == T_Sentinel) {
    uint64_t t2;
    int n2 = ora_dec_nonsorting_uvarint(b + n, len - n, &t2);
    if (n2 < 0) return -1;
    *typ = (int)t2;
    n += n2;
  }
  return n;
}
/* PeekValueLengthWithOffsetsAndType (:3420-3486) for the types a SQL table row can hold here */
static int64_t peek_value_len(const uint8_t *b, int64_t len, int data_off, int typ) {
  const uint8_t *d = b + data_off;
  int64_t rem = len - data_off;
  uint64_t u;
  int64_t s;
  int n, n2;
  switch (typ) {
    case T_Null: case T_True: case T_False: return data_off;
    case T_Int: n = ora_dec_stdlib_varint(d, rem, &s); return n < 0 ? -1 : data_off + n;
    case T_Float: return data_off + 8;
    case T_Bytes: case T_Array: n = ora_dec_nonsorting_uvarint(d, rem, &u); return n < 0 ? -1 : data_off + n + (int64_t)u;
    case T_Decimal: { /* binary.Uvarint length + bytes (:3464-3466) */
      uint64_t x = 0; int sh = 0; int i = 0;
      for (; i < rem && i < 10; i++) { x |= (uint64_t)(d[i] & 0x7f) << sh; sh += 7; if (d[i] < 0x80) break; }
      if (i >= rem) return -1;
      return data_off + i + 1 + (int64_t)x;
    }
    case T_Time: n = ora_dec_stdlib_varint(d, rem, &s); if (n < 0) return -1;
      n2 = ora_dec_stdlib_varint(d + n, rem - n, &s); return n2 < 0 ? -1 : data_off + n + n2;
    case T_Duration: { int tot = 0; for (int k = 0; k < 3; k++) { n = ora_dec_stdlib_varint(d + tot, rem - tot, &s); if (n < 0) return -1; tot += n; } return data_off + tot; }
    case T_UUID: return data_off + 16;
    default: return -1;
  }
}

/* ---------------------------------------------------------------- table description */
typedef struct ora_kv_table { /* what the ENCODER writes: a table with one column family */
  int32_t table_id, index_id;
  int32_t n_key_cols, n_val_cols;
  const int32_t *key_types; /* ORA_* */
  const int32_t *key_dirs;  /* 0 ascending, 1 descending (catenumpb.IndexColumn_Direction) */
  const int32_t *val_col_ids; /* strictly increasing column ids */
  const int32_t *val_types;
} ora_kv_table;

typedef struct ora_kv_fetch { /* what the DECODER is asked for (fetchpb.IndexFetchSpec subset) */
  int32_t key_prefix_len; /* IndexFetchSpec.KeyPrefixLength */
  int32_t n_key_cols;
  const int32_t *key_types;
  const int32_t *key_dirs;
  const int32_t *key_out; /* output ordinal or -1 (not fetched) */
  int32_t n_val_cols;     /* fetched value columns, by increasing column id */
  int32_t n_out;
  const int32_t *val_col_ids;
  const int32_t *val_types;
  const int32_t *val_out;
} ora_kv_fetch;

typedef struct bytebuf { uint8_t *p; int64_t n, cap; } bytebuf;
static void bb_reserve(bytebuf *b, int64_t extra) {
  if (b->n + extra <= b->cap) return;
  int64_t nc = b->cap ? b->cap : 4096;
  while (nc < b->n + extra) nc *= 2;
  b->p = (uint8_t *)realloc(b->p, (size_t)nc);
  if (!b->p) abort();
  b->cap = nc;
}
static int is_null(const ora_col *c, int64_t i) { return c->valid && !((c->valid[i >> 3] >> (i & 7)) & 1); }
static int64_t get_int(const ora_col *c, int64_t i) {
  switch (c->type) {
    case ORA_INT16: return ((const int16_t *)c->data)[i];
    case ORA_INT32: return ((const int32_t *)c->data)[i];
    case ORA_BOOL: return ((const uint8_t *)c->data)[i] != 0;
    default: return ((const int64_t *)c->data)[i];
  }
}

/* Column families (descpb.ColumnFamilyDescriptor; fetchpb.IndexFetchSpec.{MaxKeysPerRow, MaxFamilyID,
 * FamilyDefaultColumns}, index_fetch.proto:112-190).  NULL where a function takes one = a table with the single family 0. */
typedef struct ora_kv_families {
  int32_t n_families;             /* families in ID order; family_ids[0] == 0 (the sentinel family always exists) */
  int32_t mvcc_header_every;      /* ENCODER: every k-th KV's value carries an MVCC extended-encoding header (0: never) */
  const int32_t *family_ids;
  const int32_t *default_col_ids; /* DefaultColumnID per family (0: none) */
  const int32_t *val_family;      /* ENCODER: per value column of the table, index into family_ids */
} ora_kv_families;

static void put_value_header(bytebuf *val, int64_t salt, int tag, int ext_header) {
  /* roachpb.Value.RawBytes: [extended prelude: BE32 header length, 0x65, header bytes] checksum[4] tag[1] (data.go:284-289,
   * 316, 479-481).  The header bytes stand for a marshalled enginepb.MVCCValueHeader: the decoder skips them by length. */
  if (ext_header) {
    uint32_t hl = 3 + (uint32_t)(salt % 5);
    bb_reserve(val, 5 + hl);
    val->p[val->n++] = (uint8_t)(hl >> 24); val->p[val->n++] = (uint8_t)(hl >> 16); val->p[val->n++] = (uint8_t)(hl >> 8); val->p[val->n++] = (uint8_t)hl;
    val->p[val->n++] = 101; /* ValueType_MVCC_EXTENDED_ENCODING_SENTINEL */
    for (uint32_t i = 0; i < hl; i++) val->p[val->n++] = (uint8_t)(0x0a + i * 7 + (uint32_t)salt); /* arbitrary */
  }
  bb_reserve(val, 8);
  uint32_t ck = (uint32_t)((uint64_t)salt * 2654435761u + 12345u); /* not verified by scans; must be skipped */
  val->p[val->n++] = (uint8_t)(ck >> 24); val->p[val->n++] = (uint8_t)(ck >> 16); val->p[val->n++] = (uint8_t)(ck >> 8); val->p[val->n++] = (uint8_t)ck;
  val->p[val->n++] = (uint8_t)tag;
}

/* Encode n rows as MVCCScan batch responses (decode.go:77-103 framing), the way rowenc.EncodePrimaryIndex writes a row
 * (rowenc/index_encoding.go:1169-1265): one KV per column family in family-ID order,
 *   key   = uvarint(table) uvarint(index) keycols... familysuffix | 0x00 wall[8] [logical[4]] (len+1)
 *           familysuffix = 0x88 for family 0, else uvarint(id) uvarint(length of that encoding) (keys.go:950-961)
 *   value = family 0 and families of several columns: checksum[4] 0x0A {tag data}* -- column ids as deltas from the previous
 *           column OF THIS VALUE; NULL columns are omitted (explicit_null_every k > 0 writes every k-th NULL as an explicit
 *           Null-typed entry, which is also legal input); a family != 0 whose tuple is empty writes no KV (:1247);
 *           a family != 0 whose ONE column is its DefaultColumnID: valueside.MarshalLegacy -- checksum[4] tag data with
 *           tag INT (stdlib varint; BOOL as 0/1), FLOAT (8 bytes big-endian) or BYTES (the raw bytes); no KV when NULL (:1203-1227).
 * A new batch starts at a ROW boundary once the current one reaches max_batch_bytes.  Output: malloc'd bytes,
 * batch_offsets[nbatches+1]. */
int ora_kv_encode2(const ora_kv_table *t, const ora_kv_families *fam, const ora_col *key_cols, const ora_col *val_cols, int64_t n,
                   int64_t max_batch_bytes, int32_t explicit_null_every, uint8_t **out, int64_t *out_len,
                   int64_t **batch_offsets, int64_t *nbatches) {
  bytebuf bb = {0, 0, 0};
  int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 2));
  int64_t nb = 0, batch_start = 0, nulls_seen = 0, kv_seq = 0;
  offs[0] = 0;
  bytebuf key = {0, 0, 0}, val = {0, 0, 0}, rowbuf = {0, 0, 0};
  const int32_t zero = 0;
  const int nfam = fam ? fam->n_families : 1;
  for (int64_t r = 0; r < n; r++) {
    key.n = 0;
    rowbuf.n = 0;
    bb_reserve(&key, 32);
    key.n += ora_enc_uvarint_asc(key.p + key.n, (uint64_t)t->table_id);
    key.n += ora_enc_uvarint_asc(key.p + key.n, (uint64_t)t->index_id);
    for (int c = 0; c < t->n_key_cols; c++) {
      const ora_col *col = &key_cols[c];
      const int desc = t->key_dirs[c];
      if (is_null(col, r)) { bb_reserve(&key, 1); key.p[key.n++] = desc ? encodedNullDesc : encodedNull; continue; }
      switch (t->key_types[c]) {
        case ORA_FLOAT64:
          bb_reserve(&key, 9);
          key.n += desc ? ora_enc_float_desc(key.p + key.n, ((const double *)col->data)[r])
                        : ora_enc_float_asc(key.p + key.n, ((const double *)col->data)[r]);
          break;
        case ORA_BYTES: {
          uint32_t a = col->offsets[r], e = col->offsets[r + 1];
          bb_reserve(&key, 2 * (int64_t)(e - a) + 3);
          key.n += desc ? ora_enc_bytes_desc(key.p + key.n, (const uint8_t *)col->data + a, e - a)
                        : ora_enc_bytes_asc(key.p + key.n, (const uint8_t *)col->data + a, e - a);
          break;
        }
        default: /* bool, ints: varint (key_encoding.go:113-135) */
          bb_reserve(&key, 9);
          key.n += desc ? ora_enc_varint_desc(key.p + key.n, get_int(col, r)) : ora_enc_varint_asc(key.p + key.n, get_int(col, r));
      }
    }
    const int64_t row_key_len = key.n;
    for (int fi = 0; fi < nfam; fi++) {
      const int32_t fam_id = fam ? fam->family_ids[fi] : 0;
      const int32_t def_col = fam ? fam->default_col_ids[fi] : 0;
      int ncols_in_fam = 0, only = -1;
      for (int c = 0; c < t->n_val_cols; c++)
        if ((fam ? fam->val_family[c] : zero) == fi) { ncols_in_fam++; only = c; }
      const int ext = fam && fam->mvcc_header_every > 0 && kv_seq % fam->mvcc_header_every == 0;
      val.n = 0;
      if (fam_id != 0 && ncols_in_fam == 1 && t->val_col_ids[only] == def_col) { /* legacy single-column value */
        const ora_col *col = &val_cols[only];
        if (is_null(col, r)) continue; /* no KV */
        switch (t->val_types[only]) {
          case ORA_FLOAT64: {
            put_value_header(&val, r * 7 + fi, 2 /* FLOAT */, ext);
            uint64_t u; memcpy(&u, (const double *)col->data + r, 8);
            bb_reserve(&val, 8); put_be64(val.p + val.n, u); val.n += 8;
            break;
          }
          case ORA_BYTES: {
            uint32_t a = col->offsets[r], e = col->offsets[r + 1];
            put_value_header(&val, r * 7 + fi, 3 /* BYTES */, ext);
            bb_reserve(&val, (int64_t)(e - a) + 1);
            memcpy(val.p + val.n, (const uint8_t *)col->data + a, e - a); val.n += e - a;
            break;
          }
          default: /* ints, dates, bool: Value.SetInt / SetBool (data.go) */
            put_value_header(&val, r * 7 + fi, 1 /* INT */, ext);
            bb_reserve(&val, 12);
            val.n += ora_enc_stdlib_varint(val.p + val.n, get_int(col, r));
        }
      } else {
        put_value_header(&val, r * 7 + fi, ValueType_TUPLE, ext);
        const int64_t hdr = val.n;
        int32_t last_id = 0;
        for (int c = 0; c < t->n_val_cols; c++) {
          if ((fam ? fam->val_family[c] : zero) != fi) continue;
          const ora_col *col = &val_cols[c];
          const int32_t id = t->val_col_ids[c];
          if (is_null(col, r)) {
            nulls_seen++;
            if (explicit_null_every > 0 && nulls_seen % explicit_null_every == 0) {
              bb_reserve(&val, 12);
              val.n += ora_enc_value_tag(val.p + val.n, (uint32_t)(id - last_id), T_Null);
              last_id = id;
            }
            continue;
          }
          bb_reserve(&val, 32);
          switch (t->val_types[c]) {
            case ORA_BOOL:
              val.n += ora_enc_value_tag(val.p + val.n, (uint32_t)(id - last_id), get_int(col, r) ? T_True : T_False);
              break;
            case ORA_FLOAT64: {
              val.n += ora_enc_value_tag(val.p + val.n, (uint32_t)(id - last_id), T_Float);
              uint64_t u; memcpy(&u, (const double *)col->data + r, 8);
              put_be64(val.p + val.n, u); val.n += 8;
              break;
            }
            case ORA_BYTES: {
              uint32_t a = col->offsets[r], e = col->offsets[r + 1];
              val.n += ora_enc_value_tag(val.p + val.n, (uint32_t)(id - last_id), T_Bytes);
              bb_reserve(&val, (int64_t)(e - a) + 12);
              val.n += ora_enc_nonsorting_uvarint(val.p + val.n, e - a);
              memcpy(val.p + val.n, (const uint8_t *)col->data + a, e - a); val.n += e - a;
              break;
            }
            default:
              val.n += ora_enc_value_tag(val.p + val.n, (uint32_t)(id - last_id), T_Int);
              val.n += ora_enc_stdlib_varint(val.p + val.n, get_int(col, r));
          }
          last_id = id;
        }
        if (fam_id != 0 && val.n == hdr) continue; /* empty tuple of a later family: no KV (:1247) */
      }
      /* the KV's key: row key + family suffix + MVCC timestamp */
      key.n = row_key_len;
      bb_reserve(&key, 32);
      if (fam_id == 0) key.p[key.n++] = (uint8_t)intZero;
      else {
        int fl = ora_enc_uvarint_asc(key.p + key.n, (uint64_t)fam_id);
        key.n += fl;
        key.n += ora_enc_uvarint_asc(key.p + key.n, (uint64_t)fl);
      }
      key.p[key.n++] = 0x00; /* MVCC sentinel */
      put_be64(key.p + key.n, (uint64_t)(1600000000000000000ll + r)); key.n += 8;
      if (r % 3 != 0) { /* with a logical component: 12-byte timestamp */
        uint32_t lg = (uint32_t)(r % 1000);
        key.p[key.n++] = (uint8_t)(lg >> 24); key.p[key.n++] = (uint8_t)(lg >> 16); key.p[key.n++] = (uint8_t)(lg >> 8); key.p[key.n++] = (uint8_t)lg;
        key.p[key.n++] = 13;
      } else key.p[key.n++] = 9;
      const int64_t rec = 8 + key.n + val.n;
      bb_reserve(&rowbuf, rec);
      uint32_t vl = (uint32_t)val.n, kl = (uint32_t)key.n;
      memcpy(rowbuf.p + rowbuf.n, &vl, 4); memcpy(rowbuf.p + rowbuf.n + 4, &kl, 4); /* little-endian host */
      memcpy(rowbuf.p + rowbuf.n + 8, key.p, (size_t)key.n);
      memcpy(rowbuf.p + rowbuf.n + 8 + key.n, val.p, (size_t)val.n);
      rowbuf.n += rec;
      kv_seq++;
    }
    /* rows stay whole inside a batch (kvpb.ScanOptions.WholeRowsOfSize for multi-family indexes) */
    if (bb.n > batch_start && bb.n - batch_start + rowbuf.n > max_batch_bytes) { offs[++nb] = bb.n; batch_start = bb.n; }
    bb_reserve(&bb, rowbuf.n);
    memcpy(bb.p + bb.n, rowbuf.p, (size_t)rowbuf.n);
    bb.n += rowbuf.n;
  }
  if (bb.n > batch_start || nb == 0) offs[++nb] = bb.n;
  free(key.p); free(val.p); free(rowbuf.p);
  *out = bb.p; *out_len = bb.n; *batch_offsets = offs; *nbatches = nb;
  return 0;
}
int ora_kv_encode(const ora_kv_table *t, const ora_col *key_cols, const ora_col *val_cols, int64_t n,
                  int64_t max_batch_bytes, int32_t explicit_null_every, uint8_t **out, int64_t *out_len,
                  int64_t **batch_offsets, int64_t *nbatches) {
  return ora_kv_encode2(t, NULL, key_cols, val_cols, n, max_batch_bytes, explicit_null_every, out, out_len, batch_offsets, nbatches);
}
void ora_kv_free(void *p) { free(p); }

/* ---------------------------------------------------------------- decoder */
typedef struct kvcol { int type; int64_t n, cap; uint8_t *data; uint8_t *valid; int has_nulls; uint32_t *off; uint8_t *buf; int64_t buflen, bufcap; } kvcol;
typedef struct ora_kv_result { int ncols; int64_t nrows; kvcol *cols; } ora_kv_result;
static int tw(int t);
static int tw(int t) { return t == ORA_BOOL ? 1 : t == ORA_INT16 ? 2 : t == ORA_INT32 ? 4 : 8; }
static void kc_reserve(kvcol *c, int64_t need) {
  if (need <= c->cap) return;
  int64_t nc = c->cap ? c->cap : 1024;
  while (nc < need) nc *= 2;
  if (c->type == ORA_BYTES) { c->off = (uint32_t *)realloc(c->off, (size_t)(nc + 1) * 4); if (c->cap == 0) c->off[0] = 0; }
  else c->data = (uint8_t *)realloc(c->data, (size_t)nc * tw(c->type));
  size_t ob = (size_t)((c->cap + 7) / 8), nbb = (size_t)((nc + 7) / 8);
  c->valid = (uint8_t *)realloc(c->valid, nbb);
  memset(c->valid + ob, 0xff, nbb - ob);
  c->cap = nc;
}
static void kc_set_null(kvcol *c, int64_t r) {
  c->valid[r >> 3] &= (uint8_t)~(1u << (r & 7));
  c->has_nulls = 1;
  if (c->type != ORA_BYTES) memset(c->data + (size_t)r * tw(c->type), 0, (size_t)tw(c->type));
}
static void kc_set_fixed(kvcol *c, int64_t r, int64_t iv, double fv) {
  switch (c->type) {
    case ORA_BOOL: c->data[r] = iv != 0; break;
    case ORA_INT16: ((int16_t *)c->data)[r] = (int16_t)iv; break;
    case ORA_INT32: ((int32_t *)c->data)[r] = (int32_t)iv; break;
    case ORA_INT64: ((int64_t *)c->data)[r] = iv; break;
    default: ((double *)c->data)[r] = fv;
  }
}
static uint8_t *kc_bytes_room(kvcol *c, int64_t len) {
  if (c->buflen + len > c->bufcap) {
    int64_t nc = c->bufcap ? c->bufcap : 4096;
    while (nc < c->buflen + len) nc *= 2;
    c->buf = (uint8_t *)realloc(c->buf, (size_t)nc);
    c->bufcap = nc;
  }
  return c->buf + c->buflen;
}

/* processValueBytes (cfetcher.go:1313-1399) over the TUPLE payload of one KV; returns 0 or -5 */
static int process_value_bytes(const ora_kv_fetch *f, ora_kv_result *R, uint8_t *seen, int64_t row, const uint8_t *vb, int64_t vrem) {
  int32_t last_id = 0;
  int idx = 0, remaining = 0;
  for (int c = 0; c < f->n_val_cols; c++) remaining += !seen[f->val_out[c]]; /* remainingValueColsByIdx */
  while (vrem > 0 && remaining > 0) { /* :1342-1394 */
    uint32_t delta; int typ;
    int doff = dec_value_tag(vb, vrem, &delta, &typ);
    if (doff < 0) return -5;
    int32_t id = last_id + (int32_t)delta;
    last_id = id;
    int vec = -1;
    for (; idx < f->n_val_cols; idx++) {
      if (f->val_col_ids[idx] == id) { vec = idx++; break; }
      if (f->val_col_ids[idx] > id) break;
    }
    if (vec < 0) { /* not requested: skip (:1365-1373) */
      int64_t l = peek_value_len(vb, vrem, doff, typ);
      if (l < 0 || l > vrem) return -5;
      vb += l, vrem -= l;
      continue;
    }
    kvcol *oc = &R->cols[f->val_out[vec]];
    if (!seen[f->val_out[vec]]) remaining--;
    seen[f->val_out[vec]] = 1;
    if (typ == T_Null) { /* DecodeTableValueToCol value_encoding.go:35-38 */
      kc_set_null(oc, row);
      vb += doff, vrem -= doff;
      continue;
    }
    const uint8_t *d = vb + doff;
    int64_t drem = vrem - doff, used = 0;
    switch (f->val_types[vec]) {
      case ORA_BOOL:
        if (typ != T_True && typ != T_False) return -5;
        kc_set_fixed(oc, row, typ == T_True, 0);
        break;
      case ORA_FLOAT64: {
        if (drem < 8) return -5;
        uint64_t u = get_be64(d); double x; memcpy(&x, &u, 8);
        kc_set_fixed(oc, row, 0, x); used = 8;
        break;
      }
      case ORA_BYTES: {
        uint64_t l; int n = ora_dec_nonsorting_uvarint(d, drem, &l);
        if (n < 0 || (int64_t)l > drem - n) return -5;
        uint8_t *dst = kc_bytes_room(oc, (int64_t)l);
        memcpy(dst, d + n, (size_t)l);
        oc->buflen += (int64_t)l; oc->off[row + 1] = (uint32_t)oc->buflen;
        used = n + (int64_t)l;
        break;
      }
      default: {
        int64_t v; int n = ora_dec_stdlib_varint(d, drem, &v);
        if (n < 0) return -5;
        kc_set_fixed(oc, row, v, 0); used = n;
      }
    }
    vb += doff + used, vrem -= doff + used;
  }
  return 0;
}

/* colencoding.UnmarshalColumnValueToCol (key_encoding.go:226-297) through roachpb.Value.GetInt / GetBool / GetFloat /
 * GetBytes (data.go:689-790): `tag` and `data` are the simple value's tag byte and data bytes.  Returns 0 or -5. */
static int unmarshal_single(kvcol *oc, int64_t row, int tag, const uint8_t *data, int64_t dlen) {
  switch (oc->type) {
    case ORA_BOOL: {
      int64_t v;
      if (tag != 1 /* INT */ || ora_dec_stdlib_varint(data, dlen, &v) <= 0 || v > 1 || v < 0) return -5;
      kc_set_fixed(oc, row, v, 0);
      return 0;
    }
    case ORA_FLOAT64: {
      if (tag != 2 /* FLOAT */ || dlen != 8) return -5;
      uint64_t u = get_be64(data); double x; memcpy(&x, &u, 8);
      kc_set_fixed(oc, row, 0, x);
      return 0;
    }
    case ORA_BYTES: {
      if (tag != 3 /* BYTES */) return -5;
      uint8_t *dst = kc_bytes_room(oc, dlen);
      memcpy(dst, data, (size_t)dlen);
      oc->buflen += dlen; oc->off[row + 1] = (uint32_t)oc->buflen;
      return 0;
    }
    default: {
      int64_t v;
      if (tag != 1 /* INT */ || ora_dec_stdlib_varint(data, dlen, &v) <= 0) return -5;
      kc_set_fixed(oc, row, v, 0);
      return 0;
    }
  }
}

/* cFetcher.processValue (cfetcher.go:1136-1275, the primary-index branch :1168-1217) for one KV of family `family_id`.
 * Returns 0, -4 ("single entry value with no default column", :1207-1212) or -5. */
static int process_value(const ora_kv_fetch *f, const ora_kv_families *fam, ora_kv_result *R, uint8_t *seen, int64_t row,
                         int32_t family_id, const uint8_t *val, int64_t vl) {
  if (f->n_out == 0) return 0; /* :1162-1165 */
  /* Value.GetTag / dataBytes with the MVCC extended encoding (data.go:336-338, 436-448, 479-489):
   * [BE32 header length][0x65][header]checksum[4] tag[1] data */
  int64_t simple = 0; /* where the simple value (checksum, tag, data) starts */
  int extended = 0;
  if (vl > 5 && val[4] == 101) {
    extended = 1;
    simple = 5 + (int64_t)(((uint32_t)val[0] << 24) | ((uint32_t)val[1] << 16) | ((uint32_t)val[2] << 8) | val[3]);
  }
  int tag = 0; /* ValueType_UNKNOWN */
  if (!extended) { if (vl > 4) tag = val[4]; }
  else if (vl > simple + 4) tag = val[simple + 4];
  const uint8_t *data = val + simple + 5;
  const int64_t dlen = vl - simple - 5;
  if (tag == ValueType_TUPLE) return process_value_bytes(f, R, seen, row, data, dlen);
  if (family_id == 0) return 0; /* the row sentinel of the pre-family format: no value expected (:1192-1196) */
  int32_t def_col = 0;
  int found = 0;
  for (int i = 0; fam && i < fam->n_families; i++)
    if (fam->family_ids[i] == family_id && fam->default_col_ids[i] != 0) { def_col = fam->default_col_ids[i]; found = 1; break; }
  if (!found) return -4;
  int vec = -1; /* orderedColIdxMap.Get(DefaultColumnID): -1 when the column is not fetched */
  for (int c = 0; c < f->n_val_cols; c++)
    if (f->val_col_ids[c] == def_col) vec = c;
  if (vec < 0) return 0; /* processValueSingle :1285-1289 */
  /* Value.IsPresent (data.go:396-411) */
  if (vl == 0 || (extended && vl <= simple)) return 0;
  if (dlen < 0) return -5;
  const int rc = unmarshal_single(&R->cols[f->val_out[vec]], row, tag, data, dlen);
  if (rc == 0) seen[f->val_out[vec]] = 1;
  return rc;
}

/* Decode batch responses into columns: cFetcher.NextBatch's state machine (cfetcher.go:758-1093) over the concatenated
 * KVs of all batches.  stateDecodeFirstKVOfRow (:806-944): the key columns are decoded, the bytes up to their end are
 * the row prefix, the KV's value is processed; with MaxKeysPerRow == 1 the row is finished.  Otherwise
 * stateFetchNextKVWithUnfinishedRow (:946-1020): a KV whose key does not start with the row prefix belongs to the next
 * row; one that does contributes its family's columns, and the row's last possible family (MaxFamilyID) ends the row.
 * stateFinalizeRow: fillNulls (:1401-1427).
 * Returns 0, or a negative code: -1 truncated batch, -2 bad MVCC key, -3 key decode error, -4 a single-column value
 * of a family without default column, -5 value decode error. */
int ora_kv_decode2(const ora_kv_fetch *f, const ora_kv_families *fam, const int32_t *out_types, const uint8_t *repr,
                   const int64_t *batch_offsets, int64_t nbatches, ora_kv_result **res) {
  ora_kv_result *R = (ora_kv_result *)calloc(1, sizeof(*R));
  R->ncols = f->n_out;
  R->cols = (kvcol *)calloc((size_t)(f->n_out > 0 ? f->n_out : 1), sizeof(kvcol));
  for (int c = 0; c < f->n_out; c++) R->cols[c].type = out_types[c];
  int64_t row = 0;
  int rc = 0;
  uint8_t *seen = (uint8_t *)calloc((size_t)(f->n_out > 0 ? f->n_out : 1), 1);
  const int max_keys_per_row = fam ? fam->n_families : 1;
  const int32_t max_family_id = fam && fam->n_families > 0 ? fam->family_ids[fam->n_families - 1] : 0;
  int in_row = 0;                   /* stateFetchNextKVWithUnfinishedRow */
  const uint8_t *row_prefix = NULL; /* lastRowPrefix, from KeyPrefixLength on */
  int64_t row_prefix_len = 0;
#define FINALIZE_ROW()                                               \
  do {                                                               \
    for (int c_ = 0; c_ < f->n_out; c_++)                            \
      if (!seen[c_]) kc_set_null(&R->cols[c_], row);                 \
    row++;                                                           \
    in_row = 0;                                                      \
  } while (0)
  for (int64_t bi = 0; bi < nbatches && rc == 0; bi++) {
    const uint8_t *p = repr + batch_offsets[bi];
    int64_t rem = batch_offsets[bi + 1] - batch_offsets[bi];
    while (rem > 0) {
      if (rem < 8) { rc = -1; break; } /* "unexpected batch EOF" decode.go:109 */
      uint32_t vl, kl;
      memcpy(&vl, p, 4); memcpy(&kl, p + 4, 4);
      if ((int64_t)8 + kl + vl > rem) { rc = -1; break; }
      const uint8_t *mk = p + 8, *val = p + 8 + kl;
      /* SplitMVCCKey decode.go:26-49 */
      if (kl == 0) { rc = -2; break; }
      int ts_len = mk[kl - 1];
      int64_t key_end = (int64_t)kl - 1 - ts_len;
      if (ts_len == 1 || key_end < 0) { rc = -2; break; }
      const uint8_t *key = mk + f->key_prefix_len;
      int64_t klen = key_end - f->key_prefix_len;
      if (klen < 0) { rc = -3; break; }
      if (in_row && !(klen >= row_prefix_len && memcmp(key, row_prefix, (size_t)row_prefix_len) == 0))
        FINALIZE_ROW(); /* the KV is from a different row (:975-981) */
      int32_t family_id = 0;
      if (!in_row) { /* stateDecodeFirstKVOfRow */
        for (int c = 0; c < f->n_out; c++) { kc_reserve(&R->cols[c], row + 1); seen[c] = 0; if (R->cols[c].type == ORA_BYTES) R->cols[c].off[row + 1] = (uint32_t)R->cols[c].buflen; }
        const uint8_t *key0 = key;
        /* DecodeKeyValsToCols key_encoding.go:43-81 */
        for (int c = 0; c < f->n_key_cols && rc == 0; c++) {
          const int desc = f->key_dirs[c];
          const int oc = f->key_out[c];
          if (klen <= 0) { rc = -3; break; }
          if (key[0] == (desc ? encodedNullDesc : encodedNull)) { /* DecodeIfNull :101 */
            if (oc >= 0) { kc_set_null(&R->cols[oc], row); seen[oc] = 1; }
            key++, klen--;
            continue;
          }
          int64_t n;
          switch (f->key_types[c]) {
            case ORA_FLOAT64: {
              double d;
              n = desc ? ora_dec_float_desc(key, klen, &d) : ora_dec_float_asc(key, klen, &d);
              if (n < 0) { rc = -3; break; }
              if (oc >= 0) kc_set_fixed(&R->cols[oc], row, 0, d);
              break;
            }
            case ORA_BYTES: {
              int64_t dl;
              n = ora_dec_bytes(key, klen, desc, NULL, &dl);
              if (n < 0) { rc = -3; break; }
              if (oc >= 0) {
                uint8_t *dst = kc_bytes_room(&R->cols[oc], dl);
                ora_dec_bytes(key, klen, desc, dst, &dl);
                R->cols[oc].buflen += dl;
                R->cols[oc].off[row + 1] = (uint32_t)R->cols[oc].buflen;
              }
              break;
            }
            default: {
              int64_t v;
              n = desc ? ora_dec_varint_desc(key, klen, &v) : ora_dec_varint_asc(key, klen, &v);
              if (n < 0) { rc = -3; break; }
              if (oc >= 0) kc_set_fixed(&R->cols[oc], row, v, 0);
            }
          }
          if (rc) break;
          if (oc >= 0) seen[oc] = 1;
          key += n, klen -= n;
        }
        if (rc) break;
        row_prefix = key0; /* :857-858 */
        row_prefix_len = key - key0;
        if (max_family_id != 0) { /* getCurrentColumnFamilyID :1447-1467 */
          uint64_t id;
          if (ora_dec_uvarint_asc(key, klen, &id) < 0) { rc = -3; break; }
          family_id = (int32_t)id;
        }
        rc = process_value(f, fam, R, seen, row, family_id, val, vl);
        if (rc) break;
        if (max_keys_per_row == 1) FINALIZE_ROW(); /* :937-941 */
        else in_row = 1;
      } else { /* another KV of the unfinished row (:983-1020) */
        if (max_family_id != 0) {
          uint64_t id;
          if (ora_dec_uvarint_asc(key + row_prefix_len, klen - row_prefix_len, &id) < 0) { rc = -3; break; }
          family_id = (int32_t)id;
        }
        rc = process_value(f, fam, R, seen, row, family_id, val, vl);
        if (rc) break;
        if (family_id == max_family_id) FINALIZE_ROW(); /* :1013-1016 */
      }
      p += 8 + (int64_t)kl + vl;
      rem -= 8 + (int64_t)kl + vl;
    }
  }
  if (rc == 0 && in_row) FINALIZE_ROW(); /* no more KVs: the last row is complete (:953-968) */
#undef FINALIZE_ROW
  free(seen);
  R->nrows = row;
  for (int c = 0; c < f->n_out; c++) R->cols[c].n = row;
  *res = R;
  return rc;
}
int ora_kv_decode(const ora_kv_fetch *f, const int32_t *out_types, const uint8_t *repr, const int64_t *batch_offsets,
                  int64_t nbatches, ora_kv_result **res) {
  return ora_kv_decode2(f, NULL, out_types, repr, batch_offsets, nbatches, res);
}
int32_t ora_kv_result_ncols(const ora_kv_result *r) { return r->ncols; }
int64_t ora_kv_result_nrows(const ora_kv_result *r) { return r->nrows; }
void ora_kv_result_col(const ora_kv_result *r, int32_t i, ora_col *out) {
  const kvcol *c = &r->cols[i];
  out->type = c->type; out->_pad = 0;
  out->data = c->type == ORA_BYTES ? (const void *)c->buf : (const void *)c->data;
  out->valid = c->has_nulls ? c->valid : NULL;
  out->offsets = c->off;
}
void ora_kv_result_free(ora_kv_result *r) {
  if (!r) return;
  for (int i = 0; i < r->ncols; i++) { free(r->cols[i].data); free(r->cols[i].valid); free(r->cols[i].off); free(r->cols[i].buf); }
  free(r->cols); free(r);
}
