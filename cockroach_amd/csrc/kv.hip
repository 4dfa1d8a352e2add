// kv.hip -- KV -> column fetch decode on the device.
//
// Replaces colfetcher.cFetcher.NextBatch for indexes with one KV per row
// (MaxKeysPerRow == 1, pkg/sql/colfetcher/cfetcher.go:758-1093, fast path :937-941):
// MVCCScan batch framing (pkg/storage/enginepb/decode.go:77-128), DecodeKeyValsToCols /
// decodeTableKeyToCol (pkg/sql/colencoding/key_encoding.go:43-218), processValueBytes
// (cfetcher.go:1313-1399), DecodeTableValueToCol (colencoding/value_encoding.go:24-116),
// fillNulls (cfetcher.go:1401-1427) and the primitive codecs of pkg/util/encoding
// (encoding.go:374-403,569-591,788-824,2617-2673,3032-3060,3420-3486; float.go:61-98).
//
// Record boundaries form a dependent chain inside one batch response (each record's
// length is in its own header; batches are <= 10 MiB in production, rowinfra/base.go:53).
// One wavefront following one chain reads ~1 GB/s, so every batch is cut into 32 KiB
// segments that are walked concurrently: a segment's first record is SPECULATED (the first
// offset from which six consecutive headers are plausible: lengths in range, the batch's
// table/index key prefix, an MVCC timestamp-length byte) and then VERIFIED -- the walk of
// segment s must end exactly where segment s+1 started; any mismatch re-walks that segment
// from the proven offset, so the result never depends on the guess.
// Everything after that is row-parallel: one lane per KV, byte-serial parsing of ~170 B
// that neighbouring lanes share cache lines of.  BYTES outputs take two passes
// (lengths -> scan -> copy).  Bound: HBM stream of the input bytes (read three times:
// two walks + decode) plus the decoded columns.
#include <cstdlib>
#include <cstring>

#include "device.h"

namespace crgpu {

static constexpr int kMaxKeyFetch = 8, kMaxValFetch = 32, kMaxOutFetch = 32, kMaxFamilies = 16;
static constexpr int kWalkWindow = 8192;

struct FetchSpec {
  int32_t key_prefix_len, n_key, n_val, n_out;
  int32_t key_types[kMaxKeyFetch], key_dirs[kMaxKeyFetch], key_out[kMaxKeyFetch];
  int32_t val_ids[kMaxValFetch], val_types[kMaxValFetch], val_out[kMaxValFetch];
  int32_t out_types[kMaxOutFetch];
  // column families (fetchpb.IndexFetchSpec.MaxKeysPerRow / MaxFamilyID / FamilyDefaultColumns)
  int32_t max_keys_per_row, max_family_id, n_fam_defaults;
  int32_t fam_ids[kMaxFamilies], fam_default_val[kMaxFamilies];  // family id -> index into val_* of its default column, -1: not fetched
};
struct FetchOut {
  void *data[kMaxOutFetch];        // fixed-width values, or the BYTES payload buffer (pass 1)
  uint8_t *valid[kMaxOutFetch];    // one byte per row
  uint32_t *lens[kMaxOutFetch];    // BYTES: decoded length per row (pass 0)
  const uint32_t *offs[kMaxOutFetch];  // BYTES: offsets (pass 1)
  uint32_t *src[kMaxOutFetch];     // BYTES value columns: payload position inside the record (pass 0), so that
                                   // pass 1 is a plain copy and does not parse the row again (rows of several KVs:
                                   // relative to the row's first record)
  // rows of several KVs (column families): the row of every KV, whether it opens its row, where a row's first record is
  const uint32_t *row_of;
  const uint8_t *is_first;
  const int64_t *row_first;
};

enum : int32_t { KV_OK = 0, KV_TRUNCATED = 1, KV_BAD_MVCC_KEY = 2, KV_BAD_KEY = 3, KV_NO_DEFAULT_COLUMN = 4, KV_BAD_VALUE = 5 };
struct KvError {
  unsigned long long first_row;  // smallest failing row (atomicMin)
  int32_t code;
  int32_t _pad;
};

// ------------------------------------------------------------------ primitive decoders (device)
__device__ __forceinline__ int dec_uvarint_asc(const uint8_t *b, int64_t len, uint64_t *out) {  // encoding.go:569-591
  if (len <= 0) return -1;
  int length = static_cast<int>(b[0]) - 136;
  if (length <= 109) {
    *out = static_cast<uint64_t>(static_cast<int64_t>(length));
    return length < 0 ? -1 : 1;
  }
  length -= 109;
  if (length < 0 || length > 8 || len - 1 < length) return -1;
  uint64_t v = 0;
  for (int i = 0; i < length; i++) v = (v << 8) | b[1 + i];
  *out = v;
  return 1 + length;
}
__device__ __forceinline__ int dec_varint_asc(const uint8_t *b, int64_t len, int64_t *out) {  // encoding.go:374-403
  if (len <= 0) return -1;
  int length = static_cast<int>(b[0]) - 136;
  if (length < 0) {
    length = -length;
    if (len - 1 < length) return -1;
    uint64_t v = 0;
    for (int i = 0; i < length; i++) v = (v << 8) | static_cast<uint8_t>(~b[1 + i]);
    *out = static_cast<int64_t>(~v);
    return 1 + length;
  }
  uint64_t u;
  const int n = dec_uvarint_asc(b, len, &u);
  if (n < 0 || u > 0x7FFFFFFFFFFFFFFFull) return -1;
  *out = static_cast<int64_t>(u);
  return n;
}
__device__ __forceinline__ uint64_t get_be64(const uint8_t *b) {
  uint64_t u = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) u = (u << 8) | b[i];
  return u;
}
__device__ __forceinline__ int dec_float_key(const uint8_t *b, int64_t len, bool desc, double *out) {  // float.go:61-98
  if (len <= 0) return -1;
  int n;
  double r;
  switch (b[0]) {
    case 0x02: case 0x06: r = __longlong_as_double(0x7FF8000000000000ll); n = 1; break;
    case 0x04: r = 0; n = 1; break;
    case 0x03: if (len < 9) return -1; r = __longlong_as_double(static_cast<long long>(~get_be64(b + 1))); n = 9; break;
    case 0x05: if (len < 9) return -1; r = __longlong_as_double(static_cast<long long>(get_be64(b + 1))); n = 9; break;
    default: return -1;
  }
  if (desc && r != 0 && r == r) r = -r;
  *out = r;
  return n;
}
// decodeBytesInternal (encoding.go:788-824) / DecodeBytesDescending (:776-782); out may be null
__device__ __forceinline__ int64_t dec_bytes_key(const uint8_t *b, int64_t len, bool desc, uint8_t *out, uint32_t *outlen) {
  const uint8_t marker = desc ? 0x13 : 0x12, esc = desc ? 0xff : 0x00, term = desc ? 0xfe : 0x01, e00 = desc ? 0x00 : 0xff;
  if (len <= 0 || b[0] != marker) return -1;
  int64_t i = 1;
  uint32_t o = 0;
  for (;;) {
    if (i >= len) return -1;
    const uint8_t c = b[i];
    if (c != esc) {
      if (out) out[o] = desc ? static_cast<uint8_t>(~c) : c;
      o++, i++;
      continue;
    }
    if (i + 1 >= len) return -1;
    const uint8_t v = b[i + 1];
    if (v == term) {
      *outlen = o;
      return i + 2;
    }
    if (v != e00) return -1;
    if (out) out[o] = 0x00;
    o++, i += 2;
  }
}
__device__ __forceinline__ int dec_nonsorting_uvarint(const uint8_t *b, int64_t len, uint64_t *out) {  // :2663-2673
  uint64_t v = 0;
  for (int64_t i = 0; i < len; i++) {
    v += static_cast<uint64_t>(b[i] & 0x7f);
    if (b[i] < 0x80) {
      *out = v;
      return static_cast<int>(i + 1);
    }
    v <<= 7;
  }
  return -1;
}
__device__ __forceinline__ int dec_stdlib_varint(const uint8_t *b, int64_t len, int64_t *out) {  // binary.Varint (:2617)
  uint64_t ux = 0;
  int s = 0;
  for (int64_t i = 0; i < len && i < 10; i++) {
    if (b[i] < 0x80) {
      if (i == 9 && b[i] > 1) return -1;
      ux |= static_cast<uint64_t>(b[i]) << s;
      int64_t x = static_cast<int64_t>(ux >> 1);
      if (ux & 1) x = ~x;
      *out = x;
      return static_cast<int>(i + 1);
    }
    ux |= static_cast<uint64_t>(b[i] & 0x7f) << s;
    s += 7;
  }
  return -1;
}
__device__ __forceinline__ int dec_value_tag(const uint8_t *b, int64_t len, uint32_t *delta, int *typ) {  // :3032-3060
  uint64_t tag;
  int n = dec_nonsorting_uvarint(b, len, &tag);
  if (n < 0) return -1;
  *delta = static_cast<uint32_t>(tag >> 4);
  *typ = static_cast<int>(tag & 0xf);
  if (*typ == 15) {
    uint64_t t2;
    const int n2 = dec_nonsorting_uvarint(b + n, len - n, &t2);
    if (n2 < 0) return -1;
    *typ = static_cast<int>(t2);
    n += n2;
  }
  return n;
}
// PeekValueLengthWithOffsetsAndType (:3420-3486) for the value types a table row can carry
__device__ __forceinline__ int64_t peek_value_len(const uint8_t *b, int64_t len, int data_off, int typ) {
  const uint8_t *d = b + data_off;
  const int64_t rem = len - data_off;
  uint64_t u;
  int64_t s;
  int n;
  switch (typ) {
    case 1: case 10: case 11: return data_off;                       // Null, True, False
    case 3: n = dec_stdlib_varint(d, rem, &s); return n < 0 ? -1 : data_off + n;  // Int
    case 4: return data_off + 8;                                      // Float
    case 6: case 13: n = dec_nonsorting_uvarint(d, rem, &u); return n < 0 ? -1 : data_off + n + static_cast<int64_t>(u);
    case 5: {                                                         // Decimal: binary.Uvarint length + bytes
      uint64_t x = 0;
      int sh = 0, i = 0;
      for (; i < rem && i < 10; i++) {
        x |= static_cast<uint64_t>(d[i] & 0x7f) << sh;
        sh += 7;
        if (d[i] < 0x80) break;
      }
      if (i >= rem || i >= 10) return -1;
      return data_off + i + 1 + static_cast<int64_t>(x);
    }
    case 8: case 9: {                                                 // Time (2 varints), Duration (3)
      int tot = 0;
      for (int k = 0; k < (typ == 8 ? 2 : 3); k++) {
        n = dec_stdlib_varint(d + tot, rem - tot, &s);
        if (n < 0) return -1;
        tot += n;
      }
      return data_off + tot;
    }
    case 12: return data_off + 16;                                    // UUID
    default: return -1;
  }
}

// ------------------------------------------------------------------ record boundaries
static constexpr int64_t kSegBytes = 32 * 1024;
static constexpr int kSpecDepth = 6;
static constexpr int kMaxRel = 2048;  // record positions a segment keeps from its counting walk (more: the writing walk runs)
struct KvSeg {
  int64_t lo, hi;      // the batch response this segment belongs to
  int64_t begin, end;  // records that START in [begin, end) belong to the segment
  int32_t batch;
  int32_t first;       // 1: begin == lo, the first record is known
};

// The segments of a call, cut on the device from the batch boundaries: meta = [n_batches + 1] batch offsets, then
// [n_batches + 1] the number of segments in front of every batch.  (Cutting them on the host and copying 40 bytes per
// segment from pageable memory left the GPU idle for 0.3 ms in front of every 4 M-row call.)
__global__ __launch_bounds__(256) void kv_make_segs_kernel(const int64_t *__restrict__ meta, int32_t n_batches, int64_t nseg, int64_t seg_bytes,
                                                            KvSeg *__restrict__ segs) {
  const int64_t *off = meta, *first = meta + n_batches + 1;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; s < nseg; s += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int32_t lo = 0, hi = n_batches;  // the batch b with first[b] <= s < first[b + 1]
    while (hi - lo > 1) {
      const int32_t mid = (lo + hi) >> 1;
      if (first[mid] <= s) lo = mid;
      else hi = mid;
    }
    const int64_t k = s - first[lo];
    KvSeg sg;
    sg.lo = off[lo];
    sg.hi = off[lo + 1];
    sg.begin = sg.lo + k * seg_bytes;
    sg.end = sg.begin + seg_bytes < sg.hi ? sg.begin + seg_bytes : sg.hi;
    sg.batch = lo;
    sg.first = k == 0;
    segs[s] = sg;
  }
}

__device__ __forceinline__ uint32_t ld_u32_bytes(const uint8_t *p) {
  return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24);
}
// Could a record start at c?  kSpecDepth consecutive headers must stay inside the batch, carry the
// batch's key prefix and end their MVCC key with a timestamp-length byte (decode.go:26-49: 0, 9, 13 or 14).
__device__ inline bool kv_plausible(const uint8_t *__restrict__ repr, int64_t c, int64_t hi, unsigned long long prefix,
                                    int plen) {
  int64_t pos = c;
  for (int d = 0; d < kSpecDepth; d++) {
    if (pos == hi) return true;
    if (pos + 8 > hi) return false;
    const uint32_t vl = ld_u32_bytes(repr + pos), kl = ld_u32_bytes(repr + pos + 4);
    if (kl < static_cast<uint32_t>(plen) + 1u || kl > (1u << 20)) return false;
    const int64_t next = pos + 8 + static_cast<int64_t>(kl) + vl;
    if (next > hi) return false;
    for (int i = 0; i < plen; i++)
      if (repr[pos + 8 + i] != static_cast<uint8_t>(prefix >> (8 * i))) return false;
    const uint8_t last = repr[pos + 8 + kl - 1];
    if (!(last == 0 || last == 9 || last == 13 || last == 14)) return false;
    pos = next;
  }
  return true;
}

// One wavefront per segment.  MODE 0: speculate the first record (unless the segment opens its batch) and
// count; MODE 1: count again from start[] (segments flagged by the verification only); MODE 2: write the
// record offsets.  The segment is pulled through an 8 KiB LDS window with coalesced 16-byte loads; lane 0
// follows the chain of {valLen u32, keyLen u32} headers inside it.
template <int MODE>
__global__ __launch_bounds__(64) void kv_seg_walk_kernel(const uint8_t *__restrict__ repr, const KvSeg *__restrict__ segs,
                                                         int64_t nseg, int plen, int64_t *__restrict__ start,
                                                         int64_t *__restrict__ stop, uint32_t *__restrict__ count,
                                                         int32_t *__restrict__ errc, const uint8_t *__restrict__ flag,
                                                         const uint64_t *__restrict__ row_base, int64_t *__restrict__ rec_off,
                                                         uint16_t *__restrict__ rel, unsigned int *__restrict__ rel_overflow) {
  __shared__ __align__(16) uint8_t win[kWalkWindow + 16];
  __shared__ int64_t s_pos;
  __shared__ int s_stop;
  for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
    if (MODE == 1 && !flag[s]) continue;
    const KvSeg sg = segs[s];
    const int64_t hi = sg.hi;
    int64_t pos;
    if (MODE == 0) {
      pos = sg.first ? sg.lo : -1;
      if (!sg.first && sg.lo + 8 + plen <= hi) {
        unsigned long long prefix = 0;
        for (int i = 0; i < plen; i++) prefix |= static_cast<unsigned long long>(repr[sg.lo + 8 + i]) << (8 * i);
        for (int64_t base = sg.begin; base < sg.end; base += 64) {
          const int64_t c = base + threadIdx.x;
          const bool ok = c < sg.end && c < hi && kv_plausible(repr, c, hi, prefix, plen);
          const uint64_t m = __ballot(ok);
          if (m) {
            pos = base + (__ffsll(static_cast<long long>(m)) - 1);
            break;
          }
        }
      }
      if (threadIdx.x == 0) start[s] = pos;
    } else {
      pos = start[s];
    }
    int64_t nrec = 0;
    int stopc = 0;
    const int64_t out_base = MODE == 2 ? static_cast<int64_t>(row_base[s]) : 0;
    while (pos >= 0 && pos < sg.end && pos < hi) {
      // window starts at the 16-byte aligned address at or below pos
      const int64_t wbase = pos & ~int64_t(15);
      for (int i = threadIdx.x * 16; i < kWalkWindow; i += 64 * 16) {
        const int64_t g = wbase + i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 16 <= hi) {
          v = *reinterpret_cast<const uint4 *>(repr + g);
        } else if (g < hi) {
          uint32_t w[4] = {0, 0, 0, 0};
          for (int k = 0; k < 16; k++)
            if (g + k < hi) w[k >> 2] |= static_cast<uint32_t>(repr[g + k]) << (8 * (k & 3));
          v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4 *>(win + i) = v;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        int stop_code = 0;
        const int64_t wend = wbase + kWalkWindow < hi ? wbase + kWalkWindow : hi;
        while (pos < sg.end && pos < hi) {
          if (pos + 8 > wend) {
            if (pos + 8 > hi) stop_code = KV_TRUNCATED;  // "unexpected batch EOF" (decode.go:109-111)
            break;                                        // otherwise: header not in this window, reload
          }
          const uint8_t *h = win + (pos - wbase);
          const uint32_t vl = ld_u32_bytes(h), kl = ld_u32_bytes(h + 4);
          const int64_t next = pos + 8 + static_cast<int64_t>(kl) + vl;
          if (next > hi) {
            stop_code = KV_TRUNCATED;
            break;
          }
          if (MODE == 2) rec_off[out_base + nrec] = pos;
          else if (nrec < kMaxRel) rel[s * kMaxRel + nrec] = static_cast<uint16_t>(pos - sg.begin);  // (pos in [begin, end))
          nrec++;
          pos = next;
        }
        s_pos = pos;
        s_stop = stop_code;
      }
      __syncthreads();
      pos = s_pos;
      stopc = s_stop;
      __syncthreads();
      if (stopc) break;
    }
    if (MODE != 2 && threadIdx.x == 0) {
      if (nrec > kMaxRel) atomicOr(rel_overflow, 1u);
      count[s] = static_cast<uint32_t>(nrec);
      stop[s] = pos;
      errc[s] = stopc;
    }
  }
}

// the walk of segment s-1 proves where segment s starts; a speculated start that differs is replaced and flagged
__global__ __launch_bounds__(256) void kv_seg_verify_kernel(const KvSeg *__restrict__ segs, int64_t nseg,
                                                            int64_t *__restrict__ start, const int64_t *__restrict__ stop,
                                                            const int32_t *__restrict__ errc, uint8_t *__restrict__ flag,
                                                            unsigned int *__restrict__ nbad) {
  const int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  uint8_t f = 0;
  if (!segs[s].first) {
    // a truncated predecessor ends the batch: nothing after it is decoded (the error is reported)
    const int64_t expected = errc[s - 1] ? segs[s].hi : stop[s - 1];
    if (expected >= 0 && start[s] != expected) {
      start[s] = expected;
      f = 1;
    }
  }
  flag[s] = f;
  if (f) atomicAdd(nbad, 1u);
}
// record offsets from the positions the counting walk kept: every lane writes (the walk itself keeps one lane busy)
__global__ __launch_bounds__(64) void kv_rec_off_kernel(const KvSeg *__restrict__ segs, int64_t nseg, const uint32_t *__restrict__ count,
                                                        const uint64_t *__restrict__ row_base, const uint16_t *__restrict__ rel,
                                                        int64_t *__restrict__ rec_off) {
  for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
    const uint32_t c = count[s];
    const int64_t begin = segs[s].begin, base = static_cast<int64_t>(row_base[s]);
    for (uint32_t r = threadIdx.x; r < c; r += 64) rec_off[base + r] = begin + rel[s * kMaxRel + r];
  }
}
__global__ __launch_bounds__(256) void kv_seg_errors_kernel(const KvSeg *__restrict__ segs, int64_t nseg,
                                                            const int32_t *__restrict__ errc, KvError *__restrict__ err) {
  const int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= nseg || errc[s] == 0) return;
  atomicMin(&err->first_row, static_cast<unsigned long long>(segs[s].batch));
  err->code = errc[s];
}

// ------------------------------------------------------------------ row decode
__device__ __forceinline__ void put_fixed(void *dst, int type, int64_t row, int64_t iv, double fv) {
  switch (type) {
    case CRGPU_BOOL: static_cast<uint8_t *>(dst)[row] = iv != 0; break;
    case CRGPU_INT16: static_cast<int16_t *>(dst)[row] = static_cast<int16_t>(iv); break;
    case CRGPU_INT32: static_cast<int32_t *>(dst)[row] = static_cast<int32_t>(iv); break;
    case CRGPU_INT64: static_cast<int64_t *>(dst)[row] = iv; break;
    default: static_cast<double *>(dst)[row] = fv; break;
  }
}

// The key columns of one KV, decoded (first KV of a row) or only measured: DecodeKeyValsToCols (key_encoding.go:43-81).
// Returns the bytes they take, or -1.  `o` / `row` / `seen` are used when WRITE.
template <int PASS, bool WRITE>
__device__ __forceinline__ int64_t kv_key_cols(const uint8_t *key, int64_t klen, int64_t row, const FetchSpec &f, const FetchOut &o,
                                               uint32_t *seen) {
  const uint8_t *k0 = key;
  for (int c = 0; c < f.n_key; c++) {
    const bool desc = f.key_dirs[c] != 0;
    const int oc = f.key_out[c];
    if (klen <= 0) return -1;
    if (key[0] == (desc ? 0xff : 0x00)) {  // NULL (DecodeIfNull, :101-104)
      if (WRITE && oc >= 0) {
        *seen |= 1u << oc;
        if (PASS == 0) {
          o.valid[oc][row] = 0;
          if (f.out_types[oc] == CRGPU_BYTES) o.lens[oc][row] = 0;
          else put_fixed(o.data[oc], f.out_types[oc], row, 0, 0);
        }
      }
      key++, klen--;
      continue;
    }
    int64_t used;
    if (f.key_types[c] == CRGPU_FLOAT64) {
      double d;
      used = dec_float_key(key, klen, desc, &d);
      if (WRITE && used >= 0 && oc >= 0 && PASS == 0) put_fixed(o.data[oc], CRGPU_FLOAT64, row, 0, d);
    } else if (f.key_types[c] == CRGPU_BYTES) {
      uint32_t dl = 0;
      uint8_t *dst = (WRITE && PASS == 1 && oc >= 0) ? static_cast<uint8_t *>(o.data[oc]) + o.offs[oc][row] : nullptr;
      used = dec_bytes_key(key, klen, desc, dst, &dl);
      if (WRITE && used >= 0 && oc >= 0 && PASS == 0) o.lens[oc][row] = dl;
    } else {
      int64_t v;
      used = dec_varint_asc(key, klen, &v);
      if (desc) v = ~v;  // DecodeVarintDescending (:407-410)
      if (WRITE && used >= 0 && oc >= 0 && PASS == 0) put_fixed(o.data[oc], f.out_types[oc], row, v, 0);
    }
    if (used < 0) return -1;
    if (WRITE && oc >= 0) {
      *seen |= 1u << oc;
      if (PASS == 0) o.valid[oc][row] = 1;
    }
    key += used, klen -= used;
  }
  return key - k0;
}

// PASS 0: fixed-width values, validity bytes, BYTES lengths.  PASS 1: BYTES payloads.
// FAM: rows of several KVs (column families): `kv` is the record, o.row_of / o.is_first say which row it belongs to
// and whether it opens it; validity bytes, lengths and values were zeroed before (a column no KV of the row carries is
// NULL: fillNulls), every KV writes the columns its family holds.
template <int PASS, bool FAM>
__device__ __forceinline__ void kv_parse_row(const uint8_t *p, int64_t kv, int64_t rec_off_kv, const FetchSpec &f, const FetchOut &o,
                                             KvError *__restrict__ err) {
  {
    const int64_t row = FAM ? static_cast<int64_t>(o.row_of[kv]) : kv;
    const bool first = FAM ? o.is_first[kv] != 0 : true;
    const uint32_t rel_base = FAM ? static_cast<uint32_t>(rec_off_kv - o.row_first[row]) : 0u;
    const uint32_t vl = p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24);
    const uint32_t kl = p[4] | (p[5] << 8) | (p[6] << 16) | (static_cast<uint32_t>(p[7]) << 24);
    const uint8_t *mk = p + 8, *val = p + 8 + kl;
    int code = KV_OK;
    uint32_t seen = 0;
    // SplitMVCCKey (decode.go:26-49)
    const int ts_len = kl ? mk[kl - 1] : 1;
    const int64_t key_end = static_cast<int64_t>(kl) - 1 - ts_len;
    if (kl == 0 || ts_len == 1 || key_end < 0) code = KV_BAD_MVCC_KEY;
    const uint8_t *key = mk + f.key_prefix_len;
    int64_t klen = key_end - f.key_prefix_len;
    if (code == KV_OK && klen < 0) code = KV_BAD_KEY;
    int32_t family_id = 0;
    if (code == KV_OK) {
      // the first KV of a row decodes its key columns (stateDecodeFirstKVOfRow, cfetcher.go:830-858); a later KV of the
      // row only carries the row prefix in front of its family id
      const int64_t used = first ? kv_key_cols<PASS, true>(key, klen, row, f, o, &seen) : kv_key_cols<PASS, false>(key, klen, row, f, o, &seen);
      if (used < 0) {
        code = KV_BAD_KEY;
      } else if (f.max_family_id != 0) {  // getCurrentColumnFamilyID (cfetcher.go:1447-1467)
        uint64_t id;
        if (dec_uvarint_asc(key + used, klen - used, &id) < 0) code = KV_BAD_KEY;
        family_id = static_cast<int32_t>(id);
      }
    }
    // processValue (cfetcher.go:1136-1217).  RawBytes = [BE32 header length, 0x65, MVCC value header] checksum[4] tag[1]
    // data: Value.GetTag / dataBytes with the extended encoding (roachpb/data.go:336-338, 436-448, 479-489)
    if (code == KV_OK && f.n_out > 0) {
      int64_t simple = 0;
      const bool extended = vl > 5 && val[4] == 101;
      if (extended) simple = 5 + ((static_cast<int64_t>(val[0]) << 24) | (val[1] << 16) | (val[2] << 8) | val[3]);
      int tag = 0;  // ValueType_UNKNOWN
      if (!extended) {
        if (vl > 4) tag = val[4];
      } else if (static_cast<int64_t>(vl) > simple + 4) {
        tag = val[simple + 4];
      }
      const uint8_t *vb = val + simple + 5;
      int64_t vrem = static_cast<int64_t>(vl) - simple - 5;
      if (tag == 10) {  // TUPLE
        int32_t last_id = 0;
        int idx = 0, remaining = f.n_val;
        while (vrem > 0 && remaining > 0) {  // processValueBytes (:1342-1394)
          uint32_t delta;
          int typ;
          const int doff = dec_value_tag(vb, vrem, &delta, &typ);
          if (doff < 0) {
            code = KV_BAD_VALUE;
            break;
          }
          const int32_t id = last_id + static_cast<int32_t>(delta);
          last_id = id;
          int vec = -1;
          for (; idx < f.n_val; idx++) {
            if (f.val_ids[idx] == id) {
              vec = idx++;
              break;
            }
            if (f.val_ids[idx] > id) break;
          }
          if (vec < 0) {  // not requested: skip by its encoded length (:1365-1373)
            const int64_t l = peek_value_len(vb, vrem, doff, typ);
            if (l < 0 || l > vrem) {
              code = KV_BAD_VALUE;
              break;
            }
            vb += l, vrem -= l;
            continue;
          }
          const int oc = f.val_out[vec];
          const int vt = f.val_types[vec];
          seen |= 1u << oc;
          remaining--;
          if (typ == 1) {  // Null (value_encoding.go:35-38)
            if (PASS == 0) {
              o.valid[oc][row] = 0;
              if (vt == CRGPU_BYTES) o.lens[oc][row] = 0;
              else put_fixed(o.data[oc], vt, row, 0, 0);
            }
            vb += doff, vrem -= doff;
            continue;
          }
          const uint8_t *d = vb + doff;
          const int64_t drem = vrem - doff;
          int64_t used = 0;
          if (vt == CRGPU_BOOL) {
            if (typ != 10 && typ != 11) code = KV_BAD_VALUE;
            else if (PASS == 0) put_fixed(o.data[oc], vt, row, typ == 10, 0);
          } else if (vt == CRGPU_FLOAT64) {
            if (drem < 8) code = KV_BAD_VALUE;
            else {
              if (PASS == 0) put_fixed(o.data[oc], vt, row, 0, __longlong_as_double(static_cast<long long>(get_be64(d))));
              used = 8;
            }
          } else if (vt == CRGPU_BYTES) {
            uint64_t l;
            const int nn = dec_nonsorting_uvarint(d, drem, &l);
            if (nn < 0 || static_cast<int64_t>(l) > drem - nn) code = KV_BAD_VALUE;
            else {
              if (PASS == 0) {
                o.lens[oc][row] = static_cast<uint32_t>(l);
                o.src[oc][row] = rel_base + static_cast<uint32_t>(d + nn - p);
              } else {
                uint8_t *dst = static_cast<uint8_t *>(o.data[oc]) + o.offs[oc][row];
                for (uint64_t t = 0; t < l; t++) dst[t] = d[nn + t];
              }
              used = nn + static_cast<int64_t>(l);
            }
          } else {
            int64_t v;
            const int nn = dec_stdlib_varint(d, drem, &v);
            if (nn < 0) code = KV_BAD_VALUE;
            else {
              if (PASS == 0) put_fixed(o.data[oc], vt, row, v, 0);
              used = nn;
            }
          }
          if (code != KV_OK) break;
          if (PASS == 0) o.valid[oc][row] = 1;
          vb += doff + used, vrem -= doff + used;
        }
      } else if (family_id != 0) {
        // a single-column family in the legacy form (:1191-1213): the value's tag names the type, its data is the datum
        // (processValueSingle :1280-1311, colencoding.UnmarshalColumnValueToCol key_encoding.go:226-297)
        int vec = -2;  // -2: the family has no default column
        for (int i = 0; i < f.n_fam_defaults; i++)
          if (f.fam_ids[i] == family_id) vec = f.fam_default_val[i];
        if (vec == -2) {
          code = KV_NO_DEFAULT_COLUMN;
        } else if (vec >= 0 && vl != 0 && !(extended && static_cast<int64_t>(vl) <= simple)) {  // fetched, and Value.IsPresent (data.go:396-411)
          const int oc = f.val_out[vec];
          const int vt = f.val_types[vec];
          if (vrem < 0) {
            code = KV_BAD_VALUE;
          } else if (vt == CRGPU_FLOAT64) {  // GetFloat (data.go:699-712)
            if (tag != 2 || vrem != 8) code = KV_BAD_VALUE;
            else if (PASS == 0) put_fixed(o.data[oc], vt, row, 0, __longlong_as_double(static_cast<long long>(get_be64(vb))));
          } else if (vt == CRGPU_BYTES) {  // GetBytes (:689-694): the rest of the value
            if (tag != 3) code = KV_BAD_VALUE;
            else if (PASS == 0) {
              o.lens[oc][row] = static_cast<uint32_t>(vrem);
              o.src[oc][row] = rel_base + static_cast<uint32_t>(vb - p);
            } else {
              uint8_t *dst = static_cast<uint8_t *>(o.data[oc]) + o.offs[oc][row];
              for (int64_t t = 0; t < vrem; t++) dst[t] = vb[t];
            }
          } else {  // GetInt / GetBool (:765-790)
            int64_t v;
            if (tag != 1 || dec_stdlib_varint(vb, vrem, &v) <= 0 || (vt == CRGPU_BOOL && (v > 1 || v < 0))) code = KV_BAD_VALUE;
            else if (PASS == 0) put_fixed(o.data[oc], vt, row, v, 0);
          }
          if (code == KV_OK) {
            seen |= 1u << oc;
            if (PASS == 0) o.valid[oc][row] = 1;
          }
        }
      }
      // (family 0 with another tag: the row sentinel of the pre-family format, no value expected, :1192-1196)
    }
    if (PASS == 0) {
      // fillNulls (cfetcher.go:1401-1427); a failed row is zero-filled so that later passes stay in bounds.  FAM: the
      // outputs were zeroed before the pass, only a failing KV clears what it may have half-written.
      if (!FAM || code != KV_OK) {
        for (int c = 0; c < f.n_out; c++) {
          if (code != KV_OK || !(seen & (1u << c))) {
            o.valid[c][row] = 0;
            if (f.out_types[c] == CRGPU_BYTES) o.lens[c][row] = 0;
            else put_fixed(o.data[c], f.out_types[c], row, 0, 0);
          }
        }
      }
      if (code != KV_OK) {
        const unsigned long long old = atomicMin(&err->first_row, static_cast<unsigned long long>(row));
        if (static_cast<unsigned long long>(row) <= old) err->code = code;
      }
    }
  }
}

// Rows of several KVs: does KV i open a row?  The reference decides KV by KV (cfetcher.go:946-1020): a KV belongs to
// the unfinished row when its key starts with the row prefix (the bytes up to the end of the key columns of the row's
// first KV), and the KV of the last family (MaxFamilyID) ends the row.  Row-parallel form: all KVs of a row repeat the
// row's key columns byte for byte, so KV i continues KV i-1's row exactly when its key starts with KV i-1's OWN key
// columns -- unless KV i-1 was itself a continuation carrying the last family.  (Deeper chains of KVs that repeat a
// prefix after the last family cannot be written by EncodePrimaryIndex.)
__device__ __forceinline__ bool kv_key_of(const uint8_t *repr, int64_t off, const FetchSpec &f, const FetchOut &o, const uint8_t **key,
                                          int64_t *klen, int64_t *cols_len, int32_t *family) {
  const uint8_t *p = repr + off;
  const uint32_t kl = p[4] | (p[5] << 8) | (p[6] << 16) | (static_cast<uint32_t>(p[7]) << 24);
  const uint8_t *mk = p + 8;
  const int ts_len = kl ? mk[kl - 1] : 1;
  const int64_t key_end = static_cast<int64_t>(kl) - 1 - ts_len;
  *cols_len = -1;
  *family = -1;
  if (kl == 0 || ts_len == 1 || key_end < f.key_prefix_len) return false;
  *key = mk + f.key_prefix_len;
  *klen = key_end - f.key_prefix_len;
  uint32_t seen = 0;
  const int64_t used = kv_key_cols<0, false>(*key, *klen, 0, f, o, &seen);
  if (used < 0) return false;
  *cols_len = used;
  uint64_t id;
  if (f.max_family_id != 0 && dec_uvarint_asc(*key + used, *klen - used, &id) >= 0) *family = static_cast<int32_t>(id);
  return true;
}
__device__ __forceinline__ bool kv_continues(const uint8_t *key, int64_t klen, const uint8_t *pkey, int64_t pcols) {
  if (pcols < 0 || klen < pcols) return false;
  for (int64_t i = 0; i < pcols; i++)
    if (key[i] != pkey[i]) return false;
  return true;
}
__global__ __launch_bounds__(256) void kv_row_first_kernel(const uint8_t *__restrict__ repr, const int64_t *__restrict__ rec_off, int64_t n,
                                                           FetchSpec f, FetchOut o, uint8_t *__restrict__ is_first, uint32_t *__restrict__ first32) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    bool first = true;
    if (i > 0) {
      const uint8_t *key = nullptr, *k1 = nullptr, *k2 = nullptr;
      int64_t klen = 0, cols = 0, kl1 = 0, c1 = 0, kl2 = 0, c2 = 0;
      int32_t fam = 0, fam1 = 0, fam2 = 0;
      const bool ok0 = kv_key_of(repr, rec_off[i], f, o, &key, &klen, &cols, &fam);
      const bool ok1 = kv_key_of(repr, rec_off[i - 1], f, o, &k1, &kl1, &c1, &fam1);
      bool cont = ok0 && ok1 && kv_continues(key, klen, k1, c1);
      if (cont && fam1 == f.max_family_id && i > 1) {  // did KV i-1 end its row as a continuation with the last family?
        const bool ok2 = kv_key_of(repr, rec_off[i - 2], f, o, &k2, &kl2, &c2, &fam2);
        if (ok2 && kv_continues(k1, kl1, k2, c2)) cont = false;
      }
      first = !cont;
    }
    is_first[i] = first;
    first32[i] = first;
  }
}
// row_of[kv] = (rows opened up to and including kv) - 1; row_first[row] = the offset of its first record
__global__ __launch_bounds__(256) void kv_row_index_kernel(const uint64_t *__restrict__ excl, const uint8_t *__restrict__ is_first,
                                                           const int64_t *__restrict__ rec_off, int64_t n, uint32_t *__restrict__ row_of,
                                                           int64_t *__restrict__ row_first) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t r = static_cast<uint32_t>(excl[i]) + (is_first[i] ? 1u : 0u) - 1u;
    row_of[i] = r;
    if (is_first[i]) row_first[r] = rec_off[i];
  }
}

// One lane per KV.  The records of a 256-row tile are contiguous in the batch (~45 KiB for
// lineitem): the tile is copied into LDS with coalesced 16-byte loads and parsed from there
// (byte-serial parsing straight from global memory ran at 0.35 TB/s: one load instruction per
// byte).  A tile whose span does not fit (huge values, or rows of different buffers) is parsed
// from global memory.
static constexpr int kDecodeStage = 60 * 1024;
template <int PASS, bool FAM>
__global__ __launch_bounds__(256) void kv_decode_kernel(const uint8_t *__restrict__ repr, const int64_t *__restrict__ rec_off,
                                                        int64_t n, FetchSpec f_arg, FetchOut o_arg, KvError *__restrict__ err) {
  __shared__ __align__(16) uint8_t stage[kDecodeStage];
  __shared__ int64_t s_end;
  // The parser indexes the column tables and the output pointers with per-lane indices (which value column a lane is
  // at depends on its row's bytes).  As kernel arguments they live in scalar registers, where a per-lane index costs a
  // loop over the distinct lane values per access (most of the kernel's instructions: 1300 scalar + 1000 vector per 64
  // rows); from LDS it is one ds_read.
  __shared__ FetchSpec s_f;
  __shared__ FetchOut s_o;
  if (threadIdx.x == 0) {
    s_f = f_arg;
    s_o = o_arg;
  }
  __syncthreads();
  const FetchSpec &f = s_f;
  const FetchOut &o = s_o;
  const int64_t ntiles = (n + 255) / 256;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * 256, row = row0 + threadIdx.x;
    const int64_t last = (row0 + 255 < n ? row0 + 255 : n - 1);
    const int64_t span_lo = rec_off[row0] & ~int64_t(15);
    if (row == last) {  // end of the tile's last record, from its header
      const uint8_t *h = repr + rec_off[row];
      const uint32_t vl = h[0] | (h[1] << 8) | (h[2] << 16) | (static_cast<uint32_t>(h[3]) << 24);
      const uint32_t kl = h[4] | (h[5] << 8) | (h[6] << 16) | (static_cast<uint32_t>(h[7]) << 24);
      s_end = rec_off[row] + 8 + static_cast<int64_t>(kl) + vl;
    }
    __syncthreads();
    const int64_t span = s_end - span_lo;
    const bool staged = span > 0 && span <= kDecodeStage;
    if (staged) {
      // 16-byte loads; the tail past s_end stays inside the caller's buffer (16 bytes of slack are required)
      for (int64_t i = threadIdx.x * 16; i < span; i += 256 * 16)
        *reinterpret_cast<uint4 *>(stage + i) = *reinterpret_cast<const uint4 *>(repr + span_lo + i);
    }
    __syncthreads();
    if (row < n) {
      const int64_t off = rec_off[row];
      if (staged && off >= span_lo && off < s_end) kv_parse_row<PASS, FAM>(stage + (off - span_lo), row, off, f, o, err);
      else kv_parse_row<PASS, FAM>(repr + off, row, off, f, o, err);
    }
    __syncthreads();
  }
}

// Pass 1 when no BYTES KEY column is fetched (those need unescaping): copy the value payloads found by pass 0.
// (row_start[row]: the row's record, or the first record of a row of several KVs)
__global__ __launch_bounds__(256) void kv_copy_bytes_kernel(const uint8_t *__restrict__ repr, const int64_t *__restrict__ row_start,
                                                            int64_t n, FetchSpec f, FetchOut o) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < n; row += stride) {
    const uint8_t *rec = repr + row_start[row];
    for (int c = 0; c < f.n_out; c++) {
      if (f.out_types[c] != CRGPU_BYTES) continue;
      const uint32_t a0 = o.offs[c][row], len = o.offs[c][row + 1] - a0;
      if (len == 0) continue;
      const uint8_t *s = rec + o.src[c][row];
      uint8_t *dst = static_cast<uint8_t *>(o.data[c]) + a0;
      for (uint32_t t = 0; t < len; t++) dst[t] = s[t];
    }
  }
}

__global__ __launch_bounds__(256) void kv_narrow_offsets_kernel(const uint64_t *__restrict__ in, int64_t n,
                                                                uint32_t *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = static_cast<uint32_t>(in[i]);
}

static inline int kv_grid(int64_t n, int block) {
  int64_t want = (n + block - 1) / block, cap = static_cast<int64_t>(num_cus()) * 8;
  if (want < 1) want = 1;
  return static_cast<int>(want < cap ? want : cap);
}

}  // namespace crgpu

using namespace crgpu;

struct crgpu_decoder : crgpu::Handle {
  FetchSpec spec;
  std::vector<int64_t> segs_meta_host;  // batch offsets + segments in front of every batch (see kv_make_segs_kernel)
  DevBuf segs_meta;
  DevBuf seg_rel;
  DevBuf segs, seg_start, seg_stop, seg_count, seg_err, seg_flag, seg_base, nbad, rec_off, err, scratch, offs64;
  DevBuf is_first, first32, first_scan, row_of, row_first;  // rows of several KVs (column families)
  std::vector<OutCol> out;
  std::vector<DevBuf> valid_bytes, lens, srcpos, offs64c;
  std::vector<crgpu_dcol> views;
  // Pinned read-back area: copies into pageable memory complete one by one (a round trip each, ~25 us); into pinned
  // memory they queue and one wait fetches them all.  [0, 8) the walk's verdict words, [8, 16) the record total,
  // [16, 32) the walk's error word, [32, 48) the decode's error word, [64, ...) the byte totals of the BYTES columns
  uint8_t *rb = nullptr;
  ~crgpu_decoder() {
    if (rb != nullptr) (void)hipHostFree(rb);
  }
};
static constexpr size_t kKvReadbackBytes = 64 + 8 * static_cast<size_t>(kMaxOutFetch);

// bitmap[c] bit i = bytes[c][i] != 0 for every output column c (blockIdx.y); rows past n read as present, like pack_bitmap
struct KvPackArgs {
  const uint8_t *bytes[kMaxOutFetch];
  uint8_t *bitmap[kMaxOutFetch];
  int64_t n;
  int32_t ncols, _pad;
};
__global__ __launch_bounds__(256) void kv_pack_valid_kernel(KvPackArgs a) {
  // (selected with a uniform loop: indexing the kernel arguments with blockIdx.y directly costs a scalar loop per access)
  const uint8_t *bytes = nullptr;
  uint8_t *bitmap = nullptr;
  for (int c = 0; c < a.ncols; c++)
    if (c == static_cast<int>(blockIdx.y)) bytes = a.bytes[c], bitmap = a.bitmap[c];
  const int64_t nbytes = (a.n + 7) >> 3;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; b < nbytes; b += stride) {
    uint32_t bits = 0;
    if (b * 8 + 8 <= a.n) {
      const unsigned long long w = *reinterpret_cast<const unsigned long long *>(bytes + b * 8);  // eight rows at once
#pragma unroll
      for (int k = 0; k < 8; k++) bits |= (((w >> (8 * k)) & 0xFFull) != 0 ? 1u : 0u) << k;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int64_t i = b * 8 + k;
        bits |= (i < a.n ? (bytes[i] != 0 ? 1u : 0u) : 1u) << k;
      }
    }
    bitmap[b] = static_cast<uint8_t>(bits);
  }
}

extern "C" crgpu_status crgpu_dev_decoder_create(const crgpu_fetch_spec *s, crgpu_decoder **out) {
  CRGPU_TRY(ensure_init());
  if (s == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_decoder_create: null argument");
  if (s->n_key_cols < 0 || s->n_key_cols > kMaxKeyFetch || s->n_val_cols < 0 || s->n_val_cols > kMaxValFetch ||
      s->n_out_cols < 0 || s->n_out_cols > kMaxOutFetch || s->key_prefix_len < 0)
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_decoder_create: at most 8 key columns, 32 value columns, 32 outputs");
  auto d = new crgpu_decoder();
  {
    crgpu_status _os = d->open();
    if (_os != CRGPU_OK) { delete d; return _os; }
  }
  CRGPU_ENTER(d);
  {
    void *rb = nullptr;
    const hipError_t e = hipHostMalloc(&rb, kKvReadbackBytes, hipHostMallocDefault);
    if (e != hipSuccess) {
      delete d;
      return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
    }
    d->rb = static_cast<uint8_t *>(rb);
  }
  FetchSpec &f = d->spec;
  std::memset(&f, 0, sizeof(f));
  f.key_prefix_len = s->key_prefix_len;
  f.n_key = s->n_key_cols;
  f.n_val = s->n_val_cols;
  f.n_out = s->n_out_cols;
  auto bad = [&](const char *msg) {
    delete d;
    return fail(CRGPU_ERR_INVALID, msg);
  };
  for (int c = 0; c < f.n_out; c++) {
    f.out_types[c] = s->out_types[c];
    if (!valid_type(f.out_types[c])) return bad("crgpu_dev_decoder_create: unsupported output column type");
  }
  uint32_t used = 0;
  for (int c = 0; c < f.n_key; c++) {
    f.key_types[c] = s->key_types[c];
    f.key_dirs[c] = s->key_dirs[c];
    f.key_out[c] = s->key_out[c];
    if (!valid_type(f.key_types[c])) return bad("crgpu_dev_decoder_create: unsupported key column type");
    if (f.key_dirs[c] != 0 && f.key_dirs[c] != 1) return bad("crgpu_dev_decoder_create: invalid direction");  // key_encoding.go:97-99
    if (f.key_out[c] >= f.n_out) return bad("crgpu_dev_decoder_create: key output ordinal out of range");
    if (f.key_out[c] >= 0) {
      if (f.out_types[f.key_out[c]] != f.key_types[c] || (used >> f.key_out[c]) & 1) return bad("crgpu_dev_decoder_create: key output type mismatch or duplicate output");
      used |= 1u << f.key_out[c];
    }
  }
  for (int c = 0; c < f.n_val; c++) {
    f.val_ids[c] = s->val_col_ids[c];
    f.val_types[c] = s->val_types[c];
    f.val_out[c] = s->val_out[c];
    if (c > 0 && f.val_ids[c] <= f.val_ids[c - 1]) return bad("crgpu_dev_decoder_create: value column ids must increase");
    if (f.val_out[c] < 0 || f.val_out[c] >= f.n_out || f.out_types[f.val_out[c]] != f.val_types[c] || (used >> f.val_out[c]) & 1)
      return bad("crgpu_dev_decoder_create: value output ordinal out of range, type mismatch or duplicate output");
    used |= 1u << f.val_out[c];
  }
  if (f.n_out < 32 && used != (1u << f.n_out) - 1) return bad("crgpu_dev_decoder_create: an output column has no source");
  // column families
  f.max_keys_per_row = s->max_keys_per_row;
  f.max_family_id = s->max_family_id;
  f.n_fam_defaults = s->n_family_defaults;
  if (f.max_keys_per_row < 0 || f.max_family_id < 0 || f.n_fam_defaults < 0 || f.n_fam_defaults > kMaxFamilies ||
      (f.n_fam_defaults > 0 && (s->family_ids == nullptr || s->family_default_col_ids == nullptr)))
    return bad("crgpu_dev_decoder_create: bad column family description (at most 16 family default columns)");
  for (int i = 0; i < f.n_fam_defaults; i++) {
    f.fam_ids[i] = s->family_ids[i];
    f.fam_default_val[i] = -1;  // the family's default column is not fetched: its values are skipped
    for (int c = 0; c < f.n_val; c++)
      if (f.val_ids[c] == s->family_default_col_ids[i]) f.fam_default_val[i] = c;
  }
  d->out = std::vector<OutCol>(f.n_out);
  d->valid_bytes = std::vector<DevBuf>(f.n_out);
  d->lens = std::vector<DevBuf>(f.n_out);
  d->srcpos = std::vector<DevBuf>(f.n_out);
  d->offs64c = std::vector<DevBuf>(f.n_out);
  d->views.resize(f.n_out);
  *out = d;
  return CRGPU_OK;
}

extern "C" void crgpu_dev_decoder_destroy(crgpu_decoder *d) {
  if (d == nullptr) return;
  if (d->st != nullptr) (void)hipStreamSynchronize(d->st);  // its buffers return to the pool: no kernel may still use them
  delete d;
}

static crgpu_status kv_report_error(const KvError &e, const char *phase) {
  if (e.first_row == ~0ull) return CRGPU_OK;
  static const char *names[] = {"", "truncated batch response", "invalid MVCC key", "key column decode error",
                                "single entry value with no default column", "value column decode error"};
  return fail(CRGPU_ERR_INVALID, std::string("crgpu_dev_decode_kv: ") + names[e.code >= 1 && e.code <= 5 ? e.code : 0] +
                                     " (" + phase + " " + std::to_string(e.first_row) + ")");
}

static crgpu_status crgpu_dev_decode_kv_impl(crgpu_decoder *d, const uint8_t *repr_dev, const int64_t *batch_offsets_host,
                                            int64_t n_batches, crgpu_dev_result *res) {
  if (d == nullptr || res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_decode_kv: null argument");
  CRGPU_ENTER(d);
  CRGPU_CHECK_CANCEL(d);
  if (n_batches < 0 || (n_batches > 0 && (batch_offsets_host == nullptr || repr_dev == nullptr)))
    return fail(CRGPU_ERR_INVALID, "crgpu_dev_decode_kv: null batches");
  if (n_batches > 0x7FFFFFFF) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_decode_kv: too many batches");
  for (int64_t b = 0; b < n_batches; b++)
    if (batch_offsets_host[b + 1] < batch_offsets_host[b]) return fail(CRGPU_ERR_INVALID, "crgpu_dev_decode_kv: batch offsets must not decrease");
  if ((reinterpret_cast<uintptr_t>(repr_dev) & 15u) != 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_decode_kv: the batch buffer must be 16-byte aligned");
  hipStream_t st = stream();
  const FetchSpec &f = d->spec;
  int64_t n = 0;
  CRGPU_TRY(d->err.ensure(sizeof(KvError)));
  CRGPU_HIP(hipMemsetAsync(d->err.p, 0xFF, 8, st));
  CRGPU_HIP(hipMemsetAsync(d->err.as<uint8_t>() + 8, 0, 8, st));
  if (n_batches > 0) {
    // cut the batches into segments: counted here, batch by batch; written by a kernel
    std::vector<int64_t> &meta = d->segs_meta_host;  // stays alive until the copy below has run (no wait for it here)
    meta.resize(static_cast<size_t>(2 * (n_batches + 1)));
    int64_t nseg = 0;
    for (int64_t b = 0; b < n_batches; b++) {
      const int64_t len = batch_offsets_host[b + 1] - batch_offsets_host[b];
      meta[static_cast<size_t>(b)] = batch_offsets_host[b];
      meta[static_cast<size_t>(n_batches + 1 + b)] = nseg;
      nseg += len > 0 ? (len + kSegBytes - 1) / kSegBytes : 1;
    }
    meta[static_cast<size_t>(n_batches)] = batch_offsets_host[n_batches];
    meta[static_cast<size_t>(2 * n_batches + 1)] = nseg;
    const size_t ns = static_cast<size_t>(nseg);
    CRGPU_TRY(d->segs_meta.ensure(meta.size() * 8));
    CRGPU_TRY(d->segs.ensure(ns * sizeof(KvSeg)));
    CRGPU_TRY(d->seg_start.ensure(ns * 8));
    CRGPU_TRY(d->seg_stop.ensure(ns * 8));
    CRGPU_TRY(d->seg_count.ensure(ns * 4));
    CRGPU_TRY(d->seg_err.ensure(ns * 4));
    CRGPU_TRY(d->seg_flag.ensure(ns));
    CRGPU_TRY(d->seg_base.ensure((ns + 1) * 8));
    CRGPU_TRY(d->nbad.ensure(8));  // [0] segments whose guess was wrong (per round), [1] a segment kept more than kMaxRel records
    CRGPU_TRY(d->seg_rel.ensure(ns * kMaxRel * 2));
    CRGPU_HIP(hipMemsetAsync(d->nbad.p, 0, 8, st));
    CRGPU_HIP(hipMemcpyAsync(d->segs_meta.p, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(kv_make_segs_kernel, dim3(kv_grid(nseg, 256)), dim3(256), 0, st, d->segs_meta.as<int64_t>(), static_cast<int32_t>(n_batches),
                       nseg, kSegBytes, d->segs.as<KvSeg>());
    CRGPU_HIP(hipGetLastError());
    const int plen = f.key_prefix_len < 8 ? f.key_prefix_len : 8;
    const unsigned wgrid = static_cast<unsigned>(nseg < static_cast<int64_t>(num_cus()) * 32 ? nseg : num_cus() * 32);
    const unsigned vgrid = static_cast<unsigned>((nseg + 255) / 256);
    {
      ProfScope prof("kv_walk");
      hipLaunchKernelGGL(kv_seg_walk_kernel<0>, dim3(wgrid), dim3(64), 0, st, repr_dev, d->segs.as<KvSeg>(), nseg, plen,
                         d->seg_start.as<int64_t>(), d->seg_stop.as<int64_t>(), d->seg_count.as<uint32_t>(),
                         d->seg_err.as<int32_t>(), static_cast<const uint8_t *>(nullptr), static_cast<const uint64_t *>(nullptr),
                         static_cast<int64_t *>(nullptr), d->seg_rel.as<uint16_t>(), d->nbad.as<unsigned int>() + 1);
      CRGPU_HIP(hipGetLastError());
    }
    // The boundaries are verified, the segments' errors collected and their record counts scanned in one go; the verdict
    // (segments whose guess was wrong), the grand total and the error word come back in ONE round trip.  Only if a guess
    // was wrong (not seen outside crafted inputs) are the affected segments walked again and the round repeated.
    uint64_t total = 0;
    unsigned int rel_overflow = 0;
    for (int64_t iter = 0;; iter++) {
      if (iter > nseg) return fail(CRGPU_ERR_HIP, "crgpu_dev_decode_kv: record boundaries did not converge (internal error)");
      CRGPU_HIP(hipMemsetAsync(d->nbad.p, 0, 4, st));
      hipLaunchKernelGGL(kv_seg_verify_kernel, dim3(vgrid), dim3(256), 0, st, d->segs.as<KvSeg>(), nseg, d->seg_start.as<int64_t>(),
                         d->seg_stop.as<int64_t>(), d->seg_err.as<int32_t>(), d->seg_flag.as<uint8_t>(), d->nbad.as<unsigned int>());
      CRGPU_HIP(hipGetLastError());
      hipLaunchKernelGGL(kv_seg_errors_kernel, dim3(vgrid), dim3(256), 0, st, d->segs.as<KvSeg>(), nseg, d->seg_err.as<int32_t>(),
                         d->err.as<KvError>());
      CRGPU_HIP(hipGetLastError());
      CRGPU_TRY(exclusive_scan_u32(d->seg_count.as<uint32_t>(), nseg, d->seg_base.as<uint64_t>(), d->scratch, nullptr));
      unsigned int nb2[2] = {0, 0};
      KvError e;
      CRGPU_HIP(hipMemcpyAsync(d->rb, d->nbad.p, 8, hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipMemcpyAsync(d->rb + 8, d->seg_base.as<uint64_t>() + nseg, 8, hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipMemcpyAsync(d->rb + 16, d->err.p, sizeof(e), hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipStreamSynchronize(st));
      std::memcpy(nb2, d->rb, 8);
      std::memcpy(&total, d->rb + 8, 8);
      std::memcpy(&e, d->rb + 16, sizeof(e));
      const unsigned int nbad = nb2[0];
      rel_overflow = nb2[1];
      if (nbad == 0) {
        if (e.first_row != ~0ull) return kv_report_error(e, "batch");
        break;
      }
      // errors seen by a walk that started at a wrong boundary mean nothing: clear the word, walk those segments again
      CRGPU_HIP(hipMemsetAsync(d->err.p, 0xFF, 8, st));
      CRGPU_HIP(hipMemsetAsync(d->err.as<uint8_t>() + 8, 0, 8, st));
      ProfScope prof("kv_walk");
      hipLaunchKernelGGL(kv_seg_walk_kernel<1>, dim3(wgrid), dim3(64), 0, st, repr_dev, d->segs.as<KvSeg>(), nseg, plen,
                         d->seg_start.as<int64_t>(), d->seg_stop.as<int64_t>(), d->seg_count.as<uint32_t>(),
                         d->seg_err.as<int32_t>(), d->seg_flag.as<uint8_t>(), static_cast<const uint64_t *>(nullptr),
                         static_cast<int64_t *>(nullptr), d->seg_rel.as<uint16_t>(), d->nbad.as<unsigned int>() + 1);
      CRGPU_HIP(hipGetLastError());
    }
    n = static_cast<int64_t>(total);
    if (n > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_decode_kv: more than 2^31 rows in one call");
    if (n > 0) {
      CRGPU_TRY(d->rec_off.ensure(static_cast<size_t>(n) * 8));
      ProfScope prof("kv_walk");
      if (std::getenv("CRGPU_KV_WALK_TWICE") != nullptr) rel_overflow = 1;  // (tests: the form below)
      if (rel_overflow != 0)  // a segment of more than kMaxRel records (records of < 16 bytes): walk again, writing
        hipLaunchKernelGGL(kv_seg_walk_kernel<2>, dim3(wgrid), dim3(64), 0, st, repr_dev, d->segs.as<KvSeg>(), nseg, plen,
                           d->seg_start.as<int64_t>(), d->seg_stop.as<int64_t>(), d->seg_count.as<uint32_t>(),
                           d->seg_err.as<int32_t>(), static_cast<const uint8_t *>(nullptr), d->seg_base.as<uint64_t>(),
                           d->rec_off.as<int64_t>(), static_cast<uint16_t *>(nullptr), static_cast<unsigned int *>(nullptr));
      else
        hipLaunchKernelGGL(kv_rec_off_kernel, dim3(wgrid), dim3(64), 0, st, d->segs.as<KvSeg>(), nseg, d->seg_count.as<uint32_t>(),
                           d->seg_base.as<uint64_t>(), d->seg_rel.as<uint16_t>(), d->rec_off.as<int64_t>());
      CRGPU_HIP(hipGetLastError());
    }
  }
  // Rows of several KVs: which KV opens a row, the row of every KV, every row's first record.  From here on n counts ROWS
  // and nkv the records.
  const int64_t nkv = n;
  const bool fam_rows = f.max_keys_per_row > 1 && nkv > 0;
  FetchOut o;
  std::memset(&o, 0, sizeof(o));
  if (fam_rows) {
    CRGPU_TRY(d->is_first.ensure(static_cast<size_t>(nkv)));
    CRGPU_TRY(d->first32.ensure(static_cast<size_t>(nkv) * 4));
    CRGPU_TRY(d->first_scan.ensure(static_cast<size_t>(nkv + 1) * 8));
    CRGPU_TRY(d->row_of.ensure(static_cast<size_t>(nkv) * 4));
    CRGPU_TRY(d->row_first.ensure(static_cast<size_t>(nkv) * 8));
    uint64_t nrows = 0;
    {
      ProfScope prof("kv_decode");
      hipLaunchKernelGGL(kv_row_first_kernel, dim3(kv_grid(nkv, 256)), dim3(256), 0, st, repr_dev, d->rec_off.as<int64_t>(), nkv, f, o,
                         d->is_first.as<uint8_t>(), d->first32.as<uint32_t>());
      CRGPU_HIP(hipGetLastError());
      CRGPU_TRY(exclusive_scan_u32(d->first32.as<uint32_t>(), nkv, d->first_scan.as<uint64_t>(), d->scratch, &nrows));
      hipLaunchKernelGGL(kv_row_index_kernel, dim3(kv_grid(nkv, 256)), dim3(256), 0, st, d->first_scan.as<uint64_t>(),
                         d->is_first.as<uint8_t>(), d->rec_off.as<int64_t>(), nkv, d->row_of.as<uint32_t>(), d->row_first.as<int64_t>());
      CRGPU_HIP(hipGetLastError());
    }
    n = static_cast<int64_t>(nrows);
    o.row_of = d->row_of.as<uint32_t>();
    o.is_first = d->is_first.as<uint8_t>();
    o.row_first = d->row_first.as<int64_t>();
  }
  const int64_t cap = n > 0 ? n : 1;
  bool any_bytes = false;
  for (int c = 0; c < f.n_out; c++) {
    OutCol &oc = d->out[c];
    oc.type = f.out_types[c];
    oc.nullable = true;
    CRGPU_TRY(d->valid_bytes[c].ensure(static_cast<size_t>(cap)));
    CRGPU_TRY(oc.valid.ensure(static_cast<size_t>((cap + 7) / 8 + 1)));
    o.valid[c] = d->valid_bytes[c].as<uint8_t>();
    if (oc.type == CRGPU_BYTES) {
      any_bytes = true;
      CRGPU_TRY(d->lens[c].ensure(static_cast<size_t>(cap) * 4));
      CRGPU_TRY(oc.offsets.ensure(static_cast<size_t>(cap + 1) * 4));
      o.lens[c] = d->lens[c].as<uint32_t>();
      CRGPU_TRY(d->srcpos[c].ensure(static_cast<size_t>(cap) * 4));
      o.src[c] = d->srcpos[c].as<uint32_t>();
      if (n == 0) CRGPU_HIP(hipMemsetAsync(oc.offsets.p, 0, 4, st));
      if (fam_rows) CRGPU_HIP(hipMemsetAsync(o.lens[c], 0, static_cast<size_t>(cap) * 4, st));
    } else {
      CRGPU_TRY(oc.data.ensure(static_cast<size_t>(cap) * type_width(oc.type)));
      o.data[c] = oc.data.p;
      if (fam_rows) CRGPU_HIP(hipMemsetAsync(o.data[c], 0, static_cast<size_t>(cap) * type_width(oc.type), st));
    }
    // rows of several KVs: a column that no KV of the row carries is NULL (fillNulls); every KV sets what it holds
    if (fam_rows) CRGPU_HIP(hipMemsetAsync(o.valid[c], 0, static_cast<size_t>(cap), st));
  }
  if (n > 0) {
    {
      ProfScope prof("kv_decode");
      if (fam_rows)
        hipLaunchKernelGGL((kv_decode_kernel<0, true>), dim3(kv_grid(nkv, 256)), dim3(256), 0, st, repr_dev, d->rec_off.as<int64_t>(), nkv, f, o,
                           d->err.as<KvError>());
      else
        hipLaunchKernelGGL((kv_decode_kernel<0, false>), dim3(kv_grid(n, 256)), dim3(256), 0, st, repr_dev, d->rec_off.as<int64_t>(), n, f, o,
                           d->err.as<KvError>());
      CRGPU_HIP(hipGetLastError());
    }
    // BYTES columns: lengths -> offsets; all scans are queued, then the totals and the error word come back together
    std::vector<uint64_t> totals(static_cast<size_t>(f.n_out), 0);
    if (any_bytes) {
      for (int c = 0; c < f.n_out; c++) {
        if (f.out_types[c] != CRGPU_BYTES) continue;
        CRGPU_TRY(d->offs64c[c].ensure(static_cast<size_t>(n + 1) * 8));
        CRGPU_TRY(exclusive_scan_u32(d->lens[c].as<uint32_t>(), n, d->offs64c[c].as<uint64_t>(), d->scratch, nullptr));
        CRGPU_HIP(hipMemcpyAsync(d->rb + 64 + 8 * c, d->offs64c[c].as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
      }
    }
    {
      KvError e;
      CRGPU_HIP(hipMemcpyAsync(d->rb + 32, d->err.p, sizeof(e), hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipStreamSynchronize(st));
      std::memcpy(&e, d->rb + 32, sizeof(e));
      for (int c = 0; c < f.n_out; c++)
        if (f.out_types[c] == CRGPU_BYTES) std::memcpy(&totals[c], d->rb + 64 + 8 * c, 8);
      CRGPU_TRY(kv_report_error(e, "row"));
    }
    if (any_bytes) {
      for (int c = 0; c < f.n_out; c++) {
        if (f.out_types[c] != CRGPU_BYTES) continue;
        const uint64_t total = totals[c];
        if (total > 0xFFFFFFFFull) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_decode_kv: a BYTES column exceeds 4 GiB; decode fewer batches per call");
        hipLaunchKernelGGL(kv_narrow_offsets_kernel, dim3(kv_grid(n + 1, 256)), dim3(256), 0, st, d->offs64c[c].as<uint64_t>(), n + 1,
                           d->out[c].offsets.as<uint32_t>());
        CRGPU_TRY(d->out[c].data.ensure(static_cast<size_t>(total ? total : 1)));
        o.data[c] = d->out[c].data.p;
        o.offs[c] = d->out[c].offsets.as<uint32_t>();
      }
      bool bytes_key = false;
      for (int c = 0; c < f.n_key; c++) bytes_key |= f.key_types[c] == CRGPU_BYTES && f.key_out[c] >= 0;
      ProfScope prof("kv_decode");
      if (bytes_key && fam_rows)
        hipLaunchKernelGGL((kv_decode_kernel<1, true>), dim3(kv_grid(nkv, 256)), dim3(256), 0, st, repr_dev, d->rec_off.as<int64_t>(), nkv, f, o,
                           d->err.as<KvError>());
      else if (bytes_key)
        hipLaunchKernelGGL((kv_decode_kernel<1, false>), dim3(kv_grid(n, 256)), dim3(256), 0, st, repr_dev, d->rec_off.as<int64_t>(), n, f, o,
                           d->err.as<KvError>());
      else
        hipLaunchKernelGGL(kv_copy_bytes_kernel, dim3(kv_grid(n, 256)), dim3(256), 0, st, repr_dev,
                           fam_rows ? d->row_first.as<int64_t>() : d->rec_off.as<int64_t>(), n, f, o);
      CRGPU_HIP(hipGetLastError());
    }
    {  // validity bytes -> bitmaps, all output columns in ONE launch (a launch per column was ~16 per call)
      KvPackArgs pa;
      std::memset(&pa, 0, sizeof(pa));
      pa.ncols = f.n_out;
      pa.n = n;
      for (int c = 0; c < f.n_out; c++) {
        pa.bytes[c] = o.valid[c];
        pa.bitmap[c] = d->out[c].valid.as<uint8_t>();
      }
      const int64_t nbytes = (n + 7) >> 3;
      hipLaunchKernelGGL(kv_pack_valid_kernel, dim3(kv_grid(nbytes, 256), static_cast<unsigned>(f.n_out)), dim3(256), 0, st, pa);
      CRGPU_HIP(hipGetLastError());
    }
  }
  CRGPU_HIP(hipStreamSynchronize(st));
  for (int c = 0; c < f.n_out; c++) d->views[c] = d->out[c].view();
  res->n = n;
  res->ncols = f.n_out;
  res->_pad = 0;
  res->cols = d->views.data();
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_decode_kv(crgpu_decoder *d, const uint8_t *repr_dev, const int64_t *batch_offsets_host,
                                            int64_t n_batches, crgpu_dev_result *res) {
  const crgpu_status s_ = crgpu_dev_decode_kv_impl(d, repr_dev, batch_offsets_host, n_batches, res);
  if (s_ == CRGPU_OK && d != nullptr) note_rows(d, n_batches, res->n);
  return s_;
}
