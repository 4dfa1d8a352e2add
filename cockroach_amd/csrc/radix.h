// radix.h -- radix-partitioned hash join for large build sides (internal).
//
// A probe of an N_b-row table that does not fit the caches costs one random HBM
// sector per row (~26 G accesses/s on MI355X, measured) no matter how the table is
// laid out.  For big joins both inputs are therefore partitioned ONCE by the top
// bits of fmix64(key) into 1024 partitions (one pass, every HBM write a full
// 128-byte line), the build side is refined into sub-partitions of <= 3072 rows that
// each get a 3072-slot open-addressing table built in LDS, and the probe kernel runs
// partition by partition with all workgroups of an XCD on the same partition so that
// partition's tables (<= 3 MiB) stay in that XCD's L2.
#pragma once

#include "device.h"

namespace crgpu {

static constexpr int kRadixP1 = 1024;        // level-1 partitions: hash bits 54..63
static constexpr int kRadixMaxSubBits = 6;   // level-2 sub-partitions per partition: hash bits from 47 up (47..52 for the join; a caller may ask for 7: 47..53)
static constexpr int kRadixSlots = 3072;     // slots of one sub-partition's table (48 KiB, built in LDS)
static constexpr int kRadixSubCap = 2560;    // rows one sub-partition may hold (load factor <= 0.83; ~0.67 typical)
static constexpr unsigned long long kRadixEmpty = 0xFFFFFFFFFFFFFFFFull;
static constexpr int kRadixBuckets = 1024;   // displacement buckets per sub-partition of a perfect-hash table (one byte each)
static constexpr int kRadixMaxFused = 16;

enum RadixSide : int32_t { R_KEY = 0, R_PROBE_PAY = 1, R_BUILD_VAL = 2, R_BUILD_KEY = 3 };

struct RadixCopy {
  uint8_t *dst;
  uint8_t *dst_valid;  // one byte per output row, or null when the column cannot be NULL
  int32_t width;
  int32_t side;
};

struct RadixEmit {
  int32_t ncopy;
  int32_t _pad;
  RadixCopy copy[kRadixMaxFused];
  int32_t *pair_p;  // optional: probe row ids (payload is the row id)
  int32_t *pair_b;  // optional: build row ids (value is the row id), -1 when unmatched
  unsigned long long *out_count;
};

// Direct-addressed tables ("dense" build keys): when the build side's one integer key column spans a range no wider
// than the slots of the sub-partition tables (hi - lo < 2^(10 + sub_bits + kRadixDirectBits)), the key word of a row
// is x = key - lo and NOTHING is hashed: level-1 partition = the top bits of x, sub-partition = the next sub_bits,
// slot = the low kRadixDirectBits bits.  Unique keys cannot collide, so a sub-partition's table is filled by plain
// placement (no cuckoo walk) and a lookup reads exactly ONE slot -- the probe kernel is bound by its divergent table
// reads, 1.6 per row with the 3-way cuckoo tables.  A probe key outside [lo, lo + span) has no key,
// exactly like a NULL or an out-of-range column of a packed multi-column key.
static constexpr int kRadixDirectBits = 11;  // 2048 of a table's 3072 slots are addressed directly
struct RadixDirect {
  uint64_t lo;        // build side's smallest key (as the packed key word)
  uint64_t span;      // key words x = key - lo with x < span exist; 0: hashed tables
  uint32_t sub_bits;  // sub-partition bits the tables were built with
  uint32_t _pad;
};

// One side of the join as {packed key, 64-bit payload} rows.
struct RadixSrc {
  KeyPack kp;          // the key word of a row (device.h)
  const uint8_t *pay;  // payload column, or null => the payload is the row id
  int32_t pay_width;
  int32_t fast;  // int64 key without NULLs + 8-byte payload (or none: the row id), 16-byte aligned: vector loads
  RadixDirect direct;  // span != 0: direct-addressed tables (set by radix_build / radix_probe from the table)
};

struct RadixSpecial {  // the one build key whose packed value equals kRadixEmpty
  unsigned int present;
  unsigned int _pad;
  unsigned long long val;
};

struct RadixTable {
  bool ready = false;
  bool chd = false;       // the tables are perfect-hash tables: disp holds one displacement byte per (sub-partition, bucket)
  DevBuf disp;
  bool dup = false;       // built by radix_build_dup: a slot's value is {first grouped row << 16 | rows of the key}
  DevBuf pay_sorted;      // dup: the build payloads grouped by key, sub-partition by sub-partition
  DevBuf vals, counters;  // dup probe: every probe row's slot value (count pass -> emit pass); {output rows, rows seen}
  int sub_bits = 0;
  uint32_t nsub = 1;  // sub-partitions per partition of the last split: 2^sub_bits, or the any_nsub it was asked for
  int64_t n = 0;
  DevBuf rows_a, rows_b;  // level-1 output; level-2 output (kRadixSubCap rows reserved per sub-partition)
  DevBuf table, special;  // kRadixSlots slots per sub-partition
  DevBuf hist, totals, starts, sub_count, flag;
  DevBuf p_rows, p_hist, p_totals, p_starts;  // probe-side partition buffers
  // probe side partitioned without a histogram pass: fixed-capacity (partition, chunk) cells in p_rows
  DevBuf p_cell_cnt, p_cell_pref;
  uint32_t p_nchunks = 0, p_cell_cap = 0;
  bool cells_failed = false;  // a cell overflowed once (skewed keys): this table's probes use the histogram form
  bool has_dup_keys = false;  // radix_build(check_dup): two build rows share a key (the table keeps one of them)
  RadixDirect direct = {0, 0, 0, 0};  // span != 0: the tables are direct-addressed (radix_build with src.direct set)
};

// Level 1 only: t.rows_a holds partition p's rows in [starts[p], starts[p] + totals[p]) (uint32 arrays of kRadixP1).
crgpu_status radix_partition_level1(RadixTable &t, const RadixSrc &src, int64_t n);
// Levels 1 and 2 only (hash aggregation of high-cardinality inputs reuses them): t.rows_b holds the rows of
// sub-partition q in [q * subcap, q * subcap + sub_count[q]), q < kRadixP1 << t.sub_bits.
// *ok = false when a sub-partition overflowed its region of `subcap` rows.
// About 2^log2_rows_per_sub rows per sub-partition, at most 2^max_sub_bits (<= 7) sub-partitions per partition.
// any_nsub != 0: exactly that many sub-partitions per partition instead (any number up to 128, chosen by a multiply-shift
// of hash bits the partition does not use; t.nsub says how many, q < kRadixP1 * t.nsub) -- for callers that never look a
// sub-partition up by hash afterwards.
crgpu_status radix_partition_split(RadixTable &t, const RadixSrc &src, int64_t n, bool *ok, uint32_t subcap, int log2_rows_per_sub,
                                   int max_sub_bits, uint32_t any_nsub = 0);
// Hash aggregation over a BYTES key column whose values are at most 16 bytes long -- or over two integer key columns --:
// 32-byte rows {k0, k1, value, length} (the key's bytes little-endian in two zero-padded words) through the same two
// partition levels.
struct WideSrc {
  DCol key;                        // BYTES, no NULLs -- or the first of two integer key columns: {k0, k1} = the two
  DCol key2;                       // sign-extended values, length 16 (type 0: absent)
  const unsigned long long *val;   // optional 8-byte value column (no NULLs), else the value is 0
  const unsigned long long *val2;  // non-null: ONE integer key column in `key` and two value columns -- the row is
                                   // {key, 0, val, val2} and its fourth word is no part of the key (wmask = 0 below)
};
__device__ __forceinline__ uint64_t wide_key_hash(uint64_t k0, uint64_t k1, uint64_t len) {
  return fmix64(k0 ^ fmix64(k1 + len * 0x9E3779B97F4A7C15ull));
}
// (wmask: all ones when the fourth word belongs to the key -- the length --, zero when it carries a second value)
crgpu_status radix_partition_split_wide(RadixTable &t, const WideSrc &src, int64_t n, bool *ok, uint32_t subcap,
                                        int log2_rows_per_sub, int max_sub_bits, uint32_t any_nsub = 0);
// Builds the CSR table; *ok = false (and ready = false) when a sub-partition overflows
// kRadixSubCap, i.e. the build keys are not (nearly) unique: the caller falls back.
// check_dup: the table kernels also verify that the build keys are unique and report it in t.has_dup_keys (the join
// paths do not need it: first writer wins and they want one match; the fused group-join does).
// src.direct.span != 0 asks for direct-addressed tables (the caller measured the key range: radix_direct_fits).
crgpu_status radix_build(RadixTable &t, const RadixSrc &src, int64_t n, bool *ok, bool check_dup = false);
// The sub-partition bits direct-addressed tables need for a key range of `span` values, or -1 when the range is too
// wide for them (or so sparse -- fewer than span / 4 build rows -- that hashed tables are the better use of memory).
int radix_direct_sub_bits(uint64_t span, int64_t n_build);
// Probes n rows.  partitioned: radix-partition the probe rows first (large probes);
// otherwise look every row up directly (three dependent random reads per row).
// nulls_pass: additionally emit the probe rows whose key is NULL as unmatched
// (LEFT OUTER / LEFT ANTI).
crgpu_status radix_probe(RadixTable &t, int join_type, const RadixSrc &src, int64_t n, const RadixEmit &emit,
                         bool partitioned);

// Duplicate build keys (INNER / LEFT OUTER with a non-distinct build side): the same two partition levels, then per
// sub-partition the rows are GROUPED by key in LDS; the table holds one slot per distinct key whose value is the
// {first row, row count} of its group in pay_sorted.  *ok = false when a sub-partition overflows (heavily skewed keys):
// the caller takes the chained table.
crgpu_status radix_build_dup(RadixTable &t, const RadixSrc &src, int64_t n, bool *ok);
// Probe in two passes: count (looks every probe row up once, keeps its slot value, sums the output rows) ...
crgpu_status radix_probe_dup_count(RadixTable &t, int join_type, const RadixSrc &src, int64_t n, bool partitioned,
                                   uint64_t *out_rows);
// ... then emit into outputs sized by the count (expands every probe row into its matches, tile by tile).
crgpu_status radix_probe_dup_emit(RadixTable &t, int join_type, const RadixSrc &src, int64_t n, const RadixEmit &emit,
                                  bool partitioned);

}  // namespace crgpu
