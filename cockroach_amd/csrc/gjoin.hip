// gjoin.hip -- hash group-join: aggregate WHILE probing, no join output is materialised.
//
// Replaces colexec.hashGroupJoiner (pkg/sql/colexec/hash_group_joiner.go:20-76), which the reference itself calls "the
// naive implementation ... simply uses the hash joiner and the hash aggregator directly.  TODO: add optimized
// implementation", for the plan shape its spec defines (execinfrapb.HashGroupJoinerSpec, processors_sql.proto:1035-1072):
// "the join's equality columns are exactly the same as the aggregation's grouping columns", here with one INT64
// equality column, unique build keys (the planner's right-side-is-key case) and an INNER join -- or an outer join that
// keeps the BUILD side (CRGPU_RIGHT_OUTER: every build row is a group, also without a probe row; TPC-H Q13's
// customer LEFT OUTER JOIN orders GROUP BY c_custkey with customer as the build side).  Everything else keeps
// the composition crgpu_dev_join_probe -> crgpu_dev_agg_consume, which stays device-resident.
//
// Why not atomics from the join's probe kernel: random HBM atomics run at ~24 G/s on this part (experiments/atomics.hip)
// -- two accumulator words per probe row would cost 45 ms for config J's probe side, three times the whole join.  The
// groups must be in LDS while their rows stream by.  Every group IS a build row, and the radix join already leaves the
// build rows in sub-partitions of <= 2560 rows, each with a 48 KiB cuckoo table image (radix.hip).  So:
//   build   radix_build, unchanged: 1024 partitions x <= 64 sub-partitions, one table image each
//   probe   the probe rows go through the SAME two partition levels (radix_partition_split with the build side's
//           sub-partition bits; fixed-capacity regions, an overflowing one -- skewed keys -- reports
//           CRGPU_ERR_UNSUPPORTED and the caller composes)
//   fuse    one workgroup per sub-partition: its table image is copied into LDS (coalesced), accumulator planes
//           [slot] sit beside it, the sub-partition's probe rows stream through: <= 3 LDS slot reads per row instead of
//           L2 requests, LDS atomics on the slot's accumulators; at the end the slots that saw a row are emitted
//           (key, build payload, aggregates) with one global atomicAdd per workgroup.
// Traffic: the probe rows cross HBM three times (scatter in/out, split in/out, fuse in) and nothing else is written but
// one row per group; the composition writes the joined rows, reads them, partitions them again for the aggregation.
//
// Roofline: HBM streaming (the three probe-side passes).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>

#include "device.h"
#include "radix.h"

namespace crgpu {

static constexpr int kGjMaxAggs = 8;
static constexpr int kGjMaxPlanes = 4;            // 48 KiB table + 4 x 24 KiB planes fit the 160 KiB of LDS
static constexpr int kGjSlots = kRadixSlots + 1;  // + the build key equal to the empty marker (RadixSpecial)
enum { GJ_IN_KEY = 0, GJ_IN_PROBE = 1, GJ_IN_BUILD = 2 };
enum { GJ_COUNT_ROWS = 0, GJ_SUM_F64, GJ_AVG_F64, GJ_SUM_INT, GJ_MIN_I64, GJ_MAX_I64, GJ_MIN_F64, GJ_MAX_F64, GJ_ANY, GJ_COUNT_PROBE };

struct GjOp {
  int8_t kind, input, plane, plane2;  // plane2: SUM_INT's high word
  int8_t owner, _p[3];                // owner: this aggregation updates `plane` (SUM and AVG of the column share one)
  uint8_t *out;        // [group]
  uint8_t *out_valid;  // OUTER, aggregations over the probe side: one byte per group, 0 = NULL (no probe row); else nullptr
};
struct GjArgs {
  const ulonglong2 *prows;
  const uint32_t *pcount;
  uint32_t psubcap;
  uint32_t nplanes;
  const ulonglong2 *table;
  const RadixSpecial *special;
  int32_t naggs;
  int32_t _pad;
  GjOp ops[kGjMaxAggs];
  unsigned long long *out_count;
  unsigned int *err;
};

// total order for float MIN / MAX: NaN smallest (hash_min_max_agg.eg.go:966-981), as in agg.hip
__device__ __forceinline__ unsigned long long gj_f64_ordered(double f) {
  if (f != f) return 0ull;
  const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(f));
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double gj_ordered_f64(unsigned long long m) {
  if (m == 0ull) return __longlong_as_double(0x7FF8000000000000ll);
  const unsigned long long b = (m >> 63) ? (m & 0x7FFFFFFFFFFFFFFFull) : ~m;
  return __longlong_as_double(static_cast<long long>(b));
}
__device__ __forceinline__ void gj_slots(uint64_t h, uint32_t *s) {  // = radix.hip rslots
  const uint64_t g = fmix64(h + 0x9E3779B97F4A7C15ULL);
  s[0] = static_cast<uint32_t>((static_cast<uint64_t>(static_cast<uint32_t>(h)) * kRadixSlots) >> 32);
  s[1] = static_cast<uint32_t>((static_cast<uint64_t>(static_cast<uint32_t>(g)) * kRadixSlots) >> 32);
  s[2] = static_cast<uint32_t>((static_cast<uint64_t>(static_cast<uint32_t>(g >> 32)) * kRadixSlots) >> 32);
}

// OUTER: the join keeps the build side -- every build row is emitted; without a probe row its COUNT_ROWS is 1 (the one
// joined row, NULL on the probe side), COUNT over the probe side 0 and the other probe-side aggregations NULL.
template <bool OUTER>
__global__ __launch_bounds__(1024) void gjoin_fused_kernel(GjArgs a) {
  constexpr int THREADS = 1024, RPT = 4;
  extern __shared__ __align__(16) unsigned char gj_lds[];
  ulonglong2 *tab = reinterpret_cast<ulonglong2 *>(gj_lds);                                  // [kRadixSlots]
  unsigned long long *planes = reinterpret_cast<unsigned long long *>(tab + kRadixSlots);    // [nplanes][kGjSlots]
  __shared__ uint32_t s_emit, s_base_lo, s_base_hi;
  const uint32_t q = blockIdx.x;
  const uint32_t n = a.pcount[q];
  if (!OUTER && n == 0) return;  // no probe row reached this sub-partition: none of its groups exists (INNER)
  // OUTER: the build row whose key equals the empty marker lives in a side slot; the sub-partition its key hashes to emits it
  const uint32_t nsub = gridDim.x / static_cast<uint32_t>(kRadixP1);
  const uint64_t he = fmix64(kRadixEmpty);
  const bool special_here = OUTER && a.special->present != 0 &&
                            q == static_cast<uint32_t>(he >> 54) * nsub + (static_cast<uint32_t>(he >> 47) & (nsub - 1));
  const ulonglong2 *img = a.table + static_cast<int64_t>(q) * kRadixSlots;
  for (int i = threadIdx.x; i < kRadixSlots; i += THREADS) tab[i] = img[i];
  // plane 0 = rows per slot; the others start at what their aggregation needs
  for (uint32_t p = 0; p < a.nplanes; p++) {
    unsigned long long init = 0;
    for (int f = 0; f < a.naggs; f++) {
      if (a.ops[f].plane != static_cast<int8_t>(p)) continue;
      if (a.ops[f].kind == GJ_MIN_I64) init = 0x7FFFFFFFFFFFFFFFull;
      else if (a.ops[f].kind == GJ_MAX_I64) init = 0x8000000000000000ull;
      else if (a.ops[f].kind == GJ_MIN_F64) init = ~0ull;
    }
    for (int i = threadIdx.x; i < kGjSlots; i += THREADS) planes[p * kGjSlots + i] = init;
  }
  if (threadIdx.x == 0) s_emit = 0;
  __syncthreads();
  const ulonglong2 *rows = a.prows + static_cast<int64_t>(q) * a.psubcap;
  for (uint32_t t0 = 0; t0 < n; t0 += THREADS * RPT) {
    unsigned long long k[RPT], pay[RPT];
    uint32_t slot[RPT], s1[RPT], s2[RPT], cur[RPT];
    bool look[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const uint32_t i = t0 + j * THREADS + threadIdx.x;
      look[j] = i < n;
      k[j] = 0, pay[j] = 0, slot[j] = 0xFFFFFFFFu, s1[j] = s2[j] = cur[j] = 0;
      if (look[j]) {
        const ulonglong2 r = rows[i];
        k[j] = r.x, pay[j] = r.y;
        if (k[j] == kRadixEmpty) {
          if (a.special->present) slot[j] = kRadixSlots;
          look[j] = false;
        } else {
          uint32_t s[3];
          gj_slots(fmix64(k[j]), s);
          cur[j] = s[0], s1[j] = s[1], s2[j] = s[2];
        }
      }
    }
#pragma unroll
    for (int c = 1; c <= 3; c++) {
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        if (!look[j]) continue;
        const unsigned long long tk = tab[cur[j]].x;
        if (tk == k[j]) {
          slot[j] = cur[j];
          look[j] = false;
        } else if (tk == kRadixEmpty || c == 3) {
          look[j] = false;
        } else {
          cur[j] = c == 1 ? s1[j] : s2[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      if (slot[j] == 0xFFFFFFFFu) continue;
      atomicAdd(&planes[slot[j]], 1ull);
      for (int f = 0; f < a.naggs; f++) {
        const GjOp &op = a.ops[f];
        if (op.plane <= 0 || !op.owner) continue;
        unsigned long long *w = &planes[op.plane * kGjSlots + slot[j]];
        switch (op.kind) {
          case GJ_SUM_F64:
          case GJ_AVG_F64: unsafeAtomicAdd(reinterpret_cast<double *>(w), __longlong_as_double(static_cast<long long>(pay[j]))); break;
          case GJ_SUM_INT: {
            const unsigned long long old = atomicAdd(w, pay[j]);
            const unsigned long long hi = ((pay[j] >> 63) ? ~0ull : 0ull) + ((old + pay[j]) < old ? 1ull : 0ull);
            if (hi) atomicAdd(&planes[op.plane2 * kGjSlots + slot[j]], hi);
            break;
          }
          case GJ_MIN_I64: atomicMin(reinterpret_cast<long long *>(w), static_cast<long long>(pay[j])); break;
          case GJ_MAX_I64: atomicMax(reinterpret_cast<long long *>(w), static_cast<long long>(pay[j])); break;
          case GJ_MIN_F64: atomicMin(w, gj_f64_ordered(__longlong_as_double(static_cast<long long>(pay[j])))); break;
          case GJ_MAX_F64: atomicMax(w, gj_f64_ordered(__longlong_as_double(static_cast<long long>(pay[j])))); break;
          default: break;
        }
      }
    }
  }
  __syncthreads();
  // emit the groups that saw a row: positions inside the workgroup first, one global reservation
  uint32_t mine[(kGjSlots + THREADS - 1) / THREADS], nmine = 0, local[(kGjSlots + THREADS - 1) / THREADS];
  for (int i = threadIdx.x, t = 0; i < kGjSlots; i += THREADS, t++) {
    mine[t] = 0xFFFFFFFFu;
    const bool emit = OUTER ? (i < kRadixSlots ? tab[i].x != kRadixEmpty : special_here) : planes[i] != 0ull;
    if (emit) {
      mine[t] = static_cast<uint32_t>(i);
      local[t] = atomicAdd(&s_emit, 1u);
    }
    nmine = t + 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long b = s_emit ? atomicAdd(a.out_count, static_cast<unsigned long long>(s_emit)) : 0ull;
    s_base_lo = static_cast<uint32_t>(b);
    s_base_hi = static_cast<uint32_t>(b >> 32);
  }
  __syncthreads();
  const unsigned long long base = (static_cast<unsigned long long>(s_base_hi) << 32) | s_base_lo;
  for (uint32_t t = 0; t < nmine; t++) {
    if (mine[t] == 0xFFFFFFFFu) continue;
    const uint32_t s = mine[t];
    const unsigned long long pos = base + local[t];
    const unsigned long long key = s == static_cast<uint32_t>(kRadixSlots) ? kRadixEmpty : tab[s].x;
    const unsigned long long bval = s == static_cast<uint32_t>(kRadixSlots) ? a.special->val : tab[s].y;
    const unsigned long long cnt = planes[s];
    for (int f = 0; f < a.naggs; f++) {
      const GjOp &op = a.ops[f];
      unsigned long long v = 0;
      const unsigned long long w = op.plane > 0 ? planes[op.plane * kGjSlots + s] : 0ull;
      switch (op.kind) {
        case GJ_COUNT_ROWS: v = OUTER && cnt == 0 ? 1ull : cnt; break;
        case GJ_COUNT_PROBE: v = cnt; break;
        case GJ_ANY: v = op.input == GJ_IN_KEY ? key : bval; break;
        case GJ_SUM_F64: v = w; break;
        case GJ_AVG_F64:
          v = cnt == 0 ? 0ull : static_cast<unsigned long long>(__double_as_longlong(__longlong_as_double(static_cast<long long>(w)) / static_cast<double>(cnt)));
          break;
        case GJ_SUM_INT: {
          const unsigned long long hi = planes[op.plane2 * kGjSlots + s];
          if (hi != ((w >> 63) ? ~0ull : 0ull)) *a.err = 1;  // the exact total leaves int64 (tree.ErrIntOutOfRange)
          v = w;
          break;
        }
        case GJ_MIN_F64:
        case GJ_MAX_F64: v = static_cast<unsigned long long>(__double_as_longlong(gj_ordered_f64(w))); break;
        default: v = w; break;  // MIN / MAX over INT64
      }
      if (OUTER && op.out_valid != nullptr) {
        op.out_valid[pos] = cnt != 0;
        if (cnt == 0) v = 0;
      }
      reinterpret_cast<unsigned long long *>(op.out)[pos] = v;
    }
  }
}

// validity bytes (one per group, written in emission order) -> bitmap; one thread per output byte
__global__ __launch_bounds__(256) void gjoin_pack_valid_kernel(const uint8_t *__restrict__ bytes, int64_t n, uint8_t *__restrict__ bitmap) {
  const int64_t nb = (n + 7) >> 3;
  for (int64_t b = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; b < nb; b += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    uint32_t m = 0;
    for (int j = 0; j < 8; j++)
      if (b * 8 + j < n && bytes[b * 8 + j] != 0) m |= 1u << j;
    bitmap[b] = static_cast<uint8_t>(m);
  }
}

}  // namespace crgpu

using namespace crgpu;

struct crgpu_gjoin : crgpu::Handle {
  int probe_type = 0, build_type = 0;  // payload column types (0: none)
  int naggs = 0, nplanes = 1;
  bool outer = false;             // CRGPU_RIGHT_OUTER: the build side is kept
  bool nullable[kGjMaxAggs] = {};  // outer: aggregations over the probe side (NULL for a build row without a probe row)
  DevBuf valid_bytes[kGjMaxAggs];
  GjOp ops[kGjMaxAggs];
  int out_type[kGjMaxAggs];
  RadixTable tb, tp;
  int64_t n_build = 0;
  bool built = false;
  DevBuf counter, err;
  std::vector<OutCol> out;
  std::vector<crgpu_dcol> views;
};

extern "C" crgpu_status crgpu_dev_gjoin_create(const crgpu_gjoin_spec *spec, crgpu_gjoin **out) {
  CRGPU_TRY(ensure_init());
  if (spec == nullptr || out == nullptr || (spec->n_aggs > 0 && spec->aggs == nullptr)) return fail(CRGPU_ERR_INVALID, "crgpu_dev_gjoin_create: null argument");
  const char *compose = "crgpu_dev_gjoin_create: outside the fused form (compose crgpu_dev_join_probe -> crgpu_dev_agg_consume)";
  if ((spec->join_type != CRGPU_INNER && spec->join_type != CRGPU_RIGHT_OUTER) || spec->key_type != CRGPU_INT64) return fail(CRGPU_ERR_UNSUPPORTED, compose);
  if (spec->n_aggs < 1 || spec->n_aggs > kGjMaxAggs) return fail(CRGPU_ERR_UNSUPPORTED, compose);
  for (int t : {spec->probe_payload_type, spec->build_payload_type})
    if (t != 0 && t != CRGPU_INT64 && t != CRGPU_FLOAT64) return fail(CRGPU_ERR_UNSUPPORTED, compose);
  std::unique_ptr<crgpu_gjoin> g(new crgpu_gjoin());
  CRGPU_TRY(g->open());
  g->probe_type = spec->probe_payload_type;
  g->build_type = spec->build_payload_type;
  g->naggs = spec->n_aggs;
  g->outer = spec->join_type == CRGPU_RIGHT_OUTER;
  for (int f = 0; f < spec->n_aggs; f++) {
    const crgpu_gjoin_agg &d = spec->aggs[f];
    GjOp &op = g->ops[f];
    std::memset(&op, 0, sizeof(op));
    op.input = static_cast<int8_t>(d.input);
    op.plane = 0;
    op.plane2 = -1;
    const int in_t = d.input == GJ_IN_KEY ? CRGPU_INT64 : d.input == GJ_IN_PROBE ? g->probe_type : d.input == GJ_IN_BUILD ? g->build_type : -1;
    if (d.func != CRGPU_COUNT_ROWS && (in_t <= 0)) return fail(CRGPU_ERR_INVALID, "crgpu_dev_gjoin_create: aggregation over a column the spec does not have");
    bool needs_plane = false;
    switch (d.func) {
      case CRGPU_COUNT_ROWS: op.kind = GJ_COUNT_ROWS; g->out_type[f] = CRGPU_INT64; break;
      case CRGPU_COUNT:  // no NULL inputs: COUNT(col) = the rows -- but the probe side's columns of an unmatched build row are NULL
        op.kind = g->outer && d.input == GJ_IN_PROBE ? GJ_COUNT_PROBE : GJ_COUNT_ROWS;
        g->out_type[f] = CRGPU_INT64;
        break;
      case CRGPU_ANY_NOT_NULL:
        if (d.input == GJ_IN_PROBE) return fail(CRGPU_ERR_UNSUPPORTED, compose);  // "first in arrival order" has no meaning here
        op.kind = GJ_ANY; g->out_type[f] = in_t; break;
      case CRGPU_SUM:
      case CRGPU_AVG:
        if (d.input != GJ_IN_PROBE || in_t != CRGPU_FLOAT64) return fail(CRGPU_ERR_UNSUPPORTED, compose);
        op.kind = d.func == CRGPU_SUM ? GJ_SUM_F64 : GJ_AVG_F64; g->out_type[f] = CRGPU_FLOAT64; needs_plane = true; break;
      case CRGPU_SUM_INT:
        if (d.input != GJ_IN_PROBE || in_t != CRGPU_INT64) return fail(CRGPU_ERR_UNSUPPORTED, compose);
        op.kind = GJ_SUM_INT; g->out_type[f] = CRGPU_INT64; needs_plane = true; break;
      case CRGPU_MIN:
      case CRGPU_MAX:
        if (d.input != GJ_IN_PROBE) return fail(CRGPU_ERR_UNSUPPORTED, compose);
        if (in_t == CRGPU_FLOAT64) op.kind = d.func == CRGPU_MIN ? GJ_MIN_F64 : GJ_MAX_F64;
        else op.kind = d.func == CRGPU_MIN ? GJ_MIN_I64 : GJ_MAX_I64;
        g->out_type[f] = in_t; needs_plane = true; break;
      default: return fail(CRGPU_ERR_UNSUPPORTED, compose);
    }
    g->nullable[f] = g->outer && needs_plane;  // (every aggregation with a plane is one over the probe side)
    if (needs_plane) {
      // SUM and AVG of the same column share a plane
      int share = -1;
      for (int e = 0; e < f && share < 0; e++)
        if ((g->ops[e].kind == GJ_SUM_F64 || g->ops[e].kind == GJ_AVG_F64) && (op.kind == GJ_SUM_F64 || op.kind == GJ_AVG_F64)) share = g->ops[e].plane;
      if (share >= 0) {
        op.plane = static_cast<int8_t>(share);
      } else {
        op.owner = 1;
        op.plane = static_cast<int8_t>(g->nplanes++);
        if (op.kind == GJ_SUM_INT) op.plane2 = static_cast<int8_t>(g->nplanes++);
      }
      if (g->nplanes > kGjMaxPlanes) return fail(CRGPU_ERR_UNSUPPORTED, compose);
    }
  }
  g->out = std::vector<OutCol>(spec->n_aggs);
  g->views.resize(spec->n_aggs);
  CRGPU_TRY(g->counter.ensure(8));
  CRGPU_TRY(g->err.ensure(4));
  static DeviceOnce attr_set;
  if (attr_set.need()) {
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gjoin_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kRadixSlots * 16 + kGjMaxPlanes * kGjSlots * 8));
    CRGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gjoin_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  kRadixSlots * 16 + kGjMaxPlanes * kGjSlots * 8));
    attr_set.mark();
  }
  *out = g.release();
  return CRGPU_OK;
}

extern "C" void crgpu_dev_gjoin_destroy(crgpu_gjoin *g) {
  if (g == nullptr) return;
  if (g->st != nullptr) (void)hipStreamSynchronize(g->st);
  delete g;
}

static crgpu_status gj_src(const crgpu_dcol *key, const crgpu_dcol *pay, int pay_type, int64_t n, RadixSrc *out) {
  if (key == nullptr || key->type != CRGPU_INT64 || (n > 0 && key->data == nullptr)) return fail(CRGPU_ERR_INVALID, "crgpu_dev_gjoin: the key column must be INT64");
  if (pay_type != 0 && (pay == nullptr || pay->type != pay_type || (n > 0 && pay->data == nullptr)))
    return fail(CRGPU_ERR_INVALID, "crgpu_dev_gjoin: payload column differs from the spec");
  if (pay_type != 0 && pay->valid != nullptr) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_gjoin: payload columns with NULLs take the composed form");
  RadixSrc s;
  std::memset(&s, 0, sizeof(s));
  s.kp.n = 1;
  s.kp.c[0] = to_dcol(*key);
  s.pay = pay_type != 0 ? static_cast<const uint8_t *>(pay->data) : nullptr;
  s.pay_width = pay_type != 0 ? 8 : 0;
  s.fast = key->valid == nullptr && s.pay != nullptr && (reinterpret_cast<uintptr_t>(key->data) & 15u) == 0 &&
           (reinterpret_cast<uintptr_t>(s.pay) & 15u) == 0;
  *out = s;
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_gjoin_build(crgpu_gjoin *g, const crgpu_dcol *key, const crgpu_dcol *payload, int64_t n) {
  if (g == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_gjoin_build: null handle");
  CRGPU_ENTER(g);
  CRGPU_CHECK_CANCEL(g);
  if (n < 1 || n > (int64_t(1) << 30)) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_gjoin_build: between 1 and 2^30 build rows");
  RadixSrc src;
  CRGPU_TRY(gj_src(key, payload, g->build_type, n, &src));
  // (a build row with a NULL key joins nothing, but an outer join that keeps the build side still owes it a group)
  if (g->outer && key->valid != nullptr) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_gjoin_build: RIGHT OUTER with a nullable build key (compose join -> aggregate)");
  bool ok = false;
  g->built = false;
  CRGPU_TRY(radix_build(g->tb, src, n, &ok, /*check_dup=*/true));
  if (!ok) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_gjoin_build: build keys are not unique enough for the partitioned tables (compose join -> aggregate)");
  // every group IS a build row: a key held by two build rows would be counted once where the join emits it twice
  if (g->tb.has_dup_keys) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_gjoin_build: duplicate build keys (compose join -> aggregate)");
  g->n_build = n;
  g->built = true;
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_gjoin_probe(crgpu_gjoin *g, const crgpu_dcol *key, const crgpu_dcol *payload, int64_t n,
                                              crgpu_dev_result *res) {
  if (g == nullptr || res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_gjoin_probe: null argument");
  CRGPU_ENTER(g);
  CRGPU_CHECK_CANCEL(g);
  if (!g->built) return fail(CRGPU_ERR_STATE, "crgpu_dev_gjoin_probe: build first");
  if (n < 0 || n > (int64_t(1) << 30)) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_gjoin_probe: at most 2^30 probe rows per call");
  hipStream_t st = stream();
  const int64_t cap = g->n_build;
  for (int f = 0; f < g->naggs; f++) {
    OutCol &o = g->out[f];
    o.type = g->out_type[f];
    o.nullable = g->nullable[f];
    CRGPU_TRY(o.data.ensure(static_cast<size_t>(cap) * 8));
    g->ops[f].out = o.data.as<uint8_t>();
    g->ops[f].out_valid = nullptr;
    if (g->nullable[f]) {
      CRGPU_TRY(g->valid_bytes[f].ensure(static_cast<size_t>(cap)));
      CRGPU_TRY(o.valid.ensure(static_cast<size_t>((cap + 7) / 8 + 1)));
      g->ops[f].out_valid = g->valid_bytes[f].as<uint8_t>();
    }
  }
  uint64_t ngroups = 0;
  if (n > 0 || g->outer) {
    const int sb = g->tb.sub_bits;
    uint32_t subcap = 8;
    if (n == 0) {  // outer, no probe row: every region is empty
      CRGPU_TRY(g->tp.rows_b.ensure(16));
      CRGPU_TRY(g->tp.sub_count.ensure((static_cast<size_t>(kRadixP1) << sb) * 4));
      CRGPU_HIP(hipMemsetAsync(g->tp.sub_count.p, 0, (static_cast<size_t>(kRadixP1) << sb) * 4, st));
    } else {
      RadixSrc src;
      CRGPU_TRY(gj_src(key, payload, g->probe_type, n, &src));
      // the probe rows through the build side's two partition levels
      int lg = 0;
      while ((int64_t(1) << lg) < n) lg++;
      int l2rps = lg - 10 - sb;
      // (a probe side much smaller than the build side: INNER leaves it to the composition, which touches only the
      // matching groups; the outer form emits every build row anyway and takes it, its regions nearly empty)
      if (l2rps < 0 && !g->outer) l2rps = 0;
      const double mean = static_cast<double>(n) / static_cast<double>(int64_t(kRadixP1) << sb);
      // regions of twice the mean: probe keys repeat, so a sub-partition's row count varies far more than a Poisson count
      subcap = static_cast<uint32_t>(mean * 2.0 + 8.0 * std::sqrt(mean) + 64.0);
      subcap = (subcap + 7u) & ~7u;
      bool ok = false;
      CRGPU_TRY(radix_partition_split(g->tp, src, n, &ok, subcap, l2rps, sb));
      if (g->tp.sub_bits != sb)
        return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_gjoin_probe: the probe side is too small for the build side's partitioning (compose join -> aggregate)");
      if (!ok) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_gjoin_probe: skewed probe keys overflow a sub-partition (compose join -> aggregate)");
    }
    CRGPU_HIP(hipMemsetAsync(g->counter.p, 0, 8, st));
    CRGPU_HIP(hipMemsetAsync(g->err.p, 0, 4, st));
    GjArgs a;
    std::memset(&a, 0, sizeof(a));
    a.prows = g->tp.rows_b.as<ulonglong2>();
    a.pcount = g->tp.sub_count.as<uint32_t>();
    a.psubcap = subcap;
    a.nplanes = static_cast<uint32_t>(g->nplanes);
    a.table = g->tb.table.as<ulonglong2>();
    a.special = g->tb.special.as<RadixSpecial>();
    a.naggs = g->naggs;
    for (int f = 0; f < g->naggs; f++) a.ops[f] = g->ops[f];
    a.out_count = g->counter.as<unsigned long long>();
    a.err = g->err.as<unsigned int>();
    const size_t lds = static_cast<size_t>(kRadixSlots) * 16 + static_cast<size_t>(g->nplanes) * kGjSlots * 8;
    {
      ProfScope prof("gjoin_fused");
      if (g->outer) hipLaunchKernelGGL(gjoin_fused_kernel<true>, dim3(static_cast<unsigned int>(kRadixP1) << sb), dim3(1024), lds, st, a);
      else hipLaunchKernelGGL(gjoin_fused_kernel<false>, dim3(static_cast<unsigned int>(kRadixP1) << sb), dim3(1024), lds, st, a);
      CRGPU_HIP(hipGetLastError());
      // (the kernel strides: any grid will do)
      const unsigned pack_grid = static_cast<unsigned>(std::min<int64_t>(((cap + 7) / 8 + 255) / 256, static_cast<int64_t>(num_cus()) * 8));
      if (g->outer)  // (the number of groups is known: every build row)
        for (int f = 0; f < g->naggs; f++)
          if (g->nullable[f])
            hipLaunchKernelGGL(gjoin_pack_valid_kernel, dim3(pack_grid), dim3(256), 0, st, g->valid_bytes[f].as<uint8_t>(), cap,
                               g->out[f].valid.as<uint8_t>());
      CRGPU_HIP(hipGetLastError());
    }
    unsigned int err = 0;
    CRGPU_HIP(hipMemcpyAsync(&ngroups, g->counter.p, 8, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipMemcpyAsync(&err, g->err.p, 4, hipMemcpyDeviceToHost, st));
    CRGPU_HIP(hipStreamSynchronize(st));
    if (err) return fail(CRGPU_ERR_INT_OUT_OF_RANGE, "integer out of range");
    if (g->outer && ngroups != static_cast<uint64_t>(cap)) return fail(CRGPU_ERR_HIP, "crgpu_dev_gjoin_probe: build rows and emitted groups differ (internal error)");
  }
  for (int f = 0; f < g->naggs; f++) g->views[f] = g->out[f].view();
  res->n = static_cast<int64_t>(ngroups);
  res->ncols = g->naggs;
  res->_pad = 0;
  res->cols = g->views.data();
  return CRGPU_OK;
}
