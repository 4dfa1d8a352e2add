// partition.hip -- hash repartition: the device half of the HashRouter replacement.
//
// Replaces colexechash.TupleHashDistributor.Distribute
// (pkg/sql/colexec/colexechash/hash_utils.go:163-208: initHash64 -> rehash per
// hash column -> finalizeHash(% numOutputs) -> one selection vector per output)
// and the per-output row copies HashRouter.processNextBatch makes
// (pkg/sql/colflow/routers.go:683-715).  Instead of P selection vectors over
// 1024-row batches, the whole input is regrouped in two streaming passes into
// P contiguous regions (the send buffers of the all-to-all):
//   pass 1: dest = hash64 % P per row + per-tile histograms in LDS (dest kept as uint16 only for BYTES hash columns);
//   scan  : exclusive scan of the P x ntiles histogram (partition-major);
//   pass 2: each 4096-row tile counting-sorts its rows by destination in LDS, column by column, and writes every
//           destination's run base[p][tile] + 0.. as whole lines (part_scatter_lds_kernel).
// Same hash as the reference, so a mixed CPU/GPU cluster co-locates equal keys.
//
// Roofline: HBM streaming; algorithmic bytes/row = row width read + row width written.
#include <cstdlib>
#include <cstring>

#include "device.h"

namespace crgpu {

static constexpr int kPartBlock = 256;
static constexpr int kPartItems = 16;
static constexpr int kPartTile = kPartBlock * kPartItems;
static constexpr int kMaxOutputs = 1024;
static constexpr int kMaxScatter = 16;

__global__ __launch_bounds__(kPartBlock) void part_hist_kernel(KeyCols keys, int64_t n, uint64_t init, uint32_t P,
                                                               int64_t ntiles, uint16_t *__restrict__ dest,
                                                               uint32_t *__restrict__ tile_hist) {
  extern __shared__ uint32_t s_hist[];
  for (uint32_t p = threadIdx.x; p < P; p += kPartBlock) s_hist[p] = 0;
  __syncthreads();
  const int64_t tile = blockIdx.x;
  const int64_t base = tile * kPartTile;
  const bool pow2 = (P & (P - 1)) == 0;
  // all of a thread's rows are requested before the first is hashed (LDS atomics per row are not the bound: counting in
  // registers and adding up per wave measured slower)
  uint32_t dd[kPartItems];
#pragma unroll 4
  for (int k = 0; k < kPartItems; k++) {
    const int64_t i = base + static_cast<int64_t>(k) * kPartBlock + threadIdx.x;
    dd[k] = 0xFFFFFFFFu;
    if (i < n) {
      const uint64_t h = hash_row64(keys, i, init);
      dd[k] = pow2 ? static_cast<uint32_t>(h & (P - 1)) : static_cast<uint32_t>(h % P);  // finalizeHash
    }
  }
#pragma unroll
  for (int k = 0; k < kPartItems; k++) {
    if (dd[k] == 0xFFFFFFFFu) continue;
    if (dest != nullptr) dest[base + static_cast<int64_t>(k) * kPartBlock + threadIdx.x] = static_cast<uint16_t>(dd[k]);
    atomicAdd(&s_hist[dd[k]], 1u);
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < P; p += kPartBlock) tile_hist[static_cast<int64_t>(p) * ntiles + tile] = s_hist[p];
}

struct ScatterCol {
  const uint8_t *src;
  uint8_t *dst;
  const uint8_t *src_valid;
  uint8_t *dst_valid;  // byte per row, packed afterwards
  int32_t width;
  int32_t _pad;
};
struct ScatterArgs {
  int64_t n, ntiles;
  uint32_t P;
  int32_t ncols;
  const uint16_t *dest;    // null (part_scatter_lds_kernel only): the destination is recomputed from `keys`
  const uint64_t *bases;  // [P * ntiles + 1]
  int32_t *src_of_pos;    // optional: inverse map for columns not scattered here
  ScatterCol cols[kMaxScatter];
  KeyCols keys;
  uint64_t init;
};

__global__ __launch_bounds__(kPartBlock) void part_scatter_kernel(ScatterArgs a) {
  extern __shared__ uint32_t s_cursor[];  // next free rank per partition within this tile
  for (uint32_t p = threadIdx.x; p < a.P; p += kPartBlock) s_cursor[p] = 0;
  __syncthreads();
  const int64_t tile = blockIdx.x;
  const int64_t base = tile * kPartTile;
  int64_t pos[kPartItems];
#pragma unroll
  for (int k = 0; k < kPartItems; k++) {
    const int64_t i = base + static_cast<int64_t>(k) * kPartBlock + threadIdx.x;
    pos[k] = -1;
    if (i < a.n) {
      const uint32_t d = a.dest[i];
      const uint32_t r = atomicAdd(&s_cursor[d], 1u);
      pos[k] = static_cast<int64_t>(a.bases[static_cast<int64_t>(d) * a.ntiles + tile] + r);
    }
  }
  for (int c = 0; c < a.ncols; c++) {
    const ScatterCol sc = a.cols[c];
#pragma unroll
    for (int k = 0; k < kPartItems; k++) {
      if (pos[k] < 0) continue;
      const int64_t i = base + static_cast<int64_t>(k) * kPartBlock + threadIdx.x;
      switch (sc.width) {
        case 1: sc.dst[pos[k]] = sc.src[i]; break;
        case 2: reinterpret_cast<uint16_t *>(sc.dst)[pos[k]] = reinterpret_cast<const uint16_t *>(sc.src)[i]; break;
        case 4: reinterpret_cast<uint32_t *>(sc.dst)[pos[k]] = reinterpret_cast<const uint32_t *>(sc.src)[i]; break;
        default: reinterpret_cast<uint64_t *>(sc.dst)[pos[k]] = reinterpret_cast<const uint64_t *>(sc.src)[i]; break;
      }
      if (sc.dst_valid) sc.dst_valid[pos[k]] = (sc.src_valid[i >> 3] >> (i & 7)) & 1;
    }
  }
  if (a.src_of_pos != nullptr) {
#pragma unroll
    for (int k = 0; k < kPartItems; k++)
      if (pos[k] >= 0) a.src_of_pos[pos[k]] = static_cast<int32_t>(base + static_cast<int64_t>(k) * kPartBlock + threadIdx.x);
  }
}

// The scatter of crgpu_dev_partition: a tile's rows are counting-sorted by destination in LDS first, column by column,
// and leave in sorted order -- lane l writes position base[d] + l of its destination's run, so every store instruction
// covers whole lines (the per-row form above writes P partial lines per wave).  With fixed-width hash columns the
// destination is recomputed here instead of being written by the histogram pass and read back (4 B/row less traffic).
static constexpr int kLdsBlock = 1024;                     // 16 waves: 4 rows per thread keep the kernel at < 64 VGPRs
static constexpr int kLdsItems = kPartTile / kLdsBlock;
static constexpr size_t part_lds_bytes(uint32_t P) {
  return static_cast<size_t>(kPartTile) * (8 + 2 + 1) + static_cast<size_t>((P + 1) & ~1u) * 4 + static_cast<size_t>(P) * 8;
}
__device__ __forceinline__ void part_load_col(const ScatterCol &sc, int64_t base, uint32_t rows, uint32_t tid,
                                              uint64_t *v, uint8_t *b) {
#pragma unroll
  for (int k = 0; k < kLdsItems; k++) {
    const uint32_t s = k * kLdsBlock + tid;
    v[k] = 0, b[k] = 1;
    if (s >= rows) continue;
    const int64_t i = base + s;
    switch (sc.width) {
      case 1: v[k] = sc.src[i]; break;
      case 2: v[k] = reinterpret_cast<const uint16_t *>(sc.src)[i]; break;
      case 4: v[k] = reinterpret_cast<const uint32_t *>(sc.src)[i]; break;
      default: v[k] = reinterpret_cast<const uint64_t *>(sc.src)[i]; break;
    }
    if (sc.dst_valid) b[k] = (sc.src_valid[i >> 3] >> (i & 7)) & 1;
  }
}
__global__ __launch_bounds__(kLdsBlock) void part_scatter_lds_kernel(ScatterArgs a) {
  extern __shared__ __align__(16) unsigned char s_raw[];
  uint64_t *s_stage = reinterpret_cast<uint64_t *>(s_raw);                  // [kPartTile] one column's values, sorted
  uint16_t *s_sdest = reinterpret_cast<uint16_t *>(s_stage + kPartTile);    // [kPartTile] destination of sorted slot s
  uint8_t *s_vbyte = reinterpret_cast<uint8_t *>(s_sdest + kPartTile);      // [kPartTile] validity of the staged column
  uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_vbyte + kPartTile);      // [P] rows per destination, then first slots
  long long *s_gbase = reinterpret_cast<long long *>(s_cnt + ((a.P + 1) & ~1u));  // [P] output position of slot s = s_gbase[d] + s
  __shared__ uint32_t s_wsum[4];
  const uint32_t tid = threadIdx.x, P = a.P;
  for (uint32_t p = tid; p < P; p += kLdsBlock) s_cnt[p] = 0;
  __syncthreads();
  const int64_t tile = blockIdx.x;
  const int64_t base = tile * kPartTile;
  const uint32_t rows = a.n - base < kPartTile ? static_cast<uint32_t>(a.n - base) : static_cast<uint32_t>(kPartTile);
  const bool pow2 = (P & (P - 1)) == 0;
  uint32_t d[kLdsItems], lpos[kLdsItems];
#pragma unroll
  for (int k = 0; k < kLdsItems; k++) {
    const uint32_t s = k * kLdsBlock + tid;
    d[k] = 0, lpos[k] = 0;
    if (s < rows) {
      const int64_t i = base + s;
      if (a.dest != nullptr) {
        d[k] = a.dest[i];
      } else {
        const uint64_t h = hash_row64(a.keys, i, a.init);
        d[k] = pow2 ? static_cast<uint32_t>(h & (P - 1)) : static_cast<uint32_t>(h % P);
      }
      lpos[k] = atomicAdd(&s_cnt[d[k]], 1u);  // rank among the tile's rows of this destination
    }
  }
  __syncthreads();
  // exclusive scan of s_cnt by the first 256 threads: thread t owns entries [4t, 4t + 4)  (P <= 1024)
  uint32_t c4[4] = {0, 0, 0, 0}, sum = 0, incl = 0;
  if (tid < 256) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t idx = 4 * tid + e;
      c4[e] = idx < P ? s_cnt[idx] : 0u;
      sum += c4[e];
    }
    incl = wave_incl_scan_u32(sum);
    if (lane_id() == 63) s_wsum[tid >> 6] = incl;
  }
  __syncthreads();
  if (tid < 256) {
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < (tid >> 6); w++) run += s_wsum[w];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t idx = 4 * tid + e;
      if (idx < P) {
        s_cnt[idx] = run;
        s_gbase[idx] = static_cast<long long>(a.bases[static_cast<int64_t>(idx) * a.ntiles + tile]) - static_cast<long long>(run);
        run += c4[e];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kLdsItems; k++) {
    if (k * kLdsBlock + tid < rows) {
      lpos[k] += s_cnt[d[k]];
      s_sdest[lpos[k]] = static_cast<uint16_t>(d[k]);
    }
  }
  __syncthreads();
  // column c + 1 is requested before column c is staged and written: one exposed load latency per tile, not one per column
  uint64_t vn[kLdsItems];
  uint8_t bn[kLdsItems];
  part_load_col(a.cols[0], base, rows, tid, vn, bn);
  for (int c = 0; c < a.ncols; c++) {
    const ScatterCol sc = a.cols[c];
#pragma unroll
    for (int k = 0; k < kLdsItems; k++) {
      if (k * kLdsBlock + tid >= rows) continue;
      s_stage[lpos[k]] = vn[k];
      if (sc.dst_valid) s_vbyte[lpos[k]] = bn[k];
    }
    if (c + 1 < a.ncols) part_load_col(a.cols[c + 1], base, rows, tid, vn, bn);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kLdsItems; k++) {
      const uint32_t s = k * kLdsBlock + tid;
      if (s >= rows) continue;
      const int64_t g = static_cast<int64_t>(s_gbase[s_sdest[s]] + static_cast<long long>(s));
      const uint64_t v = s_stage[s];
      switch (sc.width) {
        case 1: sc.dst[g] = static_cast<uint8_t>(v); break;
        case 2: reinterpret_cast<uint16_t *>(sc.dst)[g] = static_cast<uint16_t>(v); break;
        case 4: __builtin_nontemporal_store(static_cast<uint32_t>(v), reinterpret_cast<uint32_t *>(sc.dst) + g); break;
        default: __builtin_nontemporal_store(v, reinterpret_cast<uint64_t *>(sc.dst) + g); break;
      }
      if (sc.dst_valid) sc.dst_valid[g] = s_vbyte[s];
    }
    __syncthreads();
  }
  if (a.src_of_pos != nullptr) {
    uint32_t *s_row = reinterpret_cast<uint32_t *>(s_stage);
#pragma unroll
    for (int k = 0; k < kLdsItems; k++)
      if (k * kLdsBlock + tid < rows) s_row[lpos[k]] = k * kLdsBlock + tid;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kLdsItems; k++) {
      const uint32_t s = k * kLdsBlock + tid;
      if (s < rows) a.src_of_pos[s_gbase[s_sdest[s]] + static_cast<long long>(s)] = static_cast<int32_t>(base + s_row[s]);
    }
  }
}

// ---- no-allocation variant (crgpu_dev_partition_noalloc): its scratch -- destination codes, per-tile histograms,
// scanned bases, scan tile sums -- lives in this file-scope array, one copy per GPU the library runs on, held until
// the process ends.  `scratch` itself is a host-side stand-in for the symbol, never a device address: host code only
// ever uses the per-device pointer hipGetSymbolAddress returned (noalloc_setup_device, called by crgpu_init).
static constexpr int64_t kNoAllocMaxRows = int64_t(1) << 22;
static constexpr int kNoAllocMaxOutputs = 64;
static constexpr size_t kNoAllocMaxTiles = static_cast<size_t>(kNoAllocMaxRows / kPartTile);
static constexpr size_t kNoAllocDestBytes = static_cast<size_t>(kNoAllocMaxRows) * 2;                       // uint16 per row
static constexpr size_t kNoAllocHistBytes = kNoAllocMaxTiles * kNoAllocMaxOutputs * 4;                      // uint32 per (output, tile)
static constexpr size_t kNoAllocBasesBytes = (kNoAllocMaxTiles * kNoAllocMaxOutputs + 1 + 31) / 32 * 32 * 8;  // uint64 per (output, tile) + total
static constexpr size_t kNoAllocScanBytes = 4096;                                                           // tile sums of the scan (<= 66 words)
static constexpr size_t kNoAllocScratchBytes = kNoAllocDestBytes + kNoAllocHistBytes + kNoAllocBasesBytes + kNoAllocScanBytes;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];
static void *g_scratch_ptr[kMaxDevices];  // each GPU holds its own copy of the array at its own address

crgpu_status noalloc_setup_device(int device) {  // with `device` current (crgpu_init)
  if (device < 0 || device >= kMaxDevices) return fail(CRGPU_ERR_INVALID, "noalloc_setup_device: device ordinal out of range");
  void *ptr = nullptr;
  CRGPU_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  if (ptr == nullptr) return fail(CRGPU_ERR_HIP, "noalloc_setup_device: hipGetSymbolAddress returned a null address");
  g_scratch_ptr[device] = ptr;
  return CRGPU_OK;
}

__global__ void part_counts_kernel(const uint64_t *__restrict__ bases, int64_t ntiles, uint32_t P,
                                   uint64_t *__restrict__ counts) {
  for (uint32_t p = threadIdx.x; p < P; p += blockDim.x)
    counts[p] = bases[static_cast<int64_t>(p + 1) * ntiles] - bases[static_cast<int64_t>(p) * ntiles];
}

}  // namespace crgpu

using namespace crgpu;

struct crgpu_partitioner : crgpu::Handle {
  std::vector<int> types, hash_cols;
  uint64_t init = 1;
  int P = 1;
  DevBuf dest, tile_hist, bases, src_of_pos, counts_dev, scratch;
  std::vector<OutCol> out;
  std::vector<DevBuf> valid_bytes;
  std::vector<crgpu_dcol> views;
};

extern "C" crgpu_status crgpu_dev_partitioner_create(const int32_t *types, int32_t n_cols, const int32_t *hash_cols,
                                                     int32_t n_hash, uint64_t init, int32_t num_outputs,
                                                     crgpu_partitioner **out) {
  CRGPU_TRY(ensure_init());
  if (types == nullptr || out == nullptr || (n_hash > 0 && hash_cols == nullptr))
    return fail(CRGPU_ERR_INVALID, "crgpu_dev_partitioner_create: null argument");
  if (n_cols < 1 || n_cols > kMaxCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partitioner_create: column count out of range");
  if (n_hash < 0 || n_hash > kMaxKeyCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partitioner_create: at most 8 hash columns");
  if (num_outputs < 1 || num_outputs > kMaxOutputs)
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partitioner_create: between 1 and 1024 outputs");
  if (init == 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_partitioner_create: init hash value must be non-zero (hash_utils.go:64)");
  auto p = new crgpu_partitioner();
  {
    crgpu_status _os = p->open();
    if (_os != CRGPU_OK) { delete p; return _os; }
  }
  CRGPU_ENTER(p);
  for (int i = 0; i < n_cols; i++) {
    if (!valid_type(types[i])) { delete p; return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partitioner_create: unsupported column type"); }
    p->types.push_back(types[i]);
  }
  for (int i = 0; i < n_hash; i++) {
    if (hash_cols[i] < 0 || hash_cols[i] >= n_cols) { delete p; return fail(CRGPU_ERR_INVALID, "crgpu_dev_partitioner_create: hash column ordinal out of range"); }
    p->hash_cols.push_back(hash_cols[i]);
  }
  p->init = init;
  p->P = num_outputs;
  p->out = std::vector<OutCol>(n_cols);
  p->valid_bytes = std::vector<DevBuf>(n_cols);
  p->views.resize(n_cols);
  *out = p;
  return CRGPU_OK;
}

extern "C" void crgpu_dev_partitioner_destroy(crgpu_partitioner *p) {
  if (p == nullptr) return;
  if (p->st != nullptr) (void)hipStreamSynchronize(p->st);  // its buffers return to the pool: no kernel may still use them
  delete p;
}

static crgpu_status crgpu_dev_partition_impl(crgpu_partitioner *p, const crgpu_dcol *cols, int64_t n,
                                            crgpu_dev_result *res, int64_t *counts_host) {
  if (p == nullptr || res == nullptr || counts_host == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_partition: null argument");
  CRGPU_ENTER(p);
  CRGPU_CHECK_CANCEL(p);
  if (n < 0 || n > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partition: at most 2^31 rows per call");
  if (n > 0 && cols == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_partition: null columns");
  const int nc = static_cast<int>(p->types.size());
  hipStream_t st = stream();
  std::vector<DCol> in;
  for (int c = 0; c < nc; c++) {
    DCol d{p->types[c], 0, nullptr, nullptr, nullptr};
    if (n > 0) {
      if (cols[c].type != p->types[c]) return fail(CRGPU_ERR_INVALID, "crgpu_dev_partition: column type differs from the spec");
      if (cols[c].type == CRGPU_BYTES && cols[c].offsets == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_partition: BYTES column without offsets");
      if (cols[c].type != CRGPU_BYTES && cols[c].data == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_partition: null data pointer");
      d = to_dcol(cols[c]);
    }
    in.push_back(d);
  }
  for (int q = 0; q < p->P; q++) counts_host[q] = 0;
  if (n == 0) {
    for (int c = 0; c < nc; c++) {
      OutCol &o = p->out[c];
      o.type = p->types[c];
      o.nullable = cols != nullptr && cols[c].valid != nullptr;  // an empty input keeps its columns' nullability
      CRGPU_TRY(o.data.ensure(16));
      if (o.nullable) CRGPU_TRY(o.valid.ensure(8));
      if (o.type == CRGPU_BYTES) {
        CRGPU_TRY(o.offsets.ensure(4));
        CRGPU_HIP(hipMemsetAsync(o.offsets.p, 0, 4, st));
      }
      p->views[c] = o.view();
    }
    CRGPU_HIP(hipStreamSynchronize(st));
    res->n = 0; res->ncols = nc; res->_pad = 0; res->cols = p->views.data();
    return CRGPU_OK;
  }

  const int64_t ntiles = (n + kPartTile - 1) / kPartTile;
  const uint32_t P = static_cast<uint32_t>(p->P);
  KeyCols keys;
  keys.n = static_cast<int>(p->hash_cols.size());
  keys._pad = 0;
  for (int k = 0; k < keys.n; k++) keys.c[k] = in[p->hash_cols[k]];
  // BYTES hash columns: the destination is computed once and kept (2 B/row); fixed-width ones are hashed again by the scatter
  bool keep_dest = false;
  for (int k = 0; k < keys.n; k++) keep_dest = keep_dest || keys.c[k].type == CRGPU_BYTES;
  const char *lds_env = std::getenv("CRGPU_PARTITION_LDS");  // "0": the per-row scatter (tests, measurements)
  const bool lds_scatter = lds_env == nullptr || lds_env[0] != '0';
  if (!lds_scatter) keep_dest = true;
  if (keep_dest) CRGPU_TRY(p->dest.ensure(static_cast<size_t>(n) * 2));
  CRGPU_TRY(p->tile_hist.ensure(static_cast<size_t>(ntiles) * P * 4));
  CRGPU_TRY(p->bases.ensure((static_cast<size_t>(ntiles) * P + 1) * 8));
  CRGPU_TRY(p->counts_dev.ensure(static_cast<size_t>(P) * 8));

  ScatterArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n = n;
  a.ntiles = ntiles;
  a.P = P;
  a.dest = keep_dest ? p->dest.as<uint16_t>() : nullptr;
  a.bases = p->bases.as<uint64_t>();
  a.keys = keys;
  a.init = p->init;
  std::vector<int> later;
  std::vector<int> scattered;
  for (int c = 0; c < nc; c++) {
    OutCol &o = p->out[c];
    o.type = in[c].type;
    o.nullable = in[c].valid != nullptr;
    if (in[c].type != CRGPU_BYTES && a.ncols < kMaxScatter) {
      CRGPU_TRY(o.data.ensure(static_cast<size_t>(n) * type_width(in[c].type)));
      ScatterCol &sc = a.cols[a.ncols++];
      scattered.push_back(c);
      sc.src = static_cast<const uint8_t *>(in[c].data);
      sc.dst = o.data.as<uint8_t>();
      sc.width = type_width(in[c].type);
      sc.src_valid = in[c].valid;
      sc.dst_valid = nullptr;
      if (o.nullable) {
        CRGPU_TRY(p->valid_bytes[c].ensure(static_cast<size_t>(n)));
        CRGPU_TRY(o.valid.ensure(static_cast<size_t>((n + 7) / 8 + 1)));
        sc.dst_valid = p->valid_bytes[c].as<uint8_t>();
      }
    } else {
      later.push_back(c);
    }
  }
  if (!later.empty()) {
    CRGPU_TRY(p->src_of_pos.ensure(static_cast<size_t>(n) * 4));
    a.src_of_pos = p->src_of_pos.as<int32_t>();
  }
  {
    ProfScope prof("partition");
    hipLaunchKernelGGL(part_hist_kernel, dim3(ntiles), dim3(kPartBlock), P * 4, st, keys, n, p->init, P, ntiles,
                       keep_dest ? p->dest.as<uint16_t>() : nullptr, p->tile_hist.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
    CRGPU_TRY(exclusive_scan_u32(p->tile_hist.as<uint32_t>(), ntiles * P, p->bases.as<uint64_t>(), p->scratch, nullptr));
    if (lds_scatter)
      hipLaunchKernelGGL(part_scatter_lds_kernel, dim3(ntiles), dim3(kLdsBlock), part_lds_bytes(P), st, a);
    else
      hipLaunchKernelGGL(part_scatter_kernel, dim3(ntiles), dim3(kPartBlock), P * 4, st, a);
    CRGPU_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(part_counts_kernel, dim3(1), dim3(256), 0, st, p->bases.as<uint64_t>(), ntiles, P,
                     p->counts_dev.as<uint64_t>());
  CRGPU_HIP(hipGetLastError());
  for (size_t s = 0; s < scattered.size(); s++)
    if (a.cols[s].dst_valid != nullptr)
      CRGPU_TRY(pack_bitmap(a.cols[s].dst_valid, n, p->out[scattered[s]].valid.as<uint8_t>()));
  for (int c : later) CRGPU_TRY(materialize(in[c], p->src_of_pos.as<int32_t>(), n, false, p->out[c], p->scratch));
  std::vector<uint64_t> cnt(P);
  CRGPU_HIP(hipMemcpyAsync(cnt.data(), p->counts_dev.p, static_cast<size_t>(P) * 8, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  for (uint32_t q = 0; q < P; q++) counts_host[q] = static_cast<int64_t>(cnt[q]);
  for (int c = 0; c < nc; c++) p->views[c] = p->out[c].view();
  res->n = n;
  res->ncols = nc;
  res->_pad = 0;
  res->cols = p->views.data();
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_dev_partition(crgpu_partitioner *p, const crgpu_dcol *cols, int64_t n,
                                            crgpu_dev_result *res, int64_t *counts_host) {
  const crgpu_status s_ = crgpu_dev_partition_impl(p, cols, n, res, counts_host);
  if (s_ == CRGPU_OK && p != nullptr) note_rows(p, n, n);
  return s_;
}

// The one entry point that allocates nothing (include/crgpu.h): same kernels as crgpu_dev_partition, scratch from the
// per-device static array, outputs and counts in caller-owned device memory, queued on the device's default stream.
extern "C" crgpu_status crgpu_dev_partition_noalloc(const int32_t *types, int32_t n_cols, const int32_t *hash_cols,
                                                    int32_t n_hash, uint64_t init, int32_t num_outputs,
                                                    const crgpu_dcol *cols, int64_t n, void *const *out_data,
                                                    int64_t *counts_dev) {
  CRGPU_ENTER(nullptr);
  if (types == nullptr || cols == nullptr || out_data == nullptr || counts_dev == nullptr || (n_hash > 0 && hash_cols == nullptr))
    return fail(CRGPU_ERR_INVALID, "crgpu_dev_partition_noalloc: null argument");
  if (n_cols < 1 || n_cols > kMaxScatter)
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partition_noalloc: between 1 and 16 columns");
  if (n_hash < 0 || n_hash > kMaxKeyCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partition_noalloc: at most 8 hash columns");
  if (num_outputs < 1 || num_outputs > kNoAllocMaxOutputs)
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partition_noalloc: between 1 and 64 outputs (the static scratch is sized for that)");
  if (n < 0 || n > kNoAllocMaxRows)
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partition_noalloc: at most 2^22 rows per call (the static scratch is sized for that); "
                                       "use crgpu_dev_partition for larger inputs");
  if (init == 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_partition_noalloc: init hash value must be non-zero (hash_utils.go:64)");
  for (int c = 0; c < n_cols; c++) {
    if (!valid_type(types[c]) || types[c] == CRGPU_BYTES)
      return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partition_noalloc: fixed-width columns only");
    if (cols[c].type != types[c]) return fail(CRGPU_ERR_INVALID, "crgpu_dev_partition_noalloc: column type differs from the spec");
    if (cols[c].valid != nullptr) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_partition_noalloc: columns with NULLs take crgpu_dev_partition");
    if (n > 0 && (cols[c].data == nullptr || out_data[c] == nullptr)) return fail(CRGPU_ERR_INVALID, "crgpu_dev_partition_noalloc: null data pointer");
  }
  for (int k = 0; k < n_hash; k++)
    if (hash_cols[k] < 0 || hash_cols[k] >= n_cols) return fail(CRGPU_ERR_INVALID, "crgpu_dev_partition_noalloc: hash column ordinal out of range");
  const int dev = current_device();
  uint8_t *base = dev >= 0 ? static_cast<uint8_t *>(g_scratch_ptr[dev]) : nullptr;
  if (base == nullptr) return fail(CRGPU_ERR_STATE, "crgpu_dev_partition_noalloc: the device's scratch was not resolved (crgpu_init)");
  hipStream_t st = stream();
  const uint32_t P = static_cast<uint32_t>(num_outputs);
  if (n == 0) {
    CRGPU_HIP(hipMemsetAsync(counts_dev, 0, static_cast<size_t>(P) * 8, st));
    mark_default_work();
    return CRGPU_OK;
  }
  const int64_t ntiles = (n + kPartTile - 1) / kPartTile;  // <= kNoAllocMaxTiles by the row limit above
  uint16_t *dest = reinterpret_cast<uint16_t *>(base);
  uint32_t *tile_hist = reinterpret_cast<uint32_t *>(base + kNoAllocDestBytes);
  uint64_t *bases = reinterpret_cast<uint64_t *>(base + kNoAllocDestBytes + kNoAllocHistBytes);
  uint64_t *scan_ts = reinterpret_cast<uint64_t *>(base + kNoAllocDestBytes + kNoAllocHistBytes + kNoAllocBasesBytes);
  if (scan_scratch_words(ntiles * P) * 8 > kNoAllocScanBytes) return fail(CRGPU_ERR_STATE, "crgpu_dev_partition_noalloc: scan scratch too small");
  KeyCols keys;
  keys.n = n_hash;
  keys._pad = 0;
  for (int k = 0; k < n_hash; k++) keys.c[k] = to_dcol(cols[hash_cols[k]]);
  ScatterArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n = n;
  a.ntiles = ntiles;
  a.P = P;
  a.dest = dest;
  a.bases = bases;
  a.ncols = n_cols;
  for (int c = 0; c < n_cols; c++) {
    a.cols[c].src = static_cast<const uint8_t *>(cols[c].data);
    a.cols[c].dst = static_cast<uint8_t *>(out_data[c]);
    a.cols[c].width = type_width(types[c]);
  }
  // nothing in the scratch has to start at zero: every (output, tile) histogram entry, every destination code and every
  // scanned base this call reads is written by this call first; what an earlier, larger call left beyond them is never read
  // (no ProfScope here: creating its events would be an allocation)
  hipLaunchKernelGGL(part_hist_kernel, dim3(ntiles), dim3(kPartBlock), P * 4, st, keys, n, init, P, ntiles, dest, tile_hist);
  CRGPU_HIP(hipGetLastError());
  CRGPU_TRY(exclusive_scan_u32_ws(tile_hist, ntiles * P, bases, scan_ts, nullptr));
  hipLaunchKernelGGL(part_scatter_kernel, dim3(ntiles), dim3(kPartBlock), P * 4, st, a);
  CRGPU_HIP(hipGetLastError());
  hipLaunchKernelGGL(part_counts_kernel, dim3(1), dim3(256), 0, st, bases, ntiles, P, reinterpret_cast<uint64_t *>(counts_dev));
  CRGPU_HIP(hipGetLastError());
  mark_default_work();
  return CRGPU_OK;
}
