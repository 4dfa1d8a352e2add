// host_ops.hip -- layer (2) of include/crgpu.h: the operator boundary a cgo shell binds.
//
// Takes coldata.Batch-shaped HOST memory exactly as Go lays it out (pkg/col/coldata:
// dense value slices, 32-byte Bytes elements + buffer, validity bitmaps, []int
// selection vectors), stages it (deselect + Bytes -> offsets/data), moves it to HBM
// in large pieces, drives the bulk device API and hands results back in caller-owned
// output batches -- the colexecop.Operator Next() contract
// (pkg/sql/colexecop/operator.go:22-58).  Nothing borrowed is retained across calls.
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "common.h"

namespace crgpu {

static constexpr int64_t kFlushRows = int64_t(1) << 20;  // staged probe / agg / router rows before a device round trip
static constexpr int kElemSize = 32;                     // coldata.ElementSize (bytes.go:76-78)
static constexpr int kMaxInline = 30;                    // coldata.BytesMaxInlineLength (bytes.go:80-82)

// ------------------------------------------------------------------ host-side staging
// std::vector<uint8_t>::resize zero-fills, which doubles the memory traffic of a buffer that is overwritten at once
// (a D2H copy, a memcpy): this allocator default-initialises instead, i.e. leaves the bytes alone.
// Large staging and result buffers are PINNED (hipHostMalloc) and pooled: a transfer from or to pageable memory is
// staged through the runtime's own bounce buffers (D2H measured 29 GB/s against ~55 from pinned memory), and a fresh
// pageable buffer pays a page fault per 4 KiB on first touch.  Blocks are powers of two from 64 KiB, returned to a
// process-wide free list (kept until the process ends; beyond CRGPU_PINNED_MAX_BYTES, 8 GiB, blocks are pageable).
class PinnedPool {
 public:
  static PinnedPool &get() {
    static PinnedPool *p = new PinnedPool();  // never destroyed: hipHostFree after the runtime has shut down is not safe
    return *p;
  }
  void *alloc(size_t bytes) {
    if (bytes < kMin) return std::malloc(bytes ? bytes : 1);
    const int b = bucket(bytes);
    const size_t size = size_t(1) << b;
    {
      std::lock_guard<std::mutex> g(mu_);
      if (!free_[b].empty()) {
        void *p = free_[b].back();
        free_[b].pop_back();
        own(p, size);
        return p;
      }
      if (pinned_bytes_ + size > max_bytes_) return std::malloc(bytes);
      pinned_bytes_ += size;
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, size, hipHostMallocPortable) != hipSuccess || p == nullptr) {
      (void)hipGetLastError();
      std::lock_guard<std::mutex> g(mu_);
      pinned_bytes_ -= size;
      return std::malloc(bytes);
    }
    std::lock_guard<std::mutex> g(mu_);
    pinned_.emplace(p, nullptr);
    own(p, size);
    return p;
  }
  void release(void *p, size_t bytes) {
    if (p == nullptr) return;
    if (bytes >= kMin) {
      std::lock_guard<std::mutex> g(mu_);
      auto it = pinned_.find(p);
      if (it != pinned_.end()) {
        account_mem(it->second, 0, -(int64_t(1) << bucket(bytes)));
        it->second = nullptr;
        free_[bucket(bytes)].push_back(p);
        return;
      }
    }
    std::free(p);
  }

 private:
  static constexpr size_t kMin = size_t(64) << 10;
  static int bucket(size_t bytes) {
    int b = 16;
    while ((size_t(1) << b) < bytes) b++;
    return b;
  }
  PinnedPool() {
    const char *e = std::getenv("CRGPU_PINNED_MAX_BYTES");
    max_bytes_ = e != nullptr ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : size_t(8) << 30;
  }
  std::mutex mu_;
  std::vector<void *> free_[64];
  void own(void *p, size_t size) {  // mu_ held: the block belongs to the handle whose call took it
    Handle *o = current_handle();
    pinned_[p] = o;
    account_mem(o, 0, static_cast<int64_t>(size));
  }
  std::unordered_map<void *, Handle *> pinned_;  // pinned block -> current owner (nullptr: idle or unowned)
  size_t pinned_bytes_ = 0, max_bytes_ = 0;
};

template <class T>
struct NoInitAlloc {
  using value_type = T;
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
  T *allocate(size_t n) {
    void *p = PinnedPool::get().alloc(n * sizeof(T));
    if (p == nullptr) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, size_t n) noexcept { PinnedPool::get().release(p, n * sizeof(T)); }
  template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
  template <class U> bool operator==(const NoInitAlloc<U> &) const { return true; }
  template <class U> bool operator!=(const NoInitAlloc<U> &) const { return false; }
};
using ByteVec = std::vector<uint8_t, NoInitAlloc<uint8_t>>;

struct HostCol {
  int type = 0;
  ByteVec data;                   // fixed-width values or BYTES payload
  std::vector<uint32_t> offsets;  // BYTES: n+1
  std::vector<uint8_t> present;   // one byte per row; may be shorter than n (missing rows are present) while !any_null
  bool any_null = false;
  int64_t n = 0;
  void init(int t) {
    type = t;
    clear();
  }
  void fill_present() {
    if (present.size() < static_cast<size_t>(n)) present.resize(static_cast<size_t>(n), 1);
  }
  void clear() {
    data.clear();
    offsets.clear();
    present.clear();
    if (type == CRGPU_BYTES) offsets.push_back(0);
    any_null = false;
    n = 0;
  }
};

static inline bool host_is_null(const crgpu_col &c, int64_t i) {
  return c.maybe_has_nulls && c.nulls != nullptr && !((c.nulls[i >> 3] >> (i & 7)) & 1);
}

// element.get (bytes.go:93-103)
static inline const uint8_t *elem_get(const crgpu_col &c, int64_t i, uint32_t *len) {
  const uint8_t *e = static_cast<const uint8_t *>(c.data) + i * kElemSize;
  if (e[31]) {  // inlined
    *len = e[30];
    return e;
  }
  int64_t off, l;
  std::memcpy(&off, e, 8);
  std::memcpy(&l, e + 8, 8);
  *len = static_cast<uint32_t>(l);
  return c.bytes_buf + off;
}

// AppendTuples with deselection (colexecutils/utils.go:204)
static crgpu_status stage_batch(std::vector<HostCol> &cols, const crgpu_batch *b) {
  if (b == nullptr) return fail(CRGPU_ERR_INVALID, "null batch");
  if (b->length < 0) return fail(CRGPU_ERR_INVALID, "negative batch length");
  if (b->ncols != static_cast<int>(cols.size())) return fail(CRGPU_ERR_INVALID, "batch column count differs from the spec");
  if (b->length == 0) return CRGPU_OK;
  if (b->cols == nullptr) return fail(CRGPU_ERR_INVALID, "batch without columns");
  const int64_t n = b->length;
  for (size_t ci = 0; ci < cols.size(); ci++) {
    const crgpu_col &c = b->cols[ci];
    HostCol &h = cols[ci];
    if (c.type != h.type) return fail(CRGPU_ERR_INVALID, "batch column type differs from the spec");
    if (c.data == nullptr) return fail(CRGPU_ERR_INVALID, "batch column without data");
    // the per-row present bytes are kept only from the first NULL on (a Go batch without NULLs costs one memcpy per column)
    const bool no_nulls = !c.maybe_has_nulls || c.nulls == nullptr;
    const size_t old = static_cast<size_t>(h.n);
    const bool track = !no_nulls || h.any_null;
    if (track) {
      h.fill_present();
      h.present.resize(old + n, 1);
    }
    if (h.type == CRGPU_BYTES) {
      for (int64_t r = 0; r < n; r++) {
        const int64_t i = b->sel ? b->sel[r] : r;
        const bool isnull = !no_nulls && host_is_null(c, i);
        if (isnull) {
          h.present[old + r] = 0;
          h.any_null = true;
        } else {
          uint32_t len;
          const uint8_t *p = elem_get(c, i, &len);
          if (h.data.size() + len > 0xFFFFFFF0ull) return fail(CRGPU_ERR_UNSUPPORTED, "a staged BYTES column exceeds 4 GiB");
          h.data.insert(h.data.end(), p, p + len);
        }
        h.offsets.push_back(static_cast<uint32_t>(h.data.size()));
      }
    } else {
      const int w = type_width(h.type);
      const size_t oldb = h.data.size();
      h.data.resize(oldb + static_cast<size_t>(n) * w);  // (ByteVec: not zero-filled)
      const uint8_t *src = static_cast<const uint8_t *>(c.data);
      if (b->sel == nullptr) {
        std::memcpy(h.data.data() + oldb, src, static_cast<size_t>(n) * w);
      } else {
        for (int64_t r = 0; r < n; r++) std::memcpy(h.data.data() + oldb + r * w, src + b->sel[r] * w, w);
      }
      if (!no_nulls) {
        for (int64_t r = 0; r < n; r++) {
          if (host_is_null(c, b->sel ? b->sel[r] : r)) {
            h.present[old + r] = 0;
            h.any_null = true;
          }
        }
      }
    }
    h.n += n;
  }
  return CRGPU_OK;
}

// ------------------------------------------------------------------ device copies of staged tables
struct DevTable {
  std::vector<DevBuf> data, valid, offsets;
  std::vector<crgpu_dcol> cols;
  int64_t n = 0;
};

static crgpu_status upload_table(const std::vector<HostCol> &h, DevTable &t) {
  const size_t nc = h.size();
  if (t.data.size() != nc) {
    t.data = std::vector<DevBuf>(nc);
    t.valid = std::vector<DevBuf>(nc);
    t.offsets = std::vector<DevBuf>(nc);
    t.cols.resize(nc);
  }
  t.n = nc ? h[0].n : 0;
  std::vector<uint8_t> bitmap;
  for (size_t c = 0; c < nc; c++) {
    const HostCol &hc = h[c];
    crgpu_dcol &d = t.cols[c];
    d.type = hc.type;
    d._pad = 0;
    CRGPU_TRY(t.data[c].ensure(hc.data.size() ? hc.data.size() : 16));
    if (!hc.data.empty()) CRGPU_HIP(hipMemcpy(t.data[c].p, hc.data.data(), hc.data.size(), hipMemcpyHostToDevice));
    d.data = t.data[c].p;
    d.offsets = nullptr;
    if (hc.type == CRGPU_BYTES) {
      CRGPU_TRY(t.offsets[c].ensure(hc.offsets.size() * 4));
      CRGPU_HIP(hipMemcpy(t.offsets[c].p, hc.offsets.data(), hc.offsets.size() * 4, hipMemcpyHostToDevice));
      d.offsets = t.offsets[c].as<uint32_t>();
    }
    d.valid = nullptr;
    if (hc.any_null) {
      bitmap.assign(static_cast<size_t>((hc.n + 7) / 8), 0xFF);
      for (int64_t i = 0; i < hc.n; i++)
        if (!hc.present[i]) bitmap[i >> 3] &= static_cast<uint8_t>(~(1u << (i & 7)));
      CRGPU_TRY(t.valid[c].ensure(bitmap.size() ? bitmap.size() : 1));
      CRGPU_HIP(hipMemcpy(t.valid[c].p, bitmap.data(), bitmap.size(), hipMemcpyHostToDevice));
      d.valid = t.valid[c].as<uint8_t>();
    }
  }
  return CRGPU_OK;
}

// Tables of fixed-width columns without NULLs can be uploaded asynchronously (their sizes are known up front); the
// staged host columns must then stay untouched until the copies have run.
static bool plain_fixed(const std::vector<HostCol> &h) {
  for (const HostCol &c : h)
    if (c.type == CRGPU_BYTES || c.any_null) return false;
  return !h.empty();
}
static crgpu_status upload_table_async(const std::vector<HostCol> &h, DevTable &t) {
  const size_t nc = h.size();
  if (t.data.size() != nc) {
    t.data = std::vector<DevBuf>(nc);
    t.valid = std::vector<DevBuf>(nc);
    t.offsets = std::vector<DevBuf>(nc);
    t.cols.resize(nc);
  }
  t.n = h[0].n;
  for (size_t c = 0; c < nc; c++) {
    crgpu_dcol &d = t.cols[c];
    d.type = h[c].type;
    d._pad = 0;
    CRGPU_TRY(t.data[c].ensure(h[c].data.size() ? h[c].data.size() : 16));
    if (!h[c].data.empty()) CRGPU_HIP(hipMemcpyAsync(t.data[c].p, h[c].data.data(), h[c].data.size(), hipMemcpyHostToDevice, stream()));
    d.data = t.data[c].p;
    d.offsets = nullptr;
    d.valid = nullptr;
  }
  return CRGPU_OK;
}

// result of a bulk call copied back to the host, served in output batches
struct HostResult {
  std::vector<HostCol> cols;
  int64_t n = 0, cursor = 0;
  hipEvent_t ready = nullptr;  // set while the device-to-host copies of the columns are still in flight
  HostResult() = default;
  HostResult(const HostResult &) = delete;
  HostResult &operator=(const HostResult &) = delete;
  ~HostResult() {
    if (ready != nullptr) {
      (void)hipEventSynchronize(ready);  // the copies write into `cols`
      (void)hipEventDestroy(ready);
    }
  }
  crgpu_status wait() {
    if (ready == nullptr) return CRGPU_OK;
    const hipError_t e = hipEventSynchronize(ready);
    (void)hipEventDestroy(ready);
    ready = nullptr;
    CRGPU_HIP(e);
    return CRGPU_OK;
  }
};


static crgpu_status download_result(const crgpu_dev_result &r, HostResult &out) {
  out.n = r.n;
  out.cursor = 0;
  out.cols.resize(r.ncols);
  CRGPU_HIP(hipStreamSynchronize(stream()));
  std::vector<uint8_t> bitmap;
  for (int c = 0; c < r.ncols; c++) {
    const crgpu_dcol &d = r.cols[c];
    HostCol &h = out.cols[c];
    h.init(d.type);
    h.n = r.n;
    if (d.valid != nullptr && r.n > 0) {
      h.present.assign(static_cast<size_t>(r.n), 1);
      bitmap.resize(static_cast<size_t>((r.n + 7) / 8));
      CRGPU_HIP(hipMemcpy(bitmap.data(), d.valid, bitmap.size(), hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < r.n; i++) {
        h.present[i] = (bitmap[i >> 3] >> (i & 7)) & 1;
        if (!h.present[i]) h.any_null = true;
      }
    }
    if (d.type == CRGPU_BYTES) {
      h.offsets.assign(static_cast<size_t>(r.n + 1), 0);
      if (r.n > 0) CRGPU_HIP(hipMemcpy(h.offsets.data(), d.offsets, static_cast<size_t>(r.n + 1) * 4, hipMemcpyDeviceToHost));
      h.data.resize(h.offsets[r.n]);
      if (!h.data.empty()) CRGPU_HIP(hipMemcpy(h.data.data(), d.data, h.data.size(), hipMemcpyDeviceToHost));
    } else {
      h.data.resize(static_cast<size_t>(r.n) * type_width(d.type));
      if (!h.data.empty()) CRGPU_HIP(hipMemcpy(h.data.data(), d.data, h.data.size(), hipMemcpyDeviceToHost));
    }
  }
  return CRGPU_OK;
}

// The same without waiting, for results made of fixed-width columns that cannot be NULL: the copies are enqueued behind
// the kernels that produce the columns and `out.ready` tells when they have landed.  The device columns stay valid
// until the handle's next call, whose kernels run behind these copies on the same stream.
static crgpu_status download_result_async(const crgpu_dev_result &r, HostResult &out) {
  for (int c = 0; c < r.ncols; c++)
    if (r.cols[c].type == CRGPU_BYTES || r.cols[c].valid != nullptr) return download_result(r, out);
  out.n = r.n;
  out.cursor = 0;
  out.cols.resize(r.ncols);
  for (int c = 0; c < r.ncols; c++) {
    HostCol &h = out.cols[c];
    h.init(r.cols[c].type);
    h.n = r.n;
    h.data.resize(static_cast<size_t>(r.n) * type_width(r.cols[c].type));
    if (!h.data.empty()) CRGPU_HIP(hipMemcpyAsync(h.data.data(), r.cols[c].data, h.data.size(), hipMemcpyDeviceToHost, stream()));
  }
  CRGPU_HIP(hipEventCreateWithFlags(&out.ready, hipEventDisableTiming));
  CRGPU_HIP(hipEventRecord(out.ready, stream()));
  return CRGPU_OK;
}

// Copy rows [cursor, cursor+k) of `src` into the caller's output batch (coldata layout).
// Returns false (and fills bytes_buf_len with the required sizes) if a BYTES spill buffer is too small.
static crgpu_status emit_rows(HostResult &src, crgpu_out_batch *out, bool *emitted) {
  *emitted = false;
  out->length = 0;
  if (out->ncols != static_cast<int>(src.cols.size())) return fail(CRGPU_ERR_INVALID, "output batch column count differs from the operator's output schema");
  if (out->capacity <= 0) return fail(CRGPU_ERR_INVALID, "output batch without capacity");
  const int64_t k = std::min<int64_t>(out->capacity, src.n - src.cursor);
  if (k <= 0) return CRGPU_OK;
  const bool borrow = (out->flags & CRGPU_OUT_BORROW) != 0;
  // pass 1: BYTES spill requirements
  bool fits = true;
  for (size_t c = 0; c < src.cols.size(); c++) {
    crgpu_out_col &o = out->cols[c];
    const HostCol &h = src.cols[c];
    if (o.type != h.type) return fail(CRGPU_ERR_INVALID, "output batch column type differs from the operator's output schema");
    o.bytes_buf_len = 0;
    if (h.type != CRGPU_BYTES) continue;
    int64_t need = 0;
    for (int64_t r = 0; r < k; r++) {
      const uint32_t len = h.offsets[src.cursor + r + 1] - h.offsets[src.cursor + r];
      if (len > static_cast<uint32_t>(kMaxInline)) need += len;
    }
    o.bytes_buf_len = need;
    if (need > o.bytes_buf_cap) fits = false;
  }
  if (!fits) return CRGPU_OK;  // length stays 0, bytes_buf_len tells the caller how much to allocate
  for (size_t c = 0; c < src.cols.size(); c++) {
    crgpu_out_col &o = out->cols[c];
    const HostCol &h = src.cols[c];
    if (o.data == nullptr && !(borrow && h.type != CRGPU_BYTES)) return fail(CRGPU_ERR_INVALID, "output column without data buffer");
    o.has_nulls = 0;
    if (o.nulls != nullptr) std::memset(o.nulls, 0xFF, static_cast<size_t>((out->capacity + 7) / 8));
    for (int64_t r = 0; h.any_null && r < k; r++) {
      if (!h.present[src.cursor + r]) {
        if (o.nulls == nullptr) return fail(CRGPU_ERR_INVALID, "output column needs a null bitmap");
        o.nulls[r >> 3] &= static_cast<uint8_t>(~(1u << (r & 7)));
        o.has_nulls = 1;
      }
    }
    if (h.type == CRGPU_BYTES) {
      uint8_t *elems = static_cast<uint8_t *>(o.data);
      int64_t spill = 0;
      for (int64_t r = 0; r < k; r++) {
        const uint32_t a = h.offsets[src.cursor + r], len = h.offsets[src.cursor + r + 1] - a;
        uint8_t *e = elems + r * kElemSize;
        std::memset(e, 0, kElemSize);
        if (len <= static_cast<uint32_t>(kMaxInline)) {  // element.set (bytes.go:105-112)
          if (len) std::memcpy(e, h.data.data() + a, len);
          e[30] = static_cast<uint8_t>(len);
          e[31] = 1;
        } else {
          const int64_t l64 = len;
          std::memcpy(e, &spill, 8);
          std::memcpy(e + 8, &l64, 8);
          std::memcpy(e + 16, &l64, 8);
          std::memcpy(o.bytes_buf + spill, h.data.data() + a, len);
          spill += len;
        }
      }
    } else {
      const int w = type_width(h.type);
      if (borrow)  // the values stay where the transfer put them (pinned, library-owned, alive until the next output call)
        o.data = const_cast<uint8_t *>(h.data.data()) + static_cast<size_t>(src.cursor) * w;
      else
        std::memcpy(o.data, h.data.data() + static_cast<size_t>(src.cursor) * w, static_cast<size_t>(k) * w);
    }
  }
  src.cursor += k;
  out->length = k;
  *emitted = true;
  return CRGPU_OK;
}

static crgpu_status emit_from_queue(std::deque<std::unique_ptr<HostResult>> &q, crgpu_out_batch *out) {
  out->length = 0;
  while (!q.empty() && q.front()->cursor >= q.front()->n) q.pop_front();
  if (q.empty()) return CRGPU_OK;
  CRGPU_TRY(q.front()->wait());
  bool emitted;
  return emit_rows(*q.front(), out, &emitted);
}

// dst ++= src (row-wise append of staged columns)
static void append_cols(std::vector<HostCol> &dst, const std::vector<HostCol> &src) {
  for (size_t c = 0; c < src.size(); c++) {
    HostCol &d = dst[c];
    const HostCol &s = src[c];
    if (s.type == CRGPU_BYTES) {
      const uint32_t base = static_cast<uint32_t>(d.data.size());
      for (int64_t i = 1; i <= s.n; i++) d.offsets.push_back(base + s.offsets[i]);
    }
    d.data.insert(d.data.end(), s.data.begin(), s.data.end());
    if (d.any_null || s.any_null) {
      d.fill_present();
      d.present.insert(d.present.end(), s.present.begin(), s.present.end());
      d.any_null = true;
    }
    d.n += s.n;
    if (d.any_null) d.fill_present();  // (s.present may have been shorter than s.n)
  }
}

// drop all but the newest `keep` rows
static void keep_newest(std::vector<HostCol> &cols, int64_t keep) {
  for (HostCol &c : cols) {
    if (keep >= c.n) continue;
    const int64_t drop = c.n - keep;
    if (c.type == CRGPU_BYTES) {
      const uint32_t base = c.offsets[static_cast<size_t>(drop)];
      c.data.erase(c.data.begin(), c.data.begin() + base);
      c.offsets.erase(c.offsets.begin(), c.offsets.begin() + drop);
      for (auto &o : c.offsets) o -= base;
    } else {
      c.data.erase(c.data.begin(), c.data.begin() + static_cast<size_t>(drop) * type_width(c.type));
    }
    if (!c.present.empty()) {
      const size_t d = std::min(c.present.size(), static_cast<size_t>(drop));
      c.present.erase(c.present.begin(), c.present.begin() + d);
    }
    c.n = keep;
  }
}

static void init_cols(std::vector<HostCol> &cols, const std::vector<int> &types) {
  cols.resize(types.size());
  for (size_t i = 0; i < types.size(); i++) cols[i].init(types[i]);
}
static void clear_cols(std::vector<HostCol> &cols) {
  for (auto &c : cols) c.clear();
}

}  // namespace crgpu

using namespace crgpu;

// ====================================================================== crgpu_hash_cols
extern "C" crgpu_status crgpu_hash_cols(const crgpu_col *cols, int32_t ncols, int64_t n, const int64_t *sel,
                                        uint64_t init, int32_t width, uint64_t nbuckets, void *out_host) {
  CRGPU_ENTER(nullptr);
  if (n < 0 || ncols < 0 || ncols > kMaxKeyCols) return fail(CRGPU_ERR_INVALID, "crgpu_hash_cols: bad n / ncols");
  if (width != 32 && width != 64) return fail(CRGPU_ERR_INVALID, "crgpu_hash_cols: width must be 32 or 64");
  if (n == 0) return CRGPU_OK;
  if (cols == nullptr || out_host == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_hash_cols: null argument");
  std::vector<int> types;
  for (int c = 0; c < ncols; c++) {
    if (!valid_type(cols[c].type)) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_hash_cols: unsupported column type");
    types.push_back(cols[c].type);
  }
  std::vector<HostCol> staged;
  init_cols(staged, types);
  crgpu_batch b{n, sel, ncols, 0, cols};
  CRGPU_TRY(stage_batch(staged, &b));
  DevTable t;
  CRGPU_TRY(upload_table(staged, t));
  DevBuf out;
  CRGPU_TRY(out.ensure(static_cast<size_t>(n) * (width / 8)));
  CRGPU_TRY(crgpu_dev_hash_cols(t.cols.data(), ncols, n, init, width, nbuckets, out.p));
  CRGPU_HIP(hipMemcpy(out_host, out.p, static_cast<size_t>(n) * (width / 8), hipMemcpyDeviceToHost));
  return CRGPU_OK;
}

// ====================================================================== hash joiner shell
struct crgpu_hj : crgpu::Shell {
  crgpu_join *dev = nullptr;
  std::vector<int> left_types, right_types;
  std::vector<HostCol> build_stage, probe_stage;
  DevTable build_dev, probe_dev;
  // Probe chunks of fixed-width columns are pipelined: a chunk is uploaded asynchronously when it is full and probed
  // only when the NEXT chunk is full (or output is asked for), so the transfer runs while the shell stages more batches;
  // its result comes back asynchronously too.  probe_inflight holds the host rows of the chunk whose upload is pending.
  std::vector<HostCol> probe_inflight;
  DevTable probe_dev2;
  int probe_cur = 0;       // which device table the uploaded chunk sits in
  bool deferred = false;   // an uploaded chunk has not been probed yet
  enum State { BUILDING, PROBING, DRAINING } state = BUILDING;
  std::deque<std::unique_ptr<HostResult>> outq;
  int64_t export_cursor = 0;
  bool build_on_host = true;
  std::unique_ptr<HostResult> export_copy;
};

extern "C" crgpu_status crgpu_hj_new(const crgpu_join_spec *spec, crgpu_hj **out) {
  if (spec == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_hj_new: null argument");
  crgpu_join *dev = nullptr;
  CRGPU_TRY(crgpu_dev_join_create(spec, &dev));
  auto h = new crgpu_hj();
  h->dev = dev;
  h->inner = reinterpret_cast<Handle *>(dev);
  for (int i = 0; i < spec->n_left_cols; i++) h->left_types.push_back(spec->left_types[i]);
  for (int i = 0; i < spec->n_right_cols; i++) h->right_types.push_back(spec->right_types[i]);
  init_cols(h->build_stage, h->right_types);
  init_cols(h->probe_stage, h->left_types);
  init_cols(h->probe_inflight, h->left_types);
  *out = h;
  return CRGPU_OK;
}

extern "C" void crgpu_hj_free(crgpu_hj *h) {
  if (h == nullptr) return;
  crgpu_dev_join_destroy(h->dev);
  delete h;
}

extern "C" crgpu_status crgpu_hj_reset(crgpu_hj *h) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_hj_reset: null handle");
  CRGPU_ENTER(h->inner);
  if (h->deferred) CRGPU_HIP(hipStreamSynchronize(stream()));  // an upload may still read probe_inflight
  h->deferred = false;
  CRGPU_TRY(crgpu_dev_join_reset(h->dev));
  clear_cols(h->build_stage);
  clear_cols(h->probe_stage);
  clear_cols(h->probe_inflight);
  h->outq.clear();
  h->state = crgpu_hj::BUILDING;
  h->export_cursor = 0;
  h->build_on_host = true;
  h->export_copy.reset();
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_hj_build_append(crgpu_hj *h, const crgpu_batch *right) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_hj_build_append: null handle");
  CRGPU_CHECK_CANCEL(h->inner);
  if (h->state != crgpu_hj::BUILDING) return fail(CRGPU_ERR_STATE, "crgpu_hj_build_append: the build side is already finished");
  return stage_batch(h->build_stage, right);
}

extern "C" crgpu_status crgpu_hj_build_finish(crgpu_hj *h) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_hj_build_finish: null handle");
  CRGPU_ENTER(h->inner);
  CRGPU_CHECK_CANCEL(h->inner);
  if (h->state != crgpu_hj::BUILDING) return fail(CRGPU_ERR_STATE, "crgpu_hj_build_finish: called twice");
  CRGPU_TRY(upload_table(h->build_stage, h->build_dev));
  CRGPU_TRY(crgpu_dev_join_build(h->dev, h->build_dev.cols.data(), h->build_dev.n));
  h->state = crgpu_hj::PROBING;
  return CRGPU_OK;
}

// async_result: the caller goes on staging, so the result may come back behind an event; a caller that is about to
// read the result takes the plain copies (an event per result costs ~50 us, which shows with small read-ahead)
static crgpu_status hj_probe_table(crgpu_hj *h, DevTable &t, bool async_result) {
  crgpu_dev_result r;
  CRGPU_TRY(crgpu_dev_join_probe(h->dev, t.cols.data(), t.n, &r));
  if (r.n > 0) {
    std::unique_ptr<HostResult> hr(new HostResult());
    if (async_result) CRGPU_TRY(download_result_async(r, *hr));
    else CRGPU_TRY(download_result(r, *hr));
    h->outq.push_back(std::move(hr));
  }
  return CRGPU_OK;
}

// drain: everything staged or uploaded is probed before the call returns (output was asked for, or the input ended)
static crgpu_status hj_flush_probe(crgpu_hj *h, bool drain) {
  if (h->deferred) {  // the chunk uploaded by the previous flush: its transfer ran while the next chunk was staged
    h->deferred = false;
    CRGPU_TRY(hj_probe_table(h, h->probe_cur ? h->probe_dev2 : h->probe_dev, !drain));
  }
  if (h->probe_stage.empty() || h->probe_stage[0].n == 0) return CRGPU_OK;
  if (h->build_on_host) {  // the host copy of the build side is only needed for ExportBuffered before probing starts
    clear_cols(h->build_stage);
    for (auto &c : h->build_stage) { c.data.shrink_to_fit(); c.present.shrink_to_fit(); c.offsets.shrink_to_fit(); }
    h->build_on_host = false;
  }
  if (!plain_fixed(h->probe_stage)) {
    CRGPU_TRY(upload_table(h->probe_stage, h->probe_dev));
    clear_cols(h->probe_stage);
    return hj_probe_table(h, h->probe_dev, false);
  }
  // (the other device table and probe_inflight are free: the chunk that used them has been probed, which waited for
  // its upload)
  h->probe_cur ^= 1;
  DevTable &t = h->probe_cur ? h->probe_dev2 : h->probe_dev;
  CRGPU_TRY(upload_table_async(h->probe_stage, t));
  std::swap(h->probe_stage, h->probe_inflight);
  clear_cols(h->probe_stage);
  h->deferred = true;
  if (drain) {
    h->deferred = false;
    CRGPU_TRY(hj_probe_table(h, t, false));
  }
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_hj_probe_append(crgpu_hj *h, const crgpu_batch *left) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_hj_probe_append: null handle");
  CRGPU_ENTER(h->inner);
  CRGPU_CHECK_CANCEL(h->inner);
  if (h->state != crgpu_hj::PROBING) return fail(CRGPU_ERR_STATE, "crgpu_hj_probe_append: build side not finished or probe side already finished");
  CRGPU_TRY(stage_batch(h->probe_stage, left));
  if (!h->probe_stage.empty() && h->probe_stage[0].n >= kFlushRows) CRGPU_TRY(hj_flush_probe(h, false));
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_hj_probe_finish(crgpu_hj *h) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_hj_probe_finish: null handle");
  CRGPU_ENTER(h->inner);
  CRGPU_CHECK_CANCEL(h->inner);
  if (h->state != crgpu_hj::PROBING) return fail(CRGPU_ERR_STATE, "crgpu_hj_probe_finish: not probing");
  CRGPU_TRY(hj_flush_probe(h, true));
  crgpu_dev_result r;
  CRGPU_TRY(crgpu_dev_join_emit_right(h->dev, &r));  // hjEmittingRight (hashjoiner.go:322-328)
  if (r.n > 0) {
    std::unique_ptr<HostResult> hr(new HostResult());
    CRGPU_TRY(download_result(r, *hr));
    h->outq.push_back(std::move(hr));
  }
  h->state = crgpu_hj::DRAINING;
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_hj_next_output(crgpu_hj *h, crgpu_out_batch *out) {
  if (h == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_hj_next_output: null argument");
  CRGPU_ENTER(h->inner);
  CRGPU_CHECK_CANCEL(h->inner);
  out->length = 0;
  while (!h->outq.empty() && h->outq.front()->cursor >= h->outq.front()->n) h->outq.pop_front();
  if (h->outq.empty() && h->state == crgpu_hj::PROBING) CRGPU_TRY(hj_flush_probe(h, true));
  return emit_from_queue(h->outq, out);
}

extern "C" crgpu_status crgpu_hj_export_buffered(crgpu_hj *h, crgpu_out_batch *out) {
  if (h == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_hj_export_buffered: null argument");
  CRGPU_ENTER(h->inner);
  out->length = 0;
  if (!h->export_copy) {
    h->export_copy.reset(new HostResult());
    if (h->build_on_host) {
      h->export_copy->cols = h->build_stage;  // copy of the staged build rows
      h->export_copy->n = h->build_stage.empty() ? 0 : h->build_stage[0].n;
    } else {
      crgpu_dev_result r;
      r.n = h->build_dev.n;
      r.ncols = static_cast<int32_t>(h->build_dev.cols.size());
      r._pad = 0;
      r.cols = h->build_dev.cols.data();
      CRGPU_TRY(download_result(r, *h->export_copy));
    }
    h->export_copy->cursor = 0;
  }
  if (h->export_copy->cursor >= h->export_copy->n) return CRGPU_OK;
  bool emitted;
  return emit_rows(*h->export_copy, out, &emitted);
}

// ====================================================================== hash aggregator shell
struct crgpu_ha : crgpu::Shell {
  crgpu_agg *dev = nullptr;
  std::vector<int> types;
  std::vector<HostCol> stage;
  DevTable staged_dev;
  bool finished = false;
  std::deque<std::unique_ptr<HostResult>> outq;
  // hashAggregator.inputTrackingState (hash_aggregator.go:138-153): when the planner wraps the aggregator in a disk
  // spiller, every input tuple is kept so that ExportBuffered can replay the whole input to the disk-backed fallback
  bool track = false;
  std::vector<HostCol> tracked;
  std::unique_ptr<HostResult> export_copy;
  bool released = false;
  bool ordered = false;  // AggregatorSpec.OrderedGroupCols: completed groups are emitted while the input is still read
};

extern "C" crgpu_status crgpu_agg_new(const crgpu_agg_spec *spec, crgpu_ha **out) {
  if (spec == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_agg_new: null argument");
  crgpu_agg *dev = nullptr;
  CRGPU_TRY(crgpu_dev_agg_create(spec, &dev));
  auto h = new crgpu_ha();
  h->dev = dev;
  h->inner = reinterpret_cast<Handle *>(dev);
  h->ordered = spec->n_ordered > 0;
  for (int i = 0; i < spec->n_cols; i++) h->types.push_back(spec->types[i]);
  init_cols(h->stage, h->types);
  *out = h;
  return CRGPU_OK;
}

// partially ordered input: the pieces the device aggregator has completed so far (crgpu_dev_agg_emit_ready) -> outq
static crgpu_status ha_pull_ready(crgpu_ha *h) {
  for (;;) {
    crgpu_dev_result r;
    CRGPU_TRY(crgpu_dev_agg_emit_ready(h->dev, &r));
    if (r.n == 0) return CRGPU_OK;
    std::unique_ptr<HostResult> hr(new HostResult());
    CRGPU_TRY(download_result(r, *hr));
    h->outq.push_back(std::move(hr));
  }
}

extern "C" void crgpu_agg_free(crgpu_ha *h) {
  if (h == nullptr) return;
  crgpu_dev_agg_destroy(h->dev);
  delete h;
}

extern "C" crgpu_status crgpu_agg_reset(crgpu_ha *h) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_agg_reset: null handle");
  CRGPU_ENTER(h->inner);
  CRGPU_TRY(crgpu_dev_agg_reset(h->dev));
  clear_cols(h->stage);
  clear_cols(h->tracked);
  h->export_copy.reset();
  h->released = false;
  h->outq.clear();
  h->finished = false;
  return CRGPU_OK;
}

static crgpu_status ha_flush(crgpu_ha *h) {
  const int64_t n = h->stage.empty() ? 0 : h->stage[0].n;
  if (n == 0) return CRGPU_OK;
  if (h->track) append_cols(h->tracked, h->stage);  // before the device call: a failing call (OOM) must not lose them
  if (h->released) {
    clear_cols(h->stage);
    return fail(CRGPU_ERR_STATE, "crgpu_agg: the device state was released for export");
  }
  crgpu_status st = upload_table(h->stage, h->staged_dev);
  if (getenv("CRGPU_TEST_FAIL_AGG_UPLOAD") != nullptr && st == CRGPU_OK) st = fail(CRGPU_ERR_OOM, "crgpu_agg: injected upload failure (CRGPU_TEST_FAIL_AGG_UPLOAD)");
  if (st != CRGPU_OK) {
    // With tracking on the rows are in `tracked` already: the stage is cleared, or crgpu_agg_export_buffered would
    // append them to `tracked` a second time and hand them out twice.  Without tracking the stage keeps them.
    if (h->track) clear_cols(h->stage);
    return st;
  }
  st = crgpu_dev_agg_consume(h->dev, h->staged_dev.cols.data(), n);
  clear_cols(h->stage);  // (tracked rows keep the batch when tracking is on)
  if (st == CRGPU_OK && h->ordered) st = ha_pull_ready(h);
  if (st == CRGPU_OK && h->ordered && h->track) {
    // the groups of completed runs are out: only the rows behind the last change of the ordered columns are still
    // needed for ExportBuffered (hashAggregator.resetBucketsAndTrackingState, hash_aggregator_tmpl.go:355-378)
    int64_t pending = 0;
    st = crgpu_dev_agg_pending_rows(h->dev, &pending);
    if (st == CRGPU_OK) keep_newest(h->tracked, pending);
  }
  return st;
}

// colexecop.BufferingInMemoryOperator for the aggregator (hash_aggregator.go:455-500)
extern "C" crgpu_status crgpu_agg_track_input(crgpu_ha *h, int32_t on) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_agg_track_input: null handle");
  if (!h->stage.empty() && (h->stage[0].n > 0 || (!h->tracked.empty() && h->tracked[0].n > 0)))
    return fail(CRGPU_ERR_STATE, "crgpu_agg_track_input: call before the first append");
  h->track = on != 0;
  if (h->track) init_cols(h->tracked, h->types);
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_agg_export_buffered(crgpu_ha *h, crgpu_out_batch *out) {
  if (h == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_agg_export_buffered: null argument");
  out->length = 0;
  if (!h->track) return fail(CRGPU_ERR_STATE, "crgpu_agg_export_buffered: input tracking was not enabled (crgpu_agg_track_input)");
  if (!h->export_copy) {
    // everything consumed so far plus what is still staged: the whole input, in arrival order
    if (!h->stage.empty() && h->stage[0].n > 0) {
      append_cols(h->tracked, h->stage);
      clear_cols(h->stage);
    }
    h->export_copy.reset(new HostResult());
    h->export_copy->cols = std::move(h->tracked);
    h->export_copy->n = h->export_copy->cols.empty() ? 0 : h->export_copy->cols[0].n;
    h->export_copy->cursor = 0;
    init_cols(h->tracked, h->types);
  }
  if (h->export_copy->cursor >= h->export_copy->n) return CRGPU_OK;  // zero-length batch: all tuples have been exported
  bool emitted;
  return emit_rows(*h->export_copy, out, &emitted);
}

extern "C" crgpu_status crgpu_agg_release_before_export(crgpu_ha *h) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_agg_release_before_export: null handle");
  CRGPU_ENTER(h->inner);
  // the hash table and the accumulators go back to the pool; only the tracked input is kept (hash_aggregator.go:475-492)
  CRGPU_TRY(crgpu_dev_agg_reset(h->dev));
  h->staged_dev = DevTable();
  h->outq.clear();
  h->released = true;
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_agg_release_after_export(crgpu_ha *h) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_agg_release_after_export: null handle");
  h->export_copy.reset();
  clear_cols(h->tracked);
  for (auto &c : h->tracked) { c.data.shrink_to_fit(); c.present.shrink_to_fit(); c.offsets.shrink_to_fit(); }
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_agg_append(crgpu_ha *h, const crgpu_batch *in) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_agg_append: null handle");
  CRGPU_ENTER(h->inner);
  CRGPU_CHECK_CANCEL(h->inner);
  if (h->finished) return fail(CRGPU_ERR_STATE, "crgpu_agg_append: input already finished");
  if (h->types.empty()) {
    // no input columns at all (COUNT_ROWS only): nothing to stage but the row count matters
    return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_agg_append: aggregation without input columns");
  }
  // Ordered grouping columns: completed groups are read (crgpu_agg_next_output) before more input goes in, as the
  // reference's state machine does (hashAggregatorOutputting runs dry before hashAggregatorBuffering resumes); the
  // tracked input can then never hold rows of groups that were queued for output but not handed out.
  if (h->ordered) {
    while (!h->outq.empty() && h->outq.front()->cursor >= h->outq.front()->n) h->outq.pop_front();
    if (!h->outq.empty()) return fail(CRGPU_ERR_STATE, "crgpu_agg_append: ordered aggregation has completed groups to read first (crgpu_agg_next_output until length 0)");
  }
  CRGPU_TRY(stage_batch(h->stage, in));
  if (h->stage[0].n >= kFlushRows) CRGPU_TRY(ha_flush(h));
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_agg_finish(crgpu_ha *h) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_agg_finish: null handle");
  CRGPU_ENTER(h->inner);
  CRGPU_CHECK_CANCEL(h->inner);
  if (h->finished) return fail(CRGPU_ERR_STATE, "crgpu_agg_finish: called twice");
  CRGPU_TRY(ha_flush(h));
  crgpu_dev_result r;
  CRGPU_TRY(crgpu_dev_agg_finish(h->dev, &r));
  if (r.n > 0) {
    std::unique_ptr<HostResult> hr(new HostResult());
    CRGPU_TRY(download_result(r, *hr));
    h->outq.push_back(std::move(hr));
  }
  if (h->ordered) CRGPU_TRY(ha_pull_ready(h));
  h->finished = true;
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_agg_next_output(crgpu_ha *h, crgpu_out_batch *out) {
  if (h == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_agg_next_output: null argument");
  CRGPU_ENTER(h->inner);
  CRGPU_CHECK_CANCEL(h->inner);
  out->length = 0;
  // the hash aggregator emits only after its input is exhausted -- unless the input is ordered on some grouping
  // columns: then the groups of every completed run are there already (hash_aggregator_tmpl.go:256-306)
  if (!h->finished && !h->ordered) return CRGPU_OK;
  return emit_from_queue(h->outq, out);
}

// ====================================================================== unordered distinct shell
struct crgpu_ud : crgpu::Shell {
  crgpu_distinct *dev = nullptr;
  std::vector<int> types;
  std::vector<HostCol> stage;
  DevTable staged_dev;
  bool finished = false;
  std::deque<std::unique_ptr<HostResult>> outq;
};

extern "C" crgpu_status crgpu_distinct_new(const int32_t *types, int32_t n_cols, const int32_t *distinct_cols,
                                           int32_t n_distinct, int32_t nulls_are_distinct, int32_t error_on_dup,
                                           crgpu_ud **out) {
  if (out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_distinct_new: null argument");
  crgpu_distinct *dev = nullptr;
  CRGPU_TRY(crgpu_dev_distinct_create(types, n_cols, distinct_cols, n_distinct, nulls_are_distinct, error_on_dup, &dev));
  auto h = new crgpu_ud();
  h->dev = dev;
  h->inner = reinterpret_cast<Handle *>(dev);
  for (int i = 0; i < n_cols; i++) h->types.push_back(types[i]);
  init_cols(h->stage, h->types);
  *out = h;
  return CRGPU_OK;
}

extern "C" void crgpu_distinct_free(crgpu_ud *h) {
  if (h == nullptr) return;
  crgpu_dev_distinct_destroy(h->dev);
  delete h;
}

extern "C" crgpu_status crgpu_distinct_reset(crgpu_ud *h) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_distinct_reset: null handle");
  CRGPU_ENTER(h->inner);
  CRGPU_TRY(crgpu_dev_distinct_reset(h->dev));
  clear_cols(h->stage);
  h->outq.clear();
  h->finished = false;
  return CRGPU_OK;
}

// staged rows -> device -> the new tuples come back as one host result (input order is kept end to end)
static crgpu_status ud_flush(crgpu_ud *h) {
  const int64_t n = h->stage.empty() ? 0 : h->stage[0].n;
  if (n == 0) return CRGPU_OK;
  CRGPU_TRY(upload_table(h->stage, h->staged_dev));
  clear_cols(h->stage);
  crgpu_dev_result r;
  const crgpu_status st = crgpu_dev_distinct_consume(h->dev, h->staged_dev.cols.data(), n, &r);
  if (st != CRGPU_OK && st != CRGPU_ERR_DUPLICATE) return st;
  if (r.n > 0) {
    std::unique_ptr<HostResult> hr(new HostResult());
    CRGPU_TRY(download_result(r, *hr));
    h->outq.push_back(std::move(hr));
  }
  return st;  // CRGPU_ERR_DUPLICATE: errorOnDup plans raise their message; the output is still queued
}

extern "C" crgpu_status crgpu_distinct_append(crgpu_ud *h, const crgpu_batch *in) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_distinct_append: null handle");
  CRGPU_ENTER(h->inner);
  CRGPU_CHECK_CANCEL(h->inner);
  if (h->finished) return fail(CRGPU_ERR_STATE, "crgpu_distinct_append: input already finished");
  CRGPU_TRY(stage_batch(h->stage, in));
  if (!h->stage.empty() && h->stage[0].n >= kFlushRows) CRGPU_TRY(ud_flush(h));
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_distinct_finish(crgpu_ud *h) {
  if (h == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_distinct_finish: null handle");
  CRGPU_ENTER(h->inner);
  CRGPU_CHECK_CANCEL(h->inner);
  h->finished = true;
  return ud_flush(h);
}

extern "C" crgpu_status crgpu_distinct_next_output(crgpu_ud *h, crgpu_out_batch *out) {
  if (h == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_distinct_next_output: null argument");
  CRGPU_ENTER(h->inner);
  CRGPU_CHECK_CANCEL(h->inner);
  out->length = 0;
  while (!h->outq.empty() && h->outq.front()->cursor >= h->outq.front()->n) h->outq.pop_front();
  // the operator streams (unordered_distinct.go:113-119): nothing queued => push what is staged through the device
  if (h->outq.empty()) CRGPU_TRY(ud_flush(h));
  return emit_from_queue(h->outq, out);
}

// colexecop.BufferingInMemoryOperator for unordered DISTINCT (unordered_distinct.go:123-150): only the tuples of the last
// input that are new and not yet emitted are "buffered" -- the hash table itself stays, because the external distinct
// keeps filtering against it (unorderedDistinctFilterer) -- so ReleaseBeforeExport / ReleaseAfterExport are no-ops there
// and have no entry point here.  What is staged is first pushed through the device, then the unemitted new tuples are
// handed out in input order.
extern "C" crgpu_status crgpu_distinct_export_buffered(crgpu_ud *h, crgpu_out_batch *out) {
  if (h == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_distinct_export_buffered: null argument");
  CRGPU_ENTER(h->inner);
  out->length = 0;
  while (!h->outq.empty() && h->outq.front()->cursor >= h->outq.front()->n) h->outq.pop_front();
  if (h->outq.empty()) {
    const crgpu_status st = ud_flush(h);
    if (st != CRGPU_OK && st != CRGPU_ERR_DUPLICATE) return st;
  }
  return emit_from_queue(h->outq, out);
}

// ====================================================================== hash router shell
struct crgpu_router : crgpu::Shell {
  crgpu_partitioner *dev = nullptr;
  std::vector<int> types;
  int num_outputs = 0;
  std::vector<HostCol> stage;
  DevTable staged_dev;
  std::vector<std::deque<std::unique_ptr<HostResult>>> outq;
  std::mutex mu;  // Run() and every output's Next() live on different goroutines (routers.go:486-488)
};

extern "C" crgpu_status crgpu_route_new(const int32_t *types, int32_t n_cols, const int32_t *hash_cols, int32_t n_hash,
                                        int32_t num_outputs, crgpu_router **out) {
  if (out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_route_new: null argument");
  crgpu_partitioner *dev = nullptr;
  // HashRouter uses the default init hash value 1 (routers.go:557)
  CRGPU_TRY(crgpu_dev_partitioner_create(types, n_cols, hash_cols, n_hash, 1, num_outputs, &dev));
  auto r = new crgpu_router();
  r->dev = dev;
  r->inner = reinterpret_cast<Handle *>(dev);
  r->num_outputs = num_outputs;
  for (int i = 0; i < n_cols; i++) r->types.push_back(types[i]);
  init_cols(r->stage, r->types);
  r->outq.resize(num_outputs);
  *out = r;
  return CRGPU_OK;
}

extern "C" void crgpu_route_free(crgpu_router *r) {
  if (r == nullptr) return;
  crgpu_dev_partitioner_destroy(r->dev);
  delete r;
}

static crgpu_status route_flush(crgpu_router *r) {
  const int64_t n = r->stage.empty() ? 0 : r->stage[0].n;
  if (n == 0) return CRGPU_OK;
  CRGPU_TRY(upload_table(r->stage, r->staged_dev));
  crgpu_dev_result res;
  std::vector<int64_t> counts(r->num_outputs, 0);
  CRGPU_TRY(crgpu_dev_partition(r->dev, r->staged_dev.cols.data(), n, &res, counts.data()));
  clear_cols(r->stage);
  HostResult all;
  CRGPU_TRY(download_result(res, all));
  int64_t start = 0;
  for (int p = 0; p < r->num_outputs; p++) {
    const int64_t k = counts[p];
    if (k > 0) {
      std::unique_ptr<HostResult> hr(new HostResult());
      hr->n = k;
      hr->cols.resize(all.cols.size());
      for (size_t c = 0; c < all.cols.size(); c++) {
        const HostCol &s = all.cols[c];
        HostCol &d = hr->cols[c];
        d.init(s.type);
        d.n = k;
        if (s.any_null) {  // (conservative: this slice may hold none of the NULLs)
          d.present.assign(s.present.begin() + start, s.present.begin() + start + k);
          d.any_null = true;
        }
        if (s.type == CRGPU_BYTES) {
          const uint32_t a = s.offsets[start], b = s.offsets[start + k];
          d.data.assign(s.data.begin() + a, s.data.begin() + b);
          d.offsets.resize(static_cast<size_t>(k + 1));
          for (int64_t i = 0; i <= k; i++) d.offsets[i] = s.offsets[start + i] - a;
        } else {
          const int w = type_width(s.type);
          d.data.assign(s.data.begin() + start * w, s.data.begin() + (start + k) * w);
        }
      }
      r->outq[p].push_back(std::move(hr));
    }
    start += k;
  }
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_route_partition(crgpu_router *r, const crgpu_batch *in) {
  if (r == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_route_partition: null handle");
  std::lock_guard<std::mutex> lk(r->mu);
  CRGPU_ENTER(r->inner);
  CRGPU_CHECK_CANCEL(r->inner);
  CRGPU_TRY(stage_batch(r->stage, in));
  if (!r->stage.empty() && r->stage[0].n >= kFlushRows) CRGPU_TRY(route_flush(r));
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_route_finish(crgpu_router *r) {
  if (r == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_route_finish: null handle");
  std::lock_guard<std::mutex> lk(r->mu);
  CRGPU_ENTER(r->inner);
  CRGPU_CHECK_CANCEL(r->inner);
  return route_flush(r);
}

extern "C" crgpu_status crgpu_route_next_output(crgpu_router *r, int32_t output, crgpu_out_batch *out) {
  if (r == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_route_next_output: null argument");
  if (output < 0 || output >= r->num_outputs) return fail(CRGPU_ERR_INVALID, "crgpu_route_next_output: output index out of range");
  std::lock_guard<std::mutex> lk(r->mu);
  CRGPU_ENTER(r->inner);
  CRGPU_CHECK_CANCEL(r->inner);
  out->length = 0;
  auto &q = r->outq[output];
  while (!q.empty() && q.front()->cursor >= q.front()->n) q.pop_front();
  if (q.empty()) CRGPU_TRY(route_flush(r));  // nothing buffered for this output: push what is staged through the device
  return emit_from_queue(q, out);
}
