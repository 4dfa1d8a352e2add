// common.h -- internal declarations shared by the HIP translation units of libcrgpu.
// gfx950 (MI355X, CDNA4) only: 64-wide wavefronts, 256 CUs in 8 XCDs, 160 KB LDS/CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <string>
#include <vector>

#include "crgpu.h"

namespace crgpu {

// ------------------------------------------------------------------ errors
void set_error(const std::string &msg);
crgpu_status fail(crgpu_status code, const std::string &msg);
crgpu_status hip_fail(hipError_t e, const char *what, const char *file, int line);

#define CRGPU_HIP(expr)                                                              \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) return ::crgpu::hip_fail(_e, #expr, __FILE__, __LINE__);   \
  } while (0)
#define CRGPU_TRY(expr)                          \
  do {                                           \
    crgpu_status _s = (expr);                    \
    if (_s != CRGPU_OK) return _s;               \
  } while (0)

// ------------------------------------------------------------------ runtime
// One library context per HIP device (crgpu_init): a default stream for the handle-less calls, the CU count, the
// no-allocation scratch pointer.  Every OS thread has a "library device" (the last crgpu_init on that thread, else the
// first device the process initialised); every handle remembers the device it was created on and owns a HIP stream, so
// two operator chains on two goroutines do not serialise on each other and one process can drive several GPUs.
static constexpr int kMaxDevices = 16;
hipStream_t stream();        // the calling thread's current stream: the handle's inside a handle call, else the device's default stream
crgpu_status ensure_init();  // binds the calling thread to its library device (lazily crgpu_init); cheap when already bound
int num_cus();
int current_device();        // the calling thread's library device (-1 before any init)
void mark_default_work();    // a handle-less call queued asynchronous work on the default stream: handle calls order themselves after it
crgpu_status noalloc_setup_device(int device);  // partition.hip: resolves this device's copy of the static scratch array (called by crgpu_init with the device current)

// "Once per device" latch for per-device HIP state set lazily (hipFuncSetAttribute applies to the current device only;
// one process may drive several GPUs).  Two threads racing through need()/mark() both do the idempotent work.
struct DeviceOnce {
  std::atomic<uint32_t> mask{0};
  bool need() const {
    const int d = current_device();
    return d < 0 || d >= 32 || !((mask.load(std::memory_order_acquire) >> d) & 1u);
  }
  void mark() {
    const int d = current_device();
    if (d >= 0 && d < 32) mask.fetch_or(1u << d, std::memory_order_release);
  }
};

static constexpr uint32_t kHandleMagic = 0x43524748u;  // "CRGH"
// First base of every handle struct (crgpu_join, crgpu_agg, ...): crgpu_cancel() accepts any of them.
struct Handle {
  uint32_t magic = kHandleMagic;
  int device = -1;
  hipStream_t st = nullptr;
  hipEvent_t ev = nullptr;      // orders st after the work queued so far on the device's default stream
  uint64_t seen_epoch = 0;
  std::atomic<int32_t> cancelled{0};  // set by crgpu_cancel from any thread; sticky until reset
  // bytes this handle's calls hold: device buffers (DevBuf) and pinned host staging, for the caller's per-operator
  // memory account (crgpu_handle_mem_usage; colmem.Allocator in the reference)
  std::atomic<int64_t> mem_dev{0}, mem_pinned{0};
  // per-operator execution statistics (crgpu_handle_stats; execinfrapb.ComponentStats in the reference): calls that named
  // this handle outermost and the wall time inside them, rows handed in and rows produced
  std::atomic<int64_t> st_calls{0}, st_ns{0}, st_rows_in{0}, st_rows_out{0};
  Handle() = default;
  Handle(const Handle &) = delete;
  Handle &operator=(const Handle &) = delete;
  crgpu_status open();  // binds to the calling thread's library device and creates the stream
  void close();
  ~Handle() { close(); }
};
// First base of the layer-2 shells (crgpu_hj, crgpu_ha, crgpu_ud, crgpu_router): they run on their device handle's
// stream and are cancelled through it.
static constexpr uint32_t kShellMagic = 0x43524753u;  // "CRGS"
struct Shell {
  uint32_t magic = kShellMagic;
  Handle *inner = nullptr;
  ~Shell() { magic = 0; }
};
// RAII for one extern "C" call: binds the thread to the handle's device (or, without a handle, to its library device),
// makes the handle's stream the thread's current stream() and restores both on exit.  Nests (layer 2 calls layer 1).
struct CallScope {
  hipStream_t prev_stream;
  Handle *prev_handle;
  Handle *timed = nullptr;  // the handle this scope made current (outermost call): its time is taken at exit
  int64_t t0_ns = 0;
  int prev_device;
  bool prev_in_call;
  crgpu_status st;
  explicit CallScope(Handle *h);
  ~CallScope();
};
#define CRGPU_ENTER(h)            \
  ::crgpu::CallScope _scope(h);   \
  if (_scope.st != CRGPU_OK) return _scope.st
#define CRGPU_CHECK_CANCEL(h)                                                                         \
  do {                                                                                                \
    if ((h)->cancelled.load(std::memory_order_relaxed)) return ::crgpu::fail(CRGPU_ERR_CANCELLED, "operation cancelled (crgpu_cancel)"); \
  } while (0)

Handle *current_handle();  // the handle of the outermost extern "C" call on this thread (nullptr: a handle-less call)
void account_mem(Handle *owner, int64_t device_delta, int64_t pinned_delta);  // no-op for a handle that is gone
inline void note_rows(Handle *h, int64_t rows_in, int64_t rows_out) {
  h->st_rows_in.fetch_add(rows_in, std::memory_order_relaxed);
  h->st_rows_out.fetch_add(rows_out, std::memory_order_relaxed);
}
// Grow-only device buffer; every byte is accounted in crgpu_mem_usage() and in its owner's mem_dev.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int device = -1;  // the device the block lives on (its pool)
  Handle *owner = nullptr;  // the handle whose call allocated the block (accounting only; checked against the live set)
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  // Ensures capacity >= bytes.  Contents are NOT preserved on growth unless keep=true.
  crgpu_status ensure(size_t bytes, bool keep = false);
  void release();
  template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// HIP-event timing per kernel family (crgpu_prof_*).
struct ProfScope {
  const char *name;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool on;
  explicit ProfScope(const char *n);
  ~ProfScope();
};

// ------------------------------------------------------------------ device-side column model
static constexpr int kMaxKeyCols = 8;
static constexpr int kMaxCols = 64;

struct DCol {  // POD mirror of crgpu_dcol passed to kernels by value
  int32_t type;
  int32_t _pad;
  const void *data;
  const uint8_t *valid;
  const uint32_t *offsets;
};
static inline DCol to_dcol(const crgpu_dcol &c) { return DCol{c.type, 0, c.data, c.valid, c.offsets}; }

struct KeyCols {
  int32_t n;
  int32_t _pad;
  DCol c[kMaxKeyCols];
};

static inline int type_width(int t) {
  switch (t) {
    case CRGPU_BOOL: return 1;
    case CRGPU_INT16: return 2;
    case CRGPU_INT32: return 4;
    case CRGPU_INT64: return 8;
    case CRGPU_FLOAT64: return 8;
    default: return 0;
  }
}
static inline bool is_int_type(int t) { return t == CRGPU_INT16 || t == CRGPU_INT32 || t == CRGPU_INT64; }
static inline bool valid_type(int t) { return t >= CRGPU_BOOL && t <= CRGPU_BYTES; }

// An output column owned by the library: data / validity-bytes / offsets buffers.
struct OutCol {
  int type = 0;
  DevBuf data;     // fixed-width values or BYTES payload
  DevBuf valid;    // validity bitmap (bit = 1 present); only meaningful if nullable
  DevBuf offsets;  // BYTES: n+1 uint32
  bool nullable = false;
  crgpu_dcol view() const {
    crgpu_dcol d;
    d.type = type;
    d._pad = 0;
    d.data = data.p;
    d.valid = nullable ? valid.as<uint8_t>() : nullptr;
    d.offsets = type == CRGPU_BYTES ? offsets.as<uint32_t>() : nullptr;
    return d;
  }
};

// ------------------------------------------------------------------ shared launch helpers (gather.hip)
// dst[i] = src[idx[i]] for fixed-width columns; idx < 0 (as int32) => NULL row (zero value).
crgpu_status gather_fixed(const DCol &src, const int32_t *idx, int64_t n, void *dst);
// Validity of gathered rows: out bitmap bit i = src valid at idx[i] (idx<0 => 0). Returns via *any_null_dev (uint32 flag) if non-null.
crgpu_status gather_valid(const DCol &src, const int32_t *idx, int64_t n, uint8_t *dst_bitmap);
// BYTES gather: computes lengths, scans into dst_offsets[n+1], ensures dst_data capacity, copies payloads.
crgpu_status gather_bytes(const DCol &src, const int32_t *idx, int64_t n, DevBuf &dst_offsets, DevBuf &dst_data,
                          DevBuf &scratch);
// Materialise column `src` through row ids `idx` into `out` (all families, validity included when
// `force_nullable` or the source has a validity bitmap).
crgpu_status materialize(const DCol &src, const int32_t *idx, int64_t n, bool force_nullable, OutCol &out,
                         DevBuf &scratch);
// Exclusive prefix sum of uint32 counts into uint64 offsets[n+1] (offsets[n] = total); returns total on host.
crgpu_status exclusive_scan_u32(const uint32_t *counts, int64_t n, uint64_t *offsets, DevBuf &scratch,
                                uint64_t *total_host);
// Same with caller-provided tile-sum scratch of scan_scratch_words(n) uint64 words (allocates nothing).
size_t scan_scratch_words(int64_t n);
crgpu_status exclusive_scan_u32_ws(const uint32_t *counts, int64_t n, uint64_t *offsets, uint64_t *tile_sums,
                                   uint64_t *total_host);
// idx_out[k] = i for every i in [0,n) with flags[i] != 0 (stable order); *count_host = number selected.
// skip_if_all: when every row is flagged idx_out is NOT written (the caller copies instead of gathering).
crgpu_status select_flagged(const uint8_t *flags, int64_t n, int32_t *idx_out, DevBuf &scratch,
                            int64_t *count_host, bool skip_if_all = false);
// bytes-per-row -> bitmap packing (dst bit i = src[i] != 0)
crgpu_status pack_bitmap(const uint8_t *bytes, int64_t n, uint8_t *bitmap);
// hash rows (hash.hip): out32/out64 device arrays
crgpu_status launch_hash(const KeyCols &keys, int64_t n, uint64_t init, int width, uint64_t nbuckets, void *out);

}  // namespace crgpu
