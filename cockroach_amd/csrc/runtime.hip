// runtime.hip -- library lifecycle, per-device contexts, per-handle streams, error reporting, memory accounting,
// HIP-event profiling, cancellation.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_set>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace crgpu {

struct DeviceCtx {
  std::atomic<bool> ready{false};
  hipStream_t stream = nullptr;  // default stream: handle-less calls (hash, select, project, raw copies)
  int cus = 256;
  std::atomic<uint64_t> epoch{0};  // bumped by every asynchronous handle-less call
  // streams and events of closed handles, reused by the next handles: a query plan opens and closes a dozen operators,
  // and creating a stream per operator costs tens of microseconds each (guarded by g_mu)
  std::vector<std::pair<hipStream_t, hipEvent_t>> idle;
};
static constexpr size_t kMaxIdleStreams = 64;

static thread_local std::string t_err;
static thread_local int t_device = -1;            // this thread's library device
static thread_local hipStream_t t_stream = nullptr;  // current handle stream inside a handle call
static thread_local bool t_in_call = false;
static thread_local Handle *t_handle = nullptr;   // handle of the outermost call in progress
static std::mutex g_live_mu;
static std::unordered_set<Handle *> g_live;        // handles between open() and close()
static std::atomic<int64_t> g_mem{0};
static std::atomic<int64_t> g_handles{0};
static std::atomic<int> g_default_device{-1};
static DeviceCtx g_dev[kMaxDevices];
static std::mutex g_mu;

void set_error(const std::string &msg) { t_err = msg; }
crgpu_status fail(crgpu_status code, const std::string &msg) {
  t_err = msg;
  return code;
}
crgpu_status hip_fail(hipError_t e, const char *what, const char *file, int line) {
  char buf[512];
  snprintf(buf, sizeof(buf), "HIP error %d (%s) in %s at %s:%d", static_cast<int>(e), hipGetErrorString(e), what,
           file, line);
  t_err = buf;
  return e == hipErrorOutOfMemory ? CRGPU_ERR_OOM : CRGPU_ERR_HIP;
}

int current_device() { return t_device; }
hipStream_t stream() {
  if (t_stream != nullptr) return t_stream;
  return t_device >= 0 ? g_dev[t_device].stream : nullptr;
}
int num_cus() { return t_device >= 0 ? g_dev[t_device].cus : 256; }
void mark_default_work() {
  if (t_device >= 0) g_dev[t_device].epoch.fetch_add(1, std::memory_order_release);
}

// hipSetDevice only affects the calling OS thread, and cgo calls arrive on arbitrary threads: every entry point
// re-binds (one hipGetDevice when nothing changed).
static crgpu_status bind_thread(int device) {
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device) CRGPU_HIP(hipSetDevice(device));
  t_device = device;
  return CRGPU_OK;
}

static crgpu_status init_device(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(CRGPU_ERR_NO_DEVICE, "crgpu_init: no HIP device visible (the GPU path has no CPU fallback)");
  if (device < 0 || device >= count || device >= kMaxDevices)
    return fail(CRGPU_ERR_INVALID, "crgpu_init: device ordinal out of range");
  CRGPU_TRY(bind_thread(device));
  DeviceCtx &d = g_dev[device];
  if (d.ready.load(std::memory_order_acquire)) return CRGPU_OK;
  std::lock_guard<std::mutex> lk(g_mu);
  if (d.ready.load(std::memory_order_acquire)) return CRGPU_OK;
  hipDeviceProp_t prop;
  CRGPU_HIP(hipGetDeviceProperties(&prop, device));
  d.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  CRGPU_HIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
  CRGPU_TRY(noalloc_setup_device(device));
  int none = -1;
  g_default_device.compare_exchange_strong(none, device);
  d.ready.store(true, std::memory_order_release);
  return CRGPU_OK;
}

crgpu_status ensure_init() {
  int device = t_device;
  if (device < 0) device = g_default_device.load();
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  if (!g_dev[device < kMaxDevices && device >= 0 ? device : 0].ready.load(std::memory_order_acquire)) return init_device(device);
  return bind_thread(device);
}

crgpu_status Handle::open() {
  CRGPU_TRY(ensure_init());
  device = t_device;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto &idle = g_dev[device].idle;
    if (!idle.empty()) {
      st = idle.back().first;
      ev = idle.back().second;
      idle.pop_back();
    }
  }
  if (st == nullptr) {
    CRGPU_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CRGPU_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  g_handles.fetch_add(1);
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.insert(this);
  }
  return CRGPU_OK;
}
void Handle::close() {
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.erase(this);
  }
  if (st != nullptr) {
    (void)hipStreamSynchronize(st);
    bool kept = false;
    if (ev != nullptr && device >= 0 && device < kMaxDevices) {
      std::lock_guard<std::mutex> lk(g_mu);
      auto &idle = g_dev[device].idle;
      if (g_dev[device].ready.load() && idle.size() < kMaxIdleStreams) {
        idle.emplace_back(st, ev);  // drained above: the next handle starts on an empty stream
        kept = true;
      }
    }
    if (kept) {
      ev = nullptr;
    } else {
      (void)hipStreamDestroy(st);
    }
    st = nullptr;
    g_handles.fetch_sub(1);
  }
  if (ev != nullptr) {
    (void)hipEventDestroy(ev);
    ev = nullptr;
  }
  magic = 0;
}

CallScope::CallScope(Handle *h) : prev_stream(t_stream), prev_handle(t_handle), prev_device(t_device), prev_in_call(t_in_call), st(CRGPU_OK) {
  if (h == nullptr) {
    if (!t_in_call) {  // a handle-less call nested in a handle call keeps the handle's stream
      st = ensure_init();
      t_stream = nullptr;
      t_handle = nullptr;
    }
    t_in_call = true;
    return;
  }
  if (h->magic != kHandleMagic || h->st == nullptr) {
    st = fail(CRGPU_ERR_INVALID, "invalid or destroyed handle");
    return;
  }
  if (!g_dev[h->device].ready.load(std::memory_order_acquire)) {
    st = fail(CRGPU_ERR_STATE, "the handle's device was shut down (crgpu_shutdown)");
    return;
  }
  st = bind_thread(h->device);
  if (st != CRGPU_OK) return;
  if (!t_in_call || t_handle == nullptr) {  // memory and time are accounted to the outermost handle of a call
    t_handle = h;
    timed = h;
    t0_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  if (!t_in_call || t_stream == nullptr) {
    t_stream = h->st;
    // results of handle-less calls queued on the default stream (projections, selections, hashes) feed handle calls
    DeviceCtx &d = g_dev[h->device];
    const uint64_t ep = d.epoch.load(std::memory_order_acquire);
    if (ep != h->seen_epoch) {
      hipError_t e = hipEventRecord(h->ev, d.stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(h->st, h->ev, 0);
      if (e != hipSuccess) {
        st = hip_fail(e, "order after the default stream", __FILE__, __LINE__);
        return;
      }
      h->seen_epoch = ep;
    }
  }  // else: a layer-2 shell already made ITS device handle current; nested device handles run on that stream
  t_in_call = true;
}
CallScope::~CallScope() {
  if (timed != nullptr) {
    const int64_t t1 = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    timed->st_ns.fetch_add(t1 - t0_ns, std::memory_order_relaxed);
    timed->st_calls.fetch_add(1, std::memory_order_relaxed);
  }
  t_stream = prev_stream;
  t_handle = prev_handle;
  t_in_call = prev_in_call;
  if (prev_device >= 0 && prev_device != t_device && g_dev[prev_device].ready.load()) (void)bind_thread(prev_device);
}

Handle *current_handle() { return t_handle; }
void account_mem(Handle *owner, int64_t device_delta, int64_t pinned_delta) {
  if (owner == nullptr) return;
  std::lock_guard<std::mutex> lk(g_live_mu);
  if (g_live.count(owner) == 0) return;
  owner->mem_dev.fetch_add(device_delta);
  owner->mem_pinned.fetch_add(pinned_delta);
}

// ------------------------------------------------------------------ device memory pool
// A query builds fresh operators (fresh handles) every time it runs, as the reference does; hipMalloc / hipFree cost
// 0.1-1 ms apiece and hipFree synchronises the whole device, which at SF10 was most of a TPC-H query's wall time.
// Freed DevBuf blocks are therefore kept per device in a size-ordered free list and handed out again (best fit within
// 1.25 x); crgpu_mem_usage() counts blocks in use, crgpu_mem_cached() the idle ones, crgpu_mem_trim() returns them to
// the driver, and an allocation failure trims before it gives up.  CRGPU_POOL_MAX_BYTES (default 96 GiB) bounds the
// idle bytes per device; blocks are only pooled once the stream that used them has been synchronised (DevBuf::ensure
// and the destroy entry points do that before releasing).
static std::multimap<size_t, void *> g_pool[kMaxDevices];
static int64_t g_pool_bytes[kMaxDevices];
static std::mutex g_pool_mu;
static int64_t pool_limit() {
  static const int64_t lim = [] {
    const char *v = getenv("CRGPU_POOL_MAX_BYTES");
    return (v != nullptr && *v != 0) ? atoll(v) : (int64_t(96) << 30);
  }();
  return lim;
}
static void *pool_take(int dev, size_t want, size_t *got) {
  if (dev < 0 || dev >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = g_pool[dev].lower_bound(want);
  if (it == g_pool[dev].end() || it->first > want + want / 4) return nullptr;
  void *p = it->second;
  *got = it->first;
  g_pool_bytes[dev] -= static_cast<int64_t>(it->first);
  g_pool[dev].erase(it);
  return p;
}
static bool pool_put(int dev, void *p, size_t cap) {
  if (dev < 0 || dev >= kMaxDevices) return false;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_pool_bytes[dev] + static_cast<int64_t>(cap) > pool_limit()) return false;
  g_pool[dev].emplace(cap, p);
  g_pool_bytes[dev] += static_cast<int64_t>(cap);
  return true;
}
static void pool_trim(int dev) {
  std::multimap<size_t, void *> take;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    take.swap(g_pool[dev]);
    g_pool_bytes[dev] = 0;
  }
  for (auto &kv : take) (void)hipFree(kv.second);
}

crgpu_status DevBuf::ensure(size_t bytes, bool keep) {
  if (bytes <= cap) return CRGPU_OK;
  size_t ncap = cap ? cap : 256;
  while (ncap < bytes) ncap = ncap < (size_t(1) << 30) ? ncap * 2 : ncap + (ncap >> 1);
  ncap = (ncap + 255) & ~size_t(255);
  const int dev = t_device;
  size_t got = 0;
  void *np = pool_take(dev, ncap, &got);
  if (np != nullptr) {
    ncap = got;
  } else {
    hipError_t e = hipMalloc(&np, ncap);
    if (e != hipSuccess) {  // give the idle blocks back to the driver, then retry (with the exact size last)
      (void)hipGetLastError();
      if (dev >= 0) pool_trim(dev);
      e = hipMalloc(&np, ncap);
      if (e != hipSuccess && ncap > bytes) {
        (void)hipGetLastError();
        ncap = (bytes + 255) & ~size_t(255);
        e = hipMalloc(&np, ncap);
      }
    }
    if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  }
  {
    // Test aid (CRGPU_POOL_POISON=<byte>): every block a buffer receives is filled with that byte first.  Pooled blocks
    // keep the contents of their previous owner and fresh ones are zero, so a kernel that forgets to write part of its
    // output can pass by luck; the GPU suite runs clean with 0 and with 165 (see tests/conftest.py).
    static const int poison = getenv("CRGPU_POOL_POISON") != nullptr ? atoi(getenv("CRGPU_POOL_POISON")) : -1;
    if (poison >= 0) {
      hipError_t c = hipMemsetAsync(np, poison & 0xFF, ncap, stream());
      if (c == hipSuccess) c = hipStreamSynchronize(stream());
      if (c != hipSuccess) {
        (void)hipFree(np);
        return hip_fail(c, "hipMemsetAsync(poison)", __FILE__, __LINE__);
      }
    }
  }
  if (keep && p != nullptr && cap > 0) {
    hipError_t c = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, stream());
    if (c == hipSuccess) c = hipStreamSynchronize(stream());
    if (c != hipSuccess) {
      (void)hipFree(np);
      return hip_fail(c, "hipMemcpyAsync(grow)", __FILE__, __LINE__);
    }
  }
  if (p != nullptr) {
    (void)hipStreamSynchronize(stream());  // kernels that still read the old block
    device = device >= 0 ? device : dev;
    if (!pool_put(device, p, cap)) (void)hipFree(p);
    g_mem -= static_cast<int64_t>(cap);
    account_mem(owner, -static_cast<int64_t>(cap), 0);
  }
  p = np;
  cap = ncap;
  device = dev;
  owner = t_handle;
  g_mem += static_cast<int64_t>(cap);
  account_mem(owner, static_cast<int64_t>(cap), 0);
  return CRGPU_OK;
}

void DevBuf::release() {
  if (p != nullptr) {
    if (!pool_put(device, p, cap)) (void)hipFree(p);
    g_mem -= static_cast<int64_t>(cap);
    account_mem(owner, -static_cast<int64_t>(cap), 0);
  }
  p = nullptr;
  cap = 0;
  owner = nullptr;
}

// ------------------------------------------------------------------ profiling
struct ProfEntry {
  double ms = 0;
  int64_t launches = 0;
};
static std::atomic<bool> g_prof_on{false};
static std::map<std::string, ProfEntry> g_prof;
struct Pending {
  std::string name;
  hipEvent_t e0, e1;
};
static std::vector<Pending> g_pending;

static void drain_pending() {  // g_mu held
  for (auto &p : g_pending) {
    float ms = 0;
    if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      g_prof[p.name].ms += ms;
      g_prof[p.name].launches += 1;
    }
    (void)hipEventDestroy(p.e0);
    (void)hipEventDestroy(p.e1);
  }
  g_pending.clear();
}

ProfScope::ProfScope(const char *n) : name(n), on(g_prof_on.load(std::memory_order_relaxed)) {
  if (!on) return;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    on = false;
    return;
  }
  (void)hipEventRecord(e0, stream());
}
ProfScope::~ProfScope() {
  if (!on) return;
  (void)hipEventRecord(e1, stream());
  std::lock_guard<std::mutex> lk(g_mu);
  g_pending.push_back(Pending{name, e0, e1});
}

}  // namespace crgpu

using namespace crgpu;

extern "C" {

crgpu_status crgpu_init(int32_t device) { return init_device(device); }

void crgpu_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  drain_pending();
  if (g_handles.load() > 0) {
    // live handles own streams and buffers on these devices: tearing the contexts down under them would leave them
    // launching on destroyed streams.  Free the handles first.
    set_error("crgpu_shutdown: handles are still alive; nothing was torn down");
    return;
  }
  for (int i = 0; i < kMaxDevices; i++) {
    DeviceCtx &d = g_dev[i];
    if (!d.ready.load()) continue;
    pool_trim(i);
    d.ready.store(false);
    for (auto &se : d.idle) {  // (g_mu is held)
      (void)hipStreamDestroy(se.first);
      (void)hipEventDestroy(se.second);
    }
    d.idle.clear();
    if (d.stream != nullptr) {
      (void)hipStreamSynchronize(d.stream);
      (void)hipStreamDestroy(d.stream);
      d.stream = nullptr;
    }
  }
  g_default_device.store(-1);
  t_device = -1;
}

const char *crgpu_last_error(void) { return t_err.c_str(); }
int64_t crgpu_mem_usage(void) { return g_mem.load(); }
int64_t crgpu_live_handles(void) { return g_handles.load(); }
int64_t crgpu_mem_cached(void) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  int64_t t = 0;
  for (int i = 0; i < kMaxDevices; i++) t += g_pool_bytes[i];
  return t;
}
void crgpu_mem_trim(void) {
  for (int i = 0; i < kMaxDevices; i++)
    if (g_dev[i].ready.load()) pool_trim(i);
}

crgpu_status crgpu_handle_mem_usage(void *handle, int64_t *device_bytes, int64_t *pinned_bytes) {
  Handle *h = static_cast<Handle *>(handle);
  if (h != nullptr && h->magic == kShellMagic) h = static_cast<Shell *>(handle)->inner;  // layer-2 shell: its device handle
  if (h == nullptr || h->magic != kHandleMagic) return fail(CRGPU_ERR_INVALID, "crgpu_handle_mem_usage: not a live handle");
  if (device_bytes != nullptr) *device_bytes = h->mem_dev.load();
  if (pinned_bytes != nullptr) *pinned_bytes = h->mem_pinned.load();
  return CRGPU_OK;
}

crgpu_status crgpu_handle_stats(void *handle, crgpu_handle_stats_t *out) {
  Handle *h = static_cast<Handle *>(handle);
  if (h != nullptr && h->magic == kShellMagic) h = static_cast<Shell *>(handle)->inner;
  if (h == nullptr || h->magic != kHandleMagic || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_handle_stats: not a live handle");
  out->calls = h->st_calls.load();
  out->busy_ns = h->st_ns.load();
  out->rows_in = h->st_rows_in.load();
  out->rows_out = h->st_rows_out.load();
  out->device_bytes = h->mem_dev.load();
  out->pinned_bytes = h->mem_pinned.load();
  return CRGPU_OK;
}

crgpu_status crgpu_cancel(void *handle) {
  Handle *h = static_cast<Handle *>(handle);
  if (h != nullptr && h->magic == kShellMagic) h = static_cast<Shell *>(handle)->inner;  // layer-2 shell: cancel its device handle
  if (h == nullptr || h->magic != kHandleMagic) return fail(CRGPU_ERR_INVALID, "crgpu_cancel: not a live handle");
  h->cancelled = 1;
  return CRGPU_OK;
}

// The caller-owned blocks come from the same per-device free list as the library's own buffers (a query that
// allocates its selection vectors and projected columns through these paid a hipMalloc and a device-synchronising
// hipFree per column: a third of a TPC-H chain's wall time at SF10).  g_user: block -> (capacity, device).
static std::mutex g_user_mu;
static std::unordered_map<void *, std::pair<size_t, int>> g_user;
crgpu_status crgpu_dev_malloc(void **out, size_t bytes) {
  CRGPU_ENTER(nullptr);
  if (out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_malloc: null out");
  *out = nullptr;
  size_t cap = ((bytes ? bytes : 1) + 255) & ~size_t(255);
  const int dev = t_device;
  size_t got = 0;
  void *p = pool_take(dev, cap, &got);
  if (p != nullptr) {
    cap = got;
  } else {
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) {  // give the idle blocks back to the driver, then retry
      (void)hipGetLastError();
      if (dev >= 0) pool_trim(dev);
      e = hipMalloc(&p, cap);
    }
    if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  }
  {
    std::lock_guard<std::mutex> lk(g_user_mu);
    g_user[p] = std::make_pair(cap, dev);
  }
  *out = p;
  return CRGPU_OK;
}
void crgpu_dev_free(void *p) {
  if (p == nullptr) return;
  std::pair<size_t, int> info(0, -1);
  {
    std::lock_guard<std::mutex> lk(g_user_mu);
    auto it = g_user.find(p);
    if (it != g_user.end()) {
      info = it->second;
      g_user.erase(it);
    }
  }
  if (info.first == 0) {  // not one of ours (or freed twice): the driver decides
    (void)hipFree(p);
    return;
  }
  // a pooled block may be handed out again at once: whatever still reads or writes this one must be done first
  // (hipFree waits for the device too) -- on the block's device, which need not be this thread's current one
  int cur = -1;
  (void)hipGetDevice(&cur);
  if (info.second >= 0 && cur != info.second) (void)hipSetDevice(info.second);
  (void)hipDeviceSynchronize();
  if (info.second >= 0 && cur >= 0 && cur != info.second) (void)hipSetDevice(cur);
  if (!pool_put(info.second, p, info.first)) (void)hipFree(p);
}
crgpu_status crgpu_dev_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes) {
  CRGPU_ENTER(nullptr);
  if (bytes == 0) return CRGPU_OK;
  CRGPU_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return CRGPU_OK;
}
crgpu_status crgpu_dev_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes) {
  CRGPU_ENTER(nullptr);
  if (bytes == 0) return CRGPU_OK;
  CRGPU_HIP(hipDeviceSynchronize());  // whichever handle's stream produced src_dev
  CRGPU_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return CRGPU_OK;
}
crgpu_status crgpu_dev_synchronize(void) {
  CRGPU_ENTER(nullptr);
  CRGPU_HIP(hipDeviceSynchronize());
  return CRGPU_OK;
}

void crgpu_prof_enable(int32_t on) { g_prof_on.store(on != 0); }
void crgpu_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  drain_pending();
  g_prof.clear();
}
crgpu_status crgpu_prof_get(const char *name, double *total_ms, int64_t *launches) {
  std::lock_guard<std::mutex> lk(g_mu);
  drain_pending();
  auto it = g_prof.find(name ? name : "");
  if (total_ms) *total_ms = it == g_prof.end() ? 0.0 : it->second.ms;
  if (launches) *launches = it == g_prof.end() ? 0 : it->second.launches;
  return CRGPU_OK;
}

}  // extern "C"
