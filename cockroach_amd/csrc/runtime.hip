// runtime.hip -- library lifecycle, error reporting, memory accounting, HIP-event profiling.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>

#include "common.h"

namespace crgpu {

static thread_local std::string t_err;
static std::atomic<int64_t> g_mem{0};
static hipStream_t g_stream = nullptr;
static int g_device = -1;
static int g_cus = 256;
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

hipStream_t stream() { return g_stream; }
int num_cus() { return g_cus; }

crgpu_status ensure_init() {
  if (g_stream != nullptr) return CRGPU_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return crgpu_init(dev);
}

crgpu_status DevBuf::ensure(size_t bytes, bool keep) {
  if (bytes <= cap) return CRGPU_OK;
  size_t ncap = cap ? cap : 256;
  while (ncap < bytes) ncap = ncap < (size_t(1) << 30) ? ncap * 2 : ncap + (ncap >> 1);
  ncap = (ncap + 255) & ~size_t(255);
  void *np = nullptr;
  hipError_t e = hipMalloc(&np, ncap);
  if (e != hipSuccess && ncap > bytes) {  // retry with the exact size before giving up
    ncap = (bytes + 255) & ~size_t(255);
    e = hipMalloc(&np, ncap);
  }
  if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  if (keep && p != nullptr && cap > 0) {
    hipError_t c = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, stream());
    if (c == hipSuccess) c = hipStreamSynchronize(stream());
    if (c != hipSuccess) {
      (void)hipFree(np);
      return hip_fail(c, "hipMemcpyAsync(grow)", __FILE__, __LINE__);
    }
  }
  if (p != nullptr) {
    (void)hipStreamSynchronize(stream());
    (void)hipFree(p);
    g_mem -= static_cast<int64_t>(cap);
  }
  p = np;
  cap = ncap;
  g_mem += static_cast<int64_t>(cap);
  return CRGPU_OK;
}

void DevBuf::release() {
  if (p != nullptr) {
    (void)hipFree(p);
    g_mem -= static_cast<int64_t>(cap);
  }
  p = nullptr;
  cap = 0;
}

// ------------------------------------------------------------------ profiling
struct ProfEntry {
  double ms = 0;
  int64_t launches = 0;
};
static bool g_prof_on = false;
static std::map<std::string, ProfEntry> g_prof;
struct Pending {
  std::string name;
  hipEvent_t e0, e1;
};
static std::vector<Pending> g_pending;

static void drain_pending() {
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

ProfScope::ProfScope(const char *n) : name(n), on(g_prof_on) {
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

crgpu_status crgpu_init(int32_t device) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_stream != nullptr && g_device == device) return CRGPU_OK;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(CRGPU_ERR_NO_DEVICE, "crgpu_init: no HIP device visible (the GPU path has no CPU fallback)");
  if (device < 0 || device >= count) return fail(CRGPU_ERR_INVALID, "crgpu_init: device ordinal out of range");
  CRGPU_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  CRGPU_HIP(hipGetDeviceProperties(&prop, device));
  g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (g_stream != nullptr) {
    (void)hipStreamDestroy(g_stream);
    g_stream = nullptr;
  }
  CRGPU_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  g_device = device;
  return CRGPU_OK;
}

void crgpu_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  drain_pending();
  if (g_stream != nullptr) {
    (void)hipStreamSynchronize(g_stream);
    (void)hipStreamDestroy(g_stream);
    g_stream = nullptr;
  }
  g_device = -1;
}

const char *crgpu_last_error(void) { return t_err.c_str(); }
int64_t crgpu_mem_usage(void) { return g_mem.load(); }

crgpu_status crgpu_dev_malloc(void **out, size_t bytes) {
  CRGPU_TRY(ensure_init());
  if (out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_malloc: null out");
  *out = nullptr;
  CRGPU_HIP(hipMalloc(out, bytes ? bytes : 1));
  return CRGPU_OK;
}
void crgpu_dev_free(void *p) {
  if (p != nullptr) (void)hipFree(p);
}
crgpu_status crgpu_dev_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes) {
  CRGPU_TRY(ensure_init());
  if (bytes == 0) return CRGPU_OK;
  CRGPU_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return CRGPU_OK;
}
crgpu_status crgpu_dev_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes) {
  CRGPU_TRY(ensure_init());
  if (bytes == 0) return CRGPU_OK;
  CRGPU_HIP(hipStreamSynchronize(stream()));
  CRGPU_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return CRGPU_OK;
}
crgpu_status crgpu_dev_synchronize(void) {
  CRGPU_TRY(ensure_init());
  CRGPU_HIP(hipStreamSynchronize(stream()));
  return CRGPU_OK;
}

void crgpu_prof_enable(int32_t on) { g_prof_on = on != 0; }
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
