// exchange.hip -- the one exchange step of the path, behind the C-ABI: hash-repartitioned regions swapped between the
// GPUs of a node with RCCL over xGMI.
//
// Replaces the data movement of {HashRouter -> routerOutputOp -> colrpc.Outbox -> gRPC FlowStream -> colrpc.Inbox}
// (pkg/sql/colflow/routers.go:592-715, pkg/sql/colflow/colrpc/outbox.go:153, inbox.go): where the reference serialises
// every routed batch to Arrow and streams it over gRPC, each rank here holds -- after crgpu_dev_partition -- one
// contiguous region per destination and the regions are swapped in place: a tiny all-gather of the counts, then per
// column ONE grouped {ncclSend, ncclRecv} x world (an all-to-all-v; xGMI is point to point, every peer pair uses its own
// link).  BYTES columns travel as a length column plus the payload bytes; validity as one byte per row (bitmaps of
// adjacent regions share bytes).  One process per GPU, or one thread per GPU: a communicator is bound to the device the
// calling thread's library context is on.  RCCL is loaded on first use (dlopen "librccl.so.1": the copy the process
// already holds, e.g. PyTorch's, else ROCm's).
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include <rccl/rccl.h>

#include "device.h"

namespace crgpu {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static crgpu_status rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.lib != nullptr) return CRGPU_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) return fail(CRGPU_ERR_UNSUPPORTED, std::string("crgpu_exchange: RCCL (librccl.so.1) cannot be loaded: ") + dlerror());
#define CRGPU_SYM(field, name)                                                                         \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                             \
  if (g_rccl.field == nullptr) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_exchange: RCCL lacks " name)
  CRGPU_SYM(GetUniqueId, "ncclGetUniqueId");
  CRGPU_SYM(CommInitRank, "ncclCommInitRank");
  CRGPU_SYM(CommDestroy, "ncclCommDestroy");
  CRGPU_SYM(GroupStart, "ncclGroupStart");
  CRGPU_SYM(GroupEnd, "ncclGroupEnd");
  CRGPU_SYM(Send, "ncclSend");
  CRGPU_SYM(Recv, "ncclRecv");
  CRGPU_SYM(AllGather, "ncclAllGather");
  CRGPU_SYM(GetErrorString, "ncclGetErrorString");
#undef CRGPU_SYM
  g_rccl.lib = h;
  return CRGPU_OK;
}

static crgpu_status nccl_fail(ncclResult_t r, const char *what) {
  return fail(CRGPU_ERR_HIP, std::string("RCCL error in ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
}
#define CRGPU_NCCL(expr)                                        \
  do {                                                          \
    ncclResult_t _r = (expr);                                   \
    if (_r != ncclSuccess) return ::crgpu::nccl_fail(_r, #expr); \
  } while (0)

// validity bitmap -> one byte per row (regions of different destinations share bitmap bytes)
__global__ __launch_bounds__(256) void valid_to_bytes_kernel(const uint8_t *__restrict__ bitmap, int64_t n, uint8_t *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (bitmap[i >> 3] >> (i & 7)) & 1;
}
__global__ __launch_bounds__(256) void offsets_to_lens_kernel(const uint32_t *__restrict__ offs, int64_t n, uint32_t *__restrict__ lens) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) lens[i] = offs[i + 1] - offs[i];
}
__global__ __launch_bounds__(256) void narrow_u64_kernel(const uint64_t *__restrict__ in, int64_t n, uint32_t *__restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = static_cast<uint32_t>(in[i]);
}
static inline int xgrid(int64_t n) {
  int64_t want = (n + 255) / 256;
  const int64_t cap = static_cast<int64_t>(num_cus()) * 8;
  if (want < 1) want = 1;
  return static_cast<int>(want < cap ? want : cap);
}

}  // namespace crgpu

using namespace crgpu;

struct crgpu_exchange : crgpu::Handle {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  DevBuf counts_dev, all_counts_dev, tmp_a, tmp_b, offs64, scratch;
  std::vector<OutCol> out;
  std::vector<DevBuf> valid_bytes;
  std::vector<crgpu_dcol> views;
};

extern "C" crgpu_status crgpu_exchange_unique_id(uint8_t *id_out) {
  CRGPU_ENTER(nullptr);
  if (id_out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_unique_id: null argument");
  CRGPU_TRY(rccl_load());
  ncclUniqueId id;
  CRGPU_NCCL(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == CRGPU_EXCHANGE_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id_out, &id, sizeof(id));
  return CRGPU_OK;
}

extern "C" crgpu_status crgpu_exchange_create(const uint8_t *id, int32_t world, int32_t rank, crgpu_exchange **out) {
  CRGPU_TRY(ensure_init());
  if (id == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_create: null argument");
  if (world < 1 || world > 1024 || rank < 0 || rank >= world) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_create: bad world size / rank");
  CRGPU_TRY(rccl_load());
  auto x = new crgpu_exchange();
  crgpu_status os = x->open();
  if (os != CRGPU_OK) {
    delete x;
    return os;
  }
  x->world = world;
  x->rank = rank;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclResult_t r = g_rccl.CommInitRank(&x->comm, world, uid, rank);  // collective: every rank of the node calls it
  if (r != ncclSuccess) {
    delete x;
    return nccl_fail(r, "ncclCommInitRank");
  }
  *out = x;
  return CRGPU_OK;
}

extern "C" void crgpu_exchange_destroy(crgpu_exchange *x) {
  if (x == nullptr) return;
  if (x->st != nullptr) (void)hipStreamSynchronize(x->st);
  if (x->comm != nullptr && g_rccl.CommDestroy != nullptr) (void)g_rccl.CommDestroy(x->comm);
  delete x;
}

// one grouped all-to-all-v of `elem` byte wide items: region p of `src` (scnt[p] items at soff[p]) goes to rank p
static crgpu_status a2a_bytes(crgpu_exchange *x, const uint8_t *src, const std::vector<int64_t> &soff, const std::vector<int64_t> &scnt,
                              uint8_t *dst, const std::vector<int64_t> &roff, const std::vector<int64_t> &rcnt, size_t elem) {
  hipStream_t st = stream();
  // items travel in their own width (counts stay far below 2^31 for 8-byte columns of 2^29 rows) and in pieces of at
  // most 2^25 items: a single ncclSend / ncclRecv of 2 GiB and more was seen to deliver wrong data
  const ncclDataType_t dt = elem == 8 ? ncclInt64 : (elem == 4 ? ncclInt32 : ncclChar);
  const size_t unit = (elem == 8 || elem == 4) ? elem : 1;
  const int64_t per_item = static_cast<int64_t>(elem / unit);
  constexpr int64_t kPiece = int64_t(1) << 25;
  CRGPU_NCCL(g_rccl.GroupStart());
  for (int p = 0; p < x->world; p++) {
    const int64_t sn = scnt[p] * per_item, rn = rcnt[p] * per_item;
    for (int64_t o = 0; o < sn; o += kPiece)
      CRGPU_NCCL(g_rccl.Send(src + (soff[p] * per_item + o) * unit, static_cast<size_t>(sn - o < kPiece ? sn - o : kPiece), dt, p, x->comm, st));
    for (int64_t o = 0; o < rn; o += kPiece)
      CRGPU_NCCL(g_rccl.Recv(dst + (roff[p] * per_item + o) * unit, static_cast<size_t>(rn - o < kPiece ? rn - o : kPiece), dt, p, x->comm, st));
  }
  CRGPU_NCCL(g_rccl.GroupEnd());
  return CRGPU_OK;
}
// what every rank sends to every rank: gathers `mine` (world values) of all ranks; theirs[p] = what rank p sends here
static crgpu_status swap_counts(crgpu_exchange *x, const std::vector<int64_t> &mine, std::vector<int64_t> &theirs) {
  hipStream_t st = stream();
  const int W = x->world;
  CRGPU_TRY(x->counts_dev.ensure(static_cast<size_t>(W) * 8));
  CRGPU_TRY(x->all_counts_dev.ensure(static_cast<size_t>(W) * W * 8));
  CRGPU_HIP(hipMemcpyAsync(x->counts_dev.p, mine.data(), static_cast<size_t>(W) * 8, hipMemcpyHostToDevice, st));
  CRGPU_NCCL(g_rccl.AllGather(x->counts_dev.p, x->all_counts_dev.p, static_cast<size_t>(W) * 8, ncclChar, x->comm, st));
  std::vector<int64_t> all(static_cast<size_t>(W) * W);
  CRGPU_HIP(hipMemcpyAsync(all.data(), x->all_counts_dev.p, all.size() * 8, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
  theirs.resize(W);
  for (int p = 0; p < W; p++) theirs[p] = all[static_cast<size_t>(p) * W + x->rank];
  return CRGPU_OK;
}
static void prefix(const std::vector<int64_t> &cnt, std::vector<int64_t> &off, int64_t *total) {
  off.resize(cnt.size());
  int64_t t = 0;
  for (size_t i = 0; i < cnt.size(); i++) {
    off[i] = t;
    t += cnt[i];
  }
  *total = t;
}

extern "C" crgpu_status crgpu_exchange_all_to_all(crgpu_exchange *x, const crgpu_dcol *cols, int32_t n_cols,
                                                  const int64_t *send_counts, crgpu_dev_result *res, int64_t *recv_counts) {
  if (x == nullptr || res == nullptr || send_counts == nullptr || recv_counts == nullptr)
    return fail(CRGPU_ERR_INVALID, "crgpu_exchange_all_to_all: null argument");
  CRGPU_ENTER(x);
  CRGPU_CHECK_CANCEL(x);
  if (n_cols < 0 || n_cols > kMaxCols || (n_cols > 0 && cols == nullptr)) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_all_to_all: bad columns");
  const int W = x->world;
  hipStream_t st = stream();
  std::vector<int64_t> scnt(send_counts, send_counts + W), rcnt, soff, roff;
  int64_t n = 0, m = 0;
  for (int p = 0; p < W; p++)
    if (scnt[p] < 0) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_all_to_all: negative count");
  prefix(scnt, soff, &n);
  if (n > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_exchange_all_to_all: at most 2^31 rows per call");
  for (int c = 0; c < n_cols; c++) {
    if (!valid_type(cols[c].type)) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_exchange_all_to_all: unsupported column type");
    if (n > 0 && cols[c].type == CRGPU_BYTES && cols[c].offsets == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_all_to_all: BYTES column without offsets");
    if (n > 0 && cols[c].type != CRGPU_BYTES && cols[c].data == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_all_to_all: null data pointer");
  }
  CRGPU_TRY(swap_counts(x, scnt, rcnt));
  prefix(rcnt, roff, &m);
  if (m > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_exchange_all_to_all: more than 2^31 rows arrive; exchange in smaller pieces");
  for (int p = 0; p < W; p++) recv_counts[p] = rcnt[p];
  if (static_cast<int>(x->out.size()) != n_cols) {
    x->out = std::vector<OutCol>(n_cols);
    x->valid_bytes = std::vector<DevBuf>(n_cols);
    x->views.resize(n_cols);
  }
  ProfScope prof("exchange");
  for (int c = 0; c < n_cols; c++) {
    const crgpu_dcol &col = cols[c];
    OutCol &o = x->out[c];
    o.type = col.type;
    // validity: every rank must agree on whether the column carries a bitmap (the plan's nullability), so it is exchanged
    // whenever THIS rank has one and assumed absent otherwise -- callers pass a bitmap on every rank or on none
    o.nullable = col.valid != nullptr;
    if (col.type == CRGPU_BYTES) {
      // lengths travel as a uint32 column; the payload as bytes with its own per-destination byte counts
      CRGPU_TRY(x->tmp_a.ensure(static_cast<size_t>(n > 0 ? n : 1) * 4));
      if (n > 0) hipLaunchKernelGGL(offsets_to_lens_kernel, dim3(xgrid(n)), dim3(256), 0, st, col.offsets, n, x->tmp_a.as<uint32_t>());
      CRGPU_TRY(x->tmp_b.ensure(static_cast<size_t>(m > 0 ? m : 1) * 4));
      CRGPU_TRY(a2a_bytes(x, x->tmp_a.as<uint8_t>(), soff, scnt, x->tmp_b.as<uint8_t>(), roff, rcnt, 4));
      std::vector<uint32_t> bounds(W + 1, 0);  // payload offsets at the region boundaries
      if (n > 0) {
        for (int p = 0; p <= W; p++) {
          const int64_t row = p < W ? soff[p] : n;
          CRGPU_HIP(hipMemcpyAsync(&bounds[p], col.offsets + row, 4, hipMemcpyDeviceToHost, st));
        }
        CRGPU_HIP(hipStreamSynchronize(st));
      }
      std::vector<int64_t> bs(W), br, bso(W), bro;
      for (int p = 0; p < W; p++) {
        bs[p] = static_cast<int64_t>(bounds[p + 1]) - bounds[p];
        bso[p] = bounds[p];
      }
      CRGPU_TRY(swap_counts(x, bs, br));
      int64_t btotal = 0;
      prefix(br, bro, &btotal);
      if (btotal > 0xFFFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_exchange_all_to_all: a BYTES column exceeds 4 GiB after the exchange");
      CRGPU_TRY(o.data.ensure(static_cast<size_t>(btotal > 0 ? btotal : 1)));
      CRGPU_TRY(a2a_bytes(x, static_cast<const uint8_t *>(col.data), bso, bs, o.data.as<uint8_t>(), bro, br, 1));
      // received lengths -> offsets
      CRGPU_TRY(o.offsets.ensure(static_cast<size_t>(m + 1) * 4));
      CRGPU_TRY(x->offs64.ensure(static_cast<size_t>(m + 1) * 8));
      CRGPU_TRY(exclusive_scan_u32(x->tmp_b.as<uint32_t>(), m, x->offs64.as<uint64_t>(), x->scratch, nullptr));
      hipLaunchKernelGGL(narrow_u64_kernel, dim3(xgrid(m + 1)), dim3(256), 0, st, x->offs64.as<uint64_t>(), m + 1, o.offsets.as<uint32_t>());
      CRGPU_HIP(hipGetLastError());
    } else {
      const size_t w = static_cast<size_t>(type_width(col.type));
      CRGPU_TRY(o.data.ensure((static_cast<size_t>(m > 0 ? m : 1)) * w));
      CRGPU_TRY(a2a_bytes(x, static_cast<const uint8_t *>(col.data), soff, scnt, o.data.as<uint8_t>(), roff, rcnt, w));
    }
    if (o.nullable) {
      CRGPU_TRY(x->tmp_a.ensure(static_cast<size_t>(n > 0 ? n : 1)));
      CRGPU_TRY(x->valid_bytes[c].ensure(static_cast<size_t>(m > 0 ? m : 1)));
      CRGPU_TRY(o.valid.ensure(static_cast<size_t>((m + 7) / 8 + 1)));
      if (n > 0) hipLaunchKernelGGL(valid_to_bytes_kernel, dim3(xgrid(n)), dim3(256), 0, st, col.valid, n, x->tmp_a.as<uint8_t>());
      CRGPU_TRY(a2a_bytes(x, x->tmp_a.as<uint8_t>(), soff, scnt, x->valid_bytes[c].as<uint8_t>(), roff, rcnt, 1));
      CRGPU_TRY(pack_bitmap(x->valid_bytes[c].as<uint8_t>(), m, o.valid.as<uint8_t>()));
    }
    x->views[c] = o.view();
  }
  CRGPU_HIP(hipStreamSynchronize(st));
  res->n = m;
  res->ncols = n_cols;
  res->_pad = 0;
  res->cols = x->views.data();
  return CRGPU_OK;
}
