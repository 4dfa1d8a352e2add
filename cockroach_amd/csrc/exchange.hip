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

#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>

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

// The communicator of a group of exchange handles (crgpu_exchange_create + its crgpu_exchange_clone s): destroyed with
// the last of them.
struct CommBox {
  ncclComm_t comm = nullptr;
  ~CommBox();
};

struct crgpu_exchange : crgpu::Handle {
  std::shared_ptr<CommBox> box;
  ncclComm_t comm = nullptr;  // == box->comm
  int world = 1, rank = 0;
  DevBuf counts_dev, all_counts_dev, offs64, scratch;
  std::vector<OutCol> out;
  std::vector<DevBuf> valid_bytes, valid_send, lens_send, lens_recv;  // per column: all columns of a call are in flight together
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
  x->box = std::make_shared<CommBox>();
  ncclResult_t r = g_rccl.CommInitRank(&x->box->comm, world, uid, rank);  // collective: every rank of the node calls it
  if (r != ncclSuccess) {
    delete x;
    return nccl_fail(r, "ncclCommInitRank");
  }
  x->comm = x->box->comm;
  *out = x;
  return CRGPU_OK;
}

CommBox::~CommBox() {
  if (comm != nullptr && g_rccl.CommDestroy != nullptr) (void)g_rccl.CommDestroy(comm);
}

extern "C" crgpu_status crgpu_exchange_clone(crgpu_exchange *x, crgpu_exchange **out) {
  if (x == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_clone: null argument");
  CRGPU_ENTER(x);
  auto y = new crgpu_exchange();
  crgpu_status os = y->open();
  if (os != CRGPU_OK) {
    delete y;
    return os;
  }
  y->world = x->world;
  y->rank = x->rank;
  y->box = x->box;
  y->comm = x->comm;
  *out = y;
  return CRGPU_OK;
}

extern "C" void crgpu_exchange_destroy(crgpu_exchange *x) {
  if (x == nullptr) return;
  if (x->st != nullptr) (void)hipStreamSynchronize(x->st);
  delete x;  // the communicator goes with the last handle that shares it
}

// Self-regions (source rank == destination rank) are device-to-device copies on the handle's stream; with
// CRGPU_EXCHANGE_SELF=rccl they travel through ncclSend / ncclRecv like every other region (what the one-GPU tests use to
// drive RCCL: with world size 1 every region is a self-region).
static bool self_through_rccl() {
  const char *e = std::getenv("CRGPU_EXCHANGE_SELF");
  return e != nullptr && e[0] == 'r';
}

// one grouped all-to-all-v of `elem` byte wide items: scnt[p] items at soff[p] of `src` go to rank p, rcnt[p] items from
// rank p land at roff[p] of `dst`
static crgpu_status a2a_bytes(crgpu_exchange *x, const uint8_t *src, const std::vector<int64_t> &soff, const std::vector<int64_t> &scnt,
                              uint8_t *dst, const std::vector<int64_t> &roff, const std::vector<int64_t> &rcnt, size_t elem) {
  hipStream_t st = stream();
  // items travel in their own width (counts stay far below 2^31 for 8-byte columns of 2^29 rows) and in pieces of at
  // most 2^25 items: a single ncclSend / ncclRecv of 2 GiB and more was seen to deliver wrong data
  const ncclDataType_t dt = elem == 8 ? ncclInt64 : (elem == 4 ? ncclInt32 : ncclChar);
  const size_t unit = (elem == 8 || elem == 4) ? elem : 1;
  const int64_t per_item = static_cast<int64_t>(elem / unit);
  constexpr int64_t kPiece = int64_t(1) << 25;
  const bool self_rccl = self_through_rccl();
  if (!self_rccl && scnt[x->rank] > 0) {
    if (scnt[x->rank] != rcnt[x->rank]) return fail(CRGPU_ERR_STATE, "crgpu_exchange: self-region counts disagree");
    CRGPU_HIP(hipMemcpyAsync(dst + roff[x->rank] * elem, src + soff[x->rank] * elem, static_cast<size_t>(scnt[x->rank]) * elem,
                             hipMemcpyDeviceToDevice, st));
  }
  if (x->world == 1 && !self_rccl) return CRGPU_OK;
  CRGPU_NCCL(g_rccl.GroupStart());
  for (int p = 0; p < x->world; p++) {
    if (p == x->rank && !self_rccl) continue;
    const int64_t sn = scnt[p] * per_item, rn = rcnt[p] * per_item;
    for (int64_t o = 0; o < sn; o += kPiece)
      CRGPU_NCCL(g_rccl.Send(src + (soff[p] * per_item + o) * unit, static_cast<size_t>(sn - o < kPiece ? sn - o : kPiece), dt, p, x->comm, st));
    for (int64_t o = 0; o < rn; o += kPiece)
      CRGPU_NCCL(g_rccl.Recv(dst + (roff[p] * per_item + o) * unit, static_cast<size_t>(rn - o < kPiece ? rn - o : kPiece), dt, p, x->comm, st));
  }
  CRGPU_NCCL(g_rccl.GroupEnd());
  return CRGPU_OK;
}
// every rank's `words` uint64 values, gathered: all[p * words + i] = value i of rank p (one all-gather + one wait)
static crgpu_status gather_words(crgpu_exchange *x, const std::vector<uint64_t> &mine, std::vector<uint64_t> &all) {
  hipStream_t st = stream();
  const size_t W = static_cast<size_t>(x->world), words = mine.size();
  all.assign(W * words, 0);
  if (W == 1) {  // nothing to agree on with nobody
    all = mine;
    return CRGPU_OK;
  }
  CRGPU_TRY(x->counts_dev.ensure(words * 8));
  CRGPU_TRY(x->all_counts_dev.ensure(W * words * 8));
  CRGPU_HIP(hipMemcpyAsync(x->counts_dev.p, mine.data(), words * 8, hipMemcpyHostToDevice, st));
  CRGPU_NCCL(g_rccl.AllGather(x->counts_dev.p, x->all_counts_dev.p, words * 8, ncclChar, x->comm, st));
  CRGPU_HIP(hipMemcpyAsync(all.data(), x->all_counts_dev.p, all.size() * 8, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));
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

// The exchange proper.  Region p of this rank's columns = rows [soff[p], soff[p] + scnt[p]) goes to rank p (all-to-all:
// disjoint regions in rank order; broadcast: every region is the whole input).  Three collectives carry control
// information so that all ranks take every decision together -- a rank that returned early would leave its peers blocked
// inside a grouped send/recv:
//   1. one all-gather of {row counts per destination, payload bytes per destination of every BYTES column, a bit per
//      column "this rank holds a validity bitmap"}: receive counts, the agreed nullability of every column (the OR: a
//      rank without a bitmap sends all-ones validity) and the size limits (every rank computes EVERY rank's totals from
//      the gathered matrix and fails identically);
//   2. after all buffers are allocated, one all-gather of a status word: a rank-local allocation failure aborts the
//      exchange on every rank;
//   3. the data: per column one grouped {ncclSend, ncclRecv} x (world - 1) + a device copy of the self-region.
static crgpu_status exchange_impl(crgpu_exchange *x, const char *who, const crgpu_dcol *cols, int32_t n_cols, int64_t n,
                                  const std::vector<int64_t> &soff, const std::vector<int64_t> &scnt, crgpu_dev_result *res,
                                  int64_t *recv_counts) {
  const int W = x->world;
  hipStream_t st = stream();
  for (int c = 0; c < n_cols; c++) {
    if (!valid_type(cols[c].type)) return fail(CRGPU_ERR_UNSUPPORTED, std::string(who) + ": unsupported column type");
    if (n > 0 && cols[c].type == CRGPU_BYTES && cols[c].offsets == nullptr) return fail(CRGPU_ERR_INVALID, std::string(who) + ": BYTES column without offsets");
    if (n > 0 && cols[c].type != CRGPU_BYTES && cols[c].data == nullptr) return fail(CRGPU_ERR_INVALID, std::string(who) + ": null data pointer");
  }
  std::vector<int> bcols;  // the BYTES columns
  for (int c = 0; c < n_cols; c++)
    if (cols[c].type == CRGPU_BYTES) bcols.push_back(c);
  const size_t nb = bcols.size();
  // ---- 1. what this rank sends
  const size_t words = static_cast<size_t>(W) * (1 + nb) + 1;
  std::vector<uint64_t> mine(words, 0), all;
  for (int p = 0; p < W; p++) mine[p] = static_cast<uint64_t>(scnt[p]);
  std::vector<std::vector<int64_t>> bso(nb, std::vector<int64_t>(W, 0)), bs(nb, std::vector<int64_t>(W, 0));
  if (n > 0 && nb > 0) {  // payload offsets at the region boundaries
    std::vector<uint32_t> bounds(nb * 2 * W, 0);
    for (size_t k = 0; k < nb; k++)
      for (int p = 0; p < W; p++) {
        CRGPU_HIP(hipMemcpyAsync(&bounds[(k * W + p) * 2], cols[bcols[k]].offsets + soff[p], 4, hipMemcpyDeviceToHost, st));
        CRGPU_HIP(hipMemcpyAsync(&bounds[(k * W + p) * 2 + 1], cols[bcols[k]].offsets + soff[p] + scnt[p], 4, hipMemcpyDeviceToHost, st));
      }
    CRGPU_HIP(hipStreamSynchronize(st));
    for (size_t k = 0; k < nb; k++)
      for (int p = 0; p < W; p++) {
        bso[k][p] = bounds[(k * W + p) * 2];
        bs[k][p] = static_cast<int64_t>(bounds[(k * W + p) * 2 + 1]) - bounds[(k * W + p) * 2];
        mine[static_cast<size_t>(W) * (1 + k) + p] = static_cast<uint64_t>(bs[k][p]);
      }
  }
  uint64_t mask = 0;
  for (int c = 0; c < n_cols; c++)
    if (cols[c].valid != nullptr) mask |= 1ull << c;
  mine[words - 1] = mask;
  CRGPU_TRY(gather_words(x, mine, all));
  // ---- what arrives, on this rank and on every other one
  std::vector<int64_t> rcnt(W), roff;
  std::vector<std::vector<int64_t>> br(nb, std::vector<int64_t>(W, 0)), bro(nb);
  uint64_t any_mask = 0;
  bool rows_over = false, bytes_over = false;
  for (int p = 0; p < W; p++) any_mask |= all[static_cast<size_t>(p) * words + words - 1];
  for (int r = 0; r < W; r++) {  // totals arriving at rank r
    int64_t rows = 0;
    for (int p = 0; p < W; p++) rows += static_cast<int64_t>(all[static_cast<size_t>(p) * words + r]);
    rows_over = rows_over || rows > 0x7FFFFFF0ll;
    for (size_t k = 0; k < nb; k++) {
      int64_t bytes = 0;
      for (int p = 0; p < W; p++) bytes += static_cast<int64_t>(all[static_cast<size_t>(p) * words + static_cast<size_t>(W) * (1 + k) + r]);
      bytes_over = bytes_over || bytes > 0xFFFFFFF0ll;
    }
  }
  if (rows_over) return fail(CRGPU_ERR_UNSUPPORTED, std::string(who) + ": more than 2^31 rows arrive at a rank; exchange in smaller pieces");
  if (bytes_over) return fail(CRGPU_ERR_UNSUPPORTED, std::string(who) + ": a BYTES column exceeds 4 GiB on a rank after the exchange");
  for (int p = 0; p < W; p++) {
    rcnt[p] = static_cast<int64_t>(all[static_cast<size_t>(p) * words + x->rank]);
    for (size_t k = 0; k < nb; k++) br[k][p] = static_cast<int64_t>(all[static_cast<size_t>(p) * words + static_cast<size_t>(W) * (1 + k) + x->rank]);
  }
  int64_t m = 0;
  prefix(rcnt, roff, &m);
  for (int p = 0; p < W; p++) recv_counts[p] = rcnt[p];
  if (static_cast<int>(x->out.size()) != n_cols) {
    x->out = std::vector<OutCol>(n_cols);
    x->valid_bytes = std::vector<DevBuf>(n_cols);
    x->valid_send = std::vector<DevBuf>(n_cols);
    x->lens_send = std::vector<DevBuf>(n_cols);
    x->lens_recv = std::vector<DevBuf>(n_cols);
    x->views.resize(n_cols);
  }
  // ---- 2. every buffer of the data phase, then the agreement that all ranks have theirs
  std::vector<int64_t> btotal(nb, 0);
  crgpu_status local = CRGPU_OK;
  auto need = [&](DevBuf &b, size_t bytes) {
    if (local == CRGPU_OK) local = b.ensure(bytes > 0 ? bytes : 1);
  };
  for (int c = 0, k = 0; c < n_cols; c++) {
    OutCol &o = x->out[c];
    o.type = cols[c].type;
    o.nullable = ((any_mask >> c) & 1ull) != 0;
    if (o.type == CRGPU_BYTES) {
      prefix(br[k], bro[k], &btotal[k]);
      need(x->lens_send[c], static_cast<size_t>(n) * 4);
      need(x->lens_recv[c], static_cast<size_t>(m) * 4);
      need(o.data, static_cast<size_t>(btotal[k]));
      need(o.offsets, static_cast<size_t>(m + 1) * 4);
      need(x->offs64, static_cast<size_t>(m + 1) * 8);
      k++;
    } else {
      need(o.data, static_cast<size_t>(m) * type_width(o.type));
    }
    if (o.nullable) {
      need(x->valid_send[c], static_cast<size_t>(n));
      need(x->valid_bytes[c], static_cast<size_t>(m));
      need(o.valid, static_cast<size_t>((m + 7) / 8 + 1));
    }
  }
  {
    std::vector<uint64_t> st_mine(1, static_cast<uint64_t>(local)), st_all;
    const crgpu_status gs = gather_words(x, st_mine, st_all);
    if (gs != CRGPU_OK) return gs;
    if (local != CRGPU_OK) return local;  // (the message of the failing ensure stands)
    for (int p = 0; p < W; p++)
      if (st_all[p] != 0) return fail(static_cast<crgpu_status>(st_all[p]), std::string(who) + ": rank " + std::to_string(p) + " could not allocate its buffers; the exchange is abandoned on every rank");
  }
  // ---- 3. the data
  ProfScope prof("exchange");
  for (int c = 0, k = 0; c < n_cols; c++) {
    const crgpu_dcol &col = cols[c];
    OutCol &o = x->out[c];
    if (col.type == CRGPU_BYTES) {
      // lengths travel as a uint32 column; the payload as bytes with its own per-destination byte counts
      if (n > 0) hipLaunchKernelGGL(offsets_to_lens_kernel, dim3(xgrid(n)), dim3(256), 0, st, col.offsets, n, x->lens_send[c].as<uint32_t>());
      CRGPU_TRY(a2a_bytes(x, x->lens_send[c].as<uint8_t>(), soff, scnt, x->lens_recv[c].as<uint8_t>(), roff, rcnt, 4));
      CRGPU_TRY(a2a_bytes(x, static_cast<const uint8_t *>(col.data), bso[k], bs[k], o.data.as<uint8_t>(), bro[k], br[k], 1));
      // received lengths -> offsets
      CRGPU_TRY(exclusive_scan_u32(x->lens_recv[c].as<uint32_t>(), m, x->offs64.as<uint64_t>(), x->scratch, nullptr));
      hipLaunchKernelGGL(narrow_u64_kernel, dim3(xgrid(m + 1)), dim3(256), 0, st, x->offs64.as<uint64_t>(), m + 1, o.offsets.as<uint32_t>());
      CRGPU_HIP(hipGetLastError());
      k++;
    } else {
      CRGPU_TRY(a2a_bytes(x, static_cast<const uint8_t *>(col.data), soff, scnt, o.data.as<uint8_t>(), roff, rcnt, static_cast<size_t>(type_width(col.type))));
    }
    if (o.nullable) {
      // validity travels as one byte per row (bitmaps of adjacent regions share bytes); a rank that holds no bitmap for
      // a column some other rank has one for sends "present" for all its rows
      if (n > 0) {
        if (col.valid != nullptr) hipLaunchKernelGGL(valid_to_bytes_kernel, dim3(xgrid(n)), dim3(256), 0, st, col.valid, n, x->valid_send[c].as<uint8_t>());
        else CRGPU_HIP(hipMemsetAsync(x->valid_send[c].p, 1, static_cast<size_t>(n), st));
      }
      CRGPU_TRY(a2a_bytes(x, x->valid_send[c].as<uint8_t>(), soff, scnt, x->valid_bytes[c].as<uint8_t>(), roff, rcnt, 1));
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

extern "C" crgpu_status crgpu_exchange_all_to_all(crgpu_exchange *x, const crgpu_dcol *cols, int32_t n_cols,
                                                  const int64_t *send_counts, crgpu_dev_result *res, int64_t *recv_counts) {
  if (x == nullptr || res == nullptr || send_counts == nullptr || recv_counts == nullptr)
    return fail(CRGPU_ERR_INVALID, "crgpu_exchange_all_to_all: null argument");
  CRGPU_ENTER(x);
  CRGPU_CHECK_CANCEL(x);
  if (n_cols < 0 || n_cols > kMaxCols || (n_cols > 0 && cols == nullptr)) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_all_to_all: bad columns");
  const int W = x->world;
  std::vector<int64_t> scnt(send_counts, send_counts + W), soff;
  int64_t n = 0;
  for (int p = 0; p < W; p++)
    if (scnt[p] < 0) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_all_to_all: negative count");
  prefix(scnt, soff, &n);
  if (n > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_exchange_all_to_all: at most 2^31 rows per call");
  return exchange_impl(x, "crgpu_exchange_all_to_all", cols, n_cols, n, soff, scnt, res, recv_counts);
}

extern "C" crgpu_status crgpu_exchange_broadcast(crgpu_exchange *x, const crgpu_dcol *cols, int32_t n_cols, int64_t n,
                                                 crgpu_dev_result *res, int64_t *recv_counts) {
  if (x == nullptr || res == nullptr || recv_counts == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_broadcast: null argument");
  CRGPU_ENTER(x);
  CRGPU_CHECK_CANCEL(x);
  if (n_cols < 0 || n_cols > kMaxCols || (n_cols > 0 && cols == nullptr)) return fail(CRGPU_ERR_INVALID, "crgpu_exchange_broadcast: bad columns");
  if (n < 0 || n > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_exchange_broadcast: at most 2^31 rows per call");
  // every destination's region is the whole input
  std::vector<int64_t> scnt(x->world, n), soff(x->world, 0);
  return exchange_impl(x, "crgpu_exchange_broadcast", cols, n_cols, n, soff, scnt, res, recv_counts);
}
