// arrow.hip -- the reference's batch wire format, to and from device columns.
//
// Replaces, for the column families of the GPU path (BOOL, INT16/32/64, FLOAT64, BYTES):
//   colserde.RecordBatchSerializer.Serialize / Deserialize   (pkg/col/colserde/record_batch.go:96-314)
//   colserde.ArrowBatchConverter.BatchToArrow / ArrowToBatch (pkg/col/colserde/arrowbatchconverter.go:130-470)
//   coldata.Bytes.Serialize / Deserialize                     (pkg/col/coldata/bytes.go:560-640)
// i.e. what colfetcher.cFetcherWrapper.NextBatch returns when `serialize` is set (cfetcher_wrapper.go:85-143; the
// direct columnar scan of storage/col_mvcc.go:112-166), what kvpb.DeserializeColumnarBatchesFromArrow consumes, and what
// colrpc.Outbox / Inbox put on the wire.
//
// A message is  [u32 metadata length][flatbuffers Message{version, header_type = RecordBatch, header, bodyLength}]
// [padding to 8][body][padding to 8].  The RecordBatch header lists, per column, a FieldNode {length, null_count} and
// its buffers {offset, length} into the body:
//   buffer 0  validity bitmap as coldata.Nulls keeps it (bit = 1: present, LSB first); length 0 when the column has no NULLs
//   fixed-width families: buffer 1 = the values (1 / 2 / 4 / 8 bytes each)
//   BYTES: buffer 1 = n + 1 int32 offsets, buffer 2 = values in the reference's "arrow-like" form (bytes.go:560-615):
//          values = [n 32-byte coldata elements | Bytes.buffer], offsets = [0, ..., 0, 32 n, len(values)]
//          (n = 1: the value itself with offsets [0, len]; n = 0: offsets [0]).  An element holds a value of <= 30 bytes
//          inline (length in byte 30, byte 31 = 1) or {bufferOffset, len, cap} as three int64 (byte 31 = 0).
// Validity bitmaps and fixed-width values have the device layout already: reading is one H2D copy per buffer (to an
// aligned place) and nothing else; BYTES columns take two small kernels (lengths -> scan -> payload copy) to become
// {uint32 offsets[n+1], flat payload}; writing does the inverse (elements + the non-inlined values' buffer).
// The flatbuffers metadata (a few hundred bytes) is parsed and produced on the host.
//
// Roofline: PCIe for the copies; the BYTES kernels stream 32 B/row of elements + the payload (HBM).
#include <cstring>

#include "device.h"

namespace crgpu {

static constexpr int kElem = 32;        // coldata.ElementSize
static constexpr int kMaxInline = 30;   // coldata.BytesMaxInlineLength

// ------------------------------------------------------------------ kernels
// elements -> value lengths (NULL rows keep whatever length their element says: the reference copies elements as-is)
__global__ __launch_bounds__(256) void arrow_elem_len_kernel(const uint8_t *__restrict__ elems, int64_t n, uint64_t buf_len,
                                                             uint32_t *__restrict__ lens, uint32_t *__restrict__ bad) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint8_t *e = elems + i * kElem;
    uint32_t len;
    if (e[31]) {
      len = e[30];
      if (len > static_cast<uint32_t>(kMaxInline)) { *bad = 1; len = 0; }
    } else {
      long long off, l;
      __builtin_memcpy(&off, e, 8);
      __builtin_memcpy(&l, e + 8, 8);
      if (off < 0 || l < 0 || static_cast<uint64_t>(off) + static_cast<uint64_t>(l) > buf_len) { *bad = 1; l = 0; }
      len = static_cast<uint32_t>(l);
    }
    lens[i] = len;
  }
}
// one wave per 64 rows would waste lanes on short values; one lane per value, 8 bytes at a time
__global__ __launch_bounds__(256) void arrow_elem_copy_kernel(const uint8_t *__restrict__ elems, const uint8_t *__restrict__ buf,
                                                              int64_t n, const uint64_t *__restrict__ off64,
                                                              uint32_t *__restrict__ off32, uint8_t *__restrict__ dst) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i <= n; i += stride) {
    const uint64_t o = off64[i];
    off32[i] = static_cast<uint32_t>(o);
    if (i == n) break;
    const uint32_t len = static_cast<uint32_t>(off64[i + 1] - o);
    const uint8_t *e = elems + i * kElem;
    const uint8_t *src = e;
    if (!e[31]) {
      long long boff;
      __builtin_memcpy(&boff, e, 8);
      src = buf + boff;
    }
    uint32_t t = 0;
    for (; t + 8 <= len; t += 8) {
      unsigned long long v;
      __builtin_memcpy(&v, src + t, 8);
      __builtin_memcpy(dst + o + t, &v, 8);
    }
    for (; t < len; t++) dst[o + t] = src[t];
  }
}
// {offsets, payload} -> length of the value if it does not fit an element (else 0)
__global__ __launch_bounds__(256) void arrow_spill_len_kernel(const uint32_t *__restrict__ offsets, const uint8_t *__restrict__ valid,
                                                              int64_t n, uint32_t *__restrict__ lens) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t len = offsets[i + 1] - offsets[i];
    if (valid != nullptr && !((valid[i >> 3] >> (i & 7)) & 1)) len = 0;  // a NULL is the zero element, as coldata keeps it
    lens[i] = len > static_cast<uint32_t>(kMaxInline) ? len : 0;
  }
}
__global__ __launch_bounds__(256) void arrow_elem_build_kernel(const uint32_t *__restrict__ offsets, const uint8_t *__restrict__ data,
                                                               const uint8_t *__restrict__ valid, int64_t n,
                                                               const uint64_t *__restrict__ spill_off, uint8_t *__restrict__ elems,
                                                               uint8_t *__restrict__ buf) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t a0 = offsets[i];
    uint32_t len = offsets[i + 1] - a0;
    const bool null = valid != nullptr && !((valid[i >> 3] >> (i & 7)) & 1);
    uint8_t *e = elems + i * kElem;
    unsigned long long w[4] = {0, 0, 0, 0};
    if (null) {
      w[3] = 1ull << 56;  // element{inlinedLength: 0, inlined: true}: what Bytes.Set(i, nil) leaves
    } else if (len <= static_cast<uint32_t>(kMaxInline)) {
      uint8_t tmp[kElem];
      for (int t = 0; t < kElem; t++) tmp[t] = 0;
      for (uint32_t t = 0; t < len; t++) tmp[t] = data[a0 + t];
      tmp[30] = static_cast<uint8_t>(len);
      tmp[31] = 1;
      __builtin_memcpy(w, tmp, kElem);
    } else {
      const uint64_t so = spill_off[i];
      w[0] = so, w[1] = len, w[2] = len;  // {bufferOffset, len, cap}; byte 31 = 0: not inlined
      uint32_t t = 0;
      for (; t + 8 <= len; t += 8) {
        unsigned long long v;
        __builtin_memcpy(&v, data + a0 + t, 8);
        __builtin_memcpy(buf + so + t, &v, 8);
      }
      for (; t < len; t++) buf[so + t] = data[a0 + t];
    }
    reinterpret_cast<ulonglong2 *>(e)[0] = make_ulonglong2(w[0], w[1]);
    reinterpret_cast<ulonglong2 *>(e)[1] = make_ulonglong2(w[2], w[3]);
  }
}

static inline int agrid(int64_t n) {
  int64_t want = (n + 255) / 256;
  const int64_t cap = static_cast<int64_t>(num_cus()) * 8;
  if (want < 1) want = 1;
  return static_cast<int>(want < cap ? want : cap);
}

// ------------------------------------------------------------------ flatbuffers, the little of it the format needs
struct FbReader {
  const uint8_t *b;
  size_t len;
  bool ok = true;
  template <typename T> T rd(size_t pos) {
    T v{};
    if (pos + sizeof(T) > len) { ok = false; return v; }
    std::memcpy(&v, b + pos, sizeof(T));
    return v;
  }
  // position of field `slot_off` (4, 6, 8, ...) of the table at `tab`, or 0 when absent (flatbuffers Table.Offset)
  size_t field(size_t tab, uint16_t slot_off) {
    const int32_t so = rd<int32_t>(tab);
    const size_t vt = static_cast<size_t>(static_cast<int64_t>(tab) - so);
    if (!ok || vt + 4 > len) { ok = false; return 0; }
    const uint16_t vt_len = rd<uint16_t>(vt);
    if (slot_off + 2 > vt_len) return 0;
    const uint16_t o = rd<uint16_t>(vt + slot_off);
    return o ? tab + o : 0;
  }
  size_t indirect(size_t pos) { return pos + rd<uint32_t>(pos); }
};

// Written the way the flatbuffers builder lays a finished buffer out: root offset, vtables in front of their tables,
// referenced objects behind the referring ones.
struct FbWriter {
  std::vector<uint8_t> out;
  void pad_to(size_t a) { while (out.size() % a) out.push_back(0); }
  template <typename T> size_t put(T v) {
    const size_t at = out.size();
    out.resize(at + sizeof(T));
    std::memcpy(out.data() + at, &v, sizeof(T));
    return at;
  }
  template <typename T> void patch(size_t at, T v) { std::memcpy(out.data() + at, &v, sizeof(T)); }
};

}  // namespace crgpu

using namespace crgpu;

struct crgpu_arrow : crgpu::Handle {
  std::vector<int> types;
  // read side
  std::vector<OutCol> out;
  std::vector<crgpu_dcol> views;
  std::vector<DevBuf> stage_vals, stage_offs;  // BYTES: the serialized values / offsets buffers as they arrived
  DevBuf lens, off64, scratch, flag;
  // write side
  std::vector<DevBuf> w_elems;  // BYTES: [elements | buffer] per column
  std::vector<uint8_t> msg;     // the finished message (host)
};

static int arrow_num_buffers(int type) { return type == CRGPU_BYTES ? 3 : 2; }  // record_batch.go:33-45

extern "C" crgpu_status crgpu_dev_arrow_create(const int32_t *types, int32_t n_cols, crgpu_arrow **out) {
  CRGPU_TRY(ensure_init());
  if (types == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_create: null argument");
  if (n_cols < 1 || n_cols > kMaxCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_arrow_create: column count out of range");
  for (int i = 0; i < n_cols; i++)
    if (!valid_type(types[i])) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_arrow_create: column family outside the GPU path (the CPU converter stays)");
  auto h = new crgpu_arrow();
  crgpu_status os = h->open();
  if (os != CRGPU_OK) {
    delete h;
    return os;
  }
  h->types.assign(types, types + n_cols);
  h->out = std::vector<OutCol>(n_cols);
  h->views.resize(n_cols);
  h->stage_vals = std::vector<DevBuf>(n_cols);
  h->stage_offs = std::vector<DevBuf>(n_cols);
  h->w_elems = std::vector<DevBuf>(n_cols);
  *out = h;
  return CRGPU_OK;
}

extern "C" void crgpu_dev_arrow_destroy(crgpu_arrow *h) {
  if (h == nullptr) return;
  if (h->st != nullptr) (void)hipStreamSynchronize(h->st);
  delete h;
}

// RecordBatchSerializer.Deserialize + ArrowBatchConverter.ArrowToBatch: `msg` is HOST memory.
extern "C" crgpu_status crgpu_dev_arrow_read(crgpu_arrow *h, const uint8_t *msg, int64_t msg_len, crgpu_dev_result *res) {
  if (h == nullptr || msg == nullptr || res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: null argument");
  CRGPU_ENTER(h);
  CRGPU_CHECK_CANCEL(h);
  hipStream_t st = stream();
  if (msg_len < 8) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: message shorter than its length prefix");
  uint32_t meta_len;
  std::memcpy(&meta_len, msg, 4);
  if (4 + static_cast<int64_t>(meta_len) > msg_len) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: metadata length beyond the message");
  FbReader fb{msg + 4, meta_len};
  const size_t mtab = fb.indirect(0);
  const size_t f_type = fb.field(mtab, 6), f_header = fb.field(mtab, 8), f_body = fb.field(mtab, 10);
  if (!fb.ok || f_header == 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: unable to decode metadata table");
  const uint8_t header_type = f_type ? fb.rd<uint8_t>(f_type) : 0;
  if (header_type != 3) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: cannot decode RecordBatch from a message of another type");
  const int64_t body_len = f_body ? fb.rd<int64_t>(f_body) : 0;
  const size_t rb = fb.indirect(f_header);
  const size_t f_len = fb.field(rb, 4), f_nodes = fb.field(rb, 6), f_bufs = fb.field(rb, 8);
  const int64_t n = f_len ? fb.rd<int64_t>(f_len) : 0;
  const int nc = static_cast<int>(h->types.size());
  if (!fb.ok || f_nodes == 0 || f_bufs == 0) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: RecordBatch without nodes / buffers");
  const size_t nodes = fb.indirect(f_nodes), bufs = fb.indirect(f_bufs);
  const uint32_t n_nodes = fb.rd<uint32_t>(nodes), n_bufs = fb.rd<uint32_t>(bufs);
  if (!fb.ok || static_cast<int>(n_nodes) != nc) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: mismatched schema and header lengths");
  int want_bufs = 0;
  for (int c = 0; c < nc; c++) want_bufs += arrow_num_buffers(h->types[c]);
  if (static_cast<int>(n_bufs) != want_bufs) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: buffer count differs from the schema's");
  if (n < 0 || n > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_arrow_read: at most 2^31 rows per batch");
  const int64_t body_at = 4 + static_cast<int64_t>(meta_len);
  if (body_len < 0 || body_at + body_len > msg_len) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: body beyond the message");
  const uint8_t *body = msg + body_at;
  struct Buf { int64_t off, len; };
  auto get_buf = [&](int k) {
    Buf b;
    b.off = fb.rd<int64_t>(bufs + 4 + static_cast<size_t>(k) * 16);
    b.len = fb.rd<int64_t>(bufs + 4 + static_cast<size_t>(k) * 16 + 8);
    return b;
  };
  CRGPU_TRY(h->flag.ensure(4));
  CRGPU_HIP(hipMemsetAsync(h->flag.p, 0, 4, st));
  bool any_bytes = false;
  for (int c = 0, k = 0; c < nc; c++) {
    const int t = h->types[c];
    const int64_t node_len = fb.rd<int64_t>(nodes + 4 + static_cast<size_t>(c) * 16);
    if (node_len != n) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: mismatched field and header lengths");
    Buf b[3];
    for (int j = 0; j < arrow_num_buffers(t); j++, k++) {
      b[j] = get_buf(k);
      if (!fb.ok || b[j].off < 0 || b[j].len < 0 || b[j].off + b[j].len > body_len)
        return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: buffer beyond the message body");
    }
    OutCol &o = h->out[c];
    o.type = t;
    // validity: absent (length 0) when the vector had no NULLs (arrowbatchconverter.go:338-341)
    o.nullable = b[0].len > 0;
    if (o.nullable) {
      if (b[0].len < (n + 7) / 8) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: validity bitmap shorter than the batch");
      CRGPU_TRY(o.valid.ensure(static_cast<size_t>((n + 7) / 8 + 1)));
      CRGPU_HIP(hipMemcpyAsync(o.valid.p, body + b[0].off, static_cast<size_t>((n + 7) / 8), hipMemcpyHostToDevice, st));
    }
    if (t != CRGPU_BYTES) {
      const int64_t need = n * type_width(t);
      // "some value buffers can be nil if the data are all zero values" (record_batch.go:139)
      if (b[1].len != 0 && b[1].len < need) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: values buffer shorter than the batch");
      CRGPU_TRY(o.data.ensure(static_cast<size_t>(need > 0 ? need : 1)));
      if (need > 0) {
        if (b[1].len == 0) CRGPU_HIP(hipMemsetAsync(o.data.p, 0, static_cast<size_t>(need), st));
        else CRGPU_HIP(hipMemcpyAsync(o.data.p, body + b[1].off, static_cast<size_t>(need), hipMemcpyHostToDevice, st));
      }
      h->views[c] = o.view();
      continue;
    }
    // BYTES (Bytes.Deserialize, bytes.go:618-640)
    any_bytes = true;
    if (b[1].len < (n + 1) * 4) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: offsets buffer shorter than the batch");
    CRGPU_TRY(o.offsets.ensure(static_cast<size_t>(n + 1) * 4));
    if (n <= 1) {
      int32_t offs[2] = {0, 0};
      std::memcpy(offs, body + b[1].off, static_cast<size_t>(n + 1) * 4);
      const int64_t vlen = n == 1 ? offs[1] - offs[0] : 0;
      if (vlen < 0 || offs[0] < 0 || offs[0] + vlen > b[2].len) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: BYTES offsets beyond the values");
      CRGPU_TRY(o.data.ensure(static_cast<size_t>(vlen > 0 ? vlen : 1)));
      const uint32_t out_offs[2] = {0, static_cast<uint32_t>(vlen)};
      CRGPU_HIP(hipMemcpyAsync(o.offsets.p, out_offs, static_cast<size_t>(n + 1) * 4, hipMemcpyHostToDevice, st));
      if (vlen > 0) CRGPU_HIP(hipMemcpyAsync(o.data.p, body + b[2].off + offs[0], static_cast<size_t>(vlen), hipMemcpyHostToDevice, st));
      CRGPU_HIP(hipStreamSynchronize(st));  // out_offs is a stack temporary
      h->views[c] = o.view();
      continue;
    }
    int32_t tail[2];
    std::memcpy(tail, body + b[1].off + (n - 1) * 4, 8);
    const int64_t elem_bytes = n * kElem;
    if (tail[0] != elem_bytes || tail[1] < tail[0] || tail[1] > b[2].len)
      return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: BYTES offsets are not the reference's [0, ..., 0, 32 n, total] form");
    const uint64_t buf_len = static_cast<uint64_t>(tail[1] - tail[0]);
    CRGPU_TRY(h->stage_vals[c].ensure(static_cast<size_t>(tail[1]) + 16));
    CRGPU_HIP(hipMemcpyAsync(h->stage_vals[c].p, body + b[2].off, static_cast<size_t>(tail[1]), hipMemcpyHostToDevice, st));
    const uint8_t *elems = h->stage_vals[c].as<uint8_t>();
    CRGPU_TRY(h->lens.ensure(static_cast<size_t>(n) * 4));
    CRGPU_TRY(h->off64.ensure(static_cast<size_t>(n + 1) * 8));
    ProfScope prof("arrow_read_bytes");
    hipLaunchKernelGGL(arrow_elem_len_kernel, dim3(agrid(n)), dim3(256), 0, st, elems, n, buf_len, h->lens.as<uint32_t>(),
                       h->flag.as<uint32_t>());
    CRGPU_HIP(hipGetLastError());
    uint64_t total = 0;
    CRGPU_TRY(exclusive_scan_u32(h->lens.as<uint32_t>(), n, h->off64.as<uint64_t>(), h->scratch, &total));
    if (total > 0xFFFFFFFFull) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_arrow_read: BYTES column beyond 4 GiB");
    CRGPU_TRY(o.data.ensure(static_cast<size_t>(total ? total : 1)));
    hipLaunchKernelGGL(arrow_elem_copy_kernel, dim3(agrid(n + 1)), dim3(256), 0, st, elems, elems + elem_bytes, n,
                       h->off64.as<uint64_t>(), o.offsets.as<uint32_t>(), o.data.as<uint8_t>());
    CRGPU_HIP(hipGetLastError());
    h->views[c] = o.view();
  }
  uint32_t bad = 0;
  if (any_bytes) CRGPU_HIP(hipMemcpyAsync(&bad, h->flag.p, 4, hipMemcpyDeviceToHost, st));
  CRGPU_HIP(hipStreamSynchronize(st));  // the caller's message may be reused after the call
  if (bad) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_read: a BYTES element points outside its buffer");
  res->n = n;
  res->ncols = nc;
  res->_pad = 0;
  res->cols = h->views.data();
  return CRGPU_OK;
}

// ArrowBatchConverter.BatchToArrow + RecordBatchSerializer.Serialize of device columns: the message is assembled in
// library-owned HOST memory, valid until the next call on the handle.
extern "C" crgpu_status crgpu_dev_arrow_write(crgpu_arrow *h, const crgpu_dcol *cols, int64_t n, const uint8_t **msg_out,
                                              int64_t *msg_len_out) {
  if (h == nullptr || msg_out == nullptr || msg_len_out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_write: null argument");
  CRGPU_ENTER(h);
  CRGPU_CHECK_CANCEL(h);
  hipStream_t st = stream();
  const int nc = static_cast<int>(h->types.size());
  if (n < 0 || n > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_arrow_write: at most 2^31 rows per batch");
  if (n > 0 && cols == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_write: null columns");
  struct Piece { const void *dev; std::vector<uint8_t> host; int64_t len; };
  std::vector<Piece> pieces;  // the body's buffers in order
  std::vector<int64_t> bytes_vals_len(nc, 0);
  for (int c = 0; c < nc; c++) {
    const int t = h->types[c];
    if (n > 0) {
      if (cols[c].type != t) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_write: column type differs from the spec");
      if (t == CRGPU_BYTES && cols[c].offsets == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_write: BYTES column without offsets");
      if (t != CRGPU_BYTES && cols[c].data == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_arrow_write: null data pointer");
    }
    const uint8_t *valid = n > 0 ? cols[c].valid : nullptr;
    pieces.push_back(Piece{valid, {}, valid ? (n + 7) / 8 : 0});
    if (t != CRGPU_BYTES) {
      pieces.push_back(Piece{n > 0 ? cols[c].data : nullptr, {}, n * type_width(t)});
      continue;
    }
    // Bytes.Serialize (bytes.go:577-615)
    std::vector<uint8_t> offs(static_cast<size_t>(n + 1) * 4, 0);
    if (n == 0) {
      pieces.push_back(Piece{nullptr, offs, 4});
      pieces.push_back(Piece{nullptr, {}, 0});
      continue;
    }
    if (n == 1) {
      uint32_t o2[2];
      CRGPU_HIP(hipMemcpyAsync(o2, cols[c].offsets, 8, hipMemcpyDeviceToHost, st));
      CRGPU_HIP(hipStreamSynchronize(st));
      uint8_t vbit = 1;
      if (valid != nullptr) {
        CRGPU_HIP(hipMemcpyAsync(&vbit, valid, 1, hipMemcpyDeviceToHost, st));
        CRGPU_HIP(hipStreamSynchronize(st));
        vbit &= 1;
      }
      const int32_t len = vbit ? static_cast<int32_t>(o2[1] - o2[0]) : 0;
      std::memcpy(offs.data() + 4, &len, 4);
      pieces.push_back(Piece{nullptr, offs, 8});
      pieces.push_back(Piece{static_cast<const uint8_t *>(cols[c].data) + o2[0], {}, len});
      continue;
    }
    CRGPU_TRY(h->lens.ensure(static_cast<size_t>(n) * 4));
    CRGPU_TRY(h->off64.ensure(static_cast<size_t>(n + 1) * 8));
    uint64_t spill = 0;
    {
      ProfScope prof("arrow_write_bytes");
      hipLaunchKernelGGL(arrow_spill_len_kernel, dim3(agrid(n)), dim3(256), 0, st, cols[c].offsets, valid, n, h->lens.as<uint32_t>());
      CRGPU_HIP(hipGetLastError());
      CRGPU_TRY(exclusive_scan_u32(h->lens.as<uint32_t>(), n, h->off64.as<uint64_t>(), h->scratch, &spill));
    }
    const int64_t total = n * kElem + static_cast<int64_t>(spill);
    if (total > 0x7FFFFFFFll)  // arrowbatchconverter.go:316-324: "serialized representation ... is too large"
      return fail(CRGPU_ERR_OOM, "crgpu_dev_arrow_write: serialized representation of a BYTES column exceeds the int32 offsets");
    CRGPU_TRY(h->w_elems[c].ensure(static_cast<size_t>(total) + 16));
    {
      ProfScope prof("arrow_write_bytes");
      hipLaunchKernelGGL(arrow_elem_build_kernel, dim3(agrid(n)), dim3(256), 0, st, cols[c].offsets,
                         static_cast<const uint8_t *>(cols[c].data), valid, n, h->off64.as<uint64_t>(), h->w_elems[c].as<uint8_t>(),
                         h->w_elems[c].as<uint8_t>() + n * kElem);
      CRGPU_HIP(hipGetLastError());
    }
    const int32_t e_end = static_cast<int32_t>(n * kElem), t_end = static_cast<int32_t>(total);
    std::memcpy(offs.data() + (n - 1) * 4, &e_end, 4);
    std::memcpy(offs.data() + n * 4, &t_end, 4);
    pieces.push_back(Piece{nullptr, offs, (n + 1) * 4});
    pieces.push_back(Piece{h->w_elems[c].p, {}, total});
  }
  int64_t body_len = 0;
  for (const Piece &p : pieces) body_len += p.len;
  const int64_t body_pad = (8 - (body_len & 7)) & 7;  // calculatePadding (record_batch.go:84-88)

  // ---- metadata (record_batch.go:121-186)
  FbWriter w;
  const size_t root_at = w.put<uint32_t>(0);
  // Message vtable: [vt_len u16][table_len u16][version][header_type][header][bodyLength]
  w.pad_to(2);
  const size_t mvt = w.out.size();
  w.put<uint16_t>(12);  // vtable bytes
  w.put<uint16_t>(24);  // table bytes: soffset 4 | header 4 | bodyLength 8 | version 2 | header_type 1 | pad 5
  w.put<uint16_t>(16);  // version
  w.put<uint16_t>(18);  // header_type
  w.put<uint16_t>(4);   // header
  w.put<uint16_t>(8);   // bodyLength
  // table start must keep its int64 field 8-aligned: table_pos + 8 = 0 mod 8
  while ((w.out.size() + 8) % 8) w.out.push_back(0);
  const size_t mtab = w.out.size();
  w.put<int32_t>(static_cast<int32_t>(mtab - mvt));
  const size_t m_header_at = w.put<uint32_t>(0);
  w.put<int64_t>(body_len);  // totalBufferLen: the body padding is not counted (record_batch.go:174)
  w.put<int16_t>(0);  // MetadataVersionV1
  w.put<uint8_t>(3);  // MessageHeaderRecordBatch
  for (int i = 0; i < 5; i++) w.put<uint8_t>(0);
  w.patch<uint32_t>(root_at, static_cast<uint32_t>(mtab - root_at));
  // RecordBatch vtable + table: [length i64][nodes][buffers]
  w.pad_to(2);
  const size_t rvt = w.out.size();
  w.put<uint16_t>(10);
  w.put<uint16_t>(24);  // soffset 4 | nodes 4 | length 8 | buffers 4 | pad 4
  w.put<uint16_t>(8);   // length
  w.put<uint16_t>(4);   // nodes
  w.put<uint16_t>(16);  // buffers
  while ((w.out.size() + 8) % 8) w.out.push_back(0);
  const size_t rtab = w.out.size();
  w.put<int32_t>(static_cast<int32_t>(rtab - rvt));
  const size_t r_nodes_at = w.put<uint32_t>(0);
  w.put<int64_t>(n);
  const size_t r_bufs_at = w.put<uint32_t>(0);
  w.put<uint32_t>(0);
  w.patch<uint32_t>(m_header_at, static_cast<uint32_t>(rtab - m_header_at));
  // nodes: vector of struct {length i64, null_count i64}; the structs are 8-aligned behind the u32 count
  while ((w.out.size() + 4) % 8) w.out.push_back(0);
  const size_t nodes_at = w.put<uint32_t>(static_cast<uint32_t>(nc));
  for (int c = 0; c < nc; c++) {
    w.put<int64_t>(n);
    w.put<int64_t>(0);  // the reference leaves the null count at 0 (arrowbatchconverter.go:354-361)
  }
  w.patch<uint32_t>(r_nodes_at, static_cast<uint32_t>(nodes_at - r_nodes_at));
  while ((w.out.size() + 4) % 8) w.out.push_back(0);
  const size_t bufs_at = w.put<uint32_t>(static_cast<uint32_t>(pieces.size()));
  int64_t off = 0;
  for (const Piece &p : pieces) {
    w.put<int64_t>(off);
    w.put<int64_t>(p.len);
    off += p.len;
  }
  w.patch<uint32_t>(r_bufs_at, static_cast<uint32_t>(bufs_at - r_bufs_at));
  // metadata + padding so that 4 + len is a multiple of 8 (record_batch.go:190-196)
  while ((4 + w.out.size()) % 8) w.out.push_back(0);
  const uint32_t meta_len = static_cast<uint32_t>(w.out.size());

  h->msg.resize(4 + static_cast<size_t>(meta_len) + static_cast<size_t>(body_len + body_pad));
  std::memcpy(h->msg.data(), &meta_len, 4);
  std::memcpy(h->msg.data() + 4, w.out.data(), meta_len);
  uint8_t *body = h->msg.data() + 4 + meta_len;
  int64_t at = 0;
  for (const Piece &p : pieces) {
    if (p.len > 0) {
      if (!p.host.empty()) std::memcpy(body + at, p.host.data(), static_cast<size_t>(p.len));
      else CRGPU_HIP(hipMemcpyAsync(body + at, p.dev, static_cast<size_t>(p.len), hipMemcpyDeviceToHost, st));
    }
    at += p.len;
  }
  std::memset(body + at, 0, static_cast<size_t>(body_pad));
  CRGPU_HIP(hipStreamSynchronize(st));
  // "To conform to the Arrow spec, zero out all trailing null values" (arrowbatchconverter.go:138-141)
  if (n & 7) {
    int64_t pos = 0;
    for (int c = 0, k = 0; c < nc; c++) {
      if (pieces[k].len > 0) body[pos + pieces[k].len - 1] &= static_cast<uint8_t>((1u << (n & 7)) - 1);
      for (int j = 0; j < arrow_num_buffers(h->types[c]); j++, k++) pos += pieces[k].len;
    }
  }
  *msg_out = h->msg.data();
  *msg_len_out = static_cast<int64_t>(h->msg.size());
  return CRGPU_OK;
}
