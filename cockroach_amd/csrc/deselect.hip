// deselect.hip -- materialise the rows a selection kept.
//
// Replaces colexecutils.deselectorOp.Next (pkg/sql/colexec/colexecutils/deselector.go:48-80): "copy the selected
// tuples of the input batch into a dense output batch".  The reference carries a selection through its operators as
// the batch's selection vector and only deselects where an operator cannot work through one; the bulk device operators
// here take dense columns, so the BOOL column a device selection produced (crgpu_dev_select_*: 1 = row passes) is
// applied once, in front of a join input (Q3: customer WHERE c_mktsegment = 'BUILDING', orders WHERE o_orderdate < d).
// In front of an aggregator the BOOL column is passed as the aggregates' FILTER instead and no row moves.
//
// Kernels: flag scan + stream compaction (select_flagged), then one gather per column (materialize).
// Roofline: HBM streaming: 1 B/row of flags + the kept rows' bytes read and written.
#include "device.h"

using namespace crgpu;

struct crgpu_deselector : crgpu::Handle {
  std::vector<int> types;
  DevBuf idx, scratch;
  std::vector<OutCol> out;
  std::vector<crgpu_dcol> views;
};

extern "C" crgpu_status crgpu_dev_deselector_create(const int32_t *types, int32_t n_cols, crgpu_deselector **out) {
  CRGPU_TRY(ensure_init());
  if (types == nullptr || out == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_deselector_create: null argument");
  if (n_cols < 1 || n_cols > kMaxCols) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_deselector_create: column count out of range");
  for (int i = 0; i < n_cols; i++)
    if (!valid_type(types[i])) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_deselector_create: unsupported column type");
  auto d = new crgpu_deselector();
  crgpu_status os = d->open();
  if (os != CRGPU_OK) {
    delete d;
    return os;
  }
  d->types.assign(types, types + n_cols);
  d->out = std::vector<OutCol>(n_cols);
  d->views.resize(n_cols);
  *out = d;
  return CRGPU_OK;
}

extern "C" void crgpu_dev_deselector_destroy(crgpu_deselector *d) {
  if (d == nullptr) return;
  if (d->st != nullptr) (void)hipStreamSynchronize(d->st);  // its buffers return to the pool: no kernel may still use them
  delete d;
}

extern "C" crgpu_status crgpu_dev_deselect(crgpu_deselector *d, const crgpu_dcol *cols, int64_t n, const uint8_t *keep_bool,
                                           crgpu_dev_result *res) {
  if (d == nullptr || res == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_deselect: null argument");
  CRGPU_ENTER(d);
  CRGPU_CHECK_CANCEL(d);
  if (n < 0 || n > 0x7FFFFFF0ll) return fail(CRGPU_ERR_UNSUPPORTED, "crgpu_dev_deselect: at most 2^31 rows per call");
  if (n > 0 && (cols == nullptr || keep_bool == nullptr)) return fail(CRGPU_ERR_INVALID, "crgpu_dev_deselect: null columns / flags");
  const int nc = static_cast<int>(d->types.size());
  int64_t kept = 0;
  if (n > 0) {
    for (int c = 0; c < nc; c++) {
      if (cols[c].type != d->types[c]) return fail(CRGPU_ERR_INVALID, "crgpu_dev_deselect: column type differs from the spec");
      if (cols[c].type == CRGPU_BYTES && cols[c].offsets == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_deselect: BYTES column without offsets");
      if (cols[c].type != CRGPU_BYTES && cols[c].data == nullptr) return fail(CRGPU_ERR_INVALID, "crgpu_dev_deselect: null data pointer");
    }
    CRGPU_TRY(d->idx.ensure(static_cast<size_t>(n) * 4));
    ProfScope prof("deselect");
    CRGPU_TRY(select_flagged(keep_bool, n, d->idx.as<int32_t>(), d->scratch, &kept));
  }
  for (int c = 0; c < nc; c++) {
    DCol src{d->types[c], 0, nullptr, nullptr, nullptr};
    if (n > 0) src = to_dcol(cols[c]);
    ProfScope prof("deselect");
    CRGPU_TRY(materialize(src, d->idx.as<int32_t>(), kept, false, d->out[c], d->scratch));
    d->views[c] = d->out[c].view();
  }
  CRGPU_HIP(hipStreamSynchronize(stream()));
  res->n = kept;
  res->ncols = nc;
  res->_pad = 0;
  res->cols = d->views.data();
  return CRGPU_OK;
}
