// TEMPORARY: entry points not implemented yet fail loudly with CRGPU_ERR_UNSUPPORTED.
#include "common.h"
using namespace crgpu;
#define NOTYET(name) return fail(CRGPU_ERR_UNSUPPORTED, name ": not implemented yet")
extern "C" {
crgpu_status crgpu_hash_cols(const crgpu_col *, int32_t, int64_t, const int64_t *, uint64_t, int32_t, uint64_t, void *) { NOTYET("crgpu_hash_cols"); }
crgpu_status crgpu_hj_new(const crgpu_join_spec *, crgpu_hj **) { NOTYET("crgpu_hj_new"); }
crgpu_status crgpu_hj_build_append(crgpu_hj *, const crgpu_batch *) { NOTYET("crgpu_hj_build_append"); }
crgpu_status crgpu_hj_build_finish(crgpu_hj *) { NOTYET("crgpu_hj_build_finish"); }
crgpu_status crgpu_hj_probe_append(crgpu_hj *, const crgpu_batch *) { NOTYET("crgpu_hj_probe_append"); }
crgpu_status crgpu_hj_probe_finish(crgpu_hj *) { NOTYET("crgpu_hj_probe_finish"); }
crgpu_status crgpu_hj_next_output(crgpu_hj *, crgpu_out_batch *) { NOTYET("crgpu_hj_next_output"); }
crgpu_status crgpu_hj_export_buffered(crgpu_hj *, crgpu_out_batch *) { NOTYET("crgpu_hj_export_buffered"); }
crgpu_status crgpu_hj_reset(crgpu_hj *) { NOTYET("crgpu_hj_reset"); }
void crgpu_hj_free(crgpu_hj *) {}
crgpu_status crgpu_agg_new(const crgpu_agg_spec *, crgpu_ha **) { NOTYET("crgpu_agg_new"); }
crgpu_status crgpu_agg_append(crgpu_ha *, const crgpu_batch *) { NOTYET("crgpu_agg_append"); }
crgpu_status crgpu_agg_finish(crgpu_ha *) { NOTYET("crgpu_agg_finish"); }
crgpu_status crgpu_agg_next_output(crgpu_ha *, crgpu_out_batch *) { NOTYET("crgpu_agg_next_output"); }
crgpu_status crgpu_agg_reset(crgpu_ha *) { NOTYET("crgpu_agg_reset"); }
void crgpu_agg_free(crgpu_ha *) {}
crgpu_status crgpu_route_new(const int32_t *, int32_t, const int32_t *, int32_t, int32_t, crgpu_router **) { NOTYET("crgpu_route_new"); }
crgpu_status crgpu_route_partition(crgpu_router *, const crgpu_batch *) { NOTYET("crgpu_route_partition"); }
crgpu_status crgpu_route_finish(crgpu_router *) { NOTYET("crgpu_route_finish"); }
crgpu_status crgpu_route_next_output(crgpu_router *, int32_t, crgpu_out_batch *) { NOTYET("crgpu_route_next_output"); }
void crgpu_route_free(crgpu_router *) {}
}
