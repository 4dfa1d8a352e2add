/*
 * crgpu.h -- C ABI of the MI355X-native hash path for CockroachDB's vectorized
 * engine (hashing -> hash table build/probe -> hash aggregation -> hash
 * repartition).  This is the drop-in boundary: exactly what a cgo shim in the
 * reference would bind (INTEGRATION.md shows the Go side).  Plain pointers and
 * sizes only; no C++/torch types; no exceptions cross it.
 *
 * Two layers are exported:
 *
 *  (1) crgpu_dev_*  -- the bulk, device-resident hot path.  Columns live in HBM
 *      in an Arrow-like layout (crgpu_dcol).  One call processes arbitrarily
 *      many rows.  This is what bench.py measures.
 *  (2) crgpu_hj_* / crgpu_agg_* / crgpu_route_* / crgpu_hash_cols -- the operator
 *      layer that takes coldata.Batch-shaped HOST memory exactly as Go lays it
 *      out (crgpu_col / crgpu_batch), stages it to the device, drives layer (1)
 *      and hands results back in <= coldata.BatchSize() output batches, i.e. the
 *      colexecop.Operator Next() contract (pkg/sql/colexecop/operator.go:22-58).
 *
 * Error style follows the in-tree cgo precedent pkg/geo/geos/geos.h:13-61:
 * every call returns a status; crgpu_last_error() gives the message for the
 * calling thread.  Handles are NOT thread-safe (Init/Next are single-goroutine,
 * operator.go:36,54) except crgpu_route_* (internally locked; routers.go:486-488).
 *
 * Reference paths below are relative to /root/reference/pkg.
 */
#ifndef CRGPU_H
#define CRGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status */
typedef int32_t crgpu_status;
#define CRGPU_OK 0
#define CRGPU_ERR_INVALID 1          /* bad argument / spec -> colexecerror.InternalError */
#define CRGPU_ERR_UNSUPPORTED 2      /* type / aggregate / spec outside the GPU path: plan the reference operator instead
                                        (pattern: supportedNatively, sql/colexec/colbuilder/execplan.go:149) */
#define CRGPU_ERR_OOM 3              /* HBM exhausted -> memory-budget error so the disk spiller engages
                                        (sql/colexec/colexecdisk/disk_spiller.go:81) */
#define CRGPU_ERR_INT_OUT_OF_RANGE 4 /* tree.ErrIntOutOfRange via colexecerror.ExpectedError
                                        (sql/colexec/colexecagg/hash_sum_int_agg.eg.go:295-298) */
#define CRGPU_ERR_HIP 5              /* HIP runtime failure */
#define CRGPU_ERR_NO_DEVICE 6        /* no gfx950 device visible */
#define CRGPU_ERR_STATE 7            /* call made in the wrong operator state */
#define CRGPU_ERR_DUPLICATE 8        /* unordered DISTINCT with errorOnDup saw a duplicate: the shell raises the plan's message
                                        (colexecbase.UpsertDistinctHelper.MaybeEmitErrorOnDup) */
#define CRGPU_ERR_CANCELLED 9        /* crgpu_cancel was called on the handle: the shell returns the context's error
                                        (colexecutils.CancelChecker, sql/colexec/colexecutils/cancel_checker.go:68) */

/* canonical type families: col/typeconv/typeconv.go:30 */
typedef enum {
  CRGPU_BOOL = 0, CRGPU_INT16 = 1, CRGPU_INT32 = 2, CRGPU_INT64 = 3, CRGPU_FLOAT64 = 4, CRGPU_BYTES = 5
} crgpu_type;

/* descpb.JoinType: sql/catalog/descpb/join_type.proto:13-83 */
typedef enum {
  CRGPU_INNER = 0, CRGPU_LEFT_OUTER = 1, CRGPU_RIGHT_OUTER = 2, CRGPU_FULL_OUTER = 3, CRGPU_LEFT_SEMI = 4,
  CRGPU_LEFT_ANTI = 5, CRGPU_INTERSECT_ALL = 6, CRGPU_EXCEPT_ALL = 7, CRGPU_RIGHT_SEMI = 8, CRGPU_RIGHT_ANTI = 9
} crgpu_join_type;

/* execinfrapb.AggregatorSpec_Func: sql/execinfrapb/processors_sql.proto:778-791 */
typedef enum {
  CRGPU_ANY_NOT_NULL = 0, CRGPU_AVG = 1, CRGPU_BOOL_AND = 2, CRGPU_BOOL_OR = 3, CRGPU_COUNT = 5, CRGPU_MAX = 7,
  CRGPU_MIN = 8, CRGPU_SUM = 10, CRGPU_SUM_INT = 11, CRGPU_COUNT_ROWS = 14
} crgpu_agg_func;

/* ------------------------------------------------------------ lifecycle */
/* Initialises HIP device `device` for the library (default stream, CU count) and makes it the CALLING THREAD's
 * library device; the first device a process initialises is also the default for threads that never called
 * crgpu_init.  Idempotent per device; a process may initialise several devices (one Go process, 8 GPUs).
 * hipSetDevice is per OS thread and cgo calls arrive on arbitrary threads, so EVERY entry point re-binds the
 * calling thread: handle calls to the device the handle was created on, handle-less calls to the thread's library
 * device.  Every handle owns its own HIP stream: calls on different handles from different threads run
 * concurrently; calls on ONE handle stay single-threaded (operator.go:36,54) except crgpu_route_* and crgpu_cancel. */
crgpu_status crgpu_init(int32_t device);
/* Tears the device contexts down.  Refused (no-op, message in crgpu_last_error) while any handle is alive: free
 * the handles first. */
void crgpu_shutdown(void);
/* Handles currently alive (every crgpu_*_create / crgpu_*_new not yet destroyed / freed). */
int64_t crgpu_live_handles(void);
/* Asks the operation running (or about to run) on `handle` -- any handle type of this header -- to stop: the flag
 * is checked at entry and between the device passes / staged chunks of a call, which then returns
 * CRGPU_ERR_CANCELLED.  Callable from any thread (the Go shell calls it when ctx.Done() fires, the way
 * colexecutils.CancelChecker polls ctx in the reference, cancel_checker.go:68; hash loops check it at
 * hash_utils_tmpl.go:151, hashtable.go:816).  Sticky until the handle's reset. */
crgpu_status crgpu_cancel(void *handle);
/* Message of the last failing call on this thread ("" if none). */
const char *crgpu_last_error(void);
/* Device bytes currently held by the library; the Go shell reports the delta
 * through colmem.Allocator.AdjustMemoryUsage (sql/colmem/allocator.go:526). */
int64_t crgpu_mem_usage(void);
/* Bytes held on behalf of ONE handle (any handle type of this header): the device buffers and the pinned host staging
 * its calls allocated and have not released.  This is the figure an operator reports to its own account: each
 * operator of the reference owns a colmem.Allocator bound to its memory monitor (sql/colmem/allocator.go:34-60,
 * AdjustMemoryUsage :526, ReleaseAll :560), and the process-wide crgpu_mem_usage cannot be split between two
 * operators that run at the same time.  Buffers allocated by a call are accounted to the handle of the OUTERMOST
 * call on the thread (a layer-2 shell resolves to its device handle); either pointer may be NULL. */
crgpu_status crgpu_handle_mem_usage(void *handle, int64_t *device_bytes, int64_t *pinned_bytes);
/* Execution statistics of ONE handle, for the operator's ComponentStats (execinfrapb.ComponentStats as collected by
 * colflow's vectorizedStatsCollector: batchInfoCollector counts batches / tuples and times Next, stats.go:41-137; the
 * process-wide crgpu_prof_* timings cannot be attributed to an operator).  calls / busy_ns: calls that named this
 * handle outermost and the wall time spent inside them (device work included: every call returns after its device
 * work); rows_in / rows_out: rows handed to build / probe / consume / partition / decode (KV batches for the decoder)
 * and rows of the results handed back. */
typedef struct {
  int64_t calls, busy_ns, rows_in, rows_out, device_bytes, pinned_bytes;
} crgpu_handle_stats_t;
crgpu_status crgpu_handle_stats(void *handle, crgpu_handle_stats_t *out);
/* Freed library buffers are kept in a per-device pool and reused by later handles (a query builds fresh operators on
 * every execution; hipMalloc/hipFree per operator would dominate).  crgpu_mem_cached: idle bytes held by the pool
 * (not part of crgpu_mem_usage); crgpu_mem_trim returns them to the driver -- call it under memory pressure from
 * other users of the GPU (the library trims by itself before reporting CRGPU_ERR_OOM).  CRGPU_POOL_MAX_BYTES
 * (environment, default 96 GiB) bounds the idle bytes per device. */
int64_t crgpu_mem_cached(void);
void crgpu_mem_trim(void);

/* Kernel timing with HIP events on the library stream (for bench.py's roofline
 * object; batchInfoCollector-style stats, sql/colflow/stats.go:41-137).
 * `name` is a kernel family: "hash", "join_build", "join_probe", "agg",
 * "partition", "decode". */
void crgpu_prof_enable(int32_t on);
void crgpu_prof_reset(void);
crgpu_status crgpu_prof_get(const char *name, double *total_ms, int64_t *launches);

/* Raw device memory helpers for callers that do not bring their own allocator
 * (the Go shell stages coldata batches through these; torch callers pass
 * tensor.data_ptr() instead).  Copies are synchronous; crgpu_dev_memcpy_d2h and
 * crgpu_dev_synchronize wait for every stream of the thread's library device.
 * crgpu_dev_malloc / crgpu_dev_free share the library's per-device free list (crgpu_mem_cached / crgpu_mem_trim):
 * a freed block is kept for the next allocation of about its size instead of going back to the driver, and
 * crgpu_dev_free waits for the device first (as hipFree does).  These blocks are the caller's: crgpu_mem_usage and
 * crgpu_handle_mem_usage do not count them. */
crgpu_status crgpu_dev_malloc(void **out, size_t bytes);
void crgpu_dev_free(void *p);
crgpu_status crgpu_dev_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
crgpu_status crgpu_dev_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
crgpu_status crgpu_dev_synchronize(void);

/* ============================================================ layer (1) */
/* Device-resident column.  Fixed-width families: dense array in `data`.
 * BYTES: `offsets[n+1]` (uint32) into the flat `data` buffer.
 * `valid`: validity bitmap with coldata.Nulls bit convention (LSB first,
 * bit = 1 => present; col/coldata/nulls.go:184-201); NULL => no NULLs. */
typedef struct crgpu_dcol {
  int32_t type; /* crgpu_type */
  int32_t _pad;
  const void *data;
  const uint8_t *valid;
  const uint32_t *offsets;
} crgpu_dcol;

/* Library-owned result batch (device memory), valid until the next call on
 * the same handle. */
typedef struct crgpu_dev_result {
  int64_t n;
  int32_t ncols;
  int32_t _pad;
  const crgpu_dcol *cols; /* host array of `ncols` descriptors */
} crgpu_dev_result;

/* Replaces colexechash.rehash + initHash + finalizeHash
 * (sql/colexec/colexechash/hash_utils.eg.go:39, hash_utils.go:65,83,104) over
 * `n` rows: out[i] = finalize(fold over cols of hash(col[i], seed)).
 * width 32: out is uint32[n], running hash truncated after every column
 * (hash tables); width 64: uint64[n] (routing).  nbuckets 0 = no finalize. */
crgpu_status crgpu_dev_hash_cols(const crgpu_dcol *cols, int32_t ncols, int64_t n, uint64_t init, int32_t width,
                                 uint64_t nbuckets, void *out_dev);

/* ---- hash join: replaces colexecjoin.hashJoiner build()/exec()/emitRight()
 * (sql/colexec/colexecjoin/hashjoiner.go:339,504,405) and colexechash.HashTable
 * FullBuild/Check/DistinctCheck (hashtable.go:477, hashtable_tmpl.go:732). */
typedef struct crgpu_join_spec {
  int32_t join_type;             /* crgpu_join_type */
  int32_t right_eq_cols_are_key; /* HashJoinerSpec.RightEqColumnsAreKey; MakeHashJoinerSpec overrides applied inside
                                    (hashjoiner.go:827-883) */
  int32_t n_left_cols, n_right_cols;
  const int32_t *left_types, *right_types; /* crgpu_type per column */
  int32_t n_eq;
  int32_t n_out_cols; /* -1: emit every column of JoinType.MakeOutputTypes; else the PostProcessSpec
                         projection (execinfrapb.PostProcessSpec.OutputColumns, hashjoiner_test.go:939-968)
                         pushed into the joiner so that unused columns are never materialised */
  const int32_t *left_eq, *right_eq; /* equality column ordinals */
  const int32_t *out_cols;           /* n_out_cols ordinals into (left columns ++ right columns) as emitted */
} crgpu_join_spec;

typedef struct crgpu_join crgpu_join;
crgpu_status crgpu_dev_join_create(const crgpu_join_spec *spec, crgpu_join **out);
/* Build side (right input), all rows at once.  The columns are BORROWED and
 * must stay valid and unchanged until reset/destroy (ht.Vals, hashtable.go:252). */
crgpu_status crgpu_dev_join_build(crgpu_join *j, const crgpu_dcol *right_cols, int64_t n);
/* Probe one chunk of the left input; `out` = joined rows for this chunk (left
 * columns then right columns, JoinType.MakeOutputTypes join_type.go:101).
 * RIGHT SEMI / RIGHT ANTI produce no probe-time output (n = 0). */
crgpu_status crgpu_dev_join_probe(crgpu_join *j, const crgpu_dcol *left_cols, int64_t n, crgpu_dev_result *out);
/* The same probe over a batch that carries a selection: the reference's selection operators only set the batch's
 * selection vector and the hash joiner reads its probe batches through it (hashJoiner.exec, hashjoiner.go:504-560:
 * `sel := batch.Selection()`; colexechash.rehash with sel, hash_utils.eg.go:39), nothing is copied in between.  keep_bool
 * is a BOOL column in device memory as crgpu_dev_select_* write it: rows whose byte is 0 are not part of the probe side
 * (they produce nothing, whatever the join type).  NULL: every row.  The packed table probes the selected rows in
 * place; the partitioned and the chained tables copy them out first (as crgpu_dev_deselect would). */
crgpu_status crgpu_dev_join_probe_selected(crgpu_join *j, const crgpu_dcol *left_cols, int64_t n, const uint8_t *keep_bool,
                                           crgpu_dev_result *out);
/* After the last probe chunk: hjEmittingRight (hashjoiner.go:322-328) --
 * unmatched build rows (RIGHT/FULL OUTER, RIGHT ANTI) or matched ones (RIGHT SEMI). */
crgpu_status crgpu_dev_join_emit_right(crgpu_join *j, crgpu_dev_result *out);
crgpu_status crgpu_dev_join_reset(crgpu_join *j); /* colexecop.Resetter */
void crgpu_dev_join_destroy(crgpu_join *j);

/* ---- hash aggregation: replaces colexec.hashAggregator onlineAgg + the
 * colexecagg *HashAgg accumulators (sql/colexec/hash_aggregator.go:354,
 * sql/colexec/colexecagg/hash_*_agg.eg.go). */
typedef struct crgpu_agg_desc {
  int32_t func;       /* crgpu_agg_func */
  int32_t col;        /* AggregatorSpec_Aggregation.ColIdx[0]; -1 for COUNT_ROWS */
  int32_t filter_col; /* FilterColIdx (BOOL column) or -1 */
  int32_t distinct;   /* AggregatorSpec_Aggregation.Distinct: COUNT / SUM / SUM_INT / AVG see the first row of every (group,
                       * input) tuple among the rows passing their FILTER (aggregators_util.go:298-373); ignored for
                       * MIN / MAX / ANY_NOT_NULL / BOOL_AND / BOOL_OR; COUNT_ROWS or computed columns =>
                       * CRGPU_ERR_UNSUPPORTED */
} crgpu_agg_desc;

typedef struct crgpu_agg_spec {
  int32_t n_cols;
  int32_t n_group;
  const int32_t *types;      /* crgpu_type per input column */
  const int32_t *group_cols; /* AggregatorSpec.GroupCols */
  int32_t n_aggs;
  int32_t n_ordered; /* AggregatorSpec.OrderedGroupCols: the input is sorted on the first n_ordered entries of group_cols
                      * (list the ordered grouping columns first); 0 = unordered input.  See crgpu_dev_agg_emit_ready. */
  const crgpu_agg_desc *aggs;
} crgpu_agg_spec;

typedef struct crgpu_agg crgpu_agg;
crgpu_status crgpu_dev_agg_create(const crgpu_agg_spec *spec, crgpu_agg **out);
/* Online: may be called any number of times (hashAggregatorAggregating state). */
crgpu_status crgpu_dev_agg_consume(crgpu_agg *a, const crgpu_dcol *cols, int64_t n);
/* One row per group, one column per aggregation (hashAggregatorOutputting,
 * hash_aggregator_tmpl.go:331-387); arbitrary row order. */
crgpu_status crgpu_dev_agg_finish(crgpu_agg *a, crgpu_dev_result *out);
/* Partially ordered input (spec.n_ordered > 0; hash_aggregator_tmpl.go:160-169, 256-306: the reference aggregates one run
 * of equal ordered-column values at a time and emits its groups before it reads on).  The groups completed by the
 * crgpu_dev_agg_consume calls so far -- every group in front of the last change of the ordered columns -- are handed
 * out in pieces, one per call, oldest first, out->n == 0 when there is none (yet): rows in the order of the runs in the
 * input, any order inside a run (what colexectestutils.PartialOrderedVerifier accepts, utils.go:1337-1400), and removed
 * from the table, which therefore holds the groups behind the last change only.  After the last consume,
 * crgpu_dev_agg_finish completes the remaining groups and returns the OLDEST piece not yet handed out; the caller then
 * calls crgpu_dev_agg_emit_ready until out->n == 0.  Without ordered columns every call returns n == 0.  Limits: no
 * computed columns; at most 2^31 rows between two changes of the ordered columns (CRGPU_ERR_UNSUPPORTED). */
crgpu_status crgpu_dev_agg_emit_ready(crgpu_agg *a, crgpu_dev_result *out);
/* Rows consumed so far whose groups have not been handed out yet: with ordered grouping columns the rows behind the last
 * completed run (the reference's aggregator drops the others from its input tracking once a run's groups are emitted:
 * resetBucketsAndTrackingState, hash_aggregator_tmpl.go:355-378 -- a shell that keeps the input for ExportBuffered
 * keeps this many of the newest rows); without ordered columns every row since the last reset. */
crgpu_status crgpu_dev_agg_pending_rows(crgpu_agg *a, int64_t *rows);
crgpu_status crgpu_dev_agg_reset(crgpu_agg *a);
void crgpu_dev_agg_destroy(crgpu_agg *a);

/* ---- scan -> selection -> projections -> aggregation in ONE pass.  The reference plans
 * colexecsel.sel*Int64Int64ConstOp and colexecproj / colexecprojconst float projections in front of the hash aggregator
 * (Q1: tpch_vec:20537-20546; colexecsel/selection_ops.eg.go:34952, colexecproj/proj_non_const_ops.eg.go:4922,8126,12196,
 * colexecprojconst/proj_const_left_ops.eg.go:4515,7503,11295) and materialises every intermediate vector.  Here the
 * last n_virtual ordinals of spec->types are COMPUTED columns, evaluated inside the aggregation kernels' row loop with
 * the roundings of the separate operators; crgpu_dev_agg_consume then receives only the first n_cols - n_virtual
 * (physical) columns.  A computed column is either
 *   CRGPU_VCOL_PROJECT_F64 (type FLOAT64):  a op b, op as in crgpu_dev_project_f64 (0 +, 1 -, 2 *, 3 a*(1-b), 4 a*(1+b));
 *     b is a physical FLOAT64 column, a a physical FLOAT64 column or an earlier computed projection (two chained
 *     projections at most); NULL if an operand is NULL.  Usable as the input of SUM / AVG / MIN / MAX / COUNT.
 *   CRGPU_VCOL_SELECT_I64_CONST (type BOOL):  a cmp cst, cmp as in crgpu_dev_select_i64_const, a a physical INT64
 *     column; false for a NULL.  Usable as an aggregation's filter_col (a WHERE in front of the aggregator is the
 *     FILTER of every aggregate plus a filtered COUNT_ROWS whose zero groups the caller drops).
 * Grouping columns must be physical.  Everything else behaves as crgpu_dev_agg_create. */
enum { CRGPU_VCOL_PROJECT_F64 = 0, CRGPU_VCOL_SELECT_I64_CONST = 1 };
typedef struct crgpu_agg_vcol {
  int32_t kind; /* CRGPU_VCOL_* */
  int32_t op;   /* projection operator / comparison */
  int32_t a, b; /* operand ordinals (b unused by a selection) */
  int64_t cst;  /* selection constant */
} crgpu_agg_vcol;
crgpu_status crgpu_dev_agg_create_fused(const crgpu_agg_spec *spec, int32_t n_virtual, const crgpu_agg_vcol *vcols,
                                        crgpu_agg **out);

/* ---- hash repartition: replaces colexechash.TupleHashDistributor.Distribute
 * (hash_utils.go:163) + the per-output row copies of HashRouter
 * (sql/colflow/routers.go:683-715).  Rows of `cols` are regrouped by
 * destination = rehash64(init, hash_cols) % num_outputs into one contiguous
 * region per destination; counts_host[p] = rows for output p (host array).
 * The exchange itself (RCCL all-to-all over xGMI) is crgpu_exchange_all_to_all below,
 * which takes exactly this result and these counts. */
typedef struct crgpu_partitioner crgpu_partitioner;
crgpu_status crgpu_dev_partitioner_create(const int32_t *types, int32_t n_cols, const int32_t *hash_cols,
                                          int32_t n_hash, uint64_t init, int32_t num_outputs,
                                          crgpu_partitioner **out);
crgpu_status crgpu_dev_partition(crgpu_partitioner *p, const crgpu_dcol *cols, int64_t n, crgpu_dev_result *out,
                                 int64_t *counts_host);
void crgpu_dev_partitioner_destroy(crgpu_partitioner *p);

/* ---- exchange: the one collective on the data path.  Replaces the movement of routed batches between nodes --
 * HashRouter.Run -> routerOutputOp -> colrpc.Outbox.Run (Arrow serialisation, gRPC FlowStream) -> colrpc.Inbox
 * (sql/colflow/routers.go:592-715, sql/colflow/colrpc/outbox.go:153) -- between the GPUs of one node: after
 * crgpu_dev_partition every rank holds one contiguous region per destination, and crgpu_exchange_all_to_all swaps
 * the regions with RCCL over xGMI (an all-gather of the counts, then one grouped ncclSend/ncclRecv set per column).
 * Bootstrap as RCCL does it: rank 0 calls crgpu_exchange_unique_id, the host ships the 128 bytes to the other ranks
 * (the Go side already ships a FlowSpec to every node: sql/colflow/vectorized_flow.go:1137), every rank calls
 * crgpu_exchange_create (collective).  The communicator lives on the device of the calling thread's library
 * context (crgpu_init): one process per GPU, or one thread per GPU.
 * all_to_all: `cols` are this rank's partitioned columns (rows grouped by destination in rank order, send_counts[p]
 * rows for rank p -- exactly crgpu_dev_partition's result and counts); `out` = the rows received, grouped by source
 * rank, recv_counts[p] from rank p (host arrays of `world` entries).  All families incl. BYTES and validity bitmaps.
 * Collective: every rank calls it with the same schema (column count and types).  What may differ per rank is agreed
 * on inside the call: a column is nullable in the result on every rank as soon as ONE rank passes a validity bitmap
 * for it (ranks without one send "present"), and failures that only one rank can see after the first collective --
 * more than 2^31 rows or 4 GiB of one BYTES column arriving somewhere, an allocation that fails -- are gathered, so
 * that every rank returns the same error instead of leaving its peers blocked in a send/recv.
 *
 * crgpu_exchange_broadcast: every rank contributes its n rows of `cols` and EVERY rank receives the concatenation of
 * all ranks' rows in rank order (recv_counts[p] = rank p's n): the MIRROR router of the reference
 * (execinfrapb.OutputRouterSpec_MIRROR, sql/execinfrapb/data.proto:153-154: "each row is sent to all output streams")
 * in place of the BY_HASH routers AddJoinStage puts on both join inputs (sql/physicalplan/physical_plan.go:1086-1157),
 * for SMALL build sides -- Q9's nation, supplier and filtered part (tpch_vec:20709-20733): every rank then joins its own
 * shard of the probe side against the whole build side and the probe side does not move at all.  The result set is
 * the same as with both sides hash-routed (each probe row meets every build row with its key on exactly one rank). */
#define CRGPU_EXCHANGE_ID_BYTES 128
typedef struct crgpu_exchange crgpu_exchange;
crgpu_status crgpu_exchange_unique_id(uint8_t *id_out /* CRGPU_EXCHANGE_ID_BYTES */);
crgpu_status crgpu_exchange_create(const uint8_t *id, int32_t world, int32_t rank, crgpu_exchange **out);
crgpu_status crgpu_exchange_all_to_all(crgpu_exchange *x, const crgpu_dcol *cols, int32_t n_cols, const int64_t *send_counts,
                                       crgpu_dev_result *out, int64_t *recv_counts);
/* A second handle on the SAME communicator (no collective, no bootstrap): its own stream and its own result buffers.
 * A result is valid until the next call on the handle that produced it, and a plan keeps several exchanged inputs alive
 * at once (Q9: three broadcast build sides and four hash-routed join inputs); the reference gives every stream its own
 * colrpc.Inbox the same way (sql/colflow/vectorized_flow.go:749-787).  Calls on handles of one communicator must be
 * issued in the same order on every rank (one thread per rank does that by construction). */
crgpu_status crgpu_exchange_clone(crgpu_exchange *x, crgpu_exchange **out);
crgpu_status crgpu_exchange_broadcast(crgpu_exchange *x, const crgpu_dcol *cols, int32_t n_cols, int64_t n,
                                      crgpu_dev_result *out, int64_t *recv_counts);
void crgpu_exchange_destroy(crgpu_exchange *x);

/* ---- the one entry point that allocates nothing.  It has NO counterpart in the reference: it exists for callers
 * that cannot call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same repartition as crgpu_dev_partition (same kernels, same destinations), without a handle: the spec travels in
 * the arguments, the regrouped columns go to CALLER-OWNED device buffers out_data[c] (room for n values each,
 * destination order, one contiguous region per destination) and the per-destination row counts to the caller's
 * DEVICE array counts_dev[num_outputs].  The kernels' scratch (destination codes, per-tile histograms, scanned
 * bases) is a static device array of 9.3 MiB that every GPU the library runs on holds until the process ends;
 * no hipMalloc / hipFree / allocating routine is called.  Limits, rejected with CRGPU_ERR_UNSUPPORTED beyond:
 * n <= 2^22 rows per call, num_outputs <= 64, <= 16 fixed-width columns (no BYTES), no validity bitmaps.
 * The work is queued on the default stream of the thread's library device and the call returns without waiting
 * (crgpu_dev_synchronize, or any later handle call, orders after it).  Every call on a GPU shares that GPU's
 * scratch array: calls that overlap on one GPU race -- issue them from one thread at a time. */
crgpu_status crgpu_dev_partition_noalloc(const int32_t *types, int32_t n_cols, const int32_t *hash_cols, int32_t n_hash,
                                         uint64_t init, int32_t num_outputs, const crgpu_dcol *cols, int64_t n,
                                         void *const *out_data, int64_t *counts_dev);

/* ---- unordered DISTINCT: replaces colexec.UnorderedDistinct.Next (sql/colexec/unordered_distinct.go:94-121) and
 * colexechash.HashTable.DistinctBuild (sql/colexec/colexechash/hashtable.go:628-649).  Every call takes one input
 * batch and returns, in input order, all columns of the rows whose `distinct_cols` tuple was not seen before (in an
 * earlier call or earlier in this batch): the first occurrence is the one emitted, as in the reference.  NULLs
 * compare equal unless nulls_are_distinct (then a row with a NULL distinct column is always emitted).  With
 * error_on_dup the call returns CRGPU_ERR_DUPLICATE when it dropped a row (the result is still valid).  At most 2^28
 * rows per call (CRGPU_ERR_UNSUPPORTED beyond: feed smaller batches); CRGPU_ERR_OOM once the stored rows plus the
 * incoming batch exceed 2^29 (a mostly-new batch is stored whole, so the bound counts rows, not distinct keys). */
typedef struct crgpu_distinct crgpu_distinct;
crgpu_status crgpu_dev_distinct_create(const int32_t *types, int32_t n_cols, const int32_t *distinct_cols,
                                       int32_t n_distinct, int32_t nulls_are_distinct, int32_t error_on_dup,
                                       crgpu_distinct **out);
crgpu_status crgpu_dev_distinct_consume(crgpu_distinct *d, const crgpu_dcol *cols, int64_t n, crgpu_dev_result *out);
crgpu_status crgpu_dev_distinct_reset(crgpu_distinct *d);
void crgpu_dev_distinct_destroy(crgpu_distinct *d);

/* ---- selection and projection between the scan and the aggregator (Q1 is
 * scan -> sel -> 3 projections -> hashAggregator, tpch_vec:20537-20546).  Whole-column device kernels with
 * caller-provided outputs; calls are ordered on the library stream with every other crgpu_dev_* call and return
 * without waiting (crgpu_dev_synchronize, or any later blocking call, completes them).
 *
 * crgpu_dev_select_i64_const: replaces colexecsel.sel{EQ,NE,LT,LE,GT,GE}Int64Int64ConstOp.Next
 * (colexecsel/selection_ops.eg.go:5876,25260,34952,...).  cmp: 0 EQ, 1 NE, 2 LT, 3 LE, 4 GT, 5 GE.  Instead of a
 * selection vector it writes a BOOL column, out_bool[i] = 1 iff row i passes (NULL never passes), to be used as the
 * FILTER column of every aggregate: no row is moved.
 *
 * crgpu_dev_project_f64: replaces colexecproj.proj{Plus,Minus,Mult}Float64Float64Op.Next
 * (colexecproj/proj_non_const_ops.eg.go:4922,8126,12196) and the constant forms
 * colexecprojconst.proj{Plus,Minus,Mult}Float64ConstFloat64Op (proj_const_left_ops.eg.go:4515,7503,11295; right
 * twins in proj_const_right_ops.eg.go).  op: 0 +, 1 -, 2 *, and two pairs of those operators done in one pass with
 * the same roundings: 3 a * (1 - b), 4 a * (1 + b).  a or b may be NULL: the operand is then the constant.
 * out_valid (bitmap, (n+7)/8 bytes) = AND of the operands' bitmaps; required when an operand has one. */
crgpu_status crgpu_dev_select_i64_const(int32_t cmp, const crgpu_dcol *col, int64_t cst, int64_t n, uint8_t *out_bool);
crgpu_status crgpu_dev_project_f64(int32_t op, const crgpu_dcol *a, double a_const, const crgpu_dcol *b, double b_const,
                                   int64_t n, double *out_data, uint8_t *out_valid);

/* crgpu_dev_select_bytes_const: replaces colexecsel.selEQBytesBytesConstOp.Next / selNEBytesBytesConstOp.Next
 * (colexecsel/selection_ops.eg.go:358,10050) and, for LIKE '%c%' patterns, selContainsBytesBytesConstOp.Next
 * (colexecsel/sel_like_ops.eg.go:229; case-sensitive form).  cmp: 0 EQ, 1 NE, 6 CONTAINS, 7 NOT CONTAINS; the
 * constant is at most 64 bytes.  Writes a BOOL column like crgpu_dev_select_i64_const (NULL never passes).
 * Q3: c_mktsegment = 'BUILDING' (tpch_vec:20588); Q9: p_name LIKE '%green%' (tpch_vec:20731). */
crgpu_status crgpu_dev_select_bytes_const(int32_t cmp, const crgpu_dcol *col, const uint8_t *cst, int32_t cst_len,
                                          int64_t n, uint8_t *out_bool);

/* crgpu_dev_extract_year: EXTRACT(year FROM d) of a DATE column (INT64 days since 1970-01-01, typeconv.go:46-47) as
 * FLOAT64 -- what the reference evaluates row by row through colexecbuiltins.defaultBuiltinFuncOperator.Next
 * (colexecbuiltins/builtin_funcs.go:44) and sem/builtins extractTimeSpanFromTimestamp (builtins.go:11946:
 * tree.DFloat(fromTime.Year())).  Q9's o_year group key (tpch_vec:20715-20720).  The column must not hold NULLs
 * (pass its validity bitmap on to the consumer otherwise; values under a NULL are unspecified). */
crgpu_status crgpu_dev_extract_year(const crgpu_dcol *days, int64_t n, double *out_data);

/* ---- deselection: replaces colexecutils.deselectorOp.Next (sql/colexec/colexecutils/deselector.go:48-80): the rows
 * whose keep_bool byte (a BOOL column produced by crgpu_dev_select_*, device memory) is non-zero, in input order,
 * as dense library-owned columns (valid until the next call on the handle).  Used in front of a join input; in
 * front of an aggregator pass the BOOL column as the aggregates' FILTER instead (no row moves). */
typedef struct crgpu_deselector crgpu_deselector;
crgpu_status crgpu_dev_deselector_create(const int32_t *types, int32_t n_cols, crgpu_deselector **out);
crgpu_status crgpu_dev_deselect(crgpu_deselector *d, const crgpu_dcol *cols, int64_t n, const uint8_t *keep_bool,
                                crgpu_dev_result *out);
void crgpu_dev_deselector_destroy(crgpu_deselector *d);

/* ---- hash group-join, fused: replaces colexec.hashGroupJoiner (sql/colexec/hash_group_joiner.go:20-76, planned at
 * colbuilder/execplan.go:1385-1530 from execinfrapb.HashGroupJoinerSpec, processors_sql.proto:1035-1072: "the join's
 * equality columns are exactly the same as the aggregation's grouping columns").  The reference composes joiner ->
 * aggregator ("TODO: add optimized implementation"); this form aggregates while probing and materialises no joined
 * row.  Scope of the fused form: INNER join on ONE INT64 equality column with unique build keys (a duplicate-heavy
 * build side => CRGPU_ERR_UNSUPPORTED at build), at most one 8-byte payload column per side (INT64 or FLOAT64, no
 * NULLs), aggregations COUNT_ROWS / COUNT, SUM / AVG (FLOAT64), SUM_INT, MIN, MAX over the probe payload and
 * ANY_NOT_NULL of the key or the build payload, within four accumulator words per group.  Anything else =>
 * CRGPU_ERR_UNSUPPORTED: the caller composes crgpu_dev_join_probe -> crgpu_dev_agg_consume (device-resident too).
 * One probe call takes the whole probe side and returns one row per group that has a match (arbitrary order):
 * column f = aggregation f.
 * CRGPU_RIGHT_OUTER (the build side is the join's right input, as in the reference's hash joiner, and is KEPT; a plan
 * whose kept side is the left one with unique keys -- TPC-H Q13's customer LEFT OUTER JOIN orders GROUP BY c_custkey --
 * swaps its inputs): one row per BUILD row, also without a probe row -- then COUNT_ROWS = 1 (the one joined row),
 * COUNT over the probe payload = 0 and SUM / AVG / SUM_INT / MIN / MAX over it NULL (those columns carry a validity
 * bitmap); the build key column must not be nullable; one probe call per build, an empty probe side included. */
typedef struct crgpu_gjoin_agg {
  int32_t func;  /* crgpu_agg_func */
  int32_t input; /* 0 the key, 1 the probe side's payload column, 2 the build side's payload column */
} crgpu_gjoin_agg;
typedef struct crgpu_gjoin_spec {
  int32_t join_type;          /* CRGPU_INNER, or CRGPU_RIGHT_OUTER: groups for every build row */
  int32_t key_type;           /* CRGPU_INT64 */
  int32_t probe_payload_type; /* 0: none */
  int32_t build_payload_type; /* 0: none */
  int32_t n_aggs;
  int32_t _pad;
  const crgpu_gjoin_agg *aggs;
} crgpu_gjoin_spec;
typedef struct crgpu_gjoin crgpu_gjoin;
crgpu_status crgpu_dev_gjoin_create(const crgpu_gjoin_spec *spec, crgpu_gjoin **out);
crgpu_status crgpu_dev_gjoin_build(crgpu_gjoin *g, const crgpu_dcol *key, const crgpu_dcol *payload, int64_t n);
crgpu_status crgpu_dev_gjoin_probe(crgpu_gjoin *g, const crgpu_dcol *key, const crgpu_dcol *payload, int64_t n,
                                   crgpu_dev_result *out);
void crgpu_dev_gjoin_destroy(crgpu_gjoin *g);

/* ---- Grace-partitioned (external) hash join / hash aggregation: HBM -> host memory -> disk.
 * Replaces colexecdisk.hashBasedPartitioner (sql/colexec/colexecdisk/hash_based_partitioner.go:143-717) around the
 * in-memory hash joiner (external_hash_joiner.go:62-143) or hash aggregator (external_hash_aggregator.go:32-95): what
 * the disk spiller (disk_spiller.go:81-200) switches to when the in-memory operator reports CRGPU_ERR_OOM / exceeds its
 * budget (feed it the operator's ExportBuffered tuples, then the rest of the input).
 *   crgpu_dev_grace_add    hbpInitialPartitioning: distribute device rows of input 0 (left / the aggregator's input)
 *                          or 1 (right = build side) over num_buckets partitions by the reference's
 *                          TupleHashDistributor hash (seed DefaultInitHashValue + 1) and move them out of HBM
 *   crgpu_dev_grace_next   the first call ends the partitioning phase; every call returns the next piece of output
 *                          (library-owned device columns, valid until the next call): partitions whose build side fits
 *                          mem_limit run through the in-memory operator, larger ones are repartitioned with the next
 *                          seed, children that shrink by < 5 % run with the budget lifted (the reference's sort-based
 *                          fallback; a real device OOM is returned as CRGPU_ERR_OOM).  *done = 1 after the last piece.
 * Output rows are those of the in-memory operator on the whole input, in partition order. */
typedef struct crgpu_grace crgpu_grace;
typedef struct crgpu_grace_opts {
  int64_t mem_limit;        /* maxPartitionSizeToProcessUsingMain: bytes of last-input rows a partition may hold (>= 64 KiB; 0: 64 MiB) */
  int64_t host_limit;       /* bytes of spilled rows kept in host memory; beyond it chunks go to files (0: unlimited) */
  const char *spill_dir;    /* directory for the disk tier, or NULL: host memory only */
  int32_t num_buckets;      /* partitions per (re)partitioning step (0: 16) */
  int32_t max_repartitions; /* recursion bound (0: 64) */
} crgpu_grace_opts;
typedef struct crgpu_grace_stats {
  int64_t partitions_processed, repartitions, fallback_partitions;
  int64_t bytes_spilled, bytes_spilled_disk;
} crgpu_grace_stats;
crgpu_status crgpu_dev_grace_join_create(const crgpu_join_spec *spec, const crgpu_grace_opts *opts, crgpu_grace **out);
crgpu_status crgpu_dev_grace_agg_create(const crgpu_agg_spec *spec, const crgpu_grace_opts *opts, crgpu_grace **out);
crgpu_status crgpu_dev_grace_add(crgpu_grace *g, int32_t input, const crgpu_dcol *cols, int64_t n);
crgpu_status crgpu_dev_grace_next(crgpu_grace *g, crgpu_dev_result *out, int32_t *done);
crgpu_status crgpu_dev_grace_stats(crgpu_grace *g, crgpu_grace_stats *out);
void crgpu_dev_grace_destroy(crgpu_grace *g);

/* ---- the batch wire format: replaces colserde.RecordBatchSerializer.Serialize / Deserialize
 * (col/colserde/record_batch.go:96-314), colserde.ArrowBatchConverter.BatchToArrow / ArrowToBatch
 * (col/colserde/arrowbatchconverter.go:130-470) and coldata.Bytes.Serialize / Deserialize (col/coldata/bytes.go:560-640)
 * for the column families of the GPU path.  crgpu_dev_arrow_read takes one serialized RecordBatch message in HOST
 * memory -- what colfetcher.cFetcherWrapper.NextBatch returns for a direct columnar scan (cfetcher_wrapper.go:85-143,
 * storage/col_mvcc.go:112-166) and what colrpc.Inbox receives -- and returns library-owned device columns (valid
 * until the next call on the handle): validity bitmaps and fixed-width values are copied as they are, BYTES columns
 * are converted on the device from the reference's [elements | buffer] form to {offsets, payload}.
 * crgpu_dev_arrow_write is the inverse (device columns, e.g. crgpu_dev_decode_kv's result, to a message in
 * library-owned HOST memory valid until the next call): with it the device decoder is a storage.CFetcherWrapper whose
 * NextBatch serializes.  Other families (Decimal, Interval, TimestampTZ, JSON, datum-backed) =>
 * CRGPU_ERR_UNSUPPORTED at create: the CPU converter stays for such schemas. */
typedef struct crgpu_arrow crgpu_arrow;
crgpu_status crgpu_dev_arrow_create(const int32_t *types, int32_t n_cols, crgpu_arrow **out);
crgpu_status crgpu_dev_arrow_read(crgpu_arrow *h, const uint8_t *msg, int64_t msg_len, crgpu_dev_result *out);
crgpu_status crgpu_dev_arrow_write(crgpu_arrow *h, const crgpu_dcol *cols, int64_t n, const uint8_t **msg_out,
                                   int64_t *msg_len_out);
void crgpu_dev_arrow_destroy(crgpu_arrow *h);

/* ---- KV fetch decode: replaces colfetcher.cFetcher.NextBatch for primary indexes
 * (sql/colfetcher/cfetcher.go:758-1093: one KV per row -- the MaxKeysPerRow == 1 fast path :937-941 -- and rows of
 * several column families, stateFetchNextKVWithUnfinishedRow :946-1020) together with
 * colencoding.DecodeKeyValsToCols (colencoding/key_encoding.go:43), processValue (cfetcher.go:1136),
 * processValueBytes (:1313) with colencoding.DecodeTableValueToCol (value_encoding.go:24), processValueSingle (:1280)
 * with colencoding.UnmarshalColumnValueToCol (key_encoding.go:226-297: INT / FLOAT / BYTES single-column families),
 * getCurrentColumnFamilyID (:1447), fillNulls (:1401) and the MVCC extended value header
 * (roachpb/data.go:288-289,336-338,436-489), which is skipped.  The spec is the subset of fetchpb.IndexFetchSpec
 * the decode needs (sql/catalog/fetchpb/index_fetch.proto:57-127: KeyPrefixLength, key columns with direction,
 * fetched columns by column id, MaxKeysPerRow, MaxFamilyID, FamilyDefaultColumns).  Types: crgpu_type; other
 * families (DECIMAL, TIMESTAMP, ...) stay on the CPU fetcher; such columns may still be PRESENT in the rows, they
 * are skipped.  Secondary-index value forms (cfetcher.go:1218-1268) stay on the CPU fetcher. */
typedef struct crgpu_fetch_spec {
  int32_t key_prefix_len;  /* IndexFetchSpec.KeyPrefixLength: tenant/table/index prefix bytes */
  int32_t n_key_cols;      /* index key columns, in key order (<= 8) */
  const int32_t *key_types;
  const int32_t *key_dirs; /* 0 ascending, 1 descending (catenumpb.IndexColumn_Direction) */
  const int32_t *key_out;  /* output ordinal, or -1 when the column is not fetched */
  int32_t n_val_cols;      /* fetched value columns, by increasing column id (<= 32) */
  int32_t n_out_cols;      /* <= 32 */
  const int32_t *val_col_ids;
  const int32_t *val_types;
  const int32_t *val_out;
  const int32_t *out_types; /* crgpu_type per output column */
  /* Column families.  All zero / NULL: the index stores one KV per row in family 0. */
  int32_t max_keys_per_row;   /* IndexFetchSpec.MaxKeysPerRow; > 1: consecutive KVs that share the row prefix are ONE row */
  int32_t max_family_id;      /* IndexFetchSpec.MaxFamilyID; != 0: the family id is read from the key (keys.MakeFamilyKey) */
  int32_t n_family_defaults;  /* IndexFetchSpec.FamilyDefaultColumns (<= 16) */
  int32_t _pad;
  const int32_t *family_ids;             /* FamilyDefaultColumn.FamilyID */
  const int32_t *family_default_col_ids; /* FamilyDefaultColumn.DefaultColumnID: a non-TUPLE value of that family is this column's datum */
} crgpu_fetch_spec;

typedef struct crgpu_decoder crgpu_decoder;
crgpu_status crgpu_dev_decoder_create(const crgpu_fetch_spec *spec, crgpu_decoder **out);
/* repr_dev: concatenated MVCCScan batch responses (storage/enginepb/decode.go:77-128) in device
 * memory, 16-byte aligned and followed by >= 16 readable bytes; batch b spans [batch_offsets_host[b], batch_offsets_host[b+1]).
 * out: one row per KV (per group of KVs sharing a row prefix when max_keys_per_row > 1; a call must hold whole rows,
 * as scans with WholeRowsOfSize deliver them) in scan order, one column per output ordinal; columns requested but
 * absent from a row are NULL.  Malformed input => CRGPU_ERR_INVALID naming the first bad row. */
crgpu_status crgpu_dev_decode_kv(crgpu_decoder *d, const uint8_t *repr_dev, const int64_t *batch_offsets_host,
                                 int64_t n_batches, crgpu_dev_result *out);
void crgpu_dev_decoder_destroy(crgpu_decoder *d);

/* ============================================================ layer (2) */
/* One coldata.Vec as Go lays it out (col/coldata/vec.go:44, native_types.go:15-37,
 * bytes.go:45-74,156-165): fixed-width families -> the slice's backing array;
 * BYTES -> &Bytes.elements[0] (32-byte elements, <= 30 bytes inline) and
 * &Bytes.buffer[0].  All memory is BORROWED for the duration of the call only
 * (cgo rule: no Go pointer is retained). */
typedef struct crgpu_col {
  int32_t type;            /* crgpu_type */
  int32_t maybe_has_nulls; /* Nulls.MaybeHasNulls() */
  const void *data;        /* values, or Bytes.elements */
  const uint8_t *nulls;    /* Nulls.NullBitmap(), bit = 1 => present; may be NULL if !maybe_has_nulls */
  const uint8_t *bytes_buf; /* Bytes.buffer */
  int64_t bytes_buf_len;
} crgpu_col;

/* One coldata.Batch (col/coldata/batch.go:24): Length(), Selection(), ColVecs(). */
typedef struct crgpu_batch {
  int64_t length;
  const int64_t *sel; /* Go []int; NULL => rows 0..length-1 */
  int32_t ncols;
  int32_t _pad;
  const crgpu_col *cols;
} crgpu_batch;

/* Caller-owned output batch (allocator.ResetMaybeReallocate on the Go side).
 * Fixed-width: `data` has room for `capacity` values.  BYTES: `data` is the
 * elements array (capacity * 32 bytes), `bytes_buf` the spill buffer of
 * `bytes_buf_cap` bytes; if a batch needs more, the call returns CRGPU_OK with
 * length = 0 and bytes_buf_len = required size so the caller can grow and retry. */
typedef struct crgpu_out_col {
  int32_t type;
  int32_t has_nulls; /* out */
  void *data;
  uint8_t *nulls; /* (capacity+7)/8 bytes */
  uint8_t *bytes_buf;
  int64_t bytes_buf_cap;
  int64_t bytes_buf_len; /* out */
} crgpu_out_col;

/* crgpu_out_batch.flags.  CRGPU_OUT_BORROW: the values of fixed-width columns are not copied into the caller's `data`
 * buffers; instead `data` is SET to point at them inside the library's pinned result buffer (`data` may be NULL on
 * entry for such columns).  The pointers stay valid until the next output call on the same handle (a router: on the
 * same output), its reset or its free -- the lifetime coldata.Batch promises a consumer anyway (colexecop/operator.go:
 * "the batch is only valid until the next call to Next"), so a Go shell can wrap them with unsafe.Slice as the Vec's
 * backing array and saves the second host copy of every output byte.  Null bitmaps and BYTES columns are written to
 * the caller's buffers as without the flag. */
enum { CRGPU_OUT_BORROW = 1 };
typedef struct crgpu_out_batch {
  int64_t capacity; /* <= coldata.BatchSize() typically */
  int64_t length;   /* out; 0 = no more output (EOF of this phase) */
  int32_t ncols;
  int32_t flags;    /* CRGPU_OUT_*; 0: every value is copied into the caller's buffers */
  crgpu_out_col *cols;
} crgpu_out_batch;

/* colexechash.rehash over host vectors with a selection vector
 * (hash_utils.eg.go:39; used by the Go router / partitioner shells). */
crgpu_status crgpu_hash_cols(const crgpu_col *cols, int32_t ncols, int64_t n, const int64_t *sel, uint64_t init,
                             int32_t width, uint64_t nbuckets, void *out_host);

/* colexecjoin.NewHashJoiner + hashJoiner.Next() (hashjoiner.go:886-910,291). */
typedef struct crgpu_hj crgpu_hj;
crgpu_status crgpu_hj_new(const crgpu_join_spec *spec, crgpu_hj **out);
crgpu_status crgpu_hj_build_append(crgpu_hj *h, const crgpu_batch *right); /* InputTwo.Next() batches */
crgpu_status crgpu_hj_build_finish(crgpu_hj *h);                           /* right input exhausted */
crgpu_status crgpu_hj_probe_append(crgpu_hj *h, const crgpu_batch *left);  /* InputOne.Next() batches */
crgpu_status crgpu_hj_probe_finish(crgpu_hj *h);                           /* left input exhausted */
/* Next(): fills `out` with up to out->capacity joined rows; length 0 once
 * everything appended so far has been emitted. */
crgpu_status crgpu_hj_next_output(crgpu_hj *h, crgpu_out_batch *out);
/* BufferingInMemoryOperator.ExportBuffered (colexecop/operator.go:167-196):
 * hands the buffered build rows back so the CPU disk-backed joiner can take over. */
crgpu_status crgpu_hj_export_buffered(crgpu_hj *h, crgpu_out_batch *out);
crgpu_status crgpu_hj_reset(crgpu_hj *h);
void crgpu_hj_free(crgpu_hj *h);

/* colexec.NewHashAggregator + hashAggregator.Next() (hash_aggregator.go:194,
 * hash_aggregator_tmpl.go:400). */
typedef struct crgpu_ha crgpu_ha;
crgpu_status crgpu_agg_new(const crgpu_agg_spec *spec, crgpu_ha **out);
crgpu_status crgpu_agg_append(crgpu_ha *h, const crgpu_batch *in);
crgpu_status crgpu_agg_finish(crgpu_ha *h); /* input exhausted */
crgpu_status crgpu_agg_next_output(crgpu_ha *h, crgpu_out_batch *out);
crgpu_status crgpu_agg_reset(crgpu_ha *h);
void crgpu_agg_free(crgpu_ha *h);
/* colexecop.BufferingInMemoryOperator for the aggregator (colexecop/operator.go:167-196; hashAggregator.ExportBuffered /
 * ReleaseBeforeExport / ReleaseAfterExport, sql/colexec/hash_aggregator.go:455-500; asserted by the disk spiller at
 * colbuilder/execplan.go:1135).  The reference keeps every input tuple in inputTrackingState when it is planned under a
 * disk spiller (hash_aggregator.go:138-153, 221-231): crgpu_agg_track_input(h, 1), called before the first append,
 * makes the shell keep the staged input on the host.  On CRGPU_ERR_OOM the Go shell calls release_before_export (device
 * state back to the pool), then export_buffered until length == 0 (the whole input so far, in arrival order, including
 * rows not yet pushed to the device), then release_after_export. */
crgpu_status crgpu_agg_track_input(crgpu_ha *h, int32_t on);
crgpu_status crgpu_agg_export_buffered(crgpu_ha *h, crgpu_out_batch *out);
crgpu_status crgpu_agg_release_before_export(crgpu_ha *h);
crgpu_status crgpu_agg_release_after_export(crgpu_ha *h);

/* colflow.NewHashRouter + processNextBatch + routerOutputOp.Next
 * (sql/colflow/routers.go:490,683,238). */
typedef struct crgpu_router crgpu_router;
crgpu_status crgpu_route_new(const int32_t *types, int32_t n_cols, const int32_t *hash_cols, int32_t n_hash,
                             int32_t num_outputs, crgpu_router **out);
crgpu_status crgpu_route_partition(crgpu_router *r, const crgpu_batch *in); /* processNextBatch */
crgpu_status crgpu_route_finish(crgpu_router *r);                          /* zero-length batch = EOF to all */
crgpu_status crgpu_route_next_output(crgpu_router *r, int32_t output, crgpu_out_batch *out);
void crgpu_route_free(crgpu_router *r);

/* colexec.NewUnorderedDistinct + UnorderedDistinct.Next (sql/colexec/unordered_distinct.go:22-40,94-121;
 * constructed at colbuilder/execplan.go:1207).  The operator streams: next_output pushes whatever is staged through
 * the device when nothing is queued, so output batches hold the new tuples of the input seen so far, in input order.
 * CRGPU_ERR_DUPLICATE from append/finish/next_output (error_on_dup plans) => the shell raises the plan's message. */
typedef struct crgpu_ud crgpu_ud;
crgpu_status crgpu_distinct_new(const int32_t *types, int32_t n_cols, const int32_t *distinct_cols, int32_t n_distinct,
                                int32_t nulls_are_distinct, int32_t error_on_dup, crgpu_ud **out);
crgpu_status crgpu_distinct_append(crgpu_ud *h, const crgpu_batch *in);
crgpu_status crgpu_distinct_finish(crgpu_ud *h); /* input exhausted */
crgpu_status crgpu_distinct_next_output(crgpu_ud *h, crgpu_out_batch *out);
crgpu_status crgpu_distinct_reset(crgpu_ud *h);
void crgpu_distinct_free(crgpu_ud *h);
/* UnorderedDistinct.ExportBuffered (sql/colexec/unordered_distinct.go:123-134; asserted at colbuilder/execplan.go:1216):
 * the new tuples of the input seen so far that were not emitted yet, in input order; length 0 when there are none.  The
 * table of seen tuples stays (the external distinct filters against it), so ReleaseBeforeExport / ReleaseAfterExport
 * are no-ops in the reference (:136-150) and the Go shell implements them empty. */
crgpu_status crgpu_distinct_export_buffered(crgpu_ud *h, crgpu_out_batch *out);

#ifdef __cplusplus
}
#endif
#endif /* CRGPU_H */
