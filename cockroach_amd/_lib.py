"""ctypes loader for libcrgpu.so (the C-ABI declared in include/crgpu.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is
visible, every entry point raises.  This mirrors the Go-side cgo stub of
INTEGRATION.md one-to-one (same structs, same field order).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcrgpu.so")

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_OOM, ERR_INT_OUT_OF_RANGE, ERR_HIP, ERR_NO_DEVICE, ERR_STATE, ERR_DUPLICATE, ERR_CANCELLED = range(10)


class CrgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"crgpu status {code}: {msg}")
        self.code = code


class DCol(C.Structure):  # crgpu_dcol
    _fields_ = [("type", C.c_int32), ("_pad", C.c_int32), ("data", C.c_void_p), ("valid", C.c_void_p),
                ("offsets", C.c_void_p)]


class DevResult(C.Structure):  # crgpu_dev_result
    _fields_ = [("n", C.c_int64), ("ncols", C.c_int32), ("_pad", C.c_int32), ("cols", C.POINTER(DCol))]


class JoinSpec(C.Structure):  # crgpu_join_spec
    _fields_ = [("join_type", C.c_int32), ("right_eq_cols_are_key", C.c_int32), ("n_left_cols", C.c_int32),
                ("n_right_cols", C.c_int32), ("left_types", C.POINTER(C.c_int32)),
                ("right_types", C.POINTER(C.c_int32)), ("n_eq", C.c_int32), ("n_out_cols", C.c_int32),
                ("left_eq", C.POINTER(C.c_int32)), ("right_eq", C.POINTER(C.c_int32)),
                ("out_cols", C.POINTER(C.c_int32))]


class HandleStats(C.Structure):  # crgpu_handle_stats_t
    _fields_ = [(n, C.c_int64) for n in ("calls", "busy_ns", "rows_in", "rows_out", "device_bytes", "pinned_bytes")]


class AggDesc(C.Structure):  # crgpu_agg_desc
    _fields_ = [("func", C.c_int32), ("col", C.c_int32), ("filter_col", C.c_int32), ("distinct", C.c_int32)]


class AggSpec(C.Structure):  # crgpu_agg_spec
    _fields_ = [("n_cols", C.c_int32), ("n_group", C.c_int32), ("types", C.POINTER(C.c_int32)),
                ("group_cols", C.POINTER(C.c_int32)), ("n_aggs", C.c_int32), ("n_ordered", C.c_int32),
                ("aggs", C.POINTER(AggDesc))]


class AggVcol(C.Structure):  # crgpu_agg_vcol
    _fields_ = [("kind", C.c_int32), ("op", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("cst", C.c_int64)]


class GjoinAgg(C.Structure):  # crgpu_gjoin_agg
    _fields_ = [("func", C.c_int32), ("input", C.c_int32)]


class GjoinSpec(C.Structure):  # crgpu_gjoin_spec
    _fields_ = [("join_type", C.c_int32), ("key_type", C.c_int32), ("probe_payload_type", C.c_int32),
                ("build_payload_type", C.c_int32), ("n_aggs", C.c_int32), ("_pad", C.c_int32), ("aggs", C.POINTER(GjoinAgg))]


class GraceOpts(C.Structure):  # crgpu_grace_opts
    _fields_ = [("mem_limit", C.c_int64), ("host_limit", C.c_int64), ("spill_dir", C.c_char_p), ("num_buckets", C.c_int32),
                ("max_repartitions", C.c_int32)]


class GraceStats(C.Structure):  # crgpu_grace_stats
    _fields_ = [("partitions_processed", C.c_int64), ("repartitions", C.c_int64), ("fallback_partitions", C.c_int64),
                ("bytes_spilled", C.c_int64), ("bytes_spilled_disk", C.c_int64)]


class FetchSpec(C.Structure):  # crgpu_fetch_spec
    _fields_ = [("key_prefix_len", C.c_int32), ("n_key_cols", C.c_int32), ("key_types", C.POINTER(C.c_int32)),
                ("key_dirs", C.POINTER(C.c_int32)), ("key_out", C.POINTER(C.c_int32)), ("n_val_cols", C.c_int32),
                ("n_out_cols", C.c_int32), ("val_col_ids", C.POINTER(C.c_int32)), ("val_types", C.POINTER(C.c_int32)),
                ("val_out", C.POINTER(C.c_int32)), ("out_types", C.POINTER(C.c_int32)),
                ("max_keys_per_row", C.c_int32), ("max_family_id", C.c_int32), ("n_family_defaults", C.c_int32),
                ("_pad", C.c_int32), ("family_ids", C.POINTER(C.c_int32)), ("family_default_col_ids", C.POINTER(C.c_int32))]


class Col(C.Structure):  # crgpu_col (host, coldata layout)
    _fields_ = [("type", C.c_int32), ("maybe_has_nulls", C.c_int32), ("data", C.c_void_p), ("nulls", C.c_void_p),
                ("bytes_buf", C.c_void_p), ("bytes_buf_len", C.c_int64)]


class Batch(C.Structure):  # crgpu_batch
    _fields_ = [("length", C.c_int64), ("sel", C.c_void_p), ("ncols", C.c_int32), ("_pad", C.c_int32),
                ("cols", C.POINTER(Col))]


class OutCol(C.Structure):  # crgpu_out_col
    _fields_ = [("type", C.c_int32), ("has_nulls", C.c_int32), ("data", C.c_void_p), ("nulls", C.c_void_p),
                ("bytes_buf", C.c_void_p), ("bytes_buf_cap", C.c_int64), ("bytes_buf_len", C.c_int64)]


class OutBatch(C.Structure):  # crgpu_out_batch
    _fields_ = [("capacity", C.c_int64), ("length", C.c_int64), ("ncols", C.c_int32), ("flags", C.c_int32),
                ("cols", C.POINTER(OutCol))]


OUT_BORROW = 1  # crgpu_out_batch.flags: fixed-width values are handed out in place (crgpu.h)


# every symbol include/crgpu.h declares: (name, restype, argtypes)
_VP, _I32, _I64, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
SYMBOLS = [
    ("crgpu_init", _I32, [_I32]),
    ("crgpu_shutdown", None, []),
    ("crgpu_last_error", C.c_char_p, []),
    ("crgpu_mem_usage", _I64, []),
    ("crgpu_handle_mem_usage", _I32, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("crgpu_handle_stats", _I32, [_VP, C.POINTER(HandleStats)]),
    ("crgpu_live_handles", _I64, []),
    ("crgpu_mem_cached", _I64, []),
    ("crgpu_mem_trim", None, []),
    ("crgpu_cancel", _I32, [_VP]),
    ("crgpu_prof_enable", None, [_I32]),
    ("crgpu_prof_reset", None, []),
    ("crgpu_prof_get", _I32, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(_I64)]),
    ("crgpu_dev_malloc", _I32, [C.POINTER(_VP), C.c_size_t]),
    ("crgpu_dev_free", None, [_VP]),
    ("crgpu_dev_memcpy_h2d", _I32, [_VP, _VP, C.c_size_t]),
    ("crgpu_dev_memcpy_d2h", _I32, [_VP, _VP, C.c_size_t]),
    ("crgpu_dev_synchronize", _I32, []),
    ("crgpu_dev_hash_cols", _I32, [C.POINTER(DCol), _I32, _I64, _U64, _I32, _U64, _VP]),
    ("crgpu_dev_join_create", _I32, [C.POINTER(JoinSpec), C.POINTER(_VP)]),
    ("crgpu_dev_join_build", _I32, [_VP, C.POINTER(DCol), _I64]),
    ("crgpu_dev_join_probe", _I32, [_VP, C.POINTER(DCol), _I64, C.POINTER(DevResult)]),
    ("crgpu_dev_join_probe_selected", _I32, [_VP, C.POINTER(DCol), _I64, _VP, C.POINTER(DevResult)]),
    ("crgpu_dev_join_emit_right", _I32, [_VP, C.POINTER(DevResult)]),
    ("crgpu_dev_join_reset", _I32, [_VP]),
    ("crgpu_dev_join_destroy", None, [_VP]),
    ("crgpu_dev_agg_create", _I32, [C.POINTER(AggSpec), C.POINTER(_VP)]),
    ("crgpu_dev_agg_create_fused", _I32, [C.POINTER(AggSpec), _I32, C.POINTER(AggVcol), C.POINTER(_VP)]),
    ("crgpu_dev_agg_consume", _I32, [_VP, C.POINTER(DCol), _I64]),
    ("crgpu_dev_agg_finish", _I32, [_VP, C.POINTER(DevResult)]),
    ("crgpu_dev_agg_emit_ready", _I32, [_VP, C.POINTER(DevResult)]),
    ("crgpu_dev_agg_pending_rows", _I32, [_VP, C.POINTER(C.c_int64)]),
    ("crgpu_dev_agg_reset", _I32, [_VP]),
    ("crgpu_dev_agg_destroy", None, [_VP]),
    ("crgpu_dev_partitioner_create", _I32, [C.POINTER(_I32), _I32, C.POINTER(_I32), _I32, _U64, _I32, C.POINTER(_VP)]),
    ("crgpu_dev_partition", _I32, [_VP, C.POINTER(DCol), _I64, C.POINTER(DevResult), C.POINTER(_I64)]),
    ("crgpu_dev_partitioner_destroy", None, [_VP]),
    ("crgpu_exchange_unique_id", _I32, [_VP]),
    ("crgpu_exchange_create", _I32, [_VP, _I32, _I32, C.POINTER(_VP)]),
    ("crgpu_exchange_all_to_all", _I32, [_VP, C.POINTER(DCol), _I32, C.POINTER(_I64), C.POINTER(DevResult), C.POINTER(_I64)]),
    ("crgpu_exchange_clone", _I32, [_VP, C.POINTER(_VP)]),
    ("crgpu_exchange_broadcast", _I32, [_VP, C.POINTER(DCol), _I32, _I64, C.POINTER(DevResult), C.POINTER(_I64)]),
    ("crgpu_exchange_destroy", None, [_VP]),
    ("crgpu_dev_partition_noalloc", _I32, [C.POINTER(_I32), _I32, C.POINTER(_I32), _I32, _U64, _I32, C.POINTER(DCol), _I64,
                                           C.POINTER(_VP), _VP]),
    ("crgpu_dev_select_i64_const", _I32, [_I32, C.POINTER(DCol), _I64, _I64, _VP]),
    ("crgpu_dev_project_f64", _I32, [_I32, C.POINTER(DCol), C.c_double, C.POINTER(DCol), C.c_double, _I64, _VP, _VP]),
    ("crgpu_dev_select_bytes_const", _I32, [_I32, C.POINTER(DCol), C.c_char_p, _I32, _I64, _VP]),
    ("crgpu_dev_extract_year", _I32, [C.POINTER(DCol), _I64, _VP]),
    ("crgpu_dev_gjoin_create", _I32, [C.POINTER(GjoinSpec), C.POINTER(_VP)]),
    ("crgpu_dev_gjoin_build", _I32, [_VP, C.POINTER(DCol), C.POINTER(DCol), _I64]),
    ("crgpu_dev_gjoin_probe", _I32, [_VP, C.POINTER(DCol), C.POINTER(DCol), _I64, C.POINTER(DevResult)]),
    ("crgpu_dev_gjoin_destroy", None, [_VP]),
    ("crgpu_dev_grace_join_create", _I32, [C.POINTER(JoinSpec), C.POINTER(GraceOpts), C.POINTER(_VP)]),
    ("crgpu_dev_grace_agg_create", _I32, [C.POINTER(AggSpec), C.POINTER(GraceOpts), C.POINTER(_VP)]),
    ("crgpu_dev_grace_add", _I32, [_VP, _I32, C.POINTER(DCol), _I64]),
    ("crgpu_dev_grace_next", _I32, [_VP, C.POINTER(DevResult), C.POINTER(_I32)]),
    ("crgpu_dev_grace_stats", _I32, [_VP, C.POINTER(GraceStats)]),
    ("crgpu_dev_grace_destroy", None, [_VP]),
    ("crgpu_dev_arrow_create", _I32, [C.POINTER(_I32), _I32, C.POINTER(_VP)]),
    ("crgpu_dev_arrow_read", _I32, [_VP, _VP, _I64, C.POINTER(DevResult)]),
    ("crgpu_dev_arrow_write", _I32, [_VP, C.POINTER(DCol), _I64, C.POINTER(_VP), C.POINTER(_I64)]),
    ("crgpu_dev_arrow_destroy", None, [_VP]),
    ("crgpu_dev_deselector_create", _I32, [C.POINTER(_I32), _I32, C.POINTER(_VP)]),
    ("crgpu_dev_deselect", _I32, [_VP, C.POINTER(DCol), _I64, _VP, C.POINTER(DevResult)]),
    ("crgpu_dev_deselector_destroy", None, [_VP]),
    ("crgpu_dev_distinct_create", _I32, [C.POINTER(_I32), _I32, C.POINTER(_I32), _I32, _I32, _I32, C.POINTER(_VP)]),
    ("crgpu_dev_distinct_consume", _I32, [_VP, C.POINTER(DCol), _I64, C.POINTER(DevResult)]),
    ("crgpu_dev_distinct_reset", _I32, [_VP]),
    ("crgpu_dev_distinct_destroy", None, [_VP]),
    ("crgpu_dev_decoder_create", _I32, [C.POINTER(FetchSpec), C.POINTER(_VP)]),
    ("crgpu_dev_decode_kv", _I32, [_VP, _VP, C.POINTER(_I64), _I64, C.POINTER(DevResult)]),
    ("crgpu_dev_decoder_destroy", None, [_VP]),
    ("crgpu_hash_cols", _I32, [C.POINTER(Col), _I32, _I64, _VP, _U64, _I32, _U64, _VP]),
    ("crgpu_hj_new", _I32, [C.POINTER(JoinSpec), C.POINTER(_VP)]),
    ("crgpu_hj_build_append", _I32, [_VP, C.POINTER(Batch)]),
    ("crgpu_hj_build_finish", _I32, [_VP]),
    ("crgpu_hj_probe_append", _I32, [_VP, C.POINTER(Batch)]),
    ("crgpu_hj_probe_finish", _I32, [_VP]),
    ("crgpu_hj_next_output", _I32, [_VP, C.POINTER(OutBatch)]),
    ("crgpu_hj_export_buffered", _I32, [_VP, C.POINTER(OutBatch)]),
    ("crgpu_hj_reset", _I32, [_VP]),
    ("crgpu_hj_free", None, [_VP]),
    ("crgpu_agg_new", _I32, [C.POINTER(AggSpec), C.POINTER(_VP)]),
    ("crgpu_agg_append", _I32, [_VP, C.POINTER(Batch)]),
    ("crgpu_agg_finish", _I32, [_VP]),
    ("crgpu_agg_next_output", _I32, [_VP, C.POINTER(OutBatch)]),
    ("crgpu_agg_reset", _I32, [_VP]),
    ("crgpu_agg_free", None, [_VP]),
    ("crgpu_agg_track_input", _I32, [_VP, _I32]),
    ("crgpu_agg_export_buffered", _I32, [_VP, C.POINTER(OutBatch)]),
    ("crgpu_agg_release_before_export", _I32, [_VP]),
    ("crgpu_agg_release_after_export", _I32, [_VP]),
    ("crgpu_route_new", _I32, [C.POINTER(_I32), _I32, C.POINTER(_I32), _I32, _I32, C.POINTER(_VP)]),
    ("crgpu_route_partition", _I32, [_VP, C.POINTER(Batch)]),
    ("crgpu_route_finish", _I32, [_VP]),
    ("crgpu_route_next_output", _I32, [_VP, _I32, C.POINTER(OutBatch)]),
    ("crgpu_route_free", None, [_VP]),
    ("crgpu_distinct_new", _I32, [C.POINTER(_I32), _I32, C.POINTER(_I32), _I32, _I32, _I32, C.POINTER(_VP)]),
    ("crgpu_distinct_append", _I32, [_VP, C.POINTER(Batch)]),
    ("crgpu_distinct_finish", _I32, [_VP]),
    ("crgpu_distinct_next_output", _I32, [_VP, C.POINTER(OutBatch)]),
    ("crgpu_distinct_reset", _I32, [_VP]),
    ("crgpu_distinct_free", None, [_VP]),
    ("crgpu_distinct_export_buffered", _I32, [_VP, C.POINTER(OutBatch)]),
]

_lib = None


def load():
    """Load libcrgpu.so and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CrgpuError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() "
                         "(make -C cockroach_amd/csrc); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != OK:
        msg = load().crgpu_last_error()
        raise CrgpuError(status, msg.decode() if msg else "")
