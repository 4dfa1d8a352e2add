"""Thin Python face of the bulk device API (include/crgpu.h, layer 1).

Used by bench.py and the GPU parity tests.  It only moves pointers around:
every operation is a call into libcrgpu.so through ctypes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import coldata as cd


class DeviceBuffer:
    """Raw HBM allocation owned by Python (crgpu_dev_malloc / crgpu_dev_free)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_malloc(C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value

    @staticmethod
    def from_numpy(arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        b = DeviceBuffer(arr.nbytes)
        if arr.nbytes:
            _lib.check(_lib.load().crgpu_dev_memcpy_h2d(b.ptr, arr.ctypes.data, arr.nbytes))
        return b

    def to_numpy(self, dtype, count) -> np.ndarray:
        return download(self.ptr, dtype, count)

    def free(self):
        if self.ptr:
            _lib.load().crgpu_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # interpreter shutdown
            pass


def download(ptr, dtype, count) -> np.ndarray:
    out = np.empty(int(count), dtype=dtype)
    if out.nbytes:
        _lib.check(_lib.load().crgpu_dev_memcpy_d2h(out.ctypes.data, ptr, out.nbytes))
    return out


class DeviceColumn:
    """One device-resident column (crgpu_dcol) plus whatever keeps its memory alive."""

    def __init__(self, typ, n, data_ptr, valid_ptr=None, offsets_ptr=None, keep=None):
        self.type, self.n = typ, int(n)
        self.data_ptr, self.valid_ptr, self.offsets_ptr = data_ptr, valid_ptr, offsets_ptr
        self.keep = keep

    def dcol(self) -> _lib.DCol:
        return _lib.DCol(self.type, 0, self.data_ptr, self.valid_ptr, self.offsets_ptr)


def upload(vec: cd.Vec) -> DeviceColumn:
    keep = []
    data = DeviceBuffer.from_numpy(vec.data)
    keep.append(data)
    valid = offs = None
    if vec.valid is not None:
        valid = DeviceBuffer.from_numpy(vec.valid)
        keep.append(valid)
    if vec.type == cd.BYTES:
        offs = DeviceBuffer.from_numpy(vec.offsets)
        keep.append(offs)
    return DeviceColumn(vec.type, len(vec), data.ptr, valid.ptr if valid else None, offs.ptr if offs else None, keep)


def from_torch(typ, tensor, valid=None, offsets=None) -> DeviceColumn:
    """Wrap torch CUDA tensors (no copy)."""
    n = (offsets.numel() - 1) if typ == cd.BYTES else tensor.numel()
    return DeviceColumn(typ, n, tensor.data_ptr(), None if valid is None else valid.data_ptr(),
                        None if offsets is None else offsets.data_ptr(), keep=(tensor, valid, offsets))


def dcol_array(cols):
    arr = (_lib.DCol * max(len(cols), 1))()
    for i, c in enumerate(cols):
        arr[i] = c.dcol()
    return arr


def result_to_vecs(res: _lib.DevResult):
    """Copy a library-owned device result back to host Vecs."""
    n, out = int(res.n), []
    for i in range(res.ncols):
        c = res.cols[i]
        valid = download(c.valid, np.uint8, (n + 7) // 8) if c.valid else None
        if c.type == cd.BYTES:
            offs = download(c.offsets, np.uint32, n + 1)
            total = int(offs[-1]) if n else 0
            data = download(c.data, np.uint8, total)
            out.append(cd.Vec(cd.BYTES, data, valid, offs))
        else:
            out.append(cd.Vec(c.type, download(c.data, cd.NP_DTYPE[c.type], n), valid))
    return out


def result_columns(res: _lib.DevResult):
    """Device-resident views of a result (valid until the next call on the handle)."""
    return [DeviceColumn(res.cols[i].type, res.n, res.cols[i].data, res.cols[i].valid, res.cols[i].offsets)
            for i in range(res.ncols)]


def _i32arr(xs):
    xs = list(xs)
    return (C.c_int32 * max(len(xs), 1))(*xs)


def hash_cols(cols, n=None, init=1, width=64, nbuckets=0) -> np.ndarray:
    """colexechash.rehash over device columns; returns the hashes on the host."""
    n = cols[0].n if n is None else n
    dt = np.uint32 if width == 32 else np.uint64
    out = DeviceBuffer(max(n, 1) * np.dtype(dt).itemsize)
    _lib.check(_lib.load().crgpu_dev_hash_cols(dcol_array(cols), len(cols), n, init, width, nbuckets, out.ptr))
    return out.to_numpy(dt, n)


class Join:
    """crgpu_join handle: build once, probe chunk by chunk, then emit_right."""

    def __init__(self, join_type, left_types, right_types, left_eq, right_eq, right_eq_cols_are_key=False,
                 out_cols=None):
        self._keep = (_i32arr(left_types), _i32arr(right_types), _i32arr(left_eq), _i32arr(right_eq),
                      _i32arr(out_cols or []))
        spec = _lib.JoinSpec(join_type, int(bool(right_eq_cols_are_key)), len(left_types), len(right_types),
                             self._keep[0], self._keep[1], len(left_eq), -1 if out_cols is None else len(out_cols),
                             self._keep[2], self._keep[3], self._keep[4])
        self.h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_join_create(C.byref(spec), C.byref(self.h)))
        self._build_cols = None

    def build(self, cols, n=None):
        self._build_cols = cols  # borrowed by the library until reset/destroy
        n = (cols[0].n if cols else 0) if n is None else n
        _lib.check(_lib.load().crgpu_dev_join_build(self.h, dcol_array(cols), n))

    def probe(self, cols, n=None) -> _lib.DevResult:
        n = (cols[0].n if cols else 0) if n is None else n
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_dev_join_probe(self.h, dcol_array(cols), n, C.byref(res)))
        return res

    def probe_selected(self, cols, keep, n=None) -> _lib.DevResult:
        """Probe the rows of `cols` whose byte in `keep` (a BOOL DeviceColumn from select_*) is non-zero."""
        n = (cols[0].n if cols else 0) if n is None else n
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_dev_join_probe_selected(self.h, dcol_array(cols), n, keep.data_ptr if keep is not None else None,
                                                             C.byref(res)))
        return res

    def emit_right(self) -> _lib.DevResult:
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_dev_join_emit_right(self.h, C.byref(res)))
        return res

    def reset(self):
        _lib.check(_lib.load().crgpu_dev_join_reset(self.h))

    def close(self):
        if self.h:
            _lib.load().crgpu_dev_join_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Agg:
    """crgpu_agg handle (online hash aggregation)."""

    def __init__(self, types, group_cols, aggs, n_ordered=0):
        """aggs: list of (func, col or -1, filter_col or -1[, distinct]); n_ordered: the input is sorted on the first
        n_ordered grouping columns (AggregatorSpec.OrderedGroupCols) -- see emit_ready."""
        self._types, self._groups = _i32arr(types), _i32arr(group_cols)
        self._aggs = (_lib.AggDesc * max(len(aggs), 1))()
        for i, a in enumerate(aggs):
            self._aggs[i] = _lib.AggDesc(a[0], a[1], a[2] if len(a) > 2 else -1, int(a[3]) if len(a) > 3 else 0)
        spec = _lib.AggSpec(len(types), len(group_cols), self._types, self._groups, len(aggs), n_ordered, self._aggs)
        self.h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_agg_create(C.byref(spec), C.byref(self.h)))

    def consume(self, cols, n=None):
        n = (cols[0].n if cols else 0) if n is None else n
        _lib.check(_lib.load().crgpu_dev_agg_consume(self.h, dcol_array(cols), n))

    def finish(self) -> _lib.DevResult:
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_dev_agg_finish(self.h, C.byref(res)))
        return res

    def emit_ready(self) -> _lib.DevResult:
        """Partially ordered input: the next piece of completed groups (n == 0: none yet / no more)."""
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_dev_agg_emit_ready(self.h, C.byref(res)))
        return res

    def reset(self):
        _lib.check(_lib.load().crgpu_dev_agg_reset(self.h))

    def close(self):
        if self.h:
            _lib.load().crgpu_dev_agg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Partitioner:
    """crgpu_partitioner handle: hash-repartition rows into num_outputs contiguous regions."""

    def __init__(self, types, hash_cols, num_outputs, init=1):
        self._types, self._hc = _i32arr(types), _i32arr(hash_cols)
        self.num_outputs = num_outputs
        self.h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_partitioner_create(self._types, len(types), self._hc, len(hash_cols), init,
                                                            num_outputs, C.byref(self.h)))

    def partition(self, cols, n=None):
        n = (cols[0].n if cols else 0) if n is None else n
        res = _lib.DevResult()
        counts = (C.c_int64 * self.num_outputs)()
        _lib.check(_lib.load().crgpu_dev_partition(self.h, dcol_array(cols), n, C.byref(res), counts))
        return res, np.array(list(counts), dtype=np.int64)

    def close(self):
        if self.h:
            _lib.load().crgpu_dev_partitioner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Exchange:
    """crgpu_exchange handle: RCCL all-to-all of partitioned regions between the GPUs of a node."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * Exchange.ID_BYTES)()
        _lib.check(_lib.load().crgpu_exchange_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid: bytes, world: int, rank: int, _clone_of=None):
        self.world, self.rank = world, rank
        self.h = C.c_void_p()
        if _clone_of is not None:
            _lib.check(_lib.load().crgpu_exchange_clone(_clone_of.h, C.byref(self.h)))
            return
        self._uid = (C.c_uint8 * Exchange.ID_BYTES)(*uid)
        _lib.check(_lib.load().crgpu_exchange_create(self._uid, world, rank, C.byref(self.h)))

    def clone(self) -> "Exchange":
        """A second handle on the same communicator (own stream, own result buffers): crgpu_exchange_clone."""
        return Exchange(None, self.world, self.rank, _clone_of=self)

    def all_to_all(self, cols, send_counts):
        """cols: partitioned DeviceColumns; returns (DevResult of the received rows, recv_counts)."""
        sc = (C.c_int64 * self.world)(*[int(v) for v in send_counts])
        rc = (C.c_int64 * self.world)()
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_exchange_all_to_all(self.h, dcol_array(cols), len(cols), sc, C.byref(res), rc))
        return res, np.array(list(rc), dtype=np.int64)

    def broadcast(self, cols, n=None):
        """cols: this rank's rows; returns (DevResult of all ranks' rows in rank order, recv_counts)."""
        n = (cols[0].n if cols else 0) if n is None else n
        rc = (C.c_int64 * self.world)()
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_exchange_broadcast(self.h, dcol_array(cols), len(cols), n, C.byref(res), rc))
        return res, np.array(list(rc), dtype=np.int64)

    def close(self):
        if self.h:
            _lib.load().crgpu_exchange_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


CMP = {"EQ": 0, "NE": 1, "LT": 2, "LE": 3, "GT": 4, "GE": 5}
BIN = {"+": 0, "-": 1, "*": 2, "*(1-)": 3, "*(1+)": 4}


def select_i64_const(cmp, col: DeviceColumn, cst, out=None) -> DeviceColumn:
    """crgpu_dev_select_i64_const: BOOL column, 1 where `col <cmp> cst` holds (NULL never passes).
    `out`: a DeviceBuffer of >= n bytes to reuse."""
    out = out or DeviceBuffer(max(col.n, 1))
    d = col.dcol()
    _lib.check(_lib.load().crgpu_dev_select_i64_const(CMP[cmp], C.byref(d), int(cst), col.n, out.ptr))
    return DeviceColumn(cd.BOOL, col.n, out.ptr, None, None, keep=(out, col))


def project_f64(op, a, b, out=None) -> DeviceColumn:
    """crgpu_dev_project_f64: a <op> b over FLOAT64 columns; either operand may be a Python float (constant).
    `out`: a DeviceBuffer of >= 8 n bytes to reuse."""
    cols = [x for x in (a, b) if isinstance(x, DeviceColumn)]
    n = cols[0].n
    out = out or DeviceBuffer(max(n, 2) * 8)
    nullable = any(c.valid_ptr for c in cols)
    valid = DeviceBuffer((n + 7) // 8 + 1) if nullable else None
    da = a.dcol() if isinstance(a, DeviceColumn) else None
    db = b.dcol() if isinstance(b, DeviceColumn) else None
    _lib.check(_lib.load().crgpu_dev_project_f64(BIN[op], C.byref(da) if da is not None else None, 0.0 if da is not None else float(a),
                                                 C.byref(db) if db is not None else None, 0.0 if db is not None else float(b),
                                                 n, out.ptr, valid.ptr if valid else None))
    return DeviceColumn(cd.FLOAT64, n, out.ptr, valid.ptr if valid else None, None, keep=(out, valid, a, b))


BCMP = {"EQ": 0, "NE": 1, "CONTAINS": 6, "NOT CONTAINS": 7}


def select_bytes_const(cmp, col: DeviceColumn, cst: bytes, out=None) -> DeviceColumn:
    """crgpu_dev_select_bytes_const: BOOL column, 1 where `col <cmp> cst` holds (EQ, NE, CONTAINS = LIKE '%cst%')."""
    out = out or DeviceBuffer(max(col.n, 1))
    d = col.dcol()
    _lib.check(_lib.load().crgpu_dev_select_bytes_const(BCMP[cmp], C.byref(d), bytes(cst), len(cst), col.n, out.ptr))
    return DeviceColumn(cd.BOOL, col.n, out.ptr, None, None, keep=(out, col))


def extract_year(col: DeviceColumn, out=None) -> DeviceColumn:
    """crgpu_dev_extract_year: EXTRACT(year FROM date) as FLOAT64."""
    out = out or DeviceBuffer(max(col.n, 2) * 8)
    d = col.dcol()
    _lib.check(_lib.load().crgpu_dev_extract_year(C.byref(d), col.n, out.ptr))
    return DeviceColumn(cd.FLOAT64, col.n, out.ptr, col.valid_ptr, None, keep=(out, col))


class Deselector:
    """crgpu_deselector handle: the rows a BOOL column keeps, as dense device columns (deselectorOp)."""

    def __init__(self, types):
        self._types = _i32arr(types)
        self.h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_deselector_create(self._types, len(types), C.byref(self.h)))

    def deselect(self, cols, keep: DeviceColumn, n=None) -> _lib.DevResult:
        n = (cols[0].n if cols else 0) if n is None else n
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_dev_deselect(self.h, dcol_array(cols), n, keep.data_ptr, C.byref(res)))
        return res

    def close(self):
        if self.h:
            _lib.load().crgpu_dev_deselector_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupJoin:
    """crgpu_gjoin handle: the fused hash group-join (aggregate while probing).  aggs: [(func, input)] with input
    0 = the key, 1 = the probe side's payload, 2 = the build side's payload.  Raises CrgpuError(ERR_UNSUPPORTED) for
    shapes outside the fused form: compose Join -> Agg then."""

    def __init__(self, aggs, probe_payload_type=0, build_payload_type=0, join_type=0, key_type=cd.INT64):
        self._aggs = (_lib.GjoinAgg * max(len(aggs), 1))()
        for i, (f, inp) in enumerate(aggs):
            self._aggs[i] = _lib.GjoinAgg(f, inp)
        spec = _lib.GjoinSpec(join_type, key_type, probe_payload_type, build_payload_type, len(aggs), 0, self._aggs)
        self.h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_gjoin_create(C.byref(spec), C.byref(self.h)))

    def build(self, key, payload=None, n=None):
        n = key.n if n is None else n
        self._keep = (key, payload)
        pd = payload.dcol() if payload is not None else None
        _lib.check(_lib.load().crgpu_dev_gjoin_build(self.h, C.byref(key.dcol()), C.byref(pd) if pd is not None else None, n))

    def probe(self, key, payload=None, n=None) -> _lib.DevResult:
        n = key.n if n is None else n
        res = _lib.DevResult()
        pd = payload.dcol() if payload is not None else None
        _lib.check(_lib.load().crgpu_dev_gjoin_probe(self.h, C.byref(key.dcol()), C.byref(pd) if pd is not None else None, n,
                                                     C.byref(res)))
        return res

    def close(self):
        if self.h:
            _lib.load().crgpu_dev_gjoin_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Grace:
    """crgpu_grace handle: the Grace-partitioned (external) hash join / hash aggregation (hashBasedPartitioner).
    add() partitions device rows and spills them; results() yields the output piece by piece as host Vecs."""

    def __init__(self, handle, keep):
        self.h, self._keep = handle, keep

    @staticmethod
    def _opts(mem_limit, num_buckets, host_limit, spill_dir, max_repartitions):
        return _lib.GraceOpts(int(mem_limit), int(host_limit), spill_dir.encode() if spill_dir else None, int(num_buckets),
                              int(max_repartitions))

    @classmethod
    def join(cls, join_type, left_types, right_types, left_eq, right_eq, right_eq_cols_are_key=False, mem_limit=0,
             num_buckets=0, host_limit=0, spill_dir=None, max_repartitions=0):
        keep = (_i32arr(left_types), _i32arr(right_types), _i32arr(left_eq), _i32arr(right_eq), _i32arr([]))
        spec = _lib.JoinSpec(join_type, int(bool(right_eq_cols_are_key)), len(left_types), len(right_types), keep[0], keep[1],
                             len(left_eq), -1, keep[2], keep[3], keep[4])
        opts = cls._opts(mem_limit, num_buckets, host_limit, spill_dir, max_repartitions)
        h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_grace_join_create(C.byref(spec), C.byref(opts), C.byref(h)))
        return cls(h, keep)

    @classmethod
    def agg(cls, types, group_cols, aggs, mem_limit=0, num_buckets=0, host_limit=0, spill_dir=None, max_repartitions=0):
        descs = (_lib.AggDesc * max(len(aggs), 1))()
        for i, a in enumerate(aggs):
            descs[i] = _lib.AggDesc(a[0], a[1], a[2] if len(a) > 2 else -1, 0)
        keep = (_i32arr(types), _i32arr(group_cols), descs)
        spec = _lib.AggSpec(len(types), len(group_cols), keep[0], keep[1], len(aggs), 0, descs)
        opts = cls._opts(mem_limit, num_buckets, host_limit, spill_dir, max_repartitions)
        h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_grace_agg_create(C.byref(spec), C.byref(opts), C.byref(h)))
        return cls(h, keep)

    def add(self, input_idx, cols, n=None):
        n = (cols[0].n if cols else 0) if n is None else n
        _lib.check(_lib.load().crgpu_dev_grace_add(self.h, input_idx, dcol_array(cols), n))

    def results(self):
        while True:
            res, done = _lib.DevResult(), C.c_int32(0)
            _lib.check(_lib.load().crgpu_dev_grace_next(self.h, C.byref(res), C.byref(done)))
            if done.value:
                return
            yield result_to_vecs(res)

    def stats(self) -> _lib.GraceStats:
        st = _lib.GraceStats()
        _lib.check(_lib.load().crgpu_dev_grace_stats(self.h, C.byref(st)))
        return st

    def close(self):
        if self.h:
            _lib.load().crgpu_dev_grace_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ArrowSerde:
    """crgpu_arrow handle: serialized RecordBatch messages (colserde wire format) <-> device columns."""

    def __init__(self, types):
        self._types = _i32arr(types)
        self.h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_arrow_create(self._types, len(types), C.byref(self.h)))

    def read(self, msg: bytes) -> _lib.DevResult:
        buf = np.frombuffer(msg, dtype=np.uint8)
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_dev_arrow_read(self.h, buf.ctypes.data if buf.size else None, buf.size, C.byref(res)))
        return res

    def write(self, cols, n=None) -> bytes:
        n = (cols[0].n if cols else 0) if n is None else n
        out, ln = C.c_void_p(), C.c_int64()
        _lib.check(_lib.load().crgpu_dev_arrow_write(self.h, dcol_array(cols), n, C.byref(out), C.byref(ln)))
        return C.string_at(out.value, ln.value)

    def close(self):
        if self.h:
            _lib.load().crgpu_dev_arrow_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FilteredAgg:
    """Hash aggregation under a WHERE clause without moving rows: the selection's BOOL column becomes the FILTER of
    every aggregate (group-key ANY_NOT_NULLs excepted) plus a hidden COUNT_ROWS, and groups that no passing row
    reached are dropped from the result.  Same result as `sel -> hashAggregator` in the reference's plan."""

    def __init__(self, types, group_cols, aggs):
        self.where = len(types)
        self.naggs = len(aggs)
        rewritten = []
        for a in aggs:
            f, c = a[0], a[1]
            flt = a[2] if len(a) > 2 else -1
            if flt >= 0:
                raise _lib.CrgpuError(_lib.ERR_UNSUPPORTED, "FilteredAgg: an aggregate with its own FILTER")
            rewritten.append((f, c, -1) if (f == 0 and c in group_cols) else (f, c, self.where))
        rewritten.append((14, -1, self.where))  # hidden COUNT_ROWS of the passing rows
        self.agg = Agg(list(types) + [cd.BOOL], group_cols, rewritten)

    def consume(self, cols, where_col, n=None):
        self.agg.consume(list(cols) + [where_col], n)

    def finish_vecs(self):
        vecs = result_to_vecs(self.agg.finish())
        keep = np.flatnonzero(vecs[-1].data > 0)
        return [v.take(keep) for v in vecs[:self.naggs]]

    def reset(self):
        self.agg.reset()

    def close(self):
        self.agg.close()


PROJ_OP = {"+": 0, "-": 1, "*": 2, "*(1-)": 3, "*(1+)": 4}
CMP_OP = {"EQ": 0, "NE": 1, "LT": 2, "LE": 3, "GT": 4, "GE": 5}


class FusedAgg(Agg):
    """crgpu_dev_agg_create_fused: scan -> selection -> projections -> hash aggregation in one pass.  `types` are the
    physical input columns; `vcols` define computed columns at the ordinals len(types), len(types) + 1, ...:
    ("project", op, a, b) -- FLOAT64, op in PROJ_OP -- or ("select", cmp, a, constant) -- BOOL, cmp in CMP_OP.
    aggs may name them as col / filter_col.  With `where` (the ordinal of a selection) the selection becomes the FILTER
    of every aggregate (ANY_NOT_NULL of a group column excepted) plus a hidden COUNT_ROWS, and finish_vecs() drops the
    groups no passing row reached: the result of `sel -> projections -> hashAggregator` in the reference's plan."""

    def __init__(self, types, group_cols, aggs, vcols, where=None):
        self.naggs = len(aggs)
        self.where = where
        aggs = [(a[0], a[1], a[2] if len(a) > 2 else -1) for a in aggs]
        if where is not None:
            if any(a[2] >= 0 for a in aggs):
                raise _lib.CrgpuError(_lib.ERR_UNSUPPORTED, "FusedAgg: an aggregate with its own FILTER under a WHERE")
            aggs = [(f, c, -1) if (f == 0 and c in group_cols) else (f, c, where) for f, c, _ in aggs] + [(14, -1, where)]
        all_types = list(types)
        self._vc = (_lib.AggVcol * max(len(vcols), 1))()
        for i, v in enumerate(vcols):
            if v[0] == "project":
                self._vc[i] = _lib.AggVcol(0, PROJ_OP[v[1]], v[2], v[3], 0)
                all_types.append(cd.FLOAT64)
            else:
                self._vc[i] = _lib.AggVcol(1, CMP_OP[v[1]], v[2], -1, int(v[3]))
                all_types.append(cd.BOOL)
        self._types, self._groups = _i32arr(all_types), _i32arr(group_cols)
        self._aggs = (_lib.AggDesc * max(len(aggs), 1))()
        for i, a in enumerate(aggs):
            self._aggs[i] = _lib.AggDesc(a[0], a[1], a[2], 0)
        spec = _lib.AggSpec(len(all_types), len(group_cols), self._types, self._groups, len(aggs), 0, self._aggs)
        self.h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_agg_create_fused(C.byref(spec), len(vcols), self._vc, C.byref(self.h)))

    def finish_vecs(self):
        vecs = result_to_vecs(self.finish())
        if self.where is None:
            return vecs
        keep = np.flatnonzero(vecs[-1].data > 0)
        return [v.take(keep) for v in vecs[:self.naggs]]


class Distinct:
    """crgpu_distinct handle (unordered DISTINCT): consume() returns the first occurrences of new tuples, in input order."""

    def __init__(self, types, distinct_cols, nulls_are_distinct=False, error_on_dup=False):
        self._types, self._dc = _i32arr(types), _i32arr(distinct_cols)
        self.h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_distinct_create(self._types, len(types), self._dc, len(distinct_cols),
                                                         int(bool(nulls_are_distinct)), int(bool(error_on_dup)), C.byref(self.h)))

    def consume(self, cols, n=None) -> _lib.DevResult:
        n = (cols[0].n if cols else 0) if n is None else n
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_dev_distinct_consume(self.h, dcol_array(cols), n, C.byref(res)))
        return res

    def reset(self):
        _lib.check(_lib.load().crgpu_dev_distinct_reset(self.h))

    def close(self):
        if self.h:
            _lib.load().crgpu_dev_distinct_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Decoder:
    """crgpu_decoder handle: KV batch responses (device bytes) -> columns (cFetcher for one-KV-per-row indexes)."""

    def __init__(self, key_prefix_len, key_types, key_dirs, key_out, val_col_ids, val_types, val_out, out_types,
                 max_keys_per_row=0, max_family_id=0, family_defaults=()):
        """family_defaults: [(family id, default column id)] (IndexFetchSpec.FamilyDefaultColumns)."""
        fam = list(family_defaults)
        self._keep = (_i32arr(key_types), _i32arr(key_dirs), _i32arr(key_out), _i32arr(val_col_ids), _i32arr(val_types),
                      _i32arr(val_out), _i32arr(out_types), _i32arr([f for f, _ in fam]), _i32arr([c for _, c in fam]))
        k = self._keep
        spec = _lib.FetchSpec(key_prefix_len, len(list(key_types)), k[0], k[1], k[2], len(list(val_col_ids)),
                              len(list(out_types)), k[3], k[4], k[5], k[6], max_keys_per_row, max_family_id, len(fam), 0,
                              k[7], k[8])
        self.h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_decoder_create(C.byref(spec), C.byref(self.h)))

    def decode(self, repr_buf: DeviceBuffer, batch_offsets):
        offs = np.ascontiguousarray(batch_offsets, dtype=np.int64)
        res = _lib.DevResult()
        _lib.check(_lib.load().crgpu_dev_decode_kv(self.h, repr_buf.ptr, offs.ctypes.data_as(C.POINTER(C.c_int64)),
                                                   len(offs) - 1, C.byref(res)))
        return res

    def close(self):
        if self.h:
            _lib.load().crgpu_dev_decoder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prof_enable(on=True):
    _lib.load().crgpu_prof_enable(int(on))


def prof_reset():
    _lib.load().crgpu_prof_reset()


def prof_get(name):
    ms, n = C.c_double(0), C.c_int64(0)
    _lib.check(_lib.load().crgpu_prof_get(name.encode(), C.byref(ms), C.byref(n)))
    return ms.value, n.value
