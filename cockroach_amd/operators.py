"""Host-side mirror of the reference's operator interface for the hot path, over
layer (2) of the C-ABI (include/crgpu.h): the same shells a Go maintainer writes
with cgo, written here in Python because the Go toolchain is
absent from this image.  Same names, argument meaning and error behaviour as

  colexecop.Operator            Init / Next            (sql/colexecop/operator.go:22-58)
  colexecjoin.NewHashJoiner     hashJoiner.Next        (sql/colexec/colexecjoin/hashjoiner.go:886,291)
  colexec.NewHashAggregator     hashAggregator.Next    (sql/colexec/hash_aggregator.go:194)
  colflow.NewHashRouter         Run / routerOutputOp   (sql/colflow/routers.go:490,238)

Every batch crosses the boundary in the coldata layout (32-byte Bytes elements,
validity bitmaps, []int selection vectors); the library stages, uploads,
computes on the MI355X and fills caller-owned output batches.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import coldata as cd


class Operator:
    """colexecop.Operator: Init once, then Next until a zero-length batch."""

    def init(self):
        pass

    def next(self) -> cd.Batch:  # pragma: no cover - interface
        raise NotImplementedError


class BatchSource(Operator):
    """Feeds a fixed list of batches, then zero-length batches forever
    (colexectestutils.opTestInput / colexecop.RepeatableBatchSource)."""

    def __init__(self, batches, types):
        self.batches = list(batches)
        self.types = list(types)
        self.pos = 0

    def init(self):
        self.pos = 0

    def next(self):
        if self.pos >= len(self.batches):
            return cd.Batch([cd.Vec.from_values(t, []) for t in self.types], 0)
        b = self.batches[self.pos]
        self.pos += 1
        return b

    def reset(self):
        self.pos = 0


# ---------------------------------------------------------------- marshalling (what cgo does with unsafe.Pointer)
def _marshal_batch(batch: cd.Batch):
    keep = []
    ncols = len(batch.cols)
    cols = (_lib.Col * max(ncols, 1))()
    for i, v in enumerate(batch.cols):
        c = cols[i]
        c.type = v.type
        if v.type == cd.BYTES:
            elems, buf = cd.bytes_to_elements(v)
            if elems.size == 0:
                elems = np.zeros((1, 32), dtype=np.uint8)
            keep += [elems, buf]
            c.data = elems.ctypes.data
            c.bytes_buf = buf.ctypes.data if buf.size else None
            c.bytes_buf_len = buf.size
        else:
            data = v.data if v.data.size else np.zeros(1, dtype=v.data.dtype)
            keep.append(data)
            c.data = data.ctypes.data
            c.bytes_buf = None
            c.bytes_buf_len = 0
        if v.valid is not None:
            keep.append(v.valid)
            c.maybe_has_nulls = 1
            c.nulls = v.valid.ctypes.data
        else:
            c.maybe_has_nulls = 0
            c.nulls = None
    b = _lib.Batch()
    b.length = batch.length
    if batch.sel is not None:
        keep.append(batch.sel)
        b.sel = batch.sel.ctypes.data
    else:
        b.sel = None
    b.ncols = ncols
    b.cols = cols
    keep.append(cols)
    return b, keep


BORROW_OUTPUT = False  # default of _OutBatch(borrow=): the shells ask for CRGPU_OUT_BORROW output batches


class _OutBatch:
    """Caller-owned output batch (what allocator.ResetMaybeReallocate gives the Go shell)."""

    def __init__(self, types, capacity, bytes_cap=0, borrow=None):
        self.types = list(types)
        self.capacity = capacity
        self.borrow = BORROW_OUTPUT if borrow is None else borrow  # CRGPU_OUT_BORROW: fixed-width values are read where the library left them
        self.cols = (_lib.OutCol * max(len(types), 1))()
        self.bufs = []
        for i, t in enumerate(types):
            c = self.cols[i]
            c.type = t
            data = np.zeros(capacity * (32 if t == cd.BYTES else cd.WIDTH[t]), dtype=np.uint8)
            nulls = np.zeros((capacity + 7) // 8, dtype=np.uint8)
            spill = np.zeros(max(bytes_cap, 1), dtype=np.uint8)
            self.bufs.append([data, nulls, spill])
            c.data = data.ctypes.data
            c.nulls = nulls.ctypes.data
            c.bytes_buf = spill.ctypes.data
            c.bytes_buf_cap = bytes_cap
        self.b = _lib.OutBatch()
        self.b.capacity = capacity
        self.b.ncols = len(types)
        self.b.flags = _lib.OUT_BORROW if self.borrow else 0
        self.b.cols = self.cols
        self.grow_retries = 0

    def call(self, fn, *args):
        """Invoke fn(*args, &out); grow the Bytes spill buffers and retry when asked to."""
        for _ in range(3):
            _lib.check(fn(*args, C.byref(self.b)))
            if self.b.length > 0:
                return self.to_batch()
            need = [(i, self.cols[i].bytes_buf_len) for i in range(len(self.types))
                    if self.cols[i].bytes_buf_len > self.cols[i].bytes_buf_cap]
            if not need:
                return self.to_batch()
            self.grow_retries += 1
            for i, sz in need:
                spill = np.zeros(int(sz), dtype=np.uint8)
                self.bufs[i][2] = spill
                self.cols[i].bytes_buf = spill.ctypes.data
                self.cols[i].bytes_buf_cap = int(sz)
        raise AssertionError("output batch still too small after growing")

    def to_batch(self) -> cd.Batch:
        n = int(self.b.length)
        vecs = []
        for i, t in enumerate(self.types):
            data, nulls, spill = self.bufs[i]
            valid = nulls[: (n + 7) // 8].copy() if self.cols[i].has_nulls else None
            if valid is not None and n % 8:
                valid[-1] |= (0xFF << (n % 8)) & 0xFF
            if t == cd.BYTES:
                vecs.append(cd.elements_to_bytes(data, spill, n, valid))
            elif self.borrow:  # the library's buffer is only valid until the next call: copy out of it here
                raw = np.frombuffer(C.string_at(self.cols[i].data, n * cd.WIDTH[t]), dtype=cd.NP_DTYPE[t]) if n else np.zeros(0, cd.NP_DTYPE[t])
                assert n == 0 or self.cols[i].data != data.ctypes.data, "CRGPU_OUT_BORROW left the caller's buffer in place"
                vecs.append(cd.Vec(t, raw.copy(), valid))
            else:
                vecs.append(cd.Vec(t, data[: n * cd.WIDTH[t]].view(cd.NP_DTYPE[t]).copy(), valid))
        return cd.Batch(vecs, n)


def _i32(xs):
    return (C.c_int32 * max(len(xs), 1))(*xs)


def hash_cols(vecs, n=None, sel=None, init=1, width=32, nbuckets=0):
    """colexechash.rehash over host vectors with a selection vector (hash_utils.eg.go:39)."""
    n = (len(sel) if sel is not None else len(vecs[0])) if n is None else n
    b, keep = _marshal_batch(cd.Batch(vecs, n, sel))
    out = np.zeros(n, dtype=np.uint32 if width == 32 else np.uint64)
    _lib.check(_lib.load().crgpu_hash_cols(b.cols, len(vecs), n, b.sel, init, width, nbuckets,
                                           out.ctypes.data if n else None))
    return out


# ---------------------------------------------------------------- hash joiner
def join_output_types(join_type, left_types, right_types):
    """JoinType.MakeOutputTypes (sql/sqlbase/join_type.go:101)."""
    if join_type in (4, 5, 6, 7):  # LEFT_SEMI, LEFT_ANTI, INTERSECT_ALL, EXCEPT_ALL
        return list(left_types)
    if join_type in (8, 9):  # RIGHT_SEMI, RIGHT_ANTI
        return list(right_types)
    return list(left_types) + list(right_types)


class HashJoiner(Operator):
    """colexecjoin.NewHashJoiner(spec, leftSource, rightSource)."""

    def __init__(self, join_type, left_types, right_types, left_eq, right_eq, right_eq_cols_are_key,
                 left: Operator, right: Operator, out_cols=None, capacity=cd.BATCH_SIZE, lookahead=1):
        self.lib = _lib.load()
        self.left, self.right = left, right
        all_out = join_output_types(join_type, left_types, right_types)
        self.out_types = all_out if out_cols is None else [all_out[i] for i in out_cols]
        self._keep = (_i32(left_types), _i32(right_types), _i32(left_eq), _i32(right_eq),
                      _i32(out_cols if out_cols is not None else []))
        spec = _lib.JoinSpec(join_type, int(bool(right_eq_cols_are_key)), len(left_types), len(right_types),
                             self._keep[0], self._keep[1], len(left_eq), -1 if out_cols is None else len(out_cols),
                             self._keep[2], self._keep[3], self._keep[4])
        self.h = C.c_void_p()
        _lib.check(self.lib.crgpu_hj_new(C.byref(spec), C.byref(self.h)))
        self.right_types = list(right_types)
        self.capacity = capacity
        self.lookahead = max(1, lookahead)
        self.out = _OutBatch(self.out_types, capacity)
        self.state = "building"

    def init(self):
        self.left.init()
        self.right.init()

    def _build(self):
        while True:  # hjBuilding: consume the whole right input (hashjoiner.go:339-365)
            b = self.right.next()
            if b.length == 0:
                break
            mb, keep = _marshal_batch(b)
            _lib.check(self.lib.crgpu_hj_build_append(self.h, C.byref(mb)))
        _lib.check(self.lib.crgpu_hj_build_finish(self.h))
        self.state = "probing"

    def next(self) -> cd.Batch:
        if self.state == "building":
            self._build()
        while True:
            if self.state == "probing":
                for _ in range(self.lookahead):
                    b = self.left.next()
                    if b.length == 0:
                        _lib.check(self.lib.crgpu_hj_probe_finish(self.h))
                        self.state = "draining"
                        break
                    mb, keep = _marshal_batch(b)
                    _lib.check(self.lib.crgpu_hj_probe_append(self.h, C.byref(mb)))
            out = self.out.call(self.lib.crgpu_hj_next_output, self.h)
            if out.length > 0 or self.state == "draining":
                return out

    def export_buffered(self):
        """colexecop.BufferingInMemoryOperator.ExportBuffered: one batch per call, zero-length at the end."""
        ob = _OutBatch(self.right_types, self.capacity)
        return ob.call(self.lib.crgpu_hj_export_buffered, self.h)

    def reset(self):
        _lib.check(self.lib.crgpu_hj_reset(self.h))
        for src in (self.left, self.right):
            if hasattr(src, "reset"):
                src.reset()
        self.state = "building"

    def close(self):
        if self.h:
            self.lib.crgpu_hj_free(self.h)
            self.h = None


# ---------------------------------------------------------------- hash aggregator
AGG_OUTPUT_TYPE = {  # colexecagg.ProcessAggregations output types for the supported overloads
    1: cd.FLOAT64, 2: cd.BOOL, 3: cd.BOOL, 5: cd.INT64, 10: cd.FLOAT64, 11: cd.INT64, 14: cd.INT64,
}


class HashAggregator(Operator):
    """colexec.NewHashAggregator(args): aggs = [(func, col, filter_col[, distinct])]; ordered_cols =
    AggregatorSpec.OrderedGroupCols (a subset of group_cols the input is sorted on): the groups of a run of equal
    ordered values are emitted as soon as the run is over (hash_aggregator_tmpl.go:160-169, 256-306)."""

    def __init__(self, types, group_cols, aggs, input: Operator, capacity=cd.BATCH_SIZE, track_input=False, ordered_cols=()):
        self.lib = _lib.load()
        ordered_cols = list(ordered_cols)
        if any(c not in group_cols for c in ordered_cols):
            raise ValueError("ordered columns must be grouping columns")
        group_cols = ordered_cols + [c for c in group_cols if c not in ordered_cols]  # the C-ABI wants them first
        self.ordered = len(ordered_cols) > 0
        self.input = input
        self.types = list(types)
        self.capacity = capacity
        descs = (_lib.AggDesc * max(len(aggs), 1))()
        self.out_types = []
        for i, a in enumerate(aggs):  # (func, col, filter_col[, distinct])
            f, c, fc = a[0], a[1], a[2]
            descs[i] = _lib.AggDesc(f, c, fc, int(bool(a[3])) if len(a) > 3 else 0)
            self.out_types.append(AGG_OUTPUT_TYPE.get(f, types[c] if c >= 0 else cd.INT64))
        self._keep = (_i32(types), _i32(group_cols), descs)
        spec = _lib.AggSpec(len(types), len(group_cols), self._keep[0], self._keep[1], len(aggs), len(ordered_cols), descs)
        self.h = C.c_void_p()
        _lib.check(self.lib.crgpu_agg_new(C.byref(spec), C.byref(self.h)))
        if track_input:  # planned under a disk spiller: hashAggregator.inputTrackingState (hash_aggregator.go:138-153)
            _lib.check(self.lib.crgpu_agg_track_input(self.h, 1))
        self.out = _OutBatch(self.out_types, capacity)
        self._export = None
        self.state = "aggregating"

    def init(self):
        self.input.init()

    def next(self):
        if self.state == "aggregating":
            while True:
                if self.ordered:  # groups of completed runs go out while the input is still being read
                    out = self.out.call(self.lib.crgpu_agg_next_output, self.h)
                    if out.length > 0:
                        return out
                b = self.input.next()
                if b.length == 0:
                    break
                mb, keep = _marshal_batch(b)
                _lib.check(self.lib.crgpu_agg_append(self.h, C.byref(mb)))
            _lib.check(self.lib.crgpu_agg_finish(self.h))
            self.state = "outputting"
        return self.out.call(self.lib.crgpu_agg_next_output, self.h)

    def export_buffered(self):
        """hashAggregator.ExportBuffered (hash_aggregator.go:455-471): the tracked input, one batch per call."""
        if self._export is None:
            self._export = _OutBatch(self.types, self.capacity)
        return self._export.call(self.lib.crgpu_agg_export_buffered, self.h)

    def release_before_export(self):
        _lib.check(self.lib.crgpu_agg_release_before_export(self.h))

    def release_after_export(self):
        _lib.check(self.lib.crgpu_agg_release_after_export(self.h))

    def reset(self):
        _lib.check(self.lib.crgpu_agg_reset(self.h))
        if hasattr(self.input, "reset"):
            self.input.reset()
        self.state = "aggregating"

    def close(self):
        if self.h:
            self.lib.crgpu_agg_free(self.h)
            self.h = None


# ---------------------------------------------------------------- unordered distinct
class UnorderedDistinct(Operator):
    """colexec.NewUnorderedDistinct(allocator, input, distinctCols, typs, nullsAreDistinct, errorOnDup)
    (unordered_distinct.go:22-40): Next() returns the new distinct tuples of the input read so far, in input order."""

    def __init__(self, types, distinct_cols, input: Operator, nulls_are_distinct=False, error_on_dup="",
                 capacity=cd.BATCH_SIZE, lookahead=1):
        self.lib = _lib.load()
        self.input = input
        self.error_on_dup = error_on_dup
        self._keep = (_i32(types), _i32(distinct_cols))
        self.h = C.c_void_p()
        _lib.check(self.lib.crgpu_distinct_new(self._keep[0], len(types), self._keep[1], len(distinct_cols),
                                               int(bool(nulls_are_distinct)), int(bool(error_on_dup)), C.byref(self.h)))
        self.out = _OutBatch(list(types), capacity)
        self._export = _OutBatch(list(types), capacity)
        self.lookahead = max(1, lookahead)
        self.state = "reading"

    def init(self):
        self.input.init()

    def _check(self, st):
        if st == _lib.ERR_DUPLICATE:  # UpsertDistinctHelper.MaybeEmitErrorOnDup
            raise DuplicateError(self.error_on_dup)
        _lib.check(st)

    def next(self):
        while True:
            if self.state == "reading":
                for _ in range(self.lookahead):
                    b = self.input.next()
                    if b.length == 0:
                        self._check(self.lib.crgpu_distinct_finish(self.h))
                        self.state = "draining"
                        break
                    mb, keep = _marshal_batch(b)
                    self._check(self.lib.crgpu_distinct_append(self.h, C.byref(mb)))
            out = self.out.call(self._next_output, self.h)
            if out.length > 0 or self.state == "draining":
                return out

    def export_buffered(self):
        """UnorderedDistinct.ExportBuffered (unordered_distinct.go:123-134): new tuples not emitted yet."""
        return self._export.call(self.lib.crgpu_distinct_export_buffered, self.h)

    def _next_output(self, h, out):
        st = self.lib.crgpu_distinct_next_output(h, out)
        if st == _lib.ERR_DUPLICATE:
            raise DuplicateError(self.error_on_dup)
        return st

    def reset(self):
        _lib.check(self.lib.crgpu_distinct_reset(self.h))
        if hasattr(self.input, "reset"):
            self.input.reset()
        self.state = "reading"

    def close(self):
        if self.h:
            self.lib.crgpu_distinct_free(self.h)
            self.h = None


class DuplicateError(Exception):
    """The plan's errorOnDup message (colexecbase.UpsertDistinctHelper)."""


# ---------------------------------------------------------------- hash router
class _RouterOutput(Operator):
    def __init__(self, router, idx):
        self.router, self.idx = router, idx
        self.out = _OutBatch(router.types, router.capacity)

    def next(self):
        return self.out.call(self.router.lib.crgpu_route_next_output, self.router.h, self.idx)


class HashRouter:
    """colflow.NewHashRouter(unlimitedAllocators, input, types, hashCols, ...): Run() drains the input,
    each output is an Operator."""

    def __init__(self, types, hash_cols, num_outputs, input: Operator, capacity=cd.BATCH_SIZE):
        self.lib = _lib.load()
        self.types = list(types)
        self.capacity = capacity
        self.input = input
        self.h = C.c_void_p()
        _lib.check(self.lib.crgpu_route_new(_i32(types), len(types), _i32(hash_cols), len(hash_cols), num_outputs,
                                            C.byref(self.h)))
        self.outputs = [_RouterOutput(self, i) for i in range(num_outputs)]

    def run(self):
        self.input.init()
        while True:  # processNextBatch until the zero-length batch (routers.go:683-715)
            b = self.input.next()
            if b.length == 0:
                break
            mb, keep = _marshal_batch(b)
            _lib.check(self.lib.crgpu_route_partition(self.h, C.byref(mb)))
        _lib.check(self.lib.crgpu_route_finish(self.h))

    def close(self):
        if self.h:
            self.lib.crgpu_route_free(self.h)
            self.h = None
