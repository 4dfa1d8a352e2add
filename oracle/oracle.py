"""ctypes wrapper around the CPU oracle (oracle/crdb_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (cockroach_amd/) never
imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from cockroach_amd import coldata as cd

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libcrdb_oracle.so")


class OraCol(C.Structure):
    _fields_ = [("type", C.c_int32), ("_pad", C.c_int32), ("data", C.c_void_p),
                ("valid", C.c_void_p), ("offsets", C.c_void_p)]


class OraAggSpec(C.Structure):
    _fields_ = [("func", C.c_int32), ("col", C.c_int32), ("filter_col", C.c_int32), ("distinct", C.c_int32)]


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if not force and os.path.exists(_LIB_PATH) and all(
            os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.ora_memhash.restype = C.c_uint64
        L.ora_memhash.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.ora_memhash64.restype = C.c_uint64
        L.ora_memhash64.argtypes = [C.c_uint64, C.c_uint64]
        L.ora_hash_cols32.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.ora_hash_cols64.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.ora_distribute.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_void_p]
        L.ora_hash_join.restype = C.c_void_p
        L.ora_hash_join.argtypes = [C.c_int32, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_uint32]
        L.ora_unordered_distinct.restype = C.c_void_p
        L.ora_unordered_distinct.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int32, C.c_int32,
                                             C.c_uint32, C.c_void_p]
        L.ora_hash_agg.restype = C.c_void_p
        L.ora_hash_agg.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_int32, C.c_int32, C.c_uint32, C.c_void_p]
        L.ora_result_ncols.restype = C.c_int32
        L.ora_result_ncols.argtypes = [C.c_void_p]
        L.ora_result_nrows.restype = C.c_int64
        L.ora_result_nrows.argtypes = [C.c_void_p]
        L.ora_result_col.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ora_result_free.argtypes = [C.c_void_p]
        L.ora_test_set_hash_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ora_test_last_join_state.argtypes = [C.c_void_p] * 8
        L.ora_test_last_agg_state.argtypes = [C.c_void_p] * 4
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_cols(vecs):
    """list[Vec] -> (ctypes array of OraCol, keepalive)."""
    arr = (OraCol * max(len(vecs), 1))()
    for i, v in enumerate(vecs):
        arr[i].type = v.type
        arr[i].data = v.data.ctypes.data if v.data.size else None
        arr[i].valid = None if v.valid is None else v.valid.ctypes.data
        arr[i].offsets = v.offsets.ctypes.data if v.offsets is not None else None
    return arr, vecs


def _result_to_vecs(res, prefix="ora_result"):
    L = lib()
    if not res:
        return None
    f_ncols, f_nrows = getattr(L, prefix + "_ncols"), getattr(L, prefix + "_nrows")
    f_col, f_free = getattr(L, prefix + "_col"), getattr(L, prefix + "_free")
    ncols, nrows = f_ncols(res), f_nrows(res)
    out = []
    for i in range(ncols):
        c = OraCol()
        f_col(res, i, C.byref(c))
        valid = None
        if c.valid:
            valid = np.ctypeslib.as_array(C.cast(c.valid, C.POINTER(C.c_uint8)), shape=((nrows + 7) // 8,)).copy()
        if c.type == cd.BYTES:
            offs = np.ctypeslib.as_array(C.cast(c.offsets, C.POINTER(C.c_uint32)), shape=(nrows + 1,)).copy()
            nb = int(offs[-1]) if nrows else 0
            data = (np.ctypeslib.as_array(C.cast(c.data, C.POINTER(C.c_uint8)), shape=(nb,)).copy()
                    if nb else np.zeros(0, np.uint8))
            out.append(cd.Vec(cd.BYTES, data, valid, offs))
        else:
            dt = np.dtype(cd.NP_DTYPE[c.type])
            if nrows:
                raw = np.ctypeslib.as_array(C.cast(c.data, C.POINTER(C.c_uint8)), shape=(nrows * dt.itemsize,)).copy()
                data = raw.view(dt)
            else:
                data = np.zeros(0, dt)
            out.append(cd.Vec(c.type, data, valid))
    f_free(res)
    return out


# ---------------------------------------------------------------------------
def memhash(b: bytes, seed: int) -> int:
    buf = C.create_string_buffer(b, len(b)) if len(b) else C.create_string_buffer(1)
    return lib().ora_memhash(C.cast(buf, C.c_void_p), seed & (2**64 - 1), len(b))


def memhash64(v: int, seed: int) -> int:
    return lib().ora_memhash64(v & (2**64 - 1), seed & (2**64 - 1))


def hash_cols(vecs, n=None, sel=None, init=1, nbuckets=0, width=64):
    """initHash -> rehash per column -> finalizeHash (nbuckets=0: raw hash)."""
    n = (len(sel) if sel is not None else len(vecs[0])) if n is None else n
    cols, _k = make_cols(vecs)
    selp = None if sel is None else np.ascontiguousarray(sel, dtype=np.int64)
    if width == 32:
        out = np.zeros(n, dtype=np.uint32)
        lib().ora_hash_cols32(cols, len(vecs), n, _ptr(selp), init, nbuckets, _ptr(out))
    else:
        out = np.zeros(n, dtype=np.uint64)
        lib().ora_hash_cols64(cols, len(vecs), n, _ptr(selp), init, nbuckets, _ptr(out))
    return out


def distribute(vecs, hash_cols_idx, num_outputs, init=1, batch_size=cd.BATCH_SIZE):
    n = len(vecs[0])
    cols, _k = make_cols(vecs)
    hc = np.ascontiguousarray(hash_cols_idx, dtype=np.int32)
    dest = np.zeros(n, dtype=np.int32)
    counts = np.zeros(num_outputs, dtype=np.int64)
    lib().ora_distribute(cols, len(vecs), _ptr(hc), len(hc), n, init, num_outputs, batch_size, _ptr(dest), _ptr(counts))
    return dest, counts


def hash_join(join_type, left, right, left_eq, right_eq, right_eq_cols_are_key=False,
              batch_size=cd.BATCH_SIZE, initial_buckets=256):
    lc, _k1 = make_cols(left)
    rc, _k2 = make_cols(right)
    le = np.ascontiguousarray(left_eq, dtype=np.int32)
    re_ = np.ascontiguousarray(right_eq, dtype=np.int32)
    nl = len(left[0]) if left else 0
    nr = len(right[0]) if right else 0
    res = lib().ora_hash_join(join_type, lc, len(left), nl, rc, len(right), nr, _ptr(le), _ptr(re_), len(le),
                              int(bool(right_eq_cols_are_key)), batch_size, initial_buckets)
    return _result_to_vecs(res)


class OracleError(Exception):
    def __init__(self, code):
        super().__init__({1: "integer out of range", 2: "unsupported aggregate/type"}.get(code, str(code)))
        self.code = code


def hash_agg(cols, group_cols, aggs, batch_size=cd.BATCH_SIZE, max_buffered=cd.MAX_BATCH_SIZE, initial_buckets=256):
    """aggs: list of (func, col or -1, filter_col or -1[, distinct])."""
    cc, _k = make_cols(cols)
    gc = np.ascontiguousarray(group_cols, dtype=np.int32)
    specs = (OraAggSpec * max(len(aggs), 1))()
    for i, a in enumerate(aggs):
        specs[i].func, specs[i].col = a[0], a[1]
        specs[i].filter_col = a[2] if len(a) > 2 else -1
        specs[i].distinct = int(bool(a[3])) if len(a) > 3 else 0
    err = C.c_int32(0)
    n = len(cols[0]) if cols else 0
    res = lib().ora_hash_agg(cc, len(cols), n, _ptr(gc), len(gc), specs, len(aggs), batch_size, max_buffered,
                             initial_buckets, C.byref(err))
    if err.value:
        raise OracleError(err.value)
    return _result_to_vecs(res)


def unordered_distinct(cols, distinct_cols, nulls_are_distinct=False, batch_size=cd.BATCH_SIZE, initial_buckets=128):
    """colexec.UnorderedDistinct (unordered_distinct.go:94-121): returns (output vecs in emission order, dropped) where
    dropped says whether any batch lost a tuple (the errorOnDup condition)."""
    cc, _k = make_cols(cols)
    dc = np.ascontiguousarray(distinct_cols, dtype=np.int32)
    dropped = C.c_int32(0)
    n = len(cols[0]) if cols else 0
    res = lib().ora_unordered_distinct(cc, len(cols), n, _ptr(dc), len(dc), int(bool(nulls_are_distinct)), batch_size,
                                       initial_buckets, C.byref(dropped))
    return _result_to_vecs(res), bool(dropped.value)


def select_i64_const(cmp, vec, cst):
    """colexecsel.sel<CMP>Int64Int64ConstOp (selection_ops.eg.go:34952 for LE): the indices that stay selected, in
    order.  A NULL never passes (:34971)."""
    v = vec.data.astype(np.int64)
    r = {"EQ": v == cst, "NE": v != cst, "LT": v < cst, "LE": v <= cst, "GT": v > cst, "GE": v >= cst}[cmp]
    return np.flatnonzero(r & vec.present())


def project_f64(op, a, b):
    """colexecproj.proj{Plus,Minus,Mult}Float64Float64Op and the constant forms (proj_non_const_ops.eg.go:4922,8126,
    12196; proj_const_left_ops.eg.go:4515,7503,11295): IEEE double arithmetic row by row, NULL if an operand is NULL."""
    vecs = [x for x in (a, b) if isinstance(x, cd.Vec)]
    n = len(vecs[0])
    x = a.data if isinstance(a, cd.Vec) else np.full(n, float(a))
    y = b.data if isinstance(b, cd.Vec) else np.full(n, float(b))
    with np.errstate(all="ignore"):
        r = {"+": lambda: x + y, "-": lambda: x - y, "*": lambda: x * y,
             "*(1-)": lambda: x * (1.0 - y), "*(1+)": lambda: x * (1.0 + y)}[op]()  # the last two: two operators in a row
    pres = np.ones(n, dtype=bool)
    for v in vecs:
        pres &= v.present()
    return cd.Vec.from_numpy(cd.FLOAT64, r, None if pres.all() else pres)


def select_bytes_const(cmp, vec, cst: bytes):
    """colexecsel.selEQBytesBytesConstOp / selNEBytesBytesConstOp (selection_ops.eg.go:358,10050: bytes.Compare with
    the constant) and selContainsBytesBytesConstOp (sel_like_ops.eg.go:229: bytes.Contains(arg, constArg) != negate):
    the indices that stay selected, in order.  A NULL never passes."""
    vals = vec.to_values()
    test = {"EQ": lambda v: v == cst, "NE": lambda v: v != cst, "CONTAINS": lambda v: cst in v,
            "NOT CONTAINS": lambda v: cst not in v}[cmp]
    return np.array([i for i, v in enumerate(vals) if v is not None and test(v)], dtype=np.int64)


def extract_year(vec):
    """EXTRACT(year FROM date) (sem/builtins/builtins.go:11946: tree.DFloat(fromTime.Year())) over a DATE column held
    as int64 days since the epoch (pgdate), evaluated row by row by colexecbuiltins.defaultBuiltinFuncOperator
    (builtin_funcs.go:44).  Proleptic Gregorian, like Go's time.Time and numpy's datetime64."""
    years = vec.data.astype("datetime64[D]").astype("datetime64[Y]").astype(np.int64) + 1970
    return cd.Vec.from_numpy(cd.FLOAT64, years.astype(np.float64), vec.valid)


def deselect(vecs, idx):
    """colexecutils.deselectorOp (deselector.go:48-80): the selected tuples as dense vectors."""
    return [v.take(np.asarray(idx, dtype=np.int64)) for v in vecs]


def set_test_hash(mapping):
    """Inject h(value) for the reference's worked examples; {} resets."""
    vals = np.array(list(mapping.keys()), dtype=np.int64)
    hs = np.array(list(mapping.values()), dtype=np.uint32)
    lib().ora_test_set_hash_table(_ptr(vals) if len(vals) else None, _ptr(hs) if len(hs) else None, len(vals))


def last_join_state(nkeys_hint=4096):
    first = np.zeros(1 << 16, dtype=np.uint32)
    nfirst = C.c_int32(0)
    nxt = np.zeros(nkeys_hint + 1, dtype=np.uint32)
    same = np.zeros(nkeys_hint + 1, dtype=np.uint32)
    vis = np.zeros(nkeys_hint + 1, dtype=np.uint8)
    nkeys = C.c_int64(0)
    head = np.zeros(8192, dtype=np.uint32)
    nhead = C.c_int32(0)
    lib().ora_test_last_join_state(_ptr(first), C.addressof(nfirst), _ptr(nxt), _ptr(same), _ptr(vis),
                                   C.addressof(nkeys), _ptr(head), C.addressof(nhead))
    k = nkeys.value
    return dict(first=first[: nfirst.value].tolist(), next=nxt[: k + 1].tolist(), same=same[: k + 1].tolist(),
                visited=vis[: k + 1].astype(bool).tolist(), head_id=head[: nhead.value].tolist())


def last_agg_state(nkeys_hint=4096):
    first = np.zeros(1 << 16, dtype=np.uint32)
    nfirst = C.c_int32(0)
    nxt = np.zeros(nkeys_hint + 1, dtype=np.uint32)
    nkeys = C.c_int64(0)
    lib().ora_test_last_agg_state(_ptr(first), C.addressof(nfirst), _ptr(nxt), C.addressof(nkeys))
    return dict(first=first[: nfirst.value].tolist(), next=nxt[: nkeys.value + 1].tolist())


# ---------------------------------------------------------------------------
# KV codec (oracle/crdb_kv.c): encoder + cFetcher-style decoder
# ---------------------------------------------------------------------------
class OraKvTable(C.Structure):
    _fields_ = [("table_id", C.c_int32), ("index_id", C.c_int32), ("n_key_cols", C.c_int32), ("n_val_cols", C.c_int32),
                ("key_types", C.c_void_p), ("key_dirs", C.c_void_p), ("val_col_ids", C.c_void_p), ("val_types", C.c_void_p)]


class OraKvFetch(C.Structure):
    _fields_ = [("key_prefix_len", C.c_int32), ("n_key_cols", C.c_int32), ("key_types", C.c_void_p),
                ("key_dirs", C.c_void_p), ("key_out", C.c_void_p), ("n_val_cols", C.c_int32), ("n_out", C.c_int32),
                ("val_col_ids", C.c_void_p), ("val_types", C.c_void_p), ("val_out", C.c_void_p)]


class OraKvFamilies(C.Structure):
    _fields_ = [("n_families", C.c_int32), ("mvcc_header_every", C.c_int32), ("family_ids", C.c_void_p),
                ("default_col_ids", C.c_void_p), ("val_family", C.c_void_p)]


def _i32arr(xs):
    return np.ascontiguousarray(np.asarray(list(xs), dtype=np.int32))


class KvFamilies:
    """Column families of a table (descpb.ColumnFamilyDescriptor): family ids in increasing order starting with 0, the
    DefaultColumnID of each (0: none) and, for the encoder, the family (an index into family_ids) of every value column.
    mvcc_header_every k > 0: the encoder gives every k-th KV's value an MVCC extended-encoding header."""

    def __init__(self, family_ids, default_col_ids, val_family=None, mvcc_header_every=0):
        self.family_ids, self.default_col_ids = list(family_ids), list(default_col_ids)
        self.val_family = list(val_family) if val_family is not None else []
        self.mvcc_header_every = mvcc_header_every

    def struct(self):
        keep = [_i32arr(self.family_ids), _i32arr(self.default_col_ids), _i32arr(self.val_family or [0])]
        return OraKvFamilies(len(self.family_ids), self.mvcc_header_every, keep[0].ctypes.data, keep[1].ctypes.data,
                             keep[2].ctypes.data), keep


def _kv_setup():
    L = lib()
    if getattr(L, "_kv_ready", False):
        return L
    for name in ("ora_enc_uvarint_asc", "ora_enc_varint_asc", "ora_enc_varint_desc", "ora_enc_nonsorting_uvarint",
                 "ora_enc_stdlib_varint"):
        getattr(L, name).restype = C.c_int
    L.ora_enc_uvarint_asc.argtypes = [C.c_void_p, C.c_uint64]
    L.ora_enc_varint_asc.argtypes = [C.c_void_p, C.c_int64]
    L.ora_enc_varint_desc.argtypes = [C.c_void_p, C.c_int64]
    L.ora_enc_nonsorting_uvarint.argtypes = [C.c_void_p, C.c_uint64]
    L.ora_enc_stdlib_varint.argtypes = [C.c_void_p, C.c_int64]
    L.ora_enc_float_asc.restype = C.c_int
    L.ora_enc_float_asc.argtypes = [C.c_void_p, C.c_double]
    L.ora_enc_float_desc.restype = C.c_int
    L.ora_enc_float_desc.argtypes = [C.c_void_p, C.c_double]
    L.ora_enc_bytes_asc.restype = C.c_int64
    L.ora_enc_bytes_asc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.ora_enc_bytes_desc.restype = C.c_int64
    L.ora_enc_bytes_desc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    for name, outt in (("ora_dec_uvarint_asc", C.c_uint64), ("ora_dec_varint_asc", C.c_int64),
                       ("ora_dec_varint_desc", C.c_int64), ("ora_dec_float_asc", C.c_double),
                       ("ora_dec_float_desc", C.c_double), ("ora_dec_nonsorting_uvarint", C.c_uint64),
                       ("ora_dec_stdlib_varint", C.c_int64)):
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int64, C.POINTER(outt)]
    L.ora_dec_bytes.restype = C.c_int64
    L.ora_dec_bytes.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]
    L.ora_kv_encode.restype = C.c_int
    L.ora_kv_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.ora_kv_encode2.restype = C.c_int
    L.ora_kv_encode2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.ora_kv_decode2.restype = C.c_int
    L.ora_kv_decode2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    L.ora_kv_free.argtypes = [C.c_void_p]
    L.ora_kv_decode.restype = C.c_int
    L.ora_kv_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    L.ora_kv_result_ncols.restype = C.c_int32
    L.ora_kv_result_ncols.argtypes = [C.c_void_p]
    L.ora_kv_result_nrows.restype = C.c_int64
    L.ora_kv_result_nrows.argtypes = [C.c_void_p]
    L.ora_kv_result_col.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.ora_kv_result_free.argtypes = [C.c_void_p]
    L._kv_ready = True
    return L


def kv_enc(kind: str, value) -> bytes:
    """One primitive encoding: kind in uvarint_asc, varint_asc, varint_desc, float_asc, float_desc,
    bytes_asc, bytes_desc, nonsorting_uvarint, stdlib_varint."""
    L = _kv_setup()
    if kind.startswith("bytes"):
        data = bytes(value)
        buf = C.create_string_buffer(2 * len(data) + 8)
        src = C.create_string_buffer(data, max(len(data), 1))
        n = getattr(L, "ora_enc_" + kind)(buf, src, len(data))
    else:
        buf = C.create_string_buffer(16)
        n = getattr(L, "ora_enc_" + kind)(buf, value)
    return buf.raw[:n]


def kv_dec(kind: str, data: bytes):
    """Inverse of kv_enc: returns (value, bytes consumed)."""
    L = _kv_setup()
    src = C.create_string_buffer(bytes(data), max(len(data), 1))
    if kind.startswith("bytes"):
        out = C.create_string_buffer(len(data) + 1)
        outlen = C.c_int64()
        n = L.ora_dec_bytes(src, len(data), 1 if kind.endswith("desc") else 0, out, C.byref(outlen))
        return (out.raw[:outlen.value] if n >= 0 else None), n
    ctype = {"uvarint_asc": C.c_uint64, "nonsorting_uvarint": C.c_uint64, "float_asc": C.c_double,
             "float_desc": C.c_double}.get(kind, C.c_int64)
    v = ctype()
    n = getattr(L, "ora_dec_" + kind)(src, len(data), C.byref(v))
    return (v.value if n >= 0 else None), n


class KvTable:
    """The encoder's table: key columns (type, direction) and value columns (id, type), one family."""

    def __init__(self, table_id, index_id, key_types, key_dirs, val_col_ids, val_types):
        self.table_id, self.index_id = table_id, index_id
        self.key_types, self.key_dirs = list(key_types), list(key_dirs)
        self.val_col_ids, self.val_types = list(val_col_ids), list(val_types)

    def prefix_len(self):
        return len(kv_enc("uvarint_asc", self.table_id)) + len(kv_enc("uvarint_asc", self.index_id))


def kv_encode(table: KvTable, key_vecs, val_vecs, max_batch_bytes=10 << 20, explicit_null_every=0, families=None):
    """Rows -> (uint8 array of concatenated MVCCScan batch responses, int64 batch offsets[nbatches+1]).
    families: a KvFamilies (one KV per family and row, rowenc.EncodePrimaryIndex); None: the single family 0."""
    L = _kv_setup()
    n = len(key_vecs[0]) if key_vecs else len(val_vecs[0])
    keep = [_i32arr(table.key_types), _i32arr(table.key_dirs), _i32arr(table.val_col_ids), _i32arr(table.val_types)]
    t = OraKvTable(table.table_id, table.index_id, len(table.key_types), len(table.val_types),
                   keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, keep[3].ctypes.data)
    kc, _k1 = make_cols(key_vecs)
    vc, _k2 = make_cols(val_vecs)
    out, out_len, offs, nb = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    fs, _fkeep = families.struct() if families is not None else (None, None)
    rc = L.ora_kv_encode2(C.byref(t), C.byref(fs) if fs is not None else None, kc, vc, n, max_batch_bytes, explicit_null_every,
                          C.byref(out), C.byref(out_len), C.byref(offs), C.byref(nb))
    assert rc == 0
    data = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(out_len.value, 1),))[:out_len.value].copy()
    offsets = np.ctypeslib.as_array(C.cast(offs, C.POINTER(C.c_int64)), shape=(nb.value + 1,)).copy()
    L.ora_kv_free(out)
    L.ora_kv_free(offs)
    return data, offsets


class KvDecodeError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle kv decode error {code}")
        self.code = code


def kv_decode(key_prefix_len, key_types, key_dirs, key_out, val_col_ids, val_types, val_out, out_types, data, offsets,
              families=None):
    """cFetcher-style decode of batch responses -> list[Vec] (one per output ordinal).  families: a KvFamilies for
    tables with several column families (rows of several KVs); None: the single family 0."""
    L = _kv_setup()
    keep = [_i32arr(key_types), _i32arr(key_dirs), _i32arr(key_out), _i32arr(val_col_ids), _i32arr(val_types),
            _i32arr(val_out), _i32arr(out_types)]
    f = OraKvFetch(key_prefix_len, len(key_types), keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data,
                   len(val_col_ids), len(out_types), keep[3].ctypes.data, keep[4].ctypes.data, keep[5].ctypes.data)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    res = C.c_void_p()
    fs, _fkeep = families.struct() if families is not None else (None, None)
    rc = L.ora_kv_decode2(C.byref(f), C.byref(fs) if fs is not None else None, keep[6].ctypes.data,
                          data.ctypes.data if data.size else None, offsets.ctypes.data, len(offsets) - 1, C.byref(res))
    if rc != 0:
        L.ora_kv_result_free(res)
        raise KvDecodeError(rc)
    return _result_to_vecs(res, prefix="ora_kv_result")
