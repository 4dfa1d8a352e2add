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
    _fields_ = [("func", C.c_int32), ("col", C.c_int32), ("filter_col", C.c_int32), ("_pad", C.c_int32)]


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


def _result_to_vecs(res):
    L = lib()
    if not res:
        return None
    ncols, nrows = L.ora_result_ncols(res), L.ora_result_nrows(res)
    out = []
    for i in range(ncols):
        c = OraCol()
        L.ora_result_col(res, i, C.byref(c))
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
    L.ora_result_free(res)
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
    """aggs: list of (func, col or -1, filter_col or -1)."""
    cc, _k = make_cols(cols)
    gc = np.ascontiguousarray(group_cols, dtype=np.int32)
    specs = (OraAggSpec * max(len(aggs), 1))()
    for i, a in enumerate(aggs):
        specs[i].func, specs[i].col = a[0], a[1]
        specs[i].filter_col = a[2] if len(a) > 2 else -1
    err = C.c_int32(0)
    n = len(cols[0]) if cols else 0
    res = lib().ora_hash_agg(cc, len(cols), n, _ptr(gc), len(gc), specs, len(aggs), batch_size, max_buffered,
                             initial_buckets, C.byref(err))
    if err.value:
        raise OracleError(err.value)
    return _result_to_vecs(res)


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
