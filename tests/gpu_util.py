"""Helpers for the -m gpu parity tests: every call goes through the C-ABI
(ctypes on libcrgpu.so); the oracle is only the checker."""
from __future__ import annotations

import numpy as np

from cockroach_amd import coldata as cd
from cockroach_amd import device as dev
from tests import helpers as H


def gpu_join(join_type, left, right, left_eq, right_eq, right_key=False, probe_chunks=1):
    """left/right: list[Vec].  Returns output rows (tuples) = probe outputs + emit_right."""
    jt = H.JOIN_TYPE[join_type] if isinstance(join_type, str) else join_type
    j = dev.Join(jt, [v.type for v in left], [v.type for v in right], left_eq, right_eq, right_key)
    rcols = [dev.upload(v) for v in right]
    j.build(rcols, len(right[0]) if right else 0)
    nleft = len(left[0]) if left else 0
    rows = []
    bounds = np.linspace(0, nleft, probe_chunks + 1).astype(np.int64)
    for a, b in zip(bounds[:-1], bounds[1:]):
        chunk = [v.slice(int(a), int(b)) for v in left] if probe_chunks > 1 else left
        lcols = [dev.upload(v) for v in chunk]
        res = j.probe(lcols, int(b - a))
        rows += H.vecs_to_tuples(dev.result_to_vecs(res), int(res.n))
    res = j.emit_right()
    rows += H.vecs_to_tuples(dev.result_to_vecs(res), int(res.n))
    j.close()
    return rows


def gpu_join_vecs(join_type, left, right, left_eq, right_eq, right_key=False, probe_chunks=1):
    """Same, but returns host Vecs of the probe result (probe_chunks > 1: one list of Vecs per chunk) and of emit_right
    (for large inputs)."""
    jt = H.JOIN_TYPE[join_type] if isinstance(join_type, str) else join_type
    j = dev.Join(jt, [v.type for v in left], [v.type for v in right], left_eq, right_eq, right_key)
    rcols = [dev.upload(v) for v in right]
    j.build(rcols, len(right[0]))
    nleft = len(left[0])
    bounds = np.linspace(0, nleft, probe_chunks + 1).astype(np.int64)
    parts = []
    for s, e in zip(bounds[:-1], bounds[1:]):
        chunk = [v.slice(int(s), int(e)) for v in left] if probe_chunks > 1 else left
        parts.append(dev.result_to_vecs(j.probe([dev.upload(v) for v in chunk], int(e - s))))
    b = dev.result_to_vecs(j.emit_right())
    j.close()
    return (parts[0] if probe_chunks == 1 else parts), b


def rows_as_sorted_array(vecs, n=None):
    """Order-insensitive canonical form of a result made of fixed-width columns:
    a lexicographically sorted int64 matrix (NULL -> extra flag column)."""
    if not vecs:
        return np.zeros((n or 0, 0), dtype=np.int64)
    cols = []
    for v in vecs:
        pres = v.present()
        if v.type == cd.FLOAT64:
            raw = np.where(pres, v.data, 0.0).astype(np.float64).view(np.int64)
        else:
            raw = np.where(pres, v.data.astype(np.int64), 0)
        cols.append(pres.astype(np.int64))
        cols.append(raw)
    m = np.stack(cols, axis=1)
    order = np.lexsort(m.T[::-1])
    return m[order]


def gpu_agg(cols, group_cols, aggs, chunks=1):
    a = dev.Agg([v.type for v in cols], group_cols, aggs)
    n = len(cols[0]) if cols else 0
    bounds = np.linspace(0, n, chunks + 1).astype(np.int64)
    for s, e in zip(bounds[:-1], bounds[1:]):
        chunk = [v.slice(int(s), int(e)) for v in cols] if chunks > 1 else cols
        a.consume([dev.upload(v) for v in chunk], int(e - s))
    res = a.finish()
    out = dev.result_to_vecs(res)
    a.close()
    return out


def wrap_torch_ptr(ptr, n, dtype, device):
    """torch view (no copy) of `n` elements of device memory at `ptr`."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    typestr = {torch.int64: "<i8", torch.float64: "<f8", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    h.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(h, device=device)


def wrap_torch(col, n, dtype, device):
    """torch view (no copy) of the data buffer of a device column (library-owned result or input)."""
    return wrap_torch_ptr(col.data_ptr, n, dtype, device)
