"""GPU test of the one entry point that allocates nothing, crgpu_dev_partition_noalloc: called through the C-ABI,
its output must equal crgpu_dev_partition's (the existing entry point with the same kernels) bit for bit -- per
destination the same rows as multisets, the same counts -- on two calls in a row, the second with a smaller input
(the per-GPU static scratch keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

from cockroach_amd import _lib
from cockroach_amd import coldata as cd

pytestmark = pytest.mark.gpu


def _noalloc(dev, cols, hash_cols, P, n):
    types = [c.type for c in cols]
    dcols = [dev.upload(c) for c in cols]
    outs = [dev.DeviceBuffer(max(n, 1) * cd.NP_DTYPE[t]().itemsize) for t in types]
    counts = dev.DeviceBuffer(P * 8)
    out_ptrs = (C.c_void_p * len(outs))(*[o.ptr for o in outs])
    lib = _lib.load()
    mem_before = lib.crgpu_mem_usage()
    _lib.check(lib.crgpu_dev_partition_noalloc(dev._i32arr(types), len(types), dev._i32arr(hash_cols), len(hash_cols), 1, P,
                                               dev.dcol_array(dcols), n, out_ptrs, counts.ptr))
    assert lib.crgpu_mem_usage() == mem_before  # the library allocated nothing
    _lib.check(lib.crgpu_dev_synchronize())
    got_counts = counts.to_numpy(np.int64, P)
    got = [o.to_numpy(cd.NP_DTYPE[t], n) for o, t in zip(outs, types)]
    return got, got_counts


def _existing(dev, cols, hash_cols, P, n):
    p = dev.Partitioner([c.type for c in cols], hash_cols, P)
    res, counts = p.partition([dev.upload(c) for c in cols], n)
    out = [v.data.copy() for v in dev.result_to_vecs(res)]
    p.close()
    return out, counts


def _same_regions(a, ca, b, cb):
    assert ca.tolist() == cb.tolist()

    def rows(cols, start, k):  # the k rows of one region as a byte matrix, sorted
        m = np.concatenate([np.ascontiguousarray(x[start:start + k]).view(np.uint8).reshape(k, -1) for x in cols], axis=1)
        return m[np.lexsort(m.T[::-1])]

    start = 0
    for k in ca.tolist():
        if k:
            assert np.array_equal(rows(a, start, k), rows(b, start, k))
        start += k


def test_partition_noalloc_matches_partition_twice():
    from cockroach_amd import device as dev
    rng = np.random.default_rng(7)
    for n, P in [(1_000_003, 8), (70_001, 7)]:  # the second call is smaller: stale scratch must not matter
        k = cd.Vec.from_numpy(cd.INT64, rng.integers(-10**6, 10**6, size=n, dtype=np.int64))
        w = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n))
        b = cd.Vec.from_numpy(cd.INT32, rng.integers(-5, 5, size=n, dtype=np.int32))
        cols = [k, w, b]
        got, gc = _noalloc(dev, cols, [0, 2], P, n)
        want, wc = _existing(dev, cols, [0, 2], P, n)
        for g, wv in zip(got, want):
            assert g.dtype.itemsize == wv.dtype.itemsize
        _same_regions(got, gc, want, wc)


def test_partition_noalloc_known_answer_and_limits():
    from cockroach_amd import device as dev
    v = cd.Vec.from_numpy(cd.INT64, np.arange(1024, dtype=np.int64))
    got, counts = _noalloc(dev, [v], [0], 4, 1024)
    assert counts.tolist() == [273, 252, 287, 212]  # pkg/sql/colflow/routers_test.go:698-781
    assert sorted(got[0].tolist()) == list(range(1024))
    lib = _lib.load()
    d = dev.upload(v)
    out = dev.DeviceBuffer(8 * 1024)
    counts_b = dev.DeviceBuffer(8 * 128)
    ptrs = (C.c_void_p * 1)(out.ptr)
    # beyond the stated maxima: a clear error, nothing launched
    st = lib.crgpu_dev_partition_noalloc(dev._i32arr([cd.INT64]), 1, dev._i32arr([0]), 1, 1, 4, dev.dcol_array([d]), (1 << 22) + 1, ptrs, counts_b.ptr)
    assert st == _lib.ERR_UNSUPPORTED and b"2^22" in lib.crgpu_last_error()
    st = lib.crgpu_dev_partition_noalloc(dev._i32arr([cd.INT64]), 1, dev._i32arr([0]), 1, 1, 65, dev.dcol_array([d]), 1024, ptrs, counts_b.ptr)
    assert st == _lib.ERR_UNSUPPORTED
    # empty input
    got, counts = _noalloc(dev, [cd.Vec.from_numpy(cd.INT64, np.zeros(0, np.int64))], [0], 4, 0)
    assert counts.tolist() == [0, 0, 0, 0]
