"""crgpu_handle_mem_usage: the bytes one handle holds, for the caller's per-operator account (the reference gives every
operator its own colmem.Allocator, sql/colmem/allocator.go:34-60)."""
import ctypes as C

import numpy as np
import pytest

from cockroach_amd import _lib, coldata as cd

pytestmark = pytest.mark.gpu


def usage(handle):
    d, p = C.c_int64(-1), C.c_int64(-1)
    _lib.check(_lib.load().crgpu_handle_mem_usage(handle, C.byref(d), C.byref(p)))
    return d.value, p.value


def test_device_handles_are_accounted_separately():
    import torch
    from cockroach_amd import device as dev
    lib = _lib.load()
    _lib.check(lib.crgpu_init(0))
    n = 1 << 20
    k = torch.arange(n, device="cuda", dtype=torch.int64)
    v = torch.ones(n, device="cuda", dtype=torch.int64)
    torch.cuda.synchronize()
    a = dev.Agg([cd.INT64, cd.INT64], [0], [(0, 0, -1), (11, 1, -1)])
    b = dev.Agg([cd.INT64, cd.INT64], [0], [(0, 0, -1), (11, 1, -1)])
    try:
        a0, _ = usage(a.h)
        b0, _ = usage(b.h)
        a.consume([dev.from_torch(cd.INT64, k), dev.from_torch(cd.INT64, v)], n)
        a1, _ = usage(a.h)
        b1, _ = usage(b.h)
        assert a1 > a0 and a1 >= n * 16          # the groups' keys and sums at least
        assert b1 == b0                          # the other operator's account did not move
        b.consume([dev.from_torch(cd.INT64, k[:1024]), dev.from_torch(cd.INT64, v[:1024])], 1024)
        b2, _ = usage(b.h)
        assert b0 < b2 < a1
        assert a1 + b2 <= lib.crgpu_mem_usage()  # both are part of the process-wide figure
        res = a.finish()
        assert int(res.n) == n
    finally:
        a.close()
        held = usage(b.h)[0]
        assert held > 0                          # closing one handle leaves the other's account alone
        b.close()


def test_shell_resolves_to_its_device_handle_and_counts_pinned_staging():
    from cockroach_amd import operators as ops
    lib = _lib.load()
    _lib.check(lib.crgpu_init(0))
    n = 1 << 18
    keys = cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64) % 1000)
    vals = cd.Vec.from_numpy(cd.INT64, np.ones(n, dtype=np.int64))
    types = [cd.INT64, cd.INT64]
    batches = [cd.Batch([keys.take(np.arange(lo, min(n, lo + 1024))), vals.take(np.arange(lo, min(n, lo + 1024)))], min(n, lo + 1024) - lo)
               for lo in range(0, n, 1024)]
    ha = ops.HashAggregator(types, [0], [(0, 0, -1), (11, 1, -1)], ops.BatchSource(batches, types))
    try:
        ha.init()
        rows = 0
        while True:
            b = ha.next()
            if b.length == 0:
                break
            rows += b.length
        assert rows == 1000
        d, p = usage(ha.h)  # the shell handle: staged host rows (pinned) and the device handle's buffers
        assert d > 0 and p > 0
    finally:
        ha.close()


def test_rejects_what_is_not_a_handle():
    lib = _lib.load()
    junk = (C.c_uint8 * 64)()
    d = C.c_int64()
    assert lib.crgpu_handle_mem_usage(C.cast(junk, C.c_void_p), C.byref(d), None) == _lib.ERR_INVALID


def test_handle_stats_count_rows_calls_and_time():
    """crgpu_handle_stats: per-operator calls, busy time, rows in / out (ComponentStats, colflow/stats.go:41-137)."""
    import torch
    from cockroach_amd import device as dev
    lib = _lib.load()
    _lib.check(lib.crgpu_init(0))

    def stats(h):
        s = _lib.HandleStats()
        _lib.check(lib.crgpu_handle_stats(h, C.byref(s)))
        return s

    nb, npr = 1 << 16, 1 << 18
    bk = torch.randperm(nb, device="cuda", dtype=torch.int64)
    pk = torch.randint(0, 2 * nb, (npr,), device="cuda", dtype=torch.int64)
    torch.cuda.synchronize()
    j = dev.Join(0, [cd.INT64], [cd.INT64], [0], [0], True)
    other = dev.Join(0, [cd.INT64], [cd.INT64], [0], [0], True)
    try:
        c0 = stats(j.h).calls  # (create is a call on the handle too)
        assert c0 <= 1 and stats(j.h).rows_in == 0
        j.build([dev.from_torch(cd.INT64, bk)], nb)
        res = j.probe([dev.from_torch(cd.INT64, pk)], npr)
        hits = int((pk < nb).sum().item())
        assert int(res.n) == hits
        s = stats(j.h)
        assert s.calls == c0 + 2 and s.busy_ns > 0
        assert s.rows_in == nb + npr and s.rows_out == hits
        assert s.device_bytes > 0
        o = stats(other.h)
        assert o.calls <= 1 and o.rows_in == 0 and o.rows_out == 0  # another operator's statistics are its own
    finally:
        j.close()
        other.close()
