"""GPU tests of the library runtime: per-handle streams driven from several threads at once, the thread/device
binding (crgpu_init on one thread, work from another: hipSetDevice is per OS thread), cancellation, shutdown
refused under live handles."""
import threading

import numpy as np
import pytest

from cockroach_amd import _lib
from cockroach_amd import coldata as cd

pytestmark = pytest.mark.gpu


def _join_once(dev, oracle, seed, nb=20_000, npr=200_000):
    rng = np.random.default_rng(seed)
    bk = cd.Vec.from_numpy(cd.INT64, rng.permutation(nb * 2)[:nb].astype(np.int64))
    bv = cd.Vec.from_numpy(cd.INT64, bk.data * 3 + 1)
    pk = cd.Vec.from_numpy(cd.INT64, rng.integers(0, nb * 2, size=npr, dtype=np.int64))
    pw = cd.Vec.from_numpy(cd.INT64, np.arange(npr, dtype=np.int64))
    j = dev.Join(0, [cd.INT64, cd.INT64], [cd.INT64, cd.INT64], [0], [0], True)
    j.build([dev.upload(bk), dev.upload(bv)], nb)
    got = dev.result_to_vecs(j.probe([dev.upload(pk), dev.upload(pw)], npr))
    j.close()
    g = np.stack([v.data for v in got], 1)
    hit = np.isin(pk.data, bk.data)
    want = np.stack([pk.data[hit], pw.data[hit], pk.data[hit], pk.data[hit] * 3 + 1], 1)
    assert np.array_equal(g[np.lexsort(g.T[::-1])], want[np.lexsort(want.T[::-1])])


def _agg_once(dev, seed, n=300_000):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 1000, size=n, dtype=np.int64)
    x = rng.integers(-100, 100, size=n, dtype=np.int64)
    a = dev.Agg([cd.INT64, cd.INT64], [0], [(0, 0, -1), (11, 1, -1), (14, -1, -1)])
    half = n // 2
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k[:half])), dev.upload(cd.Vec.from_numpy(cd.INT64, x[:half]))], half)
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, k[half:])), dev.upload(cd.Vec.from_numpy(cd.INT64, x[half:]))], n - half)
    got = dev.result_to_vecs(a.finish())
    a.close()
    g = np.stack([v.data for v in got], 1)
    keys = np.unique(k)
    want = np.stack([keys, np.array([x[k == q].sum() for q in keys]), np.array([(k == q).sum() for q in keys])], 1)
    assert np.array_equal(g[np.argsort(g[:, 0])], want)


def test_two_handles_from_two_threads(oracle):
    """A joiner and an aggregator driven concurrently from two threads (two goroutines' operator chains): each handle
    owns its stream, results stay exact.  crgpu_init ran on the main thread only."""
    from cockroach_amd import device as dev
    _lib.check(_lib.load().crgpu_init(0))
    errs = []

    def run(fn, *args):
        try:
            for r in range(4):
                fn(*args, r)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=run, args=(lambda s: _join_once(dev, oracle, s),)),
          threading.Thread(target=run, args=(lambda s: _agg_once(dev, s),)),
          threading.Thread(target=run, args=(lambda s: _join_once(dev, oracle, 100 + s),))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_init_on_one_thread_work_on_another():
    from cockroach_amd import device as dev
    lib = _lib.load()
    out = {}

    def init():
        out["st"] = lib.crgpu_init(0)

    t = threading.Thread(target=init)
    t.start()
    t.join()
    assert out["st"] == _lib.OK
    done = []
    t2 = threading.Thread(target=lambda: (_agg_once(dev, 5), done.append(1)))
    t2.start()
    t2.join()
    assert done == [1]


def test_cancel_and_reset():
    from cockroach_amd import device as dev
    lib = _lib.load()
    k = cd.Vec.from_numpy(cd.INT64, np.arange(1000, dtype=np.int64))
    a = dev.Agg([cd.INT64], [0], [(0, 0, -1), (14, -1, -1)])
    a.consume([dev.upload(k)], 1000)
    assert lib.crgpu_cancel(a.h) == _lib.OK
    with pytest.raises(_lib.CrgpuError) as ei:
        a.consume([dev.upload(k)], 1000)
    assert ei.value.code == _lib.ERR_CANCELLED
    with pytest.raises(_lib.CrgpuError):
        a.finish()
    a.reset()  # Reset clears the flag (colexecop.Resetter)
    a.consume([dev.upload(k)], 1000)
    assert int(a.finish().n) == 1000
    a.close()
    assert lib.crgpu_cancel(None) == _lib.ERR_INVALID


def test_shutdown_refused_under_live_handles():
    from cockroach_amd import device as dev
    lib = _lib.load()
    a = dev.Agg([cd.INT64], [0], [(14, -1, -1)])
    live = lib.crgpu_live_handles()
    assert live >= 1
    lib.crgpu_shutdown()
    assert b"still alive" in lib.crgpu_last_error()
    k = cd.Vec.from_numpy(cd.INT64, np.arange(10, dtype=np.int64))
    a.consume([dev.upload(k)], 10)  # the handle still works
    assert int(a.finish().n) == 10
    a.close()
    assert lib.crgpu_live_handles() == live - 1


def test_caller_blocks_come_from_the_free_list():
    """crgpu_dev_malloc / crgpu_dev_free share the per-device free list of the library's own buffers: a freed block is
    kept (crgpu_mem_cached grows by its size) and handed to the next request of about that size; blocks the caller holds
    are not part of crgpu_mem_usage; crgpu_mem_trim returns the idle ones to the driver; freeing NULL and a pointer the
    library never handed out are harmless."""
    import ctypes as C
    lib = _lib.load()
    _lib.check(lib.crgpu_init(0))
    lib.crgpu_mem_cached.restype = C.c_int64
    lib.crgpu_mem_usage.restype = C.c_int64
    lib.crgpu_mem_trim()
    used0, cached0 = lib.crgpu_mem_usage(), lib.crgpu_mem_cached()
    size = (37 << 20) + 123
    p = C.c_void_p()
    _lib.check(lib.crgpu_dev_malloc(C.byref(p), C.c_size_t(size)))
    assert p.value and lib.crgpu_mem_usage() == used0
    first = p.value
    lib.crgpu_dev_free(p)
    grown = lib.crgpu_mem_cached() - cached0
    assert size <= grown < size + 4096
    q = C.c_void_p()
    _lib.check(lib.crgpu_dev_malloc(C.byref(q), C.c_size_t(size - 1000)))  # within a quarter of the cached block
    assert q.value == first and lib.crgpu_mem_cached() == cached0
    lib.crgpu_dev_free(q)
    lib.crgpu_dev_free(None)
    lib.crgpu_mem_trim()
    assert lib.crgpu_mem_cached() == 0

