"""GPU parity: hash aggregation over partially ordered input (AggregatorSpec.OrderedGroupCols;
hash_aggregator_tmpl.go:160-169, 256-306).  The reference's own cases (hash_aggregator_test.go:104-397: oneOrderedCol,
twoOrderedCol, PartialOrderWithNulls, orderedColSpansMultMaxBufferBoundary, orderedColUnique, spilloverAfterEmit) are
checked the way the reference checks them -- colexectestutils.PartialOrderedVerifier: the runs of equal ordered values
in input order, any order inside a run -- at layer 1 (crgpu_dev_agg_emit_ready) and through the operator; random inputs
against the oracle, whose rows are cut into the same runs."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from cockroach_amd import device as dev
from cockroach_amd import operators as ops
from cockroach_amd import _lib
from tests import helpers as H

pytestmark = pytest.mark.gpu
F = H.AGG_FUNC
CASES = H.ordered_agg_cases()


def _ordered_first(group_cols, ordered_cols):
    return list(ordered_cols) + [g for g in group_cols if g not in ordered_cols]


def run_ordered(cols, group_cols, ordered_cols, aggs, bounds):
    """Feeds rows [bounds[i], bounds[i+1]) per consume call; returns (rows, rows handed out before finish)."""
    a = dev.Agg([v.type for v in cols], _ordered_first(group_cols, ordered_cols), aggs, n_ordered=len(ordered_cols))
    rows, early = [], 0
    try:
        for s, e in zip(bounds[:-1], bounds[1:]):
            if e == s:
                continue
            a.consume([dev.upload(v.slice(int(s), int(e))) for v in cols], int(e - s))
            while True:
                r = a.emit_ready()
                if r.n == 0:
                    break
                rows += H.vecs_to_tuples(dev.result_to_vecs(r))
        early = len(rows)
        r = a.finish()
        while r.n > 0:
            rows += H.vecs_to_tuples(dev.result_to_vecs(r))
            r = a.emit_ready()
    finally:
        a.close()
    return rows, early


def test_the_reference_has_ordered_cases():
    assert {c["name"] for c in CASES} >= {"oneOrderedCol", "twoOrderedCol", "PartialOrderWithNulls", "orderedColUnique",
                                          "orderedColSpansMultMaxBufferBoundary", "spilloverAfterEmit"}


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_ordered_agg_golden_case_layer1(case):
    cols, group_cols, aggs, expected = H.prepare_agg_case(case)
    n = len(case["input"])
    oc = case["ordered_cols"]
    for step in (n, 1, 3, 5):
        bounds = list(range(0, n, step)) + [n]
        rows, _ = run_ordered(cols, group_cols, oc, aggs, bounds)
        H.assert_partial_order(rows, expected, oc, f"{case['id']} step={step}")


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_ordered_agg_golden_case_operator(case):
    cols, group_cols, aggs, expected = H.prepare_agg_case(case)
    types = [v.type for v in cols]
    for batch_size in (1, 2, 3, 16, 1024):
        n = len(cols[0])
        batches = [cd.Batch([v.slice(s, min(n, s + batch_size)) for v in cols], min(n, s + batch_size) - s)
                   for s in range(0, n, batch_size)]
        src = ops.BatchSource(batches, types)
        ha = ops.HashAggregator(types, group_cols, aggs, src, capacity=4, ordered_cols=case["ordered_cols"])
        ha.init()
        for attempt in ("first run", "after Reset"):
            rows = []
            while True:
                b = ha.next()
                if b.length == 0:
                    break
                rows += H.vecs_to_tuples(b.cols, b.length)
            H.assert_partial_order(rows, expected, case["ordered_cols"], f"{case['id']} bs={batch_size} {attempt}")
            ha.reset()
        ha.close()


def _oracle_in_run_order(oracle, cols, group_cols, ordered_cols, aggs):
    """The oracle's groups, ordered by the first input row of their run of ordered values."""
    want = H.vecs_to_tuples(oracle.hash_agg(cols, group_cols, aggs))
    first = {}
    vals = [cols[c].to_values() for c in ordered_cols]
    for i in range(len(cols[0])):
        first.setdefault(tuple(repr(v[i]) for v in vals), i)
    # output column p carries ANY_NOT_NULL of ordered column p (the aggregations below are laid out that way)
    want.sort(key=lambda r: first[tuple(repr(r[p]) for p in range(len(ordered_cols)))])
    return want


@pytest.mark.parametrize("kind", ["int", "bytes", "two"])
def test_ordered_agg_random_vs_oracle(oracle, kind):
    rng = np.random.default_rng({"int": 1, "bytes": 2, "two": 3}[kind])
    n = 120_000
    run = np.sort(rng.integers(0, 900, size=n))  # ~130 rows per run; the first run is the NULL run
    if kind == "bytes":
        o0 = cd.Vec.from_values(cd.BYTES, [None if r == 0 else b"k%05d" % r + b"x" * int(r % 11) for r in run])
    else:
        o0 = cd.Vec.from_numpy(cd.INT64, run * 7 - 1000, run != 0)
    o1 = cd.Vec.from_numpy(cd.INT32, (np.arange(n) // 17 % 5).astype(np.int32)) if kind == "two" else None
    g = cd.Vec.from_numpy(cd.INT16, rng.integers(0, 9, size=n).astype(np.int16), rng.random(n) > 0.05)
    x = cd.Vec.from_numpy(cd.INT64, rng.integers(-10**9, 10**9, size=n, dtype=np.int64), rng.random(n) > 0.1)
    f = cd.Vec.from_numpy(cd.FLOAT64, rng.random(n) * 64 - 32, rng.random(n) > 0.1)
    s = cd.Vec.from_values(cd.BYTES, [None if i % 13 == 0 else b"v%d" % (i % 1000) for i in range(n)])
    if kind == "two":
        # the second ordered column is only sorted inside a run of the first
        o1 = cd.Vec.from_numpy(cd.INT32, np.concatenate([np.sort((np.arange(c) * 5 // max(c, 1)).astype(np.int32))
                                                        for c in np.bincount(run)]).astype(np.int32))
        cols, ordered, group = [o0, o1, g, x, f, s], [0, 1], [0, 1, 2]
        lead = [(F["ANY_NOT_NULL"], 0, -1), (F["ANY_NOT_NULL"], 1, -1), (F["ANY_NOT_NULL"], 2, -1)]
        xi, fi, si = 3, 4, 5
    else:
        cols, ordered, group = [o0, g, x, f, s], [0], [1, 0]  # (grouping columns given in the other order on purpose)
        lead = [(F["ANY_NOT_NULL"], 0, -1), (F["ANY_NOT_NULL"], 1, -1)]
        xi, fi, si = 2, 3, 4
    aggs = lead + [(F["COUNT_ROWS"], -1, -1), (F["SUM_INT"], xi, -1), (F["MIN"], xi, -1), (F["SUM"], fi, -1), (F["AVG"], fi, -1),
                   (F["MAX"], si, -1), (F["COUNT"], si, -1), (F["SUM"], xi, -1)]
    want = _oracle_in_run_order(oracle, cols, group, ordered, aggs)
    for bounds in ([0, n], [0, 1, 7, 4099, 4100, 50_001, 99_999, n], list(range(0, n, 4096)) + [n]):
        rows, early = run_ordered(cols, group, ordered, aggs, bounds)
        H.assert_partial_order(rows, want, list(range(len(ordered))), f"{kind} {len(bounds) - 1} calls")
        assert early > 0.9 * len(rows), "groups of completed runs must be handed out before the input ends"


def test_ordered_agg_with_distinct_and_filter_aggregations(oracle):
    """The ordered form wraps an ordinary aggregator, so DISTINCT and FILTER aggregations come along: per run of the
    ordered column, COUNT(DISTINCT x), SUM(DISTINCT x) FILTER (WHERE f), COUNT_ROWS FILTER (WHERE f) -- the DISTINCT
    tables are reset with every piece, which is right because a group never spans two pieces."""
    rng = np.random.default_rng(9)
    n = 60_000
    run = np.sort(rng.integers(0, 400, size=n))
    o0 = cd.Vec.from_numpy(cd.INT64, run * 3)
    g = cd.Vec.from_numpy(cd.INT32, rng.integers(0, 4, size=n).astype(np.int32))
    x = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 12, size=n, dtype=np.int64), rng.random(n) > 0.1)
    f = cd.Vec.from_numpy(cd.BOOL, rng.integers(0, 2, size=n).astype(np.uint8), rng.random(n) > 0.05)
    cols = [o0, g, x, f]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["ANY_NOT_NULL"], 1, -1), (F["COUNT"], 2, -1, True), (F["SUM_INT"], 2, 3, True),
            (F["COUNT_ROWS"], -1, 3), (F["SUM_INT"], 2, -1)]
    want = _oracle_in_run_order(oracle, cols, [0, 1], [0], aggs)
    for bounds in ([0, n], [0, 5, 8191, 30_001, n], list(range(0, n, 2048)) + [n]):
        rows, early = run_ordered(cols, [0, 1], [0], aggs, bounds)
        H.assert_partial_order(rows, want, [0], f"{len(bounds) - 1} calls")


def test_ordered_agg_one_long_run_and_reset(oracle):
    """No change of the ordered column at all: nothing is ready before finish; a second use after reset gives the same."""
    rng = np.random.default_rng(5)
    n = 50_000
    o0 = cd.Vec.from_numpy(cd.INT64, np.full(n, 42, dtype=np.int64))
    g = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 3000, size=n, dtype=np.int64))
    x = cd.Vec.from_numpy(cd.FLOAT64, rng.random(n))
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["ANY_NOT_NULL"], 1, -1), (F["SUM"], 2, -1), (F["COUNT_ROWS"], -1, -1)]
    want = H.vecs_to_tuples(oracle.hash_agg([o0, g, x], [0, 1], aggs))
    a = dev.Agg([cd.INT64, cd.INT64, cd.FLOAT64], [0, 1], aggs, n_ordered=1)
    for _ in range(2):
        for s in range(0, n, 10_000):
            a.consume([dev.upload(v.slice(s, s + 10_000)) for v in (o0, g, x)], 10_000)
            assert a.emit_ready().n == 0
        rows = H.vecs_to_tuples(dev.result_to_vecs(a.finish()))
        assert a.emit_ready().n == 0
        H.assert_same_multiset(rows, want)
        a.reset()
    a.close()


def test_ordered_agg_spec_errors():
    with pytest.raises(_lib.CrgpuError):  # more ordered columns than grouping columns
        dev.Agg([cd.INT64, cd.INT64], [0], [(F["COUNT_ROWS"], -1, -1)], n_ordered=2)
    with pytest.raises(_lib.CrgpuError) as e:
        import ctypes as C  # (spec built by hand: the python FusedAgg class has no n_ordered argument)
        types = dev._i32arr([cd.FLOAT64, cd.FLOAT64, cd.FLOAT64])
        groups = dev._i32arr([0])
        descs = (_lib.AggDesc * 1)(_lib.AggDesc(F["SUM"], 2, -1, 0))
        vc = (_lib.AggVcol * 1)(_lib.AggVcol(0, 0, 0, 1, 0))
        spec = _lib.AggSpec(3, 1, types, groups, 1, 1, descs)
        h = C.c_void_p()
        _lib.check(_lib.load().crgpu_dev_agg_create_fused(C.byref(spec), 1, vc, C.byref(h)))
    assert e.value.code == _lib.ERR_UNSUPPORTED
    # an unordered handle never has anything ready
    a = dev.Agg([cd.INT64], [0], [(F["COUNT_ROWS"], -1, -1)])
    a.consume([dev.upload(cd.Vec.from_numpy(cd.INT64, np.arange(10, dtype=np.int64)))], 10)
    assert a.emit_ready().n == 0
    assert a.finish().n == 10
    a.close()


def test_ordered_agg_input_tracking_drops_emitted_runs(oracle):
    """Planned under a disk spiller (crgpu_agg_track_input) the shell keeps the input for ExportBuffered; with ordered
    grouping columns the reference drops the tuples of a run from that tracking once the run's groups have been emitted
    (resetBucketsAndTrackingState, hash_aggregator_tmpl.go:355-378).  Here: 1.3 M rows in runs of 1000; after the first
    device round trip (2^20 staged rows) the completed runs' groups are read, then ExportBuffered must hand out exactly
    the rows that belong to no emitted group -- a contiguous tail of what was appended -- and emitted groups + the
    aggregation of that tail must be the aggregation of everything appended.  Appending while completed groups are
    unread is a state error."""
    import ctypes as C
    from tests.test_gpu_operators import _export_all
    rng = np.random.default_rng(77)
    n, bs = 1_300_000, 4096
    o0 = cd.Vec.from_numpy(cd.INT64, (np.arange(n) // 1000).astype(np.int64))
    g = cd.Vec.from_numpy(cd.INT32, rng.integers(0, 3, size=n).astype(np.int32))
    x = cd.Vec.from_numpy(cd.INT64, rng.integers(-1000, 1000, size=n, dtype=np.int64))
    cols, types = [o0, g, x], [cd.INT64, cd.INT32, cd.INT64]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["ANY_NOT_NULL"], 1, -1), (F["SUM_INT"], 2, -1), (F["COUNT_ROWS"], -1, -1)]
    batches = [cd.Batch([v.slice(s, min(n, s + bs)) for v in cols], min(n, s + bs) - s) for s in range(0, n, bs)]
    ha = ops.HashAggregator(types, [0, 1], aggs, ops.BatchSource(batches, types), capacity=1024, track_input=True, ordered_cols=[0])
    ha.init()
    fed = 0
    emitted = []
    for b in batches:
        mb, _keep = ops._marshal_batch(b)
        _lib.check(ha.lib.crgpu_agg_append(ha.h, C.byref(mb)))
        fed += b.length
        out = ha.out.call(ha.lib.crgpu_agg_next_output, ha.h)
        if out.length > 0:  # the first device round trip has happened: completed runs are ready
            emitted += H.vecs_to_tuples(out.cols, out.length)
            nxt = batches[fed // bs]
            mb2, _k2 = ops._marshal_batch(nxt)
            with pytest.raises(_lib.CrgpuError) as e:
                _lib.check(ha.lib.crgpu_agg_append(ha.h, C.byref(mb2)))
            assert e.value.code == _lib.ERR_STATE
            while True:
                out = ha.out.call(ha.lib.crgpu_agg_next_output, ha.h)
                if out.length == 0:
                    break
                emitted += H.vecs_to_tuples(out.cols, out.length)
            break
    assert emitted and fed >= (1 << 20)
    ha.release_before_export()
    tail = _export_all(ha)
    all_rows = H.vecs_to_tuples(cols)
    assert 0 < len(tail) < 5000, "only the rows behind the last completed run are still tracked"
    assert tail == all_rows[fed - len(tail):fed]
    assert max(r[0] for r in emitted) < min(r[0] for r in tail)
    tail_cols = [v.slice(fed - len(tail), fed) for v in cols]
    rest = H.vecs_to_tuples(oracle.hash_agg(tail_cols, [0, 1], aggs)) if tail else []
    want = H.vecs_to_tuples(oracle.hash_agg([v.slice(0, fed) for v in cols], [0, 1], aggs))
    H.assert_same_multiset(emitted + rest, want, "emitted groups + aggregation of the exported tail")
    ha.release_after_export()
    ha.close()


def test_ordered_agg_large_single_call(oracle):
    """1.5 M rows in ONE call, runs of two rows: 750 000 groups in one piece -- more rows and more groups than a capped
    launch grid has threads, NULLs in the value column so that the second sub-batch needs a shifted validity bitmap."""
    rng = np.random.default_rng(12)
    n = 1_500_001
    o0 = cd.Vec.from_numpy(cd.INT64, (np.arange(n) // 2).astype(np.int64) * 5 - 3)
    x = cd.Vec.from_numpy(cd.INT64, rng.integers(-50, 50, size=n, dtype=np.int64), rng.random(n) > 0.2)
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT"], 1, -1), (F["COUNT_ROWS"], -1, -1)]
    want = H.vecs_to_tuples(oracle.hash_agg([o0, x], [0], aggs))
    want.sort(key=lambda r: r[0])
    rows, early = run_ordered([o0, x], [0], [0], aggs, [0, n])
    assert rows == want  # (one group per run: the partial order is a total order here)
    assert early == len(rows) - 1
