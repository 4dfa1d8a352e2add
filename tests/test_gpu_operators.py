"""GPU parity of the operator boundary (layer 2 of include/crgpu.h) driven the way
the reference's test runner drives operators (colexectestutils.RunTestsWithTyps,
utils.go:330-420): the same case tables run with several input batch sizes,
with and without selection vectors, with small output batches, and after Reset.
Batches cross the C-ABI in the coldata layout (32-byte Bytes elements)."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from cockroach_amd import operators as ops
from tests import helpers as H

pytestmark = pytest.mark.gpu

JOIN_CASES = H.join_cases()
AGG_CASES = H.agg_cases()


def make_batches(vecs, batch_size, use_sel, rng):
    """Split columns into batches of <= batch_size tuples.  With use_sel the physical
    batch holds extra rows (copies of other tuples) and a selection vector picks the
    real ones in order -- the reference's runner does the same (utils.go:1044-1100)."""
    n = len(vecs[0]) if vecs else 0
    out = []
    for a in range(0, n, batch_size):
        b = min(n, a + batch_size)
        m = b - a
        if not use_sel:
            out.append(cd.Batch([v.slice(a, b) for v in vecs], m))
            continue
        phys = 2 * m + 1
        sel = np.sort(rng.choice(phys, size=m, replace=False)).astype(np.int64)
        src = rng.integers(0, n, size=phys)  # garbage rows: copies of arbitrary tuples
        src[sel] = np.arange(a, b)
        out.append(cd.Batch([v.take(src) for v in vecs], m, sel))
    return out


def drain(op, limit=100000):
    rows = []
    for _ in range(limit):
        b = op.next()
        if b.length == 0:
            return rows
        rows += H.vecs_to_tuples(b.cols, b.length)
    raise AssertionError("operator did not terminate")


def run_join(case_or_args, left, right, batch_size, use_sel, capacity, rng, lookahead=1, reset_rounds=0):
    jt, lt, rt, leq, req, key = case_or_args
    lsrc = ops.BatchSource(make_batches(left, batch_size, use_sel, rng), lt)
    rsrc = ops.BatchSource(make_batches(right, batch_size, use_sel, rng), rt)
    hj = ops.HashJoiner(jt, lt, rt, leq, req, key, lsrc, rsrc, capacity=capacity, lookahead=lookahead)
    hj.init()
    rows = drain(hj)
    for _ in range(reset_rounds):
        hj.reset()
        again = drain(hj)
        H.assert_same_multiset(again, rows, "after Reset")
    hj.close()
    return rows


@pytest.mark.parametrize("case", JOIN_CASES, ids=[c["id"] for c in JOIN_CASES])
def test_hash_joiner_golden_cases(case):
    expected = H.expected_join_rows(case)
    lt = [H.type_id(t) for t in case["left_types"]]
    rt = [H.type_id(t) for t in case["right_types"]]
    left = H.tuples_to_vecs(case["left"], lt)
    right = H.tuples_to_vecs(case["right"], rt)
    args = (H.JOIN_TYPE[case["join_type"]], lt, rt, case["left_eq"], case["right_eq"], case["right_eq_cols_are_key"])
    rng = np.random.default_rng(7)
    for batch_size, use_sel, capacity in ((1024, False, 1024), (3, True, 2), (1, False, 1024)):
        rows = run_join(args, left, right, batch_size, use_sel, capacity, rng,
                        reset_rounds=1 if batch_size == 1024 else 0)
        H.assert_same_multiset(H.project_join_output(case, rows), expected,
                               f"{case['id']} batch={batch_size} sel={use_sel} cap={capacity}")


@pytest.mark.parametrize("jt", ["INNER", "LEFT_OUTER", "FULL_OUTER", "LEFT_ANTI", "RIGHT_SEMI", "EXCEPT_ALL"])
def test_hash_joiner_random_vs_oracle(oracle, jt):
    rng = np.random.default_rng(123)
    nl, nr = 20_000, 3_000
    lk = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 4000, nl, dtype=np.int64), rng.random(nl) > 0.03)
    rk = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 4000, nr, dtype=np.int64), rng.random(nr) > 0.03)
    left, right = [lk], [rk]
    if jt != "EXCEPT_ALL":
        left.append(cd.Vec.from_values(cd.BYTES, [b"L%d" % i * (1 + i % 9) for i in range(nl)]))  # inline + spilled values
        right.append(cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(nr)))
    lt, rt = [v.type for v in left], [v.type for v in right]
    want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], left, right, [0], [0], False))
    got = run_join((H.JOIN_TYPE[jt], lt, rt, [0], [0], False), left, right, 1024, True, 1024, rng, lookahead=8)
    H.assert_same_multiset(got, want, jt)


def test_hash_joiner_projection_and_bytes_spill_growth(oracle):
    rng = np.random.default_rng(5)
    n = 500
    lk = cd.Vec.from_numpy(cd.INT32, rng.integers(0, 50, n).astype(np.int32))
    lp = cd.Vec.from_values(cd.BYTES, [bytes([65 + i % 26]) * (25 + i % 80) for i in range(n)])
    rk = cd.Vec.from_numpy(cd.INT64, np.arange(50, dtype=np.int64))
    rp = cd.Vec.from_values(cd.BYTES, [b"r" * (i * 3) for i in range(50)])
    want = H.vecs_to_tuples(oracle.hash_join(0, [lk, lp], [rk, rp], [0], [0], True))
    want = [(r[1], r[3]) for r in want]
    lsrc = ops.BatchSource(make_batches([lk, lp], 64, False, rng), [cd.INT32, cd.BYTES])
    rsrc = ops.BatchSource(make_batches([rk, rp], 64, False, rng), [cd.INT64, cd.BYTES])
    hj = ops.HashJoiner(0, [cd.INT32, cd.BYTES], [cd.INT64, cd.BYTES], [0], [0], True, lsrc, rsrc, out_cols=[1, 3],
                        capacity=100)
    hj.init()
    got = drain(hj)
    assert hj.out.grow_retries > 0  # the "bytes_buf too small -> length 0 + required size" protocol was exercised
    hj.close()
    H.assert_same_multiset(got, want)


def test_hash_joiner_export_buffered_and_state_errors():
    from cockroach_amd import _lib
    rng = np.random.default_rng(9)
    rk = cd.Vec.from_numpy(cd.INT64, np.arange(3000, dtype=np.int64), np.arange(3000) % 11 != 0)
    rp = cd.Vec.from_values(cd.BYTES, [b"v%d" % i * (1 + i % 20) for i in range(3000)])
    lk = cd.Vec.from_numpy(cd.INT64, np.arange(10, dtype=np.int64))
    lsrc = ops.BatchSource(make_batches([lk], 1024, False, rng), [cd.INT64])
    rsrc = ops.BatchSource(make_batches([rk, rp], 1000, True, rng), [cd.INT64, cd.BYTES])
    hj = ops.HashJoiner(0, [cd.INT64], [cd.INT64, cd.BYTES], [0], [0], True, lsrc, rsrc)
    hj.init()
    hj._build()
    # ExportBuffered while the build rows are still staged on the host ...
    rows = []
    while True:
        b = hj.export_buffered()
        if b.length == 0:
            break
        rows += H.vecs_to_tuples(b.cols, b.length)
    assert rows == H.vecs_to_tuples([rk, rp])  # same tuples, same order
    with pytest.raises(_lib.CrgpuError) as e:
        mb, _keep = ops._marshal_batch(cd.Batch([rk, rp], 3000))
        import ctypes as C
        _lib.check(hj.lib.crgpu_hj_build_append(hj.h, C.byref(mb)))
    assert e.value.code == _lib.ERR_STATE
    assert len(drain(hj)) == 9  # key 0 is NULL on the build side
    hj.close()
    # ... and after probing started (the rows then come back from HBM)
    lsrc.reset(), rsrc.reset()
    hj = ops.HashJoiner(0, [cd.INT64], [cd.INT64, cd.BYTES], [0], [0], True, lsrc, rsrc)
    hj.init()
    assert hj.next().length == 9
    rows = []
    while True:
        b = hj.export_buffered()
        if b.length == 0:
            break
        rows += H.vecs_to_tuples(b.cols, b.length)
    assert rows == H.vecs_to_tuples([rk, rp])
    hj.close()


def test_host_hash_cols_with_selection_vector(oracle):
    rng = np.random.default_rng(3)
    n = 5000
    a = cd.Vec.from_numpy(cd.INT64, rng.integers(-1000, 1000, n, dtype=np.int64), rng.random(n) > 0.1)
    b = cd.Vec.from_values(cd.BYTES, [None if i % 17 == 0 else b"k%d" % i * (i % 12) for i in range(n)])
    c = cd.Vec.from_numpy(cd.FLOAT64, rng.choice([0.0, -0.0, np.nan, 1.25], n))
    sel = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.int64)
    dense = [v.take(sel) for v in (a, b, c)]
    for width in (32, 64):
        got = ops.hash_cols([a, b, c], sel=sel, width=width)
        assert np.array_equal(got, oracle.hash_cols(dense, width=width))
    got = ops.hash_cols([a], width=32, nbuckets=1 << 10)
    assert np.array_equal(got, oracle.hash_cols([a], width=32, nbuckets=1 << 10))


@pytest.mark.parametrize("case", AGG_CASES, ids=[c["id"] for c in AGG_CASES])
def test_hash_aggregator_golden_cases(case):
    cols, group_cols, aggs, expected = H.prepare_agg_case(case)
    if not aggs:
        pytest.skip("no aggregation of this case is in scope")
    types = [v.type for v in cols]
    rng = np.random.default_rng(1)
    for batch_size, use_sel, capacity in ((1024, False, 1024), (2, True, 1), (4096, True, 3)):
        src = ops.BatchSource(make_batches(cols, batch_size, use_sel, rng), types)
        ha = ops.HashAggregator(types, group_cols, aggs, src, capacity=capacity)
        ha.init()
        rows = drain(ha)
        H.assert_same_multiset(rows, expected, f"{case['id']} batch={batch_size} sel={use_sel}")
        ha.reset()
        H.assert_same_multiset(drain(ha), expected, f"{case['id']} after Reset")
        ha.close()


def test_hash_aggregator_random_vs_oracle(oracle):
    rng = np.random.default_rng(77)
    n = 50_000
    g1 = cd.Vec.from_numpy(cd.INT32, rng.integers(0, 300, n).astype(np.int32), rng.random(n) > 0.02)
    g2 = cd.Vec.from_values(cd.BYTES, [[b"a", b"bb", b"c" * 40, None][i] for i in rng.integers(0, 4, n)])
    x = cd.Vec.from_numpy(cd.INT64, rng.integers(-10**6, 10**6, n, dtype=np.int64), rng.random(n) > 0.1)
    f = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n), rng.random(n) > 0.1)
    cols = [g1, g2, x, f]
    aggs = [(0, 0, -1), (0, 1, -1), (11, 2, -1), (10, 3, -1), (1, 3, -1), (7, 2, -1), (8, 3, -1), (5, 3, -1), (14, -1, -1),
            (7, 1, -1), (8, 1, -1)]  # ... and MIN / MAX over the BYTES column
    want = H.vecs_to_tuples(oracle.hash_agg(cols, [0, 1], aggs))
    types = [v.type for v in cols]
    src = ops.BatchSource(make_batches(cols, 1024, True, rng), types)
    ha = ops.HashAggregator(types, [0, 1], aggs, src)
    ha.init()
    H.assert_same_multiset(drain(ha), want)
    ha.close()


@pytest.mark.parametrize("num_outputs", [1, 2, 4, 7])
def test_hash_router_vs_distributor(oracle, num_outputs):
    rng = np.random.default_rng(31)
    n = 30_000
    k = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 10**9, n, dtype=np.int64), rng.random(n) > 0.05)
    s = cd.Vec.from_values(cd.BYTES, [b"s%d" % i * (1 + i % 13) for i in range(n)])
    cols, types = [k, s], [cd.INT64, cd.BYTES]
    dest, counts = oracle.distribute(cols, [0], num_outputs)
    src = ops.BatchSource(make_batches(cols, 1024, True, rng), types)
    r = ops.HashRouter(types, [0], num_outputs, src)
    r.run()
    all_rows = H.vecs_to_tuples(cols)
    for p, out in enumerate(r.outputs):
        got = drain(out)
        want = [all_rows[i] for i in np.nonzero(dest == p)[0]]
        assert len(got) == counts[p]
        H.assert_same_multiset(got, want, f"output {p}")
    r.close()


def test_hash_router_known_answer():
    """routers_test.go:812-870: keys 0..1023 over 4 outputs -> {273, 252, 287, 212} rows."""
    k = cd.Vec.from_numpy(cd.INT64, np.arange(1024, dtype=np.int64))
    src = ops.BatchSource([cd.Batch([k], 1024)], [cd.INT64])
    r = ops.HashRouter([cd.INT64], [0], 4, src)
    r.run()
    assert [len(drain(o)) for o in r.outputs] == [273, 252, 287, 212]
    r.close()


# ---------------------------------------------------------------- unordered distinct (layer 2)
from tests.test_oracle_distinct import CASES as DISTINCT_CASES, ordered_equal  # noqa: E402


@pytest.mark.parametrize("case", DISTINCT_CASES, ids=[c["id"] for c in DISTINCT_CASES])
def test_unordered_distinct_golden_cases(case):
    """distinct_test.go:45-330 through the operator boundary: host batches (with and without selection vectors), small
    output batches, Reset, and the errorOnDup plans."""
    rng = np.random.default_rng(3)
    types = [H.type_id(t) for t in case["types"]]
    cols = H.tuples_to_vecs(case["tuples"], case["types"])
    for batch_size, use_sel, capacity, lookahead in ((1, False, 1, 1), (2, True, 3, 2), (1024, False, 1024, 1), (3, True, 2, 1)):
        src = ops.BatchSource(make_batches(cols, batch_size, use_sel, rng), types)
        ud = ops.UnorderedDistinct(types, case["distinct_cols"], src, case["nulls_are_distinct"], case["error_on_dup"],
                                   capacity=capacity, lookahead=lookahead)
        ud.init()
        if case["error_on_dup"] and not case["no_error"]:
            with pytest.raises(ops.DuplicateError) as e:
                drain(ud)
            assert str(e.value) == case["error_on_dup"]
            ud.close()
            continue
        rows = drain(ud)
        ordered_equal(rows, case["expected"], f"{case['id']} bs={batch_size} sel={use_sel}")
        ud.reset()
        ordered_equal(drain(ud), case["expected"], f"{case['id']} after Reset")
        ud.close()


def test_unordered_distinct_random_vs_oracle(oracle):
    rng = np.random.default_rng(17)
    n = 40_000
    a = rng.integers(0, 300, size=n, dtype=np.int64)
    s = [bytes([97 + int(x)]) * int(y) for x, y in zip(rng.integers(0, 3, size=n), rng.integers(0, 45, size=n))]
    f = rng.choice(np.array([0.0, -0.0, 2.5, float("nan")]), size=n)
    cols = [cd.Vec.from_numpy(cd.INT64, a, rng.random(n) > 0.1), cd.Vec.from_values(cd.BYTES, s),
            cd.Vec.from_numpy(cd.FLOAT64, f, rng.random(n) > 0.1), cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64))]
    types = [v.type for v in cols]
    want, _ = oracle.unordered_distinct(cols, [0, 1, 2])
    src = ops.BatchSource(make_batches(cols, 1000, True, rng), types)
    ud = ops.UnorderedDistinct(types, [0, 1, 2], src, capacity=777, lookahead=5)
    ud.init()
    ordered_equal(drain(ud), H.vecs_to_tuples(want), "random")
    ud.close()


# ---------------------------------------------------------------- BufferingInMemoryOperator on the aggregator / DISTINCT
def _export_all(op):
    rows = []
    for _ in range(100000):
        b = op.export_buffered()
        if b.length == 0:
            return rows
        rows += H.vecs_to_tuples(b.cols, b.length)
    raise AssertionError("ExportBuffered did not terminate")


def test_hash_aggregator_export_buffered(oracle):
    """hash_aggregator.go:455-505 as the disk spiller drives it (disk_spiller.go:81-200): after some input the spiller
    calls ReleaseBeforeExport, drains ExportBuffered into the disk-backed operator, then ReleaseAfterExport.  The
    exported tuples are the whole input so far, in arrival order (selection vectors applied)."""
    import ctypes as C
    from cockroach_amd import _lib
    rng = np.random.default_rng(21)
    n = 9_000
    g = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 50, n, dtype=np.int64), rng.random(n) > 0.05)
    s = cd.Vec.from_values(cd.BYTES, [None if i % 13 == 0 else b"s%d" % i * (1 + i % 11) for i in range(n)])
    x = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n))
    cols, types = [g, s, x], [cd.INT64, cd.BYTES, cd.FLOAT64]
    aggs = [(0, 0, -1), (10, 2, -1), (14, -1, -1)]
    batches = make_batches(cols, 700, True, rng)
    ha = ops.HashAggregator(types, [0], aggs, ops.BatchSource(batches, types), capacity=512, track_input=True)
    ha.init()
    fed = 0
    for b in batches[:9]:  # part of the input: some of it already on the device, some still staged on the host
        mb, _keep = ops._marshal_batch(b)
        _lib.check(ha.lib.crgpu_agg_append(ha.h, C.byref(mb)))
        fed += b.length
    ha.release_before_export()
    got = _export_all(ha)
    assert got == H.vecs_to_tuples(cols)[:fed]
    assert ha.export_buffered().length == 0  # "all tuples have been exported"
    with pytest.raises(_lib.CrgpuError) as e:  # the device state is gone: only Reset brings the operator back
        mb, _keep = ops._marshal_batch(batches[9])
        _lib.check(ha.lib.crgpu_agg_append(ha.h, C.byref(mb)))
        _lib.check(ha.lib.crgpu_agg_finish(ha.h))
    assert e.value.code == _lib.ERR_STATE
    ha.release_after_export()
    ha.reset()
    want = H.vecs_to_tuples(oracle.hash_agg(cols, [0], aggs))
    H.assert_same_multiset(drain(ha), want, "after export + Reset")
    # the whole input is still tracked after a complete run
    assert _export_all(ha) == H.vecs_to_tuples(cols)
    ha.close()
    # without tracking the operator says so instead of handing out nothing
    ha = ops.HashAggregator(types, [0], aggs, ops.BatchSource(batches, types))
    with pytest.raises(_lib.CrgpuError) as e:
        ha.export_buffered()
    assert e.value.code == _lib.ERR_STATE
    ha.close()


def test_hash_aggregator_export_after_failed_upload(monkeypatch):
    """The OOM protocol when the UPLOAD of a staged chunk fails (the case input tracking exists for): the flush has
    moved the staged rows to the tracked input before the device call, so ExportBuffered must hand out every input row
    exactly once (round-2 advisor finding: the stage stayed filled and was appended a second time)."""
    import ctypes as C
    from cockroach_amd import _lib
    rng = np.random.default_rng(23)
    n = 5_000
    g = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 50, n, dtype=np.int64))
    x = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n))
    cols, types = [g, x], [cd.INT64, cd.FLOAT64]
    aggs = [(0, 0, -1), (10, 1, -1)]
    batches = make_batches(cols, 512, False, rng)
    ha = ops.HashAggregator(types, [0], aggs, ops.BatchSource(batches, types), capacity=512, track_input=True)
    ha.init()
    for b in batches:
        mb, _keep = ops._marshal_batch(b)
        _lib.check(ha.lib.crgpu_agg_append(ha.h, C.byref(mb)))
    monkeypatch.setenv("CRGPU_TEST_FAIL_AGG_UPLOAD", "1")
    with pytest.raises(_lib.CrgpuError) as e:
        _lib.check(ha.lib.crgpu_agg_finish(ha.h))
    assert e.value.code == _lib.ERR_OOM
    monkeypatch.delenv("CRGPU_TEST_FAIL_AGG_UPLOAD")
    ha.release_before_export()
    assert _export_all(ha) == H.vecs_to_tuples(cols)
    ha.release_after_export()
    ha.close()


def test_unordered_distinct_export_buffered(oracle):
    """unordered_distinct.go:123-134: ExportBuffered hands out the new tuples of the input read so far that Next has
    not returned yet; the table of seen tuples stays, so a later batch is still filtered against them."""
    import ctypes as C
    from cockroach_amd import _lib
    rng = np.random.default_rng(22)
    n = 6_000
    k = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 900, n, dtype=np.int64), rng.random(n) > 0.05)
    p = cd.Vec.from_values(cd.BYTES, [b"p%d" % i for i in range(n)])
    cols, types = [k, p], [cd.INT64, cd.BYTES]
    want, _ = oracle.unordered_distinct(cols, [0])
    want = H.vecs_to_tuples(want)
    batches = make_batches(cols, 500, True, rng)
    ud = ops.UnorderedDistinct(types, [0], ops.BatchSource(batches, types), capacity=100)
    ud.init()
    for b in batches[:6]:
        mb, _keep = ops._marshal_batch(b)
        _lib.check(ud.lib.crgpu_distinct_append(ud.h, C.byref(mb)))
    first = ud.out.call(ud._next_output, ud.h)  # one output batch through Next ...
    got = H.vecs_to_tuples(first.cols, first.length) + _export_all(ud)  # ... the rest through ExportBuffered
    want6, _ = oracle.unordered_distinct([v.slice(0, 3000) for v in cols], [0])
    ordered_equal(got, H.vecs_to_tuples(want6), "first six batches")
    for b in batches[6:]:
        mb, _keep = ops._marshal_batch(b)
        _lib.check(ud.lib.crgpu_distinct_append(ud.h, C.byref(mb)))
    _lib.check(ud.lib.crgpu_distinct_finish(ud.h))
    ud.state = "draining"
    ordered_equal(got + drain(ud), want, "whole input")
    ud.close()


def _late_null_batches(vecs, batch_size, first_null_row):
    """Batches whose columns carry no null bitmap at all (Nulls.MaybeHasNulls() false) until `first_null_row`, and bitmaps
    from there on: the staging area keeps per-row present bytes only from the first NULL on (host_ops.hip, stage_batch),
    so the rows staged before it must still come out present."""
    n = len(vecs[0])
    out = []
    for a in range(0, n, batch_size):
        b = min(n, a + batch_size)
        cols = []
        for v in vecs:
            s = v.slice(a, b)
            if b <= first_null_row:
                assert s.valid is None or s.present().all()
                s = cd.Vec(s.type, s.data, None, s.offsets)
            cols.append(s)
        out.append(cd.Batch(cols, b - a))
    return out


def test_operators_nulls_appear_after_null_free_batches(oracle):
    rng = np.random.default_rng(2024)
    n, first = 6000, 3500
    pres = np.ones(n, dtype=bool)
    pres[first:] = rng.random(n - first) > 0.2
    k = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 500, n, dtype=np.int64), pres)
    x = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n), np.roll(pres, 7) | (np.arange(n) < first))
    s = cd.Vec.from_values(cd.BYTES, [None if (i >= first and i % 5 == 0) else b"v%d" % (i % 50) * (1 + i % 3) for i in range(n)])
    cols, types = [k, x, s], [cd.INT64, cd.FLOAT64, cd.BYTES]
    # join: both sides arrive that way; the right side is the same table shifted
    right = [cd.Vec.from_numpy(cd.INT64, k.data[::-1].copy(), pres), cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64))]
    rt = [cd.INT64, cd.INT64]
    for jt in ("INNER", "LEFT_OUTER", "FULL_OUTER"):
        want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], cols, right, [0], [0], False))
        hj = ops.HashJoiner(H.JOIN_TYPE[jt], types, rt, [0], [0], False, ops.BatchSource(_late_null_batches(cols, 512, first), types),
                            ops.BatchSource(_late_null_batches(right, 512, first), rt), capacity=1024, lookahead=4)
        hj.init()
        H.assert_same_multiset(drain(hj), want, jt)
        hj.close()
    # aggregation
    aggs = [(0, 0, -1), (5, 1, -1), (10, 1, -1), (7, 2, -1), (14, -1, -1)]
    want = H.vecs_to_tuples(oracle.hash_agg(cols, [0], aggs))
    ha = ops.HashAggregator(types, [0], aggs, ops.BatchSource(_late_null_batches(cols, 512, first), types))
    ha.init()
    H.assert_same_multiset(drain(ha), want)
    ha.close()
    # router: rows and their NULLs reach the right outputs
    dest, counts = oracle.distribute(cols, [0], 3)
    r = ops.HashRouter(types, [0], 3, ops.BatchSource(_late_null_batches(cols, 512, first), types))
    r.run()
    all_rows = H.vecs_to_tuples(cols)
    for p, out in enumerate(r.outputs):
        H.assert_same_multiset(drain(out), [all_rows[i] for i in np.nonzero(dest == p)[0]], f"output {p}")
    r.close()
    # distinct on (k, s)
    ud = ops.UnorderedDistinct(types, [0, 2], ops.BatchSource(_late_null_batches(cols, 512, first), types))
    ud.init()
    got = drain(ud)
    seen, want = set(), []
    for row in all_rows:
        if (row[0], row[2]) not in seen:
            seen.add((row[0], row[2]))
            want.append(row)
    assert got == want
    ud.close()


@pytest.mark.parametrize("jt", ["INNER", "LEFT_OUTER"])
def test_hash_joiner_pipelined_probe_chunks(jt):
    """Five probe chunks of 2^20 rows read ahead before the output is drained: the shell uploads a chunk asynchronously,
    probes it one chunk later and copies plain results back behind an event (INNER), or takes the synchronous copies
    (LEFT OUTER: nullable build columns).  Every probe row must come out exactly once, with its own build row; a
    Reset in the middle of a pipelined probe starts over cleanly."""
    rng = np.random.default_rng(77)
    nb, npr, bs = 1 << 16, 5 * (1 << 20) + 777, 1 << 16
    bk = rng.permutation(nb * 2)[:nb].astype(np.int64)
    pk = rng.integers(0, nb * 2, size=npr, dtype=np.int64)
    pw = np.arange(npr, dtype=np.int64)
    types = [cd.INT64, cd.INT64]
    left = [cd.Vec.from_numpy(cd.INT64, pk), cd.Vec.from_numpy(cd.INT64, pw)]
    right = [cd.Vec.from_numpy(cd.INT64, bk), cd.Vec.from_numpy(cd.INT64, bk * 3 + 1)]
    lsrc = ops.BatchSource(make_batches(left, bs, False, rng), types)
    rsrc = ops.BatchSource(make_batches(right, bs, False, rng), types)
    hj = ops.HashJoiner(H.JOIN_TYPE[jt], types, types, [0], [0], True, lsrc, rsrc, capacity=bs, lookahead=64)
    hj.init()
    present = np.zeros(nb * 2, dtype=bool)
    present[bk] = True
    hit = present[pk]

    def collect(stop_after=None):
        cols = [[], [], [], []]
        valid3 = []
        rows = 0
        while True:
            b = hj.next()
            if b.length == 0:
                break
            for c in range(4):
                cols[c].append(np.array(b.cols[c].data[:b.length]))
            valid3.append(b.cols[3].present()[:b.length])
            rows += b.length
            if stop_after is not None and rows >= stop_after:
                return None
        return [np.concatenate(c) for c in cols], np.concatenate(valid3)

    assert collect(stop_after=1 << 20) is None  # leave the joiner in the middle of its pipelined probe
    hj.reset()
    (k, w, rk, rv), v3 = collect()
    hj.close()
    order = np.argsort(w, kind="stable")
    k, w, rk, rv, v3 = k[order], w[order], rk[order], rv[order], v3[order]
    if jt == "INNER":
        assert np.array_equal(w, pw[hit]) and np.array_equal(k, pk[hit])
        assert np.array_equal(rk, k) and np.array_equal(rv, k * 3 + 1) and v3.all()
    else:
        assert np.array_equal(w, pw) and np.array_equal(k, pk)
        assert np.array_equal(v3, hit)
        assert np.array_equal(rk[hit], pk[hit]) and np.array_equal(rv[hit], pk[hit] * 3 + 1)


# ---------------------------------------------------------------- CRGPU_OUT_BORROW (crgpu_out_batch.flags)
@pytest.mark.parametrize("which", ["joiner", "joiner_pipelined", "aggregator", "router", "distinct"])
def test_borrowed_output_batches_equal_copied_ones(oracle, which, monkeypatch):
    """With CRGPU_OUT_BORROW the shells hand out fixed-width values where the transfer put them instead of copying
    them into the caller's buffers (`data` is set to the place; null bitmaps and BYTES columns are written as ever):
    the same rows as without the flag.  _OutBatch.to_batch asserts that the pointer it reads is not its own buffer."""
    monkeypatch.setattr(ops, "BORROW_OUTPUT", True)
    if which == "joiner":
        test_hash_joiner_random_vs_oracle(oracle, "FULL_OUTER")
        test_hash_joiner_random_vs_oracle(oracle, "INNER")
    elif which == "joiner_pipelined":
        test_hash_joiner_pipelined_probe_chunks("INNER")
    elif which == "aggregator":
        test_hash_aggregator_random_vs_oracle(oracle)
    elif which == "router":
        test_hash_router_vs_distributor(oracle, 4)
    else:
        test_unordered_distinct_random_vs_oracle(oracle)
