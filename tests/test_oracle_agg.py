"""Pins the oracle's hashAggregator against the reference's literal case
tables (pkg/sql/colexec/aggregators_test.go:186-785,
hash_aggregator_test.go:34-397) and the random recipe of TestAggregatorRandom
(aggregators_test.go:841-983).  DECIMAL columns run as FLOAT64 and
CONCAT_AGG aggregations are projected away (out of scope); DISTINCT aggregations run."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests import helpers as H

CASES = H.agg_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_agg_case(oracle, case):
    cols, group_cols, aggs, expected = H.prepare_agg_case(case)
    rng = np.random.default_rng(1)
    for bs, maxbuf in ((1, 4), (2, 3), (1024, 4096), (3, 4096)):
        for nb in (1, int(rng.integers(1, 8)), 256):   # TestingKnobs.HashTableNumBuckets = 1 + rand(7)
            out = oracle.hash_agg(cols, group_cols, aggs, batch_size=bs, max_buffered=maxbuf, initial_buckets=nb)
            H.assert_same_multiset(H.vecs_to_tuples(out), expected, f"{case['id']} bs={bs} nb={nb}")


def test_empty_input_gives_no_rows(oracle):
    # hash_aggregator_tmpl.go:229-239
    cols = [cd.Vec.from_values(cd.INT64, []), cd.Vec.from_values(cd.INT64, [])]
    out = oracle.hash_agg(cols, [0], [(H.AGG_FUNC["ANY_NOT_NULL"], 0, -1), (H.AGG_FUNC["SUM_INT"], 1, -1)])
    assert len(out[0]) == 0


def test_sum_int_overflow_is_an_error(oracle):
    # hash_sum_int_agg.eg.go:295-298 -> tree.ErrIntOutOfRange
    cols = [cd.Vec.from_values(cd.INT64, [1, 1]), cd.Vec.from_values(cd.INT64, [2**63 - 1, 1])]
    with pytest.raises(oracle.OracleError) as e:
        oracle.hash_agg(cols, [0], [(H.AGG_FUNC["SUM_INT"], 1, -1)])
    assert e.value.code == 1


def test_float_min_max_nan_is_smallest(oracle):
    # hash_min_max_agg.eg.go:966-981
    cols = [cd.Vec.from_values(cd.INT64, [1, 1, 1]), cd.Vec.from_values(cd.FLOAT64, [2.0, float("nan"), -5.0])]
    out = oracle.hash_agg(cols, [0], [(H.AGG_FUNC["MIN"], 1, -1), (H.AGG_FUNC["MAX"], 1, -1)])
    assert np.isnan(out[0].to_values()[0]) and out[1].to_values() == [2.0]


@pytest.mark.parametrize("group_size", [1, 2, 256, 512])
@pytest.mark.parametrize("num_batches", [1, 2, 64])
@pytest.mark.parametrize("has_nulls", [True, False])
def test_aggregator_random(oracle, group_size, num_batches, has_nulls):
    rng = np.random.default_rng(group_size * 1000 + num_batches * 10 + int(has_nulls))
    n = cd.BATCH_SIZE * num_batches
    groups = (np.arange(n) // group_size).astype(np.int64)
    vals = 2048 * (rng.random(n) - 0.5)
    present = rng.random(n) >= 0.1 if has_nulls else np.ones(n, dtype=bool)
    exp = []
    for g in range(int(groups[-1]) + 1):
        m = groups == g
        v = vals[m][present[m]]
        rows = int(m.sum())
        if len(v) == 0:
            exp.append((rows, 0, None, None, None, None))
        else:
            s = 0.0
            for x in v:
                s += x
            exp.append((rows, len(v), s, float(v.min()), float(v.max()), s / len(v)))
    cols = [cd.Vec.from_numpy(cd.INT64, groups), cd.Vec.from_numpy(cd.FLOAT64, vals, present)]
    F = H.AGG_FUNC
    aggs = [(F["COUNT_ROWS"], -1, -1), (F["COUNT"], 1, -1), (F["SUM"], 1, -1), (F["MIN"], 1, -1), (F["MAX"], 1, -1),
            (F["AVG"], 1, -1)]
    out = oracle.hash_agg(cols, [0], aggs, initial_buckets=int(rng.integers(1, 32)))
    H.assert_same_multiset(H.vecs_to_tuples(out), exp, "random")


def test_sum_over_int_is_exact(oracle):
    """SUM over INT: 16-byte little-endian totals, pinned by Python's integers (the reference's tables hold no such case:
    their SUM inputs are DECIMAL; hash_sum_agg.eg.go:30-373 adds apd decimals, which is exact too)."""
    rng = np.random.default_rng(5)
    n = 20000
    g = rng.integers(0, 50, n, dtype=np.int64)
    x = rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, n, dtype=np.int64, endpoint=True)
    valid = rng.random(n) > 0.2
    valid[g == 7] = False
    cols = [cd.Vec.from_numpy(cd.INT64, g), cd.Vec.from_numpy(cd.INT64, x, valid)]
    out = H.vecs_to_tuples(oracle.hash_agg(cols, [0], [(0, 0, -1), (10, 1, -1)]))
    want = {}
    for gi, xi, vi in zip(g.tolist(), x.tolist(), valid.tolist()):
        want.setdefault(gi, None)
        if vi:
            want[gi] = (want[gi] or 0) + xi
    got = {r[0]: (None if r[1] is None else int.from_bytes(r[1], "little", signed=True)) for r in out}
    assert got == want and got[7] is None
