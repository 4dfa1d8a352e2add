"""Pins the oracle's HashTable internals against the worked examples the
reference keeps in its comments (exact First/Next/Same/Visited/HeadID arrays),
using an injected hash function as the comments do.

* non-distinct INNER probe: pkg/sql/colexec/colexechash/hashtable_tmpl.go:584-731
* onlineAgg:                pkg/sql/colexec/hash_aggregator.go:288-352
"""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests.helpers import AGG_FUNC


@pytest.fixture()
def parity_hash(oracle):
    # "for simplicity we will assume that our hash function is h(i) = i % 2"
    oracle.set_test_hash({-1: 1, -2: 0, -3: 1, -4: 0, -5: 1})
    yield
    oracle.set_test_hash({})


def _i64(vals):
    return cd.Vec.from_numpy(cd.INT64, np.array(vals, dtype=np.int64))


RIGHT = [-1, -2, -4, -1, -3, -2, -1, -3]


def test_join_walkthrough_first_batch(oracle, parity_hash):
    out = oracle.hash_join(0, [_i64([-3, -1, -5])], [_i64(RIGHT)], [0], [0], False, batch_size=3, initial_buckets=2)
    st = oracle.last_join_state()
    assert st["first"][:2] == [2, 1]                       # hashtable_tmpl.go:601
    assert st["next"][1:] == [4, 3, 6, 5, 7, 0, 8, 0]      # :602
    assert st["same"][1:] == [7, 0, 0, 0, 8, 0, 4, 0]      # :676
    assert st["visited"] == [True, True, False, False, True, True, False, True, True]  # :677
    assert st["head_id"] == [5, 1, 0]                      # :678
    # -3 matches keyIDs 5,8; -1 matches 1,7,4; -5 none
    rows = sorted(zip(out[0].to_values(), out[1].to_values()))
    assert rows == sorted([(-3, -3)] * 2 + [(-1, -1)] * 3)


def test_join_walkthrough_second_batch(oracle, parity_hash):
    oracle.hash_join(0, [_i64([-3, -1, -5, -1, -2, -3])], [_i64(RIGHT)], [0], [0], False, batch_size=3, initial_buckets=2)
    st = oracle.last_join_state()
    assert st["same"][1:] == [7, 6, 0, 0, 8, 0, 4, 0]      # :722
    # :723 prints Visited as all-true, but keyID 3 (value -4) is never equal to
    # any probing tuple, and _CHECK_BODY (:471-563) only marks equality matches
    # as visited, so by the reference's CODE Visited[3] stays false (the comment
    # at :707 "Visited[3] = true" is a slip in the walkthrough text).
    assert st["visited"] == [True, True, True, False, True, True, True, True, True]
    assert st["head_id"] == [1, 2, 5]                      # :724


def test_online_agg_walkthrough(oracle):
    # hash_aggregator.go:288-352: batch size 4, h(i) = i % 2
    oracle.set_test_hash({-1: 1, -2: 0, -3: 1, -4: 0})
    try:
        keys = _i64([-3, -3, -2, -1, -4, -1, -1, -4])
        out = oracle.hash_agg([keys], [0], [(AGG_FUNC["ANY_NOT_NULL"], 0, -1), (AGG_FUNC["COUNT_ROWS"], -1, -1)],
                              batch_size=4, max_buffered=4, initial_buckets=2)
    finally:
        oracle.set_test_hash({})
    # buckets = [<-3>, <-2>, <-1>, <-4>] in creation order (:347-348)
    assert out[0].to_values() == [-3, -2, -1, -4]
    assert out[1].to_values() == [2, 1, 3, 2]
    st = oracle.last_agg_state()
    # all four group heads are chained from buckets 0 and 1 of BuildScratch
    assert sorted(x for x in st["first"] if x) and len(st["next"]) == 5
