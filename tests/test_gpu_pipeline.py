"""A device-resident operator chain through the C-ABI, against the same chain run on the oracle: hash group-join
(colexec/hash_group_joiner.go:20-76: hash joiner -> hash aggregator), the probe's result columns consumed in place."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("jt", [0, 1])  # INNER, LEFT OUTER (unmatched rows aggregate NULL build columns)
def test_hash_group_join_vs_oracle(oracle, jt):
    from cockroach_amd import device as dev
    rng = np.random.default_rng(40 + jt)
    nb, npr = 3000, 50_000
    bk = cd.Vec.from_numpy(cd.INT64, rng.permutation(nb * 2)[:nb].astype(np.int64))
    bv = cd.Vec.from_numpy(cd.FLOAT64, np.round(rng.random(nb) * 10, 2))
    pk = cd.Vec.from_numpy(cd.INT64, rng.integers(0, nb * 2, size=npr, dtype=np.int64), rng.random(npr) > 0.02)
    pg = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 40, size=npr, dtype=np.int64))
    px = cd.Vec.from_numpy(cd.FLOAT64, np.round(rng.random(npr) * 100, 2))
    left, right = [pk, pg, px], [bk, bv]
    lt, rt = [v.type for v in left], [v.type for v in right]
    aggs = [(0, 1, -1), (10, 2, -1), (10, 4, -1), (5, 4, -1), (14, -1, -1)]  # over the join's output (pk, pg, px, bk, bv)
    j = dev.Join(jt, lt, rt, [0], [0], True)
    a = dev.Agg(lt + rt, [1], aggs)
    j.build([dev.upload(v) for v in right], nb)
    half = npr // 2
    for lo, hi in ((0, half), (half, npr)):
        res = j.probe([dev.upload(v.slice(lo, hi)) for v in left], hi - lo)
        a.consume(dev.result_columns(res), int(res.n))  # before the next call on the join handle
    got = dev.result_to_vecs(a.finish())
    j.close(); a.close()
    joined = oracle.hash_join(jt, left, right, [0], [0], True)
    want = oracle.hash_agg(joined, [1], aggs)
    H.assert_same_multiset(H.vecs_to_tuples(got), H.vecs_to_tuples(want), "group-join")
