"""GPU parity of the Grace-partitioned (external) hash join and hash aggregation (csrc/grace.hip, SURVEY.md 8(f) N4;
colexecdisk/hash_based_partitioner.go, external_hash_joiner.go, external_hash_aggregator.go): with a budget far below
the input, the union of the partitions' outputs must be the in-memory operator's output (the oracle's hash joiner /
aggregator on the whole input), for every join type, with forced recursive repartitioning, with keys that do not
shrink under repartitioning (fallback), and with the disk tier.  The reference's own tests run the same case tables
through the external operators with tiny memory limits (external_hash_joiner_test.go:47-90)."""
import os

import numpy as np
import pytest

from cockroach_amd import _lib
from cockroach_amd import coldata as cd
from cockroach_amd import device as dev
from tests import helpers as H

pytestmark = pytest.mark.gpu
F = H.AGG_FUNC


def run_grace(g, inputs, batch=20_000):
    for idx, cols in inputs:
        n = len(cols[0])
        for lo in range(0, max(n, 1), batch):
            hi = min(n, lo + batch)
            if hi > lo:
                g.add(idx, [dev.upload(v.slice(lo, hi)) for v in cols], hi - lo)
    rows = []
    for piece in g.results():
        rows += H.vecs_to_tuples(piece)
    return rows


def join_inputs(rng, nl, nr, nkeys, dup_heavy=False):
    lk = rng.integers(0, nkeys, nl, dtype=np.int64)
    rk = rng.integers(0, nkeys, nr, dtype=np.int64)
    if dup_heavy:
        rk[: nr // 2] = 7  # half of the build side shares one key: repartitioning cannot split it
        lk[:20] = 7  # (20 probe rows x nr / 2 build rows of output)
    left = [cd.Vec.from_numpy(cd.INT64, lk, rng.random(nl) > 0.03),
            cd.Vec.from_values(cd.BYTES, [b"L%d" % i * (1 + i % 5) for i in range(nl)])]
    right = [cd.Vec.from_numpy(cd.INT32, rk.astype(np.int32), rng.random(nr) > 0.03),  # INT32 vs INT64 keys hash alike
             cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(nr)),
             cd.Vec.from_values(cd.BYTES, [None if i % 9 == 0 else b"r%d" % i for i in range(nr)])]
    return left, right


@pytest.mark.parametrize("jt", ["INNER", "LEFT_OUTER", "RIGHT_OUTER", "FULL_OUTER", "LEFT_SEMI", "LEFT_ANTI", "RIGHT_SEMI",
                                "RIGHT_ANTI", "INTERSECT_ALL", "EXCEPT_ALL"])
def test_grace_join_all_types_vs_oracle(oracle, jt):
    rng = np.random.default_rng(H.JOIN_TYPE[jt] + 100)
    if jt in ("INTERSECT_ALL", "EXCEPT_ALL"):  # set operations compare whole rows of one schema
        a = [cd.Vec.from_numpy(cd.INT64, rng.integers(0, 3000, 40_000, dtype=np.int64), rng.random(40_000) > 0.05)]
        b = [cd.Vec.from_numpy(cd.INT64, rng.integers(0, 3000, 25_000, dtype=np.int64), rng.random(25_000) > 0.05)]
        left, right, lt, rt = a, b, [cd.INT64], [cd.INT64]
    else:
        left, right = join_inputs(rng, 60_000, 30_000, 20_000)
        lt, rt = [v.type for v in left], [v.type for v in right]
    want = H.vecs_to_tuples(oracle.hash_join(H.JOIN_TYPE[jt], left, right, [0], [0], False))
    g = dev.Grace.join(H.JOIN_TYPE[jt], lt, rt, [0], [0], mem_limit=128 << 10, num_buckets=8)
    got = run_grace(g, [(0, left), (1, right)])
    st = g.stats()
    assert st.partitions_processed >= 8 and st.bytes_spilled > 0
    g.close()
    H.assert_same_multiset(got, want, jt)


def test_grace_join_recursive_repartition_and_fallback(oracle):
    """A 64 KiB budget against a 1.3 MB build side whose rows share one key for a half: the first level's partitions do
    not fit, their children are redistributed with the next seed, and the partition holding the hot key stops
    shrinking (< 5 %) and runs through the fallback."""
    rng = np.random.default_rng(5)
    left, right = join_inputs(rng, 50_000, 60_000, 40_000, dup_heavy=True)
    lt, rt = [v.type for v in left], [v.type for v in right]
    want = H.vecs_to_tuples(oracle.hash_join(0, left, right, [0], [0], False))
    g = dev.Grace.join(0, lt, rt, [0], [0], mem_limit=64 << 10, num_buckets=4)
    got = run_grace(g, [(1, right), (0, left)])  # inputs may arrive in any order
    st = g.stats()
    assert st.repartitions >= 1 and st.fallback_partitions >= 1
    g.close()
    H.assert_same_multiset(got, want, "recursive + fallback")


def test_grace_join_disk_tier(oracle, tmp_path):
    rng = np.random.default_rng(6)
    left, right = join_inputs(rng, 40_000, 20_000, 15_000)
    lt, rt = [v.type for v in left], [v.type for v in right]
    want = H.vecs_to_tuples(oracle.hash_join(1, left, right, [0], [0], False))
    g = dev.Grace.join(1, lt, rt, [0], [0], mem_limit=256 << 10, num_buckets=8, host_limit=100 << 10, spill_dir=str(tmp_path))
    for idx, cols in ((0, left), (1, right)):
        g.add(idx, [dev.upload(v) for v in cols], len(cols[0]))
    assert len(os.listdir(tmp_path)) > 0, "chunks beyond host_limit live in files"
    rows = []
    for piece in g.results():
        rows += H.vecs_to_tuples(piece)
    st = g.stats()
    assert st.bytes_spilled_disk > 0
    g.close()
    assert os.listdir(tmp_path) == [], "spill files are removed once read"
    H.assert_same_multiset(rows, want, "LEFT OUTER through the disk tier")
    # a handle destroyed before its partitions were read removes its files too
    g = dev.Grace.join(1, lt, rt, [0], [0], mem_limit=256 << 10, host_limit=1, spill_dir=str(tmp_path))
    g.add(1, [dev.upload(v) for v in right], len(right[0]))
    assert len(os.listdir(tmp_path)) > 0
    g.close()
    assert os.listdir(tmp_path) == []


@pytest.mark.parametrize("ngroups", [5, 30_000])
def test_grace_agg_vs_oracle(oracle, ngroups):
    """External hash aggregation: groups of different partitions are disjoint, so the partitions' results concatenate."""
    rng = np.random.default_rng(ngroups)
    n = 150_000
    k1 = cd.Vec.from_numpy(cd.INT64, rng.integers(0, ngroups, n, dtype=np.int64), rng.random(n) > 0.02)
    k2 = cd.Vec.from_values(cd.BYTES, [b"ab"[i % 2:i % 2 + 1] * (1 + i % 3) for i in range(n)])
    x = cd.Vec.from_numpy(cd.INT64, rng.integers(-1000, 1000, n, dtype=np.int64), rng.random(n) > 0.1)
    f = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n))
    cols = [k1, k2, x, f]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["ANY_NOT_NULL"], 1, -1), (F["SUM_INT"], 2, -1), (F["COUNT_ROWS"], -1, -1),
            (F["MIN"], 3, -1), (F["MAX"], 1, -1), (F["AVG"], 3, -1)]
    want = H.vecs_to_tuples(oracle.hash_agg(cols, [0, 1], aggs))
    g = dev.Grace.agg([v.type for v in cols], [0, 1], aggs, mem_limit=256 << 10, num_buckets=8)
    got = run_grace(g, [(0, cols)])
    st = g.stats()
    assert st.partitions_processed >= (8 if ngroups > 100 else 1)
    g.close()
    H.assert_same_multiset(got, want, f"{ngroups} groups")


def test_grace_state_errors():
    g = dev.Grace.join(0, [cd.INT64], [cd.INT64], [0], [0])
    assert list(g.results()) == []  # no input at all
    with pytest.raises(_lib.CrgpuError) as e:
        g.add(0, [dev.upload(cd.Vec.from_values(cd.INT64, [1, 2]))], 2)
    assert e.value.code == _lib.ERR_STATE
    with pytest.raises(_lib.CrgpuError) as e:
        dev.Grace.join(0, [cd.INT64], [cd.INT64], [0], [0], num_buckets=1)
    assert e.value.code == _lib.ERR_INVALID
    g.close()
