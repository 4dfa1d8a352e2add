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


# ---------------------------------------------------------------- the fused form (csrc/gjoin.hip)
F = H.AGG_FUNC


def _fused_vs_oracle(oracle, bk, bv, pk, pv, aggs_fused, aggs_oracle):
    """aggs_oracle name the columns of the join's output (pk, pv, bk, bv); grouping by the key (column 0)."""
    from cockroach_amd import device as dev
    gj = dev.GroupJoin(aggs_fused, probe_payload_type=pv.type, build_payload_type=bv.type)
    gj.build(dev.upload(bk), dev.upload(bv))
    got = dev.result_to_vecs(gj.probe(dev.upload(pk), dev.upload(pv)))
    gj.close()
    joined = oracle.hash_join(0, [pk, pv], [bk, bv], [0], [0], True)
    want = oracle.hash_agg(joined, [0], aggs_oracle)
    H.assert_same_multiset(H.vecs_to_tuples(got), H.vecs_to_tuples(want), "fused group-join")
    return len(got[0])


@pytest.mark.parametrize("nb,npr", [(50_000, 300_000), (300_000, 2_000_000)])
def test_fused_group_join_float_payload_vs_oracle(oracle, nb, npr):
    """GROUP BY the join key, aggregates over the probe payload and the build payload, INNER: against the reference's
    composition joiner -> aggregator on the oracle.  A third of the build keys has no probe row (no group), probe keys
    miss the build side and carry NULLs, the key ~0 (the tables' empty marker) is a real key."""
    rng = np.random.default_rng(nb)
    keys = rng.permutation(nb * 3)[:nb].astype(np.int64)
    keys[0] = -1  # == the empty marker of the partitioned tables
    bk = cd.Vec.from_numpy(cd.INT64, keys)
    bv = cd.Vec.from_numpy(cd.FLOAT64, np.round(rng.random(nb) * 10, 2))
    p = rng.integers(0, nb * 2, size=npr, dtype=np.int64)
    p[rng.random(npr) < 0.001] = -1
    pk = cd.Vec.from_numpy(cd.INT64, p, rng.random(npr) > 0.02)
    pv = cd.Vec.from_numpy(cd.FLOAT64, np.round(rng.standard_normal(npr) * 100, 2))
    fused = [(F["ANY_NOT_NULL"], 0), (F["ANY_NOT_NULL"], 2), (F["SUM"], 1), (F["AVG"], 1), (F["MIN"], 1), (F["MAX"], 1),
             (F["COUNT_ROWS"], 0), (F["COUNT"], 1)]
    ora = [(F["ANY_NOT_NULL"], 0, -1), (F["ANY_NOT_NULL"], 3, -1), (F["SUM"], 1, -1), (F["AVG"], 1, -1), (F["MIN"], 1, -1),
           (F["MAX"], 1, -1), (F["COUNT_ROWS"], -1, -1), (F["COUNT"], 1, -1)]
    ng = _fused_vs_oracle(oracle, bk, bv, pk, pv, fused, ora)
    assert 0 < ng < nb


@pytest.mark.parametrize("nb,npr,empty_key_probed", [(50_000, 300_000, True), (300_000, 2_000_000, False), (200_000, 3_000, False)])
def test_fused_group_join_keeps_the_build_side_vs_oracle(oracle, nb, npr, empty_key_probed):
    """CRGPU_RIGHT_OUTER: every build row is a group, also without a probe row (COUNT_ROWS 1, COUNT over the probe side
    0, its SUM / AVG / MIN / MAX NULL) -- against the reference's composition joiner (RIGHT OUTER) -> aggregator on the
    oracle, grouped by the build key.  The build key ~0 (the tables' empty marker, kept in a side slot) with and without
    probe rows; a probe side far smaller than the build side; then the same build probed by nothing at all."""
    from cockroach_amd import device as dev
    rng = np.random.default_rng(nb + 1)
    keys = rng.permutation(nb * 3)[:nb].astype(np.int64)
    keys[0] = -1
    bk = cd.Vec.from_numpy(cd.INT64, keys)
    bv = cd.Vec.from_numpy(cd.FLOAT64, np.round(rng.random(nb) * 10, 2))
    p = rng.integers(0, nb * 2, size=npr, dtype=np.int64)
    if empty_key_probed:
        p[rng.random(npr) < 0.001] = -1
    pk = cd.Vec.from_numpy(cd.INT64, p, rng.random(npr) > 0.02)
    pv = cd.Vec.from_numpy(cd.FLOAT64, np.round(rng.standard_normal(npr) * 100, 2))
    fused = [(F["ANY_NOT_NULL"], 0), (F["ANY_NOT_NULL"], 2), (F["SUM"], 1), (F["AVG"], 1), (F["MIN"], 1), (F["MAX"], 1),
             (F["COUNT_ROWS"], 0), (F["COUNT"], 1)]
    ora = [(F["ANY_NOT_NULL"], 2, -1), (F["ANY_NOT_NULL"], 3, -1), (F["SUM"], 1, -1), (F["AVG"], 1, -1), (F["MIN"], 1, -1),
           (F["MAX"], 1, -1), (F["COUNT_ROWS"], -1, -1), (F["COUNT"], 1, -1)]
    gj = dev.GroupJoin(fused, probe_payload_type=pv.type, build_payload_type=bv.type, join_type=2)
    gj.build(dev.upload(bk), dev.upload(bv))
    got = dev.result_to_vecs(gj.probe(dev.upload(pk), dev.upload(pv)))
    joined = oracle.hash_join(2, [pk, pv], [bk, bv], [0], [0], True)  # (pk, pv, bk, bv), unmatched build rows NULL-extended
    want = oracle.hash_agg(joined, [2], ora)
    assert len(got[0]) == nb
    H.assert_same_multiset(H.vecs_to_tuples(got), H.vecs_to_tuples(want), "fused group-join, build side kept")
    # nothing probes: one row per build row, every probe-side aggregate NULL
    none_k = cd.Vec.from_numpy(cd.INT64, np.zeros(0, dtype=np.int64))
    none_v = cd.Vec.from_numpy(cd.FLOAT64, np.zeros(0))
    got = dev.result_to_vecs(gj.probe(dev.upload(none_k), dev.upload(none_v)))
    gj.close()
    want = oracle.hash_agg(oracle.hash_join(2, [none_k, none_v], [bk, bv], [0], [0], True), [2], ora)
    H.assert_same_multiset(H.vecs_to_tuples(got), H.vecs_to_tuples(want), "fused group-join, build side kept, no probe row")


def test_fused_group_join_build_side_kept_integer_payload(oracle):
    """RIGHT OUTER with an INT64 probe payload: SUM_INT / MIN NULL for build rows nobody probes, COUNT_ROWS without a
    payload at all; a nullable build key is left to the composition."""
    from cockroach_amd import _lib
    from cockroach_amd import device as dev
    rng = np.random.default_rng(10)
    nb, npr = 30_000, 200_000
    bk = cd.Vec.from_numpy(cd.INT64, np.arange(nb, dtype=np.int64) * 7 - 100)
    bv = cd.Vec.from_numpy(cd.INT64, rng.integers(-5, 5, nb, dtype=np.int64))
    pk = cd.Vec.from_numpy(cd.INT64, rng.integers(0, nb // 2, npr, dtype=np.int64) * 7 - 100)  # half of the build keys
    pv = cd.Vec.from_numpy(cd.INT64, rng.integers(-2**40, 2**40, npr, dtype=np.int64))
    fused = [(F["ANY_NOT_NULL"], 0), (F["SUM_INT"], 1), (F["MIN"], 1), (F["ANY_NOT_NULL"], 2), (F["COUNT_ROWS"], 0), (F["COUNT"], 1)]
    ora = [(F["ANY_NOT_NULL"], 2, -1), (F["SUM_INT"], 1, -1), (F["MIN"], 1, -1), (F["ANY_NOT_NULL"], 3, -1), (F["COUNT_ROWS"], -1, -1),
           (F["COUNT"], 1, -1)]
    gj = dev.GroupJoin(fused, probe_payload_type=cd.INT64, build_payload_type=cd.INT64, join_type=2)
    gj.build(dev.upload(bk), dev.upload(bv))
    got = dev.result_to_vecs(gj.probe(dev.upload(pk), dev.upload(pv)))
    gj.close()
    want = oracle.hash_agg(oracle.hash_join(2, [pk, pv], [bk, bv], [0], [0], True), [2], ora)
    H.assert_same_multiset(H.vecs_to_tuples(got), H.vecs_to_tuples(want), "fused group-join, build side kept, INT64")
    gj = dev.GroupJoin([(F["COUNT_ROWS"], 0)], join_type=2)
    gj.build(dev.upload(bk))
    got = dev.result_to_vecs(gj.probe(dev.upload(pk)))
    want = oracle.hash_agg(oracle.hash_join(2, [pk], [bk], [0], [0], True), [1], [(F["COUNT_ROWS"], -1, -1)])
    H.assert_same_multiset(H.vecs_to_tuples(got), H.vecs_to_tuples(want), "fused group-join, build side kept, COUNT_ROWS")
    gj.close()
    gj = dev.GroupJoin([(F["COUNT_ROWS"], 0)], join_type=2)
    with pytest.raises(_lib.CrgpuError) as e:
        gj.build(dev.upload(cd.Vec.from_numpy(cd.INT64, np.arange(1000, dtype=np.int64), rng.random(1000) > 0.1)))
    assert e.value.code == _lib.ERR_UNSUPPORTED
    gj.close()


def test_fused_group_join_int_payload_overflow_and_unsupported(oracle):
    from cockroach_amd import _lib
    from cockroach_amd import device as dev
    rng = np.random.default_rng(9)
    nb, npr = 20_000, 400_000
    bk = cd.Vec.from_numpy(cd.INT64, np.arange(nb, dtype=np.int64) * 7)
    bv = cd.Vec.from_numpy(cd.INT64, rng.integers(-5, 5, nb, dtype=np.int64))
    pk = cd.Vec.from_numpy(cd.INT64, rng.integers(0, nb, npr, dtype=np.int64) * 7)
    pv = cd.Vec.from_numpy(cd.INT64, rng.integers(-2**62, 2**62, npr, dtype=np.int64) // 64)
    fused = [(F["ANY_NOT_NULL"], 0), (F["SUM_INT"], 1), (F["MIN"], 1), (F["ANY_NOT_NULL"], 2), (F["COUNT_ROWS"], 0)]
    ora = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["MIN"], 1, -1), (F["ANY_NOT_NULL"], 3, -1), (F["COUNT_ROWS"], -1, -1)]
    _fused_vs_oracle(oracle, bk, bv, pk, pv, fused, ora)
    # a group whose exact total leaves int64
    big = cd.Vec.from_numpy(cd.INT64, np.full(npr, 2**62, dtype=np.int64))
    one = cd.Vec.from_numpy(cd.INT64, rng.integers(0, 100_000, npr, dtype=np.int64))  # ~4 rows of 2^62 per key
    gj = dev.GroupJoin([(F["SUM_INT"], 1)], probe_payload_type=cd.INT64)
    gj.build(dev.upload(cd.Vec.from_numpy(cd.INT64, np.arange(100_000, dtype=np.int64))))
    with pytest.raises(_lib.CrgpuError) as e:
        gj.probe(dev.upload(one), dev.upload(big))
    assert e.value.code == _lib.ERR_INT_OUT_OF_RANGE
    gj.close()
    # outside the fused form: the caller composes join -> aggregate
    for kwargs, aggs in ((dict(join_type=1), [(F["COUNT_ROWS"], 0)]),                      # LEFT OUTER
                         (dict(probe_payload_type=cd.INT64), [(F["SUM"], 1)]),            # SUM over INT => DECIMAL
                         (dict(probe_payload_type=cd.FLOAT64), [(F["ANY_NOT_NULL"], 1)]),  # arrival order
                         (dict(probe_payload_type=cd.INT64), [(F["SUM_INT"], 1), (F["MIN"], 1), (F["MAX"], 1)])):  # 5 words
        with pytest.raises(_lib.CrgpuError) as e:
            dev.GroupJoin(aggs, **kwargs)
        assert e.value.code == _lib.ERR_UNSUPPORTED
    # duplicate-heavy build side
    gj = dev.GroupJoin([(F["COUNT_ROWS"], 0)])
    with pytest.raises(_lib.CrgpuError) as e:
        gj.build(dev.upload(cd.Vec.from_numpy(cd.INT64, np.zeros(100_000, dtype=np.int64))))
    assert e.value.code == _lib.ERR_UNSUPPORTED
    gj.close()


@pytest.mark.parametrize("table", ["cuckoo", "chd"])
def test_fused_group_join_rejects_a_few_duplicate_build_keys(monkeypatch, table):
    """The fused form needs UNIQUE build keys (every group is a build row).  A handful of duplicated keys -- far from
    overflowing a sub-partition -- used to build silently (first writer wins in the tables) and count each probe row
    once where the join emits it twice; now the table kernels verify uniqueness and the build says UNSUPPORTED so the
    caller composes join -> aggregate.  Also the key equal to the tables' empty marker, held twice."""
    from cockroach_amd import _lib
    from cockroach_amd import device as dev
    if table == "chd":
        monkeypatch.setenv("CRGPU_RADIX_TABLE", "chd")
    rng = np.random.default_rng(11)
    nb = 200_000
    keys = rng.permutation(nb * 2)[:nb].astype(np.int64)
    for dup_keys in (keys[[5, 77, 1234]], np.array([-1, -1], dtype=np.int64), keys[:nb // 2]):
        k = np.concatenate([keys, dup_keys])
        gj = dev.GroupJoin([(F["COUNT_ROWS"], 0)])
        with pytest.raises(_lib.CrgpuError) as e:
            gj.build(dev.upload(cd.Vec.from_numpy(cd.INT64, k[rng.permutation(len(k))])))
        assert e.value.code == _lib.ERR_UNSUPPORTED and "duplicate" in str(e.value)
        gj.close()
    gj = dev.GroupJoin([(F["COUNT_ROWS"], 0)])  # the same keys once: builds
    gj.build(dev.upload(cd.Vec.from_numpy(cd.INT64, np.concatenate([keys, np.array([-1], dtype=np.int64)]))))
    gj.close()
