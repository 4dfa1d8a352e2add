"""GPU parity: crgpu_dev_partition vs TupleHashDistributor (oracle.distribute)."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [1, 2, 4, 7, 8, 64])
def test_partition_matches_distributor(oracle, P):
    from cockroach_amd import device as dev
    rng = np.random.default_rng(P)
    n = 70_001
    k = cd.Vec.from_numpy(cd.INT64, rng.integers(-1000, 1000, size=n, dtype=np.int64), rng.random(n) > 0.05)
    s = cd.Vec.from_values(cd.BYTES, [b"s%d" % (i % 97) for i in range(n)])
    w = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n))
    cols = [k, s, w]
    dest, counts = oracle.distribute(cols, [0, 1], P)
    p = dev.Partitioner([c.type for c in cols], [0, 1], P)
    res, got_counts = p.partition([dev.upload(c) for c in cols], n)
    assert got_counts.tolist() == counts.tolist()
    out = dev.result_to_vecs(res)
    rows = H.vecs_to_tuples(out)
    src = H.vecs_to_tuples(cols)
    start = 0
    for q in range(P):
        want = [src[i] for i in np.nonzero(dest == q)[0]]
        H.assert_same_multiset(rows[start:start + int(counts[q])], want, f"output {q}")
        start += int(counts[q])
    p.close()


def test_partition_known_answer():
    # pkg/sql/colflow/routers_test.go:698-781
    from cockroach_amd import device as dev
    v = cd.Vec.from_numpy(cd.INT64, np.arange(1024, dtype=np.int64))
    p = dev.Partitioner([cd.INT64], [0], 4)
    res, counts = p.partition([dev.upload(v)], 1024)
    assert counts.tolist() == [273, 252, 287, 212]
    assert sorted(dev.result_to_vecs(res)[0].data.tolist()) == list(range(1024))
    p.close()


def test_partition_empty():
    from cockroach_amd import device as dev
    v = cd.Vec.from_numpy(cd.INT64, np.zeros(0, dtype=np.int64))
    p = dev.Partitioner([cd.INT64], [0], 4)
    res, counts = p.partition([dev.upload(v)], 0)
    assert res.n == 0 and counts.tolist() == [0, 0, 0, 0]
    p.close()


@pytest.mark.parametrize("P", [3, 16, 1000, 1024])
@pytest.mark.parametrize("lds", ["1", "0"])
def test_partition_fixed_width_hash_columns(oracle, P, lds, monkeypatch):
    """Fixed-width hash columns only: the scatter recomputes the destination instead of reading the histogram pass's
    codes (partition.hip, part_scatter_lds_kernel); every column width, a nullable column, a BYTES passenger, a last tile
    that is not full, up to the 1024 outputs the partitioner takes.  lds = 0: the per-row scatter of round 1."""
    from cockroach_amd import device as dev
    monkeypatch.setenv("CRGPU_PARTITION_LDS", lds)
    rng = np.random.default_rng(P)
    n = 3 * 4096 + 1234
    cols = [cd.Vec.from_numpy(cd.INT64, rng.integers(-10**12, 10**12, size=n, dtype=np.int64)),
            cd.Vec.from_numpy(cd.INT32, rng.integers(-5, 5, size=n).astype(np.int32), rng.random(n) > 0.1),
            cd.Vec.from_numpy(cd.INT16, rng.integers(-300, 300, size=n).astype(np.int16)),
            cd.Vec.from_numpy(cd.BOOL, rng.random(n) > 0.5),
            cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n), rng.random(n) > 0.3),
            cd.Vec.from_values(cd.BYTES, [b"v%d" % i for i in range(n)])]
    dest, counts = oracle.distribute(cols, [0, 1, 2], P)
    p = dev.Partitioner([c.type for c in cols], [0, 1, 2], P)
    for _ in range(2):  # the second call reuses the handle's buffers
        res, got_counts = p.partition([dev.upload(c) for c in cols], n)
        assert got_counts.tolist() == counts.tolist()
        rows = H.vecs_to_tuples(dev.result_to_vecs(res))
    src = H.vecs_to_tuples(cols)
    order = np.argsort(dest, kind="stable")
    start = 0
    for q in range(P):
        k = int(counts[q])
        assert sorted(rows[start:start + k], key=repr) == sorted((src[i] for i in order[start:start + k]), key=repr), q
        start += k
    p.close()
