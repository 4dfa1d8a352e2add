"""The exchange behind the C-ABI (crgpu_exchange_*, csrc/exchange.hip) on real RCCL.  One GPU on the test box, so the
communicator has world size 1: every region is sent to and received from this rank through ncclSend / ncclRecv
(the same grouped calls N ranks issue), the count all-gather runs, BYTES and validity travel in their wire forms.
Covered here: the exchange itself, the device-resident two-stage GROUP BY (dist_agg.distributed_agg_device) against
the oracle, and bench.py's N > 1 code path (--force-distributed: repartition -> exchange -> local join, then the
two-stage GROUP BY leg), result-checked by bench.py itself.  The N = 2 bookkeeping is covered on CPU by
tests/test_exchange_gloo.py and tests/test_dist_agg_gloo.py."""
import numpy as np
import pytest

from cockroach_amd import coldata as cd
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["rccl", "copy"])
def self_regions(request, monkeypatch):
    """With one rank every region is a self-region: "rccl" sends them through ncclSend / ncclRecv like any other region
    (CRGPU_EXCHANGE_SELF=rccl), "copy" is the default device-to-device copy a rank uses for its own region."""
    if request.param == "rccl":
        monkeypatch.setenv("CRGPU_EXCHANGE_SELF", "rccl")
    else:
        monkeypatch.delenv("CRGPU_EXCHANGE_SELF", raising=False)


@pytest.fixture(scope="module")
def xch():
    from cockroach_amd import device as dev
    x = dev.Exchange(dev.Exchange.unique_id(), 1, 0)
    yield x
    x.close()


def test_self_exchange_all_families(xch, oracle):
    from cockroach_amd import device as dev
    rng = np.random.default_rng(3)
    for n in (0, 1, 50_001):
        k = cd.Vec.from_numpy(cd.INT64, rng.integers(-1000, 1000, size=n, dtype=np.int64), rng.random(n) > 0.1)
        s = cd.Vec.from_values(cd.BYTES, [None if i % 17 == 3 else b"v%d" % i * (i % 4) for i in range(n)])
        f = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n))
        b = cd.Vec.from_numpy(cd.BOOL, rng.random(n) > 0.5)
        cols = [k, s, f, b]
        p = dev.Partitioner([c.type for c in cols], [0, 1], 1)
        regions, counts = p.partition([dev.upload(c) for c in cols], n)
        assert counts.tolist() == [n]
        got, rc = xch.all_to_all(dev.result_columns(regions), counts)
        assert rc.tolist() == [n] and int(got.n) == n
        H.assert_same_multiset(H.vecs_to_tuples(dev.result_to_vecs(got)), H.vecs_to_tuples(cols), f"n={n}")
        p.close()


def test_two_stage_group_by_device_resident(xch, oracle):
    from cockroach_amd import device as dev
    from cockroach_amd import dist_agg as D
    from tests.test_dist_agg_gloo import AGGS, TYPES_GROUP, make_shard
    shard = make_shard(cd, 0, n=30_000)
    types = [v.type for v in shard]
    out = D.distributed_agg_device(xch, types, TYPES_GROUP, AGGS, [dev.upload(v) for v in shard], len(shard[0]))
    want = oracle.hash_agg(shard, TYPES_GROUP, AGGS)
    H.assert_same_multiset(H.vecs_to_tuples(out), H.vecs_to_tuples(want), "two-stage (RCCL self-exchange) vs single-stage oracle")


def test_bench_distributed_path_with_one_rank(capsys):
    """bench.py's N > 1 step -- crgpu_dev_partition of both join inputs, crgpu_exchange_all_to_all, local join -- and its
    two-stage GROUP BY leg, with one rank; bench.py checks both results itself."""
    import json
    import bench
    bench.main(["--force-distributed", "--log2-build", "18", "--log2-probe", "20", "--steps", "2", "--warmup", "1",
                "--no-cpu-baseline", "--no-other-paths", "--log2-groupby", "18"])
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["config"]["result_check"] == "ok"
    assert line["group_by"]["result_check"] == "ok" and line["group_by"]["groups"] == (1 << 18) // 4
    assert "exchange" in line["kernels_ms_per_step"] and "partition" in line["kernels_ms_per_step"]


@pytest.mark.parametrize("chunks", [2, 5])
def test_bench_distributed_probe_side_in_pieces(capsys, chunks):
    """The pipelined form of the N > 1 step (the default with more than one rank): the probe side is partitioned and
    exchanged in pieces by a second thread while the previous piece is probed; bench.py's extra pass checks every piece's
    rows, sums and routing.  Pieces of unequal size (2^20 rows in 5) included."""
    import json
    import bench
    bench.main(["--force-distributed", "--log2-build", "18", "--log2-probe", "20", "--steps", "2", "--warmup", "1",
                "--no-cpu-baseline", "--no-other-paths", "--log2-groupby", "16", "--no-q9", "--probe-chunks", str(chunks)])
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["config"]["result_check"] == "ok"


def test_self_exchange_two_gib_columns(xch):
    """2^28 rows of two int64 columns: 2 GiB per column and destination.  A single ncclSend / ncclRecv of that size
    delivered wrong data on this RCCL; the exchange moves a region in pieces of 256 MiB."""
    import torch
    from cockroach_amd import device as dev
    import bench
    n = 1 << 28
    a = torch.arange(n, device="cuda", dtype=torch.int64) * 3 + 1
    b = torch.arange(n, device="cuda", dtype=torch.int64)
    torch.cuda.synchronize()
    got, rc = xch.all_to_all([dev.from_torch(cd.INT64, a), dev.from_torch(cd.INT64, b)], [n])
    assert rc.tolist() == [n]
    d = torch.device("cuda", 0)
    assert bool((bench._wrap(torch, got.cols[0].data, n, d) == a).all()) and bool((bench._wrap(torch, got.cols[1].data, n, d) == b).all())
