"""Distributed TPC-H Q9 (BASELINE.json configs[3]; workloads/tpch_dist.py) through the C-ABI on the GPU.  One GPU on
the test box, so:
  * 2 and 3 EMULATED ranks in one process: every rank runs its part of the plan on its table shards (crgpu_dev_* for
    every operator, crgpu_dev_partition for the BY_HASH routing), the regions are swapped by hand as the all-to-all
    would deliver them, the mirrored build sides are concatenated in rank order;
  * ONE rank through RCCL (crgpu_exchange_broadcast / crgpu_exchange_all_to_all on clones of one communicator, every
    region sent and received with ncclSend / ncclRecv: CRGPU_EXCHANGE_SELF=rccl).
The union of the ranks' results must be the oracle's single-node Q9: 175 groups (tpch.go:318-336), keys bit-exact,
SUM(float) within rel 1e-9 / abs 1e-6.  World size 2 over gloo with the oracle as executor: tests/test_dist_q9_gloo.py."""
import numpy as np
import pytest
import torch

from cockroach_amd import coldata as cd
from tests import helpers as H
from workloads import tpch_dist as Q
from workloads import tpch_plans as P
from workloads import tpch_synth as T

pytestmark = pytest.mark.gpu


def _to_gpu(t):
    return {name: {k: (T.BytesCol(v.data.cuda(), v.offsets.cuda()) if isinstance(v, T.BytesCol) else v.cuda()) for k, v in cols.items()}
            for name, cols in t.items()}


def _want(t):
    from oracle.plan_exec import OracleExec
    oex = OracleExec()
    return sorted(zip(*[v.to_values() for v in P.q9(oex, {k: oex.table(v) for k, v in t.items()})]))


def _check(per_rank, want):
    keys = [set(r[:2] for r in rows) for rows in per_rank]
    for a in range(len(keys)):
        for b in range(a + 1, len(keys)):
            assert not (keys[a] & keys[b]), "a group was finalised on two ranks"
    got = sorted(sum(per_rank, []))
    assert len(got) == len(want) == 175
    for g, w in zip(got, want):
        assert g[:2] == w[:2] and abs(g[2] - w[2]) <= 1e-9 * abs(w[2]) + 1e-6


@pytest.mark.parametrize("world,semi_reduce", [(2, True), (3, True), (2, False)])
def test_q9_emulated_ranks_vs_oracle(oracle, world, semi_reduce):
    from cockroach_amd import dist_agg as D
    t, _rows = T.generate(0.02, "cpu", 42)
    want = _want(t)
    tg = _to_gpu(t)
    exs = [P.GpuExec() for _ in range(world)]
    gens = [Q.q9_dist(exs[r], {k: exs[r].table(v) for k, v in Q.shard_tables(tg, r, world).items()}, semi_reduce)
            for r in range(world)]
    adapter = Q.GpuAdapter(0)
    fabric = Q.LocalFabric(world, adapter, adapter.gx)
    res = Q.run_lockstep(gens, fabric)
    per_rank = [list(zip(*[v.to_values() for v in exs[r].to_host(res[r])])) if res[r][0].n else [] for r in range(world)]
    for ex in exs:
        ex.close()
    assert sum(1 for r in per_rank if r) == world, "175 groups should spread over all ranks"
    _check(per_rank, want)


def test_q9_one_rank_through_rccl_vs_oracle(oracle, monkeypatch):
    from cockroach_amd import device as dev
    monkeypatch.setenv("CRGPU_EXCHANGE_SELF", "rccl")  # with one rank every region is a self-region: drive RCCL anyway
    t, _rows = T.generate(0.03, "cpu", 7)
    want = _want(t)
    tg = _to_gpu(t)
    xch = dev.Exchange(dev.Exchange.unique_id(), 1, 0)
    fabric = Q.RcclFabric(xch)
    for semi_reduce in (True, False):
        ex = P.GpuExec()
        res = Q.run(Q.q9_dist(ex, {k: ex.table(v) for k, v in tg.items()}, semi_reduce), fabric)
        rows = list(zip(*[v.to_values() for v in ex.to_host(res)]))
        ex.close()
        _check([rows], want)
    fabric.close()
    xch.close()


@pytest.mark.parametrize("self_mode", ["copy", "rccl"])
def test_broadcast_and_clones_all_families(oracle, monkeypatch, self_mode):
    """crgpu_exchange_broadcast with one rank returns the rank's own rows (all families, NULLs, empty input); results of
    two clones of one communicator stay valid side by side."""
    from cockroach_amd import device as dev
    if self_mode == "rccl":
        monkeypatch.setenv("CRGPU_EXCHANGE_SELF", "rccl")
    rng = np.random.default_rng(5)
    xch = dev.Exchange(dev.Exchange.unique_id(), 1, 0)
    a, b = xch.clone(), xch.clone()
    held = []
    for x, n in ((xch, 0), (a, 1), (b, 40_001)):
        k = cd.Vec.from_numpy(cd.INT64, rng.integers(-1000, 1000, size=n, dtype=np.int64), rng.random(n) > 0.1)
        s = cd.Vec.from_values(cd.BYTES, [None if i % 17 == 3 else b"v%d" % i * (i % 4) for i in range(n)])
        f = cd.Vec.from_numpy(cd.FLOAT64, rng.standard_normal(n))
        bo = cd.Vec.from_numpy(cd.BOOL, rng.random(n) > 0.5)
        cols = [k, s, f, bo]
        got, rc = x.broadcast([dev.upload(c) for c in cols], n)
        assert rc.tolist() == [n] and int(got.n) == n
        held.append((got, cols, n))
    for got, cols, n in held:  # every clone still holds its own result
        assert H.vecs_to_tuples(dev.result_to_vecs(got)) == H.vecs_to_tuples(cols), f"n={n}"
    for x in (a, b, xch):
        x.close()
