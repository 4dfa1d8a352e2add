"""Distributed GROUP BY on CPU: world_size 2 over gloo.  cockroach_amd.dist_agg drives the reference's two-stage plan
(local aggregation -> BY_HASH exchange of the partial rows -> final aggregation -> AVG rendering,
physicalplan/aggregator_funcs.go:83-306) with the oracle plugged in as the executor, since there is no GPU in this
container; the exchange and the stage decomposition are the code the GPU path runs.  The union of the ranks' results
must equal the oracle's single-stage aggregation of all the rows, and every group must come out of exactly one rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AGGS = [(0, 0, -1), (0, 1, -1), (10, 2, -1), (1, 2, -1), (5, 2, -1), (14, -1, -1), (11, 3, -1), (8, 3, -1), (7, 2, -1),
        (14, -1, 4), (10, 2, 4), (1, 2, 4), (3, 4, -1), (2, 4, -1)]
TYPES_GROUP = [0, 1]


def make_shard(cd, rank, n=6000):
    rng = np.random.default_rng(500 + rank)
    k = rng.integers(0, 40, size=n, dtype=np.int64)
    s = [bytes([65 + int(x)]) * int(y) for x, y in zip(rng.integers(0, 3, size=n), rng.integers(0, 3, size=n))]
    f = np.round(rng.random(n) * 100, 3)
    i = rng.integers(-1000, 1000, size=n, dtype=np.int64)
    b = rng.integers(0, 2, size=n).astype(np.uint8)
    return [cd.Vec.from_numpy(cd.INT64, k, rng.random(n) > 0.05), cd.Vec.from_values(cd.BYTES, s),
            cd.Vec.from_numpy(cd.FLOAT64, f, rng.random(n) > 0.3), cd.Vec.from_numpy(cd.INT64, i, rng.random(n) > 0.1),
            cd.Vec.from_numpy(cd.BOOL, b, rng.random(n) > 0.2)]


class OracleExecutor:
    """CPU stand-in for GpuExecutor (test infrastructure)."""

    def __init__(self, O, cd, D):
        self.O, self.cd, self.D = O, cd, D

    def aggregate(self, types, group_cols, aggs, cols):
        vecs = [self.D.tcol_to_vec(c) for c in cols]
        return [self.D.vec_to_tcol(v) for v in self.O.hash_agg(vecs, group_cols, aggs)]

    def partition(self, types, hash_cols, world, cols):
        vecs = [self.D.tcol_to_vec(c) for c in cols]
        dest, counts = self.O.distribute(vecs, hash_cols, world)
        order = np.argsort(dest, kind="stable")
        return [self.D.vec_to_tcol(v.take(order)) for v in vecs], [int(x) for x in counts]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard(cd, rank, empty_rank):
    shard = make_shard(cd, rank)
    return [v.slice(0, 0) for v in shard] if rank == empty_rank else shard


def _worker(rank, world, port, out_dir, empty_rank=-1):
    sys.path.insert(0, ROOT)
    import pickle
    from cockroach_amd import coldata as cd, dist_agg as D
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        shard = _shard(cd, rank, empty_rank)
        types = [v.type for v in shard]
        ex = OracleExecutor(O, cd, D)
        out = D.distributed_agg(ex, types, TYPES_GROUP, AGGS, [D.vec_to_tcol(v) for v in shard])
        rows = list(zip(*[v.to_values() for v in out])) if out and len(out[0]) else []
        with open(os.path.join(out_dir, f"rows_{rank}.pkl"), "wb") as f:
            pickle.dump(rows, f)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_distributed_group_by_world2(tmp_path, oracle):
    import pickle
    from cockroach_amd import coldata as cd
    from tests import helpers as H
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    per_rank = [pickle.load(open(tmp_path / f"rows_{r}.pkl", "rb")) for r in range(world)]
    assert all(len(r) > 0 for r in per_rank), "the hash should spread 120 groups over both ranks"
    keys = [set((r[0], r[1]) for r in rows) for rows in per_rank]
    assert not (keys[0] & keys[1]), "a group was finalised on two ranks"
    shards = [make_shard(cd, r) for r in range(world)]
    whole = [cd.Vec.from_values(shards[0][c].type, sum((s[c].to_values() for s in shards), [])) for c in range(len(shards[0]))]
    want = oracle.hash_agg(whole, TYPES_GROUP, AGGS)
    H.assert_same_multiset(per_rank[0] + per_rank[1], H.vecs_to_tuples(want), "distributed vs single-stage")


def test_distributed_group_by_world2_one_rank_without_rows(tmp_path, oracle):
    """A rank with no local rows still takes part in every collective with the same column set (its partial result is
    empty and carries no validity of its own): the other rank's groups come out unchanged."""
    import pickle
    from cockroach_amd import coldata as cd
    from tests import helpers as H
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), 1), nprocs=world, join=True)
    per_rank = [pickle.load(open(tmp_path / f"rows_{r}.pkl", "rb")) for r in range(world)]
    want = oracle.hash_agg(make_shard(cd, 0), TYPES_GROUP, AGGS)
    H.assert_same_multiset(per_rank[0] + per_rank[1], H.vecs_to_tuples(want), "one empty rank")


def test_stage_plan_follows_dist_aggregation_table():
    from cockroach_amd import coldata as cd, dist_agg as D
    types = [cd.INT64, cd.FLOAT64]
    p = D.plan_stages(types, [0], [(0, 0, -1), (1, 1, -1), (10, 1, -1), (14, -1, -1), (5, 1, -1)])
    # AVG -> local SUM + COUNT (shared with the plain SUM and COUNT), final SUM + SUM_INT, rendered as a quotient
    assert p.local_aggs == [(0, 0, -1), (10, 1, -1), (5, 1, -1), (14, -1, -1)]
    assert p.inter_types == [cd.INT64, cd.FLOAT64, cd.INT64, cd.INT64]
    assert p.final_group_cols == [0]
    assert p.final_aggs == [(0, 0, -1), (10, 1, -1), (11, 2, -1), (10, 1, -1), (11, 3, -1), (11, 2, -1)]
    assert p.render == [("col", 0), ("div", 1, 2), ("col", 3), ("col", 4), ("col", 5)]
    with pytest.raises(ValueError):
        D.plan_stages(types, [0], [(4, 1, -1)])  # CONCAT_AGG has no local stage
