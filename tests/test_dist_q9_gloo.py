"""Distributed TPC-H Q9 (BASELINE.json configs[3]) on CPU: world_size 2 over gloo.  workloads/tpch_dist.q9_dist -- the
plan bench.py --gpus N runs over RCCL -- with the oracle as the executor and torch.distributed (gloo) as the wire,
since there is no GPU in this container: mirrored small build sides, BY_HASH routing of both inputs of the big joins
(physical_plan.go:1086-1157), two-stage GROUP BY (aggregator_funcs.go:83-306).  The union of the ranks' results must be
the oracle's single-node Q9 -- 175 groups (tpch.go:318-336) -- and every group must come out of exactly one rank."""
import os
import pickle
import socket
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SF, SEED = 0.02, 42


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, semi_reduce):
    sys.path.insert(0, ROOT)
    from cockroach_amd import coldata as cd, dist_agg as D
    from oracle import oracle as O
    from oracle.plan_exec import OracleExec
    from tests.test_dist_agg_gloo import OracleExecutor
    from workloads import tpch_dist as Q, tpch_synth as T
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        t, _rows = T.generate(SF, "cpu", SEED)
        ex = OracleExec()
        shard = {k: ex.table(v) for k, v in Q.shard_tables(t, rank, world).items()}
        fabric = Q.GlooFabric(world, Q.OracleAdapter(), OracleExecutor(O, cd, D))
        res = Q.run(Q.q9_dist(ex, shard, semi_reduce), fabric)
        rows = list(zip(*[v.to_values() for v in res])) if res and len(res[0]) else []
        with open(os.path.join(out_dir, f"q9_{rank}.pkl"), "wb") as f:
            pickle.dump(rows, f)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("semi_reduce", [True, False])
def test_distributed_q9_world2(tmp_path, oracle, semi_reduce):
    from oracle.plan_exec import OracleExec
    from workloads import tpch_plans as P, tpch_synth as T
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), semi_reduce), nprocs=world, join=True)
    per_rank = [pickle.load(open(tmp_path / f"q9_{r}.pkl", "rb")) for r in range(world)]
    assert all(len(r) > 0 for r in per_rank), "175 groups should spread over both ranks"
    keys = [set(r[:2] for r in rows) for rows in per_rank]
    assert not (keys[0] & keys[1]), "a group was finalised on two ranks"
    t, _rows = T.generate(SF, "cpu", SEED)
    oex = OracleExec()
    want = sorted(zip(*[v.to_values() for v in P.q9(oex, {k: oex.table(v) for k, v in t.items()})]))
    got = sorted(per_rank[0] + per_rank[1])
    assert len(got) == len(want) == 175
    for g, w in zip(got, want):
        assert g[:2] == w[:2] and abs(g[2] - w[2]) <= 1e-9 * abs(w[2]) + 1e-6


def test_shard_tables_partition_every_table():
    import torch
    from workloads import tpch_dist as Q, tpch_synth as T
    t, rows = T.generate(0.01, "cpu", 3)
    for world in (1, 2, 3, 8):
        shards = [Q.shard_tables(t, r, world) for r in range(world)]
        for name, cols in t.items():
            for k, v in cols.items():
                if isinstance(v, T.BytesCol):
                    data = torch.cat([s[name][k].data for s in shards])
                    lens = torch.cat([s[name][k].offsets.to(torch.int64).diff() for s in shards])
                    assert torch.equal(data, v.data) and torch.equal(lens, v.offsets.to(torch.int64).diff())
                    assert all(int(s[name][k].offsets[0]) == 0 for s in shards)
                else:
                    assert torch.equal(torch.cat([s[name][k] for s in shards]), v)
