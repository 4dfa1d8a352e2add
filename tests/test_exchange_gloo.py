"""The N>1 path on CPU: world_size 2 over gloo.  Each rank repartitions its
shard of both join inputs by the reference's router hash (here with the oracle
as the partitioner, since there is no GPU in this container), swaps the regions
with cockroach_amd.exchange.all_to_all_regions -- the code bench.py runs over
RCCL -- and the test checks the exchange invariants: every received key belongs
to the receiving rank, nothing is lost or duplicated, and the union of the local
joins equals the global join."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _partition_cpu(O, cd, cols, world):
    """Oracle stand-in for crgpu_dev_partition: regions in destination order + counts."""
    vecs = [cd.Vec.from_numpy(cd.INT64, c) for c in cols]
    dest, counts = O.distribute(vecs, [0], world)
    order = np.argsort(dest, kind="stable")
    return [c[order] for c in cols], [int(x) for x in counts]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from cockroach_amd import coldata as cd
    from cockroach_amd.exchange import all_to_all_regions
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        nb, npr = 3000, 9000
        bk = rng.permutation(nb).astype(np.int64) * world + rank  # globally unique build keys
        bv = bk * 3 + 1
        pk = rng.integers(0, nb * world, npr, dtype=np.int64)
        pw = np.arange(npr, dtype=np.int64) + rank * npr
        recv = []
        for cols in ([bk, bv], [pk, pw]):
            regions, counts = _partition_cpu(O, cd, cols, world)
            got, rc = all_to_all_regions([torch.from_numpy(c) for c in regions], counts)
            assert sum(rc) == got[0].numel() == got[1].numel()
            keys = cd.Vec.from_numpy(cd.INT64, got[0].numpy())
            dest, _ = O.distribute([keys], [0], world)
            assert (dest == rank).all(), "received a key that hashes to another rank"
            recv.append([g.numpy() for g in got])
        (rbk, rbv), (rpk, rpw) = recv
        L = [cd.Vec.from_numpy(cd.INT64, rpk), cd.Vec.from_numpy(cd.INT64, rpw)]
        R = [cd.Vec.from_numpy(cd.INT64, rbk), cd.Vec.from_numpy(cd.INT64, rbv)]
        res = O.hash_join(0, L, R, [0], [0], True)
        np.save(os.path.join(out_dir, f"join_{rank}.npy"), np.stack([v.data for v in res], 1))
        np.save(os.path.join(out_dir, f"in_{rank}.npy"), np.stack([pk, pw], 1))
        # mismatching counts must be rejected before any collective runs
        with pytest.raises(ValueError):
            all_to_all_regions([torch.zeros(5, dtype=torch.int64)], [1] * world)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_repartition_exchange_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    joined = np.concatenate([np.load(tmp_path / f"join_{r}.npy") for r in range(world)])
    probes = np.concatenate([np.load(tmp_path / f"in_{r}.npy") for r in range(world)])
    # every probe row hits exactly one (globally unique) build key: output (k, w, k, 3k+1)
    assert len(joined) == len(probes)
    joined = joined[np.argsort(joined[:, 1])]
    probes = probes[np.argsort(probes[:, 1])]
    assert np.array_equal(joined[:, 0], probes[:, 0]) and np.array_equal(joined[:, 1], probes[:, 1])
    assert np.array_equal(joined[:, 2], probes[:, 0]) and np.array_equal(joined[:, 3], probes[:, 0] * 3 + 1)
