"""The two stages of the distributed GROUP BY (cockroach_amd/dist_agg.py) on the GPU through the C-ABI.  One GPU
here, so two ranks are emulated in one process: both run the local stage (crgpu_dev_agg_* then crgpu_dev_partition),
the regions are swapped by hand exactly as exchange_tcols' all-to-all would deliver them, and both run the final
stage.  The union must equal the oracle's single-stage aggregation of all rows."""
import numpy as np
import pytest
import torch

from cockroach_amd import coldata as cd
from tests import helpers as H
from tests.test_dist_agg_gloo import AGGS, TYPES_GROUP, make_shard

pytestmark = pytest.mark.gpu


def _cut(c, lo, hi):
    from cockroach_amd.dist_agg import cut_tcol
    return cut_tcol(c, lo, hi)


def _cat(parts):
    from cockroach_amd.dist_agg import cat_tcols
    return cat_tcols(parts)


@pytest.mark.parametrize("world", [2, 3])
def test_two_stage_group_by_matches_single_stage(oracle, world):
    from cockroach_amd import dist_agg as D
    ex = D.GpuExecutor(0)
    shards = [make_shard(cd, r, n=20_000) for r in range(world)]
    types = [v.type for v in shards[0]]
    plan = D.plan_stages(types, TYPES_GROUP, AGGS)
    sent = []
    for r in range(world):
        cols = [D.vec_to_tcol(v, ex.device) for v in shards[r]]
        regions, counts = D.local_stage(ex, plan, types, TYPES_GROUP, cols, world)
        assert sum(counts) == regions[0].rows
        sent.append((regions, np.concatenate([[0], np.cumsum(counts)])))
    rows, seen = [], set()
    for dst in range(world):
        received = [_cat([_cut(sent[src][0][c], int(sent[src][1][dst]), int(sent[src][1][dst + 1])) for src in range(world)])
                    for c in range(len(plan.inter_types))]
        out = D.final_stage(ex, plan, received)
        got = H.vecs_to_tuples(out)
        keys = set((g[0], g[1]) for g in got)
        assert not (keys & seen), "a group was finalised on two ranks"
        seen |= keys
        rows += got
    whole = [cd.Vec.from_values(shards[0][c].type, sum((s[c].to_values() for s in shards), [])) for c in range(len(types))]
    H.assert_same_multiset(rows, H.vecs_to_tuples(oracle.hash_agg(whole, TYPES_GROUP, AGGS)), "two-stage vs single-stage")
