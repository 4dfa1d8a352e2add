"""TPC-H Q9 distributed over N ranks (BASELINE.json configs[3]: "TPC-H SF100 Q9 distributed hash join with HashRouter
repartition over RCCL all-to-all, 2/4/8 MI355X").

The plan is workloads/tpch_plans.q9 -- /root/reference/pkg/sql/opt/exec/execbuilder/testdata/tpch_vec:20709-20733 --
cut into stages the way the reference's physical planner distributes a join (pkg/sql/physicalplan/physical_plan.go:
1086-1157, AddJoinStage): every rank holds a shard of every table; BOTH inputs of a big join are routed BY_HASH on the
equality columns (crgpu_dev_partition with the reference's router hash + crgpu_exchange_all_to_all) and every rank joins
its bucket; SMALL build sides -- the filtered part, supplier, nation -- are mirrored to every rank instead
(crgpu_exchange_broadcast = OutputRouterSpec_MIRROR, execinfrapb/data.proto:153-154), so the probe side
of those joins does not move; the final GROUP BY is the two-stage plan of distsql_physical_planner.go:2790-2844 with the
decomposition of physicalplan/aggregator_funcs.go:83-306 (cockroach_amd.dist_agg.plan_stages).

A plan is a GENERATOR over the executor interface of workloads/tpch_plans.py: where rows have to cross ranks it yields
an exchange request -- ("broadcast", cols) or ("hash", cols, hash_cols) -- and is sent the exchanged columns.  The
driver decides what an exchange is:
  * run(gen, RcclFabric)           one process per GPU, the C-ABI exchange over RCCL (bench.py --gpus N, a Go host);
  * run(gen, GlooFabric)           world_size-2 CPU tests: the oracle as executor, torch.distributed (gloo) as the wire;
  * run_lockstep(gens, LocalFabric) several emulated ranks in ONE process (one GPU): all ranks advance to their next
                                   exchange, the regions are swapped by hand.
Every rank returns its share of the result groups; the union over the ranks is the single-node result (175 groups,
pkg/workload/tpch/tpch.go:318-336).
"""
from __future__ import annotations

from cockroach_amd import coldata as cd
from workloads import tpch_synth as T

INNER, LEFT_SEMI = 0, 4
ANY_NOT_NULL, SUM = 0, 10


# ------------------------------------------------------------------------------------------------- table shards
def shard_tables(t, rank, world):
    """Rank `rank`'s contiguous row range of every table (tensors / BytesCols; views, nothing is copied except the
    re-based offsets of BYTES columns)."""
    import torch
    out = {}
    for name, cols in t.items():
        n = None
        for v in cols.values():
            n = len(v) if isinstance(v, T.BytesCol) else v.numel()
            break
        lo, hi = n * rank // world, n * (rank + 1) // world
        sh = {}
        for k, v in cols.items():
            if isinstance(v, T.BytesCol):
                offs = v.offsets[lo:hi + 1].to(torch.int64)  # (uint32 tensors do not subtract in torch)
                b0, b1 = int(offs[0].item()), int(offs[-1].item())
                sh[k] = T.BytesCol(v.data[b0:b1], (offs - b0).to(v.offsets.dtype))
            else:
                sh[k] = v[lo:hi]
        out[name] = sh
    return out


# ------------------------------------------------------------------------------------------------- the plan
def q9_dist(ex, t, semi_reduce=True):
    """This rank's part of Q9 over its table shards `t` (executor-native columns, ex.table()).  Generator: see the
    module docstring.  Returns (nation, o_year, sum_profit) for the groups that hash to this rank.

    semi_reduce: before lineitem is hash-routed for the (l_partkey, l_suppkey) join, keep only the rows whose part is
    in the mirrored `green` set (a LEFT SEMI join on l_partkey): the join's build side holds green parts only, so the
    dropped rows could not match -- same result, ~5 % of the rows cross xGMI instead of all of them."""
    from cockroach_amd import dist_agg as D
    p, ps, l, s, n, o = t["part"], t["partsupp"], t["lineitem"], t["supplier"], t["nation"], t["orders"]
    green = yield ("broadcast", ex.filter([p["p_partkey"]], ex.select_bytes("CONTAINS", p["p_name"], b"green")))
    supp = yield ("broadcast", [s["s_suppkey"], s["s_nationkey"]])
    nation = yield ("broadcast", [n["n_nationkey"], n["n_name"]])
    # partsupp shard x green (mirrored): local; then BY_HASH on (ps_partkey, ps_suppkey)
    j1 = ex.join(INNER, [ps["ps_partkey"], ps["ps_suppkey"], ps["ps_supplycost"]], green, [0], [0], True, out=[0, 1, 2])
    j1 = yield ("hash", j1, [0, 1])
    lcols = [l["l_orderkey"], l["l_partkey"], l["l_suppkey"], l["l_quantity"], l["l_extendedprice"], l["l_discount"]]
    if semi_reduce:
        lcols = ex.join(LEFT_SEMI, lcols, green, [1], [0], True, out=[0, 1, 2, 3, 4, 5])
    lcols = yield ("hash", lcols, [1, 2])
    # this rank's bucket of lineitem x (partsupp x green) on the two-column key
    j2 = ex.join(INNER, lcols, j1, [1, 2], [0, 1], True, out=[0, 2, 3, 4, 5, 8])  # orderkey, suppkey, qty, price, disc, supplycost
    j3 = ex.join(INNER, j2, supp, [1], [0], True, out=[0, 2, 3, 4, 5, 7])          # + s_nationkey   (mirrored build side)
    j4 = ex.join(INNER, j3, nation, [5], [0], True, out=[0, 1, 2, 3, 4, 7])        # + n_name        (mirrored build side)
    # ... x orders on l_orderkey = o_orderkey: both sides BY_HASH
    j4 = yield ("hash", j4, [0])
    orders = yield ("hash", [o["o_orderkey"], o["o_orderdate"]], [0])
    j5 = ex.join(INNER, j4, orders, [0], [0], True, out=[1, 2, 3, 4, 5, 7])  # qty, price, disc, supplycost, n_name, o_orderdate
    o_year = ex.extract_year(j5[5])
    amount = ex.project("-", ex.project("*(1-)", j5[1], j5[2]), ex.project("*", j5[3], j5[0]))
    # GROUP BY nation, o_year: local aggregation -> BY_HASH of the partial rows on the group columns -> final aggregation
    types = [cd.BYTES, cd.FLOAT64, cd.FLOAT64]
    plan = D.plan_stages(types, [0, 1], [(ANY_NOT_NULL, 0, -1), (ANY_NOT_NULL, 1, -1), (SUM, 2, -1)])
    partial = ex.agg([j5[4], o_year, amount], [0, 1], [(f, c) for f, c, _ in plan.local_aggs])
    partial = yield ("hash", partial, plan.final_group_cols)
    fin = ex.agg(partial, plan.final_group_cols, [(f, c) for f, c, _ in plan.final_aggs])
    return [fin[r[1]] for r in plan.render]  # no AVG here: every rendering is a plain column


def q9_scanned_tables():
    return ["part", "partsupp", "lineitem", "supplier", "nation", "orders"]


# ------------------------------------------------------------------------------------------------- drivers
def run(gen, fabric):
    """Drive one rank's plan: every yielded exchange request is a collective on `fabric`."""
    try:
        req = next(gen)
        while True:
            req = gen.send(fabric.exchange(req))
    except StopIteration as e:
        return e.value


def run_lockstep(gens, fabric):
    """Drive several emulated ranks in one process: all of them advance to their next exchange request, `fabric`
    (a LocalFabric) swaps the regions by hand, and so on until every plan has returned."""
    reqs, results = [], [None] * len(gens)
    for g in gens:
        reqs.append(next(g))
    while True:
        resp = fabric.exchange_all(reqs)
        reqs, done = [], 0
        for r, g in enumerate(gens):
            try:
                reqs.append(g.send(resp[r]))
            except StopIteration as e:
                results[r] = e.value
                done += 1
        if done == len(gens):
            return results
        if done:
            raise RuntimeError("the ranks' plans went out of step (one returned, another asked for an exchange)")
        kinds = {q[0] for q in reqs}
        if len(kinds) != 1:
            raise RuntimeError(f"the ranks' plans went out of step: {sorted(kinds)}")


# ------------------------------------------------------------------------------------------------- fabrics
class RcclFabric:
    """The exchange behind the C-ABI: crgpu_dev_partition + crgpu_exchange_all_to_all / crgpu_exchange_broadcast on
    RCCL.  Every exchange runs on its own clone of the exchange handle (same communicator): its result stays valid while
    later exchanges run -- a join's build side is borrowed until the join is destroyed."""

    def __init__(self, xch):
        from cockroach_amd import device as dev
        self.dev, self.xch, self.handles = dev, xch, []
        self.bytes_sent = 0
        self.rows_sent = 0

    def _clone(self):
        x = self.xch.clone()
        self.handles.append(x)
        return x

    def exchange(self, req):
        dev = self.dev
        kind, cols = req[0], list(req[1])
        n = cols[0].n if cols else 0
        if kind == "broadcast":
            res, _rc = self._clone().broadcast(cols, n)
            return dev.result_columns(res)
        if kind != "hash":
            raise ValueError(f"unknown exchange request {kind!r}")
        part = dev.Partitioner([c.type for c in cols], list(req[2]), self.xch.world)
        try:
            regions, counts = part.partition(cols, n)
            got, _rc = self._clone().all_to_all(dev.result_columns(regions), counts)
            self.rows_sent += int(counts.sum()) - int(counts[self.xch.rank])
        finally:
            part.close()  # its regions have been sent (the exchange call waits for its stream)
        return dev.result_columns(got)

    def close(self):
        for h in self.handles:
            h.close()
        self.handles = []


class _TColFabric:
    """Common part of the fabrics that move rows as dist_agg.TCols: `adapter` converts executor-native columns to TCols
    and back, `pex` regroups TCols by destination (a dist_agg-style executor: GpuExecutor = crgpu_dev_partition, or the
    oracle stand-in of the CPU tests)."""

    def __init__(self, world, adapter, pex):
        self.world, self.adapter, self.pex = world, adapter, pex

    def _regions(self, req):
        """(TCol regions in destination order, counts per destination) of one rank's request."""
        from cockroach_amd import dist_agg as D
        kind = req[0]
        tcols = [D._explicit_validity(c) for c in self.adapter.to_tcols(list(req[1]))]
        if kind == "broadcast":
            n = tcols[0].rows if tcols else 0
            return [D.cat_tcols([c] * self.world) for c in tcols], [n] * self.world
        if kind != "hash":
            raise ValueError(f"unknown exchange request {kind!r}")
        regions, counts = self.pex.partition([c.type for c in tcols], list(req[2]), self.world, tcols)
        return [D._explicit_validity(c) for c in regions], counts


class GlooFabric(_TColFabric):
    """One process per rank, torch.distributed as the wire (gloo on CPU): the N > 1 bookkeeping without a GPU."""

    def __init__(self, world, adapter, pex, group=None):
        super().__init__(world, adapter, pex)
        self.group = group

    def exchange(self, req):
        from cockroach_amd import dist_agg as D
        regions, counts = self._regions(req)
        got, _rc = D.exchange_tcols(regions, counts, self.group)
        return self.adapter.from_tcols(got)


class LocalFabric(_TColFabric):
    """Emulated ranks in one process: rank r's region for rank d is cut out of r's regrouped columns and the pieces
    for d are concatenated in source-rank order -- what the all-to-all delivers."""

    def exchange_all(self, reqs):
        from cockroach_amd import dist_agg as D
        sent = []
        for req in reqs:
            regions, counts = self._regions(req)
            bounds = [0]
            for c in counts:
                bounds.append(bounds[-1] + int(c))
            sent.append((regions, bounds))
        out = []
        for dst in range(self.world):
            ncols = len(sent[0][0])
            got = [D.cat_tcols([D.cut_tcol(sent[src][0][c], sent[src][1][dst], sent[src][1][dst + 1]) for src in range(self.world)])
                   for c in range(ncols)]
            out.append(self.adapter.from_tcols(got))
        return out


class GpuAdapter:
    """GpuExec columns (device.DeviceColumn) <-> TCols on the GPU."""

    def __init__(self, device_index=0):
        from cockroach_amd import dist_agg as D
        self.gx = D.GpuExecutor(device_index)

    def to_tcols(self, cols):
        return self.gx.cols_to_tcols(cols)

    def from_tcols(self, tcols):
        return [self.gx._to_dev(c) for c in tcols]


class OracleAdapter:
    """OracleExec columns (cd.Vec) <-> TCols on the CPU (test infrastructure drives it; no oracle import here)."""

    def to_tcols(self, cols):
        from cockroach_amd import dist_agg as D
        return [D.vec_to_tcol(v) for v in cols]

    def from_tcols(self, tcols):
        from cockroach_amd import dist_agg as D
        return [D.tcol_to_vec(c) for c in tcols]
