"""TPC-H Q3 and Q9 as left-deep chains of hash joins -> projections -> hash aggregation (BASELINE.json configs[2];
plan shapes: /root/reference/pkg/sql/opt/exec/execbuilder/testdata/tpch_vec:20574-20589 (Q3), :20709-20733 (Q9);
SURVEY.md 8(d) "Config Q3/Q9": the reference's lookup/merge joins of those plans run as hash joins here).

A plan is written ONCE against a small executor interface and run by
  * GpuExec (below): every operator is a call into libcrgpu.so (device-resident columns, no host copies between
    operators), and
  * the oracle's executor (oracle/plan_exec.py, test infrastructure): the reference operators' CPU restatement,
so the parity tests compare the same chain on both sides.

Executor interface (columns are executor-native):
  select_i64(cmp, col, const) / select_bytes(cmp, col, const) -> selection (BOOL column / row indices)
  filter(cols, sel)                                  -> cols          deselectorOp
  join(jt, left, right, leq, req, right_key, out)    -> cols          hashJoiner (+ PostProcessSpec projection)
  project(op, a, b) / extract_year(col)              -> col
  agg(cols, group_cols, aggs)                        -> cols          hashAggregator
"""
from __future__ import annotations

from cockroach_amd import coldata as cd
from workloads import tpch_synth as T

INNER = 0
ANY_NOT_NULL, SUM, COUNT_ROWS = 0, 10, 14
I64, F64, BYTES = cd.INT64, cd.FLOAT64, cd.BYTES


def q3(ex, t, build_orders_by_custkey=False):
    """SELECT l_orderkey, sum(l_extendedprice * (1 - l_discount)) AS revenue, o_orderdate, o_shippriority
       FROM customer, orders, lineitem WHERE c_mktsegment = 'BUILDING' AND c_custkey = o_custkey AND
       l_orderkey = o_orderkey AND o_orderdate < DATE '1995-03-15' AND l_shipdate > DATE '1995-03-15'
       GROUP BY l_orderkey, o_orderdate, o_shippriority   (the ORDER BY ... LIMIT 10 on top is a topKSorter: not this path).
    Returns the aggregation's columns (l_orderkey, o_orderdate, o_shippriority, revenue).

    build_orders_by_custkey: the same query with the first join's sides swapped (customers probe a table built on the
    filtered ORDERS by o_custkey: ~10 rows per build key) -- the duplicate-build-key join path on the same data."""
    c, o, l = t["customer"], t["orders"], t["lineitem"]
    cust = ex.filter([c["c_custkey"]], ex.select_bytes("EQ", c["c_mktsegment"], b"BUILDING"))
    o_cols = [o["o_orderkey"], o["o_custkey"], o["o_orderdate"], o["o_shippriority"]]
    o_sel = ex.select_i64("LT", o["o_orderdate"], T.DAY_1995_03_15)
    if not build_orders_by_custkey:
        # hashJoiner(left = orders, right = customer): tpch_vec:20584-20588; the probe batches carry the selection
        j1 = ex.join(INNER, o_cols, cust, [1], [0], True, out=[0, 2, 3], left_sel=o_sel)
    else:
        j1 = ex.join(INNER, cust, ex.filter(o_cols, o_sel), [0], [1], False, out=[1, 3, 4])
    # lineitem x (orders x customer) on l_orderkey = o_orderkey; o_orderkey is a key of the build side
    j2 = ex.join(INNER, [l["l_orderkey"], l["l_extendedprice"], l["l_discount"]], j1, [0], [0], True, out=[0, 1, 2, 4, 5],
                 left_sel=ex.select_i64("GT", l["l_shipdate"], T.DAY_1995_03_15))
    revenue = ex.project("*(1-)", j2[1], j2[2])
    return ex.agg([j2[0], j2[3], j2[4], revenue], [0, 1, 2],
                  [(ANY_NOT_NULL, 0), (ANY_NOT_NULL, 1), (ANY_NOT_NULL, 2), (SUM, 3)])


def q9(ex, t):
    """SELECT nation, o_year, sum(amount) FROM (SELECT n_name AS nation, EXTRACT(year FROM o_orderdate) AS o_year,
       l_extendedprice * (1 - l_discount) - ps_supplycost * l_quantity AS amount FROM part, supplier, lineitem,
       partsupp, orders, nation WHERE s_suppkey = l_suppkey AND ps_suppkey = l_suppkey AND ps_partkey = l_partkey AND
       p_partkey = l_partkey AND o_orderkey = l_orderkey AND s_nationkey = n_nationkey AND p_name LIKE '%green%')
       GROUP BY nation, o_year.   Returns (nation, o_year, sum_profit): 175 groups (tpch.go:318-336)."""
    p, ps, l, s, n, o = t["part"], t["partsupp"], t["lineitem"], t["supplier"], t["nation"], t["orders"]
    green = ex.filter([p["p_partkey"]], ex.select_bytes("CONTAINS", p["p_name"], b"green"))
    # partsupp x part(green): ps_partkey = p_partkey
    j1 = ex.join(INNER, [ps["ps_partkey"], ps["ps_suppkey"], ps["ps_supplycost"]], green, [0], [0], True, out=[0, 1, 2])
    # lineitem x that on (l_partkey, l_suppkey) = (ps_partkey, ps_suppkey): a two-column key, unique on the build side
    j2 = ex.join(INNER, [l["l_orderkey"], l["l_partkey"], l["l_suppkey"], l["l_quantity"], l["l_extendedprice"], l["l_discount"]],
                 j1, [1, 2], [0, 1], True, out=[0, 2, 3, 4, 5, 8])   # orderkey, suppkey, qty, price, disc, supplycost
    j3 = ex.join(INNER, j2, [s["s_suppkey"], s["s_nationkey"]], [1], [0], True, out=[0, 2, 3, 4, 5, 7])  # + s_nationkey
    j4 = ex.join(INNER, j3, [n["n_nationkey"], n["n_name"]], [5], [0], True, out=[0, 1, 2, 3, 4, 7])     # + n_name
    j5 = ex.join(INNER, j4, [o["o_orderkey"], o["o_orderdate"]], [0], [0], True, out=[1, 2, 3, 4, 5, 7])  # qty, price, disc, supplycost, n_name, o_orderdate
    o_year = ex.extract_year(j5[5])
    amount = ex.project("-", ex.project("*(1-)", j5[1], j5[2]), ex.project("*", j5[3], j5[0]))
    return ex.agg([j5[4], o_year, amount], [0, 1], [(ANY_NOT_NULL, 0), (ANY_NOT_NULL, 1), (SUM, 2)])


# ------------------------------------------------------------------------------------------------- GPU executor
class GpuExec:
    """Runs a plan through the C-ABI (cockroach_amd.device): columns stay in HBM from the base tables to the result.
    Handles are kept until close() because a result's columns are owned by the handle that produced them.
    alg_bytes: SURVEY.md 8(d) accounting -- logical bytes every operator reads plus logical bytes it writes, i.e. every
    base column once per use, every intermediate written once and read once (BYTES = payload + 4-byte offsets)."""

    def __init__(self, account=False):
        from cockroach_amd import device as dev
        self.dev = dev
        self.handles = []
        self.account = account
        self.alg_bytes = 0
        self.ops = []  # (operator, rows in, rows out)

    # -- tables
    def table(self, tensors):
        """{column: tensor | BytesCol} -> {column: DeviceColumn} (no copy; the tensors must live on the GPU)."""
        out = {}
        for k, v in tensors.items():
            if isinstance(v, T.BytesCol):
                out[k] = self.dev.from_torch(BYTES, v.data, offsets=v.offsets)
            else:
                import torch
                out[k] = self.dev.from_torch(F64 if v.dtype == torch.float64 else I64, v)
        return out

    def _bytes(self, cols):
        if not self.account:
            return 0
        import numpy as np
        total = 0
        for c in cols:
            if c.type == BYTES:
                last = int(self.dev.download(c.offsets_ptr + 4 * c.n, np.uint32, 1)[0]) if c.n else 0
                total += last + 4 * c.n
            else:
                total += c.n * {cd.BOOL: 1, cd.INT16: 2, cd.INT32: 4}.get(c.type, 8)
        return total

    def _note(self, name, ins, outs):
        if self.account:
            self.alg_bytes += self._bytes(ins) + self._bytes(outs)
            self.ops.append((name, ins[0].n if ins else 0, outs[0].n if outs else 0))

    # -- operators
    def select_i64(self, cmp, col, cst):
        r = self.dev.select_i64_const(cmp, col, cst)
        self._note("select", [col], [r])
        return r

    def select_bytes(self, cmp, col, cst):
        r = self.dev.select_bytes_const(cmp, col, cst)
        self._note("select", [col], [r])
        return r

    def filter(self, cols, sel):
        d = self.dev.Deselector([c.type for c in cols])
        self.handles.append(d)
        out = self.dev.result_columns(d.deselect(cols, sel))
        self._note("filter", list(cols) + [sel], out)
        return out

    def join(self, jt, left, right, leq, req, right_key, out, left_sel=None):
        """left_sel: a BOOL column from select_*: the probe side is the selected rows of `left` (the reference's selection
        operators only set the batch's selection vector, the joiner reads through it: crgpu_dev_join_probe_selected)."""
        j = self.dev.Join(jt, [c.type for c in left], [c.type for c in right], leq, req, right_key, out_cols=out)
        self.handles.append(j)
        j.build(right)
        res = self.dev.result_columns(j.probe(left) if left_sel is None else j.probe_selected(left, left_sel))
        if self.account:  # only the columns the join reads: keys + emitted columns
            nl = len(left)
            used_l = sorted(set(leq) | {q for q in out if q < nl})
            used_r = sorted(set(req) | {q - nl for q in out if q >= nl})
            self._note("join", [left[q] for q in used_l] + [right[q] for q in used_r] + ([left_sel] if left_sel is not None else []), res)
        return res

    def project(self, op, a, b):
        r = self.dev.project_f64(op, a, b)
        self._note("project", [x for x in (a, b) if hasattr(x, "type")], [r])
        return r

    def extract_year(self, col):
        r = self.dev.extract_year(col)
        self._note("project", [col], [r])
        return r

    def agg(self, cols, group_cols, aggs):
        a = self.dev.Agg([c.type for c in cols], group_cols, [(f, c, -1) for f, c in aggs])
        self.handles.append(a)
        a.consume(cols)
        out = self.dev.result_columns(a.finish())
        self._note("agg", cols, out)
        return out

    def to_host(self, cols):
        """Device result columns -> list[cd.Vec]."""
        import numpy as np
        out = []
        for c in cols:
            n = c.n
            valid = self.dev.download(c.valid_ptr, np.uint8, (n + 7) // 8) if c.valid_ptr else None
            if c.type == BYTES:
                offs = self.dev.download(c.offsets_ptr, np.uint32, n + 1)
                out.append(cd.Vec(BYTES, self.dev.download(c.data_ptr, np.uint8, int(offs[-1]) if n else 0), valid, offs))
            else:
                out.append(cd.Vec(c.type, self.dev.download(c.data_ptr, cd.NP_DTYPE[c.type], n), valid))
        return out

    def close(self):
        for h in self.handles:
            h.close()
        self.handles = []
