"""TEST INFRASTRUCTURE ONLY (like everything under oracle/): runs the plans of workloads/tpch_plans.py with the
CPU restatement of the reference's operators -- selections produce selection vectors (colexecsel), the deselector
copies the selected tuples (colexecutils/deselector.go:48), joins and aggregations are oracle.hash_join /
oracle.hash_agg (crdb_oracle.c: hashJoiner / hashAggregator batch-at-a-time), projections are IEEE double
operations row by row.  Used by tests/ as the checker and by the benchmarks' cpu_baseline leg."""
from __future__ import annotations

import numpy as np

from cockroach_amd import coldata as cd
from oracle import oracle as O


class OracleExec:
    def __init__(self):
        self.ops = []

    def table(self, tensors):
        from workloads.tpch_synth import BytesCol
        out = {}
        for k, v in tensors.items():
            if isinstance(v, BytesCol):
                out[k] = cd.Vec(cd.BYTES, v.data.cpu().numpy(), None, v.offsets.cpu().numpy().astype(np.uint32))
            else:
                a = v.cpu().numpy()
                out[k] = cd.Vec.from_numpy(cd.FLOAT64 if a.dtype == np.float64 else cd.INT64, a)
        return out

    def select_i64(self, cmp, col, cst):
        return O.select_i64_const(cmp, col, cst)

    def select_bytes(self, cmp, col, cst):
        return O.select_bytes_const(cmp, col, cst)

    def filter(self, cols, sel):
        return O.deselect(cols, sel)

    def join(self, jt, left, right, leq, req, right_key, out, left_sel=None):
        if left_sel is not None:  # a probe batch with a selection vector: the joiner reads the selected rows only
            left = self.filter(left, left_sel)
        res = O.hash_join(jt, list(left), list(right), leq, req, right_key)
        self.ops.append(("join", len(left[0]), len(res[0]) if res else 0))
        return [res[q] for q in out]

    def project(self, op, a, b):
        return O.project_f64(op, a, b)

    def extract_year(self, col):
        return O.extract_year(col)

    def agg(self, cols, group_cols, aggs):
        return O.hash_agg(list(cols), group_cols, [(f, c, -1) for f, c in aggs])

    def to_host(self, cols):
        return list(cols)

    def close(self):
        pass
