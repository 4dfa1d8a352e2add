#!/usr/bin/env python3
"""Hash aggregation throughput (device API, inputs resident in HBM): Q1 shape (two 1-byte BYTES group columns,
4 groups, 4 SUM + 3 AVG + COUNT_ROWS over float64), mid cardinality (int64 key, SUM_INT + COUNT_ROWS + MIN + MAX)
and all-distinct keys (config G shape).  Prints one JSON line per case."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cockroach_amd import _lib, coldata as cd, device as dev  # noqa: E402


CPU_SAMPLE = 1 << 22  # rows the single-threaded oracle port aggregates for the cpu_baseline figure


def to_host_vec(c, n):
    """First n rows of a device column built by dev.from_torch, as a host Vec (CPU baseline input)."""
    data, valid, offsets = c.keep
    if c.type == cd.BYTES:
        offs = offsets[:n + 1].cpu().numpy().astype(np.uint32)
        return cd.Vec(cd.BYTES, data[:int(offs[-1])].cpu().numpy(), None, offs)
    return cd.Vec.from_numpy(c.type, data[:n].cpu().numpy())


def cpu_baseline(cols, group_cols, aggs, n):
    from oracle import oracle as O  # test infrastructure: only the reported CPU baseline uses it
    m = min(n, CPU_SAMPLE)
    host = [to_host_vec(c, m) for c in cols]
    t0 = time.perf_counter()
    out = O.hash_agg(host, group_cols, aggs)
    dt = time.perf_counter() - t0
    return {"value": m / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"first {m} rows of the same input (oracle/crdb_oracle.c hashAggregator restatement, 1 thread)",
            "groups_in_sample": len(out[0])}


def run(name, cols, types, group_cols, aggs, n, alg_bytes, reps=5, expect=None):
    torch.cuda.synchronize()  # inputs are produced on torch's stream, the library runs on its own
    a = dev.Agg(types, group_cols, aggs)
    a.consume(cols, n); res = a.finish(); ngroups = int(res.n); a.reset()
    torch.cuda.synchronize()
    def passes():
        for _ in range(reps):
            a.consume(cols, n)
            a.finish()
            a.reset()
        _lib.load().crgpu_dev_synchronize()
    # wall time with the library's profiling hooks off; the per-family kernel times from `reps` more passes with them on
    dev.prof_enable(False)
    t0 = time.perf_counter()
    passes()
    dt = (time.perf_counter() - t0) / reps
    dev.prof_enable(True); dev.prof_reset()
    passes()
    dev.prof_enable(False)
    if expect is not None and expect != ngroups:
        raise SystemExit(f"{name}: {ngroups} groups, expected {expect}")
    kern = {}
    for k in ("agg_fused_lds", "agg_fused_sumfam", "agg_fused_hbm", "agg_resolve", "agg_fixup", "agg_accum_lds", "agg_accum_hbm", "agg_reindex",
              "agg_any_fixup", "agg_part_hist", "agg_part_scatter", "agg_part_accum", "agg_bulk", "agg_bulk_distinct", "agg_bulk_part", "radix_build_hist",
              "radix_build_scatter", "radix_build_split"):
        ms, cnt = dev.prof_get(k)
        if cnt:
            kern[k] = {"ms_per_pass": round(ms / reps, 3), "launches_per_pass": cnt / reps}
    a.close()
    return {"case": name, "rows": n, "groups": ngroups, "ms": dt * 1e3, "rows_per_s": n / dt,
            "alg_GBps": alg_bytes / dt / 1e9, "frac_of_8TBps": alg_bytes / dt / 8e12, "kernels": kern,
            "cpu_baseline": cpu_baseline(cols, group_cols, aggs, n) if CPU_SAMPLE else None}


def run_q1_full(n, dv, g, rf, ls, offs, f, reps):
    """Config Q1 as SURVEY.md 8(d) words it: WHERE l_shipdate <= cutoff (~98.6 % pass), disc_price = price * (1 - disc),
    charge = disc_price * (1 + tax), GROUP BY (returnflag, linestatus): SUM(qty), SUM(price), SUM(disc_price),
    SUM(charge), AVG(qty), AVG(price), AVG(disc), COUNT_ROWS.  42 B/row of input.  The selection is a BOOL column used
    as every aggregate's FILTER (no row moves); two projection kernels (each two reference operators in one pass) write dense columns."""
    ship = torch.randint(8036, 10562, (n,), generator=g, device=dv, dtype=torch.int64)  # 1992-01-02 .. 1998-12-01 in days
    cutoff = 10562 - 36  # passes ~98.6 %
    qty, price, disc, tax = f
    base = [dev.from_torch(cd.INT64, ship), dev.from_torch(cd.BYTES, rf, offsets=offs), dev.from_torch(cd.BYTES, ls, offsets=offs)] + \
           [dev.from_torch(cd.FLOAT64, x) for x in (qty, price, disc, tax)]
    types = [cd.INT64, cd.BYTES, cd.BYTES] + [cd.FLOAT64] * 6 + [cd.BOOL]
    W = 9
    aggs = [(0, 1, -1), (0, 2, -1), (10, 3, W), (10, 4, W), (10, 7, W), (10, 8, W), (1, 3, W), (1, 4, W), (1, 5, W), (14, -1, W)]
    torch.cuda.synchronize()
    a = dev.Agg(types, [1, 2], aggs)

    bufs = [dev.DeviceBuffer(n), dev.DeviceBuffer(n * 8), dev.DeviceBuffer(n * 8)]  # the operator's output vectors, reused per batch

    def once():
        where = dev.select_i64_const("LE", base[0], cutoff, out=bufs[0])
        disc_price = dev.project_f64("*(1-)", base[4], base[5], out=bufs[1])   # price * (1 - disc)
        charge = dev.project_f64("*(1+)", disc_price, base[6], out=bufs[2])    # disc_price * (1 + tax)
        a.consume(base + [disc_price, charge, where], n)
        res = a.finish()
        ng = int(res.n)
        a.reset()
        return ng

    ngroups = once()
    _lib.load().crgpu_dev_synchronize()
    dev.prof_enable(False)  # wall time without the profiling hooks, then `reps` more passes with them for the kernel times
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    _lib.load().crgpu_dev_synchronize()
    dt = (time.perf_counter() - t0) / reps
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(reps):
        once()
    _lib.load().crgpu_dev_synchronize()
    dev.prof_enable(False)
    kern = {}
    for k in ("expr_select", "expr_project", "agg_fused_lds", "agg_fused_sumfam", "agg_resolve", "agg_fixup", "agg_accum_lds"):
        ms, cnt = dev.prof_get(k)
        if cnt:
            kern[k] = {"ms_per_pass": round(ms / reps, 3), "launches_per_pass": cnt / reps}
    a.close()
    cpu = None
    if CPU_SAMPLE:
        from oracle import oracle as O
        m = min(n, CPU_SAMPLE)
        host = [to_host_vec(c, m) for c in base]
        t1 = time.perf_counter()
        keep = O.select_i64_const("LE", host[0], cutoff)
        sel = [v.take(keep) for v in host]
        dp = O.project_f64("*", sel[4], O.project_f64("-", 1.0, sel[5]))
        ch = O.project_f64("*", dp, O.project_f64("+", 1.0, sel[6]))
        O.hash_agg(sel + [dp, ch], [1, 2], [(x[0], x[1], -1) for x in aggs])
        cdt = time.perf_counter() - t1
        cpu = {"value": m / cdt, "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"first {m} rows: numpy selection + projections, then oracle/crdb_oracle.c hashAggregator (1 thread)"}
    return {"case": "Q1 full: WHERE shipdate <= c, 2 projected columns, GROUP BY 2 bytes keys, 4 SUM + 3 AVG + COUNT_ROWS", "rows": n,
            "groups": ngroups, "ms": dt * 1e3, "rows_per_s": n / dt, "alg_GBps": n * 42.0 / dt / 1e9, "frac_of_8TBps": n * 42.0 / dt / 8e12,
            "kernels": kern, "cpu_baseline": cpu}


def run_q1_fused(n, dv, g, rf, ls, offs, f, reps):
    """Config Q1 with the selection and the projections evaluated INSIDE the aggregation kernels
    (crgpu_dev_agg_create_fused): the same plan and result as run_q1_full, no intermediate column exists in HBM."""
    ship = torch.randint(8036, 10562, (n,), generator=g, device=dv, dtype=torch.int64)
    cutoff = 10562 - 36
    qty, price, disc, tax = f
    base = [dev.from_torch(cd.INT64, ship), dev.from_torch(cd.BYTES, rf, offsets=offs), dev.from_torch(cd.BYTES, ls, offsets=offs)] + \
           [dev.from_torch(cd.FLOAT64, x) for x in (qty, price, disc, tax)]
    types = [cd.INT64, cd.BYTES, cd.BYTES] + [cd.FLOAT64] * 4
    # computed: 7 disc_price = price * (1 - disc), 8 charge = disc_price * (1 + tax), 9 WHERE ship <= cutoff
    vcols = [("project", "*(1-)", 4, 5), ("project", "*(1+)", 7, 6), ("select", "LE", 0, cutoff)]
    aggs = [(0, 1, -1), (0, 2, -1), (10, 3, -1), (10, 4, -1), (10, 7, -1), (10, 8, -1), (1, 3, -1), (1, 4, -1), (1, 5, -1), (14, -1, -1)]
    torch.cuda.synchronize()
    a = dev.FusedAgg(types, [1, 2], aggs, vcols, where=9)

    def once():
        a.consume(base, n)
        res = a.finish()
        ng = int(res.n)
        a.reset()
        return ng

    ngroups = once()
    _lib.load().crgpu_dev_synchronize()
    dev.prof_enable(False)  # wall time without the profiling hooks, then `reps` more passes with them for the kernel times
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    _lib.load().crgpu_dev_synchronize()
    dt = (time.perf_counter() - t0) / reps
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(reps):
        once()
    _lib.load().crgpu_dev_synchronize()
    dev.prof_enable(False)
    kern = {}
    for k in ("agg_fused_lds", "agg_fused_sumfam", "agg_resolve", "agg_fixup", "agg_accum_lds"):
        ms, cnt = dev.prof_get(k)
        if cnt:
            kern[k] = {"ms_per_pass": round(ms / reps, 3), "launches_per_pass": cnt / reps}
    a.close()
    cpu = None
    if CPU_SAMPLE:  # the same plan on the host: selection -> projections -> hashAggregator, one thread
        from oracle import oracle as O
        m = min(n, CPU_SAMPLE)
        host = [to_host_vec(c, m) for c in base]
        t1 = time.perf_counter()
        keep = O.select_i64_const("LE", host[0], cutoff)
        sel = [v.take(keep) for v in host]
        dp = O.project_f64("*", sel[4], O.project_f64("-", 1.0, sel[5]))
        ch = O.project_f64("*", dp, O.project_f64("+", 1.0, sel[6]))
        O.hash_agg(sel + [dp, ch], [1, 2], [(x[0], x[1], -1) for x in aggs])
        cdt = time.perf_counter() - t1
        cpu = {"value": m / cdt, "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"first {m} rows: numpy selection + projections, then oracle/crdb_oracle.c hashAggregator (1 thread), {cdt:.1f} s"}
    return {"case": "Q1 full, fused: WHERE shipdate <= c and the 2 projections evaluated inside the aggregation kernel, GROUP BY 2 bytes keys, 4 SUM + 3 AVG + COUNT_ROWS",
            "rows": n, "groups": ngroups, "ms": dt * 1e3, "rows_per_s": n / dt, "alg_GBps": n * 42.0 / dt / 1e9,
            "frac_of_8TBps": n * 42.0 / dt / 8e12, "kernels": kern, "cpu_baseline": cpu}


def measure(lg=27, only=None, reps=5, cpu_sample=None, device_index=0):
    """Yields one result dict per case (bench.py reports a few of them beside the join line)."""
    global CPU_SAMPLE
    if cpu_sample is not None:
        CPU_SAMPLE = cpu_sample
    n = 1 << lg
    _lib.check(_lib.load().crgpu_init(device_index))
    dv = torch.device("cuda", device_index)
    g = torch.Generator(device=dv); g.manual_seed(1)
    # Q1 shape
    rf = torch.randint(0, 3, (n,), generator=g, device=dv, dtype=torch.uint8) + 65
    ls = torch.randint(0, 2, (n,), generator=g, device=dv, dtype=torch.uint8) + 70
    offs = torch.arange(n + 1, device=dv, dtype=torch.int32)
    f = [torch.rand(n, generator=g, device=dv, dtype=torch.float64) for _ in range(4)]
    cols = [dev.from_torch(cd.BYTES, rf, offsets=offs), dev.from_torch(cd.BYTES, ls, offsets=offs)] + \
           [dev.from_torch(cd.FLOAT64, x) for x in f]
    types = [cd.BYTES, cd.BYTES] + [cd.FLOAT64] * 4
    aggs = [(0, 0, -1), (0, 1, -1), (10, 2, -1), (10, 3, -1), (10, 4, -1), (10, 5, -1), (1, 2, -1), (1, 3, -1), (1, 4, -1), (14, -1, -1)]
    if only is None or "q1" in only:
        yield run("Q1 shape: 2 bytes keys, <=6 groups, 4 SUM + 3 AVG + COUNT_ROWS", cols, types, [0, 1], aggs, n, n * 34.0, reps=reps)
    if only is None or "q1full" in only:
        yield run_q1_full(n, dv, g, rf, ls, offs, f, reps)
    if only is None or "q1fused" in only:
        yield run_q1_fused(n, dv, g, rf, ls, offs, f, reps)
    del cols, f, rf, ls
    for groups in (1 << 10, 1 << 13, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 21, 1 << 22, 1 << 24, n):
        if groups > n or (only is not None and str(groups) not in only):
            continue
        k = torch.randint(0, groups, (n,), generator=g, device=dv, dtype=torch.int64) if groups < n else \
            torch.randperm(n, generator=g, device=dv, dtype=torch.int64)
        x = torch.randint(-1000, 1000, (n,), generator=g, device=dv, dtype=torch.int64)
        cols = [dev.from_torch(cd.INT64, k), dev.from_torch(cd.INT64, x)]
        aggs = [(0, 0, -1), (11, 1, -1), (14, -1, -1), (8, 1, -1), (7, 1, -1)]
        yield run(f"int64 key, {groups} groups, SUM_INT + COUNT_ROWS + MIN + MAX", cols, [cd.INT64, cd.INT64], [0], aggs, n,
                  n * 16.0 + groups * 40.0, expect=int(torch.unique(k).numel()), reps=reps)
        del cols, k, x


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
    only = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else None  # e.g. "q1,1024,65536"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cpu_sample = int(sys.argv[4]) if len(sys.argv) > 4 else None  # 0: skip the CPU baseline (profiling runs)
    for r in measure(lg, only, reps, cpu_sample):
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
