#!/usr/bin/env python3
"""BASELINE.json north_star sizes on ONE MI355X: "TPC-H SF100 Q1 and Q9-style hash join + aggregation ... on 1 GPU".

* Q1 in full on 6.0e8 lineitem rows (SF100), inputs resident in HBM: WHERE l_shipdate <= c, disc_price and charge
  evaluated inside the aggregation kernels (crgpu_dev_agg_create_fused), GROUP BY (l_returnflag, l_linestatus),
  4 SUM + 3 AVG + COUNT_ROWS; 42 B/row of input (SURVEY.md 8(d) "Config Q1").  Checked against torch reductions over
  the same columns (counts exact, float sums rel 1e-9).
* Q9 at SF100 (600 M lineitems, 150 M orders, 80 M partsupp, 20 M parts; the synthetic tables of
  workloads/tpch_synth.py, ~45 GB): the plan of workloads/tpch_plans.q9 through the C-ABI, checked against the
  independent dense-lookup evaluation of tests/tpch_check.py (175 groups, pkg/workload/tpch/tpch.go:318-336).
* BASELINE.json configs[0], "TPC-H SF1 Q1 ... CPU only": the same Q1 on 6.0e6 rows through the CPU port of the
  reference operators (numpy selection + projections, oracle/crdb_oracle.c hashAggregator, one thread) -- the
  cpu_baseline of the Q1 line; the Q9 line's is the oracle's executor on the same plan at SF1.
Prints one JSON line per case."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def q1_columns(torch, n, dv, seed=1):
    g = torch.Generator(device=dv)
    g.manual_seed(seed)
    rf = torch.randint(0, 3, (n,), generator=g, device=dv, dtype=torch.uint8) + 65
    ls = torch.randint(0, 2, (n,), generator=g, device=dv, dtype=torch.uint8) + 70
    offs = torch.arange(n + 1, device=dv, dtype=torch.int32)
    qty = torch.randint(1, 51, (n,), generator=g, device=dv, dtype=torch.int32).to(torch.float64)
    price = torch.rand(n, generator=g, device=dv, dtype=torch.float64) * 1e5
    disc = torch.randint(0, 11, (n,), generator=g, device=dv, dtype=torch.int32).to(torch.float64) / 100.0
    tax = torch.randint(0, 9, (n,), generator=g, device=dv, dtype=torch.int32).to(torch.float64) / 100.0
    ship = torch.randint(8036, 10562, (n,), generator=g, device=dv, dtype=torch.int64)  # 1992-01-02 .. 1998-12-01 in days
    return ship, rf, ls, offs, qty, price, disc, tax


Q1_CUTOFF = 10562 - 36  # passes ~98.6 % (1998-12-01 minus 90 days, scaled to the synthetic date range)
Q1_AGGS = [(0, 1, -1), (0, 2, -1), (10, 3, -1), (10, 4, -1), (10, 7, -1), (10, 8, -1), (1, 3, -1), (1, 4, -1), (1, 5, -1), (14, -1, -1)]


def q1_cpu_sf1(n=6_000_000):
    """configs[0]: Q1 at SF1 on the host, one thread: selection -> projections -> hashAggregator (reference operators restated)."""
    import numpy as np
    import torch
    from cockroach_amd import coldata as cd
    from oracle import oracle as O
    ship, rf, ls, offs, qty, price, disc, tax = q1_columns(torch, n, torch.device("cpu"))
    o32 = offs.numpy().astype(np.uint32)
    host = [cd.Vec.from_numpy(cd.INT64, ship.numpy()), cd.Vec(cd.BYTES, rf.numpy(), None, o32), cd.Vec(cd.BYTES, ls.numpy(), None, o32)] + \
           [cd.Vec.from_numpy(cd.FLOAT64, x.numpy()) for x in (qty, price, disc, tax)]
    t0 = time.perf_counter()
    keep = O.select_i64_const("LE", host[0], Q1_CUTOFF)
    sel = [v.take(keep) for v in host]
    dp = O.project_f64("*", sel[4], O.project_f64("-", 1.0, sel[5]))
    ch = O.project_f64("*", dp, O.project_f64("+", 1.0, sel[6]))
    out = O.hash_agg(sel + [dp, ch], [1, 2], [(x[0], x[1], -1) for x in Q1_AGGS])
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "rows/s", "cores": 1, "kind": "port", "seconds": dt, "groups": len(out[0]),
            "sample": f"BASELINE configs[0]: TPC-H SF1 Q1 ({n} lineitem rows) on the CPU, one thread: numpy selection + "
                      f"projections, oracle/crdb_oracle.c hashAggregator, {dt:.1f} s"}


def q1_sf100(device_index=0, n=600_000_000, reps=3, cpu=True):
    import torch
    from cockroach_amd import _lib, coldata as cd, device as dev
    dv = torch.device("cuda", device_index)
    ship, rf, ls, offs, qty, price, disc, tax = q1_columns(torch, n, dv)
    base = [dev.from_torch(cd.INT64, ship), dev.from_torch(cd.BYTES, rf, offsets=offs), dev.from_torch(cd.BYTES, ls, offsets=offs)] + \
           [dev.from_torch(cd.FLOAT64, x) for x in (qty, price, disc, tax)]
    types = [cd.INT64, cd.BYTES, cd.BYTES] + [cd.FLOAT64] * 4
    vcols = [("project", "*(1-)", 4, 5), ("project", "*(1+)", 7, 6), ("select", "LE", 0, Q1_CUTOFF)]
    torch.cuda.synchronize()
    a = dev.FusedAgg(types, [1, 2], Q1_AGGS, vcols, where=9)
    a.consume(base, n)
    got = a.finish_vecs()
    a.reset()
    _lib.load().crgpu_dev_synchronize()
    def passes():
        for _ in range(reps):
            a.consume(base, n)
            a.finish()
            a.reset()
        _lib.load().crgpu_dev_synchronize()
    dev.prof_enable(False)  # wall time without the library's profiling hooks, then the same passes with them for the kernel times
    t0 = time.perf_counter()
    passes()
    dt = (time.perf_counter() - t0) / reps
    dev.prof_enable(True)
    dev.prof_reset()
    passes()
    dev.prof_enable(False)
    kern = {}
    for k in ("agg_fused_lds", "agg_fused_sumfam", "agg_resolve", "agg_fixup", "agg_accum_lds"):
        ms, cnt = dev.prof_get(k)
        if cnt:
            kern[k] = {"ms_per_pass": round(ms / reps, 3), "launches_per_pass": cnt / reps}
    a.close()
    # check: the same groups from torch reductions, chunk by chunk (exact counts; float sums rel 1e-9)
    want = {}
    step = 1 << 27
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        w = ship[lo:hi] <= Q1_CUTOFF
        code = (rf[lo:hi].to(torch.int64) - 65) * 2 + (ls[lo:hi].to(torch.int64) - 70)
        dp = price[lo:hi] * (1.0 - disc[lo:hi])
        ch = dp * (1.0 + tax[lo:hi])
        for gcode in range(6):
            m = w & (code == gcode)
            acc = want.setdefault(gcode, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
            acc[0] += int(m.sum().item())
            for i, col in enumerate((qty[lo:hi], price[lo:hi], dp, ch, disc[lo:hi])):
                acc[1 + i] += float(col[m].sum().item())
        del w, code, dp, ch
    ok = len(got[0]) == sum(1 for v in want.values() if v[0] > 0)
    for r in range(len(got[0])):
        gcode = (got[0].to_values()[r][0] - 65) * 2 + (got[1].to_values()[r][0] - 70)
        cnt, sq, sp, sdp, sch, sd = want[gcode]
        vals = [got[c].data[r] for c in range(2, 10)]
        exp = [sq, sp, sdp, sch, sq / cnt, sp / cnt, sd / cnt, cnt]
        ok = ok and all(abs(v - e) <= 1e-9 * abs(e) + 1e-6 for v, e in zip(vals, exp))
    del ship, rf, ls, offs, qty, price, disc, tax, base
    torch.cuda.empty_cache()
    return {"case": f"TPC-H Q1 in full at SF100 on one GPU: {n} lineitem rows resident in HBM, WHERE + 2 projections inside the "
                    "aggregation kernel, GROUP BY 2 bytes keys, 4 SUM + 3 AVG + COUNT_ROWS",
            "rows": n, "groups": len(got[0]), "ms": dt * 1e3, "rows_per_s": n / dt, "algorithmic_bytes": n * 42.0,
            "frac_of_8TBps": n * 42.0 / dt / 8e12, "kernels": kern, "result_check": "ok" if ok else "MISMATCH",
            "cpu_baseline": q1_cpu_sf1() if cpu else None}


def q9_sf100(device_index=0, sf=100.0, reps=3, cpu_sf=1.0):
    import torch
    from cockroach_amd import _lib, device as dev
    from tests import tpch_check as K
    from workloads import tpch_plans as P, tpch_synth as T
    t, rows = T.generate(sf, f"cuda:{device_index}", 42)
    torch.cuda.synchronize()
    aux = rows["_aux"]
    ex = P.GpuExec(account=True)  # untimed: accounting, warm-up, the checked result
    res = ex.to_host(P.q9(ex, {k: ex.table(v) for k, v in t.items()}))
    alg = ex.alg_bytes
    ex.close()
    nat, year, profit = K.q9_expected(t, aux)
    got = sorted(zip(res[0].to_values(), res[1].to_values(), res[2].to_values()))
    want = sorted(zip([T.NATIONS[i] for i in nat.tolist()], [float(y) for y in year.tolist()], profit.tolist()))
    ok = len(got) == len(want) and all(g[:2] == w[:2] and abs(g[2] - w[2]) <= 1e-9 * abs(w[2]) + 1e-6 for g, w in zip(got, want))
    del nat, year, profit
    torch.cuda.empty_cache()
    def passes():
        for _ in range(reps):
            ex = P.GpuExec()
            P.q9(ex, {k: ex.table(v) for k, v in t.items()})
            _lib.check(_lib.load().crgpu_dev_synchronize())
            ex.close()
    dev.prof_enable(False)  # wall time without the library's profiling hooks, then the same passes with them for the kernel times
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    passes()
    dt = (time.perf_counter() - t0) / reps
    dev.prof_enable(True)
    dev.prof_reset()
    passes()
    dev.prof_enable(False)
    kern = {}
    for f in ("expr_select", "expr_project", "deselect", "join_build", "join_probe", "radix_build_hist", "radix_build_scatter",
              "radix_build_split", "radix_build_table", "radix_probe_hist", "radix_probe_scatter", "agg"):
        ms, cnt = dev.prof_get(f)
        if cnt:
            kern[f] = round(ms / reps, 3)
    scanned = ["part", "partsupp", "lineitem", "supplier", "nation", "orders"]
    in_rows = sum(rows[k] for k in scanned)
    del t
    torch.cuda.empty_cache()
    _lib.load().crgpu_mem_trim()
    cpu = None
    if cpu_sf:
        from oracle.plan_exec import OracleExec
        tc, rc = T.generate(cpu_sf, "cpu", 42)
        oex = OracleExec()
        otabs = {k: oex.table(v) for k, v in tc.items()}
        t1 = time.perf_counter()
        P.q9(oex, otabs)
        dtc = time.perf_counter() - t1
        nc = sum(rc[k] for k in scanned)
        cpu = {"value": nc / dtc, "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"the same plan at SF{cpu_sf:g} ({nc} scanned rows) on the oracle executor (reference operators restated, 1 thread), {dtc:.1f} s"}
    return {"case": f"TPC-H Q9 at SF{sf:g} on one GPU (synthetic tables resident in HBM): six tables, five hash joins, GROUP BY nation, o_year",
            "rows": in_rows, "groups": len(got), "ms": dt * 1e3, "rows_per_s": in_rows / dt, "algorithmic_bytes": alg,
            "frac_of_8TBps": alg / dt / 8e12, "kernels_ms": kern, "result_check": "ok" if ok else "MISMATCH", "cpu_baseline": cpu}


def measure(device_index=0, reps=3, cpu=True, q9_sf=100.0):
    import torch
    from cockroach_amd import _lib
    torch.cuda.set_device(device_index)
    _lib.check(_lib.load().crgpu_init(device_index))
    out = [q1_sf100(device_index, reps=reps, cpu=cpu)]
    _lib.load().crgpu_mem_trim()
    out.append(q9_sf100(device_index, sf=q9_sf, reps=reps, cpu_sf=1.0 if cpu else 0.0))
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--q9-sf", type=float, default=100.0)
    a = ap.parse_args()
    for r in measure(0, a.reps, not a.no_cpu, a.q9_sf):
        print(json.dumps(r), flush=True)
