#!/usr/bin/env python3
"""The sub-measurements SURVEY.md 8(d) asks for beside the headline join: hash-only (colexechash.rehash), build-only
(config J build side), HashRouter repartition (TupleHashDistributor + per-output copies) and the config G variants
(SUM_INT + COUNT_ROWS over all-distinct int64 keys, each key 4x, 16-byte BYTES keys).  Inputs resident in HBM; each
line carries the single-threaded CPU port timed on a bounded sample of the same input.  One JSON line per case."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cockroach_amd import _lib, coldata as cd, device as dev  # noqa: E402

CPU_SAMPLE = 1 << 22
MIX = 0x9E3779B97F4A7C15 - (1 << 64)  # odd multiplier: k -> k * MIX is a bijection of the int64 values


def _sync():
    torch.cuda.synchronize()
    _lib.load().crgpu_dev_synchronize()


def _timed(fn, reps):
    _sync()  # inputs are produced on torch's stream, the library runs on its own: offsets read early are garbage
    fn()
    _sync()
    # wall time with the library's profiling hooks off (two events per kernel family and launch); the per-family kernel
    # times come from `reps` more passes with them on (_kernels divides by reps)
    dev.prof_enable(False)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    _lib.load().crgpu_dev_synchronize()
    dt = (time.perf_counter() - t0) / reps
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(reps):
        fn()
    _lib.load().crgpu_dev_synchronize()
    dev.prof_enable(False)
    return dt


def _kernels(names, reps):
    out = {}
    for k in names:
        ms, cnt = dev.prof_get(k)
        if cnt:
            out[k] = {"ms_per_pass": round(ms / reps, 3), "launches_per_pass": cnt / reps}
    return out


def _line(case, rows, dt, alg_bytes, kernels, cpu):
    return {"case": case, "rows": rows, "ms": dt * 1e3, "rows_per_s": rows / dt, "alg_GBps": alg_bytes / dt / 1e9,
            "frac_of_8TBps": alg_bytes / dt / 8e12, "kernels": kernels, "cpu_baseline": cpu}


def _cpu(fn, m, what):
    if not CPU_SAMPLE:
        return None
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    return {"value": m / dt, "unit": "rows/s", "cores": 1, "kind": "port", "sample": f"first {m} rows of the same input ({what}, 1 thread)"}


def bench_hash(lg, reps, dv, g):
    n = 1 << lg
    k = torch.randint(-2**62, 2**62, (n,), generator=g, device=dv, dtype=torch.int64)
    out = torch.empty(n, device=dv, dtype=torch.int64)
    col = dev.from_torch(cd.INT64, k)
    arr = dev.dcol_array([col])
    lib = _lib.load()

    def go():
        _lib.check(lib.crgpu_dev_hash_cols(arr, 1, n, 1, 64, 0, C.c_void_p(out.data_ptr())))
    dt = _timed(go, reps)
    kern = _kernels(["hash"], reps)
    m = min(n, CPU_SAMPLE)
    cpu = None
    if CPU_SAMPLE:
        from oracle import oracle as O  # test infrastructure: the reported CPU baseline only
        hv = cd.Vec.from_numpy(cd.INT64, k[:m].cpu().numpy())
        cpu = _cpu(lambda: O.hash_cols([hv], width=64), m, "oracle rehash restatement")
    return _line(f"hash only: rehash of one int64 column, 64-bit running hash, 2^{lg} rows", n, dt, n * 16.0, kern, cpu)


def bench_build(lg, reps, dv, g):
    n = 1 << lg
    k = torch.randperm(n, generator=g, device=dv, dtype=torch.int64)
    v = k * 3 + 1
    cols = [dev.from_torch(cd.INT64, k), dev.from_torch(cd.INT64, v)]
    j = dev.Join(0, [cd.INT64, cd.INT64], [cd.INT64, cd.INT64], [0], [0], True, out_cols=[1, 3])

    def go():
        j.reset()
        j.build(cols, n)
    dt = _timed(go, reps)
    kern = _kernels(["radix_build_hist", "radix_build_scatter", "radix_build_split", "radix_build_table", "join_build"], reps)
    j.close()
    return _line(f"build only: config J build side, 2^{lg} unique int64 keys + int64 payload", n, dt, n * 16.0, kern, None)


def bench_general_join(reps, dv, g, lg_b=24, lg_p=26, dup=4, two_cols=False):
    """The general join path: duplicate build keys (radix_build_dup: rows grouped per sub-partition; the chained table only
    takes over for heavily skewed keys and for the join types that track build matches).  INNER join,
    build keys drawn at random from nb/dup values (Poisson(dup) rows per key, `dup` on average); rows/s = (build + probe rows) / time for build + probe."""
    nb, npr = 1 << lg_b, 1 << lg_p
    bk = torch.randint(0, nb // dup, (nb,), generator=g, device=dv, dtype=torch.int64)
    bv = torch.arange(nb, device=dv, dtype=torch.int64)
    pk = torch.randint(0, nb // dup, (npr,), generator=g, device=dv, dtype=torch.int64)
    pw = torch.arange(npr, device=dv, dtype=torch.int64)
    if two_cols:  # the same keys as two equality columns: equality is decided on the build columns, not in the table
        build = [dev.from_torch(cd.INT64, bk >> 8), dev.from_torch(cd.INT64, bv), dev.from_torch(cd.INT64, bk & 255)]
        probe = [dev.from_torch(cd.INT64, pk >> 8), dev.from_torch(cd.INT64, pw), dev.from_torch(cd.INT64, pk & 255)]
        j = dev.Join(0, [cd.INT64] * 3, [cd.INT64] * 3, [0, 2], [0, 2], False, out_cols=[1, 4])
    else:
        build = [dev.from_torch(cd.INT64, bk), dev.from_torch(cd.INT64, bv)]
        probe = [dev.from_torch(cd.INT64, pk), dev.from_torch(cd.INT64, pw)]
        j = dev.Join(0, [cd.INT64, cd.INT64], [cd.INT64, cd.INT64], [0], [0], False, out_cols=[1, 3])
    out_rows = [0]

    def go():
        j.reset()
        j.build(build, nb)
        out_rows[0] = int(j.probe(probe, npr).n)
    dt = _timed(go, reps)
    kern = _kernels(["join_build", "join_probe", "radix_build_hist", "radix_build_scatter", "radix_build_split", "radix_build_table",
                     "radix_probe_hist", "radix_probe_scatter"], reps)
    j.close()
    cpu = None
    if CPU_SAMPLE and not two_cols:
        from oracle import oracle as O
        sb, sp = min(nb, CPU_SAMPLE // 4), min(npr, CPU_SAMPLE)
        hb = [cd.Vec.from_numpy(cd.INT64, (bk[:sb] % (sb // dup)).cpu().numpy()), cd.Vec.from_numpy(cd.INT64, bv[:sb].cpu().numpy())]
        hp = [cd.Vec.from_numpy(cd.INT64, (pk[:sp] % (sb // dup)).cpu().numpy()), cd.Vec.from_numpy(cd.INT64, pw[:sp].cpu().numpy())]
        t0 = time.perf_counter()
        O.hash_join(0, hp, hb, [0], [0], False)
        cdt = time.perf_counter() - t0
        cpu = {"value": (sb + sp) / cdt, "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{sb} build x {sp} probe rows of the same shape (keys folded to keep {dup} duplicates per key; oracle hashJoiner, 1 thread)"}
    r = _line(f"general join (duplicate build keys, grouped radix table{', two int64 key columns packed into one key word' if two_cols else ''}): 2^{lg_b} build (random keys, {dup} rows per key on average) x 2^{lg_p} probe, INNER, {out_rows[0]} output rows",
              nb + npr, dt, 16.0 * nb + 16.0 * npr + 16.0 * out_rows[0], kern, cpu)
    r["output_rows"] = out_rows[0]
    return r


def wrap_torch_bitmap(col, n, dv):
    """bool tensor of a device column's validity bitmap (bit = 1: present); all True without one."""
    if not col.valid_ptr:
        return torch.ones(n, dtype=torch.bool, device=dv)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import wrap_torch_ptr
    by = wrap_torch_ptr(col.valid_ptr, (n + 7) // 8, torch.uint8, dv)
    bits = (by.unsqueeze(1) >> torch.arange(8, device=dv, dtype=torch.uint8)) & 1
    return bits.reshape(-1)[:n].to(torch.bool)


def bench_group_join_fused(reps, dv, g, lg_b=24, lg_p=27, outer=False):
    """The same group-join through the fused form (crgpu_dev_gjoin_*: aggregate while probing, no joined row is
    materialised), same inputs and the same torch check as bench_join_then_agg.  outer: the join keeps the build side
    (CRGPU_RIGHT_OUTER): a group per built order, SUM NULL and COUNT_ROWS 1 where no lineitem matches."""
    nb, npr = 1 << lg_b, 1 << lg_p
    okeys = torch.randperm(nb, generator=g, device=dv, dtype=torch.int64)[: nb // 2].contiguous() * 7 + 3
    lkey = torch.randint(0, nb, (npr,), generator=g, device=dv, dtype=torch.int64) * 7 + 3
    price = torch.randint(1, 100000, (npr,), generator=g, device=dv, dtype=torch.int64).to(torch.float64) / 100.0
    bk, pk, pv = dev.from_torch(cd.INT64, okeys), dev.from_torch(cd.INT64, lkey), dev.from_torch(cd.FLOAT64, price)
    gj = dev.GroupJoin([(0, 0), (10, 1), (14, 0)], probe_payload_type=cd.FLOAT64, join_type=2 if outer else 0)
    out = {}

    def go():
        gj.build(bk, None, nb // 2)
        out["res"] = gj.probe(pk, pv, npr)
    dt = _timed(go, reps)
    kern = _kernels(["radix_build_hist", "radix_build_scatter", "radix_build_split", "radix_build_table", "gjoin_fused"], reps)
    res = out["res"]
    ng = int(res.n)
    rc = dev.result_columns(res)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import wrap_torch
    gk = wrap_torch(rc[0], ng, torch.int64, dv).clone()
    gs = wrap_torch(rc[1], ng, torch.float64, dv).clone()
    gc = wrap_torch(rc[2], ng, torch.int64, dv).clone()
    hit = torch.isin(lkey, okeys)
    uk, inv = torch.unique(lkey[hit], return_inverse=True)
    want_s = torch.zeros(uk.numel(), device=dv, dtype=torch.float64).index_add_(0, inv, price[hit])
    want_c = torch.bincount(inv, minlength=uk.numel())
    order = torch.argsort(gk)
    if outer:
        # every built order is a group; the matched ones carry the sums and counts above, the others NULL and 1
        valid = wrap_torch_bitmap(rc[1], ng, dv)
        sk = gk[order]
        pos = torch.searchsorted(sk, uk)
        matched = torch.zeros(ng, dtype=torch.bool, device=dv)
        matched[pos] = True
        ok = ng == nb // 2 and torch.equal(sk, torch.sort(okeys).values) and torch.equal(valid[order], matched) and \
            torch.equal(gc[order][pos], want_c) and bool((gc[order][~matched] == 1).all()) and \
            bool(torch.allclose(gs[order][pos], want_s, rtol=1e-9, atol=1e-6))
    else:
        ok = ng == uk.numel() and torch.equal(gk[order], uk) and torch.equal(gc[order], want_c) and \
            bool(torch.allclose(gs[order], want_s, rtol=1e-9, atol=1e-6))
    if not ok:
        raise SystemExit("fused group-join: result check failed")
    gj.close()
    r = _line(f"hash group-join, FUSED{', build side kept (RIGHT OUTER)' if outer else ''} (aggregate while probing): 2^{lg_b - 1} built orders x 2^{lg_p} lineitems -> GROUP BY key: SUM + COUNT_ROWS, {ng} groups",
              nb // 2 + npr, dt, 16.0 * (nb // 2) + 16.0 * npr + 24.0 * ng, kern, None)
    r["result_check"] = "ok"
    return r


def bench_join_then_agg(reps, dv, g, lg_b=24, lg_p=27):
    """Hash group-join as the reference runs it (colexec/hash_group_joiner.go:20-76: hash joiner -> hash aggregator),
    device-resident: the join's output columns go straight into crgpu_dev_agg_consume.  Q3/Q18-like shape: 2^lg_b
    orders (unique key, half of them pass an upstream filter and are built), 2^lg_p lineitems probe, then
    GROUP BY l_orderkey: SUM(price), COUNT_ROWS.  Checked against torch reductions."""
    nb, npr = 1 << lg_b, 1 << lg_p
    okeys = torch.randperm(nb, generator=g, device=dv, dtype=torch.int64)[: nb // 2].contiguous() * 7 + 3
    odate = torch.randint(0, 2500, (nb // 2,), generator=g, device=dv, dtype=torch.int64)
    lkey = torch.randint(0, nb, (npr,), generator=g, device=dv, dtype=torch.int64) * 7 + 3
    price = torch.randint(1, 100000, (npr,), generator=g, device=dv, dtype=torch.int64).to(torch.float64) / 100.0
    build = [dev.from_torch(cd.INT64, okeys), dev.from_torch(cd.INT64, odate)]
    probe = [dev.from_torch(cd.INT64, lkey), dev.from_torch(cd.FLOAT64, price)]
    j = dev.Join(0, [cd.INT64, cd.FLOAT64], [cd.INT64, cd.INT64], [0], [0], True, out_cols=[0, 1])  # emits (l_orderkey, price)
    a = dev.Agg([cd.INT64, cd.FLOAT64], [0], [(0, 0, -1), (10, 1, -1), (14, -1, -1)])
    out = {}

    def go():
        j.reset()
        a.reset()
        j.build(build, nb // 2)
        res = j.probe(probe, npr)
        out["joined"] = int(res.n)
        a.consume(dev.result_columns(res), int(res.n))
        out["res"] = a.finish()
    dt = _timed(go, reps)
    kern = _kernels(["radix_build_hist", "radix_build_scatter", "radix_build_split", "radix_build_table", "radix_probe_hist",
                     "radix_probe_scatter", "join_probe"] + AGG_KERNELS, reps)
    # check: groups = distinct probed keys that exist on the build side; sums per key
    res = out["res"]
    ng = int(res.n)
    rc = dev.result_columns(res)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import wrap_torch
    gk = wrap_torch(rc[0], ng, torch.int64, dv).clone()
    gs = wrap_torch(rc[1], ng, torch.float64, dv).clone()
    gc = wrap_torch(rc[2], ng, torch.int64, dv).clone()
    hit = torch.isin(lkey, okeys)
    uk, inv = torch.unique(lkey[hit], return_inverse=True)
    want_s = torch.zeros(uk.numel(), device=dv, dtype=torch.float64).index_add_(0, inv, price[hit])
    want_c = torch.bincount(inv, minlength=uk.numel())
    order = torch.argsort(gk)
    ok = ng == uk.numel() and torch.equal(gk[order], uk) and torch.equal(gc[order], want_c) and \
        bool(torch.allclose(gs[order], want_s, rtol=1e-9, atol=1e-6)) and out["joined"] == int(hit.sum())
    if not ok:
        raise SystemExit("join -> aggregate pipeline: result check failed")
    j.close(); a.close()
    r = _line(f"hash group-join, device-resident: 2^{lg_b - 1} built orders x 2^{lg_p} lineitems -> {out['joined']} joined rows -> GROUP BY key: SUM + COUNT_ROWS, {ng} groups",
              nb // 2 + npr, dt, 16.0 * (nb // 2) + 16.0 * npr + 24.0 * ng, kern, None)
    r["result_check"] = "ok"
    return r


def bench_router(lg, reps, dv, g, outputs):
    n = 1 << lg
    k = torch.randint(0, 1 << 40, (n,), generator=g, device=dv, dtype=torch.int64)
    w = torch.arange(n, device=dv, dtype=torch.int64)
    cols = [dev.from_torch(cd.INT64, k), dev.from_torch(cd.INT64, w)]
    p = dev.Partitioner([cd.INT64, cd.INT64], [0], outputs)
    counts = [None]

    def go():
        _res, counts[0] = p.partition(cols, n)
    dt = _timed(go, reps)
    kern = _kernels(["partition"], reps)
    assert int(counts[0].sum()) == n
    m = min(n, CPU_SAMPLE)
    cpu = None
    if CPU_SAMPLE:
        from oracle import oracle as O
        hv = [cd.Vec.from_numpy(cd.INT64, k[:m].cpu().numpy()), cd.Vec.from_numpy(cd.INT64, w[:m].cpu().numpy())]
        cpu = _cpu(lambda: O.distribute(hv, [0], outputs), m, "oracle TupleHashDistributor restatement, selection vectors only")
    p.close()
    return _line(f"router repartition: (int64 key, int64 payload) into {outputs} outputs, 2^{lg} rows", n, dt, n * 32.0, kern, cpu)


AGG_KERNELS = ["agg_fused_lds", "agg_fused_sumfam", "agg_fused_hbm", "agg_resolve", "agg_fixup", "agg_accum_lds", "agg_accum_hbm", "agg_reindex",
               "agg_any_fixup", "agg_part_hist", "agg_part_scatter", "agg_part_accum", "agg_bulk", "agg_bulk_distinct", "agg_bulk_part", "radix_build_hist",
               "radix_build_scatter", "radix_build_split"]


def _agg(case, cols, types, aggs, n, alg_bytes, reps, expect):
    a = dev.Agg(types, [0], aggs)
    ng = [0]

    def go():
        a.reset()
        a.consume(cols, n)
        ng[0] = int(a.finish().n)
    dt = _timed(go, reps)
    kern = _kernels(AGG_KERNELS, reps)
    if ng[0] != expect:
        raise SystemExit(f"{case}: {ng[0]} groups, expected {expect}")
    cpu = None
    if CPU_SAMPLE:
        from oracle import oracle as O
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from bench_agg import to_host_vec
        m = min(n, CPU_SAMPLE)
        host = [to_host_vec(c, m) for c in cols]
        cpu = _cpu(lambda: O.hash_agg(host, [0], aggs), m, "oracle hashAggregator restatement")
    a.close()
    r = _line(case, n, dt, alg_bytes, kern, cpu)
    r["groups"] = ng[0]
    return r


def bench_config_g(lg, reps, dv, g, which):
    n = 1 << lg
    aggs = [(0, 0, -1), (11, 1, -1), (14, -1, -1)]  # ANY_NOT_NULL(k), SUM_INT(x), COUNT_ROWS
    x = torch.randint(-1000, 1000, (n,), generator=g, device=dv, dtype=torch.int64)
    if which == "distinct":
        k = torch.randperm(n, generator=g, device=dv, dtype=torch.int64) * MIX
        cols = [dev.from_torch(cd.INT64, k), dev.from_torch(cd.INT64, x)]
        return _agg(f"config G: 2^{lg} rows, all keys distinct, SUM_INT + COUNT_ROWS", cols, [cd.INT64, cd.INT64], aggs, n,
                    n * 16.0 + n * 24.0, reps, n)
    if which == "x4":
        k = (torch.randperm(n, generator=g, device=dv, dtype=torch.int64) >> 2) * MIX
        cols = [dev.from_torch(cd.INT64, k), dev.from_torch(cd.INT64, x)]
        return _agg(f"config G: 2^{lg} rows, each key 4x, SUM_INT + COUNT_ROWS", cols, [cd.INT64, cd.INT64], aggs, n,
                    n * 16.0 + (n // 4) * 24.0, reps, n // 4)
    # 16-byte BYTES keys: lowercase hex of the (distinct) key
    k = torch.randperm(n, generator=g, device=dv, dtype=torch.int64) * MIX
    hexd = torch.tensor(list(b"0123456789abcdef"), device=dv, dtype=torch.uint8)
    data = torch.empty((n, 16), device=dv, dtype=torch.uint8)
    for d in range(16):
        data[:, d] = hexd[((k >> (60 - 4 * d)) & 15)]
    del k
    offs = torch.arange(n + 1, device=dv, dtype=torch.int32) * 16
    cols = [dev.from_torch(cd.BYTES, data.reshape(-1), offsets=offs), dev.from_torch(cd.INT64, x)]
    return _agg(f"config G: 2^{lg} rows, distinct 16-byte BYTES keys (hex), SUM_INT + COUNT_ROWS", cols, [cd.BYTES, cd.INT64], aggs, n,
                n * 28.0 + n * 36.0, reps, n)


def bench_distinct(lg, reps, dv, g, nkeys):
    """Unordered DISTINCT on (key) of (key, payload) rows, one call; algorithmic bytes = rows in + distinct rows out."""
    n = 1 << lg
    k = torch.randperm(n, generator=g, device=dv, dtype=torch.int64) * MIX if nkeys >= n else \
        torch.randint(0, nkeys, (n,), generator=g, device=dv, dtype=torch.int64) * MIX
    w = torch.arange(n, device=dv, dtype=torch.int64)
    cols = [dev.from_torch(cd.INT64, k), dev.from_torch(cd.INT64, w)]
    d = dev.Distinct([cd.INT64, cd.INT64], [0])
    m = [0]

    def go():
        d.reset()
        m[0] = int(d.consume(cols, n).n)
    dt = _timed(go, reps)
    kern = _kernels(["distinct_resolve", "distinct_publish", "distinct_gather", "distinct_reindex", "distinct_bulk", "distinct_bulk_part", "radix_build_hist", "radix_build_scatter", "radix_build_split"], reps)
    cpu = None
    if CPU_SAMPLE:
        from oracle import oracle as O
        s = min(n, CPU_SAMPLE)
        hv = [cd.Vec.from_numpy(cd.INT64, k[:s].cpu().numpy()), cd.Vec.from_numpy(cd.INT64, w[:s].cpu().numpy())]
        cpu = _cpu(lambda: O.unordered_distinct(hv, [0]), s, "oracle UnorderedDistinct restatement")
    d.close()
    r = _line(f"unordered DISTINCT on the int64 key of (key, payload), 2^{lg} rows, {m[0]} distinct", n, dt, (n + m[0]) * 16.0, kern, cpu)
    r["groups"] = m[0]
    return r


def measure(only=None, reps=3, cpu_sample=None, device_index=0, lg_hash=29, lg_rows=27, lg_bytes=26):
    global CPU_SAMPLE
    if cpu_sample is not None:
        CPU_SAMPLE = cpu_sample
    _lib.check(_lib.load().crgpu_init(device_index))
    dv = torch.device("cuda", device_index)
    g = torch.Generator(device=dv); g.manual_seed(7)
    want = lambda name: only is None or name in only  # noqa: E731
    if want("hash"):
        yield bench_hash(lg_hash, reps, dv, g)
    if want("build"):
        yield bench_build(lg_rows, reps, dv, g)
    if want("join_agg"):
        yield bench_join_then_agg(reps, dv, g)
    if want("join_agg_fused"):
        yield bench_group_join_fused(reps, dv, g)
        yield bench_group_join_fused(reps, dv, g, outer=True)
    if want("general_join"):
        yield bench_general_join(reps, dv, g)
    if want("general_join2"):
        yield bench_general_join(reps, dv, g, two_cols=True)
    if want("router"):
        yield bench_router(lg_rows, reps, dv, g, int(os.environ.get("ROUTER_OUTPUTS", "8")))
    if want("g_distinct"):
        yield bench_config_g(lg_rows, reps, dv, g, "distinct")
    if want("g_x4"):
        yield bench_config_g(lg_rows, reps, dv, g, "x4")
    if want("distinct"):
        for nkeys in (1 << 10, 1 << 20, 1 << lg_rows):
            yield bench_distinct(lg_rows, reps, dv, g, nkeys)
    if want("g_bytes"):
        yield bench_config_g(lg_bytes, reps, dv, g, "bytes")


def main():
    only = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else None
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cpu_sample = int(sys.argv[3]) if len(sys.argv) > 3 else None
    for r in measure(only, reps, cpu_sample):
        print(json.dumps(r), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
