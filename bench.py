#!/usr/bin/env python3
"""bench.py -- BASELINE.json configs[1]: "Synthetic hash join microbench: int64
keys, 2^27 build x 2^29 probe rows, 1 MI355X" (SURVEY.md 8(d) config J).

One step = one full hash join of the batch: build the table on (k, v)
(2^27 rows, k a permutation) + probe it with (k, w) (2^29 rows, 100 % hit) +
materialise (w, v) (2^29 rows).  Inputs are resident in HBM before the timed
region.  N > 1 (one process per GPU, RCCL): every rank holds its own 2^27 x
2^29 shard (weak scaling), hash-repartitions both inputs by key with the
reference's router hash (crgpu_dev_partition), exchanges the regions with one
all-to-all per column and joins locally -- BASELINE.json configs[3] shape.

Prints ONE JSON line (see the repo contract); `roofline` is for the dominant
kernel of the step (largest HIP-event time; on config J the radix probe kernel),
`cpu_baseline` times the oracle port on a bounded sample at N=1.
`roofline.traffic` comes from the committed rocprofv3 --pmc passes of this same
command (profiles/r01/bench_J_radix_pmc_summary.txt, scripts/prof_pmc.sh): it is
only reported for the default sizes those passes were taken at.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def np_log(x):
    import math
    return math.log(x)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2-build", type=int, default=27)
    ap.add_argument("--log2-probe", type=int, default=29)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2-build", type=int, default=25)
    ap.add_argument("--cpu-log2-probe", type=int, default=27)
    ap.add_argument("--no-other-paths", action="store_true", help="skip the aggregation / decode side measurements")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the N > 1 code path (hash repartition -> crgpu_exchange over RCCL -> local join, then the "
                         "two-stage GROUP BY) even with one rank: a self-exchange through RCCL")
    ap.add_argument("--probe-chunks", type=int, default=0,
                    help="N > 1: pieces the probe side is repartitioned and exchanged in; piece c + 1 is partitioned and "
                         "exchanged by a second thread while piece c is probed (1: one piece, no overlap; 0 = 4 with more "
                         "than one rank, 1 for --force-distributed with one rank, where the 'exchange' is a device copy "
                         "that competes with the probe for the same HBM: 27.0 ms against 25.4)")
    ap.add_argument("--log2-groupby", type=int, default=26, help="rows per GPU of the distributed GROUP BY leg (N > 1)")
    ap.add_argument("--no-q9", action="store_true", help="skip the distributed TPC-H Q9 leg of an N > 1 run")
    ap.add_argument("--q9-sf", type=float, default=0.0,
                    help="scale factor of the distributed TPC-H Q9 leg (BASELINE.json configs[3]); 0: SF100 on N > 1 GPUs, "
                         "SF10 for --force-distributed with one rank")
    ap.add_argument("--variant", default="hit100", choices=["hit100", "hit50", "zipf", "null1", "sparse"],
                    help="probe-key variants of config J (SURVEY.md 8(d)): 100%% hit (the headline), 50%% hit, "
                         "Zipf-like skew (theta=1), 1%% NULL probe keys")
    return ap.parse_args(argv)


def cpu_baseline(lb, lp):
    """The oracle port (oracle/crdb_oracle.c: reference algorithm, single thread,
    1024-tuple batches) on a bounded sample of the same workload."""
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    from cockroach_amd import coldata as cd
    rng = np.random.default_rng(42)
    nb, npr = 1 << lb, 1 << lp
    bk = rng.permutation(nb).astype(np.int64)
    bv = bk * 3 + 1
    pk = rng.integers(0, nb, size=npr, dtype=np.int64)
    pw = np.arange(npr, dtype=np.int64)
    L = [cd.Vec.from_numpy(cd.INT64, pk), cd.Vec.from_numpy(cd.INT64, pw)]
    R = [cd.Vec.from_numpy(cd.INT64, bk), cd.Vec.from_numpy(cd.INT64, bv)]
    cl, _k1 = O.make_cols(L)
    cr, _k2 = O.make_cols(R)
    eq = np.array([0], dtype=np.int32)
    t0 = time.perf_counter()
    res = O.lib().ora_hash_join(0, cl, 2, npr, cr, 2, nb, eq.ctypes.data, eq.ctypes.data, 1, 1, 1024, 256)
    dt = time.perf_counter() - t0
    nout = O.lib().ora_result_nrows(res)
    O.lib().ora_result_free(res)
    assert nout == npr
    return {"value": (nb + npr) / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"same workload at 2^{lb} build x 2^{lp} probe rows (oracle/crdb_oracle.c, 1 thread, "
                      f"1024-tuple batches), {dt:.1f} s"}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline_pway(lb, lp, P):
    """SURVEY.md 8(d) baseline (2): the reference's distributed shape on the host cores -- P "nodes", each routing its
    1/P share of both inputs through a HashRouter (TupleHashDistributor over P outputs, oracle/crdb_oracle.c
    ora_distribute) to P hash joiners, one thread each (ora_hash_join on its bucket).  The routing is inside the timed
    region.  Threads, not processes: the oracle's calls release the GIL."""
    import threading
    import numpy as np
    from oracle import oracle as O
    from cockroach_amd import coldata as cd
    rng = np.random.default_rng(42)
    nb, npr = 1 << lb, 1 << lp
    bk = rng.permutation(nb).astype(np.int64)
    bv = bk * 3 + 1
    pk = rng.integers(0, nb, size=npr, dtype=np.int64)
    pw = np.arange(npr, dtype=np.int64)
    sides = {"L": [pk, pw], "R": [bk, bv]}
    routed = {(s_, i): None for s_ in sides for i in range(P)}

    def route(i):  # node i: its 1/P share of both inputs through a HashRouter each (physical_plan.go:1120-1157)
        for name, (k, v) in sides.items():
            lo, hi = len(k) * i // P, len(k) * (i + 1) // P
            ks_, vs_ = k[lo:hi], v[lo:hi]
            dest, counts = O.distribute([cd.Vec.from_numpy(cd.INT64, ks_)], [0], P)
            order = np.argsort(dest, kind="stable")  # the router's per-output row copies
            ks, vs = ks_[order], vs_[order]
            ends = np.cumsum(counts)
            routed[(name, i)] = [(ks[e - c:e], vs[e - c:e]) for c, e in zip(counts, ends)]

    outs = [0] * P

    def join(p):  # node p's joiner: what the P routers sent to bucket p
        lk, lw = (np.concatenate([routed[("L", i)][p][j] for i in range(P)]) for j in (0, 1))
        rk, rv = (np.concatenate([routed[("R", i)][p][j] for i in range(P)]) for j in (0, 1))
        L = [cd.Vec.from_numpy(cd.INT64, lk), cd.Vec.from_numpy(cd.INT64, lw)]
        R = [cd.Vec.from_numpy(cd.INT64, rk), cd.Vec.from_numpy(cd.INT64, rv)]
        cl, _k1 = O.make_cols(L)
        cr, _k2 = O.make_cols(R)
        eq = np.array([0], dtype=np.int32)
        res = O.lib().ora_hash_join(0, cl, 2, len(lk), cr, 2, len(rk), eq.ctypes.data, eq.ctypes.data, 1, 1, 1024, 256)
        outs[p] = O.lib().ora_result_nrows(res)
        O.lib().ora_result_free(res)

    t0 = time.perf_counter()
    ts = [threading.Thread(target=route, args=(i,)) for i in range(P)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    t_route = time.perf_counter() - t0
    ts = [threading.Thread(target=join, args=(p,)) for p in range(P)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.perf_counter() - t0
    assert sum(outs) == npr
    return {"value": (nb + npr) / dt, "unit": "rows/s", "cores": P, "kind": "port", "cpu": cpu_model(),
            "sample": f"same workload at 2^{lb} build x 2^{lp} probe rows: {P} threads route their share of both inputs "
                      f"(ora_distribute, {t_route:.1f} s), then {P} joiner threads (ora_hash_join per bucket), {dt:.1f} s in all"}


PMC_SUMMARY = os.path.join(ROOT, "profiles", "r03", "bench_J_pmc_summary.txt")
KERNEL_SYMBOL = {"join_probe": "radix_probe_kernel<0, 0", "radix_build_split": "radix_split_kernel",
                 "radix_build_table": "_table_kernel"}


def pmc_traffic(prof_name, streamed_read_bytes):
    """HBM bytes per launch of one kernel from the committed --pmc passes: WRITE_SIZE (KiB) is exact;
    FETCH_SIZE (KiB) counts the wide coalesced input stream at half its size on gfx950
    (MI355X_MICROARCH.md, HBM), so the missing half of that known stream is added back."""
    sym = KERNEL_SYMBOL.get(prof_name)
    if sym is None or not os.path.exists(PMC_SUMMARY):
        return None
    fetch = write = None
    with open(PMC_SUMMARY) as f:
        for line in f:
            if not line.startswith("pmc,") or sym not in line:
                continue
            val = float(line.split("per_launch=")[1].split(",")[0])
            if ",FETCH_SIZE," in line:
                fetch = val
            elif ",WRITE_SIZE," in line:
                write = val
    if fetch is None or write is None:
        return None
    return fetch * 1024.0 + streamed_read_bytes / 2.0 + write * 1024.0


def main(argv=None):
    args = parse(argv)
    import torch
    import torch.distributed as dist
    from cockroach_amd import coldata as cd
    from cockroach_amd import device as dev
    from cockroach_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    _lib.check(_lib.load().crgpu_init(local_rank))
    distributed = world > 1 or args.force_distributed
    xch = None
    if distributed:
        # the exchange lives behind the C-ABI (crgpu_exchange_*: RCCL directly); torch.distributed only carries the
        # bootstrap id, the barriers and the timing reductions of this harness
        uid = torch.zeros(dev.Exchange.ID_BYTES, dtype=torch.uint8, device=device)
        if rank == 0:
            uid = torch.tensor(list(dev.Exchange.unique_id()), dtype=torch.uint8, device=device)
        if world > 1:
            dist.broadcast(uid, 0)
        xch = dev.Exchange(bytes(uid.cpu().tolist()), world, rank)

    nb, npr = 1 << args.log2_build, 1 << args.log2_probe
    g = torch.Generator(device=device)
    g.manual_seed(42 + rank)
    # build (k, v): k = a permutation of the GLOBAL key domain shard-by-shard (unique keys), v = 3k+1
    bk = torch.randperm(nb, generator=g, device=device, dtype=torch.int64)
    if world > 1:
        bk = bk * world + rank  # globally unique keys, spread over all destinations
    bv = bk * 3 + 1
    g.manual_seed(43 + rank)
    pk_valid = None
    if args.variant == "hit50":
        pk = torch.randint(0, 2 * nb * world, (npr,), generator=g, device=device, dtype=torch.int64)
    elif args.variant == "zipf":
        # P(rank r) ~ 1/r over [1, nb*world): inverse CDF of the continuous 1/x density, ranks spread over the key
        # domain by an odd multiplier (a bijection mod 2^k)
        u = torch.rand(npr, generator=g, device=device, dtype=torch.float64)
        r = torch.exp(u * float(np_log(nb * world))).to(torch.int64).clamp_(1, nb * world - 1)
        pk = (r * 0x9E3779B1) % (nb * world)
        del u, r
    else:
        pk = torch.randint(0, nb * world, (npr,), generator=g, device=device, dtype=torch.int64)  # 100 % hit
    if args.variant == "null1":
        isnull = torch.rand(npr, generator=g, device=device) < 0.01
        bits = (~isnull).view(-1, 8).to(torch.uint8)  # coldata.Nulls: bit = 1 => present, LSB first
        pk_valid = (bits * (2 ** torch.arange(8, device=device, dtype=torch.uint8))).sum(1, dtype=torch.uint8).contiguous()
    pw = torch.arange(npr, device=device, dtype=torch.int64) + rank * npr
    pk_base = pk  # the key's position in the dense domain: what the result check reasons about
    if args.variant == "sparse":
        # the same 100 %-hit workload over keys 7919 apart: the build keys are not dense, so the join takes the hashed
        # (cuckoo) sub-partition tables instead of the direct-addressed ones
        bk = bk * 7919 + 13
        bv = bk * 3 + 1
        pk = pk * 7919 + 13
    torch.cuda.synchronize()

    I64 = cd.INT64
    join = dev.Join(0, [I64, I64], [I64, I64], [0], [0], True, out_cols=[1, 3])  # INNER, output (w, v)
    part_b = xch_b = None
    parts_p, xchs_p = [], []
    chunks = (args.probe_chunks if args.probe_chunks > 0 else (4 if world > 1 else 1)) if distributed else 1
    if distributed:
        part_b = dev.Partitioner([I64, I64], [0], world)
        # one exchange handle per join input: a result is owned by its handle until that handle's next call; the probe
        # side travels in `chunks` pieces through two (partitioner, exchange) pairs, so that piece c + 1 is partitioned
        # and exchanged while piece c is probed (crgpu_exchange_clone: the same communicator, own stream and buffers)
        xch_b = xch
        for _ in range(min(2, chunks)):
            parts_p.append(dev.Partitioner([I64, I64], [0], world))
            xchs_p.append(xch.clone())

    def exchange(part, x, cols_t, n):
        """hash-repartition (crgpu_dev_partition) + RCCL all-to-all (crgpu_exchange_all_to_all); device columns."""
        res, counts = part.partition([dev.from_torch(I64, t) for t in cols_t], n)
        got, rc = x.all_to_all(dev.result_columns(res), counts)
        return dev.result_columns(got), int(rc.sum())

    def probe_pipelined(check):
        """The probe side in `chunks` pieces: a producer thread partitions and exchanges piece c + 1 (its own handles and
        streams; the only thread that touches the communicator meanwhile) while this thread probes piece c.  Every call
        of the library waits for its own stream before it returns, so a piece handed over is complete, and a slot is
        only reused once the probe that read it has returned.  Returns (last result, output rows of all pieces)."""
        import queue
        import threading
        nslots = len(parts_p)
        bounds = [npr * c // chunks for c in range(chunks + 1)]
        free = [threading.Semaphore(1) for _ in range(nslots)]
        ready = queue.Queue()
        stop = threading.Event()

        def produce():
            try:
                for c in range(chunks):
                    s = c % nslots
                    free[s].acquire()
                    if stop.is_set():
                        return
                    lo, hi = bounds[c], bounds[c + 1]
                    cols, n = exchange(parts_p[s], xchs_p[s], [pk[lo:hi], pw[lo:hi]], hi - lo)
                    ready.put((s, cols, n, None))
            except BaseException as e:  # noqa: BLE001 - handed to the consumer
                ready.put((None, None, 0, e))

        t = threading.Thread(target=produce, name="probe-side exchange", daemon=True)
        t.start()
        res, total = None, 0
        try:
            for _ in range(chunks):
                try:
                    s, cols, n, err = ready.get(timeout=300)
                except queue.Empty:  # a peer died or the collectives went out of step: fail, do not hang the job
                    sys.stderr.write(f"rank {rank}: no probe piece arrived within 300 s; giving up\n")
                    os._exit(4)
                if err is not None:
                    raise err
                res = join.probe(cols, n)
                total += int(res.n)
                if check is not None:
                    check(res)
                free[s].release()
        finally:
            stop.set()
            for f in free:
                f.release()
            t.join(timeout=300)
            if t.is_alive():
                sys.stderr.write(f"rank {rank}: the probe-side exchange thread did not finish; giving up\n")
                os._exit(4)
        return res, total

    def step(check=None):
        """One pass; returns (result of the last probe call, output rows of the pass, columns kept alive)."""
        if not distributed:
            b_cols = [dev.from_torch(I64, t) for t in (bk, bv)]
            p_cols = [dev.from_torch(I64, t) for t in (pk, pw)]
            nbl, npl = nb, npr
            if pk_valid is not None:
                p_cols[0] = dev.from_torch(I64, pk, valid=pk_valid)
        else:
            b_cols, nbl = exchange(part_b, xch_b, [bk, bv], nb)
            if chunks > 1:
                join.build(b_cols, nbl)
                res, total = probe_pipelined(check)
                return res, total, b_cols, None
            p_cols, npl = exchange(parts_p[0], xchs_p[0], [pk, pw], npr)
        join.build(b_cols, nbl)
        res = join.probe(p_cols, npl)
        if check is not None:
            check(res)
        return res, int(res.n), b_cols, p_cols

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res, out_rows, _b, _p = step()
    dev.prof_enable(True)
    dev.prof_reset()
    barrier()
    t0 = time.perf_counter()
    out_rows = 0
    for _ in range(args.steps):
        res, out_rows, _b, _p = step()
    barrier()
    dt = time.perf_counter() - t0
    dev.prof_enable(False)

    # result check outside the timed region (size-independent property: sum(v) == sum(3k+1) over probe keys)
    # The union of all ranks' build keys is a permutation of [0, nb * world) (rank r holds k * world + r), so a probe row
    # hits iff its key is below nb * world (and not NULL): what THIS rank's probe rows must contribute to the output of
    # the whole job -- rows, sum(v) = sum(3k + 1), sum(w) -- is known from its inputs, wherever the exchange sent them.
    hit = pk_base < nb * world
    if args.variant == "null1":
        hit = hit & ~isnull
    want = torch.stack([hit.sum(), ((pk * 3 + 1) * hit).sum(), (pw * hit).sum()]).to(torch.int64)
    acc = {"rows": 0, "v": torch.zeros((), dtype=torch.int64, device=device), "w": torch.zeros((), dtype=torch.int64, device=device),
           "misrouted": torch.zeros(1, dtype=torch.int64, device=device)}

    def check_result(r):
        """Adds one probe result's row count, sum(v), sum(w) (int64 sums wrap alike) and its misrouted rows: every output
        row must sit on the rank its key routes to -- k = (v - 1) / 3, dest = rehash64(1, k) % world."""
        n_out = int(r.n)
        if n_out == 0:
            return
        w_o = _wrap(torch, r.cols[0].data, n_out, device)
        v_o = _wrap(torch, r.cols[1].data, n_out, device)
        acc["rows"] += n_out
        acc["v"] += v_o.sum()
        acc["w"] += w_o.sum()
        if distributed:
            ns = min(n_out, 1 << 24)
            k_out = ((v_o[:ns] - 1) // 3).contiguous()
            dest = torch.empty(ns, dtype=torch.int64, device=device)
            torch.cuda.synchronize()
            _lib.check(_lib.load().crgpu_dev_hash_cols(dev.dcol_array([dev.from_torch(I64, k_out)]), 1, ns, 1, 64, world, dest.data_ptr()))
            _lib.check(_lib.load().crgpu_dev_synchronize())
            acc["misrouted"] += (dest != rank).sum().reshape(1)
        torch.cuda.synchronize()  # the result's buffers go back to the join with its next call

    if chunks > 1:
        # a piece's result lives until the next probe call: one more pass, untimed, checks every piece as it comes out
        res, out_rows, _b, _p = step(check=check_result)
    else:
        check_result(res)
    got = torch.stack([torch.tensor(acc["rows"], device=device), acc["v"], acc["w"]]).to(torch.int64)
    misrouted = acc["misrouted"]
    ok_rows = acc["rows"] == out_rows
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0].item())
        both = torch.cat([want, got, misrouted])
        dist.all_reduce(both, op=dist.ReduceOp.SUM)
        want, got, misrouted = both[:3], both[3:6], both[6:]
    ok = bool((want == got).all().item()) and int(misrouted.item()) == 0 and ok_rows
    # per-kernel HIP-event times recorded on the library stream (crgpu_prof_*); algorithmic bytes per row
    # of each kernel's own input/output
    kern_rows = {"join_build": (nb, 16.0), "join_probe": (npr, 32.0),
                 "radix_build_hist": (nb, 8.0), "radix_build_scatter": (nb, 32.0), "radix_build_split": (nb, 32.0),
                 "radix_build_table": (nb, 40.0), "radix_probe_hist": (npr, 8.0), "radix_probe_scatter": (npr, 32.0),
                 "partition": (nb + npr, 32.0), "exchange": (nb + npr, 32.0)}
    kern = {}
    for name in kern_rows:
        ms, cnt = dev.prof_get(name)
        if cnt:
            kern[name] = (ms, cnt)
    if rank == 0:
        rows_per_step = (nb + npr) * world
        value = rows_per_step * args.steps / dt
        # roofline of the dominant kernel (largest share of the step)
        dom = max(kern, key=lambda k: kern[k][0]) if kern else "join_probe"
        dom_ms, dom_n = kern.get(dom, (0.0, 0))
        rows_l, bpr = kern_rows[dom]
        if world > 1:
            rows_l = rows_l  # per-rank launch: the local share is the same size on average (uniform hash)
        dom_alg = bpr * rows_l
        avg_s = (dom_ms / max(dom_n, 1)) / 1e3
        achieved = dom_alg / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic = None
        if world == 1 and args.log2_build == 27 and args.log2_probe == 29:
            traffic = pmc_traffic(dom, 16.0 * rows_l)  # the kernel's 16 B/row input stream
        line = {
            "metric": "rows/s hash join (build+probe rows per second)", "value": value, "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": f"hash join microbench: int64 keys, 2^{args.log2_build} build x "
                                   f"2^{args.log2_probe} probe rows per GPU, INNER, unique build keys, "
                                   f"{ {'hit100': '100% hit', 'hit50': '50% hit', 'zipf': 'Zipf-like probe keys (theta=1)', 'null1': '1% NULL probe keys', 'sparse': '100% hit, build keys 7919 apart (hashed tables)'}[args.variant]}, "
                                   "output (w, v)",
                       "parallelism": "single GPU" if world == 1 else f"hash repartition all-to-all over {world} GPUs",
                       "result_check": "ok" if ok else "MISMATCH"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_s * 1e3, "launches": dom_n,
                         "algorithmic_bytes_per_launch": dom_alg},
            "kernels_ms_per_step": {k: v[0] / max(args.steps, 1) for k, v in kern.items()},
            "whole_join_hbm_frac": (16.0 * nb + 32.0 * npr) * world / (dt / args.steps) / 1e9 / (HBM_PEAK_GBS * world),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_log2_build, args.cpu_log2_probe)
            # SURVEY.md 8(d) baseline (2): HashRouter + P joiners on the host cores this job may use (16 on the GPU box)
            P = max(2, min(16, len(os.sched_getaffinity(0))))
            line["cpu_baseline_partitioned"] = cpu_baseline_pway(args.cpu_log2_build, args.cpu_log2_probe, P)
    join.close()
    if distributed:
        gb = distributed_group_by(args, torch, dist, dev, cd, _lib, xch_b, world, rank, device)
        for h in [part_b] + parts_p + xchs_p:
            h.close()
        part_b = None
        parts_p, xchs_p = [], []
        del bk, bv, pk, pk_base, pw, res, _b, _p, hit
        torch.cuda.empty_cache()
        _lib.load().crgpu_mem_trim()
        q9 = None
        if not args.no_q9:
            q9 = distributed_q9(args, torch, dist, dev, _lib, xch_b, world, rank, device)
        if rank == 0:
            line["group_by"] = gb
            if q9 is not None:
                line["tpch_q9"] = q9
                ok = ok and q9["result_check"] == "ok"
    if rank == 0:
        if world == 1 and not args.no_other_paths and not args.force_distributed:
            # the other rows of the hot path, measured in the same run (not part of `value`): DESIGN.md 4.3 / 4.4
            line["other_paths"] = other_paths(local_rank)
        print(json.dumps(line), flush=True)
    for h in [part_b] + parts_p + xchs_p + [xch_b]:
        if h is not None:
            h.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)
    return line if rank == 0 else None


def distributed_group_by(args, torch, dist, dev, cd, _lib, xch, world, rank, device):
    """BASELINE.json configs[4] shape on N GPUs: high-cardinality GROUP BY as the reference distributes it
    (distsql_physical_planner.go:2790-2844, physicalplan/aggregator_funcs.go:83-306): local hash aggregation ->
    BY_HASH repartition of the PARTIAL rows on the group key -> all-to-all -> final aggregation (COUNT_ROWS becomes a
    final SUM_INT of the local counts).  Rows (k int64, x int64): every key appears 4 times, twice on this rank and
    twice on the next one, so both stages have work.  Everything stays in HBM and goes through the C-ABI."""
    import time
    n = 1 << args.log2_groupby
    g = torch.Generator(device=device)
    g.manual_seed(77 + rank)
    base = torch.arange(n // 4, device=device, dtype=torch.int64)
    mult = -7046029254386353131  # 0x9E3779B97F4A7C15 as int64: an odd multiplier, a bijection on the wrapped 64-bit ring
    mine = (base * world + rank) * mult
    nxt = (base * world + (rank + 1) % world) * mult
    k = torch.cat([mine, mine, nxt, nxt])[torch.randperm(n, generator=g, device=device)]
    x = torch.ones(n, device=device, dtype=torch.int64)
    torch.cuda.synchronize()
    I64 = cd.INT64
    ANY, SUM_INT, COUNT_ROWS = 0, 11, 14
    part = dev.Partitioner([I64, I64, I64], [0], world)

    def once():
        local = dev.Agg([I64, I64], [0], [(ANY, 0, -1), (SUM_INT, 1, -1), (COUNT_ROWS, -1, -1)])
        local.consume([dev.from_torch(I64, k), dev.from_torch(I64, x)], n)
        partial = dev.result_columns(local.finish())
        regions, counts = part.partition(partial, partial[0].n)
        got, rc = xch.all_to_all(dev.result_columns(regions), counts)
        final = dev.Agg([I64, I64, I64], [0], [(ANY, 0, -1), (SUM_INT, 1, -1), (SUM_INT, 2, -1)])
        rcols = dev.result_columns(got)
        final.consume(rcols, int(rc.sum()))
        out = dev.result_columns(final.finish())
        groups = out[0].n
        s = int(_wrap(torch, out[1].data_ptr, groups, device).sum().item()) if groups else 0
        c = int(_wrap(torch, out[2].data_ptr, groups, device).sum().item()) if groups else 0
        local.close()
        final.close()
        return groups, s, c

    once()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        groups, s, c = once()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / reps
    tot = torch.tensor([float(groups), float(s), float(c), dt], device=device, dtype=torch.float64)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(mx[3].item())
    # every key occurs 4 times in the union of the ranks' inputs: n * world / 4 groups, sum(x) = count = n * world
    ok = int(tot[0].item()) == n * world // 4 and int(tot[1].item()) == n * world and int(tot[2].item()) == n * world
    part.close()
    return {"case": f"two-stage GROUP BY: 2^{args.log2_groupby} rows (k int64, x int64) per GPU, each key 4x, SUM_INT + COUNT_ROWS",
            "rows_per_s": n * world / dt, "ms": dt * 1e3, "groups": int(tot[0].item()), "result_check": "ok" if ok else "MISMATCH"}


def distributed_q9(args, torch, dist, dev, _lib, xch, world, rank, device):
    """BASELINE.json configs[3]: TPC-H Q9 at SF100 over N GPUs (workloads/tpch_dist.py): every rank holds a contiguous
    shard of every table; the filtered part, supplier and nation are mirrored (crgpu_exchange_broadcast), partsupp x
    green, lineitem (semi-reduced by the green parts), the join result and orders are routed BY_HASH
    (crgpu_dev_partition + crgpu_exchange_all_to_all), the GROUP BY runs in two stages.  Every rank generates the same
    synthetic tables from one seed and keeps its shard.  The union of the ranks' groups is checked against the
    independent dense evaluation of tests/tpch_check.py (175 groups, tpch.go:318-336)."""
    import time
    from tests import tpch_check as K
    from workloads import tpch_dist as Q, tpch_plans as P, tpch_synth as T
    sf = args.q9_sf or (100.0 if world > 1 else 10.0)
    t, rows = T.generate(sf, device, 42)
    torch.cuda.synchronize()
    nat, year, profit = K.q9_expected(t, rows["_aux"])
    want = sorted(zip([T.NATIONS[i] for i in nat.tolist()], [float(y) for y in year.tolist()], profit.tolist()))
    shard = Q.shard_tables(t, rank, world)
    if world > 1:  # keep the shard only
        shard = {name: {k: (T.BytesCol(v.data.clone(), v.offsets.clone()) if isinstance(v, T.BytesCol) else v.clone())
                        for k, v in cols.items()} for name, cols in shard.items()}
    del t, nat, year, profit
    rows.pop("_aux")
    torch.cuda.empty_cache()
    fabric = Q.RcclFabric(xch)

    def once():
        ex = P.GpuExec()
        res = Q.run(Q.q9_dist(ex, {k: ex.table(v) for k, v in shard.items()}), fabric)
        out = list(zip(*[v.to_values() for v in ex.to_host(res)])) if res[0].n else []
        ex.close()
        fabric.close()
        return out

    mine = once()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    reps = 3
    dev.prof_reset()
    dev.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / reps
    dev.prof_enable(False)
    kern = {}
    for f in ("exchange", "partition", "join_probe", "join_build", "agg", "expr_select", "expr_project", "deselect",
              "radix_build_scatter", "radix_build_split", "radix_build_table", "radix_probe_scatter"):
        ms, cnt = dev.prof_get(f)
        if cnt:
            kern[f] = round(ms / reps, 3)
    everyone = [mine]
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0].item())
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
    keys = [set(r[:2] for r in part) for part in everyone]
    disjoint = sum(len(k) for k in keys) == len(set().union(*keys))
    got = sorted(sum(everyone, []))
    ok = disjoint and len(got) == len(want) and all(
        g[:2] == w[:2] and abs(g[2] - w[2]) <= 1e-9 * abs(w[2]) + 1e-6 for g, w in zip(got, want))
    in_rows = sum(rows[k] for k in Q.q9_scanned_tables())
    return {"case": f"TPC-H Q9 at SF{sf:g}, synthetic tables sharded over {world} GPU(s): mirrored part(green)/supplier/nation, "
                    "BY_HASH exchange of partsupp x green, lineitem (semi-reduced), the join result and orders, two-stage GROUP BY",
            "rows": in_rows, "ms": dt * 1e3, "rows_per_s": in_rows / dt, "groups": len(got), "groups_per_rank": [len(p) for p in everyone],
            "kernels_ms_rank0": kern, "result_check": "ok" if ok else "MISMATCH"}


def other_paths(device_index):
    """Hash aggregation (Q1 shape, Q1 in full with its WHERE and projections, 1024 and 2^20 groups, all-distinct = config G shape; 2^27 rows) and KV fetch decode
    (lineitem-shaped rows, 10 MiB batch responses), inputs resident in HBM, each with the single-threaded CPU port
    timed on a bounded sample; then SURVEY.md 8(d)'s sub-measurements: hash only, build only, the router's repartition
    and the config G variants.  scripts/bench_agg.py, bench_decode.py and bench_misc.py are the standalone forms."""
    import importlib.util

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    out = {}
    try:
        agg = load("bench_agg")
        out["hash_aggregation"] = [
            {k: r[k] for k in ("case", "rows", "groups", "ms", "rows_per_s", "frac_of_8TBps", "cpu_baseline")}
            for r in agg.measure(27, ["q1", "q1full", "q1fused", "1024", str(1 << 20), str(1 << 27)], reps=3, cpu_sample=1 << 21, device_index=device_index)]
        dec = load("bench_decode").measure(1 << 21, 4, 10 << 20, device_index)
        out["kv_fetch_decode"] = {k: dec[k] for k in ("rows", "input_bytes", "wall_ms", "kv_walk_ms", "kv_decode_ms", "rows_per_s",
                                                      "input_GBps_wall", "q1_pipeline_decode_plus_agg_ms", "q1_pipeline_rows_per_s",
                                                      "q1_pipeline_checked", "cpu_baseline")}
        import torch
        torch.cuda.empty_cache()
        _lib_trim()
        # BASELINE.json configs[2]: TPC-H Q3 / Q9 multi-way hash joins + aggregation at SF10 (scripts/bench_tpch.py)
        out["tpch_q3_q9_sf10"] = [
            {k: r[k] for k in ("case", "rows", "groups", "ms", "rows_per_s", "algorithmic_bytes", "frac_of_8TBps", "kernels_ms",
                               "result_check", "cpu_baseline")}
            for r in load("bench_tpch").measure(10.0, reps=3, cpu_sf=5.0, device_index=device_index)]
        torch.cuda.empty_cache()
        _lib_trim()
        # north_star's sizes on one GPU: Q1 in full on 6e8 rows, Q9 at SF100 (scripts/bench_sf100.py); the Q1 line's
        # cpu_baseline is BASELINE.json configs[0] (SF1 Q1 on the CPU)
        out["tpch_sf100_one_gpu"] = load("bench_sf100").measure(device_index, reps=3, cpu=True, q9_sf=100.0)
        torch.cuda.empty_cache()
        _lib_trim()
        misc = load("bench_misc")
        out["sub_kernels_and_variants"] = [
            {k: r[k] for k in ("case", "rows", "ms", "rows_per_s", "frac_of_8TBps", "kernels", "cpu_baseline")}
            for r in misc.measure(None, reps=3, cpu_sample=1 << 21, device_index=device_index)]
    except Exception as e:  # the join line must not be lost to a failure in the side measurements
        out["error"] = repr(e)
    return out


def _lib_trim():
    """Return the library's idle pooled device memory to the driver (the side measurements each need most of HBM)."""
    from cockroach_amd import _lib
    _lib.load().crgpu_mem_trim()


def _wrap(torch, ptr, n, device):
    """torch view (no copy) of a library-owned int64 device buffer."""
    import ctypes as C

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i8", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(h, device=device)


if __name__ == "__main__":
    main()
