#!/usr/bin/env python3
"""Where the wall time of TPC-H Q9 at SF10 goes on the host: every executor call timed with and without a device
synchronisation after it (the difference is what the call left queued)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from cockroach_amd import _lib
    from workloads import tpch_plans as P, tpch_synth as T
    sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    _lib.check(_lib.load().crgpu_init(0))
    t, rows = T.generate(sf, "cuda:0", 42)
    torch.cuda.synchronize()
    lib = _lib.load()

    class Timed(P.GpuExec):
        log = []

    def wrap(name, sync):
        fn = getattr(P.GpuExec, name)

        def inner(self, *a, **k):
            t0 = time.perf_counter()
            r = fn(self, *a, **k)
            t1 = time.perf_counter()
            if sync:
                lib.crgpu_dev_synchronize()
            t2 = time.perf_counter()
            Timed.log.append((name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, getattr(r[0], "n", None) if isinstance(r, list) else getattr(r, "n", None)))
            return r
        return inner

    for sync in (False, True):
        for name in ("select_bytes", "select_i64", "filter", "join", "project", "extract_year", "agg"):
            setattr(Timed, name, wrap(name, sync))
        for rep in range(3):
            Timed.log = []
            ex = Timed()
            t0 = time.perf_counter()
            tabs = {k: ex.table(v) for k, v in t.items()}
            t1 = time.perf_counter()
            (P.q3 if (len(sys.argv) > 2 and sys.argv[2] == "q3") else P.q9)(ex, tabs)
            lib.crgpu_dev_synchronize()
            t2 = time.perf_counter()
            ex.close()
            t3 = time.perf_counter()
        print(f"sync_after_each_call={sync}: tables {1e3*(t1-t0):.3f} ms, plan {1e3*(t2-t1):.3f} ms, close {1e3*(t3-t2):.3f} ms")
        for name, ms, tail, n in Timed.log:
            print(f"   {name:14s} call {ms:7.3f} ms  + queued {tail:6.3f} ms  rows out {n}")


if __name__ == "__main__":
    main()
