#!/usr/bin/env python3
"""Diagnostic: cost of the fused aggregation pass as a function of the key shape and the aggregation list."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cockroach_amd import _lib, coldata as cd, device as dev  # noqa: E402

def run(name, cols, types, group_cols, aggs, n, reps=3):
    torch.cuda.synchronize()
    a = dev.Agg(types, group_cols, aggs)
    a.consume(cols, n); a.finish(); a.reset()
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(reps):
        a.consume(cols, n); a.finish(); a.reset()
    _lib.load().crgpu_dev_synchronize()
    ms, cnt = dev.prof_get("agg_fused_lds")
    print("%-60s fused %.3f ms" % (name, ms / reps), flush=True)
    a.close()

def main():
    n = 1 << 27
    _lib.check(_lib.load().crgpu_init(0))
    dv = torch.device("cuda", 0)
    g = torch.Generator(device=dv); g.manual_seed(1)
    rf = torch.randint(0, 3, (n,), generator=g, device=dv, dtype=torch.uint8) + 65
    ls = torch.randint(0, 2, (n,), generator=g, device=dv, dtype=torch.uint8) + 70
    offs = torch.arange(n + 1, device=dv, dtype=torch.int32)
    f = [torch.rand(n, generator=g, device=dv, dtype=torch.float64) for _ in range(4)]
    k6 = torch.randint(0, 6, (n,), generator=g, device=dv, dtype=torch.int64)
    k1024 = torch.randint(0, 1024, (n,), generator=g, device=dv, dtype=torch.int64)
    bcols = [dev.from_torch(cd.BYTES, rf, offsets=offs), dev.from_torch(cd.BYTES, ls, offsets=offs)] + [dev.from_torch(cd.FLOAT64, x) for x in f]
    btypes = [cd.BYTES, cd.BYTES] + [cd.FLOAT64] * 4
    for kname, kc in (("int64 key 6 groups", k6), ("int64 key 1024 groups", k1024)):
        icols = [dev.from_torch(cd.INT64, kc)] + [dev.from_torch(cd.FLOAT64, x) for x in f]
        itypes = [cd.INT64] + [cd.FLOAT64] * 4
        run(kname + ": COUNT_ROWS", icols, itypes, [0], [(14, -1, -1)], n)
        run(kname + ": 1 SUM", icols, itypes, [0], [(10, 1, -1)], n)
        run(kname + ": 4 SUM", icols, itypes, [0], [(10, 1, -1), (10, 2, -1), (10, 3, -1), (10, 4, -1)], n)
        run(kname + ": 4 SUM + 3 AVG + COUNT_ROWS", icols, itypes, [0], [(10, 1, -1), (10, 2, -1), (10, 3, -1), (10, 4, -1), (1, 1, -1), (1, 2, -1), (1, 3, -1), (14, -1, -1)], n)
    run("2 bytes keys: COUNT_ROWS", bcols, btypes, [0, 1], [(14, -1, -1)], n)
    run("2 bytes keys: 1 SUM", bcols, btypes, [0, 1], [(10, 2, -1)], n)
    run("2 bytes keys: 4 SUM", bcols, btypes, [0, 1], [(10, 2, -1), (10, 3, -1), (10, 4, -1), (10, 5, -1)], n)
    run("1 bytes key: COUNT_ROWS", bcols, btypes, [0], [(14, -1, -1)], n)

if __name__ == "__main__":
    main()
