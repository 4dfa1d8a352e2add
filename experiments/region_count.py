"""Bulk aggregation (sub form) at row counts just above a power of two: the number of regions follows the row count
(default) or is the next power of two (CRGPU_BULK_REGION_ROWS=0, the form before).  Prints ms per consume+finish."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from cockroach_amd import coldata as cd, device as dev
F = {"ANY_NOT_NULL": 0, "SUM_INT": 11, "COUNT_ROWS": 14}

def run(n, per):
    g = n // per
    k = (torch.randperm(g, device="cuda", dtype=torch.int64) * 1_000_003 - 11).repeat(per)[:n]
    k = k[torch.randperm(k.numel(), device="cuda")].contiguous()
    x = torch.randint(-2**40, 2**40, (k.numel(),), device="cuda", dtype=torch.int64)
    n = k.numel()
    cols = [dev.from_torch(cd.INT64, k), dev.from_torch(cd.INT64, x)]
    aggs = [(F["ANY_NOT_NULL"], 0, -1), (F["SUM_INT"], 1, -1), (F["COUNT_ROWS"], -1, -1)]
    a = dev.Agg([cd.INT64, cd.INT64], [0], aggs)
    best = 1e9
    for it in range(6):
        a.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.consume(cols, n)
        r = a.finish()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    a.close()
    return best * 1e3

for n in [(1 << 26) + (1 << 20), 80_000_000, 100_000_000, (1 << 25) + (1 << 21)]:
    for per in (1, 4):
        print(f"n={n} keys x{per}: {run(n, per):.3f} ms", flush=True)
