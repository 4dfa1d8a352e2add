#!/usr/bin/env python3
"""Diagnostic: one aggregation shape (int64 key, G groups, COUNT_ROWS [+ SUMs]) for rocprofv3 counter passes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cockroach_amd import _lib, coldata as cd, device as dev  # noqa: E402

def main():
    groups = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nsum = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n = 1 << 27
    _lib.check(_lib.load().crgpu_init(0))
    dv = torch.device("cuda", 0)
    g = torch.Generator(device=dv); g.manual_seed(1)
    k = torch.randint(0, groups, (n,), generator=g, device=dv, dtype=torch.int64)
    f = [torch.rand(n, generator=g, device=dv, dtype=torch.float64) for _ in range(max(nsum, 1))]
    cols = [dev.from_torch(cd.INT64, k)] + [dev.from_torch(cd.FLOAT64, x) for x in f]
    types = [cd.INT64] + [cd.FLOAT64] * len(f)
    aggs = [(14, -1, -1)] + [(10, 1 + i, -1) for i in range(nsum)]
    torch.cuda.synchronize()
    a = dev.Agg(types, [0], aggs)
    for _ in range(3):
        a.consume(cols, n); a.finish(); a.reset()
    _lib.load().crgpu_dev_synchronize()
    a.close()

if __name__ == "__main__":
    main()
