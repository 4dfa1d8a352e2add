#!/usr/bin/env python3
"""Diagnostic: all-distinct int64 keys through the device aggregator; reports group count, COUNT_ROWS total and,
when groups are missing, which keys / input positions were lost or merged."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cockroach_amd import _lib, coldata as cd, device as dev  # noqa: E402


def main():
    _lib.check(_lib.load().crgpu_init(0))
    dv = torch.device("cuda", 0)
    for lg in [int(x) for x in sys.argv[1:]] or [22, 24, 25, 26, 27]:
        n = 1 << lg
        g = torch.Generator(device=dv); g.manual_seed(1)
        k = torch.randperm(n, generator=g, device=dv, dtype=torch.int64)
        assert int(torch.unique(k).numel()) == n
        a = dev.Agg([cd.INT64], [0], [(0, 0, -1), (14, -1, -1)])
        a.consume([dev.from_torch(cd.INT64, k)], n)
        res = a.finish()
        vecs = dev.result_to_vecs(res)
        keys, cnt = vecs[0].data, vecs[1].data
        print(f"lg={lg} groups={int(res.n)} expected={n} count_total={int(cnt.sum())} max_count={int(cnt.max())}", flush=True)
        if int(res.n) != n or int(cnt.sum()) != n:
            seen = np.zeros(n, dtype=np.uint8)
            seen[keys] = 1
            missing = np.nonzero(seen == 0)[0]
            print("  missing keys:", missing.size, missing[:16])
            kh = k.cpu().numpy()
            pos = np.empty(n, dtype=np.int64)
            pos[kh] = np.arange(n)
            mp = np.sort(pos[missing])
            print("  input positions of missing keys:", mp[:16], "...", mp[-4:])
            print("  position deltas:", np.unique(np.diff(mp))[:16])
            dup = keys[cnt > 1]
            print("  keys with count>1:", dup.size, dup[:16], "positions", np.sort(pos[dup])[:16])
        a.close()
        del k


if __name__ == "__main__":
    main()
