#!/usr/bin/env python3
"""Selective join probe (TPC-H Q9 shape): 6e7 probe rows against a 4.3e5-row unique build side, ~5 % of the probe rows
match.  Variants: one / two key columns, few / several emitted probe columns.  Prints ms per probe and the kernel time."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cockroach_amd import _lib, coldata as cd, device as dev  # noqa: E402


def run(two_cols, npay, nb=434_648, npr=60_000_000, reps=3):
    dv = torch.device("cuda", 0)
    g = torch.Generator(device=dv); g.manual_seed(3)
    space = nb * 20
    bk = torch.randperm(space, generator=g, device=dv, dtype=torch.int64)[:nb].contiguous()
    pk = torch.randint(0, space, (npr,), generator=g, device=dv, dtype=torch.int64)
    if two_cols:
        b_cols = [dev.from_torch(cd.INT64, bk // 1000), dev.from_torch(cd.INT64, bk % 1000)]
        p_keys = [dev.from_torch(cd.INT64, pk // 1000), dev.from_torch(cd.INT64, pk % 1000)]
    else:
        b_cols, p_keys = [dev.from_torch(cd.INT64, bk)], [dev.from_torch(cd.INT64, pk)]
    nk = len(b_cols)
    b_cols.append(dev.from_torch(cd.FLOAT64, torch.rand(nb, generator=g, device=dv, dtype=torch.float64)))
    pays = [dev.from_torch(cd.FLOAT64, torch.rand(npr, generator=g, device=dv, dtype=torch.float64)) for _ in range(npay)]
    lt = [cd.INT64] * nk + [cd.FLOAT64] * npay
    rt = [cd.INT64] * nk + [cd.FLOAT64]
    out = list(range(nk, nk + npay)) + [nk + npay + nk]
    torch.cuda.synchronize()  # the inputs are produced on torch's stream, the library runs on its own
    j = dev.Join(0, lt, rt, list(range(nk)), list(range(nk)), True, out_cols=out)
    j.build(b_cols, nb)
    res = j.probe(p_keys + pays, npr)
    _lib.load().crgpu_dev_synchronize()
    dev.prof_enable(True); dev.prof_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = j.probe(p_keys + pays, npr)
    _lib.load().crgpu_dev_synchronize()
    dt = (time.perf_counter() - t0) / reps
    ms, cnt = dev.prof_get("join_probe")
    print(json.dumps({"two_key_cols": two_cols, "probe_payload_cols": npay, "out_rows": int(res.n), "ms": round(dt * 1e3, 3),
                      "kernel_ms": round(ms / max(cnt, 1), 3)}), flush=True)
    j.close()


if __name__ == "__main__":
    _lib.check(_lib.load().crgpu_init(0))
    for _ in range(2):  # twice: the second round runs on recycled allocations
        for two in (False, True):
            for npay in (1, 4):
                run(two, npay)
