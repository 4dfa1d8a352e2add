#!/usr/bin/env python3
"""PCIe-inclusive rate of the operator boundary (layer 2, include/crgpu.h: crgpu_hj_*): host coldata batches of 1024
rows in, caller-owned 1024-row host batches out -- what a Go shell behind colexecop.Operator.Next() would drive.
Never bench.py's `value` (that is layer 1 with inputs resident in HBM); DESIGN.md section 2 quotes this number.
Batches are marshalled before the timed region (cgo passes pointers, it does not copy)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cockroach_amd import _lib, coldata as cd, operators as ops  # noqa: E402


def batches(cols, bs):
    n = len(cols[0])
    out = []
    for s in range(0, n, bs):
        e = min(n, s + bs)
        out.append(ops._marshal_batch(cd.Batch([cd.Vec(cd.INT64, c[s:e], None) for c in cols], e - s)))
    return out


def main():
    lg_b = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    lg_p = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    bs = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    lookahead = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    lib = _lib.load()
    _lib.check(lib.crgpu_init(0))
    rng = np.random.default_rng(42)
    nb, npr = 1 << lg_b, 1 << lg_p
    bk = rng.permutation(nb).astype(np.int64)
    bv = bk * 3 + 1
    pk = rng.integers(0, nb, size=npr, dtype=np.int64)
    pw = np.arange(npr, dtype=np.int64)
    build, probe = batches([bk, bv], bs), batches([pk, pw], bs)
    I64 = cd.INT64
    keep = (ops._i32([I64, I64]), ops._i32([I64, I64]), ops._i32([0]), ops._i32([0]), ops._i32([1, 3]))
    spec = _lib.JoinSpec(0, 1, 2, 2, keep[0], keep[1], 1, 2, keep[2], keep[3], keep[4])
    h = C.c_void_p()
    _lib.check(lib.crgpu_hj_new(C.byref(spec), C.byref(h)))
    out = ops._OutBatch([I64, I64], bs)
    res = {}
    for rep in range(2):  # first pass warms allocations
        if rep:
            _lib.check(lib.crgpu_hj_reset(h))
        t0 = time.perf_counter()
        for mb, _k in build:
            _lib.check(lib.crgpu_hj_build_append(h, C.byref(mb)))
        _lib.check(lib.crgpu_hj_build_finish(h))
        t1 = time.perf_counter()
        rows = 0
        for i, (mb, _k) in enumerate(probe):
            _lib.check(lib.crgpu_hj_probe_append(h, C.byref(mb)))
            if (i + 1) % lookahead:
                continue  # the shell reads ahead: one device round trip per `lookahead` batches
            while True:
                _lib.check(lib.crgpu_hj_next_output(h, C.byref(out.b)))
                if out.b.length == 0:
                    break
                rows += out.b.length
        _lib.check(lib.crgpu_hj_probe_finish(h))
        while True:
            _lib.check(lib.crgpu_hj_next_output(h, C.byref(out.b)))
            if out.b.length == 0:
                break
            rows += out.b.length
        t2 = time.perf_counter()
        assert rows == npr, (rows, npr)
        res = {"build_s": t1 - t0, "probe_s": t2 - t1}
    lib.crgpu_hj_free(h)
    total = res["build_s"] + res["probe_s"]
    print(json.dumps({"lookahead_batches": lookahead, "case": f"layer 2 (host coldata batches of {bs} rows in and out, PCIe included): INNER join 2^{lg_b} build x 2^{lg_p} probe, int64 key + int64 payload",
                      "rows_per_s": (nb + npr) / total, "build_rows_per_s": nb / res["build_s"], "probe_rows_per_s": npr / res["probe_s"],
                      "host_GBps_in_plus_out": (16.0 * nb + 32.0 * npr) / total / 1e9, **res}))


if __name__ == "__main__":
    main()
