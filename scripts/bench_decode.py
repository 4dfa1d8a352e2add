#!/usr/bin/env python3
"""Throughput of crgpu_dev_decode_kv on lineitem-shaped rows (SURVEY.md 8(d), config Q1 point (ii)):
rows encoded by the oracle's encoder (test infrastructure, outside the timed region), batch responses
resident in HBM, the Q1 columns fetched.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cockroach_amd import _lib, coldata as cd, device as dev  # noqa: E402
from oracle import oracle as O  # noqa: E402


def measure(n=1 << 22, reps=8, batch_bytes=10 << 20, device_index=0):
    _lib.check(_lib.load().crgpu_init(device_index))
    rng = np.random.default_rng(3)
    kt, kd = [cd.INT64, cd.INT64], [0, 0]
    vt = [cd.INT64, cd.INT64, cd.FLOAT64, cd.FLOAT64, cd.FLOAT64, cd.FLOAT64, cd.BYTES, cd.BYTES, cd.INT64, cd.INT64,
          cd.INT64, cd.BYTES, cd.BYTES, cd.BYTES]
    widths = {6: 1, 7: 1, 11: 25, 12: 10, 13: 44}
    ids = list(range(3, 17))
    keys = [cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64) // 4 + 1),
            cd.Vec.from_numpy(cd.INT64, np.arange(n, dtype=np.int64) % 4 + 1)]
    vals = []
    for i, t in enumerate(vt):
        if t == cd.INT64:
            vals.append(cd.Vec.from_numpy(cd.INT64, rng.integers(0, 200_000, n, dtype=np.int64)))
        elif t == cd.FLOAT64:
            vals.append(cd.Vec.from_numpy(cd.FLOAT64, np.round(rng.random(n) * 1e5, 2)))
        elif i == 6:  # l_returnflag in {A, N, R}
            vals.append(cd.Vec.bytes_from_fixed(np.array([65, 78, 82], np.uint8)[rng.integers(0, 3, (n, 1))]))
        elif i == 7:  # l_linestatus in {F, O}
            vals.append(cd.Vec.bytes_from_fixed(np.array([70, 79], np.uint8)[rng.integers(0, 2, (n, 1))]))
        else:
            vals.append(cd.Vec.bytes_from_fixed(rng.integers(65, 91, (n, widths[i])).astype(np.uint8)))
    t = O.KvTable(110, 1, kt, kd, ids, vt)
    data, offs = O.kv_encode(t, keys, vals, max_batch_bytes=batch_bytes)
    one = len(data)
    # replicate the encoded batches to get enough concurrent batch walks
    data = np.tile(np.concatenate([data, np.zeros((-one) % 16, np.uint8)]), reps)
    step = one + (-one) % 16
    all_offs = np.concatenate([np.concatenate([offs[:-1], [offs[-1]]])[None, :] + r * step for r in range(reps)], 0)
    batches = []
    for r in range(reps):
        for b in range(len(offs) - 1):
            batches.append((all_offs[r, b], all_offs[r, b + 1]))
    # batches must be contiguous for the API: pad records are not allowed, so decode each replica's range
    # through a flat offsets array where replica boundaries coincide (padding is zero bytes => use one call per replica)
    buf = dev.DeviceBuffer.from_numpy(np.concatenate([data, np.zeros(16, np.uint8)]))
    sel = [2, 3, 4, 5, 6, 7, 8]
    d = dev.Decoder(t.prefix_len(), kt, kd, [-1, -1], [ids[i] for i in sel], [vt[i] for i in sel], list(range(7)),
                    [vt[i] for i in sel])
    lib = _lib.load()

    class Sub:  # a view into the device buffer
        def __init__(self, ptr):
            self.ptr = ptr

    def run():
        rows = 0
        for r in range(reps):
            res = d.decode(Sub(buf.ptr + r * step), offs)
            rows += int(res.n)
        return rows

    run()
    dev.prof_enable(False)  # wall time without the library's profiling hooks; the kernel times from one more pass with them
    lib.crgpu_dev_synchronize()
    t0 = time.perf_counter()
    rows = run()
    lib.crgpu_dev_synchronize()
    dt = time.perf_counter() - t0
    dev.prof_enable(True)
    dev.prof_reset()
    run()
    lib.crgpu_dev_synchronize()
    dev.prof_enable(False)
    walk_ms, _ = dev.prof_get("kv_walk")
    dec_ms, _ = dev.prof_get("kv_decode")
    in_bytes = one * reps
    # CPU baseline: the oracle's cFetcher restatement (single thread) on one replica of the same batches
    spec = (t.prefix_len(), kt, kd, [-1, -1], [ids[i] for i in sel], [vt[i] for i in sel], list(range(7)), [vt[i] for i in sel])
    one_data = data[:one]
    c0 = time.perf_counter()
    cpu_out = O.kv_decode(*spec, one_data, offs)
    cdt = time.perf_counter() - c0
    cpu = {"value": len(cpu_out[0]) / cdt, "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": f"one replica ({one} bytes, {len(cpu_out[0])} rows) of the same batches (oracle/crdb_kv.c, 1 thread)"}
    out_bytes = rows * (4 * 8 + 2 * (1 + 4) + 8)
    # Q1 point (ii) as a device-resident pipeline: decode -> hash aggregation of the decoded columns (GROUP BY
    # l_returnflag, l_linestatus; 4 SUM + 3 AVG + COUNT_ROWS; no filter / projections: those are the next row, N2)
    ftypes = [vt[i] for i in sel]
    aggs = [(0, 4, -1), (0, 5, -1), (10, 0, -1), (10, 1, -1), (10, 2, -1), (10, 3, -1), (1, 0, -1), (1, 1, -1), (1, 2, -1), (14, -1, -1)]
    a = dev.Agg(ftypes, [4, 5], aggs)

    def pipeline(nrep):
        a.reset()
        for r in range(nrep):
            res = d.decode(Sub(buf.ptr + r * step), offs)
            a.consume(dev.result_columns(res), int(res.n))
        return dev.result_to_vecs(a.finish())

    got = pipeline(1)
    want = O.hash_agg(cpu_out, [4, 5], aggs)
    gk = sorted(zip(*[v.to_values() for v in got]))
    wk = sorted(zip(*[v.to_values() for v in want]))
    assert len(gk) == len(wk) and all(x[:2] == y[:2] and x[9] == y[9] and np.allclose(x[2:9], y[2:9], rtol=1e-9) for x, y in zip(gk, wk)), \
        "decode -> aggregate pipeline differs from the oracle"
    pipeline(reps)
    lib.crgpu_dev_synchronize()
    p0 = time.perf_counter()
    pipeline(reps)
    lib.crgpu_dev_synchronize()
    pdt = time.perf_counter() - p0
    a.close()
    d.close()
    return ({"rows": rows, "input_bytes": in_bytes, "bytes_per_row": in_bytes / rows, "batches_per_call": len(offs) - 1,
                      "wall_ms": dt * 1e3, "kv_walk_ms": walk_ms, "kv_decode_ms": dec_ms,
                      "rows_per_s": rows / dt, "input_GBps_wall": in_bytes / dt / 1e9,
                      "decode_kernel_GBps": (in_bytes + out_bytes) / (dec_ms / 1e3) / 1e9 if dec_ms else None,
                      "walk_kernel_GBps": 2 * in_bytes / (walk_ms / 1e3) / 1e9 if walk_ms else None,
                      "q1_pipeline_decode_plus_agg_ms": pdt * 1e3, "q1_pipeline_rows_per_s": rows / pdt, "q1_pipeline_groups": len(gk),
                      "q1_pipeline_checked": "one replica vs oracle kv_decode -> hash_agg: keys and counts exact, sums rel 1e-9",
                      "cpu_baseline": cpu})


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    batch_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 10 << 20
    print(json.dumps(measure(n, reps, batch_bytes)))


if __name__ == "__main__":
    main()
