#!/usr/bin/env python3
"""Busy / idle time of the GPU from a rocprofv3 --kernel-trace CSV: total kernel time, the idle gaps between consecutive
kernels (a latency-bound chain shows up as many 20-100 us gaps), and the largest gaps with the kernels around them.
Usage: kernel_gaps.py <kernel_trace.csv> [substring of the first kernel of the window] [n windows to skip]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")
gaps = [(rows[i + 1][0] - rows[i][1], rows[i][2][:60], rows[i + 1][2][:60]) for i in range(len(rows) - 1)]
hist = {}
for g, _, _ in gaps:
    b = "<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else ">=1ms"
    hist[b] = hist.get(b, [0, 0]); hist[b][0] += 1; hist[b][1] += max(g, 0)
for b, (c, t) in hist.items():
    print(f"  gaps {b}: {c} totalling {t / 1e6:.3f} ms")
