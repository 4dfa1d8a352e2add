#!/usr/bin/env python3
"""Kernel statistics out of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes
NAME_results.db on this ROCm): per kernel calls / total / average duration, like the CSV form of --stats.
usage: rocpd_stats.py results.db [csv-out]"""
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    return [(r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total) for r in rows]


if __name__ == "__main__":
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"\n')
    for r in stats(sys.argv[1]):
        out.write('"%s",%d,%d,%.1f,%d,%d,%.2f\n' % r)
