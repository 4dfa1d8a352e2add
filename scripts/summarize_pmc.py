#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by scripts/prof_pmc.sh: per kernel the mean
duration (kernel trace) and the per-launch counter values (pmc passes)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
print(f"# {root}: per-kernel summary (durations from --kernel-trace --stats, counters per launch from --pmc passes)")
for f in glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")
            if "crgpu" in name:
                print(f"duration,{name},calls={row.get('Calls')},avg_ns={row.get('AverageNs')},total_ns={row.get('TotalDurationNs')},pct={row.get('Percentage')}")
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d) or os.path.basename(d) == "kt":
        continue
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if "crgpu" not in name:
                    continue
                acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[name].add(row.get("Dispatch_Id"))
    for name, ctrs in acc.items():
        n = max(len(launches[name]), 1)
        for c, v in ctrs.items():
            print(f"pmc,{name},{c},per_launch={v / n:.1f},launches={n}")
