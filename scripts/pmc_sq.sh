#!/bin/bash
# SQ instruction-mix / wait counters of one python command: two --pmc passes (8 SQ counters each), each a run of its own.
# Usage: scripts/pmc_sq.sh <tag> <script.py> [args...]   -> gpurun_out/sq_<tag>/summary.txt
set -u
tag=$1; shift
out=gpurun_out/sq_$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$out/a" -o pmc --output-format csv -- python3 "$@" > "$out/a.log" 2>&1 || { echo "pass a failed"; tail -5 "$out/a.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM -d "$out/b" -o pmc --output-format csv -- python3 "$@" > "$out/b.log" 2>&1 || { echo "pass b failed"; tail -5 "$out/b.log"; exit 1; }
python3 - "$out" <<'PY' > "$out/summary.txt"
import csv, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in "ab":
    for r in csv.DictReader(open(f"{out}/{d}/pmc_counter_collection.csv")):
        k = r["Kernel_Name"].split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:6]:
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:24s} {x / cnt[(k, c)]:16.0f}  per launch ({cnt[(k, c)]} launches)")
PY
cat "$out/summary.txt"
