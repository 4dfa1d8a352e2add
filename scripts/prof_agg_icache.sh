#!/bin/bash
# instruction / scalar cache counters of the aggregation kernels (one --pmc pass)
set -u
tag=$1; cases=$2; lg=${3:-27}
out=gpurun_out/prof_agg_$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES -d "$out/IC" -o pmc --output-format csv -- python3 scripts/bench_agg.py $lg $cases 2 0 > "$out/IC.log" 2>&1 || { echo "IC pass failed"; tail -5 "$out/IC.log"; exit 1; }
python3 scripts/summarize_pmc.py "$out" > "$out/summary_ic.txt"
grep -E "accum_kernel" "$out/summary_ic.txt"
