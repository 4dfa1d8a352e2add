#!/bin/bash
# rocprofv3 passes over scripts/bench_agg.py (one case): kernel trace + SQ counter passes (<= 8 SQ counters each)
set -u
tag=$1; cases=$2; lg=${3:-27}
out=gpurun_out/prof_agg_$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 scripts/bench_agg.py $lg $cases 2 0 > "$out/kt.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$out/kt.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d "$out/SQ1" -o pmc --output-format csv -- python3 scripts/bench_agg.py $lg $cases 2 0 > "$out/SQ1.log" 2>&1 || { echo "SQ1 pass failed"; tail -5 "$out/SQ1.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$out/SQ2" -o pmc --output-format csv -- python3 scripts/bench_agg.py $lg $cases 2 0 > "$out/SQ2.log" 2>&1 || { echo "SQ2 pass failed"; tail -5 "$out/SQ2.log"; }
python3 scripts/summarize_pmc.py "$out" > "$out/summary.txt"
grep -E "agg_" "$out/summary.txt"
