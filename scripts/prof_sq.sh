#!/bin/bash
# SQ instruction-mix counters for bench.py kernels (one --pmc pass, <= 8 SQ counters)
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args="--steps 1 --warmup 1 --no-cpu-baseline --no-other-paths $*"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d "$out/SQ" -o pmc --output-format csv -- python3 bench.py $args > "$out/SQ.log" 2>&1 || { echo "SQ pass failed"; tail -5 "$out/SQ.log"; exit 1; }
python3 scripts/summarize_pmc.py "$out" > "$out/summary_sq.txt"
grep -E "probe_kernel|scatter_kernel" "$out/summary_sq.txt"
