#!/bin/bash
# rocprofv3 evidence for any python command: one kernel-trace pass and one --pmc pass per counter group (never combined
# with other trace domains; each pass a run of its own).  Usage: scripts/prof_pmc_cmd.sh <tag> <script.py> [args...]
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 "$@" > "$out/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$out/kt.log"; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -s KILL 240 rocprofv3 --pmc $grp -d "$out/$name" -o pmc --output-format csv -- python3 "$@" > "$out/$name.log" 2>&1 || { echo "pmc pass $grp failed"; tail -5 "$out/$name.log"; exit 1; }
done
python3 scripts/summarize_pmc.py "$out" > "$out/summary.txt"
cat "$out/summary.txt"
