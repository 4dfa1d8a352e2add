#!/bin/bash
# rocprofv3 evidence for bench.py: one kernel-trace pass and one --pmc pass per counter group
# (never combined with other trace domains).  Usage: scripts/prof_pmc.sh <tag> [bench args...]
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args="--steps 1 --warmup 1 --no-cpu-baseline --no-other-paths $*"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python3 bench.py $args > "$out/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$out/kt.log"; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -s KILL 240 rocprofv3 --pmc $grp -d "$out/$name" -o pmc --output-format csv -- python3 bench.py $args > "$out/$name.log" 2>&1 || { echo "pmc pass $grp failed"; tail -5 "$out/$name.log"; exit 1; }
done
python3 scripts/summarize_pmc.py "$out" > "$out/summary.txt"
cat "$out/summary.txt"
