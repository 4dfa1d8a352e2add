#!/bin/bash
# rocprofv3 evidence for every published number of a round, one tag per measured path (each tag: one kernel-trace pass
# and one --pmc pass per counter group; see prof_pmc.sh / prof_pmc_cmd.sh).  Writes gpurun_out/prof_<tag>/summary.txt and
# the bench JSON lines; copy what is to be judged into profiles/rNN/.
# Usage (on the GPU box, from the repo root): scripts/collect_profiles.sh [tags...]   (default: all)
set -u
tags=${*:-"J J_sparse agg_q1 tpch decode router sf100 dist"}
for t in $tags; do
  echo "== $t"; date +%T
  case $t in
    J)        scripts/prof_pmc.sh J || exit 1 ;;
    J_sparse) scripts/prof_pmc.sh J_sparse --variant sparse || exit 1 ;;
    agg_q1)   scripts/prof_pmc_cmd.sh agg_q1 scripts/bench_agg.py 27 q1,q1fused 3 0 || exit 1 ;;
    tpch)     scripts/prof_pmc_cmd.sh tpch scripts/bench_tpch.py --sf 10 --reps 2 --cpu-sf 0 --no-check || exit 1 ;;
    decode)   scripts/prof_pmc_cmd.sh decode scripts/bench_decode.py 2097152 4 || exit 1 ;;
    router)   scripts/prof_pmc_cmd.sh router scripts/bench_misc.py router 3 0 || exit 1 ;;
    sf100)    export TMPDIR=/tmp; mkdir -p gpurun_out/prof_sf100
              timeout -s KILL 280 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sf100/kt -o kt --output-format csv -- python3 scripts/bench_sf100.py --reps 3 --no-cpu > gpurun_out/prof_sf100/kt.log 2>&1 || { tail -5 gpurun_out/prof_sf100/kt.log; exit 1; }
              python3 scripts/summarize_pmc.py gpurun_out/prof_sf100 > gpurun_out/prof_sf100/summary.txt
              grep '^{' gpurun_out/prof_sf100/kt.log > gpurun_out/prof_sf100/bench_sf100_under_profiler.jsonl ;;
    dist)     timeout -k 10 280 python3 bench.py --force-distributed --steps 5 --warmup 2 > gpurun_out/bench_force_distributed.json 2> gpurun_out/bench_force_distributed.err || { tail -5 gpurun_out/bench_force_distributed.err; exit 1; } ;;
  esac
done
date +%T
