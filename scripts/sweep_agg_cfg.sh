#!/bin/bash
# Rebuild agg.hip with different (rows per thread, op group) settings on the GPU box and time the low-cardinality cases.
set -u
base="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -Wno-unused-result"
for cfg in "$@"; do
  k=${cfg%%,*}; rest=${cfg#*,}; g=${rest%%,*}; extra=""; [ "$rest" != "$g" ] && extra="-D${rest#*,}"
  touch cockroach_amd/csrc/agg.hip
  make -s -C cockroach_amd/csrc CXXFLAGS="$base -DCRGPU_ACC_ROWS=$k -DCRGPU_OP_GROUP=$g $extra" > gpurun_out/sweep_build.log 2>&1 || { echo "build failed for $cfg"; tail -5 gpurun_out/sweep_build.log; exit 1; }
  echo "== rows/thread=$k op group=$g $extra"
  timeout -k 10 120 python3 -u scripts/probe_agg_cost.py 2>/dev/null | grep fused || exit 1
done
