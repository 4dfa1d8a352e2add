#!/bin/bash
# Rebuild agg.hip on the GPU box with an extra -D flag (timing experiments only) and run one bench_agg selection.
set -u
base="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -Wno-unused-result"
sel=$1; shift
for flag in "$@"; do
  touch cockroach_amd/csrc/agg.hip
  extra=""; [ "$flag" != "none" ] && extra="-D$flag"
  make -s -C cockroach_amd/csrc CXXFLAGS="$base $extra" > gpurun_out/sweep_build.log 2>&1 || { echo "build failed for $flag"; tail -5 gpurun_out/sweep_build.log; exit 1; }
  echo "== $flag"
  timeout -k 10 200 python3 -u scripts/bench_agg.py 27 $sel 3 0 2>/dev/null | cut -c1-700 || exit 1
done
