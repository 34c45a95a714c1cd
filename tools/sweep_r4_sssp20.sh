#!/bin/bash
# GPU box: SSSP frontier knobs under the driver's command (one launch set of 20 clouds)
R=$GRAFT_REPO_ROOT
cd $R
for p in "" "6=8" "6=6" "6=2" "10=4096" "10=1024" "10=512" "8=64" "8=16" "13=0" ""; do
  echo "== params [$p]"
  ST_SKELETON_PARAMS="$p" ST_BENCH_MIN_UPTIME_S=15 timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | python tools/show_bench.py /dev/stdin | head -1
done
