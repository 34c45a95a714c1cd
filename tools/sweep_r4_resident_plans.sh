#!/bin/bash
# GPU box: the driver's 20 steps (inputs resident) under explicit launch-set plans, large set first
R=$GRAFT_REPO_ROOT
cd $R
for p in "" "14,6" "16,4" "12,8" "15,5" "17,3" "13,7" ""; do
  echo "== plan [$p]"
  ST_BENCH_PLAN="$p" ST_BENCH_MIN_UPTIME_S=15 timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | python tools/show_bench.py /dev/stdin | head -1
done
