#!/bin/bash
# GPU box: the upload-inclusive rate of the driver's 20 steps under explicit launch-set plans (ST_BENCH_PLAN)
R=$GRAFT_REPO_ROOT
cd $R
for p in "" "4,16" "6,14" "2,18" "3,5,12" "5,15" "8,12"; do
  echo "== plan [$p]"
  ST_BENCH_PLAN="$p" ST_BENCH_MIN_UPTIME_S=15 timeout 300 python bench.py --steps 20 --streams 3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/show_bench.py /dev/stdin | head -2
done
