#!/bin/bash
# GPU box: the driver's 20 steps under launch-set plans x SSSP form (persistent launch = tuning code 12 bit 0)
R=$GRAFT_REPO_ROOT
cd $R
run() {  # label, env params, bench flags
  echo "== $1"
  ST_SKELETON_PARAMS="$2" ST_BENCH_MIN_UPTIME_S=20 timeout 300 python bench.py --steps 20 $3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/show_bench.py /dev/stdin | head -1
}
for rep in 1 2; do
run "one set of 20" "" ""
run "one set of 20, persistent SSSP" "12=1" ""
run "10 + 10" "" "--batch 10"
run "10 + 10, persistent SSSP" "12=1" "--batch 10"
run "7 + 7 + 6, persistent SSSP" "12=1" "--batch 7 --streams 3"
run "10 + 10, persistent SSSP, no helpers" "12=257" "--batch 10"
done
