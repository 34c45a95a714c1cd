#!/bin/bash
# GPU box: tuning code 12 (persistent SSSP, helper workgroups of the branch selection) at bench level, after the key collision with
# the neighbour searches' cell cap was removed (smart_tree_amd/skeleton/tuning.py: until then "12=v" also set that cap to v / 100)
R=$GRAFT_REPO_ROOT
cd $R
run() {  # label, env params, bench flags
  echo "== $1 [$2] $3"
  ST_SKELETON_PARAMS="$2" ST_BENCH_MIN_UPTIME_S=20 timeout 400 python bench.py $3 --no-cpu-baseline --no-extras 2>/dev/null | python tools/show_bench.py /dev/stdin | head -2
}
for rep in 1 2; do
run "20 steps, default (helpers by size: on)" "" "--steps 20"
run "20 steps, no helpers" "12=256" "--steps 20"
run "20 steps, persistent SSSP" "12=1" "--steps 20"
run "20 steps, 10 + 10, persistent SSSP" "12=1" "--steps 20 --batch 10"
done
run "384 steps, default (helpers by size: off)" "" ""
run "384 steps, helpers forced on" "12=41216" ""
run "384 steps, persistent SSSP" "12=1" ""
run "384 steps, default (helpers by size: off)" "" ""
