#!/bin/bash
# per-seed skeleton-stage times of one cloud alone: default strategy (chip-wide claim for long paths) and the batch strategy
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_seeds.txt; : > $O
cd $R
for S in 0 1 2 3; do
for P in "" "14=1,1=1048576,2=1048576,3=1"; do
echo "== seed $S params [$P]" >> $O
python tools/diag_phases.py 1000000 0.02 0 $S "$P" 2>&1 | grep -E "^params|ticks|phase" | cut -c1-420 >> $O
done
done
