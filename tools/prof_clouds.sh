#!/bin/bash
# Developer aid: rocprofv3 kernel stats of tools/time_clouds.py (both bench clouds, 13 steps each).
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cc -- python $GRAFT_REPO_ROOT/tools/time_clouds.py > $GRAFT_REPO_ROOT/gpurun_out/prof_cc.log 2>&1
cd $GRAFT_REPO_ROOT; grep seed gpurun_out/prof_cc.log
f=$(ls -t gpurun_out/prof_cc/*/*kernel_stats.csv | head -1); head -24 $f | cut -c1-150
