#!/bin/bash
# Developer aid: kernel statistics of the configs[3]-sized pipeline (5M points, 1 cm voxels).
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 250 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4 -- python $GRAFT_REPO_ROOT/tools/time_fp16.py 5000000 0.01 0.6 > $GRAFT_REPO_ROOT/gpurun_out/prof_c4.log 2>&1
cd $GRAFT_REPO_ROOT; grep fp16= gpurun_out/prof_c4.log
f=$(ls -t gpurun_out/prof_c4/*/*kernel_stats.csv | head -1); head -22 $f | cut -c1-130
