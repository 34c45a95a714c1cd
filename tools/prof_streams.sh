#!/bin/bash
# Developer aid: kernel trace of the bench with several clouds in flight.
S=${1:-4}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_streams -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 2 --no-cpu-baseline --streams $S > $GRAFT_REPO_ROOT/gpurun_out/prof_streams.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/prof_streams.log | cut -c1-300
