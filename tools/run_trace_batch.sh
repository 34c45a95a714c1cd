#!/bin/bash
# kernel trace of single-stream runs folded per launch set: a batch of 24 clouds and one cloud alone (tools/trace_one_batch.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() {  # clouds per launch set, steps, which launch set from the end
rm -rf $R/gpurun_out/prof_tb$1
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tb$1 -- python $R/bench.py --streams 1 --steps $2 --warmup $2 --batch $1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_tb$1.log 2>&1
python $R/tools/trace_one_batch.py $R/gpurun_out/prof_tb$1 $3 > $R/gpurun_out/trace_batch$1.txt 2>&1
rm -rf $R/gpurun_out/prof_tb$1
}
run 24 48 4
run 1 16 44
