#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_s20t
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_s20t -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_s20t.log 2>&1
python $R/tools/trace_passes.py $R/gpurun_out/prof_s20t 2 > $R/gpurun_out/trace_steps20.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prof_s20t.log | head -1 >> $R/gpurun_out/trace_steps20.txt
head -1 $R/gpurun_out/prof_s20t/*/*kernel_trace.csv >> $R/gpurun_out/trace_steps20.txt
python $R/tools/trace_timeline.py $R/gpurun_out/prof_s20t 2 2 >> $R/gpurun_out/trace_steps20.txt 2>&1
python $R/tools/trace_timeline.py $R/gpurun_out/prof_s20t 2 1 >> $R/gpurun_out/trace_steps20.txt 2>&1
rm -rf $R/gpurun_out/prof_s20t
