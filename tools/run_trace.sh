R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_trace
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_trace -- python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_trace.log 2>&1
cd $R; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_trace.log | head -1; python tools/trace_gaps.py gpurun_out/prof_trace
